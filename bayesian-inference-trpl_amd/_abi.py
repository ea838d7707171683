"""ctypes binding of libtrpl_hip.so (the C ABI declared in include/trpl.h).

The library is built in-tree by `make -C bayesian-inference-trpl_amd` (or
`__graft_entry__.build()`).  There is NO fallback: if the shared object is missing or a call
fails, an exception is raised -- the product path never routes through the CPU oracle.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRPL_LIBRARY", os.path.join(_HERE, "libtrpl_hip.so"))   # override: A/B builds

# status codes / flags (include/trpl.h)
OK, ERR_ARG, ERR_HIP, ERR_NODEVICE, ERR_UNSUPPORTED = 0, 1, 2, 3, 4
FLAG_STRICT, FLAG_PL_F32, FLAG_NORMALIZE, FLAG_FP32 = 0x1, 0x2, 0x4, 0x8
KERNEL_FAST, KERNEL_FAST_PAIR, KERNEL_STRICT, KERNEL_FP32 = 0, 1, 2, 3


class TrplError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("trpl error %d: %s" % (code, msg))
        self.code = code


_vp, _i32, _i64, _u32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_double
_pd = C.POINTER(C.c_double)

# name -> argtypes, exactly the prototypes of include/trpl.h
SIGNATURES = {
    "trpl_abi_version": [],
    "trpl_last_error": [],
    "trpl_device_count": [],
    "trpl_solve_pl": [_vp, _i64, _f64, _f64, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _i64, _vp, _vp, _u32,
                      _i32, _pd],
    "trpl_solve_pl_dev": [_vp, _i64, _f64, _f64, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _i64, _vp, _vp,
                          _u32, _vp],
    "trpl_log10_clamp": [_vp, _i32, _i64, _i64, _i64, _f64, _i32, _pd],
    "trpl_log10_clamp_dev": [_vp, _i32, _i64, _i64, _i64, _f64, _vp],
    "trpl_sse_accumulate": [_vp, _vp, _i32, _i64, _i64, _i64, _vp, _vp, _i32, _pd],
    "trpl_sse_accumulate_dev": [_vp, _vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp],
    "trpl_loglik": [_vp, _i64, _i32, _vp, _f64, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                    _vp, _u32, _i32, _pd],
    "trpl_loglik_dev": [_vp, _i64, _i32, _vp, _f64, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp,
                        _vp, _vp, _u32, _vp],
    "trpl_loglik_obs": [_vp, _i64, _i32, _vp, _f64, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp,
                        _vp, _vp, _vp, _u32, _i32, _pd],
    "trpl_loglik_obs_dev": [_vp, _i64, _i32, _vp, _f64, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp,
                            _vp, _vp, _vp, _vp, _u32, _vp],
    "trpl_loglik_from_pl_dev": [_vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _u32, _vp],
    "trpl_loglik_multi": [_vp, _i64, _i32, _vp, _f64, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64,
                          _vp, _vp, _vp, _vp, _vp, _u32, _vp, _i32, _pd],
    "trpl_shard_bounds": [_i64, _i32, _i32, _vp, _vp],
    "trpl_kernel_variant": [_i64, _i32, _i64, _u32],
    "trpl_sample_box": [_u32, _i64, _i32, _vp, _vp, _vp, _u32, _vp, _i32, _pd],
    "trpl_sample_box_dev": [_u32, _i64, _i32, _vp, _vp, _vp, _u32, _vp, _vp],
    "trpl_posterior_workspace_bytes": [_i32],
    "trpl_posterior_weights": [_vp, _i64, _f64, _vp, _vp, _i32, _pd],
    "trpl_posterior_weights_dev": [_vp, _i64, _f64, _vp, _vp, _vp, _i64, _vp],
    "trpl_posterior_moments": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _pd],
    "trpl_posterior_moments_dev": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "trpl_posterior_hist": [_vp, _vp, _vp, _i64, _f64, _f64, _i32, _f64, _f64, _i32, _vp, _i32, _pd],
    "trpl_posterior_hist_dev": [_vp, _vp, _vp, _i64, _f64, _f64, _i32, _f64, _f64, _i32, _vp, _vp],
    "trpl_pcr_solve_batched": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _u32, _i32, _pd],
    "trpl_pcr_solve_batched_dev": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _u32, _vp],
}

_lib = None


def _build_if_missing():
    """A fresh checkout has no shared object (it is git-ignored): build the HIP library in-tree with
    the package Makefile when hipcc is present.  This is a build step, not a fallback -- the result
    is the same gfx950 library; set TRPL_AUTOBUILD=0 to forbid it."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.isfile("/opt/rocm/bin/hipcc") else None)
    if os.environ.get("TRPL_AUTOBUILD", "1") == "0" or hipcc is None or "TRPL_LIBRARY" in os.environ:
        return
    subprocess.check_call(["make", "-s", "-j4", "-C", _HERE, "libtrpl_hip.so", "HIPCC=" + hipcc])


def lib():
    """Load libtrpl_hip.so once; raise loudly if it is absent and cannot be built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            _build_if_missing()
        if not os.path.isfile(LIB_PATH):
            raise ImportError("%s not found: build it with `make -C %s` (hipcc, gfx950); "
                              "there is no CPU fallback" % (LIB_PATH, _HERE))
        dll = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(dll, name)
            fn.argtypes = argtypes
            fn.restype = C.c_char_p if name == "trpl_last_error" else (
                C.c_int64 if name == "trpl_posterior_workspace_bytes" else C.c_int)
        _lib = dll
    return _lib


def check(rc):
    if rc != OK:
        raise TrplError(rc, lib().trpl_last_error().decode("utf-8", "replace"))


def ptr(a):
    """Address of a numpy array / int device pointer / None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return int(a)
