"""The C-ABI shared library: it loads without a GPU, exports exactly what include/trpl.h
declares, validates arguments before touching a device, and fails loudly (no fallback) when no
device is present.  No compute call is made here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "trpl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trpl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(trpl):
    names = header_symbols()
    assert len(names) >= 13
    dll = trpl._abi.lib()
    for n in names:
        assert hasattr(dll, n), n
    assert set(trpl._abi.SIGNATURES) == set(names)       # the binding covers the whole header
    assert dll.trpl_abi_version() == 1


def test_cites_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "trpl.h")).read()
    for cite in ("pvSimPCR.py:309-401", "probs.py:64-85", "probs.py:20-62", "bayeslib.py:117-201",
                 "pvSimPCR.py:42-81"):
        assert cite in text


def test_argument_validation_needs_no_device(trpl):
    lib = trpl._abi.lib()
    z = np.zeros(16)
    pl = np.zeros((1, 11))
    rc = lib.trpl_solve_pl(z.ctypes.data, 1, 100.0, 1.0, 12, 10, 1, 7, 100, z.ctypes.data, pl.ctypes.data, 8, 11,
                           None, None, 0, 0, None)
    assert rc == trpl._abi.ERR_ARG and b"power of two" in lib.trpl_last_error()
    rc = lib.trpl_solve_pl(z.ctypes.data, 1, 100.0, 1.0, 16, 10, 1, 7, 100, z.ctypes.data, pl.ctypes.data, 2, 11,
                           None, None, 0, 0, None)
    assert rc == trpl._abi.ERR_ARG
    rc = lib.trpl_log10_clamp(pl.ctypes.data, 8, 1, 11, 5, 1e-300, 0, None)
    assert rc == trpl._abi.ERR_ARG
    rc = lib.trpl_pcr_solve_batched_dev(None, None, None, None, None, 4, 128, 8, 0, None)
    assert rc == trpl._abi.ERR_ARG
    # empty batches are a successful no-op everywhere
    assert lib.trpl_solve_pl(None, 0, 100.0, 1.0, 16, 10, 1, 7, 100, None, None, 8, 11, None, None, 0, 0, None) == 0
    assert lib.trpl_sse_accumulate(None, None, 8, 0, 5, 5, None, None, 0, None) == 0
    with pytest.raises(trpl.TrplError):
        trpl._abi.check(rc)


def test_no_cpu_fallback_without_a_device(trpl):
    lib = trpl._abi.lib()
    if lib.trpl_device_count() > 0:
        pytest.skip("a GPU is visible; the loud-failure path is exercised on CPU-only hosts")
    X = np.ones((2, 12))
    with pytest.raises(trpl.TrplError) as ei:
        trpl.solve_pl(X, 100.0, 1.0, 16, 10, np.ones(16))
    assert ei.value.code == trpl._abi.ERR_NODEVICE
    with pytest.raises(trpl.TrplError):
        trpl.fastlog(np.ones((2, 3)))


def test_shard_bounds_is_the_rule_the_rank_driver_uses(trpl):
    """trpl_shard_bounds (the sharding of trpl_loglik_multi) == trpl_amd.dist.shard_bounds (ranks)."""
    import ctypes as C
    lib = trpl._abi.lib()
    lo, hi = C.c_int64(), C.c_int64()
    for S in (0, 1, 5, 64, 1000, 65537):
        for n in (1, 2, 3, 8):
            cover = []
            for r in range(n):
                assert lib.trpl_shard_bounds(S, n, r, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == trpl.dist.shard_bounds(S, n, r)
                cover.append((lo.value, hi.value))
            assert cover[0][0] == 0 and cover[-1][1] == S and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert lib.trpl_shard_bounds(10, 2, 2, C.byref(lo), C.byref(hi)) == trpl._abi.ERR_ARG


def test_multi_device_call_fails_loudly_without_a_device(trpl):
    if trpl._abi.lib().trpl_device_count() > 0:
        pytest.skip("a device is present")
    X = np.ones((4, 13))
    with pytest.raises(trpl.TrplError) as ei:
        trpl.loglik(X, np.ones((1, 16)), 100.0, 1.0, 16, 10, [np.zeros(5)], devices="all")
    assert ei.value.code == trpl._abi.ERR_NODEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bayesian-inference-trpl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src and "trpl_oracle" not in src, f
