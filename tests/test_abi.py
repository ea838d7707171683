"""The C-ABI shared library: it loads without a GPU, exports exactly what include/trpl.h
declares, validates arguments before touching a device, and fails loudly (no fallback) when no
device is present.  No compute call is made here."""
import ctypes
import os
import re
import sys

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
    assert dll.trpl_abi_version() == 5 == trpl._abi.ABI_VERSION


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


def test_round2_entry_points_validate_without_a_device(trpl):
    lib = trpl._abi.lib()
    A = trpl._abi
    z = np.zeros(16 * 12)
    pl = np.zeros((1, 11))
    steps = np.zeros(17, dtype=np.int64)
    # more snapshots than the kernel argument block holds
    rc = lib.trpl_solve_pl_snap(z.ctypes.data, 1, 100.0, 1.0, 16, 10, 1, 7, 100, z.ctypes.data, pl.ctypes.data, 8, 11,
                                None, None, steps.ctypes.data, 17, None, None, None, 0, 0, None)
    assert rc == A.ERR_ARG and b"n_snap" in lib.trpl_last_error()
    rc = lib.trpl_solve_pl_snap_dev(z.ctypes.data, 1, 100.0, 1.0, 16, 10, 1, 7, 100, z.ctypes.data, pl.ctypes.data, 8,
                                    11, None, None, None, 3, None, None, None, 0, None)
    assert rc == A.ERR_ARG and b"snap_steps" in lib.trpl_last_error()
    # kernel-variant bits: reported choice honours them, contradictory requests are refused at the launch
    assert lib.trpl_kernel_variant(10, 128, 10, A.FLAG_KERNEL_PAIR) == A.KERNEL_FAST_PAIR
    assert lib.trpl_kernel_variant(10 ** 7, 128, 80000, A.FLAG_KERNEL_SINGLE) == A.KERNEL_FAST
    assert lib.trpl_kernel_variant(10 ** 7, 128, 80000, A.FLAG_KERNEL_PAIR | A.FLAG_STRICT) == A.KERNEL_STRICT
    assert lib.trpl_kernel_variant(10 ** 7, 256, 80000, 0) == A.KERNEL_FAST          # the paired kernel is L = 128 only
    assert A.pin_variant(0, 10 ** 7, 128, 80000) == A.FLAG_KERNEL_PAIR
    assert A.pin_variant(0, 30, 128, 80000) == A.FLAG_KERNEL_SINGLE
    assert A.pin_variant(A.FLAG_STRICT, 10 ** 7, 128, 80000) == A.FLAG_STRICT         # nothing to pin
    assert A.pin_variant(A.FLAG_KERNEL_SINGLE, 10 ** 7, 128, 80000) == A.FLAG_KERNEL_SINGLE
    with pytest.raises(ValueError):
        A.kernel_flag("both")
    # the inverse of trpl_shard_bounds
    for S in (1, 5, 64, 1000, 65537):
        for n in (1, 2, 3, 7, 8):
            for s in {0, S // 3, S // 2, S - 1}:
                r = lib.trpl_shard_of(S, n, s)
                lo, hi = trpl.dist.shard_bounds(S, n, r)
                assert lo <= s < hi
    assert lib.trpl_shard_of(10, 3, 10) == -1 and lib.trpl_shard_of(10, 0, 1) == -1
    # multi-device handle: argument errors come before any device or RCCL is touched
    assert lib.trpl_multi_create(None, 0, None) == A.ERR_ARG
    assert lib.trpl_multi_device_count(None) == 0 and lib.trpl_multi_destroy(None) == A.OK
    assert lib.trpl_multi_synchronize(None) == A.ERR_ARG
    assert lib.trpl_loglik_multi_dev(None, None, 0, 1, None, 1.0, 128, 10, 1, 7, 10, None, None, None, None, None, 1,
                                     None, None, None, None, None, None, 0) == A.ERR_ARG
    assert lib.trpl_multi_wait_stream(None, 0, None) == A.ERR_ARG
    assert lib.trpl_multi_release_stream(None, 0, None) == A.ERR_ARG


def test_library_has_no_link_dependency_on_rccl():
    """RCCL (0.5 GB) is bound with dlopen when trpl_multi_create is first called, never at load time."""
    import subprocess
    so = os.path.join(ROOT, "bayesian-inference-trpl_amd", "libtrpl_hip.so")
    out = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "NEEDED" in out and "rccl" not in out.lower()


def _copy_package_with_stand_in_makefile(tmp_path):
    """The package tree (python + csrc + include/trpl.h) copied under tmp_path with a Makefile that logs each link and
    installs the REAL library the way the real Makefile does (temporary name + rename, then the source-hash stamp)."""
    import shutil
    pkg = os.path.join(ROOT, "bayesian-inference-trpl_amd")
    dst = tmp_path / "pkg"
    dst.mkdir()
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            shutil.copy(os.path.join(pkg, f), dst / f)
    shutil.copytree(os.path.join(pkg, "csrc"), dst / "csrc", ignore=shutil.ignore_patterns("build"))
    (tmp_path / "include").mkdir()
    shutil.copy(os.path.join(ROOT, "include", "trpl.h"), tmp_path / "include" / "trpl.h")
    real = os.path.join(pkg, "libtrpl_hip.so")
    (dst / "Makefile").write_text(
        "SRCFILES := $(sort $(wildcard csrc/*.hip csrc/*.hpp)) ../include/trpl.h Makefile\n"
        "all: libtrpl_hip.so\n"
        "libtrpl_hip.so: $(SRCFILES)\n\techo build >> builds.log; sleep 1; cp %s $@.tmp.$$$$ && mv -f $@.tmp.$$$$ $@ "
        "&& cat $(SRCFILES) | sha256sum | cut -d' ' -f1 > $@.srchash\n" % real)
    code = ("import importlib.util, sys\n"
            "spec = importlib.util.spec_from_file_location('trpl_tmp', %r, submodule_search_locations=[%r])\n"
            "m = importlib.util.module_from_spec(spec); sys.modules['trpl_tmp'] = m; spec.loader.exec_module(m)\n"
            "assert m._abi.lib().trpl_abi_version() == m._abi.ABI_VERSION\n"
            "assert m._abi.library_is_current()\n"
            "print('loaded', m._abi.LIB_PATH)\n") % (str(dst / "__init__.py"), str(dst))
    env = {k: v for k, v in os.environ.items() if k not in ("TRPL_LIBRARY", "TRPL_AUTOBUILD")}
    return dst, code, env


def test_concurrent_first_imports_build_the_library_once(tmp_path):
    """Three processes that find no shared object at the same time (the ranks of torch.distributed.run on a
    fresh checkout): the build is serialised by a lock file and re-checked under it -- one build runs, every
    process loads a complete library."""
    import subprocess
    import sys
    dst, code, env = _copy_package_with_stand_in_makefile(tmp_path)
    procs = [subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for _ in range(3)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and str(dst) in o, e[-1500:]
    assert (dst / "builds.log").read_text().count("build") == 1


def test_a_library_older_than_its_sources_is_rebuilt_not_used(tmp_path):
    """ensure_built() compares the hash of csrc/ + include/trpl.h + Makefile with the one recorded at the last link
    (libtrpl_hip.so.srchash): an import after an edit rebuilds, a second import does not, and a stamp that disagrees
    although `make` sees nothing to do (file times lost in a copy) forces a full rebuild."""
    import subprocess
    import sys
    dst, code, env = _copy_package_with_stand_in_makefile(tmp_path)
    run = lambda: subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    builds = lambda: (dst / "builds.log").read_text().count("build")
    r = run(); assert r.returncode == 0, r.stderr[-1500:]
    assert builds() == 1
    r = run(); assert r.returncode == 0 and builds() == 1               # current: nothing is built
    with open(dst / "csrc" / "trpl_common.hpp", "a") as fh:              # an edit of a kernel header
        fh.write("// edited\n")
    r = run(); assert r.returncode == 0, r.stderr[-1500:]
    assert builds() == 2
    (dst / "libtrpl_hip.so.srchash").write_text("0" * 64 + "\n")        # stamp disagrees, every file time is in order
    later = os.path.getmtime(dst / "libtrpl_hip.so") + 5
    os.utime(dst / "libtrpl_hip.so", (later, later)); os.utime(dst / "libtrpl_hip.so.srchash", (later + 1, later + 1))
    r = run(); assert r.returncode == 0, r.stderr[-1500:]
    assert builds() == 3
    # a tree copied WITHOUT the (git-ignored) stamp whose binary is newer than every source: `make` sees nothing to do, and
    # must not get a stamp from anywhere but a link -- the forced pass rebuilds (round-3 advice: the stamp used to be a
    # target of its own, which an unforced `make all` wrote next to the old binary)
    os.remove(dst / "libtrpl_hip.so.srchash")
    later = os.path.getmtime(dst / "libtrpl_hip.so") + 5
    os.utime(dst / "libtrpl_hip.so", (later, later))
    r = run(); assert r.returncode == 0, r.stderr[-1500:]
    assert builds() == 4
    # where it cannot be rebuilt (TRPL_AUTOBUILD=0, or no compiler) a stale binary is at least not used SILENTLY
    with open(dst / "csrc" / "trpl_common.hpp", "a") as fh:
        fh.write("// edited again\n")
    warn = code.replace("assert m._abi.library_is_current()\n", "")
    r = subprocess.run([sys.executable, "-W", "always", "-c", warn], env=dict(env, TRPL_AUTOBUILD="0"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and builds() == 4 and "built from other sources" in r.stderr, r.stderr[-1500:]


def test_bundle_flag_encoding_matches_the_header():
    """TRPL_FLAG_BUNDLE(m) = ((m - 1) & 0xF) << 8 (include/trpl.h); the binding refuses sizes the kernels do not have."""
    import re
    import trpl_amd
    A = trpl_amd._abi
    hdr = open(os.path.join(ROOT, "include", "trpl.h")).read()
    assert re.search(r"#define TRPL_FLAG_BUNDLE\(m\) \(\(uint32_t\)\(\(\(m\) - 1\) & 0xF\) << 8\)", hdr)
    assert [A.flag_bundle(m) for m in (1, 2, 3, 4, 16)] == [0, 0x100, 0x200, 0x300, 0xF00]
    for bad in (0, 17):
        with pytest.raises(ValueError):
            A.flag_bundle(bad)
    # the library takes 16 systems per bundle up to L = 64 and 4 from L = 128 on (csrc/trpl_common.hpp: bundle_cap)
    assert [A.bundle_cap(L) for L in (8, 32, 64, 128, 512)] == [16, 16, 16, 4, 4]
    assert A.flag_bundle(13, 32) == 0xC00 and A.flag_bundle(4, 128) == 0x300
    with pytest.raises(ValueError):
        A.flag_bundle(5, 128)
    # no other flag uses bits 8-11
    others = [A.FLAG_STRICT, A.FLAG_PL_F32, A.FLAG_NORMALIZE, A.FLAG_FP32, A.FLAG_KERNEL_PAIR, A.FLAG_KERNEL_SINGLE,
              A.FLAG_MIXED, A.FLAG_SNAP_RAW, A.FLAG_FP32_LONG]
    assert all(f & 0xF00 == 0 for f in others) and len(set(others)) == len(others)


def test_pair_table_covers_every_system_of_a_period_once(trpl):
    """trpl_pair_table: the paired stepper's rule for who shares a wavefront (host logic, no device).  Every (curve, sample
    offset) of a two-sample period appears exactly once; partners have the same thickness and observation count;
    same-sample pairs are neighbouring curves of a group, a group of odd size pairs its first curve across the samples."""
    lib = trpl._abi.lib()

    def table(lengths, n_obs):
        C = len(lengths)
        ln = np.ascontiguousarray(lengths, dtype=np.float64)
        no = np.ascontiguousarray(n_obs, dtype=np.int64)
        out = [np.full(C, -1, dtype=np.int32) for _ in range(4)]
        n = lib.trpl_pair_table(ln.ctypes.data, no.ctypes.data, C, 128, 8000, 200.0, *[o.ctypes.data for o in out])
        return n, [tuple(int(o[k]) for o in out) for k in range(max(n, 0))]

    n, t = table([2000.0] * 3, [8001] * 3)                       # Power_scan
    assert n == 3 and t == [(1, 0, 2, 0), (1, 1, 2, 1), (0, 0, 0, 1)]
    n, t = table([311.0, 2000.0] * 3, [8001] * 6)                # Twothick: two groups of three
    assert n == 6 and t == [(2, 0, 4, 0), (2, 1, 4, 1), (0, 0, 0, 1), (3, 0, 5, 0), (3, 1, 5, 1), (1, 0, 1, 1)]
    n, t = table([2000.0] * 4, [8001] * 4)                       # even group: only same-sample pairs
    assert t == [(0, 0, 1, 0), (0, 1, 1, 1), (2, 0, 3, 0), (2, 1, 3, 1)]
    n, t = table([2000.0] * 3, [8001, 8001, 500])                # a different observation count is a group of its own
    assert t == [(0, 0, 1, 0), (0, 1, 1, 1), (2, 0, 2, 1)]
    assert table([2000.0], [8001])[0] == 0                       # one curve: adjacent samples, no table
    rng = np.random.default_rng(0)
    for _ in range(50):                                          # any grouping: a partition of the period's 2 C systems
        C = int(rng.integers(2, 17))
        lengths = rng.choice([311.0, 1000.0, 2000.0], C)
        n_obs = rng.choice([100, 8001], C)
        n, t = table(lengths, n_obs)
        seen = sorted([(a, oa) for a, oa, _, _ in t] + [(b, ob) for _, _, b, ob in t])
        assert n == C and seen == sorted((c, o) for c in range(C) for o in (0, 1))
        for a, oa, b, ob in t:
            assert lengths[a] == lengths[b] and n_obs[a] == n_obs[b] and ((a != b and oa == ob) or (a == b and (oa, ob) == (0, 1)))
    assert lib.trpl_pair_table(None, None, 3, 128, 8000, 200.0, None, None, None, None) == -trpl._abi.ERR_ARG


def test_python_constants_equal_the_headers_defines():
    """Every flag, kernel id and documented constant of include/trpl.h that the Python binding restates has the same
    value there (the header is the contract; the envelope constants are what the -m gpu parity tests assert)."""
    from trpl_amd import _abi as A
    hdr = open(os.path.join(ROOT, "include", "trpl.h")).read()
    defs = dict(re.findall(r"^#define (TRPL_[A-Z0-9_]+) +([0-9][0-9a-fA-Fx.e+-]*)\b", hdr, flags=re.M))
    num = lambda v: int(v, 0) if re.fullmatch(r"0[xX][0-9a-fA-F]+|\d+", v) else float(v)
    pairs = {"TRPL_FLAG_STRICT": A.FLAG_STRICT, "TRPL_FLAG_PL_F32": A.FLAG_PL_F32, "TRPL_FLAG_NORMALIZE": A.FLAG_NORMALIZE,
             "TRPL_FLAG_FP32": A.FLAG_FP32, "TRPL_FLAG_KERNEL_PAIR": A.FLAG_KERNEL_PAIR, "TRPL_FLAG_KERNEL_SINGLE": A.FLAG_KERNEL_SINGLE,
             "TRPL_FLAG_MIXED": A.FLAG_MIXED, "TRPL_FLAG_SNAP_RAW": A.FLAG_SNAP_RAW, "TRPL_FLAG_FP32_LONG": A.FLAG_FP32_LONG,
             "TRPL_FLAG_HIST32": A.FLAG_HIST32, "TRPL_FP32_MAX_STEPS": A.FP32_MAX_STEPS,
             "TRPL_KERNEL_FAST": A.KERNEL_FAST, "TRPL_KERNEL_FAST_PAIR": A.KERNEL_FAST_PAIR, "TRPL_KERNEL_STRICT": A.KERNEL_STRICT,
             "TRPL_KERNEL_FP32": A.KERNEL_FP32, "TRPL_KERNEL_MIXED": A.KERNEL_MIXED, "TRPL_KERNEL_HIST32": A.KERNEL_HIST32,
             "TRPL_PL_FLOOR_EXCESS": A.PL_FLOOR_EXCESS, "TRPL_PL_ENVELOPE_K_THICK": A.PL_ENVELOPE_K_THICK,
             "TRPL_PL_ENVELOPE_K_THIN": A.PL_ENVELOPE_K_THIN, "TRPL_PL_ENVELOPE_K_L512": A.PL_ENVELOPE_K_L512,
             "TRPL_ABI_VERSION": A.ABI_VERSION, "TRPL_FLAG_PAIR_ALWAYS_SEAM": A.FLAG_PAIR_ALWAYS_SEAM,
             "TRPL_FLAG_PAIR_ADJACENT": A.FLAG_PAIR_ADJACENT, "TRPL_FLAG_MULTI_FORCE_PAD": A.FLAG_MULTI_FORCE_PAD,
             "TRPL_MULTI_ALLOW_DUPLICATE_DEVICES": A.MULTI_ALLOW_DUPLICATE_DEVICES}
    for name, val in pairs.items():
        assert name in defs, name
        assert num(defs[name]) == val, (name, defs[name], val)
    flags = [v for k, v in pairs.items() if k.startswith("TRPL_FLAG_")]
    assert len(set(flags)) == len(flags) and all(f & (f - 1) == 0 for f in flags)          # distinct single bits
    assert not any(f & 0xF00 for f in flags)                                                # TRPL_FLAG_BUNDLE's field
    assert not any(f & (7 << 14) for f in flags)                                            # TRPL_FLAG_BDF_ORDER's field
    assert re.search(r"#define TRPL_FLAG_BDF_ORDER\(k\) \(\(uint32_t\)\(\(k\) & 0x7\) << 14\)", hdr)
    assert [A.flag_bdf_order(k) for k in (None, 0, 1, 2, 5)] == [0, 0, 1 << 14, 2 << 14, 5 << 14]
    with pytest.raises(ValueError):
        A.flag_bdf_order(6)
    # the envelope constants the -m gpu files assert are the header's (tests/gpu_common.py restates them)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_common as G
    assert G.ENVELOPE_K_THICK == A.PL_ENVELOPE_K_THICK and G.ENVELOPE_K == {2000.0: A.PL_ENVELOPE_K_THICK, 311.0: A.PL_ENVELOPE_K_THIN}
    assert G.ENVELOPE_K_L512 == A.PL_ENVELOPE_K_L512 and G.FLOOR == A.PL_FLOOR_EXCESS


def test_kernel_name_and_round5_flags_need_no_device(trpl):
    """trpl_kernel_name spells the instantiation a launch runs (bench.py's roofline.rocprof_name; rocprofv3 lists it after
    "void "), the per-call test flags that replaced the environment switches reach it, and the BDF-order field is validated."""
    A = trpl._abi
    assert A.kernel_name(196608, 128, 8000) == "trpl::pair::stepper_pair_kernel<true, false, true>"
    assert A.kernel_name(196608, 128, 8000, A.FLAG_PAIR_ALWAYS_SEAM) == "trpl::pair::stepper_pair_kernel<true, false, false>"
    assert A.kernel_name(196608, 128, 8000, A.FLAG_KERNEL_PAIR, snapshots=True) == "trpl::pair::stepper_pair_kernel<true, true, true>"
    assert A.kernel_name(64, 128, 8000) == "trpl::stepper_kernel<128, false, false, false, false, false>"
    assert A.kernel_name(10 ** 6, 128, 8000, A.FLAG_STRICT) == "trpl::stepper_kernel<128, true, false, false, false, false>"
    if A.has_experimental():                        # `make EXPERIMENTAL=1` (libtrpl_hip_exp.so under TRPL_LIBRARY)
        assert A.kernel_name(10 ** 6, 512, 8000, A.FLAG_HIST32) == "trpl::stepper_kernel<512, false, false, false, false, true>"
        assert A.kernel_name(10 ** 6, 512, 8000, A.FLAG_MIXED) == "trpl::stepper_kernel<512, false, false, true, false, false>"
    else:                                           # the default library has no such instantiation and says so
        for fl, word in ((A.FLAG_HIST32, "TRPL_FLAG_HIST32"), (A.FLAG_MIXED, "TRPL_FLAG_MIXED")):
            with pytest.raises(A.TrplError) as e:
                A.kernel_name(10 ** 6, 512, 8000, fl)
            assert e.value.code == A.ERR_UNSUPPORTED and word in str(e.value) and "EXPERIMENTAL=1" in str(e.value)
    assert A.kernel_name(10 ** 6, 128, 8000, A.flag_bundle(3, 128)) == "trpl::stepper_kernel<128, false, false, false, true, false>"
    assert A.kernel_name(10 ** 6, 512, 100, A.FLAG_FP32) == "trpl::f32::stepper_kernel<512>"
    lib = A.lib()
    import ctypes as C
    small = C.create_string_buffer(8)
    assert lib.trpl_kernel_name(10, 128, 10, 0, 0, C.addressof(small), 8) == A.ERR_ARG and b"buflen" in lib.trpl_last_error()
    assert lib.trpl_kernel_name(10, 128, 10, 0, 0, None, 0) == A.ERR_ARG
    # the library reads no process-wide switch but TRPL_RCCL_LIBRARY (no hidden globals)
    import glob
    src = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "bayesian-inference-trpl_amd", "csrc", "*.h*")))
    assert re.findall(r'getenv\("([A-Z_]+)"\)', src) == ["TRPL_RCCL_LIBRARY"]


def test_kernel_name_refuses_what_a_launch_refuses(trpl):
    """trpl_kernel_name runs the flag / shape checks of a launch (csrc/trpl_api.hip check_launch): it returns a name exactly
    for the combinations a launch accepts, and every name it returns is an instantiation the library contains (the mangled
    kernel symbols of the shared object are the instantiation list).  bench.py attaches rocprof statistics by that name, so a
    name without a kernel would silently attach nothing."""
    import itertools
    import subprocess
    A = trpl._abi
    nm = subprocess.run(["nm", "-D", "--defined-only", A.LIB_PATH], capture_output=True, text=True).stdout
    filt = subprocess.run(["c++filt"], input=nm, capture_output=True, text=True).stdout
    have = set(re.findall(r"(trpl::(?:pair::|f32::)?stepper(?:_pair)?_kernel<[^>]*>)", filt))
    assert len(have) >= 40, len(have)
    exp = A.has_experimental()
    named = refused = 0
    arith = [0, A.FLAG_STRICT, A.FLAG_FP32, A.FLAG_FP32 | A.FLAG_FP32_LONG, A.FLAG_MIXED, A.FLAG_HIST32, A.FLAG_STRICT | A.FLAG_FP32,
             A.FLAG_MIXED | A.FLAG_HIST32]
    for L, ar, kern, bundle, snap, steps in itertools.product(
            (2, 4, 64, 128, 256, 512, 1024, 96), arith, (0, A.FLAG_KERNEL_PAIR, A.FLAG_KERNEL_SINGLE, A.FLAG_KERNEL_PAIR | A.FLAG_KERNEL_SINGLE),
            (1, 2, 4, 5, 16), (False, True), (100, 8000)):
        flags = ar | kern | (((bundle - 1) & 0xF) << 8)
        # what a launch accepts, restated from include/trpl.h's flag paragraphs
        ok = L in (4, 8, 16, 32, 64, 128, 256, 512) and kern != (A.FLAG_KERNEL_PAIR | A.FLAG_KERNEL_SINGLE)
        fp32, strict, mixed, hist = bool(ar & A.FLAG_FP32), bool(ar & A.FLAG_STRICT), bool(ar & A.FLAG_MIXED), bool(ar & A.FLAG_HIST32)
        if kern == A.FLAG_KERNEL_PAIR:
            ok &= L == 128 and not (strict or fp32 or mixed or hist)
        ok &= not ((mixed or hist) and not exp)
        ok &= not (hist and (strict or fp32 or mixed or L not in (256, 512) or snap or bundle > 1))
        ok &= not (mixed and (strict or fp32 or L < 128))
        ok &= not (fp32 and (strict or L < 128 or (steps > A.FP32_MAX_STEPS and not ar & A.FLAG_FP32_LONG)))
        ok &= not (bundle > 1 and (fp32 or mixed or kern == A.FLAG_KERNEL_PAIR or bundle > A.bundle_cap(L) or (not strict and L > 128)))
        try:
            name = A.kernel_name(10 ** 6, L, steps, flags, snapshots=snap)
        except A.TrplError as e:
            assert not ok, (L, hex(flags), snap, steps, str(e))
            assert e.code in (A.ERR_ARG, A.ERR_UNSUPPORTED)
            refused += 1
            continue
        assert ok, (L, hex(flags), snap, steps, name)
        assert name in have, (name, L, hex(flags), snap, steps)
        named += 1
    assert named > 100 and refused > 1000, (named, refused)


def test_roctx_ranges_bracket_the_host_buffer_calls_when_a_profiler_provides_them(trpl, tmp_path):
    """ABI 5: the host-buffer entry points open a ROCTx range named after the reference callable they replace (pvSim /
    fastlog / prob: pvSimPCR.py:378-381, probs.py:79-84, :51-61) -- bound at run time from whatever the process exports (no
    link dependency; `rocprofv3 --marker-trace` preloads librocprofiler-sdk-roctx.so), a no-op otherwise.  A child process
    with tests/mock_roctx preloaded calls the three callables' entry points without a device: each call returns its
    ordinary status (TRPL_ERR_NODEVICE here, TRPL_OK for an empty batch) and has pushed and popped exactly one range;
    without the preload the same calls behave identically and nothing is pushed.  (On the GPU box the real marker trace is
    taken with tools/marker_probe.py -> profiles/r6_marker_trace.txt.)"""
    import subprocess
    so = str(tmp_path / "libmock_roctx.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", so, os.path.join(ROOT, "tests", "mock_roctx", "mock_roctx.c")])
    code = r"""
import ctypes as C, os, sys
sys.path.insert(0, %r)
import numpy as np
import trpl_amd
A = trpl_amd._abi
lib = A.lib()
nodev = lib.trpl_device_count() == 0
x = np.ones((4, 8), dtype=np.float32)
P = np.zeros(4); vals = np.zeros(8); mag = np.zeros(4)
X = np.ones((4, 12)); dN = np.ones(128); pl = np.empty((4, 3)); st = np.zeros(4, dtype=np.int32)
sec = C.c_double()
rcs = [lib.trpl_log10_clamp(A.ptr(x), 4, 4, 8, 8, 1e-300, 0, C.byref(sec)),
       lib.trpl_sse_accumulate(A.ptr(P), A.ptr(x), 4, 4, 8, 8, A.ptr(vals), A.ptr(mag), 0, C.byref(sec)),
       lib.trpl_solve_pl(A.ptr(X), 4, 2000.0, 0.05, 128, 2, 1, 7, 100, A.ptr(dN), A.ptr(pl), 8, 3, A.ptr(st), None, 0, 0, C.byref(sec)),
       lib.trpl_log10_clamp(A.ptr(x), 4, 0, 8, 8, 1e-300, 0, C.byref(sec))]          # an empty batch: TRPL_OK, still one range
want = ([A.ERR_NODEVICE] * 3 if nodev else [A.OK] * 3) + [A.OK]
assert rcs == want, (rcs, want)
m = os.environ.get("MOCK_ROCTX")
if m:
    mock = C.CDLL(m)
    mock.mock_roctx_name.restype = C.c_char_p
    n = mock.mock_roctx_pushes()
    print("RANGES", n, mock.mock_roctx_pops(), mock.mock_roctx_depth(), mock.mock_roctx_max_depth(),
          "|".join(mock.mock_roctx_name(i).decode() for i in range(n)))
else:
    print("RANGES none")
""" % ROOT
    env = dict(os.environ, TRPL_AUTOBUILD="0")
    env.pop("LD_PRELOAD", None)
    plain = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0 and "RANGES none" in plain.stdout, plain.stderr[-2000:]
    traced = subprocess.run([sys.executable, "-c", code], env=dict(env, LD_PRELOAD=so, MOCK_ROCTX=so), capture_output=True,
                            text=True, timeout=300)
    assert traced.returncode == 0, traced.stderr[-2000:]
    line = [ln for ln in traced.stdout.splitlines() if ln.startswith("RANGES")][0].split(" ", 5)
    assert line[1:5] == ["4", "4", "0", "1"], line                       # four calls: four ranges, all closed, never nested
    assert line[5].split("|") == ["trpl_log10_clamp (fastlog)", "trpl_sse_accumulate (prob)", "trpl_solve_pl (pvSim)",
                                  "trpl_log10_clamp (fastlog)"]
