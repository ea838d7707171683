"""Multi-GPU: contiguous sample shards, one all-gather.  trpl_loglik_multi (host buffers), trpl_multi_* /
trpl_loglik_multi_dev (device-resident, RCCL) on a one-rank communicator and on 2-4 "ranks" of one GPU against tests/mock_rccl,
sharded = single launch bit for bit across the pair threshold, bench.py's N = 2 control flow as child processes (gloo), the
rank driver over a one-rank nccl group, stream ordering against torch."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from gpu_common import DT

pytestmark = pytest.mark.gpu


def test_multi_device_entry_point_equals_single_device(trpl, gpu):
    """trpl_loglik_multi: the shards of one host thread's call (three streams on this box's one device,
    uneven shard sizes) give bit-for-bit the single-launch result, on- and off-grid, all outputs."""
    X = trpl.workloads.samples(50, seed=5)
    ini, lengths = trpl.workloads.power_scan(128)
    T, Time = 120, 3.0
    ref_info = {}
    obs0 = [np.full(T + 1, 20.0) - 0.01 * np.arange(T + 1)] * 3
    want = trpl.loglik(X, ini, lengths, Time, 128, T, obs0, info=ref_info)
    for devices in ([0], [0, 0, 0], "all"):
        info = {}
        got = trpl.loglik(X, ini, lengths, Time, 128, T, obs0, info=info, devices=devices)
        assert np.array_equal(got, want)
        for k in ("sse", "status", "iters_total"):
            assert np.array_equal(info[k], ref_info[k]), (devices, k)
    times = [np.sort(np.random.default_rng(c).uniform(0, Time, 40)) for c in range(3)]
    obs1 = [np.full(40, 19.5)] * 3
    want = trpl.loglik(X, ini, lengths, Time, 128, T, obs1, times=times)
    got = trpl.loglik(X, ini, lengths, Time, 128, T, obs1, times=times, devices=[0, 0])
    assert np.array_equal(got, want)
    # more shards than samples: empty shards are skipped
    got = trpl.loglik(X[:2], ini, lengths, Time, 128, T, obs0, devices=[0, 0, 0, 0])
    assert np.array_equal(got, trpl.loglik(X[:2], ini, lengths, Time, 128, T, obs0))
    with pytest.raises(trpl.TrplError):
        trpl.loglik(X, ini, lengths, Time, 128, T, obs0, devices=[0, 99])
    # shards large enough for the two-systems-per-wavefront kernel: other partners, same bits
    Xb = trpl.workloads.samples(10243, seed=6)
    obs2 = [np.full(41, 20.0)] * 3
    one = trpl.loglik(Xb, ini, lengths, 1.0, 128, 40, obs2)
    assert np.array_equal(trpl.loglik(Xb, ini, lengths, 1.0, 128, 40, obs2, devices=[0, 0]), one)


# ------------------------------------------------------------------ sharding invariance
def test_sharded_batch_equals_single_launch_across_the_pair_threshold(gpu):
    """The whole batch is above the paired kernel's threshold, every shard is below it: the variant is a
    property of the logical batch, so trpl_loglik_multi (8 shards) and a rank driver that pins its flags
    from the total (dist / bench.py) return bit for bit the single launch's likelihoods."""
    w = gpu.workloads
    S, T, Time = 5124, 40, 1.0
    lib = gpu._abi.lib()
    A = gpu._abi
    assert lib.trpl_kernel_variant(3 * S, 128, T, 0) == A.KERNEL_FAST_PAIR
    assert lib.trpl_kernel_variant(3 * (S // 8 + 1), 128, T, 0) == A.KERNEL_FAST
    ini, lens = w.power_scan(128)
    X = w.samples(S, seed=31)
    obs = [np.full(T + 1, 20.0) - 0.01 * np.arange(T + 1)] * 3
    one, multi = {}, {}
    want = gpu.loglik(X, ini, lens, Time, 128, T, obs, info=one)
    got = gpu.loglik(X, ini, lens, Time, 128, T, obs, info=multi, devices=[0] * 8)
    assert np.array_equal(got, want)
    for k in ("sse", "status", "iters_total"):
        assert np.array_equal(multi[k], one[k]), k
    # the rank driver: each rank launches its own shard with the flags pinned from the TOTAL
    flags = A.pin_variant(0, 3 * S, 128, T)
    assert flags == A.FLAG_KERNEL_PAIR
    for world in (2, 8):
        parts = []
        for r in range(world):
            lo, hi = gpu.dist.shard_bounds(S, world, r)
            parts.append(gpu.loglik(X[lo:hi], ini, lens, Time, 128, T, obs, kernel="pair"))
        assert np.array_equal(np.concatenate(parts), want)
    # without the pin the shards would run the other kernel: close (rounding), not identical -- the
    # documented reason for pinning
    lo, hi = gpu.dist.shard_bounds(S, 8, 1)
    unpinned = gpu.loglik(X[lo:hi], ini, lens, Time, 128, T, obs)
    assert np.allclose(unpinned, want[lo:hi], rtol=1e-10, atol=0)
    # a small batch stays on the one-system kernel in every shard
    small = gpu.loglik(X[:300], ini, lens, Time, 128, T, obs)
    assert np.array_equal(gpu.loglik(X[:300], ini, lens, Time, 128, T, obs, devices=[0, 0, 0]), small)


def test_multi_validates_observation_brackets_like_the_single_device_call(gpu):
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    X = w.samples(6)
    T, Time = 40, 1.0
    lib = gpu._abi.lib()
    A = gpu._abi
    n = 5
    obs = np.full((3, n), 19.0)
    n_obs = np.full(3, n, dtype=np.int64)
    hi = np.tile(np.array([3, 2, 5, 7, 9], dtype=np.int32), (3, 1))        # not sorted
    dx = np.full((3, n), 0.01)
    h = np.full((3, n), 0.025)
    P = np.zeros(6)
    dev = np.zeros(2, dtype=np.int32)

    def call(hi_, dx_, h_):
        return lib.trpl_loglik_multi(X.ctypes.data, 6, 3, lens.ctypes.data, Time, 128, T, 1, 7, 1000, ini.ctypes.data,
                                     obs.ctypes.data, hi_.ctypes.data, dx_.ctypes.data, h_.ctypes.data, n,
                                     n_obs.ctypes.data, P.ctypes.data, None, None, None, None, 0, dev.ctypes.data, 2, None)
    assert call(hi, dx, h) == A.ERR_ARG and b"sorted" in lib.trpl_last_error()
    good = np.tile(np.array([2, 3, 5, 7, 9], dtype=np.int32), (3, 1))
    bad_hi = good.copy(); bad_hi[1, 4] = T + 1
    assert call(bad_hi, dx, h) == A.ERR_ARG
    bad_h = h.copy(); bad_h[2, 0] = 0.0
    assert call(good, dx, bad_h) == A.ERR_ARG
    bad_dx = dx.copy(); bad_dx[0, 1] = 0.05                                # beyond the bracket
    assert call(good, bad_dx, h) == A.ERR_ARG
    assert call(good, dx, h) == A.OK
    with pytest.raises(ValueError):                                         # the Python driver refuses earlier still
        gpu.loglik(X, ini, lens, Time, 128, T, [np.full(3, 19.0)] * 3, times=[np.array([0.1, 0.2, 2.0])] * 3,
                   devices=[0, 0])


# ------------------------------------------------------------------ device-resident multi-GPU (RCCL)
@pytest.mark.parametrize("force_pad", [False, True])
def test_multi_device_resident_allgather_on_a_one_rank_communicator(gpu, force_pad):
    """trpl_multi_create (ncclCommInitAll, RCCL bound at first use) + trpl_loglik_multi_dev with the one device
    of this box: the all-gathered P[S] left in device memory equals trpl_loglik_dev's, per-shard outputs
    included; with TRPL_FLAG_MULTI_FORCE_PAD the padded exchange + unpadding pass runs instead of the direct one
    (a per-call flag since round 5: both cases in this process)."""
    _multi_dev_check(gpu, gpu._abi.FLAG_MULTI_FORCE_PAD if force_pad else 0)


def _multi_dev_check(gpu, flags=0):
    import torch
    w = gpu.workloads
    dev = torch.device("cuda", 0)
    ini, lens = w.power_scan(128)
    S, T, Time = 777, 60, 1.5
    X = torch.from_numpy(w.samples(S, seed=51)).to(dev)
    ini_d = torch.from_numpy(ini).to(dev)
    obs = torch.full((3, T + 1), 20.0, dtype=torch.float64, device=dev) - 0.01 * torch.arange(T + 1, device=dev)
    P = torch.zeros(S, dtype=torch.float64, device=dev)
    sse = torch.empty((3, S), dtype=torch.float64, device=dev)
    st = torch.empty((3, S), dtype=torch.int32, device=dev)
    it = torch.empty((3, S), dtype=torch.int64, device=dev)
    gpu.device.loglik_device(X, ini_d, lens, Time, 128, T, obs, [T + 1] * 3, P, sse, st, it)
    torch.cuda.synchronize()
    with gpu.device.MultiDevice([0]) as md:
        assert md.n == 1
        Pf = torch.full((S,), 123.0, dtype=torch.float64, device=dev)
        sse2, st2, it2 = torch.empty_like(sse), torch.empty_like(st), torch.empty_like(it)
        for _ in range(2):                                                  # the handle is reusable
            md.loglik([X], [ini_d], lens, Time, 128, T, [obs], [T + 1] * 3, [Pf], sse=[sse2], status=[st2],
                      iters_total=[it2], flags=flags)
            md.synchronize()
            assert torch.equal(Pf, P) and torch.equal(sse2, sse) and torch.equal(st2, st) and torch.equal(it2, it)
        Pg = torch.zeros(S, dtype=torch.float64, device=dev)               # optional outputs left out
        md.loglik([X], [ini_d], lens, Time, 128, T, [obs], [T + 1] * 3, [Pg], flags=flags)
        md.synchronize()
        assert torch.equal(Pg, P)
        # off-grid observations through the same entry point
        times = np.sort(np.random.default_rng(3).uniform(0, Time, 25))
        hi, dx, h = gpu.bracket_times(np.linspace(0, Time, T + 1), times)
        rep = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.tile(a, (3, 1)))).to(dev).to(dt)
        o2 = torch.full((3, 25), 19.5, dtype=torch.float64, device=dev)
        hi_d, dx_d, h_d = rep(hi, torch.int32), rep(dx, torch.float64), rep(h, torch.float64)
        P2 = torch.zeros(S, dtype=torch.float64, device=dev)
        sse3 = torch.empty((3, S), dtype=torch.float64, device=dev)
        gpu.device.loglik_obs_device(X, ini_d, lens, Time, 128, T, o2, hi_d, dx_d, h_d, [25] * 3, P2, sse3)
        md.loglik([X], [ini_d], lens, Time, 128, T, [o2], [25] * 3, [Pg], obs_hi=[hi_d], obs_dx=[dx_d], obs_h=[h_d], flags=flags)
        md.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(Pg, P2)
    with pytest.raises(gpu.TrplError):                                      # one RCCL rank per device
        gpu.device.MultiDevice([0, 0])


# ------------------------------------------------------------------ bench.py, N = 2 control flow
def test_bench_two_rank_rehearsal_gathers_the_single_rank_likelihoods(gpu, tmp_path):
    """bench.py --gpus 2 --backend gloo as fresh child processes sharing this box's GPU (the N > 1 path:
    sample shards, pinned kernel variant, all-gather, max-over-ranks timing) must print one contract line
    and gather exactly the likelihood vector a single rank computes for the same 4 096 samples."""
    common = ["--steps", "1", "--warmup", "0", "--T", "200", "--no-cpu-baseline", "--no-pcr", "--no-full-length", "--no-host-api",
              "--no-other-configs", "--no-e2e"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TRPL_AUTOBUILD="0")
    p1, p2 = str(tmp_path / "p1.npy"), str(tmp_path / "p2.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--samples-per-gpu", "4096",
                         "--dump-p", p1] + common, env=env, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    # the plain form the driver uses for N = 1, with N = 2: bench.py starts its two ranks itself (fresh children under
    # torch.distributed.run, before the parent has imported torch or touched the GPU) and relays rank 0's line
    env2 = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                         "--samples-per-gpu", "2048", "--dump-p", p2] + common,
                        env=env2, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    line1 = json.loads(r1.stdout.strip().splitlines()[-1])
    lines2 = [ln for ln in r2.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines2) == 1                                              # ONE contract line, rank 0's
    line2 = json.loads(lines2[0])
    assert line2["n_gpus"] == 2 and line2["scaling"] == "weak" and line2["config"]["samples_total"] == 4096
    rc = line2["rccl"]                                                   # the ranks are proven, not assumed
    assert rc["world"] == 2 and [d["rank"] for d in rc["devices"]] == [0, 1] and rc["backend"] == "gloo"
    assert len({d["pid"] for d in rc["devices"]}) == 2 and rc["allgather_bytes"] == 4096 * 8 and rc["allgather_us"] > 0
    assert abs(line2["value_n1_equiv"] * 2 - line2["value"]) < 1e-6 * line2["value"] and "cpu_baseline" not in line2
    assert line1["config"]["arithmetic"] == "fast" and line1["config"]["precision"] == "fp64" and "rccl" not in line1
    # a launcher that has already set WORLD_SIZE is honoured as before (the driver's N > 1 form)
    r2b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--samples-per-gpu", "2048"] + common,
                         env=env, capture_output=True, text=True, timeout=900)
    assert r2b.returncode == 0, r2b.stderr[-2000:]
    assert json.loads(r2b.stdout.strip().splitlines()[-1])["rccl"]["world"] == 2
    assert line1["config"]["samples_total"] == 4096
    assert line2["nonconverged_systems"] == line1["nonconverged_systems"] == 0
    a, b = np.load(p1), np.load(p2)
    assert a.shape == b.shape == (1, 4096) and np.array_equal(a, b)
    # ... and the single-process form (one process, trpl_loglik_multi_dev + RCCL) on this box's one device
    p3 = str(tmp_path / "p3.npy")
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "1",
                         "--samples-per-gpu", "4096", "--steps", "1", "--warmup", "0", "--T", "200", "--dump-p", p3],
                        env=env, capture_output=True, text=True, timeout=900)
    assert r3.returncode == 0, r3.stderr[-2000:]
    line3 = json.loads(r3.stdout.strip().splitlines()[-1])
    assert line3["n_gpus"] == 1 and "ncclAllGather" in line3["config"]["collective"]
    assert np.array_equal(np.load(p3), a)


def _bench(args, env, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2500:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]                               # ONE contract line
    return json.loads(lines[0])


REHEARSAL = ["--steps", "1", "--warmup", "1", "--T", "200", "--no-cpu-baseline", "--no-pcr", "--no-full-length", "--no-host-api",
             "--no-other-configs", "--no-e2e"]


@pytest.mark.parametrize("S_total", [4098, 4099])
def test_bench_three_rank_rehearsal_with_even_and_uneven_shards(gpu, tmp_path, S_total):
    """The driver's N > 1 form of bench.py with more than two ranks on this box's one GPU: `bench.py --gpus 3 --backend gloo`
    starts its ranks itself (fresh children, before the parent touches the GPU; never a re-exec).  The pool admits SIX
    processes with the GPU open per box and counts this pytest process and the launcher too (a five-rank rehearsal was
    killed by that guard at 7), so eight ranks on one card cannot run here: world size 8 is covered on the CPU
    (tests/test_dist_gloo.py) and inside one process (next test).  S_total = 4098 (three equal shards) and 4099 (1367 + 1366
    + 1366: trpl_shard_bounds) gather the single launch's likelihoods BIT FOR BIT; the line carries the rccl record (world,
    every rank's pid / device / PCI bus id, all-gather bytes and time) and value_n1_equiv."""
    env = {k: v for k, v in dict(os.environ, MASTER_ADDR="127.0.0.1", TRPL_AUTOBUILD="0").items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    W = 3
    p1, pw = str(tmp_path / "p1.npy"), str(tmp_path / "pw.npy")
    one = _bench(["--gpus", "1", "--samples-total", str(S_total), "--dump-p", p1] + REHEARSAL, env)
    many = _bench(["--gpus", str(W), "--backend", "gloo", "--samples-total", str(S_total), "--dump-p", pw] + REHEARSAL, env)
    assert many["n_gpus"] == W and many["scaling"] == "weak" and many["config"]["samples_total"] == S_total
    assert many["config"]["collective"].startswith("gloo") and many["nonconverged_systems"] == one["nonconverged_systems"] == 0
    rc = many["rccl"]
    assert rc["world"] == W and [d["rank"] for d in rc["devices"]] == list(range(W)) and len({d["pid"] for d in rc["devices"]}) == W
    assert all(d["pci_bus_id"] is not None or d["name"] for d in rc["devices"])
    assert rc["allgather_bytes_per_rank"] == -(-S_total // W) * 8 and rc["allgather_bytes"] == W * rc["allgather_bytes_per_rank"]
    assert rc["allgather_us"] > 0 and abs(many["value_n1_equiv"] * W - many["value"]) < 1e-6 * many["value"]
    assert many["library"]["srchash"] == one["library"]["srchash"] and one["library"]["sources_current"]
    a, b = np.load(p1), np.load(pw)
    assert a.shape == b.shape == (1, S_total) and np.array_equal(a, b)


@pytest.mark.parametrize("S_total", [4096, 4099])
def test_single_process_bench_with_eight_ranks_on_one_device(gpu, tmp_path, S_total):
    """configs[3]'s rank count through the one-process form: `bench.py --single-process --gpus 8 --rehearse-on-device0` --
    trpl_multi_create_ex with eight ranks on device 0, the six RCCL entry points bound to tests/mock_rccl (RCCL refuses
    duplicate devices) -- ONE process on the GPU.  4096 samples: equal shards, direct exchange; 4099: 513 x 3 + 512 x 5,
    padded exchange + unpadding.  The gathered vector (identical on all eight "devices": bench.py asserts it) equals the
    single launch bit for bit, and the line carries the same rccl / value_n1_equiv fields as the per-rank form."""
    so = str(tmp_path / "libmock_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", so,
                           os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp")])
    env = dict(os.environ, TRPL_AUTOBUILD="0")
    p1, p8 = str(tmp_path / "p1.npy"), str(tmp_path / "p8.npy")
    one = _bench(["--gpus", "1", "--samples-total", str(S_total), "--dump-p", p1] + REHEARSAL, env)
    eight = _bench(["--single-process", "--gpus", "8", "--rehearse-on-device0", "--samples-total", str(S_total), "--steps", "1",
                    "--warmup", "1", "--T", "200", "--dump-p", p8], dict(env, TRPL_RCCL_LIBRARY=so))
    assert eight["n_gpus"] == 8 and eight["rehearsal_on_one_device"] and eight["config"]["samples_total"] == S_total
    rc = eight["rccl"]
    assert rc["world"] == 8 and [d["rank"] for d in rc["devices"]] == list(range(8)) and rc["distinct_devices"] == 1
    assert sum(d["samples"] for d in rc["devices"]) == S_total and rc["padded_exchange"] == (S_total % 8 != 0)
    assert rc["allgather_bytes_per_rank"] == -(-S_total // 8) * 8
    assert abs(eight["value_n1_equiv"] * 8 - eight["value"]) < 1e-6 * eight["value"]
    assert eight["nonconverged_systems"] == one["nonconverged_systems"] == 0
    assert np.array_equal(np.load(p8), np.load(p1))
    # without a stand-in library the rehearsal switch refuses to run (RCCL would reject the duplicate devices)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "8", "--rehearse-on-device0",
                        "--T", "50"], env={k: v for k, v in env.items() if k != "TRPL_RCCL_LIBRARY"}, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "TRPL_RCCL_LIBRARY" in r.stderr


def test_bench_collective_path_over_real_rccl_on_one_gpu(gpu, tmp_path):
    """Everything of bench.py's N > 1 line that one GPU can execute ON HARDWARE: `--gpus 1 --rehearse-collectives --backend
    nccl` runs the collective code path -- process group on the nccl backend (= RCCL), the all-gather inside the timed step
    with its event pair, the max-over-ranks all-reduce, the iteration / failure all-reduces, rccl_record's all_gather_object
    and its 20 isolated all-gathers -- on a one-rank communicator.  The 2- and 3-rank rehearsals above run the same code over
    gloo; this pins the RCCL calls themselves before the driver's first multi-GPU run.  The gathered vector equals the plain
    run's bit for bit."""
    env = {k: v for k, v in dict(os.environ, TRPL_AUTOBUILD="0").items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p1, p2 = str(tmp_path / "p1.npy"), str(tmp_path / "p2.npy")
    one = _bench(["--gpus", "1", "--samples-total", "4099", "--dump-p", p1] + REHEARSAL, env)
    rc1 = _bench(["--gpus", "1", "--rehearse-collectives", "--backend", "nccl", "--samples-total", "4099", "--dump-p", p2] + REHEARSAL, env)
    assert "rccl" not in one and one["config"]["collective"] == "none"
    assert rc1["config"]["collective"] == "RCCL all_gather" and rc1["n_gpus"] == 1
    r = rc1["rccl"]
    assert r["world"] == 1 and r["backend"] == "nccl" and r["allgather_bytes"] == 4099 * 8 and r["distinct_devices"] == 1
    assert r["allgather_us_in_loop_incl_rank_skew"] is not None and r["allgather_us_in_loop_incl_rank_skew"] > 0
    assert r["allgather_us_isolated"] > 0 and r["devices"][0]["device"] == 0
    assert rc1["nonconverged_systems"] == 0 and abs(rc1["value_n1_equiv"] - rc1["value"]) < 1e-9 * rc1["value"]
    assert np.array_equal(np.load(p1), np.load(p2))


def test_rank_driver_gathers_over_rccl_on_a_one_rank_group(gpu, tmp_path):
    """The one-process-per-GPU driver with the REAL collective backend: a child process joins a 1-rank
    torch.distributed group on the `nccl` backend (= RCCL on ROCm), computes its shard with the fused call and
    gathers with dist.gather_likelihoods on the device; the gathered vector equals the direct call's."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import trpl_amd
from trpl_amd import device as tdev, workloads as wl
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
S, T, L = 301, 40, 128
ini, lens = wl.power_scan(L)
X = torch.from_numpy(wl.samples(S, seed=71)).to(dev)
ini_d = torch.from_numpy(ini).to(dev)
obs = torch.full((3, T + 1), 20.0, dtype=torch.float64, device=dev)
P = torch.zeros(S, dtype=torch.float64, device=dev)
sse = torch.empty((3, S), dtype=torch.float64, device=dev)
flags = trpl_amd._abi.pin_variant(0, 3 * S, L, T)
tdev.loglik_device(X, ini_d, lens, T * 0.025, L, T, obs, [T + 1] * 3, P, sse, flags=flags)
full = trpl_amd.dist.gather_likelihoods(P[None, :], S)
torch.cuda.synchronize()
assert full.is_cuda and tuple(full.shape) == (1, S) and torch.equal(full[0], P)
np.save(%r, full.cpu().numpy())
dist.barrier(); dist.destroy_process_group()
print("RCCL-OK", dist.is_nccl_available())
''' % (ROOT, str(tmp_path / "p.npy"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, TRPL_AUTOBUILD="0"))
    assert out.returncode == 0 and "RCCL-OK True" in out.stdout, out.stderr[-2000:]
    w = gpu.workloads
    ini, lens = w.power_scan(128)
    want = gpu.loglik(w.samples(301, seed=71), ini, lens, 1.0, 128, 40, [np.full(41, 20.0)] * 3)
    assert np.array_equal(np.load(tmp_path / "p.npy")[0], want)


def test_multi_rank_logic_with_a_stand_in_collective_library(gpu, tmp_path):
    """The N > 1 logic of trpl_loglik_multi_dev on a one-GPU box: three and four "ranks" on device 0
    (trpl_multi_create_ex with TRPL_MULTI_ALLOW_DUPLICATE_DEVICES) with the six RCCL entry points bound to tests/mock_rccl (stream-ordered
    device-to-device copies) instead of librccl -- uneven shards (padded exchange + unpadding with every rank
    index), equal shards (direct exchange), more ranks than samples, per-shard outputs.  Every rank's P[S]
    must equal the single launch bit for bit.  RCCL itself is exercised by the one-rank tests above."""
    so = str(tmp_path / "libmock_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", so,
                           os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp")])
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
import trpl_amd
from trpl_amd import device as tdev, workloads as wl
dev = torch.device("cuda", 0)
ini, lens = wl.power_scan(128)
ini_d = torch.from_numpy(ini).to(dev)
T, Time = 40, 1.0
obs = torch.full((3, T + 1), 20.0, dtype=torch.float64, device=dev) - 0.01 * torch.arange(T + 1, device=dev)
for n, S in ((3, 1000), (4, 1000), (3, 999), (4, 2), (2, 5121)):
    Xh = wl.samples(S, seed=91)
    X = torch.from_numpy(Xh).to(dev)
    P = torch.zeros(S, dtype=torch.float64, device=dev)
    sse = torch.empty((3, S), dtype=torch.float64, device=dev)
    st = torch.empty((3, S), dtype=torch.int32, device=dev)
    it = torch.empty((3, S), dtype=torch.int64, device=dev)
    flags = trpl_amd._abi.pin_variant(0, 3 * S, 128, T)
    tdev.loglik_device(X, ini_d, lens, Time, 128, T, obs, [T + 1] * 3, P, sse, st, it, flags=flags)
    torch.cuda.synchronize()
    with tdev.MultiDevice([0] * n, allow_duplicate_devices=True) as md:
        b = md.shard_bounds(S)
        Xs = [X[lo:hi].contiguous() for lo, hi in b]
        Pf = [torch.full((S,), -7.0, dtype=torch.float64, device=dev) for _ in range(n)]
        ss = [torch.empty((3, hi - lo), dtype=torch.float64, device=dev) for lo, hi in b]
        sts = [torch.empty((3, hi - lo), dtype=torch.int32, device=dev) for lo, hi in b]
        its = [torch.empty((3, hi - lo), dtype=torch.int64, device=dev) for lo, hi in b]
        for _ in range(2):
            md.loglik(Xs, [ini_d] * n, lens, Time, 128, T, [obs] * n, [T + 1] * 3, Pf, sse=ss, status=sts, iters_total=its)
            md.synchronize()
            for r, (lo, hi) in enumerate(b):
                assert torch.equal(Pf[r], P), (n, S, r)
                assert torch.equal(ss[r], sse[:, lo:hi]) and torch.equal(sts[r], st[:, lo:hi]) and torch.equal(its[r], it[:, lo:hi])
print("MOCK-OK")
''' % ROOT
    env = dict(os.environ, TRPL_RCCL_LIBRARY=so, TRPL_AUTOBUILD="0")      # the one environment switch the library reads
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "MOCK-OK" in out.stdout, (out.stdout[-500:], out.stderr[-2500:])


def test_multi_device_call_is_ordered_against_the_callers_torch_stream(gpu):
    """trpl_loglik_multi_dev works on the handle's own streams.  MultiDevice.loglik(order=True) makes them wait for what
    the caller's torch stream holds (trpl_multi_wait_stream) and makes that stream wait for the result
    (trpl_multi_release_stream), on the device: parameters written by a copy that is still QUEUED behind ~0.3 s of
    other work when loglik() is called are the ones the solve reads, and a read of P_full queued right after the call
    sees the gathered vector -- no host synchronisation anywhere in between (round-2 advisor finding)."""
    import torch
    w = gpu.workloads
    dev = torch.device("cuda:0")
    L, T, S = 128, 40, 1500
    Time = T * DT
    ini, lens = w.power_scan(L)
    Xa, Xb = w.samples(S, seed=21), w.samples(S, seed=22)
    ini_d = torch.from_numpy(ini).to(dev)
    obs = torch.full((3, T + 1), 15.0, dtype=torch.float64, device=dev)
    want = {}
    for name, Xh in (("a", Xa), ("b", Xb)):
        Xd = torch.from_numpy(Xh).to(dev)
        P = torch.zeros(S, dtype=torch.float64, device=dev); sse = torch.empty((3, S), dtype=torch.float64, device=dev)
        gpu.device.loglik_device(Xd, ini_d, lens, Time, L, T, obs, [T + 1] * 3, P, sse, flags=gpu.FLAG_KERNEL_SINGLE)
        torch.cuda.synchronize()
        want[name] = P.clone()
    assert not torch.equal(want["a"], want["b"])
    X = torch.from_numpy(Xa).to(dev)
    Xb_pinned = torch.from_numpy(Xb).pin_memory()
    Pf = torch.zeros(S, dtype=torch.float64, device=dev)
    out = torch.empty(S, dtype=torch.float64, device=dev)
    with gpu.device.MultiDevice([0]) as md:
        md.loglik([X], [ini_d], lens, Time, L, T, [obs], [T + 1] * 3, [Pf], flags=gpu.FLAG_KERNEL_SINGLE)      # warm: RCCL channels up
        md.synchronize()
        torch.cuda.synchronize()
        torch.cuda._sleep(int(6e8))                          # ~0.3 s of work ahead of the copy on the torch stream
        X.copy_(Xb_pinned, non_blocking=True)                # still queued when loglik() is called
        md.loglik([X], [ini_d], lens, Time, L, T, [obs], [T + 1] * 3, [Pf], flags=gpu.FLAG_KERNEL_SINGLE)
        out.copy_(Pf)                                        # queued on the torch stream right behind the call
        torch.cuda.synchronize()
        md.synchronize()
    assert torch.equal(out, want["b"])
