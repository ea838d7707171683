#!/usr/bin/env python3
"""Benchmark of the TRPL hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W           (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One "step" = one pass of the fused hot path (trpl_loglik_dev: time-stepping + PL + log-likelihood
of every sample x curve system, then the curve reduction; for N > 1 also the RCCL all-gather of
the per-sample likelihoods) over one batch of synthetic input that is already resident in HBM.

Workload (BASELINE.json configs[1]): Power_scan (3 excitations, thickness 2000 nm) x 65 536
random parameter samples PER GPU (weak scaling; 8 GPUs = configs[3], 524 288 samples), L = 128
nodes, fp64, dt = 0.025 ns, tol 1e-7, MAX 10 000 -- the reference's grid and tolerances.  The
number of time steps per pass is T (default 8000, the sweep length SURVEY 8d allows; the reference's
production run has T = 80 000, i.e. 10x more steps of the same size per likelihood, a 26 s pass).
The headline `value` is therefore the per-step rate, TRPL system-timesteps/s (system = sample x
curve); likelihoods/s at this T and extrapolated to T = 80 000 are reported beside it.  The first
steps after the excitation need many more inner iterations than the rest, so short passes
under-state the full-length rate (3.2 iterations per step at T = 1000, 2.2 at 8000, 2.0 at 80 000):
`--T 80000` runs the full length, `--T 1000` the transient-dominated case of the earlier profiles.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: N fresh child
processes under `python -m torch.distributed.run` (127.0.0.1 rendezvous, a free port), BEFORE this process has
imported torch or touched the GPU; rank 0's JSON line is relayed and the exit status is the children's.  The N > 1
line carries `rccl`: the world RCCL saw, every rank's device, the bytes and the time of the one all-gather.

At N = 1 the line also carries `other_configs` -- ONE event-timed pass each of BASELINE configs[2] (Twothick x 65 536
samples x 6 curves) and of one GPU's share of configs[4] (L = 512 x 32 768 of 262 144 samples: fp64 at tol 1e-7 and 1e-6, and the config
as worded -- fp32 state, labelled "screening", with its PL error against fp64) -- `roofline_hbm_pcr_L512` -- U1 on configs[4]'s 512-node
rows, fp32 and fp64 -- `library` -- which libtrpl_hip.so was measured (source hash; `roofline.traffic_source_stale` says whether the quoted
PMC profile is of that library) -- `host_api_block` -- the reference's own call sequence (pvSim -> fastlog -> prob per
curve, host buffers) on one reference-shaped 1024-sample block, PCIe included -- and `full_length`: ONE extra pass at the production length T = 80 000 over the
same resident batch (event-timed, ~26 s), so that the full-length rate is measured by every driver run.

`--single-process --gpus N` runs the same step with ONE process driving all N devices through
trpl_loglik_multi_dev (RCCL ncclCommInitAll + one ncclAllGather; P left resident on every device).

Rehearsals of the N > 1 path on a one-GPU box (tests/test_gpu_multi.py; none is a measurement): `--samples-total S` (a logical batch
that --gpus need not divide), `--backend gloo` (ranks share the GPU, host-staged collectives), `--rehearse-collectives` (the collective
code path over real RCCL on a one-rank communicator), `--single-process --rehearse-on-device0` (N ranks of one process on device 0
against a stand-in collective library).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      the dominant kernel (the fused time-stepper): achieved fp64 FLOP/s from the
                device's own iteration counters (268*L flop per inner iteration)
                over the kernel's average duration measured with events on its stream, against
                the fp64 vector peak.  It is VALU-bound by construction (state lives in
                registers), so `bound` is "valu-fp64"; `roofline_hbm_pcr` is the HBM roofline of
                the stand-alone batched PCR solve (U1, 5*L*8 B per system), the kernel the
                north-star's ">= 40 % of HBM roofline" target names; its launches rotate over four
                distinct operand sets (1.34 GB, 5x the Infinity Cache) so the rate is an HBM rate.
  cpu_baseline  the CPU oracle (the reference's algorithm restated in C, bit-identical to it)
                timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X fp64 vector: 1/2 of the 157.3 TF fp32 vector peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0               # HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
FLOP_PER_ITER_PER_NODE = 268


def self_launch(argv, n):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as fresh child processes under
    torch.distributed.run and relay what they print.  Called before this process imports torch / trpl_amd or touches
    the GPU; never an exec (a process that has initialised the GPU must not be replaced), never under a profiler."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--T", type=int, default=8000,
                    help="time steps per pass (reference production: 80000; 8000 is the survey's sweep length)")
    ap.add_argument("--samples-per-gpu", type=int, default=65536)
    ap.add_argument("--samples-total", type=int, default=None,
                    help="rehearsals: the logical batch, cut into --gpus contiguous shards by trpl_shard_bounds -- need not be "
                         "divisible by --gpus (default: --samples-per-gpu x --gpus, weak scaling)")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="run the N > 1 code path -- process group, all-gather inside the timed step, max-over-ranks timing, the "
                         "rccl record -- even at --gpus 1: with --backend nccl this is RCCL itself on one GPU (a one-rank "
                         "communicator), the only part of the N > 1 line a one-GPU box can execute on hardware.  Not a measurement")
    ap.add_argument("--rehearse-on-device0", action="store_true",
                    help="--single-process only: all --gpus 'ranks' are device 0 (TRPL_MULTI_ALLOW_DUPLICATE_DEVICES) -- the "
                         "N-rank logic on a one-GPU box; RCCL refuses duplicate devices, so TRPL_RCCL_LIBRARY must name a stand-in "
                         "(tests/mock_rccl).  Not a measurement")
    ap.add_argument("--workload", default="power_scan", choices=["power_scan", "twothick"])
    ap.add_argument("--strict", action="store_true", help="bit-reproducible arithmetic mode")
    ap.add_argument("--L", type=int, default=128, help="spatial nodes (configs[4]: 512)")
    ap.add_argument("--fp32", action="store_true", help="fp32 solver state (configs[4]); implies --tol 3 at L=512, 4 otherwise")
    ap.add_argument("--mixed", action="store_true",
                    help="fp64 state + fp32 correction solves (TRPL_FLAG_MIXED; the accurate path for configs[4])")
    ap.add_argument("--hist32", action="store_true",
                    help="fp64 arithmetic, BDF history as fp32 differences (TRPL_FLAG_HIST32, L = 256 / 512; round-4 experiment)")
    ap.add_argument("--tol", type=int, default=None, help="convergence exponent (default 7, the reference's)")
    ap.add_argument("--extra-flags", type=lambda v: int(v, 0), default=0,
                    help="further TRPL_FLAG_* bits for the timed launches (measurements: 0x20000 = always-isolating paired "
                         "kernel, 0x40000 = adjacent-sample pairing, 0x10 / 0x20 = force the paired / one-system kernel)")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the production-shape end-to-end entry (2^17 samples, real observation file, fused level; ~10 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcr", action="store_true")
    ap.add_argument("--no-host-api", action="store_true",
                    help="skip the reference-API (host-buffer) block: pvSim -> fastlog -> prob on 1024 samples (N = 1 only)")
    ap.add_argument("--no-full-length", action="store_true",
                    help="skip the one extra pass at the reference's production length T = 80000 (N = 1 only, ~26 s)")
    ap.add_argument("--full-length-T", type=int, default=80000)
    ap.add_argument("--traffic-profile", default=None,
                    help="tag of the committed profiles/<tag>_hbm_traffic.json to quote (default: highest round/version)")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives --gpus devices (trpl_multi_* / trpl_loglik_multi_dev: RCCL ncclCommInitAll + "
                         "one ncclAllGather, the likelihood vector resident on every device) instead of one rank per GPU")
    ap.add_argument("--dump-p", default=None, help="rank 0 saves the gathered likelihood vector of the last pass (.npy)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one GPU per rank); gloo = rehearsal of the N>1 control flow "
                         "with host-staged collectives (ranks may share a GPU)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the one pass each of configs[2] (Twothick) and configs[4]'s share (L = 512) (N = 1 only, ~22 s)")
    args = ap.parse_args()

    if args.single_process:
        return main_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    multi = world > 1 or args.rehearse_collectives            # the collective code path (see --rehearse-collectives)

    import trpl_amd
    from trpl_amd import workloads as wl

    # CPU baselines first: they fork worker processes, which must happen before this process
    # initialises the GPU.  Rank 0 at N = 1 only; bounded samples of the same workload.
    cpu_legs = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ini_c, lens_c = wl.power_scan(args.L) if args.workload == "power_scan" else wl.twothick(args.L)
        cpu_legs["cpu_baseline"] = cpu_baseline(wl, trpl_amd, ini_c, lens_c, args.T * 0.025, args.L, args.T, args.cpu_seconds)
        cpu_legs["cpu_baseline_scipy"] = cpu_baseline_scipy(wl, ini_c, lens_c, args.T * 0.025, args.L, args.T)
        rec = cpu_reference_recorded()
        if rec is not None:
            cpu_legs["cpu_reference_recorded"] = rec

    import torch
    import torch.distributed as dist
    from trpl_amd import device as tdev
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")      # where collectives run
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                   # --rehearse-collectives outside a launcher
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    L, T, dt_ns = args.L, args.T, 0.025
    tol = args.tol if args.tol is not None else ((3 if L >= 512 else 4) if args.fp32 else 7)
    Time = T * dt_ns
    ini, lens = wl.power_scan(L) if args.workload == "power_scan" else wl.twothick(L)
    C = len(lens)
    S_total = args.samples_total if args.samples_total is not None else args.samples_per_gpu * world
    lo, hi = trpl_amd.dist.shard_bounds(S_total, world, rank)
    S = hi - lo
    X_host = wl.samples(S_total)[lo:hi]                      # same seeded draw on every rank, own shard
    flags = (trpl_amd.FLAG_STRICT if args.strict else 0) | (trpl_amd.FLAG_MIXED if args.mixed else 0) \
        | ((trpl_amd.FLAG_FP32 | trpl_amd.FLAG_FP32_LONG) if args.fp32 else 0) \
        | (trpl_amd._abi.FLAG_HIST32 if args.hist32 else 0) | args.extra_flags     # (--fp32: a screening-mode number, flagged as such below)
    # the stepper variant is a property of the LOGICAL batch (all ranks' samples), not of this rank's shard:
    # a sample's bits then do not depend on how many GPUs the batch is cut over (include/trpl.h)
    flags = trpl_amd._abi.pin_variant(flags, S_total * C, L, T)

    # ---- inputs resident in HBM before anything is timed ----
    X = torch.from_numpy(np.ascontiguousarray(X_host)).to(dev)
    ini_d = torch.from_numpy(ini).to(dev)
    mark = torch.from_numpy((wl.MARKED_POINT * trpl_amd.UNIT_CONVERSIONS)[None, :-1].copy()).to(dev)
    obs = torch.empty((C, T + 1), dtype=torch.float64, device=dev)
    for c in range(C):                                       # synthetic observations: the solver itself at the marked point
        pl = torch.empty((1, T + 1), dtype=torch.float64, device=dev)
        tdev.solve_pl_device(mark, lens[c], Time, L, T, ini_d[c].contiguous(), pl, flags=trpl_amd.FLAG_STRICT, tol=7)
        obs[c] = torch.log10(pl[0])
    P = torch.zeros(S, dtype=torch.float64, device=dev)
    sse = torch.empty((C, S), dtype=torch.float64, device=dev)
    status = torch.empty((C, S), dtype=torch.int32, device=dev)
    iters = torch.empty((C, S), dtype=torch.int64, device=dev)
    n_obs = [T + 1] * C

    ev, ev_ag = [], []

    def step(record):
        P.zero_()
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        tdev.loglik_device(X, ini_d, lens, Time, L, T, obs, n_obs, P, sse, status, iters, flags=flags, tol=tol)
        if record:
            e1.record()
            ev.append((e0, e1))
        if multi:
            full_ = trpl_amd.dist.gather_likelihoods(P[None, :].to(cdev), S_total)
            if record and args.backend == "nccl":       # solve end -> gathered vector usable: the all-gather + the wait
                e2 = torch.cuda.Event(enable_timing=True)     # for the slowest rank's solve
                e2.record()
                ev_ag.append((e1, e2))
            return full_
        return P[None, :]

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- bookkeeping from the device's own counters ----
    n_fail = int((status != 0).sum().item())
    it_total = int(iters.sum().item())                        # inner iterations in ONE pass on this rank
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if rank == 0 and args.dump_p:
        np.save(args.dump_p, full.detach().cpu().numpy())
    if multi:
        agg = torch.tensor([it_total, n_fail], dtype=torch.float64, device=cdev)
        dist.all_reduce(agg)
        it_all, fail_all = int(agg[0].item()), int(agg[1].item())
    else:
        it_all, fail_all = it_total, n_fail
    assert full.shape == (1, S_total) and bool(torch.isfinite(full).sum() >= S_total - fail_all)

    sys_steps = S_total * C * (T + 1)                         # system-timesteps per pass, all ranks
    lost = (status > 0).to(torch.float64) * (float(T + 1) - status.to(torch.float64))     # status = 1 + failing step
    steps_lost = float(lost.sum().item())
    if multi:
        lt = torch.tensor([steps_lost], dtype=torch.float64, device=cdev)
        dist.all_reduce(lt)
        steps_lost = float(lt.item())
    value = sys_steps * args.steps / elapsed
    flop_launch = it_total * FLOP_PER_ITER_PER_NODE * L       # this rank's launch
    achieved_tf = flop_launch / (kern_ms * 1e-3) / 1e12

    # which time-stepper this launch ran (the library picks the two-systems-per-wavefront kernel for
    # fp64 L = 128 launches that fill the chip)
    variant = trpl_amd._abi.lib().trpl_kernel_variant(S_total * C, L, T, flags)
    if variant == trpl_amd._abi.KERNEL_FAST_PAIR:
        kernel_name = "pair::stepper_pair_kernel (2 x L=128 systems per wavefront; fused time-stepper + likelihood)"
    else:
        kernel_name = "%sstepper_kernel<%d%s> (fused time-stepper + likelihood)" % ("f32::" if args.fp32 else "", L,
                                                                                   ", mixed" if args.mixed else "")
        if args.hist32:
            kernel_name = kernel_name.replace(">", ", fp32-difference history>", 1)
    # the instantiation that ran, as the library names it (no guessing: trpl_kernel_name)
    rocprof_name = "void " + trpl_amd._abi.kernel_name(S_total * C, L, T, flags)
    out = {
        "metric": "TRPL timesteps/sec at %d nodes (system = parameter sample x excitation; fused solve + "
                  "log-likelihood; parameter-sample likelihoods/sec in likelihoods_per_s_*)" % L,
        "value": value,
        "unit": "system-timesteps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 state (SCREENING mode: PL errors of percents over this window), f64 reductions" if args.fp32 else ("f64 state, f32 correction solves" if args.mixed else "f64"),
        "data": "synthetic",
        "config": {"workload": "%s x %d samples/GPU (%d total), %d curves, L=%d nodes, T=%d steps of dt=0.025 ns, "
                               "tol=1e-%d, MAX=10000, %s, arithmetic=%s"
                               % (args.workload, -(-S_total // world), S_total, C, L, T, tol,
                                  "fp32 state" if args.fp32 else ("fp64 state + fp32 solves" if args.mixed else ("fp64, fp32-difference history" if args.hist32 else "fp64")),
                                  "strict" if args.strict else "fast"),
                   "arithmetic": "strict" if args.strict else "fast", "precision": "fp32 state" if args.fp32 else
                   ("fp64 state + fp32 solves" if args.mixed else "fp64"), "tol_exp": tol,
                   "samples_total": S_total, "curves": C, "L": L, "T": T, "parallelism": "sample-shard x%d" % world,
                   "collective": "none" if not multi else ("RCCL all_gather" if args.backend == "nccl" else "gloo all_gather (rehearsal)")},
        "value_n1_equiv": value / world,              # per-GPU rate: what this job's N = 1 line reports as `value`
        "likelihoods_per_s_at_T": S_total * args.steps / elapsed,
        "likelihoods_per_s_at_T80000_equiv": value / (C * 80001),
        "inner_iterations_per_s": it_all * args.steps / elapsed,
        "mean_inner_iterations_per_step": it_all / sys_steps,
        "nonconverged_systems": fail_all,
        # a flagged system stops at its failing step: the steps actually taken, beside the nominal S*C*(T+1)
        "system_timesteps_taken_per_pass": int(sys_steps - steps_lost),
        "roofline": {"kernel": kernel_name, "rocprof_name": rocprof_name,
                     "bound": "valu-fp32" if args.fp32 else "valu-fp64",
                     "achieved": achieved_tf, "peak": FP64_VECTOR_PEAK_TFLOPS * (2 if args.fp32 else 1), "unit": "TFLOP/s",
                     "frac": achieved_tf / (FP64_VECTOR_PEAK_TFLOPS * (2 if args.fp32 else 1)), "traffic": None,
                     "flop_per_launch": flop_launch, "avg_launch_ms": kern_ms,
                     "note": "268*L flop per inner iteration x device-counted iterations; HBM traffic of this "
                             "kernel is ~0.1 KB per system by construction (see profiles/)"},
    }

    if variant == trpl_amd._abi.KERNEL_FAST_PAIR:             # who shares a wavefront (scheduling only: bits do not depend on it)
        tab = [np.full(C, -1, dtype=np.int32) for _ in range(4)]
        ln_ = np.ascontiguousarray(lens, dtype=np.float64)
        no_ = np.full(C, T + 1, dtype=np.int64)
        n_tab = trpl_amd._abi.lib().trpl_pair_table(ln_.ctypes.data, no_.ctypes.data, C, L, T, Time, *[t_.ctypes.data for t_ in tab])
        out["roofline"]["wavefront_pairs"] = ("adjacent samples of one curve" if n_tab <= 0 or (flags & trpl_amd._abi.FLAG_PAIR_ADJACENT) else
                                              [{"curves": [int(tab[0][k]), int(tab[2][k])], "sample_offsets": [int(tab[1][k]), int(tab[3][k])]}
                                               for k in range(n_tab)])
    if multi:
        out["rccl"] = rccl_record(torch, dist, trpl_amd, args, rank, world, local_rank, dev, cdev, S_total, ev_ag)
    if rank == 0 and world == 1 and not args.no_other_configs and args.workload == "power_scan" and L == 128 \
            and not (args.fp32 or args.mixed or args.strict):
        out["other_configs"] = other_configs(torch, tdev, trpl_amd, wl, dev, args.samples_per_gpu, T, dt_ns)
    if rank == 0 and not args.no_pcr:
        out["roofline_hbm_pcr"] = bench_pcr(torch, tdev, dev, flags & trpl_amd.FLAG_STRICT, L=L,
                                            dtype=torch.float32 if args.fp32 else torch.float64)
    if rank == 0 and world == 1 and not args.no_pcr and not args.no_other_configs and L == 128 and not (args.fp32 or args.strict):
        # U1 at configs[4]'s grid ("LDS-pressure / bandwidth stress"): rows of 2 KB (fp32, the config's dtype) and 4 KB (fp64) -- a
        # lane holds 8 adjacent rows, loaded in node order and transposed through LDS (pcr_batched_impl.hpp)
        torch.cuda.empty_cache()
        out["roofline_hbm_pcr_L512"] = [dict(config="configs[4] grid, %s" % name, **bench_pcr(torch, tdev, dev, 0, L=512, dtype=dt))
                                        for name, dt in (("fp32", torch.float32), ("fp64", torch.float64))]
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_full_length and args.full_length_T != T:
        out["full_length"] = full_length_pass(torch, tdev, trpl_amd, dev, X, ini_d, mark, lens, L, args.full_length_T, dt_ns,
                                              flags, tol, C, args.fp32)
    if rank == 0 and world == 1 and not args.no_host_api and L == 128 and not (args.fp32 or args.mixed or args.strict):
        out["host_api_block"] = host_api_block(trpl_amd, wl, ini, lens, L, T, dt_ns)
    if rank == 0 and world == 1 and not args.no_e2e and args.workload == "power_scan" and L == 128 \
            and not (args.fp32 or args.mixed or args.strict):
        out["e2e_production"] = e2e_production(trpl_amd)
    if rank == 0:
        out["library"] = library_record(trpl_amd)
        attach_traffic(out, args.traffic_profile)
    out.update(cpu_legs)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def main_single_process(args):
    """The same step -- fused pass over 65 536 samples per GPU + the gather of the likelihoods -- with ONE process
    driving all N devices through the C ABI's device-resident multi-GPU entry (SURVEY 8e as worded: ncclCommInitAll
    over the node's devices, one ncclAllGather of S/N fp64 per rank, P[S] left on every device).  Inputs are
    resident on their devices before the clock starts; a step ends when every device's stream has drained."""
    import torch
    import trpl_amd
    from trpl_amd import device as tdev
    from trpl_amd import workloads as wl
    n = args.gpus
    rehearse = bool(args.rehearse_on_device0)
    if not rehearse and torch.cuda.device_count() < n:
        raise SystemExit("bench.py --single-process --gpus %d: only %d devices visible" % (n, torch.cuda.device_count()))
    if rehearse and not os.environ.get("TRPL_RCCL_LIBRARY"):
        raise SystemExit("bench.py --rehearse-on-device0: RCCL refuses duplicate devices; name a stand-in collective "
                         "library in TRPL_RCCL_LIBRARY (tests/mock_rccl)")
    ordinals = [0] * n if rehearse else list(range(n))
    L, T, dt_ns = args.L, args.T, 0.025
    tol = args.tol if args.tol is not None else 7
    Time = T * dt_ns
    ini, lens = wl.power_scan(L) if args.workload == "power_scan" else wl.twothick(L)
    C = len(lens)
    S_total = args.samples_total if args.samples_total is not None else args.samples_per_gpu * n
    X_host = wl.samples(S_total)
    flags = (trpl_amd.FLAG_STRICT if args.strict else 0) | (trpl_amd.FLAG_MIXED if args.mixed else 0)
    md = tdev.MultiDevice(ordinals, allow_duplicate_devices=rehearse)
    bounds = md.shard_bounds(S_total)
    Xs, inis, obss, Ps, sses, sts, its = [], [], [], [], [], [], []
    for r, (lo, hi) in enumerate(bounds):
        dev = torch.device("cuda", ordinals[r])
        with torch.cuda.device(dev):
            ini_d = torch.from_numpy(ini).to(dev)
            mark = torch.from_numpy((wl.MARKED_POINT * trpl_amd.UNIT_CONVERSIONS)[None, :-1].copy()).to(dev)
            obs = torch.empty((C, T + 1), dtype=torch.float64, device=dev)
            for c in range(C):
                pl = torch.empty((1, T + 1), dtype=torch.float64, device=dev)
                tdev.solve_pl_device(mark, lens[c], Time, L, T, ini_d[c].contiguous(), pl, flags=trpl_amd.FLAG_STRICT, tol=7)
                obs[c] = torch.log10(pl[0])
            Xs.append(torch.from_numpy(np.ascontiguousarray(X_host[lo:hi])).to(dev))
            inis.append(ini_d); obss.append(obs)
            Ps.append(torch.empty(S_total, dtype=torch.float64, device=dev))
            sses.append(torch.empty((C, hi - lo), dtype=torch.float64, device=dev))
            sts.append(torch.empty((C, hi - lo), dtype=torch.int32, device=dev))
            its.append(torch.empty((C, hi - lo), dtype=torch.int64, device=dev))
            torch.cuda.synchronize(dev)

    def step():
        md.loglik(Xs, inis, lens, Time, L, T, obss, [T + 1] * C, Ps, sse=sses, status=sts, iters_total=its, tol=tol, flags=flags)
        md.synchronize()

    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    it_all = sum(int(t.sum().item()) for t in its)
    fail_all = sum(int((t != 0).sum().item()) for t in sts)
    full = Ps[0].cpu().numpy()
    for r in range(1, n):                                      # the gathered vector is the same on every device
        assert np.array_equal(Ps[r].cpu().numpy(), full)
    assert np.isfinite(full).sum() >= S_total - fail_all
    if args.dump_p:
        np.save(args.dump_p, full[None, :])
    sys_steps = S_total * C * (T + 1)
    value = sys_steps * args.steps / elapsed
    tf = it_all * FLOP_PER_ITER_PER_NODE * L / (elapsed / args.steps) / 1e12
    # what proves the ranks of this form: the communicator's size, each rank's device and PCI bus id, the all-gather payload
    devs = []
    for r, o in enumerate(ordinals):
        props = torch.cuda.get_device_properties(o)
        devs.append({"rank": r, "device": o, "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None),
                     "samples": bounds[r][1] - bounds[r][0]})
    widest = max(hi - lo for lo, hi in bounds)
    rccl = {"world": int(md.n), "backend": "rccl (ncclCommInitAll)" if not rehearse else
            "stand-in collective library %s" % os.path.basename(os.environ["TRPL_RCCL_LIBRARY"]),
            "devices": devs, "distinct_devices": len({(d["device"], d["pci_bus_id"]) for d in devs}),
            "allgather_bytes": widest * 8 * n, "allgather_bytes_per_rank": widest * 8,
            "padded_exchange": len({hi - lo for lo, hi in bounds}) > 1,
            "note": "one ncclAllGather of ceil(S/N) fp64 per rank per step (+ the unpadding pass when shards are unequal); "
                    "its time is inside ms_per_step (single host thread: not separated here)"}
    out = {"metric": "TRPL timesteps/sec at %d nodes (system = parameter sample x excitation; fused solve + "
                     "log-likelihood), one process driving all devices" % L,
           "value": value, "unit": "system-timesteps/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
           "value_n1_equiv": value / n, "rccl": rccl, "rehearsal_on_one_device": rehearse,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "%s x %d samples/GPU (%d total), %d curves, L=%d nodes, T=%d steps of dt=0.025 ns, "
                                  "tol=1e-%d, MAX=10000, fp64" % (args.workload, -(-S_total // n), S_total, C, L, T, tol),
                      "samples_total": S_total, "curves": C, "L": L, "T": T,
                      "parallelism": "single process, sample-shard x%d (trpl_loglik_multi_dev)" % n,
                      "collective": "RCCL ncclAllGather (ncclCommInitAll), P resident on every device"},
           "likelihoods_per_s_at_T": S_total * args.steps / elapsed, "inner_iterations_per_s": it_all * args.steps / elapsed,
           "mean_inner_iterations_per_step": it_all / sys_steps, "nonconverged_systems": fail_all,
           "roofline": {"kernel": "all devices' fused time-steppers", "bound": "valu-fp64", "achieved": tf,
                        "peak": FP64_VECTOR_PEAK_TFLOPS * n, "unit": "TFLOP/s", "frac": tf / (FP64_VECTOR_PEAK_TFLOPS * n),
                        "traffic": None,
                        "note": "wall-clock over the whole step (solve + all-gather + unpadding) on all devices"}}
    md.close()
    print(json.dumps(out), flush=True)


def rccl_record(torch, dist, trpl_amd, args, rank, world, local_rank, dev, cdev, S_total, ev_ag):
    """What proves the ranks: the world size the process group reports, every rank's device (ordinal, name, PCI bus id
    where the runtime exposes it), the payload of the one collective of a step, its time inside the timed loop (from
    the end of this rank's solve to the gathered vector: includes waiting for the slowest rank) and isolated (20
    back-to-back all-gathers of the same payload after a barrier).  Collective on every rank; rank 0 keeps the dict."""
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "local_rank": local_rank, "device": int(dev.index), "name": props.name,
            "pci_bus_id": getattr(props, "pci_bus_id", None), "pid": os.getpid()}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    widest = -(-S_total // world)
    payload = torch.zeros((1, widest), dtype=torch.float64, device=cdev)
    for _ in range(3):
        trpl_amd.dist.gather_likelihoods(payload, S_total)
    iso_us = None
    dist.barrier()
    if args.backend == "nccl":
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            trpl_amd.dist.gather_likelihoods(payload, S_total)
        e1.record()
        torch.cuda.synchronize()
        iso_us = e0.elapsed_time(e1) * 1e3 / 20
    else:
        t0 = time.perf_counter()
        for _ in range(20):
            trpl_amd.dist.gather_likelihoods(payload, S_total)
        iso_us = (time.perf_counter() - t0) * 1e6 / 20
    in_loop = float(np.mean([a.elapsed_time(b) for a, b in ev_ag])) * 1e3 if ev_ag else None
    return {"world": dist.get_world_size(), "backend": dist.get_backend(), "devices": everyone,
            "distinct_devices": len({(d["device"], d["pci_bus_id"]) for d in everyone}),
            "allgather_bytes": widest * 8 * world, "allgather_bytes_per_rank": widest * 8,
            "allgather_us": in_loop if in_loop is not None else iso_us,
            "allgather_us_in_loop_incl_rank_skew": in_loop, "allgather_us_isolated": iso_us,
            "note": "one all_gather_into_tensor of ceil(S/N) fp64 per rank per step + the unpadding copies"}


def one_pass(torch, tdev, trpl_amd, wl, dev, workload, S, L, T, dt_ns, tol, flags=0):
    """ONE event-timed fused pass of another configuration (inputs resident, one short untimed launch first)."""
    Time = T * dt_ns
    ini, lens = wl.power_scan(L) if workload == "power_scan" else wl.twothick(L)
    C = len(lens)
    X = torch.from_numpy(np.ascontiguousarray(wl.samples(S))).to(dev)
    ini_d = torch.from_numpy(ini).to(dev)
    mark = torch.from_numpy((wl.MARKED_POINT * trpl_amd.UNIT_CONVERSIONS)[None, :-1].copy()).to(dev)
    obs = torch.empty((C, T + 1), dtype=torch.float64, device=dev)
    for c in range(C):
        pl = torch.empty((1, T + 1), dtype=torch.float64, device=dev)
        tdev.solve_pl_device(mark, lens[c], Time, L, T, ini_d[c].contiguous(), pl, flags=trpl_amd.FLAG_STRICT, tol=7)
        obs[c] = torch.log10(pl[0])
    P = torch.zeros(S, dtype=torch.float64, device=dev)
    sse = torch.empty((C, S), dtype=torch.float64, device=dev)
    status = torch.empty((C, S), dtype=torch.int32, device=dev)
    iters = torch.empty((C, S), dtype=torch.int64, device=dev)
    flags = trpl_amd._abi.pin_variant(flags, S * C, L, T)
    tdev.loglik_device(X, ini_d, lens, 50 * dt_ns, L, 50, obs[:, :51].contiguous(), [51] * C, P, sse, status, iters, flags=flags, tol=tol)
    P.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    tdev.loglik_device(X, ini_d, lens, Time, L, T, obs, [T + 1] * C, P, sse, status, iters, flags=flags, tol=tol)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    it = int(iters.sum().item())
    tf = it * FLOP_PER_ITER_PER_NODE * L / (ms * 1e-3) / 1e12
    variant = trpl_amd._abi.lib().trpl_kernel_variant(S * C, L, T, flags)
    fp32 = bool(flags & trpl_amd.FLAG_FP32)
    peak = FP64_VECTOR_PEAK_TFLOPS * (2 if fp32 else 1)
    return {"workload": "%s x %d samples, %d curves, L=%d nodes, T=%d steps of dt=0.025 ns, tol=1e-%d, %s, arithmetic=fast"
                        % (workload, S, C, L, T, tol, "fp32 state" if fp32 else "fp64"),
            "samples": S, "curves": C, "L": L, "T": T, "tol_exp": tol, "passes": 1, "ms": ms,
            "system_timesteps_per_s": S * C * (T + 1) / (ms * 1e-3), "likelihoods_per_s_at_T": S / (ms * 1e-3),
            "inner_iterations": it, "mean_inner_iterations_per_step": it / (S * C * (T + 1)),
            "roofline_achieved_tflops": tf, "roofline_peak_tflops": peak, "roofline_frac": tf / peak,
            "kernel": ("pair::stepper_pair_kernel" if variant == trpl_amd._abi.KERNEL_FAST_PAIR else
                       "%sstepper_kernel<%d>" % ("f32::" if fp32 else "", L)),
            "rocprof_name": "void " + trpl_amd._abi.kernel_name(S * C, L, T, flags),
            "nonconverged": int((status != 0).sum().item()), "finite_likelihoods": int(torch.isfinite(P).sum().item())}


def fp32_pl_error(torch, tdev, trpl_amd, wl, dev, L, T, dt_ns, n_sub=256, tol32=3):
    """What the fp32-STATE stepper (configs[4] as worded: TRPL_FLAG_FP32 | _FP32_LONG, tol 1e-3) does to PL(t): its PL on an
    n_sub-sample subsample of the same seeded batch against the fp64 tol-7 stepper's, relative, over every stored column of
    the window -- the reason the driver's record answers configs[4] in fp64 (pvSimPCR.py:10-11 is fp64 only)."""
    Time = T * dt_ns
    ini, lens = wl.power_scan(L)
    X = torch.from_numpy(np.ascontiguousarray(wl.samples(n_sub)[:, :12])).to(dev)
    ini_d = torch.from_numpy(ini).to(dev)
    worst = []
    f32 = trpl_amd.FLAG_FP32 | trpl_amd.FLAG_FP32_LONG
    for c in range(len(lens)):
        pl64 = torch.empty((n_sub, T + 1), dtype=torch.float64, device=dev)
        pl32 = torch.empty((n_sub, T + 1), dtype=torch.float64, device=dev)
        st64 = torch.empty(n_sub, dtype=torch.int32, device=dev)
        st32 = torch.empty(n_sub, dtype=torch.int32, device=dev)
        tdev.solve_pl_device(X, lens[c], Time, L, T, ini_d[c].contiguous(), pl64, status=st64, tol=7)
        tdev.solve_pl_device(X, lens[c], Time, L, T, ini_d[c].contiguous(), pl32, status=st32, tol=tol32, flags=f32)
        torch.cuda.synchronize()
        ok = ((st64 == 0) & (st32 == 0))[:, None] & (pl64 > 0) & torch.isfinite(pl32)
        rel = torch.where(ok, (pl32 / pl64 - 1).abs(), torch.zeros_like(pl64))
        worst.append({"curve": c, "max": float(rel.max().item()), "median": float(rel[ok].median().item()),
                      "median_last_column": float(rel[:, -1][ok[:, -1]].median().item()),
                      "max_at_column": int(rel.max(dim=0).values.argmax().item()),
                      "flagged_fp32": int((st32 != 0).sum().item()), "flagged_fp64": int((st64 != 0).sum().item())})
    return {"against": "fp64 stepper, tol 1e-7, the same %d samples (seeded batch's first rows) x %d curves, every PL column of the "
                       "T = %d window" % (n_sub, len(lens), T),
            "max": max(w["max"] for w in worst), "median": float(np.median([w["median"] for w in worst])),
            "per_curve": worst}


def other_configs(torch, tdev, trpl_amd, wl, dev, S, T, dt_ns):
    """The other single-GPU configurations BASELINE.json names, one event-timed pass each at the headline's window:
    configs[2] Twothick (311 / 2000 nm films x 3 powers = 6 curves) x S samples, and ONE GPU's share of configs[4]
    (L = 512 x 262 144 samples over 8 GPUs = 32 768 per GPU, `share_of` names the whole) in fp64 at the reference's
    tolerance 1e-7 and at tol 1e-6 -- the setting DESIGN.md section 7 recommends for that grid (an fp32 STATE, as the
    config is worded, loses the decay over thousands of steps)."""
    out = [dict(config="configs[2]", **one_pass(torch, tdev, trpl_amd, wl, dev, "twothick", S, 128, T, dt_ns, 7))]
    torch.cuda.empty_cache()
    # configs[4]: 262 144 samples over 8 GPUs; at the reference's tolerance (1e-7) and at the recommended 1e-6
    for tol in (7, 6):
        torch.cuda.empty_cache()
        out.append(dict(config="configs[4], one GPU's share of 8", share_of=262144, n_gpus_of_config=8,
                        **one_pass(torch, tdev, trpl_amd, wl, dev, "power_scan", 32768, 512, T, dt_ns, tol)))
    # configs[4] AS WORDED (fp32): the fp32-state stepper over the same window -- a SCREENING mode (include/trpl.h,
    # TRPL_FLAG_FP32): refused beyond TRPL_FP32_MAX_STEPS steps without _FP32_LONG.  Recorded with its PL error so that the
    # record itself says why the entries above are fp64.
    torch.cuda.empty_cache()
    f32 = trpl_amd.FLAG_FP32 | trpl_amd.FLAG_FP32_LONG
    out.append(dict(config="configs[4] as worded (fp32 state), one GPU's share of 8", share_of=262144, n_gpus_of_config=8,
                    screening=True, dtype="f32 state, f64 reductions",
                    pl_rel_err_vs_fp64=fp32_pl_error(torch, tdev, trpl_amd, wl, dev, 512, T, dt_ns),
                    **one_pass(torch, tdev, trpl_amd, wl, dev, "power_scan", 32768, 512, T, dt_ns, 3, flags=f32)))
    return out


def full_length_pass(torch, tdev, trpl_amd, dev, X, ini_d, mark, lens, L, T, dt_ns, flags, tol, C, fp32):
    """ONE pass at the reference's production length (parallel_bayes_gpu.py:75: T = 80 000 steps of 0.025 ns)
    over the same resident batch, timed with events on the launch stream: the headline K-step loop uses a
    shorter window because K x 26 s would not fit the driver's time limit; the per-step rate of the short
    window UNDER-states this one (more of it is the stiff start of the decay)."""
    Time = T * dt_ns
    S = X.shape[0]
    obs = torch.empty((C, T + 1), dtype=torch.float64, device=dev)
    for c in range(C):
        pl = torch.empty((1, T + 1), dtype=torch.float64, device=dev)
        tdev.solve_pl_device(mark, lens[c], Time, L, T, ini_d[c].contiguous(), pl, flags=trpl_amd.FLAG_STRICT, tol=7)
        obs[c] = torch.log10(pl[0])
    P = torch.zeros(S, dtype=torch.float64, device=dev)
    sse = torch.empty((C, S), dtype=torch.float64, device=dev)
    status = torch.empty((C, S), dtype=torch.int32, device=dev)
    iters = torch.empty((C, S), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    tdev.loglik_device(X, ini_d, lens, Time, L, T, obs, [T + 1] * C, P, sse, status, iters, flags=flags, tol=tol)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = e0.elapsed_time(e1)
    it = int(iters.sum().item())
    peak = FP64_VECTOR_PEAK_TFLOPS * (2 if fp32 else 1)
    tf = it * FLOP_PER_ITER_PER_NODE * L / (ms * 1e-3) / 1e12
    return {"T": T, "passes": 1, "ms": ms, "wall_ms": wall * 1e3,
            "system_timesteps_per_s": S * C * (T + 1) / (ms * 1e-3), "likelihoods_per_s": S / (ms * 1e-3),
            "inner_iterations": it, "mean_inner_iterations_per_step": it / (S * C * (T + 1)),
            "nonconverged_systems": int((status != 0).sum().item()),
            "roofline_achieved_tflops": tf, "roofline_frac": tf / peak,
            "finite_likelihoods": int(torch.isfinite(P).sum().item())}


def host_api_block(trpl_amd, wl, ini, lens, L, T, dt_ns, S=1024):
    """The reference's own call sequence on one reference-shaped block (parallel_bayes_gpu.py:104: sims_per_gpu =
    1024; bayeslib.py:137-196): per curve pvSim -> fastlog -> prob through the HOST-BUFFER entry points, float32 PL
    buffer, observations on the whole grid, PCIe and host work included -- the rate a caller gets who swaps only the
    three callables (the fused path above is the one to use)."""
    X = wl.samples(S)
    Time = T * dt_ns
    par = [float(lens[0]), Time, L, T, 1, (0,), 7, 10000]
    vals = np.full(T + 1, -3.0)
    mag = np.ascontiguousarray(X[:, -1])
    pl = np.empty((S, T + 1), dtype=np.float32)
    P = np.zeros(S)
    trpl_amd.pvSim(pl[:8], None, None, None, X[:8, :-1], par, ini[0], init_mode="points")       # context warm-up
    t_solve = t_log = t_prob = 0.0
    t0 = time.perf_counter()
    for c in range(len(lens)):
        par[0] = float(lens[c])
        a = time.perf_counter()
        trpl_amd.pvSim(pl, None, None, None, X[:, :-1], par, ini[c], init_mode="points")
        b = time.perf_counter()
        trpl_amd.fastlog(pl, sys.float_info.min)
        c2 = time.perf_counter()
        trpl_amd.prob(P, pl, vals, None, mag)
        d = time.perf_counter()
        t_solve += b - a; t_log += c2 - b; t_prob += d - c2
    wall = time.perf_counter() - t0
    return {"samples": S, "curves": len(lens), "T": T, "pl_dtype": "float32", "wall_s": wall, "pvSim_s": t_solve,
            "fastlog_s": t_log, "prob_s": t_prob, "system_timesteps_per_s": S * len(lens) * (T + 1) / wall,
            "finite_likelihoods": int(np.isfinite(P).sum())}


def e2e_production(trpl_amd, S=2 ** 17, T=80000, Time=2000.0):
    """The reference's production shape through the fused integration level (tools/e2e_production.py, level A): the
    entry script's configuration (parallel_bayes_gpu.py:72-131: S = 2^17 samples of the shipped box, seed 42, L = 128,
    T = 80 000 steps over 2000 ns, tol 7) with the shipped Balancedhighsurf observation file read by dataio.get_data
    (tests/golden/obs_balanced_full.csv.gz: 5601 / 8801 / 12801 points -- the fused window ends at the last observation),
    driver.bayes with one fused launch, export of <name>_BAYRAN_{P,X}.npy.  Wall time from bayes() to the files on disk."""
    import gzip
    import shutil
    import tempfile
    from trpl_amd import sampler as sm
    gold = os.path.join(ROOT, "tests", "golden")
    src = os.path.join(gold, "obs_balanced_full.csv.gz")
    if not os.path.isfile(src):
        return {"skipped": "tests/golden/obs_balanced_full.csv.gz not found"}
    work = tempfile.mkdtemp(prefix="trpl_e2e_")
    try:
        obs_csv = os.path.join(work, "Balancedhighsurf_Power_scan_Observations.csv")
        with gzip.open(src, "rb") as fh, open(obs_csv, "wb") as out:
            out.write(fh.read())
        ic_flags = {"time_cutoff": Time, "select_obs_sets": None, "noise_level": None}
        sim_flags = {"load_PL_from_file": False, "override_equal_auger": False, "override_equal_mu": False,
                     "override_equal_s": False, "log_pl": True, "self_normalize": False, "random_sample": True, "num_points": S}
        ini = trpl_amd.get_initpoints(os.path.join(gold, "exc_power_scan.csv"), ic_flags)
        e_data = trpl_amd.get_data([obs_csv], ic_flags, sim_flags, scale_f=1e-23)
        n_obs = [len(t) for t in e_data[0][0]]
        simPar = [2000.0, Time, 128, T, 1, (0, 1, 3, 10, 30, 100), 7, 10000]
        gpu_info = {"sims_per_gpu": S, "num_gpus": 1, "has_GPU": True, "max_sims_per_block": 1, "fused": True}
        np.random.seed(42)                                                   # parallel_bayes_gpu.py:35
        t0 = time.perf_counter()
        N, P, X = trpl_amd.bayes(trpl_amd.pvSim, None, None, sm.DEFAULT_MINX * sm.UNIT_CONVERSIONS,
                                 sm.DEFAULT_MAXX * sm.UNIT_CONVERSIONS, sm.DEFAULT_DO_LOG, ini, simPar, e_data, sim_flags, gpu_info)
        trpl_amd.export(os.path.join(work, "out"), P[0], X / sm.UNIT_CONVERSIONS)     # :194-198
        wall = time.perf_counter() - t0
        steps = sum(n - 1 for n in n_obs) + 3
        # the entry point the fused level routed this experiment to (observation times that are a prefix of the
        # simulation grid take the on-grid entry: driver.fused_entry_point)
        entry = trpl_amd.driver.fused_entry_point(e_data[0][0], np.linspace(0, Time, T + 1), 3,
                                                  bool(gpu_info.get("interpolate_prefix", False)))
        return {"integration": "driver.bayes, one fused launch (%s); level A of tools/e2e_production.py" % entry,
                "entry_point": entry,
                "samples": S, "curves": 3, "L": 128, "T": T, "time_ns": Time, "n_obs": n_obs,
                "observations": "Balancedhighsurf_Power_scan_Observations.csv (shipped example data) via dataio.get_data",
                "wall_to_npy_s": wall, "likelihoods_per_s": S / wall, "system_timesteps_per_s": S * steps / wall,
                "finite_likelihoods": int(np.isfinite(P[0]).sum()), "files": sorted(os.listdir(os.path.join(work, "out"))),
                "other_levels": "profiles/r5_e2e_production.json (the reference's call sequence at sims_per_gpu 1024 / 16384, "
                                "FAST vs STRICT, oracle subsample)"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def library_record(trpl_amd):
    """Which libtrpl_hip.so this line was measured with: the hash of the sources it was linked from (the Makefile's
    libtrpl_hip.so.srchash; parse_rocprof.py copies it into profiles/<tag>_hbm_traffic.json, attach_traffic compares)."""
    A = trpl_amd._abi
    try:
        with open(A.LIB_PATH + ".srchash") as fh:
            h = fh.read().strip()
    except OSError:
        h = None
    return {"path": os.path.relpath(A.LIB_PATH, ROOT), "srchash": h, "sources_current": bool(h) and h == A.source_hash(),
            "abi_version": int(A.lib().trpl_abi_version()), "experimental_steppers": bool(A.has_experimental())}


def _profile_key(path):
    """Sort key of profiles/<tag>_hbm_traffic.json by the numbers in its tag (r2_v10 > r2_v9 > r1_v9)."""
    import re
    tag = os.path.basename(path)[:-len("_hbm_traffic.json")]
    return [int(n) for n in re.findall(r"\d+", tag)], tag


def attach_traffic(out, tag=None):
    """`traffic` = HBM bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in
    separate runs, gfx950 x2 read correction) of tools/profile_round.sh, as committed under
    profiles/<tag>_hbm_traffic.json.  PMC counters cannot be read from inside this process, so the figure
    is that of ANOTHER run of the same kernels -- the profile named in `traffic_source` (chosen by tag:
    --traffic-profile, else the highest round/version number; never by file time) -- not of the run being
    timed; null if none is committed.  The stepper's traffic scales with the profile's T (the observation
    stream, ~8 B per system-step through L2); the PCR's is size-matched (65 536 x 128).
    `traffic_source_stale`: the profile records the source hash of the library it ran (`_meta.srchash`); true when that
    differs from the hash of the library loaded HERE (or the profile predates the field) -- the quoted figure then
    belongs to other kernels than the ones being timed and the profile should be re-taken (tools/profile_round.sh)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), key=_profile_key)
    if tag is not None:
        files = [f for f in files if os.path.basename(f) == tag + "_hbm_traffic.json"]
    if not files:
        return
    t = json.load(open(files[-1]))
    src = os.path.basename(files[-1])
    meta = t.get("_meta", {})
    here = (out.get("library") or {}).get("srchash")
    stale = not (here and meta.get("srchash") == here)
    # kernel names gained a template argument in round 2: accept the profile's spelling of the same kernel
    def find(name):
        base = name.split("<")[0]
        hits = [k for k in t if k != "_meta" and k.split("<")[0] == base and (k == name or name.startswith(k[:-1]) or k.startswith(name[:-1]))]
        return t[hits[0]] if hits else None
    for key, obj in ((out["roofline"]["rocprof_name"], "roofline"),
                     ("void trpl::pcr_batched_kernel<double, 128, false>", "roofline_hbm_pcr")):
        rec = find(key)
        if rec and obj in out and (obj == "roofline" or "double,128" in out[obj]["kernel"]):
            out[obj]["traffic"] = rec["hbm_bytes_per_launch"]
            out[obj]["traffic_source"] = "from profile %s (profiles/%s%s): a separate rocprofv3 --pmc run of the same " \
                                         "kernels, not this timed run" % (src[:-len("_hbm_traffic.json")], src,
                                                                          ", T=%s" % meta["T"] if "T" in meta else "")
            out[obj]["traffic_source_stale"] = stale
            out[obj]["traffic_source_srchash"] = meta.get("srchash")


def bench_pcr(torch, tdev, dev, flags, S=65536, L=128, reps=48, dtype=None, nsets=4):
    """U1: stand-alone batched PCR tridiagonal solve, HBM -> HBM, 5*L*w B per system.  The launches rotate
    over `nsets` DISTINCT operand sets (4 x 335 MB = 1.34 GB at the default size, 5x the 256 MiB Infinity
    Cache), so no launch can find its operands cached by the previous one: the rate is an HBM rate."""
    dtype = dtype or torch.float64
    w = 8 if dtype == torch.float64 else 4
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    # The five arrays of a set are views into one allocation, each placed `skew` bytes past a multiple of the
    # array size: operands that sit at the same offset modulo a large power of two (what five separate
    # 64 MiB allocations give) send a wave's four loads to the same HBM channel, which costs ~6 % here
    # (tools/pcr_layout_probe.py: 5.26 TB/s aligned, 5.56-5.58 TB/s from a 4 KiB skew on); include/trpl.h
    # recommends the skew to callers.
    skew = (4096 + 256) // w
    n = S * L
    sets = []
    for _ in range(nsets):
        buf = torch.empty(5 * (n + skew), dtype=dtype, device=dev)
        ld, d, ud, b, x = (buf[i * (n + skew): i * (n + skew) + n].view(S, L) for i in range(5))
        ld.copy_(torch.rand((S, L), dtype=dtype, device=dev, generator=g) * 2 - 1)
        ud.copy_(torch.rand((S, L), dtype=dtype, device=dev, generator=g) * 2 - 1)
        d.copy_(torch.rand((S, L), dtype=dtype, device=dev, generator=g) * 1.5 + 2.5)
        b.copy_(torch.randn((S, L), dtype=dtype, device=dev, generator=g))
        ld[:, 0] = 0
        ud[:, -1] = 0
        sets.append((ld, d, ud, b, x))
    for i in range(2 * nsets):
        ld, d, ud, b, x = sets[i % nsets]
        tdev.pcr_solve_device(ld, d, ud, b, x, flags=flags)
    torch.cuda.synchronize()
    reps -= reps % nsets
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        ld, d, ud, b, x = sets[i % nsets]
        tdev.pcr_solve_device(ld, d, ud, b, x, flags=flags)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # residual check on every operand set so a fast wrong kernel cannot hide
    res = 0.0
    for ld, d, ud, b, x in sets:
        r = d * x
        r[:, 1:] += ld[:, 1:] * x[:, :-1]
        r[:, :-1] += ud[:, :-1] * x[:, 1:]
        res = max(res, float((r - b).abs().max().item()))
        del r
    nbytes = 5 * L * w * S
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "pcr_batched_kernel<%s,%d>" % ("double" if w == 8 else "float", L), "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "systems": S, "L": L,
            "bytes_per_launch": nbytes, "avg_launch_ms": ms, "systems_per_s": S / (ms * 1e-3),
            "operand_sets": nsets, "operand_bytes_rotated": nsets * nbytes, "launches_timed": reps,
            "operand_skew_bytes": skew * w,
            "max_abs_residual": res}


def cpu_model():
    """The host CPU's model string (SURVEY 8d asks for it beside the core count)."""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(wl, trpl_amd, ini, lens, Time, L, T, budget_s):
    """The CPU oracle (kind "port": the reference's algorithm restated in C, pinned bit-exact to
    it) on this host's cores, on a bounded sample of the SAME workload (same box, seed, curves,
    grid, T): PL solve + log10 + squared error per system, as in the timed GPU pass."""
    import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    oracle.load()

    def run(n):
        Xs = wl.samples(max(n, 1))[:n]
        t0 = time.perf_counter()
        for c in range(len(lens)):
            r = oracle.pvsim(Xs[:, :-1], lens[c], Time, L, T, ini[c], nthreads=cores)
            pl = r["plI"]
            oracle.fastlog(pl)
            Pc = np.zeros(n)
            oracle.prob(Pc, pl, np.zeros(T + 1), np.ascontiguousarray(Xs[:, -1]))
        return time.perf_counter() - t0

    n1 = cores
    t1 = run(n1)
    n2 = int(min(max(n1, n1 * budget_s / max(t1, 1e-3)), 64 * n1))
    t2 = run(n2)
    rate = n2 * len(lens) * (T + 1) / t2
    return {"value": rate, "unit": "system-timesteps/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": "%d seeded samples of the same box x %d curves x T=%d steps (%.1f s on %d OpenMP threads)"
                      % (n2, len(lens), T, t2, cores),
            "likelihoods_per_s_at_T": n2 / t2}


def cpu_baseline_scipy(wl, ini, lens, Time, L, T):
    """The "scipy CPU path" named by the north star: a port (oracle/scipy_mol.py, pinned to the
    output of the reference's pvSim_fallback) of the reference's CPU model -- method of lines +
    scipy solve_ivp(BDF) per system + Simpson PL + log10 + squared error -- on one worker process
    per host core.  A different, adaptive-step algorithm: a timing baseline, not a parity target."""
    from oracle import scipy_mol
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = max(1, (2 * cores) // len(lens))
    X = wl.samples(n)
    sec, nsys = scipy_mol.timed_batch(X, ini, lens, Time, L, T, cores)
    return {"value": nsys * (T + 1) / sec, "unit": "system-timesteps/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": "%d seeded samples x %d curves, scipy solve_ivp(BDF, rtol 1e-5) sampled on T=%d output steps "
                      "(%.1f s on %d processes)" % (n, len(lens), T, sec, cores),
            "likelihoods_per_s_at_T": n / sec}


def cpu_reference_recorded():
    """What the REFERENCE'S OWN CPU path measured for BASELINE.json configs[0] (Power_scan x 64 samples through
    bayeslib.bayes(pvSim_fallback.pvSim_cpu_fallback), has_GPU False, 8 SLURM-style array tasks) where the reference can run --
    the development container -- as recorded in tests/golden/fallback64.npz by oracle/gen_golden.py case_fallback64.  A recorded
    figure of another machine, NOT timed in this run (the reference never travels to the GPU box); `cpu_baseline_scipy` is its
    pinned port timed here."""
    path = os.path.join(ROOT, "tests", "golden", "fallback64.npz")
    if not os.path.isfile(path):
        return None
    g = np.load(path)
    out = {"kind": "reference", "measured": "development container, recorded in tests/golden/fallback64.npz", "cores": int(g["ntasks"]),
           "cpu_model": str(g["cpu_model"]), "samples": 64, "curves": 3, "unit": "system-timesteps/s"}
    for tag, name in (("w8k", "bench_window"), ("full", "full_window")):
        T, wall = int(g[tag + "_T"]), float(g[tag + "_wall"])
        out[name] = {"T": T, "wall_s": wall, "value": 64 * 3 * (T + 1) / wall, "likelihoods_per_s": 64 / wall,
                     "model_seconds_per_system": float(g[tag + "_model_seconds"].sum()) / 192}
    return out


if __name__ == "__main__":
    main()
