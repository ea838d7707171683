"""Checkpoint / continue (the reference's init_mode="continue", a stub at pvSimPCR.py:357-358 whose intent the
commented block :294-306 shows): a window cut at step t0, checkpointed as five raw time levels and continued, gives
the uninterrupted run bit for bit -- PL columns, later snapshots, status and iteration totals -- in every arithmetic
mode; and the STRICT continuation is the ORACLE's uninterrupted run bit for bit."""
import numpy as np
import pytest

from gpu_common import needs_experimental

pytestmark = pytest.mark.gpu

MODES = [dict(strict=True), dict(kernel="single"), dict(kernel="pair"), dict(mixed=True)]
DT = 2.0 ** -5          # a power of two: a segment of t0 steps over t0 * DT ns has the window's time step exactly


def _case(gpu, S, seed, L=128):
    w = gpu.workloads
    ini, lens = w.twothick(L)
    return w.samples(S, seed=seed)[:, :12], lens[0], ini[0]


def _split_run(gpu, X, length, Time, L, T, ini, t0, plT=1, late=(), **mode):
    """(uninterrupted, continued): each (pl, status, iters, snapshots-at-late-steps)."""
    S = len(X)
    full_snaps = {}
    full = gpu.solve_pl(X, length, Time, L, T, ini, plT=plT, snap_steps=list(late) or None,
                        snapshots=full_snaps if late else None, **mode)
    # segment 1: to t0, recording the five newest levels raw.  Same time step: Time * t0 / T.
    ck = {}
    pl_a, st_a, it_a, _ = gpu.solve_pl(X, length, Time * t0 / T, L, t0, ini, plT=plT,
                                       snap_steps=gpu.checkpoint_steps(t0), snapshots=ck, snap_raw=True, **mode)
    # segment 2: the caller's PL matrix carries the columns of segment 1
    out = np.full((S, T // plT + 1), np.nan)
    out[:, :pl_a.shape[1]] = pl_a
    got_snaps = {}
    pl_b, st_b, it_b, _ = gpu.solve_pl(X, length, Time, L, T, None, plT=plT, out=out,
                                       resume=(t0, ck["plN"], ck["plP"], ck["plE"]), snap_steps=list(late) or None,
                                       snapshots=got_snaps if late else None, **mode)
    # the time loop runs t = 0 .. T inclusive (pvSimPCR.py:237: the step taken at t = T is computed and dropped), so
    # segment 1 has already counted the step at t0 that the continuation takes again: count it alone and subtract
    tail = np.full((S, t0 // plT + 1), np.nan)
    _, _, it_c, _ = gpu.solve_pl(X, length, Time * t0 / T, L, t0, None, plT=plT, out=tail,
                                 resume=(t0, ck["plN"], ck["plP"], ck["plE"]), **mode)
    return (full[0], full[1], full[2], full_snaps), (pl_b, st_a, st_b, it_a + it_b - it_c, got_snaps)


@pytest.mark.parametrize("mode", MODES, ids=lambda m: "-".join("%s=%s" % kv for kv in m.items()))
@pytest.mark.parametrize("t0", [4, 57, 96])
def test_continued_run_is_the_uninterrupted_run_bit_for_bit(gpu, mode, t0):
    if mode.get("mixed"):
        needs_experimental(gpu, dict(mixed=True))
    X, length, ini = _case(gpu, 9, seed=5)
    T = 160
    Time = T * DT
    assert (Time * t0 / T) / t0 == Time / T                     # the two segments share dt exactly
    late = (t0, t0 + 1, 130, T)
    (pl, st, it, sn), (pl2, st_a, st_b, it2, sn2) = _split_run(gpu, X, length, Time, 128, T, ini, t0, late=late, **mode)
    assert not st.any() and not st_a.any() and not st_b.any()
    assert not np.isnan(pl2).any()
    assert np.array_equal(pl, pl2)
    assert np.array_equal(it, it2)
    for k in ("plN", "plP", "plE"):
        assert np.array_equal(sn[k], sn2[k]), k


def test_strict_continuation_is_the_oracles_uninterrupted_run(gpu, oracle):
    X, length, ini = _case(gpu, 5, seed=77)
    T, Time, t0 = 120, 120 * DT, 48
    want = oracle.pvsim(X, length, Time, 128, T, ini, snap_steps=[100])
    _, (pl2, st_a, st_b, it2, sn2) = _split_run(gpu, X, length, Time, 128, T, ini, t0, late=(100,), strict=True)
    assert np.array_equal(pl2, want["plI"]) and np.array_equal(it2, want["iters_total"])
    for k in ("plN", "plP", "plE"):
        assert np.array_equal(sn2[k], want[k]), k


@pytest.mark.parametrize("mode", [dict(strict=True), dict(kernel="pair")], ids=["strict", "pair"])
def test_continue_with_decimated_pl_and_an_odd_batch(gpu, mode):
    """plT = 8 with t0 off the PL grid: the first column written by the continuation is the next multiple of plT;
    columns before it keep the caller's values.  An odd batch leaves the paired kernel a half-empty wavefront."""
    if mode.get("mixed"):
        needs_experimental(gpu, dict(mixed=True))
    X, length, ini = _case(gpu, 11, seed=9)
    T, Time, t0, plT = 256, 256 * DT, 99, 8
    (pl, st, it, _), (pl2, _, st_b, it2, _) = _split_run(gpu, X, length, Time, 128, T, ini, t0, plT=plT, **mode)
    assert np.array_equal(pl, pl2) and np.array_equal(it, it2) and not st_b.any()
    # columns before ceil(t0 / plT) are not touched by the continuation
    out = np.full((len(X), T // plT + 1), -3.0)
    ck = {}
    gpu.solve_pl(X, length, Time * t0 / T, 128, t0, ini, plT=plT, snap_steps=gpu.checkpoint_steps(t0), snapshots=ck,
                 snap_raw=True, **mode)
    gpu.solve_pl(X, length, Time, 128, T, None, plT=plT, out=out, resume=(t0, ck["plN"], ck["plP"], ck["plE"]), **mode)
    first = -(-t0 // plT)
    assert (out[:, :first] == -3.0).all() and np.array_equal(out[:, first:], pl[:, first:])


def test_three_segments_chain_through_the_dropin_signature(gpu):
    """pvSim(init_mode="continue"): iniPar carries the checkpoint; three segments equal one run."""
    X, length, ini = _case(gpu, 6, seed=21)
    L, T, Time = 128, 240, 240 * DT
    S = len(X)
    whole = np.empty((S, T + 1))
    gpu.pvSim(whole, None, None, None, X, [length, Time, L, T, 1, None, 7, 10000], ini, init_mode="points")
    plI = np.full((S, T + 1), np.nan)
    cuts = [80, 160, T]
    state = None
    for i, t1 in enumerate(cuts):
        # each call is told the FULL window (same dt) but stops at t1: the time-step count of the call is t1
        seg = np.full((S, t1 + 1), np.nan)
        seg[:, :0 if state is None else state[0] + 1] = plI[:, :0 if state is None else state[0] + 1]
        ck = {}
        steps = gpu.checkpoint_steps(t1)
        if state is None:
            gpu.solve_pl(X, length, Time * t1 / T, L, t1, ini, out=seg, snap_steps=steps, snapshots=ck, snap_raw=True)
        else:
            plN = np.zeros((S, 5, L)); plP = np.zeros((S, 5, L)); plE = np.zeros((S, 5, L + 1))
            # raw snapshots need the flag, which the drop-in signature has no slot for: the last segment goes
            # through pvSim, the middle one through solve_pl
            if t1 == T:
                gpu.pvSim(seg, None, None, None, X, [length, Time * t1 / T, L, t1, 1, None, 7, 10000], state,
                          init_mode="continue")
            else:
                gpu.solve_pl(X, length, Time * t1 / T, L, t1, None, out=seg, resume=state, snap_steps=steps,
                             plN=plN, plP=plP, plE=plE, snap_raw=True)
                ck = dict(plN=plN, plP=plP, plE=plE)
        plI[:, :t1 + 1] = seg
        if ck:
            state = (t1, ck["plN"], ck["plP"], ck["plE"])
    assert np.array_equal(plI, whole)


def test_device_resident_checkpoint_and_continue(gpu):
    import torch
    dv = gpu.device
    X, length, ini = _case(gpu, 300, seed=3)
    L, T, Time, t0 = 128, 96, 96 * DT, 40
    S = len(X)
    dev = torch.device("cuda:0")
    tX = torch.from_numpy(X).to(dev); tini = torch.from_numpy(np.ascontiguousarray(ini)).to(dev)
    fl = gpu.FLAG_KERNEL_PAIR
    full = torch.empty((S, T + 1), dtype=torch.float64, device=dev)
    it_full = torch.zeros(S, dtype=torch.int64, device=dev)
    dv.solve_pl_snap_device(tX, length, Time, L, T, tini, full, [], iters_total=it_full, flags=fl)
    cN = torch.zeros((S, 5, L), dtype=torch.float64, device=dev); cP = torch.zeros_like(cN)
    cE = torch.zeros((S, 5, L + 1), dtype=torch.float64, device=dev)
    first = torch.empty((S, t0 + 1), dtype=torch.float64, device=dev)
    it_a = torch.zeros(S, dtype=torch.int64, device=dev); it_b = torch.zeros_like(it_a)
    dv.solve_pl_snap_device(tX, length, Time * t0 / T, L, t0, tini, first, gpu.checkpoint_steps(t0), cN, cP, cE,
                            iters_total=it_a, flags=fl | gpu.FLAG_SNAP_RAW)
    out = torch.full((S, T + 1), float("nan"), dtype=torch.float64, device=dev)
    out[:, :t0 + 1] = first
    dv.solve_pl_resume_device(tX, length, Time, L, T, t0, cN, cP, cE, out, iters_total=it_b, flags=fl)
    torch.cuda.synchronize()
    it_c = torch.zeros_like(it_a)                               # the step at t0, counted by both segments
    dv.solve_pl_resume_device(tX, length, Time * t0 / T, L, t0, t0, cN, cP, cE, first, iters_total=it_c, flags=fl)
    torch.cuda.synchronize()
    assert torch.equal(out, full) and torch.equal(it_a + it_b - it_c, it_full)


def test_resume_arguments_are_validated(gpu):
    X, length, ini = _case(gpu, 3, seed=1)
    z = np.zeros((3, 5, 128)); ze = np.zeros((3, 5, 129))
    with pytest.raises(gpu.TrplError, match="t0"):
        gpu.solve_pl(X, length, 1.25, 128, 40, None, resume=(3, z, z, ze))
    with pytest.raises(gpu.TrplError, match="t0"):
        gpu.solve_pl(X, length, 1.25, 128, 40, None, resume=(41, z, z, ze))
    with pytest.raises(gpu.TrplError, match="FP32"):
        gpu.solve_pl(X, length, 1.25, 128, 40, None, resume=(8, z, z, ze), fp32=True)
    with pytest.raises(ValueError, match="resume levels"):
        gpu.solve_pl(X, length, 1.25, 128, 40, None, resume=(8, z[:, :4], z, ze))
    with pytest.raises(ValueError):
        gpu.checkpoint_steps(3)
    lib, A = gpu._abi.lib(), gpu._abi
    pl = np.zeros((3, 41))
    rc = lib.trpl_solve_pl_resume(A.ptr(X), 3, float(length), 1.25, 128, 40, 1, 7, 100, 8, A.ptr(z), None, A.ptr(ze),
                                  A.ptr(pl), 8, 41, None, None, None, 0, None, None, None, 0, 0, None)
    assert rc == A.ERR_ARG
    # t0 == T: the PL column of step T is (re)written from the checkpointed state and the loop's last step
    # (computed and dropped, like the reference's t = T) is taken once
    ck = {}
    full = gpu.solve_pl(X, length, 1.25, 128, 40, ini, snap_steps=gpu.checkpoint_steps(40), snapshots=ck, snap_raw=True,
                        kernel="single")
    out = np.full((3, 41), -1.0)
    _, st, it, _ = gpu.solve_pl(X, length, 1.25, 128, 40, None, out=out, resume=(40, ck["plN"], ck["plP"], ck["plE"]),
                                kernel="single")
    assert (out[:, :40] == -1.0).all() and np.array_equal(out[:, 40], full[0][:, 40]) and (it > 0).all()


def test_device_entry_points_can_be_captured_in_a_hip_graph(gpu):
    """The _dev calls only enqueue kernels on the caller's stream (no allocation, no synchronisation): a fused
    solve + likelihood step captured once in a HIP graph replays with new parameters in the same buffers and gives
    the eager call's bits."""
    import torch
    dv, w = gpu.device, gpu.workloads
    dev = torch.device("cuda:0")
    L, T, S = 128, 64, 4096 + 3
    ini, lens = w.power_scan(L)
    C = len(lens)
    Xa, Xb = w.samples(S, seed=11), w.samples(S, seed=12)
    ini_d = torch.from_numpy(ini).to(dev)
    mark = torch.from_numpy((w.MARKED_POINT * gpu.UNIT_CONVERSIONS)[None, :-1].copy()).to(dev)
    obs = torch.empty((C, T + 1), dtype=torch.float64, device=dev)
    for c in range(C):
        pl = torch.empty((1, T + 1), dtype=torch.float64, device=dev)
        dv.solve_pl_device(mark, lens[c], T * DT, L, T, ini_d[c].contiguous(), pl, flags=gpu.FLAG_STRICT)
        obs[c] = torch.log10(pl[0])
    X = torch.from_numpy(Xa).to(dev)
    P = torch.zeros(S, dtype=torch.float64, device=dev)
    sse = torch.empty((C, S), dtype=torch.float64, device=dev)
    it = torch.zeros((C, S), dtype=torch.int64, device=dev)

    def step():
        P.zero_()
        dv.loglik_device(X, ini_d, lens, T * DT, L, T, obs, [T + 1] * C, P, sse, iters_total=it, flags=gpu.FLAG_KERNEL_PAIR)

    eager = {}
    for name, Xh in (("a", Xa), ("b", Xb)):
        X.copy_(torch.from_numpy(Xh))
        step()
        torch.cuda.synchronize()
        eager[name] = (P.clone(), it.clone())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        X.copy_(torch.from_numpy(Xa))
        step()                                      # warm-up on the capture stream
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            step()
    for name, Xh in (("b", Xb), ("a", Xa), ("b", Xb)):
        X.copy_(torch.from_numpy(Xh))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(P, eager[name][0]) and torch.equal(it, eager[name][1]), name
    assert torch.isfinite(eager["a"][0]).all() and not torch.equal(eager["a"][0], eager["b"][0])


def test_continue_into_a_large_host_buffer_written_in_place(gpu):
    """A PL matrix above the direct-write threshold (8 MiB) is written by the kernel straight into the caller's mapped
    buffer: the continuation fills the columns from t0 on and leaves the earlier ones exactly as the caller had them."""
    X, length, ini = _case(gpu, 1100, seed=13)
    T, t0 = 1000, 420
    Time = T * DT
    assert 1100 * (T + 1) * 8 > 8 << 20
    full, st, it, _ = gpu.solve_pl(X, length, Time, 128, T, ini, kernel="pair")
    ck = {}
    first, *_ = gpu.solve_pl(X, length, t0 * DT, 128, t0, ini, kernel="pair", snap_steps=gpu.checkpoint_steps(t0),
                             snapshots=ck, snap_raw=True)
    out = np.full((1100, T + 1), -7.0)
    out[:, :t0 + 1] = first
    gpu.solve_pl(X, length, Time, 128, T, None, out=out, kernel="pair", resume=(t0, ck["plN"], ck["plP"], ck["plE"]))
    assert np.array_equal(out, full) and not st.any()
    marker = np.full((1100, T + 1), -7.0)
    gpu.solve_pl(X, length, Time, 128, T, None, out=marker, kernel="pair", resume=(t0, ck["plN"], ck["plP"], ck["plE"]))
    assert (marker[:, :t0] == -7.0).all() and np.array_equal(marker[:, t0:], full[:, t0:])


@pytest.mark.parametrize("kernel", ["pair", "single"])
def test_repeated_launches_give_the_same_bits(gpu, kernel):
    """No atomics, no launch-order dependence anywhere on the fused path: two launches of the same batch agree bit for
    bit in likelihoods, per-curve sums and iteration totals (a precondition of the sharding and checkpoint guarantees)."""
    w = gpu.workloads
    ini, lens = w.twothick(128)
    X = w.samples(4099, seed=8)
    T = 96
    obs = [np.full(T + 1, 17.0) - 0.01 * np.arange(T + 1)] * len(lens)
    runs = []
    for _ in range(2):
        info = {}
        P = gpu.loglik(X, ini, lens, T * DT, 128, T, obs, info=info, kernel=kernel)
        runs.append((P.copy(), info["sse"].copy(), info["iters_total"].copy()))
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b)
    assert np.isfinite(runs[0][0]).all()


@pytest.mark.parametrize("flag", ["pair", "single", "strict", "mixed"])
def test_device_resident_continue_of_a_single_system(gpu, flag):
    """trpl_solve_pl_resume_dev takes no excitation (the state comes from the checkpoint) and must not read one: with
    S = 1 the only other array of the call, matpar, holds 12 doubles -- a kernel that still loaded L excitation values
    through an alias of it read ~1 KB past the caller's tensor (round-2 advisor finding)."""
    if flag == "mixed":
        needs_experimental(gpu, dict(mixed=True))
    import torch
    dv = gpu.device
    X, length, ini = _case(gpu, 1, seed=31)
    L, T, Time, t0 = 128, 64, 64 * DT, 20
    dev = torch.device("cuda:0")
    fl = {"pair": gpu.FLAG_KERNEL_PAIR, "single": gpu.FLAG_KERNEL_SINGLE, "strict": gpu.FLAG_STRICT, "mixed": gpu.FLAG_MIXED}[flag]
    # the parameters sit at the very end of their own allocation
    tX = torch.from_numpy(X).to(dev).clone()
    tini = torch.from_numpy(np.ascontiguousarray(ini)).to(dev)
    full = torch.empty((1, T + 1), dtype=torch.float64, device=dev)
    dv.solve_pl_snap_device(tX, length, Time, L, T, tini, full, [], flags=fl)
    cN = torch.zeros((1, 5, L), dtype=torch.float64, device=dev); cP = torch.zeros_like(cN)
    cE = torch.zeros((1, 5, L + 1), dtype=torch.float64, device=dev)
    first = torch.empty((1, t0 + 1), dtype=torch.float64, device=dev)
    dv.solve_pl_snap_device(tX, length, Time * t0 / T, L, t0, tini, first, gpu.checkpoint_steps(t0), cN, cP, cE,
                            flags=fl | gpu.FLAG_SNAP_RAW)
    out = torch.full((1, T + 1), float("nan"), dtype=torch.float64, device=dev)
    out[:, :t0 + 1] = first
    st = torch.full((1,), -1, dtype=torch.int32, device=dev)
    dv.solve_pl_resume_device(tX, length, Time, L, T, t0, cN, cP, cE, out, status=st, flags=fl)
    torch.cuda.synchronize()
    assert torch.equal(out, full) and int(st[0]) == 0


@pytest.mark.parametrize("mode", MODES + [dict(strict=True, bundle=3), dict(bundle=2)],
                         ids=lambda m: "-".join("%s=%s" % kv for kv in m.items()))
def test_a_system_flagged_before_the_checkpoint_keeps_its_status_through_the_resume(gpu, oracle, mode):
    """A small iteration cap flags some systems in the first steps.  The checkpoint of a flagged system carries its
    status word (NaN payload); the continuation reports THAT status (not 1 + t0), takes no step for it (0 iterations
    instead of max_iter on NaNs at every step) and its PL / later snapshots are NaN exactly where the uninterrupted
    run's are -- status, PL and snapshots of every system bit for bit, iteration totals once the step at t0 is
    counted once."""
    if mode.get("mixed"):
        needs_experimental(gpu, dict(mixed=True))
    X, length, ini = _case(gpu, 12, seed=5)
    L, T, t0 = 128, 96, 40
    Time = T * DT
    # a cap between the systems' (bundles') largest per-step iteration counts: about half of them are flagged
    per_step = oracle.pvsim(X, length, Time, L, T, ini, mspb=mode.get("bundle", 1), want_step_iters=True, nthreads=4)["step_iters"]
    worst = np.unique(per_step.max(axis=1))
    assert len(worst) >= 2 and per_step.argmax(axis=1).max() <= t0
    cap = int(worst[len(worst) // 2])          # a step that needs >= cap iterations is flagged (pvSimPCR.py:269)
    kw = dict(MAX=cap, **mode)
    late = (t0 + 3, T)
    full_snaps = {}
    pl, st, it, _ = gpu.solve_pl(X, length, Time, L, T, ini, snap_steps=list(late), snapshots=full_snaps, **kw)
    assert (st > 0).any() and (st == 0).any() and (st[st > 0] <= t0).all()
    ck = {}
    pl_a, st_a, it_a, _ = gpu.solve_pl(X, length, Time * t0 / T, L, t0, ini, snap_steps=gpu.checkpoint_steps(t0),
                                       snapshots=ck, snap_raw=True, **kw)
    assert np.array_equal(st_a, st)
    out = np.full((len(X), T + 1), np.nan)
    out[:, :t0 + 1] = pl_a
    got = {}
    pl_b, st_b, it_b, _ = gpu.solve_pl(X, length, Time, L, T, None, out=out, resume=(t0, ck["plN"], ck["plP"], ck["plE"]),
                                       snap_steps=list(late), snapshots=got, **kw)
    assert np.array_equal(st_b, st)                            # the ORIGINAL failing step, through the checkpoint
    assert (it_b[st > 0] == 0).all() and (it_b[st == 0] > 0).all()
    assert np.array_equal(np.isnan(pl_b), np.isnan(pl)) and np.array_equal(pl_b[~np.isnan(pl)], pl[~np.isnan(pl)])
    for k in ("plN", "plP", "plE"):
        assert np.array_equal(np.isnan(got[k]), np.isnan(full_snaps[k])), k
        ok = ~np.isnan(got[k])
        assert np.array_equal(got[k][ok], full_snaps[k][ok]), k
    tail = np.full((len(X), t0 + 1), np.nan)
    it_c = gpu.solve_pl(X, length, Time * t0 / T, L, t0, None, out=tail, resume=(t0, ck["plN"], ck["plP"], ck["plE"]), **kw)[2]
    assert np.array_equal(it_a + it_b - it_c, it)
