import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_common import DT, T_BENCH, nthreads  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/liboracle.so on first use."""
    import oracle as o
    o.load()
    return o


@pytest.fixture(scope="session")
def trpl():
    """The product package; on the GPU box a missing/failed HIP library is an error, not a skip."""
    import trpl_amd
    return trpl_amd


@pytest.fixture(scope="session")
def gpu(trpl):
    if trpl._abi.lib().trpl_device_count() < 1:
        pytest.fail("-m gpu tests need a visible HIP device")
    return trpl


# ---------------------------------------------------------------------------------------------------------------------
# Oracle solutions shared by several -m gpu files (each solved once per session; seconds on the box's host cores)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="session")
def long_window(gpu, oracle):
    """configs[0] at the bench's window, solved once by the oracle (16 threads: seconds)."""
    w = gpu.workloads
    T, L, S = 8000, 128, 64
    Time = T * DT
    ini, lens = w.power_scan(L)
    X = w.samples(S)
    mark = (w.MARKED_POINT * gpu.UNIT_CONVERSIONS)[None, :-1]
    ref = [oracle.pvsim(X[:, :12], lens[c], Time, L, T, ini[c], nthreads=16) for c in range(3)]
    obs = [np.log10(oracle.pvsim(mark, lens[c], Time, L, T, ini[c])["plI"][0]) for c in range(3)]
    P = np.zeros(S)
    sse = np.zeros((3, S))
    for c in range(3):
        lg = ref[c]["plI"].copy()
        oracle.fastlog(lg)                                   # probs.fastlog
        Pc = np.zeros(S)
        oracle.prob(Pc, lg, obs[c], np.ascontiguousarray(X[:, -1]))      # probs.prob: P -= sum (lg + mag - obs)^2
        sse[c] = -Pc
        P += Pc
    return dict(T=T, L=L, S=S, Time=Time, ini=ini, lens=lens, X=X, ref=ref, obs=obs, P=P, sse=sse)


@pytest.fixture(scope="session")
def decayed(gpu, oracle):
    """Samples with tau_n, tau_p of 0.3 .. 3 ns: gone by e^-17 .. e^-170 inside a 50 ns window."""
    w = gpu.workloads
    T, L, S = 2000, 128, 48
    Time = T * DT
    ini, lens = w.power_scan(L)
    X = w.samples(S, seed=7)
    rng = np.random.default_rng(3)
    X[:, 9] = 10 ** rng.uniform(np.log10(0.3), np.log10(3.0), S)
    X[:, 10] = X[:, 9] * 10 ** rng.uniform(-0.3, 0.3, S)
    ref = [oracle.pvsim(X[:, :12], lens[c], Time, L, T, ini[c], nthreads=16) for c in range(3)]
    return dict(T=T, L=L, S=S, Time=Time, ini=ini, lens=lens, X=X, ref=ref)


@pytest.fixture(scope="session")
def twothick_window(gpu, oracle):
    w = gpu.workloads
    L, S, T = 128, 32, T_BENCH
    Time = T * DT
    ini, lens = w.twothick(L)
    X = w.samples(S)
    mark = (w.MARKED_POINT * gpu.UNIT_CONVERSIONS)[None, :-1]
    ref = [oracle.pvsim(X[:, :12], lens[c], Time, L, T, ini[c], nthreads=nthreads()) for c in range(6)]
    obs = [np.log10(oracle.pvsim(mark, lens[c], Time, L, T, ini[c])["plI"][0]) for c in range(6)]
    sse = np.zeros((6, S))
    mag = np.ascontiguousarray(X[:, -1])
    for c in range(6):
        lg = ref[c]["plI"].copy()
        oracle.fastlog(lg)
        Pc = np.zeros(S)
        oracle.prob(Pc, lg, obs[c], mag)
        sse[c] = -Pc
    return dict(L=L, S=S, T=T, Time=Time, ini=ini, lens=lens, X=X, ref=ref, obs=obs, sse=sse)


@pytest.fixture(scope="session")
def l512_window(gpu, oracle):
    w = gpu.workloads
    L, S, T, length = 512, 16, T_BENCH, 2000.0
    Time = T * DT
    ini = np.stack([w.beer_lambert(A, length, L) for A in w.POWER_SCAN_A_CM3])
    X = w.samples(S, seed=61)
    ref7 = [oracle.pvsim(X[:, :12], length, Time, L, T, ini[c], tol=7, nthreads=nthreads()) for c in range(3)]
    ref6 = [oracle.pvsim(X[:, :12], length, Time, L, T, ini[c], tol=6, nthreads=nthreads()) for c in range(3)]
    return dict(L=L, S=S, T=T, Time=Time, length=length, ini=ini, X=X, ref7=ref7, ref6=ref6)
