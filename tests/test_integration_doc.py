"""The ctypes stub printed in INTEGRATION.md section 2 is executed verbatim: documentation that drifts
from the ABI fails here."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. The ctypes stub itself"):]
    code = re.search(r"```python\n(.*?)```", sec, flags=re.S).group(1)
    return code.replace('C.CDLL("bayesian-inference-trpl_amd/libtrpl_hip.so")',
                        'C.CDLL(%r)' % os.path.join(ROOT, "bayesian-inference-trpl_amd", "libtrpl_hip.so"))


def test_documented_stub_binds_every_symbol_it_names(trpl):
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md#2", "exec"), ns)       # loads the library, sets argtypes
    assert callable(ns["pvSim"]) and callable(ns["fastlog"]) and callable(ns["prob"])


@pytest.mark.gpu
def test_documented_stub_computes_what_the_package_computes(trpl, gpu):
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md#2", "exec"), ns)
    S, T, L = 6, 40, 128
    X = trpl.workloads.samples(S, seed=4)
    ini, lengths = trpl.workloads.power_scan(L)
    simPar = [float(lengths[0]), T * 0.025, L, T, 1, (0,), 7, 10000]
    for dtype in (np.float32, np.float64):
        a = np.empty((S, T + 1), dtype=dtype)
        b = np.empty((S, T + 1), dtype=dtype)
        ns["pvSim"](a, None, None, None, X[:, :12], simPar, ini[0], None, None, 1, init_mode="points")
        trpl.pvSim(b, None, None, None, X[:, :12], simPar, ini[0], None, None, 1, init_mode="points")
        assert np.array_equal(a, b)
        ns["fastlog"](a, 2.2e-308, None, None)
        trpl.fastlog(b, 2.2e-308)
        assert np.array_equal(a, b)
        Pa, Pb = np.zeros(S), np.zeros(S)
        vals = np.linspace(20, 19, T + 1)
        ns["prob"](Pa, a, vals, None, X[:, 12].copy(), None, None)
        trpl.prob(Pb, b, vals, None, X[:, 12].copy())
        assert np.array_equal(Pa, Pb) and np.isfinite(Pa).all()
