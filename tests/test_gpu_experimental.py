"""The experimental build flavour (`make EXPERIMENTAL=1` -> libtrpl_hip_exp.so: the default library + the two measured-and-
rejected steppers TRPL_FLAG_MIXED / TRPL_FLAG_HIST32).  Under the default library their tests SKIP after
checking that the flags are refused (gpu_common.needs_experimental); here the same tests run for real, in ONE child process
that loads the experimental library through TRPL_LIBRARY -- so a run of the suite on the default library still shows that
the experimental sources work.  __graft_entry__.build() builds both flavours."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SELECT = "mixed or hist32 or cfg4_fp64_state or bundle_flag_is_validated"


def test_experimental_steppers_pass_their_tests_on_the_experimental_library(gpu):
    if gpu._abi.has_experimental():
        pytest.skip("this run already uses the experimental library: the selected tests run in-process")
    exp = os.path.join(os.path.dirname(gpu._abi.LIB_PATH), "libtrpl_hip_exp.so")
    if not os.path.isfile(exp):
        pytest.skip("libtrpl_hip_exp.so is not built (make -C bayesian-inference-trpl_amd EXPERIMENTAL=1)")
    with open(exp + ".srchash") as fh:                                       # built from the sources that are here now
        assert fh.read().strip() == gpu._abi.source_hash(), "libtrpl_hip_exp.so is stale: rebuild it with make EXPERIMENTAL=1"
    env = dict(os.environ, TRPL_LIBRARY=exp, TRPL_AUTOBUILD="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu", "-q", "-p", "no:cacheprovider",
                        "-k", SELECT], env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0, (tail, r.stdout[-3000:], r.stderr[-1500:])
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 10 and "skipped" not in tail and "failed" not in tail, tail
