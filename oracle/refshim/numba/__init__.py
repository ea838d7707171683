"""Build-owned stand-in for the `numba` import surface the reference uses.

TEST INFRASTRUCTURE ONLY.  It exists so that `oracle/gen_golden.py` can run the
reference's own numba-CUDA kernels (pvSimPCR.py, probs.py) *sequentially on the CPU*
in the development container, where numba is not installed, and capture golden
vectors.  It is never imported by the product path, the tests or the benchmark.

Why sequential execution is exact: every device loop in the
reference is written `for i in range(threadIdx.x, n, TPB)`, every cross-thread
read (PCR, the norm tree) goes through a snapshot taken before a barrier, and
blocks touch disjoint samples.  With one thread per block (TPB = 1) the loops
visit every element in index order, so the arithmetic is the reference's own.

Surface provided: njit / jit (identity), float32 / float64 (numpy dtypes),
cuda.jit, cuda.shared.array, cuda.syncthreads, cuda.synchronize, cuda.to_device,
cuda.threadIdx / blockIdx / blockDim, cuda.grid, cuda.gridsize, cuda.detect,
cuda.select_device, cuda.get_current_device.
"""
import numpy as np

from . import cuda  # noqa: F401  (from numba import cuda)

float32 = np.float32
float64 = np.float64


def _identity_decorator(*dargs, **dkwargs):
    # @njit, @njit(cache=True), @jit(nopython=True) ...
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def wrap(fn):
        return fn
    return wrap


njit = _identity_decorator
jit = _identity_decorator
