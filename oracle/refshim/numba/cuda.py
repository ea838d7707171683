"""Sequential one-thread-per-block emulation of the numba.cuda names the reference uses.

TEST INFRASTRUCTURE ONLY (see package docstring).  Launch a kernel with TPB = (1,).
"""
import numpy as np


class _Dim:
    x = 0
    y = 0
    z = 0


threadIdx = _Dim()
blockIdx = _Dim()
blockDim = _Dim()
blockDim.x = 1
_grid = {"bpg": 1}


class _DeviceArray:
    """Host array posing as a device array (no copy: kernels write in place)."""

    def __init__(self, arr):
        self.arr = arr

    def copy_to_host(self):
        return self.arr


def to_device(a):
    return _DeviceArray(np.array(a))


def _unwrap(a):
    return a.arr if isinstance(a, _DeviceArray) else a


class _Kernel:
    def __init__(self, fn):
        self.fn = fn

    def __getitem__(self, cfg):
        bpg, tpb = cfg
        if isinstance(tpb, (tuple, list)):
            tpb = tpb[0]
        if int(tpb) != 1:
            raise ValueError("sequential emulation is only exact with one thread per block")
        bpg = int(bpg)

        def launch(*args):
            args = [_unwrap(a) for a in args]
            _grid["bpg"] = bpg
            threadIdx.x = 0
            for b in range(bpg):
                blockIdx.x = b
                self.fn(*args)
            blockIdx.x = 0
        return launch


def jit(*dargs, device=False, **kw):
    def wrap(fn):
        return fn if device else _Kernel(fn)
    if len(dargs) == 1 and callable(dargs[0]):
        return wrap(dargs[0])
    return wrap


class shared:
    @staticmethod
    def array(shape, dtype):
        return np.zeros(shape, dtype=dtype)


def syncthreads():
    pass


def synchronize():
    pass


def grid(ndim):
    return blockIdx.x          # valid because blockDim.x == 1


def gridsize(ndim):
    return _grid["bpg"]


def detect():
    return True


def select_device(i):
    if i != 0:
        raise IndexError(i)


class _Device:
    MULTIPROCESSOR_COUNT = 1


def get_current_device():
    return _Device()
