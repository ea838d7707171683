"""MI355X (gfx950) drop-in for the batched TRPL drift-diffusion solve + log-likelihood hot path
of HagesLab/Bayesian-Inference-TRPL (reference pvSimPCR.py + probs.py behind bayeslib.simulate).

The directory name carries hyphens, so import it through the repo-root alias `trpl_amd`
(trpl_amd.py) or importlib.  Everything computes on the GPU through libtrpl_hip.so
(include/trpl.h); there is no CPU fallback.
"""
from . import _abi  # noqa: F401
_abi.ensure_built()            # a fresh checkout builds here, before this process can have touched the GPU
from ._abi import (FLAG_FP32, FLAG_FP32_LONG, FLAG_KERNEL_PAIR, FLAG_KERNEL_SINGLE, FLAG_MIXED, FLAG_NORMALIZE, FLAG_PL_F32, FLAG_SNAP_RAW,  # noqa: F401
                   FLAG_STRICT, TrplError)
from . import dataio, device, dist, posterior, workloads  # noqa: F401
from .dataio import export, get_data, get_initpoints  # noqa: F401
from .driver import almost_equal, bayes, bracket_times, interp_rows, is_grid_prefix, loglik, simulate  # noqa: F401
from .likelihood import fastlog, prob  # noqa: F401
from .model import checkpoint_steps, pvSim, solve_pl  # noqa: F401
from .sampler import (DEFAULT_DO_LOG, DEFAULT_MAXX, DEFAULT_MINX, PARAM_NAMES, UNIT_CONVERSIONS,  # noqa: F401
                      default_box, make_grid, random_grid)

__all__ = ["pvSim", "solve_pl", "checkpoint_steps", "fastlog", "prob", "simulate", "bayes", "loglik", "random_grid", "make_grid",
           "TrplError"]
