"""Time of the device sampler (trpl_sample_box_dev): S samples of the 13-column box, 10 random columns."""
import sys
sys.path.insert(0, ".")
import torch
import trpl_amd
from trpl_amd import device as tdev, sampler as sm

lo, hi, lg = sm.DEFAULT_MINX * sm.UNIT_CONVERSIONS, sm.DEFAULT_MAXX * sm.UNIT_CONVERSIONS, sm.DEFAULT_DO_LOG
for S in (65536, 524288):
    X = torch.empty((S, 13), dtype=torch.float64, device="cuda")
    tdev.sample_box_device(X, lo, hi, lg, seed=42)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    tdev.sample_box_device(X, lo, hi, lg, seed=42)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"S = {S}: {ms:.2f} ms  ({S * 10 / ms / 1e3:.1f} M draws/s; the H2D it replaces is {S * 13 * 8 / 1e6:.1f} MB)")
