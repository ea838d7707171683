"""Which of the two HIP runtimes in this image (torch's bundled one, /opt/rocm's) a process ends up with depends
on what is loaded first; this probe reports whether torch still sees the GPU when libtrpl_hip.so came first.
    python tools/load_order_probe.py lib_first|torch_first"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
if order == "lib_first":
    import trpl_amd
    n = trpl_amd._abi.lib().trpl_device_count()
    import torch
else:
    import torch
    ok0 = torch.cuda.is_available()
    import trpl_amd
    n = trpl_amd._abi.lib().trpl_device_count()
maps = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l]
print(order, "cwd", os.getcwd(), "trpl devices", n, "torch.cuda.is_available", torch.cuda.is_available(),
      "device_count", torch.cuda.device_count(), sorted(set(maps)))
