#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (written by tools/profile_round.sh) into profiles/<tag>_*:
the bench line, the rocprofv3 kernel stats of our kernels, and per-launch HBM traffic from the
FETCH_SIZE / WRITE_SIZE passes with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE
counts 64 B per 128-B request for wide coalesced reads: x2; both counters are in KiB)."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
OURS = ("trpl::",)

bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)

def newest(pattern):
    """gpurun merges every run of a tag into the same directory: keep the latest run's file only."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


rows = []
for f in newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in OURS):
            rows.append(r)
with open(os.path.join(dst, tag + "_rocprof_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

traffic = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in newest(os.path.join(src, "pmc_" + cname, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname or not any(k in r["Kernel_Name"] for k in OURS):
                continue
            k = r["Kernel_Name"].split("(")[0]
            d = traffic.setdefault(k, {"FETCH_SIZE": [], "WRITE_SIZE": []})
            d[cname].append(float(r["Counter_Value"]))
def median(v):
    v = sorted(v)
    n = len(v)
    return 0.0 if n == 0 else (v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2]))


out = {}
for k, d in traffic.items():
    # per-launch MEDIAN, not mean: the TCC counters are device-wide, and whatever else touches HBM while one dispatch runs is
    # counted with it (round 4, r4_v7: three stepper launches wrote 14 848 KiB each, the fourth "wrote" 180 510 KiB -- its
    # mean, 57.6 MB, went into DESIGN.md and the bench line as if it were the kernel's); min / max show such a launch
    fetch = median(d["FETCH_SIZE"])
    write = median(d["WRITE_SIZE"])
    out[k] = {"launches_fetch": len(d["FETCH_SIZE"]), "launches_write": len(d["WRITE_SIZE"]),
              "FETCH_SIZE_KiB_min_max": [min(d["FETCH_SIZE"] or [0.0]), max(d["FETCH_SIZE"] or [0.0])],
              "WRITE_SIZE_KiB_min_max": [min(d["WRITE_SIZE"] or [0.0]), max(d["WRITE_SIZE"] or [0.0])],
              "FETCH_SIZE_KiB_per_launch_raw": fetch, "WRITE_SIZE_KiB_per_launch": write,
              "hbm_read_bytes_per_launch_corrected": fetch * 1024 * 2,
              "hbm_write_bytes_per_launch": write * 1024,
              "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024}
# what the PMC passes ran (bench.py reads this back to label `traffic`)
# ... and WHICH library: bench.py's attach_traffic flags the quotation as stale when the loaded library's hash differs
out["_meta"] = {"tag": tag, "T": bench["config"].get("T"), "workload": bench["config"].get("workload"),
                "srchash": (bench.get("library") or {}).get("srchash"),
                "command": "bench.py --no-cpu-baseline --no-full-length (tools/profile_round.sh)"}
json.dump(out, open(os.path.join(dst, tag + "_hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out.items() if k != "_meta"}, indent=1))
print("value", bench["value"], "pcr frac", bench.get("roofline_hbm_pcr", {}).get("frac"))
