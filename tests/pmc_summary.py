"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel (sum over dispatches)."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        k = k.replace("void ", "").replace("rtx::(anonymous namespace)::", "").split("(rtx::")[0]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k].add((f, row["Dispatch_Id"]))
for k in sorted(tot):
    if "at::" in k or "rocclr" in k:
        continue
    c = tot[k]
    print(f"== {k}")
    for name in sorted(c):
        print(f"   {name:34s} {c[name]:.6g}")
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        print(f"   -> active-lane fraction (THREAD_CYCLES_VALU / (64*ACTIVE_INST_VALU)) = "
              f"{c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):.3f}")
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
        print(f"   -> VALU issue / SQ busy = {c['SQ_ACTIVE_INST_VALU'] / c['SQ_BUSY_CYCLES']:.3f}")
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]):
        print(f"   -> L2 hit rate = {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.4f}")

# HBM-side traffic per launch for bench.py's roofline.traffic: FETCH_SIZE is in KB and, on gfx950,
# tallies the L2's 128-byte fabric reads at 64 B (MI355X_MICROARCH.md, HBM section), hence the x2;
# the TCC_EA0_RDREQ_{32,64,128}B breakdown of the same command confirms every request is 128 B.
import json
traffic = {}
for k in tot:
    if "FETCH_SIZE" in tot[k] and ("trace_" in k):
        n = len([1 for f, d in calls[k] if "tcc2" in f])
        if n:
            traffic[k] = {"fetch_bytes_per_launch": tot[k]["FETCH_SIZE"] * 1024 * 2 / n, "launches_profiled": n,
                          "rdreq_128B": tot[k].get("TCC_EA0_RDREQ_128B_sum"), "rdreq_64B": tot[k].get("TCC_EA0_RDREQ_64B_sum"),
                          "rdreq_32B": tot[k].get("TCC_EA0_RDREQ_32B_sum")}
if len(sys.argv) > 2:
    json.dump(traffic, open(sys.argv[2], "w"), indent=1, sort_keys=True)
