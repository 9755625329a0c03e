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
