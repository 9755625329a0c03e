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

# Fabric traffic per launch and VALU issue share of the path kernel for bench.py's roofline object:
# FETCH_SIZE is in KB and, on gfx950, tallies the L2's 128-byte fabric reads at 64 B (MI355X_MICROARCH.md,
# HBM section), hence the x2.  Usage: pmc_summary.py <dir> <profiles/pmc.json> <workload key> <source note>
import importlib
import json
if len(sys.argv) > 4:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    rt = importlib.import_module("rust-raytracer_amd")
    out_path, wkey, note = sys.argv[2], sys.argv[3], sys.argv[4]
    kname = [k for k in tot if k.startswith("path_kernel<false, false, true") or k.startswith("path_ovh_kernel<false, true")]
    if kname:
        k = kname[0]
        c = tot[k]
        n_fetch = len([1 for f, d_ in calls[k] if os.path.basename(f).startswith("fetch")]) or 1
        n_req = len([1 for f, d_ in calls[k] if os.path.basename(f).startswith("tcc2")]) or 1
        n_sq = len([1 for f, d_ in calls[k] if os.path.basename(f).startswith("sq1")]) or 1
        entry = {"kernel": k, "kernel_hash": rt.kernel_source_hash(), "source": note,
                 "path_fetch_bytes_per_launch": c.get("FETCH_SIZE", 0.0) * 1024 * 2 / n_fetch, "launches_profiled": n_fetch}
        # cross-check of the x2: the fabric read requests by size (their own pass)
        if c.get("TCC_EA0_RDREQ_128B_sum") is not None:
            entry["fabric_read_bytes_by_request_size"] = (128 * c.get("TCC_EA0_RDREQ_128B_sum", 0.0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0.0)
                                                          + 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0.0)) / n_req
        if c.get("WRITE_SIZE") is not None:
            entry["path_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024 / (len([1 for f, d_ in calls[k] if os.path.basename(f).startswith("write")]) or 1)
        if c.get("TCC_HIT_sum") is not None and (c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0)):
            entry["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if c.get("GRBM_GUI_ACTIVE") and c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; one L1 tag access per CU and cycle is the L1's rate
            n_g = len([1 for f, d_ in calls[k] if os.path.basename(f).startswith("grbm")]) or 1
            n_t = len([1 for f, d_ in calls[k] if os.path.basename(f).startswith("tcp1")]) or 1
            entry["l1_tag_accesses_per_cu_cycle"] = (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / n_t / 256.0) / (c["GRBM_GUI_ACTIVE"] / n_g / 8.0)
        if c.get("SQ_INSTS_VALU") and c.get("SQ_WAVE_CYCLES"):
            # SQ_WAVE_CYCLES counts 4-cycle quads per resident wave; a wave64 VALU instruction occupies its SIMD for 4 cycles,
            # and 5 waves share a SIMD: issue share = VALU instructions / (wave quads / waves per SIMD)
            entry["valu"] = {"insts_per_launch": c["SQ_INSTS_VALU"] / n_sq,
                             "active_lane_fraction": c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]),
                             "issue_share_of_simd_time": c["SQ_INSTS_VALU"] / (c["SQ_WAVE_CYCLES"] / 5.0)}
            if c.get("GRBM_GUI_ACTIVE"):
                # 1,024 SIMDs, 4 cycles per wave64 VALU instruction, against the kernel's cycles (GRBM_GUI_ACTIVE / 8 XCDs)
                n_g = len([1 for f, d_ in calls[k] if os.path.basename(f).startswith("grbm")]) or 1
                entry["valu"]["issue_cycles_over_kernel_cycles"] = (c["SQ_INSTS_VALU"] / n_sq / 1024.0 * 4.0) / (c["GRBM_GUI_ACTIVE"] / n_g / 8.0)
        pj = json.load(open(out_path)) if os.path.exists(out_path) else {"workloads": {}}
        pj.setdefault("workloads", {})[wkey] = entry
        json.dump(pj, open(out_path, "w"), indent=1, sort_keys=True)
        print("wrote", out_path, wkey, entry)
