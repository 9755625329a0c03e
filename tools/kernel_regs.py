#!/usr/bin/env python3
"""VGPRs / scratch / occupancy / LDS per kernel of kernels.hip (compile-only; no GPU needed).
Usage: tools/kernel_regs.py [extra hipcc flags, e.g. -DRTX_TRACE_WAVES=4]"""
import os, re, subprocess, sys
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rust-raytracer_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
       "-Rpass-analysis=kernel-resource-usage", "-c", "kernels.hip", "-o", "/tmp/kregs.o"] + sys.argv[1:]
out = subprocess.run(cmd, cwd=src, capture_output=True, text=True).stderr
cur = None
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        cur = {"name": name.replace("rtx::(anonymous namespace)::", "").split("(")[0].replace("void ", "")}
    elif cur is not None:
        k, _, v = t.partition(":")
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            print("%-44s VGPRs %4s AGPRs %3s scratch %4s occ %s LDS %s" % (
                cur["name"], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("ScratchSize [bytes/lane]"),
                cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
            cur = None
