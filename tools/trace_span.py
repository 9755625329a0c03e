#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of bench.py: per frame, the span from the first trace
launch's start to the last one's end, and the per-kernel sums -- the figures bench.py's roofline
object reports from HIP events (avg_launch_ms and per_kernel.span_ms_per_step)."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
frames, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "raygen_kernel" in n:
        cur = {"t0": None, "t1": 0, "closest": 0, "shadow": 0, "count": "<true" in "", "names": set()}
        frames.append(cur)
    if cur is None or "trace_" not in n:
        continue
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    cur["t0"] = b if cur["t0"] is None else min(cur["t0"], b)
    cur["t1"] = max(cur["t1"], e)
    cur["closest" if "trace_closest" in n else "shadow"] += e - b
    cur["names"].add("<true" in n)
timed = [f for f in frames if f["t0"] is not None and True not in f["names"]]  # skip the counted pass
print(f"frames {len(frames)}, of which timed/warm-up (no visit counters) {len(timed)}")
for k, f in enumerate(timed):
    print(f"frame {k}: trace span {(f['t1'] - f['t0']) / 1e6:.3f} ms, trace_closest spans {f['closest'] / 1e6:.3f} ms, "
          f"trace_shadow spans {f['shadow'] / 1e6:.3f} ms")
if timed:
    print(f"mean trace span {sum(f['t1'] - f['t0'] for f in timed) / len(timed) / 1e6:.3f} ms")
