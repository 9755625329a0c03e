"""Per-kernel durations of the last frame of a rocprofv3 kernel trace, grouped by kernel name in launch order."""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
s = [i for i, r in enumerate(rows) if 'raygen' in r['Kernel_Name']][-1]
out = {}
for r in rows[s:]:
    m = re.search(r'(\w+_kernel)', r['Kernel_Name'])
    if m:
        out.setdefault(m.group(1), []).append(round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 3))
for k, v in out.items():
    print(k, v)
print('frame', round((int(rows[-1]['End_Timestamp']) - int(rows[s]['Start_Timestamp'])) / 1e6, 3))
