import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'raygen' in r['Kernel_Name']]
s=idx[-1]
t0=int(rows[s]['Start_Timestamp'])
for r in rows[s:]:
    n=r['Kernel_Name']
    import re
    m=re.search(r'(\w+_kernel)',n)
    n=m.group(1) if m else n[:30]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e6:8.3f} {(int(r['End_Timestamp'])-t0)/1e6:8.3f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:7.3f} q{r.get('Queue_Id','?')} {n}")
