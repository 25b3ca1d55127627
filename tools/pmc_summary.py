import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + '/trace/*/*_kernel_trace.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.OrderedDict()
    for r in rows:
        k = (r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:50], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
        agg.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in agg.items():
        if 'at::' in k[0]: continue
        v = v[1:] if len(v) > 2 else v
        print(f"trace {k[0]:52s} grid {k[1]}x{k[2]}x{k[3]} n={len(v)} avg {sum(v)/len(v):8.1f} us")
for f in sorted(glob.glob(out + '/pmc*/*/*_counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.OrderedDict()
    for r in rows:
        k = (r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:50], r['Grid_Size'])
        agg.setdefault(k, collections.OrderedDict()).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for k, d in agg.items():
        if 'at::' in k[0]: continue
        print(f.split('/')[-3], k)
        for c, v in d.items():
            v = v[1:] if len(v) > 2 else v
            print(f"    {c:30s} n={len(v)} mean={sum(v)/len(v):.4g}")
