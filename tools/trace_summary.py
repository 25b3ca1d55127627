import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'aug_kernel' in n or 'aug_rows_kernel' in n]
s, e = idx[-3], idx[-2]          # the last complete update (one aug launch per update)
t0 = int(rows[s]['Start_Timestamp']); prev = t0
agg = collections.OrderedDict()
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    n = n.split('(')[0][:44]
    print(f"{(st-t0)/1e3:8.1f} +{(en-st)/1e3:7.1f} gap {(st-prev)/1e3:5.1f} grid {r['Grid_Size_X']:>7}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']:<3} {n}")
    prev = en
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (en - st) / 1e3
print("---- per kernel (one update) ----")
tot = sum(v[1] for v in agg.values())
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:8.1f} us  x{c:<3} {n}")
print(f"sum of kernel time {tot:.1f} us; wall of the update {(int(rows[e]['Start_Timestamp'])-t0)/1e3:.1f} us")
