"""per-kernel sums of every counter of a rocprofv3 --pmc pass (csv output): python3 pmc_any.py <dir> [rows]"""
import csv, glob, sys, collections, re
f = (glob.glob(sys.argv[1] + "/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[0]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 25
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
names = []
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[:64]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[n].add(r['Dispatch_Id'])
    if r['Counter_Name'] not in names: names.append(r['Counter_Name'])
print("kernel,launches," + ",".join(names))
key = names[0]
for n, c in sorted(agg.items(), key=lambda kv: -kv[1].get(key, 0))[:rows]:
    print("%s,%d,%s" % (n.replace(',', ';'), len(cnt[n]), ",".join("%.4g" % c.get(k, 0) for k in names)))
