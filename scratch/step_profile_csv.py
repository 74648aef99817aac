"""per-kernel time of the LAST training step of a rocprofv3 --kernel-trace --output-format csv run (file given)"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
wall = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\(.*$', '', n)[:66]
    agg[n][0] += 1
    agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
tot = sum(v[1] for v in agg.values())
print("last step: %d kernels, wall %.2f ms, kernel sum %.2f ms" % (len(step), wall, tot))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print("%-68s %5d %8.3f ms %5.1f%%" % (k, v[0], v[1], 100 * v[1] / tot))
