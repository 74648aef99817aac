"""per-kernel time of the LAST training step of a rocprofv3 --kernel-trace --output-format csv run (file given)"""
import csv, sys, collections, re
def pretty(n):
    """demangled names as they are; the bf16-tensor instantiations come out mangled (_ZN12_GLOBAL__N_1...I Li4E Lb1E ...)"""
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I((?:L[ib]\d+E)+)', n)
    if m:
        args = re.findall(r'L([ib])(\d+)E', m.group(2))
        return "%s<%s> [bf16 tensors]" % (m.group(1), ", ".join(("true" if v == "1" else "false") if t == "b" else v for t, v in args))
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*$', '', n)[:66]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
# the last interval between two SGD launches that holds a whole step (bench.py's roofline_ops block launches sgd_kernel
# back to back after the timed region: those one-kernel intervals are not steps)
pairs = [(a + 1, b + 1) for a, b in zip(ends, ends[1:]) if b - a > 400]
lo, hi = pairs[-1]
step = rows[lo:hi]
wall = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    n = pretty(r['Kernel_Name'])
    agg[n][0] += 1
    agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
tot = sum(v[1] for v in agg.values())
print("last step: %d kernels, wall %.2f ms, kernel sum %.2f ms" % (len(step), wall, tot))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print("%-68s %5d %8.3f ms %5.1f%%" % (k, v[0], v[1], 100 * v[1] / tot))
