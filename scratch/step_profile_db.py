"""per-kernel time of the LAST training step from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d <dir> -o <name>`
writes <dir>/<name>_results.db): the same table scratch/step_profile.py makes from the csv output"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if 'sgd_kernel' in r[0]]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
wall = (step[-1][2] - step[0][1]) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in step:
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\(.*$', '', n)[:64]
    agg[n][0] += 1
    agg[n][1] += (e - s) / 1e6
tot = sum(v[1] for v in agg.values())
print("last step: %d kernels, wall %.2f ms, kernel sum %.2f ms" % (len(step), wall, tot))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print("%-66s %5d %8.3f ms %5.1f%%" % (k, v[0], v[1], 100 * v[1] / tot))
