"""HBM traffic of the conv family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per MI355X_MICROARCH.md:
FETCH_SIZE is reported in KiB and counts 64 B per 128-B request on gfx950 -> doubled; WRITE_SIZE in KiB, exact.
usage: pmc_traffic.py <fetch dir> <write dir> <steps+warmup> <out csv> <out json>"""
import csv, glob, json, sys, collections
def load(d, counter):
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter: continue
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        import re
        m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I((?:L[ib]\d+E)+)', n)
        if m:      # bf16-tensor instantiations come out mangled
            n = "%s<%s> [bf16 tensors]" % (m.group(1), ",".join(v for _, v in re.findall(r'L([ib])(\d+)E', m.group(2))))
        agg[n][0] += 1; agg[n][1] += float(r['Counter_Value'])
    return agg
fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
steps = int(sys.argv[3])
# forward/backward passes actually profiled: the range guard's calibration pass (two-piece math, before the first step) is one
# more than steps + warmup -- counted from a kernel that runs once per pass
passes = max([v[0] for k, v in fetch.items() if k.startswith('maxpool_fwd_kernel')] + [0]) or steps
steps = passes
rows = []
fam_bytes = fam_launch = 0
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch[k][1] + write[k][1])):
    n = max(fetch[k][0], write[k][0])
    b = (2 * fetch[k][1] + write[k][1]) * 1024
    rows.append((k, n, fetch[k][1], write[k][1], b / max(n, 1)))
    if k.startswith(('conv_nt', 'conv_stem', 'conv_wgrad_kernel', 'conv_wgw_kernel', 'slab_reduce', 'nt_split_reduce')):   # conv_nt / conv_ntw / conv_ntv
        fam_bytes += b; fam_launch += n if k.startswith('conv_') else 0
with open(sys.argv[4], 'w') as f:
    f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python3 bench.py --steps %d --warmup 1 --no-cpu-baseline --no-roofline\n" % (steps - 1))
    f.write("# units: KiB as reported; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024/launches (gfx950: FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md section HBM)\n")
    f.write("kernel,launches_%d_steps,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_bytes_per_launch\n" % steps)
    for r in rows[:60]:
        f.write("%s,%d,%.0f,%.0f,%.0f\n" % (r[0].replace(',', ';'), r[1], r[2], r[3], r[4]))
j = {"family": "conv_nt + conv_ntw + conv_ntv + conv_stem + conv_wgrad (+ their slab reduces)", "launches_per_step": fam_launch / steps,
     "hbm_bytes_per_step": fam_bytes / steps, "hbm_bytes_per_launch": fam_bytes / max(fam_launch, 1),
     "source": sys.argv[4] + " (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes, FETCH doubled)"}
json.dump(j, open(sys.argv[5], 'w'), indent=1)
print(json.dumps(j))
tot = sum((2 * fetch[k][1] + write[k][1]) * 1024 for k in set(fetch) | set(write)) / steps
print("all kernels: %.2f GB per step" % (tot / 1e9))
for r in rows[:12]: print("%-60s %5d  fetch %.1f GB write %.1f GB" % (r[0][:60], r[1], 2 * r[2] * 1024 / 1e9, r[3] * 1024 / 1e9))
