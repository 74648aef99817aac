"""per-kernel MFMA utilisation / LDS bank-conflict rate from rocprofv3 --pmc passes (derived metrics MfmaUtil, LdsBankConflict ...):
usage: pmc_util.py <counter dir> <counter name> <out csv>  -> duration-agnostic mean, min, max of the metric per kernel name"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]
name = sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != name: continue
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    agg[k].append(float(r['Counter_Value']))
rows = sorted(agg.items(), key=lambda kv: -len(kv[1]) * (sum(kv[1]) / len(kv[1])))
with open(sys.argv[3], 'w') as o:
    o.write("# rocprofv3 --kernel-trace --pmc %s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline (per launch values, all launches)\n" % name)
    o.write("kernel,launches,mean_%s,min,max\n" % name)
    for k, v in rows:
        o.write("%s,%d,%.2f,%.2f,%.2f\n" % (k.replace(',', ';'), len(v), sum(v) / len(v), min(v), max(v)))
for k, v in rows[:16]:
    print("%-62s %5d  mean %7.2f  min %7.2f  max %7.2f" % (k[:62], len(v), sum(v) / len(v), min(v), max(v)))
