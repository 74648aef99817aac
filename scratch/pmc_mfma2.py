"""MFMA-pipe busy share per kernel name from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES pass (csv)"""
import csv, glob, sys, collections, re
f = (glob.glob(sys.argv[1] + "/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[:70]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES': cnt[n] += 1
print("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-roofline")
print("# mfma_busy_share = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES * 4 SIMDs), summed over the kernel's launches")
print("kernel,launches,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,mfma_busy_share")
for n, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0))[:30]:
    m, b = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), c.get('SQ_BUSY_CU_CYCLES', 0.0)
    print("%s,%d,%.0f,%.0f,%.3f" % (n.replace(',', ';'), cnt[n], m, b, m / (4 * b) if b else 0))
