"""VGPR / scratch / occupancy of conv kernels from a -Rpass-analysis=kernel-resource-usage log: python kres.py <log> [math]"""
import re, sys
t = open(sys.argv[1]).read()
math = sys.argv[2] if len(sys.argv) > 2 else None
for b in re.split(r'remark: [^\n]*Function Name: ', t)[1:]:
    name = b.split('\n')[0].split(' ')[0]
    def g(k):
        m = re.search(k + r': (\d+)', b); return int(m.group(1)) if m else -1
    v, sc, occ = g('VGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')
    m = re.search(r'conv_nt_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELi(\d)ELb(\d)ELi(\d)', name)
    if m and (math is None or m.group(6) == math):
        print("nt", m.groups(), 'V', v, 'scratch', sc, 'occ', occ)
    m = re.search(r'conv_wgrad_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)', name)
    if m and (math is None or m.group(5) == math):
        print("wgrad", m.groups(), 'V', v, 'scratch', sc, 'occ', occ)
