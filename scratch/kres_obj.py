"""VGPR / scratch of the conv kernels straight from a built object: python scratch/kres_obj.py [obj] [filter]
(extracts the gfx950 code object with llvm-objdump --offloading and reads its kernel metadata)"""
import os, re, subprocess, sys, tempfile
B = "/opt/rocm/lib/llvm/bin/"
obj = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "dspnet_amd/csrc/_obj/conv.o")
flt = sys.argv[2] if len(sys.argv) > 2 else "conv_nt_kernel"
d = tempfile.mkdtemp()
subprocess.run(["cp", obj, d + "/x.o"], check=True)
subprocess.run([B + "llvm-objdump", "--offloading", d + "/x.o"], check=True, capture_output=True, cwd=d)
co = [f for f in os.listdir(d) if "gfx950" in f][0]
t = subprocess.run([B + "llvm-readelf", "--notes", d + "/" + co], capture_output=True, text=True).stdout
for blk in t.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    priv = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
    vg = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
    lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt in dn:
        m = re.search(r"<(.*)>", dn)
        print("%-60s vgpr %3d scratch %4d" % (m.group(1) if m else dn, vg, priv))
