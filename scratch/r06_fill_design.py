"""Fills the round-6 numbers of DESIGN.md (between the <!--R06:x--> markers) from profiles/r06_*: python scratch/r06_fill_design.py"""
import json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(R, "profiles", n)
d = json.loads(open(P("r06_default_bench_line.json")).read())
r, s = d["roofline"], d["sustained"]
sens = ""
if "sclk_mhz" in s:
    sens = " at %.0f MHz / %.0f W (median of %d sysfs samples)" % (s["sclk_mhz"]["median"], s.get("power_w", {}).get("median", float("nan")), s["sclk_mhz"]["samples"])
head = ("Headline of the closing run (`profiles/r06_default_bench_line.json`): **%.1f images/s** (%.2f ms per step over the K = %d timed steps; "
        "`sustained` over 200 further steps: %.1f images/s%s), conv family %.2f ms (forward + data gradient %.2f, weight gradient %.2f) = %.1f TFLOP/s "
        "executed = **%.3f of 833.3** by HIP events, `frac_end_to_end_3x` %.3f; boxes of this round ranged 913 - 985 images/s on the same library "
        "(±2.5 %%; every comparison in this file is same-box)."
        % (d["value"], d["ms_per_step"], d["steps"], s["value"], sens, r["conv_ms_per_step"], r["nt_ms_per_step"], r["wgrad_ms_per_step"],
           r["achieved"], r["frac"], r["frac_end_to_end_3x"]))
rows = []
names = {"f16x2": "fp32 / two fp16 pieces (default)", "x3": "fp32 / three bf16 pieces (`--math bf16x3`)", "fp32": "fp32 / fp32 MFMA (`--math fp32`)",
         "bf16": "bf16 tensors / bf16 MFMA (`--store bf16`)"}
peaks = {"f16x2": 833.3, "x3": 416.7, "fp32": 157.3, "bf16": 2500.0}
for m in ("f16x2", "x3", "fp32", "bf16"):
    b = json.loads(open(P("r06_%s_bench_line.json" % m)).read()); rr = b["roofline"]
    t = json.load(open(P("r06_%s_pmc_conv_family.json" % m)))
    extra = ""
    if m == "f16x2":
        extra = " under `rocprofv3 --kernel-trace`; **%.1f** un-profiled (`python bench.py`)" % d["value"]
    rows.append("| resnet-50 multitask 512×512, bs 32 | %s | %.1f%s | %.1f | %.1f ms = %.1f TFLOP/s: %.3f of %.1f | %.1f GB |"
                % (names[m], b["value"], extra, b["ms_per_step"], rr["conv_ms_per_step"], rr["achieved"], rr["frac"], peaks[m], t["hbm_bytes_per_step"] / 1e9))
oc = d.get("other_configs", {})
inf = json.loads(open(P("r06_infer_bench_line.json")).read())
rows.append("| inference bs 64 (configs[4]) | fp32 / f16x2 | %.0f images/s, **p50 %.2f ms** (p90 %.2f) | | | |" % (inf["images_per_s"], inf["value"], inf["p90_ms"]))
rows.append("| vgg16_reduced bs 16 / inceptionv3 1024×512 bs 8 bf16 (configs[1], configs[3]; side measurements of the default run) | | %.1f / %.1f | %.1f / %.1f | %.3f of 833.3 / %.3f of 2500 | |"
            % (oc["configs[1]"]["images_per_s"], oc["configs[3]"]["images_per_s"], oc["configs[1]"]["ms_per_step"], oc["configs[3]"]["ms_per_step"],
               oc["configs[1]"]["roofline"]["frac"], oc["configs[3]"]["roofline"]["frac"]))
cb = d["cpu_baseline"]
mb = cb.get("multibox_ops", {})
rows.append("| CPU baseline (kind \"port\", %d cores) | torch-CPU graph restatement / C multibox oracle | %.2f images/s at B = 1, %.2f at B = 4; MultiBoxTarget %.0f samples/s on one thread (%.0f over the batch), MultiBoxDetection %.0f (%.0f) | | | |"
            % (cb["cores"], cb["conv_path"]["B=1"]["value"], cb["value"], mb["MultiBoxTarget"]["1_thread"], mb["MultiBoxTarget"]["batch_parallel"],
               mb["MultiBoxDetection"]["1_thread"], mb["MultiBoxDetection"]["batch_parallel"]))
table = "| workload | tensors / math | images/s | ms/step | conv family (HIP events) | L2-miss traffic of the family per step |\n|---|---|---|---|---|---|\n" + "\n".join(rows)
# last step by kernel: family sums and the tail
ls = open(P("r06_f16x2_last_step_by_kernel.txt")).read().splitlines()
m0 = re.match(r"last step: (\d+) kernels, wall ([\d.]+) ms, kernel sum ([\d.]+) ms", ls[0])
fam = slab = 0.0
groups = {"conv_ntw": 0.0, "conv_ntv": 0.0, "conv_nt_": 0.0, "conv_stem": 0.0, "conv_wgrad": 0.0}
cnt = dict.fromkeys(groups, 0)
for l in ls[1:]:
    mm = re.match(r"(\S.*?)\s+(\d+)\s+([\d.]+) ms", l)
    if not mm: continue
    n, c, t = mm.group(1), int(mm.group(2)), float(mm.group(3))
    for k in groups:
        if (n.startswith(k.rstrip("_")) or (k == "conv_wgrad" and n.startswith("conv_wgw"))) and (k != "conv_nt_" or n.startswith("conv_nt_kernel")):
            groups[k] += t; cnt[k] += c; fam += t; break
    if n.startswith("slab_reduce_batch"): slab = t
ksum = float(m0.group(3))
tail = ksum - fam - slab
step = ("Last step of the profiled headline run by kernel (`r06_f16x2_last_step_by_kernel.txt`: %s launches, wall %s ms, kernel sum %.2f ms): "
        "`conv_ntw` %.2f ms in %d launches, `conv_ntv` %.2f in %d, `conv_nt_kernel` %.2f in %d, `conv_stem` %.2f, weight gradients %.2f in %d "
        "+ `slab_reduce_batch` %.2f: family %.2f ms; **tail = kernel sum − family − slab reduce = %.2f ms** (round 5 by the same formula: 11.7), "
        "part of which runs under other kernels on the side streams (wall < kernel sum)."
        % (m0.group(1), m0.group(2), ksum, groups["conv_ntw"], cnt["conv_ntw"], groups["conv_ntv"], cnt["conv_ntv"], groups["conv_nt_"], cnt["conv_nt_"],
           groups["conv_stem"], groups["conv_wgrad"], cnt["conv_wgrad"], slab, fam, tail))
p = os.path.join(R, "DESIGN.md")
t = open(p).read()
def put(tag, text):
    global t
    a, b = "<!--R06:%s-->" % tag, "<!--/R06:%s-->" % tag
    assert a in t and b in t, tag
    t = t[:t.index(a) + len(a)] + "\n" + text + "\n" + t[t.index(b):]
put("HEAD", head); put("TABLE", table + "\n\n" + step)
open(p, "w").write(t)
print(head); print(step)
