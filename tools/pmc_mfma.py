"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (tools/prof_mfma.sh) -> matrix-core utilisation per kernel
family: busy cycles summed over the SIMDs / (GPU-active cycles of the dispatch x 1024 SIMDs).  rocprofv3 reports
GRBM_GUI_ACTIVE summed over the 8 XCDs (ms_fused_kernel<0>: 9.46 M "cycles" for a 520 us launch = 8 x 2.27 GHz), so the
dispatch's cycle count is that value / 8.  v_mfma_f32_32x32x2_f32
holds a SIMD's matrix pipe for 64 cycles (MI355X_MICROARCH.md, per-instruction constants), so utilisation 1.0 = 157 TFLOP/s
at 2.4 GHz; the shader clock under fp32 MFMA load is lower, which is why utilisation reads higher than TFLOP/s / 157.3."""
import csv, glob, json, os, re, sys

d = sys.argv[1]
f = glob.glob(os.path.join(d, "pmc", "**", "*counter_collection.csv"), recursive=True)
if not f:
    raise SystemExit("no counter_collection.csv under %s" % d)
per = {}
for r in csv.DictReader(open(f[0])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    m = re.match(r"void gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)", name)
    ms = re.match(r"void gemm_stream_kernel<(\d+), (\d+), (true|false)", name)
    if m:
        fam = "gemm_%s_bn%s" % (("nt", "nn", "tn")[int(m.group(5))], m.group(2))
    elif ms:
        fam = "gemm_stream_%s" % ("nt" if ms.group(3) == "true" else "nn")
    elif "gemm_stream_tn_kernel" in name:
        fam = "gemm_stream_tn"
    elif re.match(r"void gemm_pers_kernel<(\d+)", name):   # persistent form of the 128 x 128 kernel: same span family
        fam = "gemm_%s_bn128" % ("nt", "nn")[int(re.match(r"void gemm_pers_kernel<(\d+)", name).group(1))]
    else:
        fam = name.split("(")[0].replace("void ", "")[:40]
    key = (fam, r["Dispatch_Id"])
    per.setdefault(key, {})[r["Counter_Name"]] = per.setdefault(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
fam = {}
for (name, _), c in per.items():
    a = fam.setdefault(name, [0.0, 0.0, 0])
    a[0] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[1] += c.get("GRBM_GUI_ACTIVE", 0.0); a[2] += 1
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace over `python3 bench.py --steps 2 --warmup 1 "
                 "--no-cpu-baseline` (tools/prof_mfma.sh); utilisation = MFMA busy cycles / (GPU-active cycles x 1024 SIMDs), GPU-active = GRBM_GUI_ACTIVE / 8 XCDs",
       "families": {}}
for k, (busy, act, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    if busy <= 0 or act <= 0:
        continue
    out["families"][k] = {"dispatches": n, "mfma_busy_cycles_per_launch": busy / n, "gpu_active_cycles_per_launch": act / n / 8.0,
                          "mfma_utilisation": busy / (act / 8.0 * 1024.0)}
json.dump(out, open(os.path.join(d, "mfma.json"), "w"), indent=1)
for k, v in list(out["families"].items())[:12]:
    print("%-28s %4d launches  util %.3f  (%.2f M busy cycles, %.3f M active cycles per launch)" %
          (k, v["dispatches"], v["mfma_utilisation"], v["mfma_busy_cycles_per_launch"] / 1e6, v["gpu_active_cycles_per_launch"] / 1e6))
