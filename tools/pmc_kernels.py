"""Per-kernel summary of rocprofv3 --pmc passes (tools/prof_sq_bench.sh + the extra passes of a round): one entry per (kernel name,
grid size) whose name contains one of the patterns -- mean counter value per dispatch over all passes, mean duration, and the
derived ratios used in DESIGN.md's cycle budgets.
usage: python tools/pmc_kernels.py <dir with p*/..._counter_collection.csv> <out.json> pattern[,pattern...]"""
import csv, glob, json, os, sys
from collections import defaultdict

src, dst, pats = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
vals = defaultdict(lambda: defaultdict(list))
durs = defaultdict(list)
meta = {}
for f in sorted(glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if not any(p in name for p in pats):
            continue
        key = "%s | grid %s" % (name.replace("(anonymous namespace)::", "").split("(")[0], r["Grid_Size"])
        vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        did = (f, r["Dispatch_Id"])
        if did not in seen:
            seen.add(did)
            durs[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        meta[key] = {"workgroup": int(r["Workgroup_Size"]), "lds_bytes": int(r["LDS_Block_Size"]), "vgpr": int(r["VGPR_Count"]),
                     "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "scratch": int(r["Scratch_Size"])}
out = {}
for key, d in vals.items():
    c = {k: sum(v) / len(v) for k, v in d.items()}
    e = dict(meta[key], dispatches=len(durs[key]), us_under_pmc=sum(durs[key]) / len(durs[key]), counters=c)
    gui = c.get("GRBM_GUI_ACTIVE")
    wc = c.get("SQ_WAVE_CYCLES")
    der = {}
    if gui:
        # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md, "DVFS give-back"): the dispatch's cycles are
        # gui / 8; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs (tools/pmc_mfma.py)
        der["gpu_cycles"] = gui / 8.0
        der["clock_ghz"] = gui / 8.0 / (e["us_under_pmc"] * 1e3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            der["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0)
        if "SQ_BUSY_CYCLES" in c:
            der["sq_busy_per_gui"] = c["SQ_BUSY_CYCLES"] / gui
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                  "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM"):
            if k in c:
                der[k.lower() + "_per_wave_cycle"] = c[k] / wc
    if c.get("SQ_WAVES"):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VALU_MFMA_MOPS_F32"):
            if k in c:
                der[k.lower() + "_per_wave"] = c[k] / c["SQ_WAVES"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        der["tcc_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    e["derived"] = der
    out[key] = e
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for key in sorted(out, key=lambda k: -out[k]["us_under_pmc"]):
    e = out[key]
    print("%s\n   %.1f us under PMC, %d dispatches, %d threads, LDS %d B, VGPR %d + AGPR %d" % (key, e["us_under_pmc"], e["dispatches"], e["workgroup"],
          e["lds_bytes"], e["vgpr"], e["agpr"]))
    for k, v in sorted(e["derived"].items()):
        print("      %-44s %.4g" % (k, v))
