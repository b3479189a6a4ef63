"""gpurun_out/<tag>/pmc_* (one rocprofv3 --pmc pass per directory) -> gpurun_out/<tag>_pmc.json: per-dispatch means of every counter for the
kernels whose name starts with <prefix>, HBM bytes with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (2 * FETCH_SIZE + WRITE_SIZE,
KB), MFMA-busy and LDS-conflict fractions, and the sha256 of the source file the numbers belong to."""
import collections, csv, glob, hashlib, json, os, sys
src, tag, prefix, srcfile = sys.argv[1:5]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --pmc <one counter group per pass> (tools/pmc_kernel.sh), per-dispatch means", "kernels": {}}
if os.path.exists(os.path.join(root, srcfile)):
    out["source_file"] = srcfile
    out["source_sha256"] = hashlib.sha256(open(os.path.join(root, srcfile), "rb").read()).hexdigest()
for k, d in agg.items():
    if not k.startswith(prefix):
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = {"counters": m, "dispatches": {c: len(v) for c, v in d.items()}}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_bytes_per_launch"] = 2.0 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_frac_of_lds_cycles"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    if m.get("SQ_WAVE_CYCLES"):
        e["wave_cycles_waiting_frac"] = m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
        e["wave_cycles_issue_stall_frac"] = m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
    out["kernels"][k] = e
p = os.path.join(os.path.dirname(src.rstrip("/")), tag + "_pmc.json")
json.dump(out, open(p, "w"), indent=1)
print("wrote", p, {k: {c: round(v) for c, v in e["counters"].items()} for k, e in out["kernels"].items()})
