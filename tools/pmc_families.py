"""gpurun_out/<tag>/pmc_* (the rocprofv3 --pmc passes of tools/pmc_kernel.sh over ONE eager training step) -> one summary per kernel family, each with the sha256
of the source file its kernels live in: gpurun_out/<tag>_<family>_pmc.json (copy to profiles/).  Per-dispatch means; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE
(KB; the gfx950 correction of MI355X_MICROARCH.md), MFMA-busy fraction of the chip, wave-cycle split.
usage: python tools/pmc_families.py gpurun_out/<tag> <tag> [grids per launch = 8]"""
import collections, csv, glob, hashlib, json, os, re, sys
src, tag = sys.argv[1:3]
grids = int(sys.argv[3]) if len(sys.argv) > 3 else 8
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAM = [("swin_block", r"^sw::", "swin_block.hip"), ("mlp96", r"^mlp96_|^mlp_(fwd|bwd)_kernel", "mlp_fused.hip"), ("norm_streaming", r"^(tail_|in_|ln_)", "norm.hip"),
       ("conv48", r"^conv48_", "conv48.hip"), ("cconv", r"^(cconv_|upconv4_)", "cconv.hip"), ("gemm_tn_grouped", r"^gemm_tn_grouped", "tn_grouped.hip"),
       ("gemm", r"^gemm_(nt|tn)_", "gemm.hip"), ("attn", r"^attn_", "attn.hip")]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[name[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for fam, rx, fn in FAM:
    out = {"source": "rocprofv3 --pmc, one counter group per pass (tools/pmc_kernel.sh) over one eager training step at %d grids; per-dispatch means" % grids, "grids_per_launch": grids,
           "source_file": "nerf-mae_amd/csrc/" + fn, "source_sha256": hashlib.sha256(open(os.path.join(root, "nerf-mae_amd/csrc", fn), "rb").read()).hexdigest(), "kernels": {}}
    for k, d in sorted(agg.items()):
        if not re.search(rx, k) or (fam == "gemm" and k.startswith("gemm_tn_grouped")):
            continue
        m = {c: sum(v) / len(v) for c, v in d.items()}
        e = {"counters": m, "dispatches": max(len(v) for v in d.values())}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["hbm_bytes_per_launch"] = 2.0 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
        if m.get("SQ_WAVE_CYCLES"):
            e["wave_cycles_active_frac"] = m.get("SQ_ACTIVE_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
            e["wave_cycles_waiting_frac"] = m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
            e["wave_cycles_issue_stall_frac"] = m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
        if m.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac_of_lds_cycles"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
        out["kernels"][k] = e
    if out["kernels"]:
        p = os.path.join(os.path.dirname(src.rstrip("/")), f"{tag}_{fam}_pmc.json")
        json.dump(out, open(p, "w"), indent=1)
        print("wrote", p, len(out["kernels"]), "kernels")
