"""gpurun_out/<tag>/pmc_* (one rocprofv3 --pmc pass per directory) + rocm_smi_*.txt -> profiles/<tag>_conv48{,_wgrad}_pmc.json and the
raw rocm-smi samples under profiles/.  The JSON records the sha256 of csrc/conv48.hip: bench.py only quotes `traffic` from a file whose
hash equals the source it runs (a stale PMC file is refused)."""
import collections, csv, glob, hashlib, json, os, shutil, sys
src, tag, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha256(open(os.path.join(root, "nerf-mae_amd", "csrc", "conv48.hip"), "rb").read()).hexdigest()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
for key, name in (("conv48_kernel", "conv48"), ("conv48_wgrad_kernel", "conv48_wgrad")):
    # (conv48_kernel has three instantiations since round 3: the plain one -- forward / input gradient -- is <0, false, false>)
    d = agg.get("conv48_kernel<0, false, false>") if key == "conv48_kernel" else None
    if d is None:
        d = next((v for k, v in agg.items() if k.startswith(key + "<") or k == key), None)
    if d is None:
        print("no counters for", key, list(agg)); continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    out = {"kernel": key, "batch_per_gpu": B, "resolution": 160, "conv48_hip_sha256": sha,
           "source": "rocprofv3 --pmc <one counter group per pass> -- python tools/bench_conv48.py %d (tools/pmc_conv48.sh), per-dispatch means" % B,
           "counters": m, "dispatches": {c: len(v) for c, v in d.items()}}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        out["hbm_bytes_per_launch"] = 2.0 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024
        out["note"] = "gfx950: FETCH_SIZE (KB) reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE (KB) as reported"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # SQ counters are summed over the 8 XCDs (32 CUs x 4 SIMDs each share...): busy fraction = MFMA busy cycles / (GUI_ACTIVE per XCD x 256 CUs x 4 SIMDs)
        out["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
    smi = os.path.join(src, "rocm_smi_%s.txt" % ("fwd" if name == "conv48" else "wgrad"))
    if os.path.exists(smi):
        dst = os.path.join(root, "profiles", "%s_rocm_smi_%s.txt" % (tag, name))
        shutil.copy(smi, dst)
        out["rocm_smi_raw"] = os.path.basename(dst)
    json.dump(out, open(os.path.join(root, "profiles", "%s_%s_pmc.json" % (tag, name)), "w"), indent=1)
    print("wrote", "%s_%s_pmc.json" % (tag, name), {k: round(v) for k, v in m.items()})
