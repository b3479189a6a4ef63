"""gpurun_out/<tag>/pmc_{FETCH_SIZE,WRITE_SIZE} -> profiles/<tag>_hbm_kernels_pmc.json: rocprofv3-reported HBM bytes per launch of the
decoder-1 elementwise kernels (the dispatches at the 160^3 size: those within 2x of the kernel's largest FETCH_SIZE), with the sha256 of
csrc/norm.hip they belong to (bench.py quotes them only for that source)."""
import collections, csv, glob, hashlib, json, os, sys
src, tag, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha256(open(os.path.join(root, "nerf-mae_amd", "csrc", "norm.hip"), "rb").read()).hexdigest()
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(k in name for k in ("tail_fwd", "tail_bwd", "in_apply", "in_bwd_apply", "in_reduce")):
            vals[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"batch_per_gpu": B, "resolution": 160, "norm_hip_sha256": sha,
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --batch-per-gpu %d --eager (tools/pmc_step.sh)" % B,
       "note": "gfx950: FETCH_SIZE (KB) reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE (KB) as reported", "kernels": {}}
for name, d in vals.items():
    if "FETCH_SIZE" not in d:
        continue
    mx = max(d["FETCH_SIZE"])
    fs = [v for v in d["FETCH_SIZE"] if v >= 0.5 * mx]
    ws = d.get("WRITE_SIZE", [0.0])
    wmx = max(ws) if ws else 0.0
    wl = [v for v in ws if v >= 0.5 * wmx] if wmx > 0 else [0.0]
    fetch, write = sum(fs) / len(fs), sum(wl) / len(wl)
    out["kernels"][name] = {"FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "hbm_bytes_per_launch": 2.0 * fetch * 1024 + write * 1024, "dispatches": len(fs)}
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", "%s_hbm_kernels_pmc.json" % tag), "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 3) for k, v in out["kernels"].items()}))
