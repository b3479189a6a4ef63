"""Condense rocprofv3 outputs (gpurun_out/prof_*/...) into the small tracked files under profiles/."""
import collections
import csv
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)
stats = os.path.join(src, "stats", "bench_kernel_stats.csv")
if os.path.exists(stats):
    rows = list(csv.DictReader(open(stats)))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:60]:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in sorted(os.listdir(src)):
    p = os.path.join(src, sub, "bench_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"][:150]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"profiles/{tag}_pmc_summary.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Counter", "Dispatches", "MeanPerDispatch"])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
        if not any(t in k for t in ("gemm_", "attn_", "in_", "ln_", "loss", "pack", "adamw", "bias_grad", "up_cat", "win_", "conv")):
            continue
        for c, x in sorted(v.items()):
            w.writerow([k, c, len(x), f"{sum(x) / len(x):.1f}"])
print("wrote", [f for f in os.listdir("profiles") if f.startswith(tag)])
