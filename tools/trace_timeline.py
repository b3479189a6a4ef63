"""every kernel of ONE replayed training step in start order (from a rocprofv3 --kernel-trace CSV directory): start (ms from the step's first kernel), duration,
queue, grid, name; then the busy time per queue and the time during which >= 2 kernels overlap.  usage: python tools/trace_timeline.py <trace dir> [step index from the end = 3]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void pack_kernel") or r["Kernel_Name"].startswith("pack_kernel")]
idx = [i for j, i in enumerate(idx) if j == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[idx[j - 1]]["Start_Timestamp"]) > 5_000_000]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seg = rows[idx[-k - 1]:idx[-k]]
t0 = int(seg[0]["Start_Timestamp"])
qs = {}
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    return n.replace("unsigned short", "bf16")[:64]
ev = []
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = qs.setdefault(r["Queue_Id"], len(qs))
    g = "(%d,%d,%d)" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])), int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
    print(f"{s / 1e6:8.3f} ms  {(e - s) / 1e3:8.1f} us  q{q}  {g:14s} {short(r['Kernel_Name'])}")
    ev.append((s, e, q))
end = max(e for _, e, _ in ev)
print(f"step: {len(seg)} kernels, {end / 1e6:.3f} ms first start -> last end")
for q in sorted(set(q for *_, q in ev)):
    print(f"  queue {q}: {sum(e - s for s, e, qq in ev if qq == q) / 1e6:.3f} ms busy, {sum(1 for *_, qq in ev if qq == q)} kernels")
pts = sorted([(s, 1) for s, _, _ in ev] + [(e, -1) for _, e, _ in ev])
cur, last, hist = 0, 0, {}
for t, d in pts:
    hist[cur] = hist.get(cur, 0) + t - last
    cur += d; last = t
print("  time with n kernels in flight:", {n: round(v / 1e6, 3) for n, v in sorted(hist.items())})
