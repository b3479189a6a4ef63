"""wall / sum / union of kernel intervals per replayed step (steps delimited by the first pack_kernel launch of each step)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
# two pack launches per step when the pack is split: keep the first of each pair
starts = [i for j, i in enumerate(idx) if j == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[idx[j - 1]]["Start_Timestamp"]) > 5_000_000]
for a, b in zip(starts[-6:-1], starts[-5:]):
    seg = rows[a:b]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    u, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce: u += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    u += ce - cs
    print(f"step: n={len(seg)} wall={(t1 - t0) / 1e6:.2f} ms  sum={busy / 1e6:.2f}  union={u / 1e6:.2f}  idle={(t1 - t0 - u) / 1e6:.2f}")
