"""per-queue account of ONE replayed step: busy time per HSA queue, time with >= 2 kernels in flight, and every stretch >= 0.3 ms during which exactly
one queue is busy (which queue, from..to, its first kernels)"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
idx = [i for j, i in enumerate(idx) if j == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[idx[j - 1]]["Start_Timestamp"]) > 5_000_000]
seg = rows[idx[-8]:idx[-7]]
t0 = int(seg[0]["Start_Timestamp"])
busy = collections.Counter()
ev = []
for r in seg:
    s, e, q = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", "?")
    busy[q] += e - s
    ev.append((s, 1, q)); ev.append((e, -1, q))
ev.sort()
act = collections.Counter()
last, multi = 0, 0
for t, d, q in ev:
    n = sum(1 for v in act.values() if v > 0)
    if n >= 2: multi += t - last
    act[q] += d
    last = t
print("busy per queue (ms):", {q: round(v / 1e6, 2) for q, v in busy.items()}, " >=2 queues in flight:", round(multi / 1e6, 2), "ms  wall:", round(last / 1e6, 2))
for q in busy:
    ks = [r for r in seg if r.get("Queue_Id", "?") == q]
    print(f"queue {q}: {len(ks)} kernels, first at {(int(ks[0]['Start_Timestamp']) - t0) / 1e6:.2f} ms, last ends {(int(ks[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms")
# switch points of the longest chain: where consecutive kernels (by start) change queue
prev = None
for r in seg:
    q = r.get("Queue_Id", "?")
    if q != prev:
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:50]
        print(f"  {(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} ms -> queue {q}  {name}")
        prev = q
