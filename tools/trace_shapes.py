"""Per-(kernel, grid) durations of ONE replayed training step from a rocprofv3 kernel trace CSV (steps are delimited by
pack_kernel launches); prints the table sorted by total time."""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
# the weight pack is two launches per step (encoder part, decoder part on the side stream): a step starts at the first of each pair
idx = [i for j, i in enumerate(idx) if j == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[idx[j - 1]]["Start_Timestamp"]) > 5_000_000]
a, b = idx[-8], idx[-7]          # a replayed step well inside the timed region
seg = rows[a:b]
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
    name = re.sub(r"unsigned short", "bf16", name).replace("void ", "")
    key = (name[:70], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg[key][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[key][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"step of {len(seg)} kernels, {tot / 1e6:.2f} ms")
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print(f"{t / 1e3:9.1f} us {n:4d}x {t / n / 1e3:8.1f} us/call  grid=({k[1]},{k[2]},{k[3]})  {k[0]}")
