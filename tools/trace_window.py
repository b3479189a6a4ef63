"""kernels of ONE replayed step that overlap the window [t0, t1] ms (from the step's start), in start order: name, grid, start, duration"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
idx = [i for j, i in enumerate(idx) if j == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[idx[j - 1]]["Start_Timestamp"]) > 5_000_000]
seg = rows[idx[-8]:idx[-7]]
t0 = int(seg[0]["Start_Timestamp"])
a, b = float(sys.argv[2]), float(sys.argv[3])
for r in seg:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if e >= a and s <= b:
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")[:60]
        print(f"{s:8.3f} ms  {1e3 * (e - s):8.1f} us  grid=({int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))},{r['Grid_Size_Y']},{r['Grid_Size_Z']})  q={r.get('Queue_Id', '?')}  {name}")
