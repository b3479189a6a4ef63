"""per-launch durations, in launch order, of the kernels of ONE replayed step whose name contains <substr> and whose grid is <gx,gy> (from a rocprofv3
kernel-trace CSV directory): shows where in the step a kernel family runs slow"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
idx = [i for j, i in enumerate(idx) if j == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[idx[j - 1]]["Start_Timestamp"]) > 5_000_000]
seg = rows[idx[-8]:idx[-7]]
t0 = int(seg[0]["Start_Timestamp"])
sub, gx, gy = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
out = []
for r in seg:
    if sub in r["Kernel_Name"] and int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) == gx and int(r["Grid_Size_Y"]) == gy:
        out.append(((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print(" ".join(f"{t:.1f}ms:{d:.0f}us" for t, d in out))
