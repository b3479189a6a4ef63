"""every launch of ONE replayed step whose kernel name contains <substr> (from a rocprofv3 kernel-trace CSV directory): start within the step, duration,
grid in workgroups, queue -- which launches of a family are the long ones, and what they run beside"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
idx = [i for j, i in enumerate(idx) if j == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[idx[j - 1]]["Start_Timestamp"]) > 5_000_000]
seg = rows[idx[-4]:idx[-3]]
t0 = int(seg[0]["Start_Timestamp"])
for sub in sys.argv[2:]:
    for r in seg:
        if sub in r["Kernel_Name"]:
            gx = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} ms  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  grid=({gx},{r['Grid_Size_Y']},{r['Grid_Size_Z']})  q{r['Queue_Id']}  {r['Kernel_Name'][:70]}")
