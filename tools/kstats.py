"""Print the top rows of a rocprofv3 kernel-stats CSV (ms per bench step)."""
import csv
import glob
import sys

src, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f, "total ms/step", tot / 1e6 / steps)
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} {float(r['AverageNs']) / 1e3:9.1f} {float(r['MaxNs']) / 1e3:9.1f}")
