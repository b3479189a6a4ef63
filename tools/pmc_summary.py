"""Average PMC counter values per kernel from rocprofv3 counter-collection CSVs (one sub-directory per pass)."""
import collections
import csv
import glob
import json
import sys

src = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"dispatches": max(len(v) for v in d.values())} for k, d in agg.items()}
print(json.dumps(out, indent=1))
