#!/bin/bash
# usage (GPU box, repo root): tools/pmc_extra.sh <kernel name substring> "<counters of pass 1>" ["<counters of pass 2>" ...] -- <command ...>
# extra rocprofv3 --pmc passes (one counter group per run, no tracing domains); prints the per-dispatch mean of every counter for the matching kernels
KN=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
export TMPDIR=/tmp
REPO=$PWD
i=0
for C in "${GROUPS_[@]}"; do
  i=$((i+1))
  (cd "$REPO" && timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcx_$i -o p -- "$@") > /tmp/pmcx_$i.log 2>&1
done
python - "$KN" <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcx_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:40s} {sum(v) / len(v):16.0f}  ({len(v)} dispatches)")
PY
