#!/bin/bash
# usage (GPU box, repo root): tools/prof_cmd.sh <n> -- <command ...>   rocprofv3 kernel stats of a command, top n kernels by total time
N=$1; shift 2
export TMPDIR=/tmp
REPO=$PWD
OUT=/tmp/prof_cmd_$$
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- "$@" > $OUT.log 2>&1)
python - "$OUT" "$N" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
if not f:
    print("no stats; log tail:"); print(open(sys.argv[1] + ".log").read()[-2000:]); sys.exit(0)
for r in list(csv.DictReader(open(f[0])))[:int(sys.argv[2])]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:10.1f} us  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms")
PY
rm -rf $OUT $OUT.log
