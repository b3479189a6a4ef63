#!/bin/bash
# usage (GPU box, repo root): tools/step_timeline.sh <out-name> [bench.py args...]   kernel trace of a bench run -> gpurun_out/<out-name>/{step_shapes,timeline}.txt
NAME=$1; shift
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/$NAME
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/st_$NAME -o t -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-sweep "$@" > $OUT/b.log 2>&1)
python $REPO/tools/trace_shapes.py /tmp/st_$NAME 80 > $OUT/step_shapes.txt 2>&1
python $REPO/tools/trace_timeline.py /tmp/st_$NAME > $OUT/timeline.txt 2>&1
tail -1 $OUT/b.log | cut -c1-200
