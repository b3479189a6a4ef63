#!/bin/bash
# usage (GPU box, repo root): tools/evidence_set.sh <tag>   -- GPU test log, bench line (+ --e2e), step tables at 8 / 4 / 1 grids, kernel stats at 8 grids and the
# swin_b line of the tree, all under gpurun_out/ (copied into profiles/ where the repo is writable)
cd /root/repo
T=${1:-rX}
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_tests.log 2>&1; tail -2 gpurun_out/${T}_gpu_tests.log
python bench.py --steps 20 --warmup 5 --e2e > gpurun_out/${T}_bench.log 2>&1; tail -1 gpurun_out/${T}_bench.log | cut -c1-300
for g in 4 1; do bash tools/step_table.sh ${T}_g$g --global-batch $g; done
bash tools/gpu_profile.sh ${T}_g8 8
python bench.py --steps 20 --warmup 5 --backbone swin_b --no-cpu-baseline --no-sweep > gpurun_out/${T}_bench_swin_b.log 2>&1; tail -1 gpurun_out/${T}_bench_swin_b.log | cut -c1-200
