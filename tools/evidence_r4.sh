#!/bin/bash
# usage (GPU box, repo root): tools/evidence_r4.sh <tag>  -- the evidence set of a tree: GPU test log, bench line (+ --e2e), step tables at 8 / 4 / 1 grids with the
# kernel stats of the 8-grid run, the swin_b line, the isolated fused-vs-unfused Swin-block timings, and PMC passes over one eager step per kernel family
cd $GRAFT_REPO_ROOT
T=${1:-r4x}
bash tools/evidence_set.sh $T
python tools/bench_swin_block.py 8 > gpurun_out/${T}_swin_block_bench.txt 2>&1
python tools/bench_swin_cold.py 8 >> gpurun_out/${T}_swin_block_bench.txt 2>&1
bash tools/pmc_kernel.sh ${T}_pmc "NONE" nerf-mae_amd/csrc/norm.hip -- python bench.py --batch-per-gpu 8 --eager --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-sweep > gpurun_out/${T}_pmc.log 2>&1
python tools/pmc_families.py gpurun_out/${T}_pmc ${T} >> gpurun_out/${T}_pmc.log 2>&1
find gpurun_out/${T}_pmc -name '*.csv' -size +20M -delete
tail -3 gpurun_out/${T}_pmc.log
bash tools/ab_round4.sh $T > gpurun_out/${T}_ab.log 2>&1
tail -3 gpurun_out/${T}_ab.log
