#!/bin/bash
# usage (GPU box, repo root): tools/evidence_r6.sh <tag>  -- the evidence set of a tree: GPU test log, bench line, step tables + timelines at 8 / 4 / 1 grids with the
# kernel stats of the 8-grid run, the swin_b line, the isolated weight-gradient / NT-GEMM benches of round 5 (kept for comparison), and PMC passes over one eager step per kernel family
cd /root/repo
T=${1:-r6x}
python -m pytest tests -m gpu -q > gpurun_out/${T}_gpu_tests.log 2>&1; tail -2 gpurun_out/${T}_gpu_tests.log
python bench.py --steps 20 --warmup 5 --e2e > gpurun_out/${T}_bench.log 2>&1; tail -1 gpurun_out/${T}_bench.log | cut -c1-300
for g in 8 4 1; do
  bash tools/step_timeline.sh ${T}_g$g --batch-per-gpu $g
  cp gpurun_out/${T}_g$g/step_shapes.txt gpurun_out/${T}_step_kernels_by_shape_${g}grids.txt
  cp gpurun_out/${T}_g$g/timeline.txt gpurun_out/${T}_timeline_${g}grids.txt
done
bash tools/gpu_profile.sh ${T}_stats8 8
cp gpurun_out/${T}_stats8/stats/bench_kernel_stats.csv gpurun_out/${T}_kernel_stats_8grids.csv
python bench.py --steps 20 --warmup 5 --backbone swin_b --no-cpu-baseline --no-sweep > gpurun_out/${T}_bench_swin_b.log 2>&1; tail -1 gpurun_out/${T}_bench_swin_b.log | cut -c1-200
python tools/bench_tns.py > gpurun_out/${T}_tn_stream_bench.txt 2>&1
python tools/bench_tng_stage2.py > gpurun_out/${T}_tng_stage2.txt 2>&1
bash tools/pmc_kernel.sh ${T}_pmc "NONE" nerf-mae_amd/csrc/norm.hip -- python bench.py --batch-per-gpu 8 --eager --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-sweep > gpurun_out/${T}_pmc.log 2>&1
python tools/pmc_families.py gpurun_out/${T}_pmc ${T} 8 >> gpurun_out/${T}_pmc.log 2>&1
find gpurun_out/${T}_pmc -name '*.csv' -size +20M -delete
rm -rf gpurun_out/${T}_pmc/pmc_*/
tail -3 gpurun_out/${T}_pmc.log
