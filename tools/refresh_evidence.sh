#!/bin/bash
# usage (GPU box, repo root): tools/refresh_evidence.sh <tag>  -- the evidence set of a tree: conv48 / HBM-kernel PMC passes, rocprofv3 step tables at
# 8 / 4 / 1 grids, the bench line (with --e2e) and the swin_b line, all under gpurun_out/; tools/pmc_to_json.py / pmc_step_to_json.py and a copy of the
# summaries into profiles/ are run afterwards where the repo is writable
cd /root/repo
T=${1:-r2n}
bash tools/pmc_conv48.sh ${T} 4 > gpurun_out/${T}_pmc_conv48.log 2>&1
bash tools/pmc_step.sh ${T}s 8 > gpurun_out/${T}_pmc_step.log 2>&1
for g in 8 4 1; do bash tools/gpu_profile.sh ${T}_g$g $g > gpurun_out/${T}_prof_$g.log 2>&1; done
python bench.py --steps 20 --warmup 5 --e2e > gpurun_out/${T}_bench.log 2>&1
python bench.py --steps 20 --warmup 5 --backbone swin_b --no-cpu-baseline > gpurun_out/${T}_bench_swin_b.log 2>&1
tail -1 gpurun_out/${T}_bench.log | cut -c1-400
