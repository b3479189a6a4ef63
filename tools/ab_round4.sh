#!/bin/bash
# usage (GPU box, repo root): tools/ab_round4.sh <tag>  -- same-box A/B of the round's switches inside the replayed step (ms per step, 30 steps each) and the
# phase timestamps of the fused forward kernels
cd $GRAFT_REPO_ROOT
T=${1:-r4x}
O=gpurun_out/${T}_swin_ab_in_step.txt
: > $O
run() { # label, grids, env...
  L=$1; G=$2; shift 2
  ms=$(env "$@" python bench.py --batch-per-gpu $G --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-sweep 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$G grids  $L  ($*)  $ms ms" | tee -a $O
}
for G in 8 4 1; do
  run "default (fused forward, split MLP, deferred LN parameter gradients, token-ordered backward at stage 2)" $G X=0
  run "unfused encoder forward" $G NMH_SWIN=0
done
run "default, second run" 8 X=0
run "one workgroup per row tile in the stage-2 MLP forward" 8 NMH_SWIN_SPLIT=0
run "LayerNorm parameter gradients by atomics in the launch" 8 NMH_LN_DEFER=0
run "window-ordered backward of the attention branch at stage 2" 8 NMH_TOKEN_BWD=0
run "default, third run" 8 X=0
NMH_SWIN_DBG=8 python tools/swin_phase_cycles.py 8 > gpurun_out/${T}_swin_phase_cycles.txt 2>&1
tail -4 gpurun_out/${T}_swin_phase_cycles.txt
