#!/bin/bash
# usage (GPU box, repo root): tools/pmc_gemm.sh <tag> <M> <N> <K> [bias] [act]  -- PMC passes over one nmh_gemm_nt shape
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES" "FETCH_SIZE WRITE_SIZE" GRBM_GUI_ACTIVE "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  D=$(echo $C | cut -d' ' -f1)
  (cd "$REPO" && timeout 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$D" -o p -- python tools/bench_gemm_one.py "$@" > "$OUT/pmc_$D.log" 2>&1)
done
cd "$REPO"
python tools/pmc_summary.py "$OUT" | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'gemm_nt' in k: print(k); [print(f'   {c:32s} {x:16.0f}') for c,x in sorted(v.items())]
"
