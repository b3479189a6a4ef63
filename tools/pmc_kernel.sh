#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernel.sh <tag> <kernel name prefix> <source file the numbers belong to> -- <command ...>
# Separate rocprofv3 --pmc passes (one counter group per run, no tracing domains: the combination gpurun refuses) over <command>; the per-dispatch
# means of the kernels whose name starts with <prefix> go to gpurun_out/<tag>_pmc.json (copy it to profiles/).
set -u
TAG=$1; PREFIX=$2; SRCFILE=$3; shift 4
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" GRBM_GUI_ACTIVE; do
  D=$(echo $C | cut -d' ' -f1)
  (cd "$REPO" && timeout 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$D" -o p -- "$@") > "$OUT/pmc_$D.log" 2>&1
done
cd "$REPO"
python tools/pmc_kernel_to_json.py "$OUT" "$TAG" "$PREFIX" "$SRCFILE"
