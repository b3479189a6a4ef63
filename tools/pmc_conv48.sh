#!/bin/bash
# usage (GPU box, repo root): tools/pmc_conv48.sh <tag> <grids>
# Separate rocprofv3 --pmc passes (one counter group per run, no tracing domains) over tools/bench_conv48.py, then the raw rocm-smi
# clock / power samples while each kernel loops.  Everything lands in gpurun_out/<tag>/; tools/pmc_to_json.py condenses it into
# profiles/<tag>_conv48_pmc.json + profiles/<tag>_conv48_wgrad_pmc.json (with the sha256 of csrc/conv48.hip the numbers belong to).
set -u
TAG=$1; B=${2:-4}
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" GRBM_GUI_ACTIVE; do
  D=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$D" -o p -- python "$REPO/tools/bench_conv48.py" "$B" > "$OUT/pmc_$D.log" 2>&1
done
cd "$REPO"
# raw clock / power samples under load (fwd, then wgrad)
for K in fwd wgrad; do
  python tools/loop_conv48.py $K "$B" 9 &
  PID=$!
  sleep 5
  for i in 1 2 3 4; do echo "--- sample $i ($K, $(date +%s.%N))"; rocm-smi --showclocks --showpower 2>&1; sleep 0.7; done > "$OUT/rocm_smi_$K.txt"
  wait $PID
done
python tools/pmc_to_json.py "$OUT" "$TAG" "$B"
