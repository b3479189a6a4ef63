#!/bin/bash
# usage (GPU box, repo root): tools/pmc_step.sh <tag> <grids per GPU>
# rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per run, no tracing domains) over a short bench run: HBM bytes per launch of
# the HBM-bound kernels of the step.  tools/pmc_step_to_json.py condenses gpurun_out/<tag>/ into profiles/<tag>_hbm_kernels_pmc.json.
set -u
TAG=$1; B=${2:-8}
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o p -- python "$REPO/bench.py" --batch-per-gpu "$B" --no-sweep --no-cpu-baseline --no-kernel-timing --eager --steps 3 --warmup 1 > "$OUT/pmc_$C.log" 2>&1
done
cd "$REPO"
python tools/pmc_step_to_json.py "$OUT" "$TAG" "$B"
