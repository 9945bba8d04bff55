#!/bin/bash
# HBM traffic counters for the bench kernels, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (TCC slot budget), counters only
# (no sys/hip/hsa trace domains).  Run on the GPU box from the repo root:
#   bash profiles/collect_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$C" -o pmc -- \
      python "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-sample 0 "$@" > "$OUT/$C.json" 2> "$OUT/$C.err"
done
cd "$ROOT"
python profiles/summarize_pmc.py "$OUT" | tee "$OUT/summary.txt"
