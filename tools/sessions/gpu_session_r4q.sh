# GPU session r4q (the round's last GPU minute): the retained-path diagnostic again, after the fix (every lane executes every
# cross-lane read) — expected: 0 mismatching hits for both kernels.
set -u
mkdir -p gpurun_out/r4q
sed -i 's#gpurun_out/r4p#gpurun_out/r4q#' tools/diag_lp_retain.py
timeout 17 python tools/diag_lp_retain.py 0.5 > gpurun_out/r4q/diag.log 2>&1
echo "diag rc=$?"; tail -2 gpurun_out/r4q/diag.log | cut -c1-300
