# GPU session r8i: the whole GPU suite on the very last tree of the round
set -u
O=$PWD/gpurun_out/r8i
mkdir -p $O
( time timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_gpu_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
