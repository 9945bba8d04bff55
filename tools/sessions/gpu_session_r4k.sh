# GPU session r4k: the tree of the final commit (r4i's code; the r4j experiment reverted): whole GPU suite + smoke, nothing else
set -u
O=gpurun_out/r4k
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log | cut -c1-300
( timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
