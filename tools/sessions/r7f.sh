# GPU session r7f: Shared::forwards at config 2 after the runs of a pass reach the task queue under one lock acquisition (dispatch was 0.9 ms per pass)
set -u
O=$PWD/gpurun_out/r7f
mkdir -p $O
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards --e2e-sweep > $O/forwards_e2e.jsonl 2> $O/forwards_e2e.err; echo "rc=$?"
grep "forwards e2e config" $O/forwards_e2e.err | cut -c1-900
