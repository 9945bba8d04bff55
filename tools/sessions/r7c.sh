# GPU session r7c: Shared::forwards through the boundary at config 2 — where the batcher's threads spend their time (collect / device pass / dispatch)
set -u
O=$PWD/gpurun_out/r7c
mkdir -p $O
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards --e2e-sweep > $O/forwards_e2e.jsonl 2> $O/forwards_e2e.err; echo "rc=$?"
grep "forwards e2e config" $O/forwards_e2e.err | cut -c1-900
