# GPU session r7x: how long a subscribe / unsubscribe waits for the table's exclusive lock while Shared::forwards runs at full load (config 2)
set -u
O=$PWD/gpurun_out/r7x
mkdir -p $O
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards --e2e-churn > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config" $O/e2e.err | cut -c1-1500
