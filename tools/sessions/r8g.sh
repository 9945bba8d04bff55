# GPU session r8g: Shared::forwards at config 3 over the batcher's shapes (more completion threads now that a run takes the table's lock once)
set -u
O=$PWD/gpurun_out/r8g
mkdir -p $O
timeout 1800 python3 bench.py --router-e2e --e2e-configs 3 --e2e-legs forwards --e2e-sweep > $O/e2e3.jsonl 2> $O/e2e3.err; echo "rc=$?"
grep -E "e2e config" $O/e2e3.err | cut -c1-420
