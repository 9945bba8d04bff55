# GPU session r6a: the round's first look — the GPU suite on the inherited tree + the new two-rank line test, then the N = 2 line (gloo, one GPU,
# 1/10 scale) with its PMC passes, then the delivery record alone (new window choice + >= 1 % oracle sample) at full size
set -u
O=$PWD/gpurun_out/r6a
mkdir -p $O
( time timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_gpu_time.txt; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
( time timeout 1200 python3 bench.py --gpus 2 --dist-backend gloo --scale 0.1 --steps 3 --warmup 1 > $O/bench_2rank_gloo_scale0.1.json 2> $O/bench_2rank_gloo_scale0.1.err ) 2> $O/bench_2rank_time.txt; echo "bench2 rc=$?"
cp gpurun_out/bench_detail_n2.json $O/bench_detail_n2_scale0.1.json 2>/dev/null
tail -c 1500 $O/bench_2rank_gloo_scale0.1.json
( time timeout 1500 python3 bench.py --deliver 0.1 --steps 3 --warmup 1 --no-pmc > $O/bench_deliver.json 2> $O/bench_deliver.err ) 2> $O/bench_deliver_time.txt; echo "deliver rc=$?"
cp gpurun_out/bench_detail_n1.json $O/bench_detail_deliver.json 2>/dev/null
tail -c 2500 $O/bench_deliver.json
grep -E "delivery (parity|oracle)" $O/bench_deliver.err | tail -4
