# GPU session r2b: new tests (formats, group, match_filters, $share mirror), FETCH calibration, bench with compact formats,
# host-out latency at config 3 full size, 2-rank gloo bench on one GPU, RCCL world-1 self test is part of pytest.
set -u
O=gpurun_out/r2b
mkdir -p $O/calib
R=$(pwd)
( timeout 900 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
tail -4 $O/pytest_gpu.log
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/calib -o pmc -- $R/tools/membench calib > $R/$O/calib/calib.txt 2> $R/$O/calib.err )
python tools/pmc_calibrate.py $O/calib --write $O/pmc_calibration.json > $O/calibration.txt 2>&1; cat $O/calibration.txt
[ -f $O/pmc_calibration.json ] && cp $O/pmc_calibration.json profiles/pmc_calibration.json
( timeout 1200 python bench.py --steps 5 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err )
tail -3 $O/bench_default.err
( timeout 600 python tools/latency.py 3 1.0 > $O/latency_cfg3_full.txt 2>&1 ); tail -12 $O/latency_cfg3_full.txt
( timeout 300 python tools/latency.py 2 1.0 > $O/latency_cfg2.txt 2>&1 ); tail -12 $O/latency_cfg2.txt
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --config 3 --scale 0.1 --dist-backend gloo --gather tuples > $O/bench_2rank_gloo_tuples.json 2> $O/bench_2rank_gloo_tuples.err ); tail -2 $O/bench_2rank_gloo_tuples.err; cat $O/bench_2rank_gloo_tuples.json | head -c 600
( timeout 300 python bench.py --steps 3 --warmup 1 --config 3 --scale 0.1 --no-pmc --no-secondary > $O/bench_1rank_scale0.1.json 2> $O/bench_1rank_scale0.1.err )
