#!/bin/bash
# Walk-kernel geometry sweep on config 2 (walk-dominated).  Rebuilds the .so per variant.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for V in "-DRGR_WALK_THREADS=256 -DRGR_WALK_WINDOW=2560" "-DRGR_WALK_THREADS=128 -DRGR_WALK_WINDOW=1280" "-DRGR_WALK_THREADS=64 -DRGR_WALK_WINDOW=640" \
         "-DRGR_WALK_THREADS=512 -DRGR_WALK_WINDOW=5120" "-DRGR_WALK_THREADS=256 -DRGR_WALK_WINDOW=1536" "-DRGR_WALK_THREADS=256 -DRGR_WALK_WINDOW=4096" \
         "-DRGR_WALK_THREADS=128 -DRGR_WALK_WINDOW=2048"; do
  RGR_EXTRA_FLAGS="$V" python -c "from rmqtt_amd import build; build.build_gpu(force=True)" 2>&1 | grep -i " error"
  R=$(timeout 200 python bench.py --config 2 --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['kernel_ms_per_step']['walk'], j['hits_per_step'], j['value'])")
  R3=$(timeout 200 python bench.py --config 3 --scale 0.1 --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['kernel_ms_per_step']['walk'], j['hits_per_step'])")
  echo "$V => cfg2 walk_ms, hits, matches/s: $R | cfg3x0.1 walk_ms, hits: $R3" | tee -a gpurun_out/sweep_walk.log
done
python -c "from rmqtt_amd import build; build.build_gpu(force=True)"
