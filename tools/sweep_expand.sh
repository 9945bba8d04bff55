#!/bin/bash
# Tuning sweep of the expand kernel geometry on the GPU box (rebuilds the .so per variant).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for V in "-DRGR_EXPAND_THREADS=256 -DRGR_EXPAND_PER_THREAD=8" "-DRGR_EXPAND_THREADS=256 -DRGR_EXPAND_PER_THREAD=4" \
         "-DRGR_EXPAND_THREADS=256 -DRGR_EXPAND_PER_THREAD=16" "-DRGR_EXPAND_THREADS=512 -DRGR_EXPAND_PER_THREAD=4" \
         "-DRGR_EXPAND_THREADS=512 -DRGR_EXPAND_PER_THREAD=8" "-DRGR_EXPAND_THREADS=128 -DRGR_EXPAND_PER_THREAD=16" \
         "-DRGR_EXPAND_THREADS=256 -DRGR_EXPAND_PER_THREAD=8 -DRGR_EXPAND_NT=1" "-DRGR_EXPAND_THREADS=1024 -DRGR_EXPAND_PER_THREAD=2"; do
  RGR_EXTRA_FLAGS="$V" python -c "from rmqtt_amd import build; build.build_gpu(force=True)" 2>&1 | grep -i error
  R=$(timeout 300 python bench.py --scale ${SWEEP_SCALE:-0.25} --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['expand_GBps'], j['kernel_ms_per_step']['expand'], j['value'])")
  echo "$V => expand_GBps, expand_ms/step, matches/s: $R" | tee -a gpurun_out/sweep_expand.log
done
python -c "from rmqtt_amd import build; build.build_gpu(force=True)"
