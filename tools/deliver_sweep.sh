# Where does the delivery variant of the expansion spend its 0.43 ms over the plain kernel (1.01 vs 0.58 ms per 2^28-hit window at 10 % v5)?
# One library build per diagnostic switch (kernels.hip RGR_DIAG_*; results of such builds are wrong on purpose), each timed with the same bench.
set -u
O=gpurun_out/r3h
mkdir -p $O
for v in BASE NO_ATTRS NO_CAND_STORE; do
  if [ $v = BASE ]; then export RGR_EXTRA_FLAGS=""; else export RGR_EXTRA_FLAGS="-DRGR_DIAG_$v"; fi
  python -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > $O/build_$v.log 2>&1
  ( timeout 300 python bench.py --config 3 --steps 2 --warmup 1 --no-pmc --no-secondary --no-d2h --deliver 0.1 > $O/deliver_$v.json 2> $O/deliver_$v.err )
  python - <<PY
import json
try:
    d=json.load(open("$O/deliver_$v.json")); k=d["kernel_ms_per_step"]; w=d["config"]["windows_per_step"]
    print("$v", "expand ms/window", round(k["expand"]/w,3), "dedup ms/window", round(d["delivery_stage"]["dedup_ms_per_step"]/w,3), "matches/s", d["value"])
except Exception as e:
    print("$v", "failed", e)
PY
done 2>&1 | tee $O/deliver_sweep.txt
export RGR_EXTRA_FLAGS=""
python -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > $O/build_restore.log 2>&1
