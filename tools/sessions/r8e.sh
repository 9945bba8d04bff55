# GPU session r8e: __graft_entry__.build() + smoke() on the last tree, and the N = 2 self-launch (gloo, one GPU, 1/10 scale)
set -u
O=$PWD/gpurun_out/r8e
mkdir -p $O
timeout 900 python3 -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.txt | cut -c1-300
timeout 1500 python3 bench.py --gpus 2 --dist-backend gloo --scale 0.1 --steps 3 --warmup 1 > $O/bench_n2.json 2> $O/bench_n2.err; echo "n2 rc=$?"
python3 - <<PY
import json
d=json.loads(open("$O/bench_n2.json").read().strip().splitlines()[-1])
print(d["value"], d["n_gpus"], d["scaling"], d["config"].get("gather"), d.get("exchange"), (d.get("cpu_baseline") or {}).get("value"), (d.get("parity_sample") or {}).get("ok"), d["config"].get("rccl_ranks"))
PY
