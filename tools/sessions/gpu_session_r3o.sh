# GPU session r3o: last of the round — whole GPU suite + smoke on the final committed tree, and a quick bench line (config 2) after bench.py's last edits
set -u
O=gpurun_out/r3o
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -3 $O/pytest_gpu.log | cut -c1-200
( timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
( timeout 200 python bench.py --config 2 --steps 3 --warmup 1 --no-pmc --no-secondary > $O/bench_cfg2.json 2> $O/bench_cfg2.err ); python -c "
import json; d=json.load(open('$O/bench_cfg2.json')); print(d['value'], d['parity_sample']['ok'], d['cpu_baseline']['value'])"
