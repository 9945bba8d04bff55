# GPU session r5x: DIAGNOSTIC builds — the tile pass (+ classification) takes 39.6 us per 2^27-hit window, 44 ms of a 592 ms pass, for a handful of
# flagged tiles: (a) without its per-block global atomic on the candidate counter, (b) 64 tiles per block (1 024 blocks at 2^27), (c) 128 + no atomic
set -u
O=gpurun_out/r5x
mkdir -p $O
for v in product a b c; do
  [ $v != product ] && cp tools/diag_$v.so.bin rmqtt_amd/librmqtt_gpu_router.so
  timeout 300 python bench.py --time-format deliver --steps 3 --warmup 1 > $O/deliver_$v.jsonl 2> $O/deliver_$v.err; echo "$v rc=$?"
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5x/deliver_*.jsonl")):
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
