# GPU session r5n: tile pass of the v5 dedup with FEWER blocks (r5m: more blocks are slower: 2048 -> 0.368, 8192 -> 0.411, 32768 -> 0.70 ms of dedup per window)
set -u
O=gpurun_out/r5n
mkdir -p $O
timeout 700 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "X=0,RGR_DEDUP_TILE_GRID=1024,RGR_DEDUP_TILE_GRID=512,RGR_DEDUP_TILE_GRID=256" > $O/ab_dedup_tile_grid.jsonl 2> $O/ab_dedup_tile_grid.err; echo "deliver rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5n/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH", d.get("delivery_parity", {}).get("mismatching_words"))
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
