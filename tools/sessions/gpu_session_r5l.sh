# GPU session r5l: does the next chunk's preparation have to cost the expansions what it costs on its own?  Low-priority stream / CU mask for
# the preparation, IDS24 and PACKED passes (one table)
set -u
O=gpurun_out/r5l
mkdir -p $O
timeout 900 python bench.py --time-format ids24,packed --steps 4 --warmup 1 --ab-env "X=0,RGR_PREP_PRIORITY=low,RGR_PREP_CUS=32,RGR_PREP_CUS=64,RGR_PREP_CUS=128" > $O/ab_prep_stream.jsonl 2> $O/ab_prep_stream.err; echo "rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5l/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH")
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"])
PY
tail -3 $O/ab_prep_stream.err | cut -c1-300
