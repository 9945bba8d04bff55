# GPU session r5z: lean delivery expansion 512 x 4 vs 256 x 8 at the 2^27-hit windows that became the default after r5i measured them at 2^28
set -u
O=gpurun_out/r5z
mkdir -p $O
timeout 500 python bench.py --time-format deliver --steps 4 --warmup 1 --ab-env "X=0,RGR_DELIVER_LEAN=2,X=1,RGR_DELIVER_LEAN=2" > $O/ab_deliver_lean_geometry_2e27.jsonl 2> $O/ab.err; echo "rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5z/ab_*.jsonl")):
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["ab_check"], "ok" if d["ok"] else "MISMATCH", d.get("delivery_parity", {}).get("mismatching_words"))
        else: print("  ", d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
