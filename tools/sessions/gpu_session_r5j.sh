# GPU session r5j: the lean delivery expansion as the default — its floor (no v5 subscription at all) and 10 % / 30 % v5 against the r5a kernel
set -u
O=gpurun_out/r5j
mkdir -p $O
for f in 0.0 0.3; do
timeout 400 python bench.py --time-format deliver --deliver $f --steps 3 --warmup 1 --ab-env "RGR_DELIVER_LEAN=0,X=0" > $O/ab_deliver_lean_v5frac$f.jsonl 2> $O/ab_deliver_lean_v5frac$f.err; echo "deliver $f rc=$?"
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5j/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH", d.get("delivery_parity", {}).get("mismatching_words"))
        else: print("  ", d["format"], d.get("env"), d.get("v5_frac"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
