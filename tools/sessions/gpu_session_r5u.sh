# GPU session r5u: DIAGNOSTIC — the candidate list's tile slices 512 entries apart instead of 2 048 (wrong for a tile with more than 512 candidates;
# none has at 10 % v5): is the sparse 16 KB-strided layout what makes a delivery window's size matter?  Same box, product library first.
set -u
O=gpurun_out/r5u
mkdir -p $O
AB="X=0,RGR_DELIVER_WINDOW_HITS=268435456,RGR_DELIVER_WINDOW_HITS=1073741824"
timeout 500 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "$AB" > $O/ab_product_stride2048.jsonl 2> $O/ab_product.err; echo "product rc=$?"
cp tools/diag_cand_stride512.so.bin rmqtt_amd/librmqtt_gpu_router.so
timeout 500 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "$AB" > $O/ab_diag_stride512.jsonl 2> $O/ab_diag.err; echo "diag rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5u/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH", d.get("delivery_parity", {}).get("mismatching_words"))
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
