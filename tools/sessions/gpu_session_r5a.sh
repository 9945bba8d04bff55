# GPU session r5a (prepared at the end of round 4, when the round's GPU minutes were spent): every variant that was written and
# checked on the host (tests/hipsim) but never measured, A/B'd on ONE table build each.  Defaults are the measured kernels; a variant
# that wins here becomes the default (and its row in DESIGN §14 moves to §15), one that loses goes to §10.  ~6 GPU-minutes.
set -u
O=gpurun_out/r5a
mkdir -p $O
# 1. the experimental parity tests first: a variant that is wrong is not worth timing
( RMQTT_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_deliver_parity.py tests/test_formats_gpu.py -m gpu -q -x --timeout 200 > $O/pytest_experimental.log 2>&1; echo "pytest rc=$?" >> $O/pytest_experimental.log ); tail -4 $O/pytest_experimental.log | cut -c1-300
# 2. delivery stage (config 3, 10 % v5): topic pass pipelined, expansion with early loads, 2^30-hit windows, batched preparation
D="X=0,RGR_DEDUP_PIPE=1,RGR_DELIVER_EARLY=1,RGR_DEDUP_PIPE=1+RGR_DELIVER_EARLY=1,RGR_DEDUP_PIPE=1+RGR_DELIVER_EARLY=1+RGR_DELIVER_WINDOW_HITS=1073741824,RGR_DEDUP_PIPE=1+RGR_DELIVER_EARLY=1+RGR_DELIVER_WINDOW_HITS=1073741824+RGR_PREP_BATCH=1"
timeout 400 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "$D" > $O/ab_deliver.jsonl 2> $O/ab_deliver.err; echo "deliver rc=$?"
# 3. batched preparation under the compact formats and the tuple headline
timeout 300 python bench.py --time-format ids24,packed,tuple --steps 4 --warmup 2 --ab-env "X=0,RGR_PREP_BATCH=1" > $O/ab_prep.jsonl 2> $O/ab_prep.err; echo "prep rc=$?"
# 3b. IDS24 through 16-byte stores; tile records of the next window fused into the expansion (last: its host-side window plan is the one
#     piece of new code no test has run — a fault there must not cost the other variants their numbers)
timeout 200 python bench.py --time-format ids24 --steps 5 --warmup 2 --ab-env "X=0,RGR_IDS24_X4=1,RGR_COMPACT_LP=2,RGR_COMPACT_LP=0,RGR_TILES_FUSED=1,RGR_TILES_FUSED=1+RGR_PREP_BATCH=1" > $O/ab_ids24_x4.jsonl 2> $O/ab_ids24_x4.err; echo "x4 rc=$?"
# 4. the lane-held IDS24 kernel where every tile holds many runs (config 3 at 1/10 scale: ~100 hits per run)
timeout 200 python bench.py --scale 0.1 --time-format ids24 --steps 6 --warmup 2 --ab-env "RGR_COMPACT_LP=0,RGR_COMPACT_LP=1,RGR_COMPACT_LP=4" > $O/ab_lp_scale0.1.jsonl 2> $O/ab_lp_scale0.1.err; echo "scale0.1 rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5a/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH")
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
