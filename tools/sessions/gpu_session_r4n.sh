# GPU session r4n (6 GPU-minutes left in the round): the lane-held compact expansion (RGR_COMPACT_LP, expand_compact.inc) against the
# tile-per-block kernel — ids24 and packed timed on ONE table build, the fastest variant of each digested over a full pass against the
# product kernel — then its GPU parity test.  The kernel source was validated on the host first (tests/test_hipsim_expand.py,
# tools/hipsim_sanitizers.sh).  2^31-hit windows ride along as the last variants (RGR_WINDOW_HITS).
set -u
O=gpurun_out/r4n
mkdir -p $O
V="RGR_COMPACT_LP=0,RGR_COMPACT_LP=4,RGR_COMPACT_LP=2,RGR_COMPACT_LP=1,RGR_COMPACT_LP=0+RGR_WINDOW_HITS=2147483648,RGR_COMPACT_LP=4+RGR_WINDOW_HITS=2147483648"
timeout 230 python bench.py --time-format ids24,packed --steps 4 --warmup 2 --ab-env "$V" > $O/ab_lp.jsonl 2> $O/ab_lp.err
echo "ab rc=$?"; python - <<'PY'
import json
for l in open("gpurun_out/r4n/ab_lp.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    if "ab_check" in d: print("CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH", d["hits"], d["seconds"])
    else: print(d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d["windows_per_step"])
PY
tail -3 $O/ab_lp.err | cut -c1-300
( timeout 150 python -m pytest tests/test_formats_gpu.py -q -x --timeout 120 -k "lane_held or equal_tuples" > $O/pytest_lp.log 2>&1; echo "pytest rc=$?" >> $O/pytest_lp.log ); tail -5 $O/pytest_lp.log | cut -c1-300
