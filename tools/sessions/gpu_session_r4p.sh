# GPU session r4p (the round's last 1.8 GPU-minutes): which of the two IDS24 expansions is wrong on the retained path at config-5 shape,
# and on which tiles (tools/diag_lp_retain.py: tuples / LP=0 / LP=1 / runs stepped window by window, compared on the device).
set -u
mkdir -p gpurun_out/r4p
timeout 46 python tools/diag_lp_retain.py 0.5 1.0 > gpurun_out/r4p/diag.log 2>&1
echo "diag rc=$?"; tail -3 gpurun_out/r4p/diag.log | cut -c1-300
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r4p/diag.json"))
    for s in d["scales"]:
        print("scale", s["scale"], "gen", s.get("gen_s"), "table", s.get("table_s"), "done", s.get("done_s"), "total", s.get("total_mismatch"))
        for w in s["windows"][:30]:
            print("  ", {k: w[k] for k in w if k not in ("bad_tile_ids_lp0", "bad_tile_ids_lp1")})
        for t in s["tiles"][:12]:
            print("  tile", t["kernel"], t["window"], t["tile"], "len", t["len"], "np", t["np"], "n_bad", t["n_bad"], "bad@", t["bad_positions"][:12], "pairs", t["pairs"][:8])
except Exception as e:
    print("no report:", e)
PY
