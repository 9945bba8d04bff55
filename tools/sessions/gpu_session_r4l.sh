# GPU session r4l: HBM traffic of the ids24 / packed expansions from the PMC counters (FETCH_SIZE, WRITE_SIZE: separate passes, kernel trace only)
set -u
O=gpurun_out/r4l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for fmt in ids24 packed; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${fmt}_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --time-format $fmt --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/$O/pmc_${fmt}_$c.json 2> $GRAFT_REPO_ROOT/$O/pmc_${fmt}_$c.err
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json
out = {}
for fmt in ("ids24", "packed"):
    rec = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot = 0.0; n = 0; dur = 0
        for f in glob.glob(f"gpurun_out/r4l/pmc_{fmt}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == c and "expand_compact_kernel" in r["Kernel_Name"]:
                    tot += float(r["Counter_Value"]); n += 1; dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        rec[c] = {"KiB": tot, "dispatches": n, "avg_us_under_pmc": round(dur / max(1, n) / 1e3, 1)}
    try:
        hits = json.load(open(f"gpurun_out/r4l/pmc_{fmt}_WRITE_SIZE.json"))["hits_per_step"]
    except Exception:
        hits = 148150579430
    rec["hits"] = hits
    rec["write_B_per_hit"] = round(rec["WRITE_SIZE"]["KiB"] * 1024 / hits, 4)
    rec["fetch_raw_B_per_hit"] = round(rec["FETCH_SIZE"]["KiB"] * 1024 / hits, 4)
    out[fmt] = rec
print(json.dumps(out))
open("gpurun_out/r4l/compact_formats_pmc_traffic.json", "w").write(json.dumps(out, indent=1))
PY
find $O -name "*.csv" -size +2M -delete
du -sh $O
