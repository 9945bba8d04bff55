#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 PMC counters over one command (counters + kernel trace only: no sys / hip / hsa trace domains).

    python tools/pmc_kernels.py --out gpurun_out/x.json --groups "SQ_WAVE_CYCLES,SQ_WAIT_ANY;SQ_INSTS_LDS,..." -- python bench.py ...

Every `;`-separated group is one pass of the command (a group must fit the block's counter slots: 8 SQ, 4 TCC).  Output: per kernel
(name cut at the first '('), per counter the sum over its dispatches, the dispatch count and the mean duration under the counters."""
import argparse, csv, glob, json, os, shutil, subprocess, sys, tempfile


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--groups", required=True)
    ap.add_argument("--match", default="", help="only kernels whose name contains one of these comma-separated needles")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    needles = [n for n in a.match.split(",") if n]
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    res = {}
    here = os.getcwd()
    for gi, grp in enumerate(a.groups.split(";")):
        counters = [c for c in grp.split(",") if c]
        work = tempfile.mkdtemp(prefix="pmck_")
        full = [rocprof, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", work, "-o", "pmc", "--"] + cmd
        # (rocprofv3 is run from the temp directory, like bench.py's own PMC passes: relative paths of the command are made absolute)
        full = [os.path.join(here, x) if x.endswith(".py") and not os.path.isabs(x) and os.path.exists(os.path.join(here, x)) else x for x in full]
        p = subprocess.run(full, cwd=tempfile.gettempdir(), env=dict(os.environ, TMPDIR=tempfile.gettempdir(), PYTHONPATH=here + os.pathsep + os.environ.get("PYTHONPATH", "")),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        print(f"group {gi} {counters}: rc={p.returncode}", file=sys.stderr)
        print("  stdout tail: " + p.stdout.decode(errors="replace")[-300:].replace("\n", " | "), file=sys.stderr)
        print("  stderr tail: " + p.stderr.decode(errors="replace")[-600:].replace("\n", " | "), file=sys.stderr)
        csvs = glob.glob(os.path.join(work, "**", "*.csv"), recursive=True)
        print(f"  csv files: {[os.path.relpath(c, work) for c in csvs][:6]}", file=sys.stderr)
        for f in glob.glob(os.path.join(work, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rgr::", "")
                if needles and not any(n in name for n in needles):
                    continue
                k = res.setdefault(name, {"dispatches": {}, "ns": {}, "counters": {}})
                c = row["Counter_Name"]
                k["counters"][c] = k["counters"].get(c, 0.0) + float(row["Counter_Value"])
                key = (gi, row["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    k["dispatches"][gi] = k["dispatches"].get(gi, 0) + 1
                    k["ns"][gi] = k["ns"].get(gi, 0) + int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        shutil.rmtree(work, ignore_errors=True)
    out = {}
    for name, k in res.items():
        n = max(k["dispatches"].values())
        out[name] = {"dispatches": n, "avg_us_under_pmc": round(sum(k["ns"].values()) / max(1, sum(k["dispatches"].values())) / 1e3, 2),
                     "per_dispatch": {c: round(v / n, 1) for c, v in sorted(k["counters"].items())}}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    for name, k in sorted(out.items(), key=lambda kv: -kv[1]["avg_us_under_pmc"] * kv[1]["dispatches"]):
        print(name, k["dispatches"], k["avg_us_under_pmc"], k["per_dispatch"])


if __name__ == "__main__":
    main()
