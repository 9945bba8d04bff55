"""bench.py's ONE stdout line stays small enough for the driver's parser and keeps the contract's keys.

Round 4's driver run printed the full record (27 KB) and BENCH_r04.json recorded `parsed: null`; the line is now the compact record
(bench.compact_line) and the full record goes to gpurun_out/bench_detail_n<N>.json.  The fixture is the full record of round 4's default
run (profiles/r04i_bench_default.json)."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.path.join(ROOT, "profiles", "r04i_bench_default.json")

CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"]


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def strict_loads(s):
    def bad(c):
        raise ValueError(f"non-standard JSON constant {c}")
    return json.loads(s, parse_constant=bad)


def test_compact_line_of_the_full_default_record(bench):
    full = json.load(open(FULL))
    line = bench.compact_line(full, "gpurun_out/bench_detail_n1.json")
    assert "\n" not in line
    assert len(line) <= bench.LINE_BUDGET
    d = strict_loads(line)
    for k in CONTRACT:
        assert k in d and d[k] == full[k], k
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert r[k] == full["roofline"][k]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c
    assert d["parity_sample"]["ok"] is True and d["parity_sample"]["exhaustive"] is True
    assert d["parity_sample"]["topics"] == full["config"]["publishes"]
    assert [f["format"] for f in d["compact_formats"]] == ["soa", "packed", "ids24", "runs"]
    assert len(d["secondary"]) == len(full["secondary"])
    for s, fs in zip(d["secondary"], full["secondary"]):
        assert s["value"] == fs["value"] and s["metric"] == fs["metric"]
        if "roofline" in fs:
            assert s["roofline"]["frac"] == fs["roofline"]["frac"]
        if "cpu_baseline" in fs:
            assert s["cpu_baseline"]["value"] == fs["cpu_baseline"]["value"]


def test_compact_line_sheds_secondary_detail_before_it_outgrows_the_budget(bench):
    full = json.load(open(FULL))
    full["secondary"] = full["secondary"] * 6          # far more than any run produces
    line = bench.compact_line(full, None)
    assert len(line) <= bench.LINE_BUDGET
    d = strict_loads(line)
    for k in CONTRACT + ["roofline", "cpu_baseline", "parity_sample"]:
        assert k in d


def test_error_secondary_survives(bench):
    full = json.load(open(FULL))
    full["secondary"] = [{"config": {"workload": "x"}, "error": "RuntimeError('boom')"}]
    d = strict_loads(bench.compact_line(full, None))
    assert d["secondary"][0]["error"].startswith("RuntimeError")


def test_pmc_kernel_classes_cover_every_tuple_expansion_the_library_launches(bench):
    """The PMC replay attributes dispatches to classes by kernel name; a default kernel whose name matches no needle loses the run's whole
    roofline record (round 5: expand_deliver_lean_kernel).  Every kernel launch_expand() can launch must be an "expand" dispatch, no compact
    expansion and no preparation kernel may be."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rmqtt_amd", "csrc", "kernels.hip")).read()
    body = src[src.index("void launch_expand("):]
    body = body[:body.index("\n}\n")]
    launched = set(re.findall(r"(\w+_kernel)<[^;]*?><<<", body))
    assert {"expand_kernel", "expand_deliver_early_kernel", "expand_deliver_lean_kernel"} <= launched
    needles = dict(bench.KCLASS)["expand"]
    for k in launched:
        assert any(nd in k + "<512, 4>" for nd in needles), k
    for other in ("expand_compact_kernel<2, 1>", "expand_compact_lp_kernel<4, 1>", "walk_kernel<false>", "tiles_kernel", "dedup_topic_kernel<3u>"):
        assert not any(nd in other for nd in needles), other


@pytest.mark.gpu
def test_two_rank_line_is_creditable():
    """`python bench.py --gpus 2` (self-launch; gloo because the box has one GPU) at 1/50 scale: the N > 1 line carries what the N = 1 line
    does — a CPU baseline timed on rank 0's host against the unsharded table, an exhaustive parity sample — and the exchange step is
    BASELINE configs[3]'s all-gatherv of subscriber hits in its run-descriptor form BY DEFAULT, with its bytes and milliseconds reported."""
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--scale", "0.02", "--steps", "2", "--warmup", "1", "--no-pmc"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = strict_loads(lines[0])
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2
    assert d["config"]["gather"] == "runs" and d["config"]["collective"] == "torch.distributed" and d["config"]["dist_backend"] == "gloo"
    assert "rccl_ranks" in d["config"]                      # null under gloo; ncclCommCount of the library's communicator under nccl
    c = d["cpu_baseline"]
    assert c and c["value"] > 0 and c["cores"] >= 1 and c["kind"] == "port" and c["single_thread"] > 0
    ps = d["parity_sample"]
    assert ps["ok"] is True and ps["exhaustive"] is True and ps["topics"] == d["config"]["publishes"]
    x = d["exchange"]
    assert x["describes_every_hit"] is True and x["hits_described"] == d["hits_per_step"]
    assert x["ms_per_step"] > 0 and x["bytes_all_ranks_per_step"] == x["runs_all_ranks"] * 16
    assert len(d["shard_hits"]) == 2 and sum(d["shard_hits"]) == d["hits_per_step"]
    assert d["roofline"]["alg_frac"] > 0 and d["roofline"]["hits_per_launch"] > 0
