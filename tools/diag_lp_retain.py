"""Diagnostic of session r4p: at config 5 the IDS24 expansion through the lane-held kernel (RGR_COMPACT_LP=1) and through the
tile-per-block kernel (RGR_COMPACT_LP=0) digested differently (profiles/r04o_*).  Which one is wrong, where, and on what kind of tile?
Four retained-path batches over the same table are stepped window by window — tuples (the format the oracle pins), IDS24 under LP=0,
IDS24 under LP=1, RUNS (the window's pair list) — the three id streams compared on the device; for the first tiles that differ the pair
list of the tile, the positions, and the ids written / expected go to gpurun_out/r4p/diag.json (rewritten as the run proceeds).
  python tools/diag_lp_retain.py [scale ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T0 = time.time()
OUT = "gpurun_out/r4p"
os.makedirs(OUT, exist_ok=True)
REPORT = {"scales": []}
TILE = 2048


def save():
    REPORT["elapsed_s"] = round(time.time() - T0, 1)
    with open(os.path.join(OUT, "diag.json.tmp"), "w") as f:
        json.dump(REPORT, f)
    os.replace(os.path.join(OUT, "diag.json.tmp"), os.path.join(OUT, "diag.json"))


def run_scale(scale):
    import torch
    import bench
    from rmqtt_amd import capi
    rep = {"scale": scale, "windows": [], "tiles": []}
    REPORT["scales"].append(rep)
    W = bench.gen_workload(5, scale)
    rep["gen_s"] = round(time.time() - T0, 1)
    r = capi.Router(device=0, collect_walk_stats=False)
    bench.build_table(r, W, W["blob"], W["offs"], None, None)
    rep["table_s"] = round(time.time() - T0, 1)
    save()
    fm = {"tuple": capi.RGR_FORMAT_TUPLE, "lp0": capi.RGR_FORMAT_IDS24, "lp1": capi.RGR_FORMAT_IDS24, "runs": capi.RGR_FORMAT_RUNS}
    B = {}
    for k, f in fm.items():
        B[k] = r.retain_batch(W["tb"], W["to"])
        B[k].set_format(f)
        os.environ["RGR_COMPACT_LP"] = "1" if k == "lp1" else "0"
        B[k].begin()
    dumped = 0
    wi = 0
    while True:
        ws = {}
        for k in fm:
            os.environ["RGR_COMPACT_LP"] = "1" if k == "lp1" else "0"
            ws[k] = B[k].next_window()
        if ws["tuple"] is None:
            break
        torch.cuda.synchronize()
        wt = ws["tuple"]
        nh = int(wt.n_hits)
        same_plan = all(ws[k] is not None and (int(ws[k].topic_begin), int(ws[k].topic_end), int(ws[k].n_hits)) == (int(wt.topic_begin), int(wt.topic_end), nh) for k in fm)
        rec = {"window": wi, "topics": [int(wt.topic_begin), int(wt.topic_end)], "hits": nh, "same_plan": bool(same_plan)}
        wi += 1
        if nh and same_plan:
            t = torch.as_tensor(bench._DevArr(wt.d_tuples, (nh, 3), "<i4"), device="cuda")
            truth = t[:, 1].contiguous().view(torch.uint8).view(nh, 4)
            rec["truth_ids_below_2_24"] = bool((truth[:, 3] == 0).all())
            truth3 = truth[:, :3]
            bad = {}
            for k in ("lp0", "lp1"):
                a = torch.as_tensor(bench._DevArr(ws[k].d_ids24, (nh, 3), "|u1"), device="cuda")
                m = (a != truth3).any(dim=1)
                bad[k] = m
                rec["mismatch_" + k] = int(m.sum())
            rep["windows"].append(rec)
            save()
            for k in ("lp0", "lp1"):
                if not rec["mismatch_" + k] or dumped >= 12:
                    continue
                pos = bad[k].nonzero()[:, 0]
                tiles = torch.unique(pos // TILE)
                rec["bad_tiles_" + k] = int(len(tiles))
                wr = ws["runs"]
                nr = int(wr.n_runs)
                roff = (torch.as_tensor(bench._DevArr(wr.d_run_off, (nr + 1,), "<i8"), device="cuda") - int(wr.offsets_bias)).cpu().numpy()
                rsrc = torch.as_tensor(bench._DevArr(wr.d_run_src, (nr,), "<i4"), device="cuda").cpu().numpy().astype(np.int64) & 0xFFFFFFFF
                tl = tiles.cpu().numpy()
                fr = np.searchsorted(roff[:-1], tl * TILE, side="right") - 1
                lr = np.searchsorted(roff[:-1], (tl + 1) * TILE - 1, side="right") - 1
                npt = lr - fr + 1
                rec["np_of_bad_tiles_" + k] = {str(int(v)): int(c) for v, c in zip(*np.unique(np.minimum(npt, 100), return_counts=True))}
                rec["ntiles"] = (nh + TILE - 1) // TILE
                rec["bad_tile_ids_" + k] = [int(x) for x in tl[:40]]
                a = torch.as_tensor(bench._DevArr(ws[k].d_ids24, (nh, 3), "|u1"), device="cuda")
                for ti, f0, l0 in list(zip(tl, fr, lr))[:6]:
                    lo_, hi_ = int(ti) * TILE, min(nh, (int(ti) + 1) * TILE)
                    got = a[lo_:hi_].to(torch.int64)
                    got = (got[:, 0] | (got[:, 1] << 8) | (got[:, 2] << 16)).cpu().numpy()
                    want = (t[lo_:hi_, 1].to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
                    d = np.flatnonzero(got != want)
                    rep["tiles"].append({"kernel": k, "window": rec["window"], "tile": int(ti), "len": hi_ - lo_, "np": int(l0 - f0 + 1),
                                         "pairs": [[int(rsrc[p]), int(roff[p] - lo_), int(roff[p + 1] - roff[p])] for p in range(int(f0), int(l0) + 1)][:200],
                                         "bad_positions": [int(x) for x in d[:400]], "n_bad": int(len(d)),
                                         "got": [int(x) for x in got], "want": [int(x) for x in want]})
                    dumped += 1
                save()
            del t, truth, truth3, bad
        else:
            rep["windows"].append(rec)
        torch.cuda.synchronize()
        if dumped >= 12:
            break
    rep["done_s"] = round(time.time() - T0, 1)
    rep["total_mismatch"] = {k: int(sum(w.get("mismatch_" + k, 0) for w in rep["windows"])) for k in ("lp0", "lp1")}
    save()
    for b in B.values():
        b.close()
    r.close()
    return rep["total_mismatch"]


if __name__ == "__main__":
    scales = [float(x) for x in sys.argv[1:]] or [0.5, 1.0]
    for s in scales:
        tm = run_scale(s)
        print("scale", s, "mismatching hits", tm, "elapsed", round(time.time() - T0, 1), flush=True)
        if any(tm.values()):
            break
