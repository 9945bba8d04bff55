"""The v5 per-client dedup kernels' SOURCE (rmqtt_amd/csrc/dedup.inc) on the host (tests/hipsim: one OS thread per GPU thread): tile
pass (contiguous tile ranges per block), classification and topic pass over synthetic windows,
against a first-position map: of a topic's candidates of one client the lowest position stays, every other one is flagged
(types.rs:524-539).  Topics inside one tile, topics across many tiles (more than the block has waves), more candidates than a table
holds (parts by client), tables forced to overflow (re-split), blocks that walk several items, tile counts that give the tile pass
blocks with uneven or empty ranges.  CPU only; the device twins are tests/test_deliver_parity.py."""
import numpy as np
import pytest

from tests.hipsim import sim

pytestmark = pytest.mark.skipif(sim.clang() is None, reason="hipsim needs clang++")


def window(rng, topic_hits, v5_frac, n_clients, first=12345):
    hit_off = np.concatenate([[first], first + np.cumsum(topic_hits)]).astype(np.uint64)
    nh = int(hit_off[-1]) - first
    is_c = rng.random(nh) < v5_frac
    pos = np.flatnonzero(is_c)
    cl = rng.integers(0, n_clients, size=len(pos)).astype(np.uint32)
    return hit_off, pos, cl


def expected(hit_off, pos, cl):
    rel = hit_off.astype(np.int64) - int(hit_off[0])
    t = np.searchsorted(rel, pos, side="right") - 1
    first = {}
    for p, c, tt in zip(pos.tolist(), cl.tolist(), t.tolist()):
        k = (tt, c)
        if k not in first or p < first[k]:
            first[k] = p
    return np.array(sorted(p for p, c, tt in zip(pos.tolist(), cl.tolist(), t.tolist()) if first[(tt, c)] != p), dtype=np.int64)


CASES = {
    # (topic hit counts, v5 fraction, clients, blocks of the topic pass, table slots)
    "small_topics": (lambda rng: rng.integers(0, 600, size=120), 0.3, 40, 1024, 4096),
    "config3_like": (lambda rng: rng.integers(9000, 34000, size=14), 0.1, 3000, 4, 4096),          # blocks walk several items each
    "heavy_v5": (lambda rng: np.array([30000, 100, 2500, 41000, 7, 2048, 2049, 12000]), 0.45, 500, 3, 4096),   # > 256 candidates per tile, parts by client
    "overflow": (lambda rng: rng.integers(3000, 9000, size=10), 0.2, 5000, 2, 64),                 # tables too small: re-split on the fly
    "one_block": (lambda rng: rng.integers(2100, 20000, size=9), 0.12, 100, 1, 4096),
    "more_blocks_than_items": (lambda rng: np.array([5000, 3, 9000]), 0.15, 64, 64, 4096),
    # the measured regime: duplicates are a percent of the candidates ...
    "rare_duplicates": (lambda rng: rng.integers(9000, 34000, size=10), 0.1, 150000, 4, 4096),
    # ... and its opposite: thousands of distinct duplicated clients per topic
    "many_duplicated_clients": (lambda rng: np.array([60000, 50000]), 0.5, 9000, 2, 4096),
}


@pytest.mark.parametrize("variant", [0, 3])        # the topic pass's probe_mode: linear probing + 8-byte clears (RGR_DEDUP_PROBE=0), double hashing + 16-byte clears (the product)
@pytest.mark.parametrize("case", sorted(CASES))
def test_dedup_source_on_host(case, variant):
    gen, frac, ncl, grid, slots = CASES[case]
    rng = np.random.default_rng(sum(map(ord, case)))
    hit_off, pos, cl = window(rng, np.asarray(gen(rng), dtype=np.int64), frac, ncl)
    got, n_items, _ = sim.dedup(variant, hit_off, pos, cl, grid_topic=grid, max_slots=slots)
    want = expected(hit_off, pos, cl)
    assert np.array_equal(got, want), (len(got), len(want), np.setdiff1d(got, want)[:5], np.setdiff1d(want, got)[:5])
    if case != "small_topics":
        assert n_items > 0


# ---------------------------------------------------------------------------------------------- exempt runs (r6)
EX_CASES = {
    # (topic hit counts, v5 fraction, clients, blocks of the topic pass, table slots, share of a topic its exempt run takes, exempt hits dropped by No Local)
    "config3_like": (lambda rng: rng.integers(9000, 34000, size=12), 0.1, 3000, 4, 4096, 0.7, 0.0),
    "mixed_topics": (lambda rng: np.array([30000, 100, 2500, 41000, 7, 5000, 4200, 12000, 600]), 0.3, 700, 3, 4096, 0.6, 0.05),     # topics too short for an exempt run beside long ones
    "one_candidate_beside_the_run": (lambda rng: np.array([6000, 6000, 6000, 6000]), 0.002, 8, 2, 4096, 0.9, 0.0),         # items for a single candidate
    "parts_and_overflow": (lambda rng: np.array([60000, 50000, 20000]), 0.5, 9000, 2, 64, 0.5, 0.1),                     # parts by client, tables re-split; dropped exempt hits
    "everything_exempt_but_a_few": (lambda rng: rng.integers(5000, 9000, size=9), 0.2, 400, 5, 4096, 0.98, 0.02),
}


@pytest.mark.parametrize("variant", [0, 3])
@pytest.mark.parametrize("case", sorted(EX_CASES))
def test_dedup_with_exempt_runs_on_host(case, variant):
    """The topic pass with exempt runs (kernels.hpp, kExemptMinRun): every topic long enough gets ONE run of at least 4 096 consecutive hits
    whose v5 hits have distinct clients (a run holds a client once) and are in NO candidate list; the remaining candidates ask the run's client
    index.  Expected: of ALL the v5 hits of a topic and client — listed or exempt, minus the exempt hits dropped by No Local — the lowest
    position stays, every other one is flagged."""
    gen, frac, ncl, grid, slots, share, p_drop = EX_CASES[case]
    rng = np.random.default_rng(sum(map(ord, case)) + 7)
    hits = np.asarray(gen(rng), dtype=np.int64)
    hit_off, pos, cl = window(rng, hits, frac, ncl)
    rel = hit_off.astype(np.int64) - int(hit_off[0])
    keep = np.ones(len(pos), dtype=bool)
    exempt, dropped, all_pos, all_cl = {}, [], [pos], [cl]
    for t, h in enumerate(hits.tolist()):
        ln = int(h * share)
        if ln < 4096:
            continue
        s = int(rel[t]) + int(rng.integers(0, h - ln + 1))
        inside = (pos >= s) & (pos < s + ln)
        keep &= ~inside                                           # the run's hits leave the candidate lists ...
        n5 = int(inside.sum())
        run_pos = np.sort(rng.choice(ln, size=min(n5, ncl, ln), replace=False)) + s
        run_cl = rng.choice(ncl, size=len(run_pos), replace=False).astype(np.uint32)      # ... and come back with DISTINCT clients
        clients = [None] * ln
        for p_, c_ in zip(run_pos.tolist(), run_cl.tolist()):
            clients[p_ - s] = c_
        exempt[t] = (s, clients)
        is_drop = rng.random(len(run_pos)) < p_drop
        dropped += run_pos[is_drop].tolist()
        all_pos.append(run_pos[~is_drop]); all_cl.append(run_cl[~is_drop])
    assert exempt
    all_pos[0], all_cl[0] = pos[keep], cl[keep]
    got, n_items, _ = sim.dedup(variant, hit_off, pos[keep], cl[keep], grid_topic=grid, max_slots=slots, exempt=exempt, dropped=dropped)
    want = expected(hit_off, np.concatenate(all_pos), np.concatenate(all_cl))
    assert np.array_equal(got, want), (len(got), len(want), np.setdiff1d(got, want)[:5], np.setdiff1d(want, got)[:5])
    assert n_items > 0
