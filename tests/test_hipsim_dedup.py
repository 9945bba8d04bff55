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


@pytest.mark.parametrize("variant", [0, 3, 7])        # (7, the product since r6s: a tile's first 256 candidates in one request) the topic pass's probe_mode: linear probing + 8-byte clears (RGR_DEDUP_PROBE=0), double hashing + 16-byte clears (the product)
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

