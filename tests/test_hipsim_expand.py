"""The compact expansion kernels' SOURCE (rmqtt_amd/csrc/expand_compact.inc) on the host: tests/hipsim runs a launch with one OS thread
per GPU thread (barriers = pthread barriers, cross-lane reads through a per-wave mirror) and this file compares every result format of
every kernel variant with a numpy expansion of the same pair list — single-run tiles, tiles with many short runs (more than a wave can
hold: the staged fallback), groups of four positions that straddle one, two and three run boundaries, a partial last tile, a window
that does not start at pair 0.  CPU only; the `-m gpu` twins of these cases are tests/test_formats_gpu.py."""
import numpy as np
import pytest

from tests.hipsim import sim

pytestmark = pytest.mark.skipif(sim.clang() is None, reason="hipsim needs clang++")

TILE = 2048


def make_case(rng, lengths, pool=1 << 16, pair_lo=0, tail_pairs=0, first_off=0):
    """A pair list with the given run lengths (the window's pairs), `pair_lo` dummy pairs before and `tail_pairs` after it."""
    subs = np.zeros(pool, dtype=sim.SUB_DTYPE)
    subs["sub_id"] = rng.integers(0, 1 << 24, size=pool, dtype=np.uint32)
    subs["qos_flags"] = rng.integers(0, 3, size=pool, dtype=np.uint32) | (rng.integers(0, 64, size=pool, dtype=np.uint32) << 8)
    lens = np.concatenate([rng.integers(1, 50, size=pair_lo), np.asarray(lengths, dtype=np.int64), rng.integers(1, 50, size=tail_pairs)]).astype(np.int64)
    src = np.array([rng.integers(0, pool - n + 1) for n in lens], dtype=np.uint32)
    off = np.concatenate([[first_off], first_off + np.cumsum(lens)]).astype(np.uint64)
    return subs, src, off, pair_lo, pair_lo + len(lengths)


def reference(fmt, subs, src, off, lo, hi):
    idx = np.concatenate([np.arange(int(src[p]), int(src[p]) + int(off[p + 1] - off[p])) for p in range(lo, hi)])
    s = subs[idx]
    if fmt == sim.FMT_PACKED:
        return s["sub_id"] | ((s["qos_flags"] & 3) << 30), None
    if fmt == sim.FMT_IDS24:
        b = np.zeros((len(s), 3), dtype=np.uint8)
        b[:, 0] = s["sub_id"] & 0xFF
        b[:, 1] = (s["sub_id"] >> 8) & 0xFF
        b[:, 2] = (s["sub_id"] >> 16) & 0xFF
        return b.reshape(-1), None
    q = (s["qos_flags"] & 3) | (((s["qos_flags"] >> 8) & 0x3F) << 2)
    return s["sub_id"].copy(), q.astype(np.uint8)


def config3_like(rng, tiles):
    """Run lengths shaped like BASELINE config 3 (profiles/r04n_config3_tile_np_distribution.txt): a few long runs per topic, a tail of
    short ones, many of length one."""
    out = []
    total = 0
    while total < tiles * TILE:
        topic = [int(rng.choice([20534, 10599, 7125, 6012, 2371]))] if rng.random() < 0.8 else []
        topic += [int(x) for x in rng.choice([1, 1, 1, 2, 3, 5, 40, 300, 863, 2371], size=rng.integers(3, 14))]
        rng.shuffle(topic)
        out += topic
        total += sum(topic)
    return out


CASES = {
    "one_long_run": lambda rng: make_case(rng, [5 * TILE + 77]),
    "aligned_runs": lambda rng: make_case(rng, [TILE, TILE, 4, 4, 8, 2 * TILE - 16], pair_lo=2, tail_pairs=3, first_off=12345),
    "config3_like": lambda rng: make_case(rng, config3_like(rng, 24), pool=1 << 17, pair_lo=5, tail_pairs=2, first_off=(1 << 33) + 5),
    "singletons": lambda rng: make_case(rng, [1] * (TILE + 900) + [3000] + [1, 2, 1, 3] * 40, pair_lo=1),                 # > 64 pairs per tile: staged fallback
    "short_runs": lambda rng: make_case(rng, [int(x) for x in rng.integers(1, 7, size=3000)], tail_pairs=1),              # every group straddles
    "mid_fanout": lambda rng: make_case(rng, [int(x) for x in rng.integers(30, 200, size=200)], pair_lo=7, first_off=999),  # 20-60 pairs per tile
    # 62-64 pairs in a tile, singletons at its end: the lanes that own the last pairs look past lane 63 for the runs after theirs
    "lane_list_full": lambda rng: make_case(rng, [700] + [30] * 20 + [1] * 41 + [707] + [900] + [25] * 40 + [1] * 23 + [125] + [640] + [40] * 31 + [1] * 31 + [137] + [5000],
                                            pair_lo=2, first_off=77),
    "tiny_window": lambda rng: make_case(rng, [3], pair_lo=1, tail_pairs=1),
    "exact_tiles": lambda rng: make_case(rng, [TILE - 1, 1, TILE - 2, 2, 1, TILE - 1, 4 * TILE]),
}
# (variant, format, tiles per block, packed side array): 0 = expand_compact_kernel, 1 = expand_compact_lp_kernel, 2 = expand_ids24_x4_kernel
KERNELS = [(0, sim.FMT_SOA, 1, False), (0, sim.FMT_PACKED, 1, True), (0, sim.FMT_PACKED, 1, False), (0, sim.FMT_IDS24, 4, True),
           (0, sim.FMT_IDS24, 1, False), (0, sim.FMT_PACKED, 4, True),
           (1, sim.FMT_PACKED, 1, True), (1, sim.FMT_PACKED, 2, True), (1, sim.FMT_PACKED, 4, True),
           (1, sim.FMT_IDS24, 1, True), (1, sim.FMT_IDS24, 2, True), (1, sim.FMT_IDS24, 4, True),
           (2, sim.FMT_IDS24, 2, True)]            # variant 2: expand_ids24_x4_kernel (RGR_IDS24_X4)


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("variant,fmt,tiles,use_packed", KERNELS)
def test_expand_compact_source_on_host(case, variant, fmt, tiles, use_packed):
    rng = np.random.default_rng(sum(map(ord, case)))
    subs, src, off, lo, hi = CASES[case](rng)
    ids, qos = sim.expand_compact(variant, fmt, tiles, subs, src, off, lo, hi, use_packed=use_packed)
    want_ids, want_qos = reference(fmt, subs, src, off, lo, hi)
    assert ids.shape == want_ids.shape
    bad = np.flatnonzero(ids != want_ids)
    assert bad.size == 0, f"{bad.size} differing elements, first at {bad[:8]} of {ids.size}"
    if want_qos is not None:
        assert (qos == want_qos).all()


# ---- property: ANY pair list (runs of 1 .. 5 000 entries, 1 .. 150 runs) expands like numpy does, through every lane-held variant
try:
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    HAVE_HYPOTHESIS = True
except ImportError:                                     # pragma: no cover
    HAVE_HYPOTHESIS = False

if HAVE_HYPOTHESIS:
    run_len = st.one_of(st.integers(1, 4), st.integers(1, 70), st.integers(1, 5000), st.sampled_from([2047, 2048, 2049, 4096]))

    @settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
    @given(lens=st.lists(run_len, min_size=1, max_size=150), tiles=st.sampled_from([1, 2, 4]), fmt=st.sampled_from([sim.FMT_PACKED, sim.FMT_IDS24]),
           pair_lo=st.integers(0, 3), first=st.integers(0, 1 << 40), seed=st.integers(0, 1 << 16), x4=st.booleans())
    def test_lane_held_expansion_any_pair_list(lens, tiles, fmt, pair_lo, first, seed, x4):
        total = 0
        kept = []
        for n in lens:                                   # at most ~30 tiles per example: the host run stays in the tens of milliseconds
            if total + n > 30 * TILE:
                break
            kept.append(n); total += n
        rng = np.random.default_rng(seed)
        subs, src, off, lo, hi = make_case(rng, kept, pool=1 << 14, pair_lo=pair_lo, tail_pairs=1, first_off=first)
        if x4:
            fmt, tiles = sim.FMT_IDS24, 2
        ids, _ = sim.expand_compact(2 if x4 else 1, fmt, tiles, subs, src, off, lo, hi)
        want, _ = reference(fmt, subs, src, off, lo, hi)
        assert np.array_equal(ids, want)


@pytest.mark.parametrize("case", ["config3_like", "mid_fanout", "lane_list_full", "short_runs"])
def test_fused_tile_records_of_the_next_window(case):
    """RGR_TILES_FUSED: the lane-held IDS24 expansion of a window and, in the same grid, the tile records of the chunk's next window
    (what tiles_kernel would have written for it)."""
    rng = np.random.default_rng(sum(map(ord, case)) + 1)
    subs, src, off, lo, hi = CASES[case](rng)
    mid = lo + max(1, (hi - lo) * 2 // 3)
    ids, bad_records = sim.expand_ids24_fused(subs, src, off, lo, mid, hi)
    want, _ = reference(sim.FMT_IDS24, subs, src, off, lo, mid)
    assert np.array_equal(ids, want)
    assert bad_records == 0
