"""count_batched_kernel / compact_batched_kernel (rmqtt_amd/csrc/prep_batched.inc, RGR_PREP_BATCH) on the host (tests/hipsim) against
count_topic / compact_topic — the per-topic functions the product's kernels and the emulator share.  Lists shorter and longer than a
batch, empty lists, filters without subscribers (dropped pairs), lists in the overflow arena, lists for the *_big kernels (left alone),
chunks that are not a multiple of the wave, stages that fill up mid-list.  CPU only."""
import numpy as np
import pytest

from tests.hipsim import sim

pytestmark = pytest.mark.skipif(sim.clang() is None, reason="hipsim needs clang++")


def world(rng, n, max_len, n_filt=5000, p_empty=0.2, slot_cap=16, big=0):
    filt = np.zeros(n_filt, dtype=sim.DESC_DTYPE)
    filt["begin"] = rng.integers(0, 1 << 24, size=n_filt)
    filt["count"] = np.where(rng.random(n_filt) < p_empty, 0, rng.integers(1, 3000, size=n_filt))
    lists = [rng.integers(0, n_filt, size=int(rng.integers(0, max_len + 1))).tolist() for _ in range(n)]
    for t in rng.choice(n, size=big, replace=False):
        lists[int(t)] = rng.integers(0, n_filt, size=int(rng.integers(65, 200))).tolist()
    return lists, filt, slot_cap


CASES = {
    "config3_like": lambda rng: world(rng, 1000, 34, slot_cap=64),
    "short_lists": lambda rng: world(rng, 777, 3, slot_cap=4),
    "overflow_arena": lambda rng: world(rng, 300, 40, slot_cap=8),
    "with_big_topics": lambda rng: world(rng, 500, 30, slot_cap=32, big=7),
    "all_empty_filters": lambda rng: world(rng, 130, 10, p_empty=1.0),
    "stage_fills_up": lambda rng: world(rng, 128, 64, p_empty=0.0, slot_cap=64),          # 64 topics x up to 64 pairs > the 512-pair stage
    "one_topic": lambda rng: world(rng, 1, 9),
}


@pytest.mark.parametrize("with_pub", [False, True])
@pytest.mark.parametrize("case", sorted(CASES))
def test_batched_count_and_compact_on_host(case, with_pub):
    rng = np.random.default_rng(sum(map(ord, case)))
    lists, filt, slot_cap = CASES[case](rng)
    pub = None
    if with_pub:
        pub = np.zeros(100 + len(lists), dtype=sim.PUB_DTYPE)
        pub["qos_retain"] = rng.integers(0, 8, size=len(pub))
    diff, pairs, hits = sim.prep(lists, filt, slot_cap, topic_base=100, pub=pub)
    assert diff == 0
    if case not in ("all_empty_filters",):
        assert pairs > 0 and hits > 0
