"""The lane-held compact expansion (kernel SOURCE on the host, tests/hipsim) over the pair lists a real pipeline builds: the emulator
(tests/emu: the product's table compilers + the kernels' per-item functions) dumps the dense (source, offset) pair arrays of a chunk
(EMU_DUMP_PAIRS) and answers the same batch itself; the kernels must expand those pairs into exactly the emulator's hits — the
subscription table with config-3-shaped filters and hot '#' filters, and the retained path (runs = ranges of the preorder value
array: the shape whose 62..64-pair tiles exposed the guarded cross-lane read on the device, profiles/r04p_*).  CPU only."""
import os

import numpy as np
import pytest

from rmqtt_amd import workload as wl
from tests.emu import emu
from tests.hipsim import sim

pytestmark = pytest.mark.skipif(sim.clang() is None, reason="hipsim needs clang++")


def read_dump(path):
    raw = open(path, "rb").read()
    pos, chunks = 0, []
    while pos < len(raw):
        P = int(np.frombuffer(raw, dtype=np.uint64, count=1, offset=pos)[0]); pos += 8
        src = np.frombuffer(raw, dtype=np.uint32, count=P, offset=pos).copy(); pos += 4 * P
        off = np.frombuffer(raw, dtype=np.uint64, count=P + 1, offset=pos).copy(); pos += 8 * (P + 1)
        chunks.append((src, off))
    return chunks


def expand_with_kernels(src, off, vals_n, window_pairs):
    """The chunk's pairs window by window through every lane-held variant; the 'entries' are position numbers, so the expansion of a
    pair list is the list of source positions — comparable across kernels without knowing the table."""
    subs = np.zeros(vals_n, dtype=sim.SUB_DTYPE)
    subs["sub_id"] = np.arange(vals_n, dtype=np.uint32) & 0xFFFFFF
    P = len(src)
    want = np.concatenate([np.arange(int(s), int(s) + int(n)) for s, n in zip(src, np.diff(off.astype(np.int64)))]) & 0xFFFFFF if P else np.zeros(0, dtype=np.int64)
    for variant, fmt, tiles in ((1, sim.FMT_IDS24, 1), (1, sim.FMT_PACKED, 2), (1, sim.FMT_IDS24, 4), (2, sim.FMT_IDS24, 2)):
        got = []
        for lo in range(0, P, window_pairs):
            hi = min(P, lo + window_pairs)
            ids, _ = sim.expand_compact(variant, fmt, tiles, subs, src, off, lo, hi)
            if fmt == sim.FMT_IDS24:
                b = ids.reshape(-1, 3).astype(np.int64)
                ids = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            got.append(ids.astype(np.int64) & 0xFFFFFF)
        got = np.concatenate(got) if got else np.zeros(0, dtype=np.int64)
        assert np.array_equal(got, want), (variant, fmt, tiles)
    return want


def test_router_pairs_of_a_real_pipeline(tmp_path, monkeypatch):
    cfg = 3
    c = wl.CONFIGS[cfg]
    n_sub = 40_000
    blob, offs, client, qos = wl.gen_subs(n_sub, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"])
    # hot filters with thousands of subscribers, as config 3 has them at full size
    extra = ["l0x0/#"] * 5000 + ["+/#"] * 2371 + ["l0x1/#"] * 2500 + ["l0x0/l1x0/#"] * 2100
    dump = str(tmp_path / "pairs.bin")
    monkeypatch.setenv("EMU_DUMP_PAIRS", dump)
    e = emu.EmuRouter(slot_cap=64)
    assert e.subscribe_bulk(blob, offs, None, qos) == 0
    sid = n_sub
    for f in sorted(set(extra)):
        fid = e.filter_add(f)
        for _ in range(extra.count(f)):
            e.sub_add(fid, sid, sid % 3); sid += 1
    tb, to = wl.gen_topics(60, wl.PUB_SEED + cfg, 0.01, c["p_blank"])
    res = e.match_batch(tb, to)
    chunks = read_dump(dump)
    assert len(chunks) == 1
    src, off = chunks[0]
    assert int(off[-1]) == len(res["tuples"]) > 150_000
    want = expand_with_kernels(src, off, int((src.astype(np.int64) + np.diff(off.astype(np.int64))).max()), window_pairs=700)
    assert len(want) == len(res["tuples"])


def test_retained_pairs_of_a_real_pipeline(tmp_path, monkeypatch):
    cfg = 5
    c = wl.CONFIGS[cfg]
    blob, offs = wl.gen_topics(30_000, wl.PUB_SEED + cfg, 0.01, c["p_blank"], c["fixed_depth"], distinct=True)
    tb, to, _, _ = wl.gen_subs(400, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"], force_wildcard=True)
    dump = str(tmp_path / "pairs.bin")
    monkeypatch.setenv("EMU_DUMP_PAIRS", dump)
    e = emu.EmuRouter()
    e.retain_add_bulk(blob, offs)
    res = e.retain_match_batch(tb, to)
    chunks = read_dump(dump)
    assert len(chunks) == 1
    src, off = chunks[0]
    assert int(off[-1]) == len(res["topic_ids"]) > 30_000
    expand_with_kernels(src, off, int((src.astype(np.int64) + np.diff(off.astype(np.int64))).max()), window_pairs=900)
