"""PUBLISH-packet topic extraction (second half): the codec's decode of a framed PUBLISH
(rmqtt-codec/src/v3/decode.rs:110-128, v5/packet/publish.rs:31-101) restated in oracle/publish_decode.py, pinned
on the reference's own decode vectors, against the per-packet scan the device runs (executed on the host through
the emulator here; through rgr_batch_create_from_publish on the GPU)."""
import random

import numpy as np
import pytest

from oracle import publish_decode as pd
from tests.emu import emu


def pack(pkts):
    offs = np.zeros(len(pkts) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(p) for p in pkts])
    return np.frombuffer(b"".join(pkts) + b"\0", dtype=np.uint8)[:int(offs[-1])].copy() if offs[-1] else np.zeros(1, dtype=np.uint8), offs


def test_oracle_on_the_references_own_vectors():
    # rmqtt-codec/src/v3/decode.rs:279-303 (test_decode_publish_packets)
    d = pd.decode_publish(b"\x3d\x0D\x00\x05topic\x43\x21data", 4)
    assert (d["topic"], d["qos"], d["retain"], d["dup"], d["packet_id"]) == (b"topic", 2, 1, 1, 0x4321)
    assert b"\x3d\x0D\x00\x05topic\x43\x21data"[d["payload_off"]:] == b"data"
    d = pd.decode_publish(b"\x30\x0b\x00\x05topicdata", 4)
    assert (d["topic"], d["qos"], d["retain"], d["dup"], d["packet_id"]) == (b"topic", 0, 0, 0, 0)
    assert b"\x30\x0b\x00\x05topicdata"[d["payload_off"]:] == b"data"
    # the other control packets of the same test are not PUBLISH frames
    for other in (b"\x40\x02\x43\x21", b"\x50\x02\x43\x21", b"\x62\x02\x43\x21", b"\x82\x12\x12\x34\x00\x04test\x01\x00\x06filter\x02", b"\xe0\x00"):
        with pytest.raises(pd.DecodeError) as e:
            pd.decode_publish(other, 4)
        assert e.value.code == pd.NOT_PUBLISH
    # varint (utils.rs:142-155): 4 bytes is the maximum
    assert pd._varint(bytes([0xFF, 0xFF, 0xFF, 0x7F]), 0, 4) == (268_435_455, 4)
    with pytest.raises(pd.DecodeError) as e:
        pd._varint(bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x01]), 0, 5)
    assert e.value.code == pd.LENGTH
    # round trip of the generator
    for v in (4, 5):
        pkt = pd.encode_publish("a/b/c", b"xyz", qos=1, retain=True, packet_id=77, version=v, properties=b"\x01\x01")
        d = pd.decode_publish(pkt, v)
        assert (d["topic"], d["qos"], d["retain"], d["packet_id"], pkt[d["payload_off"]:]) == (b"a/b/c", 1, 1, 77, b"xyz")


def _props(rng):
    out = b""
    for _ in range(rng.randint(0, 4)):
        k = rng.choice([0x01, 0x02, 0x03, 0x08, 0x09, 0x0B, 0x23, 0x26, 0x26, 0x0B, 0x11])     # 0x11 is not a PUBLISH property
        if k == 0x01:
            out += bytes([k, rng.choice([0, 1, 1, 2])])
        elif k == 0x02:
            out += bytes([k]) + rng.choice([0, 5, 70000]).to_bytes(4, "big")
        elif k in (0x03, 0x08):
            s = rng.choice([b"text", b"", b"\xc3\x28", "é".encode()])
            out += bytes([k]) + len(s).to_bytes(2, "big") + s
        elif k == 0x09:
            s = bytes(rng.randrange(256) for _ in range(rng.randint(0, 5)))
            out += bytes([k]) + len(s).to_bytes(2, "big") + s
        elif k == 0x0B:
            out += bytes([k]) + pd.encode_varint(rng.choice([0, 1, 200, 300000]))
        elif k == 0x23:
            out += bytes([k]) + rng.choice([0, 7]).to_bytes(2, "big")
        elif k == 0x26:
            out += bytes([k]) + b"\x00\x01k\x00\x02vv"
        else:
            out += bytes([k, 0, 0, 0, 1])
    return out


def _random_packets(seed, n, version):
    rng = random.Random(seed)
    levels = ["a", "bb", "", "$SYS", "+", "#", "café", "中文", "x" * 40]
    pkts = []
    for _ in range(n):
        topic = "/".join(rng.choice(levels) for _ in range(rng.randint(1, 5))).encode()
        r = rng.random()
        if r < 0.1:
            topic = rng.choice([b"\xff\xfe", b"a/\xc0\xaf", b"\xed\xa0\x80/x", b"ok/\xf4\x90\x80\x80", b"\xe2\x82"])   # invalid UTF-8 (overlong, surrogate, > U+10FFFF, truncated)
        qos = rng.choice([0, 0, 1, 2])
        pkt = bytearray(pd.encode_publish(topic, bytes(rng.randrange(256) for _ in range(rng.randint(0, 300))), qos=qos, retain=rng.random() < 0.3,
                                          dup=rng.random() < 0.2, packet_id=rng.choice([0, 1, 65535]) if rng.random() < 0.1 else rng.randint(1, 65535),
                                          version=version, properties=_props(rng) if version >= 5 else b""))
        r = rng.random()
        if r < 0.08 and len(pkt) > 3:
            del pkt[rng.randrange(2, len(pkt)):]                       # truncated frame
        elif r < 0.14:
            pkt[0] = (pkt[0] & 0xF9) | 0x06                            # qos 3
        elif r < 0.18:
            pkt[0] = rng.choice([0x10, 0x40, 0x82, 0xE0]) | (pkt[0] & 0x0F)
        elif r < 0.22:
            pkt += b"zz"                                               # trailing bytes: not one frame
        elif r < 0.26 and len(pkt) > 6:
            pkt[rng.randrange(2, min(len(pkt), 12))] ^= 1 << rng.randrange(8)
        pkts.append(bytes(pkt))
    return pkts


@pytest.mark.parametrize("version", [4, 5])
def test_device_scan_equals_oracle_on_random_and_mutated_packets(version):
    pkts = _random_packets(100 + version, 6000, version) + [b"", b"\x30", b"\x30\x00", b"\x30\x02\x00\x00", b"\x30\x80\x80\x80\x80\x01\x00\x00"]
    blob, offs = pack(pkts)
    got = emu.publish_scan(blob, offs, version)
    n_ok = 0
    kinds = set()
    for i, p in enumerate(pkts):
        try:
            d = pd.decode_publish(p, version)
        except pd.DecodeError as e:
            assert got["error"][i] == e.code, (i, p[:20], got["error"][i], e.code)
            kinds.add(e.code)
            continue
        n_ok += 1
        g = got[i]
        assert g["error"] == 0, (i, p[:20])
        assert int(g["topic_off"]) == int(offs[i]) + d["topic_pos"] and g["topic_len"] == len(d["topic"])
        assert (g["qos"], g["retain"], g["dup"], g["packet_id"], g["payload_off"]) == (d["qos"], d["retain"], d["dup"], d["packet_id"], d["payload_off"])
    assert n_ok > 2000 and kinds == {pd.NOT_PUBLISH, pd.LENGTH, pd.MALFORMED, pd.UTF8}


@pytest.mark.gpu
@pytest.mark.parametrize("version", [4, 5])
def test_publish_packet_batch_on_the_gpu_equals_topic_batch(version):
    """rgr_batch_create_from_publish: statuses, extracted fields and the match result equal the plain batch built
    from the topics the oracle decodes; malformed packets never match; the packets' own qos / retain drive the
    delivery stage when publisher ids are given."""
    from rmqtt_amd import capi
    from rmqtt_amd import workload as wl
    c = wl.CONFIGS[3]
    blob, offs, client, qos = wl.gen_subs(30_000, wl.SUB_SEED + 3, c["p_plus"], c["p_hash"], c["p_sys"])
    r = capi.Router(device=0, window_hits=20_000)
    assert r.subscribe_bulk(blob, offs, None, qos) == 0
    r.commit()
    rng = random.Random(5)
    tb, to = wl.gen_topics(4_000, wl.PUB_SEED + 3, 0.01, c["p_blank"])
    topics = wl.strings(tb, to)
    pkts = []
    for t in topics:
        pkts.append(pd.encode_publish(t, b"p" * rng.randint(0, 50), qos=rng.choice([0, 1, 2]), retain=rng.random() < 0.5, packet_id=rng.randint(1, 65535),
                                      version=version, properties=b"\x23\x00\x07" if version >= 5 and rng.random() < 0.3 else b""))
    pkts += _random_packets(9, 800, version)
    pblob, poffs = pack(pkts)
    b = r.publish_batch(pblob, poffs, version=version)
    info, status = b.publish_info(), b.status()
    exp_topics, exp_bad = [], []
    for i, p in enumerate(pkts):
        try:
            d = pd.decode_publish(p, version)
            exp_topics.append(d["topic"]); exp_bad.append(False)
            assert info["error"][i] == 0 and (info["qos"][i], info["retain"][i], info["packet_id"][i]) == (d["qos"], d["retain"], d["packet_id"])
        except pd.DecodeError as e:
            exp_topics.append(b""); exp_bad.append(True)
            assert info["error"][i] == e.code and status[i] == capi.RGR_PACKET_MALFORMED
    ref_blob, ref_offs = capi.pack(exp_topics)
    ref = r.match_batch(ref_blob, ref_offs)
    hits, nwin = b.run()
    ho = ref["hit_offsets"].astype(np.int64)
    exp_hits = sum(int(ho[i + 1] - ho[i]) for i in range(len(pkts)) if not exp_bad[i])
    assert hits == exp_hits
    for i in range(len(pkts)):
        if not exp_bad[i]:
            assert (status[i] < 0) == (ref["status"][i] < 0)
    # window by window: same tuples as the reference batch for the well-formed packets, nothing for the others
    b.begin()
    got = []
    while True:
        w = b.next_window()
        if w is None:
            break
        t, o = b.window_to_host(w)
        got.append(t)
    got = np.concatenate(got)
    keep = np.repeat(~np.array(exp_bad), np.diff(ho))
    assert np.array_equal(got["sub_id"], ref["tuples"]["sub_id"][keep]) and np.array_equal(got["topic_idx"], ref["tuples"]["topic_idx"][keep])
    b.close()
    # with publisher ids the batch carries publish attributes taken from the packets: delivery qos = min(publish, subscription)
    b2 = r.publish_batch(pblob, poffs, version=version, from_ids=np.full(len(pkts), capi.ID_NONE, dtype=np.uint32))
    b2.begin()
    w = b2.next_window()
    t, o = b2.window_to_host(w)
    pq = info["qos"][t["topic_idx"]]
    sq = qos[t["sub_id"]]
    assert np.array_equal(t["qos_flags"] & 3, np.minimum(pq, sq))
    b2.close()
    # framing that goes backwards is refused on the host (the device would compute a negative packet length from it)
    bad_offs = poffs.copy()
    bad_offs[7], bad_offs[8] = bad_offs[8], bad_offs[7] - 1
    with pytest.raises(capi.RgrError) as e:
        r.publish_batch(pblob, bad_offs, version=version)
    assert e.value.code == capi.RGR_EINVAL
    r.close()
