"""Reader for the reference's Raft snapshot of the routing table (SURVEY.md §8(f)-4;
rmqtt-plugins/rmqtt-cluster-raft/src/router.rs:387-580) — rmqtt_amd/host/raft_snapshot.cpp.

CPU tests: the C++ reader against (a) a snapshot assembled by hand, byte by byte, from postcard's wire
format, (b) snapshots written by the restated writer (oracle/raft_snapshot.py) under every compression
and cargo-feature combination, (c) damaged input: every truncation and thousands of byte flips must come
back as an error (or a different decode), never as a crash.  The decompressors are also driven directly
with streams from the real libraries (zlib, liblz4, libzstd) and with hand-made Snappy / LZ4 streams that
use every element kind.

GPU test: ClusterRouter::restore on the C++ Router mirror — the restored table answers `matches` like the
oracle's DefaultRouter holding the same relations."""
import ctypes as C
import random
import re
import struct
import zlib

import pytest

from oracle import raft_snapshot as rs
from rmqtt_amd import build

ALL_FEATURES = rs.FEAT_SHARED | rs.FEAT_LIMIT
COMPRESSIONS = [rs.NONE, rs.ZSTD, rs.LZ4, rs.ZLIB, rs.SNAPPY]


@pytest.fixture(scope="module")
def L():
    build.build_gpu()                       # the host library links against the product library (no GPU needed to load it)
    lib = C.CDLL(build.build_host_router())
    lib.rs_last_error.restype = C.c_char_p
    lib.rs_decode_dump.restype = C.c_void_p
    lib.rs_decode_dump.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_uint32]
    lib.rs_uncompress.restype = C.c_void_p
    lib.rs_uncompress.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    lib.hr_free_str.argtypes = [C.c_void_p]
    return lib


def cpp_dump(L, snap, compression=rs.NONE, features=ALL_FEATURES):
    p = L.rs_decode_dump(bytes(snap), len(snap), compression, features)
    if not p:
        return None
    s = C.string_at(p).decode()
    L.hr_free_str(p)
    return s


def cpp_uncompress(L, data, compression):
    n = C.c_uint64(0)
    p = L.rs_uncompress(bytes(data), len(data), compression, C.byref(n))
    if not p:
        return None
    out = C.string_at(p, n.value)
    L.hr_free_str(p)
    return out


# ---------------------------------------------------------------------------------------------
# a snapshot written out by hand (postcard wire format; nothing of oracle/raft_snapshot.py's writer involved)
# ---------------------------------------------------------------------------------------------
def hand_made_snapshot():
    relations = bytes([
        0x02,                                           # Vec len 2
        #   ("a/+", { "c1": (Id, V3{qos 1, group None, limit None}) })
        0x03, *b"a/+",                                  # TopicFilter
        0x01,                                           # HashMap len 1
        0x02, *b"c1",                                   # key ClientId
        0x80, 0x01,                                     # node_id u64 = 128 (two varint bytes)
        0x00,                                           # lid u16 = 0
        0x01, 0x00, 127, 0, 0, 1, 0xDB, 0x0E,           # local_addr Some(V4(127.0.0.1:1883)); 1883 = 0xDB 0x0E
        0x00,                                           # remote_addr None
        0x02, *b"c1",                                   # client_id
        0x00,                                           # username None
        0x01,                                           # create_time i64 = -1 (zigzag 1)
        0x00,                                           # SubscriptionOptions::V3
        0x01,                                           # qos (one u8)
        0x00,                                           # shared_group None
        0x00,                                           # limit_subs None
        #   ("$share-less/#", { "dev": (Id, V5{qos 2, group Some("g"), limit Some(300), no_local, !rap, rh 2, id Some(65535)}) })
        0x0D, *b"$share-less/#",
        0x01,
        0x03, *b"dev",
        0x07,                                           # node_id 7
        0xFF, 0xFF, 0x03,                               # lid u16 = 65535 (three varint bytes, top byte 3)
        0x00,                                           # local_addr None
        0x01, 0x01, *([0] * 15), 1, 0x80, 0x80, 0x01,   # remote_addr Some(V6([::1]:16384))
        0x03, *b"dev",
        0x01, 0x03, *"üb".encode(),                     # username Some("üb"): the length counts UTF-8 bytes (ü = 2, b = 1)
        0x80, 0x01,                                     # create_time = 64 (zigzag 128)
        0x01,                                           # SubscriptionOptions::V5
        0x02,                                           # qos
        0x01, 0x01, *b"g",                              # shared_group Some("g")
        0x01, 0xAC, 0x02,                               # limit_subs Some(300)
        0x01, 0x00, 0x02,                               # no_local true, retain_as_published false, retain_handling 2
        0x01, 0xFF, 0xFF, 0x03,                         # id Some(NonZeroU32 65535)
    ])
    client_states = bytes([
        0x01,                                           # Vec len 1
        0x02, *b"c1",                                   # key
        0x80, 0x01, 0x00, 0x00, 0x00, 0x02, *b"c1", 0x00, 0x01,     # Id: node 128, lid 0, no addrs, "c1", no username, ct -1
        0x01, 0x00,                                     # online true, handshaking false
        0xA0, 0x9C, 0x01,                               # handshak_duration = 10000 (zigzag 20000 = 0xA0 0x9C 0x01)
    ])
    topics_count = bytes([0x04, 0x06, 0x00])            # Counter(2, 3, StatsMergeMode::None)
    relations_count = bytes([0x04, 0x03, 0x01])         # Counter(2, -2, Sum)
    snap = b"".join(struct.pack("<Q", len(s)) + s for s in (relations, client_states, topics_count, relations_count))
    expected = "\n".join([
        "F\t2",
        "\t".join(["R", b"a/+".hex(), b"c1".hex(), "128", "0", "127.0.0.1:1883", "-", b"c1".hex(), "-", "-1", "3", "1", "-", "-", "0", "0", "0", "-"]),
        "\t".join(["R", b"$share-less/#".hex(), b"dev".hex(), "7", "65535", "-", "[::1]:16384", b"dev".hex(), "üb".encode().hex(), "64", "5", "2",
                   b"g".hex() + ".", "300", "1", "0", "2", "65535"]),
        "\t".join(["C", b"c1".hex(), "128", "0", "-", "-", b"c1".hex(), "-", "-1", "1", "0", "10000"]),
        "T\t2\t3\t0",
        "N\t2\t-2\t1",
    ]) + "\n"
    return snap, expected


def test_hand_made_snapshot(L):
    snap, expected = hand_made_snapshot()
    assert cpp_dump(L, snap) == expected
    # and the checker's own reader agrees with the hand-made bytes
    assert rs.dump(*rs.decode_snapshot(snap)) == expected
    # the same plain sections behind each compression
    secs, pos = [], 0
    for _ in range(4):
        (n,) = struct.unpack_from("<Q", snap, pos)
        secs.append(snap[pos + 8:pos + 8 + n])
        pos += 8 + n
    for c in COMPRESSIONS:
        if not rs.have(c):
            continue
        packed = [rs.compress(c, secs[0]), rs.compress(c, secs[1]), secs[2], secs[3]]
        assert cpp_dump(L, b"".join(struct.pack("<Q", len(s)) + s for s in packed), c) == expected, c


def test_features_change_the_layout(L):
    """shared_group / limit_subs exist only under their cargo features (types.rs:775-778): a snapshot read with
    the wrong feature set must not silently decode to the same thing."""
    snap, expected = hand_made_snapshot()
    assert cpp_dump(L, snap, features=0) != expected


# ---------------------------------------------------------------------------------------------
# random snapshots from the restated writer
# ---------------------------------------------------------------------------------------------
def rand_addr(rng):
    r = rng.random()
    if r < 0.4:
        return None
    if r < 0.7:
        return ("v4", tuple(rng.randrange(256) for _ in range(4)), rng.randrange(65536))
    o = [0] * 16
    for k in range(16):
        if rng.random() < 0.45:
            o[k] = rng.randrange(256)
    if rng.random() < 0.2:
        o[:12] = [0] * 10 + [0xFF, 0xFF]
    return ("v6", tuple(o), rng.randrange(65536))


def rand_id(rng, client):
    return dict(node_id=rng.choice([0, 1, 2, 127, 128, 2**32, 2**64 - 1]), lid=rng.choice([0, 1, 1883, 65535]), local_addr=rand_addr(rng),
                remote_addr=rand_addr(rng), client_id=client, username=rng.choice([None, "", "user", "ユーザー"]),
                create_time=rng.choice([0, 1, -1, 1758400000123, -2**63, 2**63 - 1]))


def rand_opts(rng):
    v5 = rng.random() < 0.5
    return dict(v5=v5, qos=rng.randrange(3), shared_group=rng.choice([None, None, "g1", "", "grp/é"]), limit_subs=rng.choice([None, None, 0, 5, 2**40]),
                no_local=v5 and rng.random() < 0.5, rap=v5 and rng.random() < 0.5, rh=rng.randrange(3) if v5 else 0,
                sub_ident=rng.choice([None, 1, 127, 128, 268435455, 2**32 - 1]) if v5 else None)


def rand_world(rng, n_filters, features=ALL_FEATURES):
    levels = ["a", "b", "sensor", "", "$SYS", "+", "温度"]
    relations = []
    for _ in range(n_filters):
        f = "/".join(rng.choice(levels) for _ in range(rng.randint(1, 5))) + rng.choice(["", "", "/#"])
        rels = []
        for k in range(rng.randint(0, 4)):
            client = f"cl{rng.randint(0, 50)}-{k}"
            rels.append((client, rand_id(rng, client), rand_opts(rng)))
        relations.append((f, rels))
    states = []
    for k in range(rng.randint(0, 8)):
        client = f"cl{k}"
        states.append((client, rand_id(rng, client), rng.random() < 0.5, rng.random() < 0.5, rng.choice([0, 1758400000123, -5])))
    n_rel = sum(len(r) for _, r in relations)
    return relations, states, (n_filters, n_filters + rng.randint(0, 9), rng.randrange(5)), (n_rel, n_rel + 3, 0)


@pytest.mark.parametrize("features", [ALL_FEATURES, rs.FEAT_SHARED, rs.FEAT_LIMIT, 0])
@pytest.mark.parametrize("compression", COMPRESSIONS)
def test_random_snapshots_every_compression_and_feature_set(L, compression, features):
    if not rs.have(compression):
        pytest.skip("library for this compression is not installed")
    rng = random.Random(compression * 16 + features)
    for trial in range(12):
        world = rand_world(rng, rng.choice([0, 1, 7, 60]), features)
        snap = rs.encode_snapshot(*world, compression=compression, features=features)
        want = rs.dump(*world, features=features)
        assert cpp_dump(L, snap, compression, features) == want, (trial, L.rs_last_error())
        assert rs.dump(*rs.decode_snapshot(snap, compression, features), features=features) == want


def test_large_snapshot(L):
    """200 k relations (a few MB per section): multi-chunk Snappy, multi-block zstd, long LZ4 matches."""
    rng = random.Random(5)
    relations = []
    for i in range(50_000):
        f = f"site{i % 97}/dev{i}/+/t{i % 13}"
        rels = [(f"c{i}-{k}", dict(node_id=1 + k, lid=0, client_id=f"c{i}-{k}", create_time=1758400000000 + i), dict(v5=bool(k & 1), qos=k % 3, no_local=False,
                                                                                                             rap=False, rh=0)) for k in range(4)]
        relations.append((f, rels))
    world = (relations, [], (50_000, 50_000, 0), (200_000, 200_000, 0))
    want = rs.dump(*world)
    for c in COMPRESSIONS:
        if rs.have(c):
            snap = rs.encode_snapshot(*world, compression=c)
            assert cpp_dump(L, snap, c) == want, c


# ---------------------------------------------------------------------------------------------
# damaged input
# ---------------------------------------------------------------------------------------------
def test_every_truncation_is_an_error(L):
    snap, expected = hand_made_snapshot()
    for k in range(len(snap)):
        assert cpp_dump(L, snap[:k]) is None, k
        assert L.rs_last_error()
    assert cpp_dump(L, snap + b"trailing bytes are ignored like postcard::from_bytes ignores them") == expected


@pytest.mark.parametrize("compression", COMPRESSIONS)
def test_byte_flips_never_crash(L, compression):
    if not rs.have(compression):
        pytest.skip("library for this compression is not installed")
    rng = random.Random(100 + compression)
    world = rand_world(rng, 25)
    snap = bytearray(rs.encode_snapshot(*world, compression=compression))
    good = cpp_dump(L, snap, compression)
    assert good == rs.dump(*world)
    errors = 0
    for _ in range(3000):
        bad = bytearray(snap)
        for _ in range(rng.choice([1, 1, 2, 5])):
            bad[rng.randrange(len(bad))] = rng.randrange(256)
        if rng.random() < 0.2:
            del bad[rng.randrange(len(bad)):]
        got = cpp_dump(L, bad, compression)
        errors += got is None
    assert errors > 300          # most damage is detected; the rest decodes to something else, which is all a reader can do


def _snap_with_relations(plain):
    cs, cnt = bytes([0]), bytes([0, 0, 0])
    return b"".join(struct.pack("<Q", len(s)) + s for s in (plain, cs, cnt, cnt))


@pytest.mark.parametrize("plain,what", [
    (bytes([1, 1, ord("a"), 1, 1, ord("c"), 0, 0, 0, 0, 1, ord("c"), 0, 0, 0, 3, 0, 0]), "invalid QoS value, 3"),
    (bytes([1, 1, ord("a"), 1, 1, ord("c"), 0, 0, 0, 0, 1, ord("c"), 0, 0, 2, 0, 0, 0]), "SubscriptionOptions variant 2"),
    (bytes([1, 1, ord("a"), 1, 1, ord("c"), 0, 0, 2, 0, 1, ord("c"), 0, 0, 0, 0, 0, 0]), "bad Option tag"),
    (bytes([1, 1, ord("a"), 1, 1, ord("c"), 0, 0, 1, 2, 0, 1, ord("c"), 0, 0, 0, 0, 0, 0]), "SocketAddr variant 2"),
    (bytes([1, 1, 0xFF, 1, 1, ord("c"), 0, 0, 0, 0, 1, ord("c"), 0, 0, 0, 0, 0, 0]), "not UTF-8"),
    (bytes([1, 3, 0xED, 0xA0]), "exceeds the data"),
    (bytes([1, 3, 0xED, 0xA0, 0x80, 0]), "not UTF-8"),                       # a surrogate: core::str::from_utf8 refuses it
    (bytes([1, 2, 0xC0, 0x80, 0]), "not UTF-8"),                              # overlong NUL
    (bytes([1, 1, ord("a"), 1, 1, ord("c"), 0, 0xFF, 0xFF, 0x04, 0, 0, 1, ord("c"), 0, 0, 0, 0, 0, 0]), "varint overflows"),   # lid > u16
    (bytes([1, 1, ord("a"), 1, 1, ord("c")] + [0x80] * 10 + [0]), "varint too long"),
    (bytes([1, 1, ord("a"), 1, 1, ord("c"), 0, 0, 0, 0, 1, ord("c"), 0, 0, 1, 0, 0, 0, 2, 0, 0, 0]), "bad bool"),
    (bytes([1, 1, ord("a"), 1, 1, ord("c"), 0, 0, 0, 0, 1, ord("c"), 0, 0, 1, 0, 0, 0, 0, 0, 3, 0]), "invalid RetainHandling value, 3"),
    (bytes([1, 1, ord("a"), 1, 1, ord("c"), 0, 0, 0, 0, 1, ord("c"), 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0]), "NonZeroU32"),
    (bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x0F]), "exceeds the data"),            # a length no input could hold must not be allocated
])
def test_specific_decode_errors(L, plain, what):
    assert cpp_dump(L, _snap_with_relations(plain)) is None
    assert what in L.rs_last_error().decode(), L.rs_last_error()
    with pytest.raises((ValueError, IndexError)):
        rs.decode_snapshot(_snap_with_relations(plain))


def test_padded_varints_are_accepted(L):
    """postcard takes a non-canonical (padded) varint as long as it fits the type."""
    plain = bytes([0x81, 0x00, 1, ord("a"), 0x80, 0x00])                    # 1 filter "a" with 0 relations, both lengths padded
    assert cpp_dump(L, _snap_with_relations(plain)) == "F\t1\nT\t0\t0\t0\nN\t0\t0\t0\n"


# ---------------------------------------------------------------------------------------------
# decompressors
# ---------------------------------------------------------------------------------------------
def payloads():
    rng = random.Random(9)
    yield b""
    yield b"x"
    yield b"abcd" * 5
    yield bytes(300_000)                                                    # very long matches
    yield bytes(rng.getrandbits(8) for _ in range(200_000))                 # incompressible: long literals
    yield (" ".join(rng.choice(["sensor/+/temp", "c12345", "#", "l3x17"]) for _ in range(40_000))).encode()


@pytest.mark.parametrize("compression", [rs.ZSTD, rs.LZ4, rs.ZLIB, rs.SNAPPY])
def test_decompressors_round_trip(L, compression):
    if not rs.have(compression):
        pytest.skip("library for this compression is not installed")
    for d in payloads():
        assert cpp_uncompress(L, rs.compress(compression, d), compression) == d, len(d)


def test_lz4_details(L):
    d = bytes(range(256)) * 40 + b"tail-literals"
    for block in (rs.lz4_block_compress(d), rs.lz4_block_compress_py(d)):
        assert cpp_uncompress(L, struct.pack("<I", len(d)) + block, rs.LZ4) == d
    # overlapping match (offset 1, the RLE idiom): token 0x1F = 1 literal, match length 15+ -> extension bytes 255, 0
    rle = bytes([0x1F, ord("z"), 1, 0, 255, 0, 0x50, *b"12345"])           # 'z' + 274 more 'z' + "12345"
    want = b"z" * (1 + 4 + 15 + 255) + b"12345"
    assert cpp_uncompress(L, struct.pack("<I", len(want)) + rle, rs.LZ4) == want
    assert cpp_uncompress(L, struct.pack("<I", len(want) + 1) + rle, rs.LZ4) is None            # size prefix disagrees
    assert b"differs" in L.rs_last_error()
    assert cpp_uncompress(L, struct.pack("<I", 10) + bytes([0x04, 0, 0]), rs.LZ4) is None       # offset 0
    assert cpp_uncompress(L, struct.pack("<I", 10) + bytes([0x14, ord("a"), 5, 0]), rs.LZ4) is None   # offset before the start
    assert cpp_uncompress(L, struct.pack("<I", 10) + bytes([0xF0, 255]), rs.LZ4) is None        # literal length runs off the end
    assert cpp_uncompress(L, b"\x01\x00", rs.LZ4) is None                                        # no size prefix
    assert cpp_uncompress(L, struct.pack("<I", 0xFFFFFFFF) + b"\x10a", rs.LZ4) is None             # a prefix no 2-byte block can expand to
    assert b"size prefix" in L.rs_last_error()


def test_snappy_details(L):
    def frame(chunks):
        return b"\xff\x06\x00\x00sNaPpY" + b"".join(bytes([k]) + len(b).to_bytes(3, "little") + b for k, b in chunks)

    def data_chunk(kind, plain, body):
        return (kind, struct.pack("<I", rs._mask(rs.crc32c(plain))) + body)

    # every element kind of the raw format: literal (short, 1 and 2 extra length bytes), copies with 1-, 2- and 4-byte offsets
    lit_a, lit_b = bytes(range(70)), bytes((i * 7) & 0xFF for i in range(300))
    # literals: length 5 in the tag, one extra length byte (70), two extra length bytes (300)
    plain = bytearray(b"abcde" + lit_a + lit_b)
    raw = bytearray(bytes([4 << 2]) + b"abcde" + bytes([60 << 2, 69]) + lit_a + bytes([61 << 2, 299 & 0xFF, 299 >> 8]) + lit_b)

    def copy(off, ln, kind):
        for _ in range(ln):
            plain.append(plain[-off])
        if kind == 1:
            raw.extend([1 | (ln - 4) << 2 | (off >> 8) << 5, off & 0xFF])
        elif kind == 2:
            raw.extend([2 | (ln - 1) << 2, off & 0xFF, off >> 8])
        else:
            raw.extend([3 | (ln - 1) << 2]); raw.extend(off.to_bytes(4, "little"))

    copy(5, 7, 1)            # overlapping, 1-byte offset
    copy(300, 11, 1)         # offset needs the tag's high bits
    copy(370, 64, 2)
    copy(1, 33, 2)           # RLE
    copy(375, 20, 3)
    block = rs.varint(len(plain)) + bytes(raw)
    assert rs.snappy_raw_decompress(block) == bytes(plain)
    ok = frame([data_chunk(0, plain, block), (0xFE, b"\0" * 9), (0x80, b"skip me"), data_chunk(1, b"uncompressed", b"uncompressed"), (0xFF, b"sNaPpY")])
    assert cpp_uncompress(L, ok, rs.SNAPPY) == bytes(plain) + b"uncompressed"
    assert cpp_uncompress(L, ok[10:], rs.SNAPPY) is None and b"stream identifier" in L.rs_last_error()
    bad_crc = frame([(0, struct.pack("<I", 12345) + block)])
    assert cpp_uncompress(L, bad_crc, rs.SNAPPY) is None and b"checksum" in L.rs_last_error()
    assert cpp_uncompress(L, frame([(0x02, b"abcdefgh")]), rs.SNAPPY) is None and b"reserved" in L.rs_last_error()
    assert cpp_uncompress(L, ok[:-3], rs.SNAPPY) is None
    short = rs.varint(len(plain) + 1) + bytes(raw)
    assert cpp_uncompress(L, frame([data_chunk(0, plain, short)]), rs.SNAPPY) is None
    far = rs.varint(4) + bytes([2 | 3 << 2, 9, 0])                                              # copy from before the chunk
    assert cpp_uncompress(L, frame([data_chunk(0, b"xxxx", far)]), rs.SNAPPY) is None


def test_zlib_and_zstd_details(L):
    d = b"relations " * 10_000
    z = zlib.compress(d, 1)
    assert cpp_uncompress(L, z, rs.ZLIB) == d
    assert cpp_uncompress(L, z + b"ignored after the stream end", rs.ZLIB) == d
    assert cpp_uncompress(L, z[:-5], rs.ZLIB) is None and b"truncated" in L.rs_last_error()
    assert cpp_uncompress(L, b"\x78\x01" + bytes(20), rs.ZLIB) is None
    if rs.have(rs.ZSTD):
        one_shot, streamed = rs.zstd_compress(d, streaming=False), rs.zstd_compress(d)
        assert cpp_uncompress(L, one_shot, rs.ZSTD) == d                     # frame header with a content size
        assert cpp_uncompress(L, streamed, rs.ZSTD) == d                     # without (what encode_all writes)
        assert cpp_uncompress(L, streamed + one_shot, rs.ZSTD) == d + d      # decode_all reads every frame
        assert cpp_uncompress(L, streamed[:-4], rs.ZSTD) is None
        assert cpp_uncompress(L, b"not a zstd frame at all", rs.ZSTD) is None
        assert cpp_uncompress(L, b"", rs.ZSTD) == b""


@pytest.mark.parametrize("octets,text", [
    ([0] * 15 + [1], "::1"),
    ([0] * 16, "::"),
    ([0x20, 0x01, 0x0D, 0xB8] + [0] * 11 + [1], "2001:db8::1"),
    ([0] * 10 + [0xFF, 0xFF, 192, 0, 2, 1], "::ffff:192.0.2.1"),
    ([0, 1, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 3], "1:0:0:2::3"),           # the longer zero run wins
    ([0, 1, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 3, 0, 4], "1::2:0:0:3:4"),         # equal runs: the first
    ([0, 1, 0, 0, 0, 2, 0, 3, 0, 4, 0, 5, 0, 6, 0, 7], "1:0:2:3:4:5:6:7"),      # a single zero group is not compressed
    ([0xFE, 0x80] + [0] * 13 + [1], "fe80::1"),
    ([0, 1] + [0] * 14, "1::"),
    ([0xAB, 0xCD, 0, 0x0F] + [0x12, 0x34] * 6, "abcd:f:1234:1234:1234:1234:1234:1234"),
])
def test_ipv6_text_is_rusts_display(L, octets, text):
    assert rs.ipv6_text(octets) == text
    i = dict(node_id=1, lid=0, remote_addr=("v6", tuple(octets), 1883), client_id="c", create_time=0)
    snap = rs.encode_snapshot([("a", [("c", i, dict(v5=False, qos=0))])], [], (1, 1, 0), (1, 1, 0))
    row = cpp_dump(L, snap).split("\n")[1].split("\t")
    assert row[6] == f"[{text}]:1883"


# ---------------------------------------------------------------------------------------------
# GPU: restore on the Router mirror
# ---------------------------------------------------------------------------------------------
class HrId(C.Structure):
    _fields_ = [("node_id", C.c_uint64), ("client_id", C.c_char_p), ("client_len", C.c_uint32), ("create_time", C.c_int64), ("lid", C.c_uint16)]


class HrOpts(C.Structure):
    _fields_ = [("v5", C.c_uint8), ("qos", C.c_uint8), ("no_local", C.c_uint8), ("rap", C.c_uint8), ("rh", C.c_uint8), ("sub_ident", C.c_uint32),
                ("shared_group", C.c_char_p), ("shared_group_len", C.c_uint32)]


def _strip_rel(s):   # oracle v3 rows carry a test-only rel_id column
    return None if s is None else re.sub(r"^(3 [^\t\n]*\t[^\t\n]*\t\d+)\t\d+", r"\1", s, flags=re.M)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_restore_on_the_router_mirror_matches_oracle(L, devices):
    from oracle import oracle as orc
    vp = C.c_void_p
    L.hr_new_sharded.restype = vp; L.hr_new_sharded.argtypes = [C.c_uint64, C.POINTER(C.c_int), C.c_uint32]
    L.hr_free.argtypes = [vp]
    L.hr_set_shared_policy.argtypes = [vp, C.c_int]; L.hr_set_shared_policy.restype = None
    L.hr_flag_mismatches.argtypes = [vp]; L.hr_flag_mismatches.restype = C.c_uint64
    L.hr_add.argtypes = [vp, C.c_char_p, C.c_uint32, C.POINTER(HrId), C.POINTER(HrOpts)]
    L.hr_remove.argtypes = [vp, C.c_char_p, C.c_uint32, C.POINTER(HrId)]
    L.hr_matches.argtypes = [vp, C.POINTER(HrId), C.c_char_p, C.c_uint32]; L.hr_matches.restype = vp
    L.hr_restore_raft.argtypes = [vp, C.c_char_p, C.c_uint64, C.c_int, C.c_uint32]
    for f in ("hr_topics", "hr_routes"):
        getattr(L, f).argtypes = [vp]; getattr(L, f).restype = C.c_int64
    L.hr_topics_tree.argtypes = [vp]; L.hr_topics_tree.restype = C.c_uint64

    devs = (C.c_int * len(devices))(*devices)
    g = L.hr_new_sharded(1, devs, len(devices))
    assert g
    L.hr_set_shared_policy(g, 1)
    rng = random.Random(31 + len(devices))
    levels = ["a", "b", "c", "", "$SYS"]

    def rand_filter():
        lv = [rng.choice(levels if i == 0 else levels[:4] + ["+"]) for i in range(rng.randint(1, 4))]
        return "/".join(lv + (["#"] if rng.random() < 0.25 else []))

    def hid(node, client, ct):
        c = client.encode()
        return HrId(node, c, len(c), ct, 0)

    def matches(o, node, client, ct, t):
        h = hid(node, client, ct)
        p = L.hr_matches(g, C.byref(h), t.encode(), len(t.encode()))
        got = None
        if p:
            got = C.string_at(p).decode()
            L.hr_free_str(p)
        assert got == _strip_rel(o.matches(orc.mk_id(node, client, ct), t)), t

    # something that must be gone after the restore
    for k in range(40):
        f = rand_filter()
        h, ho = hid(1, f"old{k}", 0), HrOpts(0, 1, 0, 0, 0, 0)
        assert L.hr_add(g, f.encode(), len(f.encode()), C.byref(h), C.byref(ho)) == 0

    def world(seed):
        rng2 = random.Random(seed)
        table = {}
        for _ in range(500):
            f = "/".join([rng2.choice(levels if i == 0 else levels[:4] + ["+"]) for i in range(rng2.randint(1, 4))] + (["#"] if rng2.random() < 0.25 else []))
            client = f"cl{rng2.randint(0, 60)}"
            v5 = rng2.random() < 0.5
            i = dict(node_id=rng2.choice([1, 1, 2, 3]), lid=0, client_id=client, create_time=rng2.randint(0, 1))
            o = dict(v5=v5, qos=rng2.randint(0, 2), shared_group=rng2.choice([None, None, None, "g1", "g2"]), limit_subs=None,
                     no_local=v5 and rng2.random() < 0.5, rap=False, rh=0, sub_ident=rng2.choice([None, 1, 2, 3]) if v5 else None)
            table.setdefault(f, {})[client] = (i, o)
        return [(f, [(c, i, o) for c, (i, o) in rels.items()]) for f, rels in table.items()]

    for round_, (seed, compression) in enumerate([(1, rs.NONE), (2, rs.ZLIB), (3, rs.LZ4)]):
        relations = world(seed)
        n_rel = sum(len(r) for _, r in relations)
        snap = rs.encode_snapshot(relations, [], (len(relations), len(relations) + 5, 0), (n_rel, n_rel + 9, 0), compression=compression)
        assert L.hr_restore_raft(g, snap, len(snap), compression, ALL_FEATURES) == 0, L.rs_last_error()
        o = orc.DefaultRouter()
        o.set_shared_policy(1)
        k = 0
        for f, rels in relations:
            for client, i, op in rels:
                assert o.add(f, orc.mk_id(i["node_id"], client, i["create_time"]),
                             orc.mk_opts(qos=op["qos"], v5=op["v5"], no_local=op["no_local"], sub_ident=op["sub_ident"] or 0, shared_group=op["shared_group"]),
                             rel_id=k) == 0
                k += 1
        assert L.hr_topics(g) == len(relations) and L.hr_routes(g) == n_rel and L.hr_topics_tree(g) == o.topics_tree()
        pubs = [(1, "cl1", 0), (2, "cl7", 1), (9, "nobody", 0)]
        for _ in range(250):
            t = "/".join(rng.choice(levels if i == 0 else levels[:4]) for i in range(rng.randint(1, 5)))
            matches(o, *rng.choice(pubs), t)
        # the restored router goes on taking SUBSCRIBE / UNSUBSCRIBE
        for k in range(60):
            f, client, node = rand_filter(), f"new{k}", rng.choice([1, 2])
            h, ho = hid(node, client, 3), HrOpts(0, 2, 0, 0, 0, 0)
            assert L.hr_add(g, f.encode(), len(f.encode()), C.byref(h), C.byref(ho)) == 0
            assert o.add(f, orc.mk_id(node, client, 3), orc.mk_opts(qos=2), rel_id=10_000 + k) == 0
        for f, rels in relations[::3]:
            client, i, _ = rels[0]
            h = hid(i["node_id"], client, i["create_time"])
            assert L.hr_remove(g, f.encode(), len(f.encode()), C.byref(h)) == o.remove(f, orc.mk_id(i["node_id"], client, i["create_time"]))
        for _ in range(150):
            t = "/".join(rng.choice(levels if i == 0 else levels[:4]) for i in range(rng.randint(1, 5)))
            matches(o, *rng.choice(pubs), t)
        assert L.hr_flag_mismatches(g) == 0

    # an invalid filter in the snapshot: Err, and the table that was there keeps answering
    bad = rs.encode_snapshot([("a/#/b", [("c", dict(node_id=1, lid=0, client_id="c", create_time=0), dict(v5=False, qos=0))])], [], (1, 1, 0), (1, 1, 0))
    assert L.hr_restore_raft(g, bad, len(bad), rs.NONE, ALL_FEATURES) == -1 and b"invalid topic filter" in L.rs_last_error()
    for _ in range(50):
        t = "/".join(rng.choice(levels if i == 0 else levels[:4]) for i in range(rng.randint(1, 5)))
        matches(o, 1, "cl1", 0, t)
    L.hr_free(g)


# ---------------------------------------------------------------------------------------------
# property-based: arbitrary strings / integers at the edges of every field
# ---------------------------------------------------------------------------------------------
from hypothesis import HealthCheck, given, settings, strategies as st   # noqa: E402

_text = st.text(max_size=12)
_addr = st.one_of(st.none(),
                  st.tuples(st.just("v4"), st.tuples(*[st.integers(0, 255)] * 4), st.integers(0, 65535)),
                  st.tuples(st.just("v6"), st.tuples(*[st.integers(0, 255)] * 16), st.integers(0, 65535)))
_id = st.fixed_dictionaries(dict(node_id=st.integers(0, 2**64 - 1), lid=st.integers(0, 65535), local_addr=_addr, remote_addr=_addr, client_id=_text,
                                 username=st.one_of(st.none(), _text), create_time=st.integers(-2**63, 2**63 - 1)))
_opts = st.one_of(
    st.fixed_dictionaries(dict(v5=st.just(False), qos=st.integers(0, 2), shared_group=st.one_of(st.none(), _text), limit_subs=st.one_of(st.none(), st.integers(0, 2**64 - 1)),
                               no_local=st.just(False), rap=st.just(False), rh=st.just(0), sub_ident=st.none())),
    st.fixed_dictionaries(dict(v5=st.just(True), qos=st.integers(0, 2), shared_group=st.one_of(st.none(), _text), limit_subs=st.one_of(st.none(), st.integers(0, 2**64 - 1)),
                               no_local=st.booleans(), rap=st.booleans(), rh=st.integers(0, 2), sub_ident=st.one_of(st.none(), st.integers(1, 2**32 - 1)))))
_relations = st.lists(st.tuples(_text, st.lists(st.tuples(_text, _id, _opts), max_size=3)), max_size=5)
_states = st.lists(st.tuples(_text, _id, st.booleans(), st.booleans(), st.integers(-2**63, 2**63 - 1)), max_size=3)
_counter = st.tuples(st.integers(-2**63, 2**63 - 1), st.integers(-2**63, 2**63 - 1), st.integers(0, 4))


@settings(max_examples=400, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(relations=_relations, states=_states, tc=_counter, rc=_counter, features=st.sampled_from([0, 1, 2, 3]), compression=st.sampled_from([rs.NONE, rs.ZLIB, rs.LZ4, rs.SNAPPY]))
def test_reader_property(L, relations, states, tc, rc, features, compression):
    snap = rs.encode_snapshot(relations, states, tc, rc, compression=compression, features=features)
    want = rs.dump(relations, states, tc, rc, features=features)
    assert cpp_dump(L, snap, compression, features) == want
    assert rs.dump(*rs.decode_snapshot(snap, compression, features), features=features) == want
