"""TEST INFRASTRUCTURE (oracle/): restatement of the reference's Raft snapshot of the routing table,
writer AND reader side, used as the test-vector generator and checker for the product's reader
(rmqtt_amd/host/raft_snapshot.cpp).  Never imported by the product.

What it follows:
  rmqtt-plugins/rmqtt-cluster-raft/src/router.rs:387-463   ClusterRouter::snapshot
  rmqtt-plugins/rmqtt-cluster-raft/src/router.rs:466-580   ClusterRouter::restore
  rmqtt-plugins/rmqtt-cluster-raft/src/router.rs:39-45     ClientStatus
  rmqtt/src/types.rs:1899-1911 (_Id), :607-610 (SubscriptionOptions), :769-779 (SubOptionsV3),
  :803-821 (SubOptionsV5), :717-731 / :864-890 (qos and retain_handling as one u8)
  rmqtt-utils/src/counter.rs:39, :337-343                   Counter, StatsMergeMode

Third-party formats restated here (none of them is in /root/reference; versions from Cargo.lock):
  postcard 1.1.3 wire format — varint(LEB128) for u16..u64/usize, zigzag for signed, one byte for
    u8/bool/Option tag, varint length + bytes for str, varint length + items for seq/map, fields in order
    for tuples/structs, varint(u32) variant index for enums;
  serde's impls for std types in non-human-readable formats — SocketAddr = variant index (0 V4, 1 V6) +
    (ip octets as a fixed tuple of u8, port u16); NonZeroU32 = u32; AtomicIsize = isize;
  lz4_flex block + u32 LE size prefix, the Snappy framing format (snap), zlib (flate2), zstd frames.

PARITY UNPINNED for this format: no snapshot written by the reference itself is available (no Rust
toolchain here, and the reference's tests hold no snapshot bytes).  Reader and writer restatements were
written separately from the format descriptions above; liblz4 / libzstd / zlib produce the real
compressed streams where they are installed.
"""
import ctypes as C
import ctypes.util
import struct
import zlib

NONE, ZSTD, LZ4, ZLIB, SNAPPY = 0, 1, 2, 3, 4          # config.rs:415-420 (+ None)
FEAT_SHARED, FEAT_LIMIT = 1, 2


# ---------------------------------------------------------------------------------------------
# postcard writer
# ---------------------------------------------------------------------------------------------
def varint(v):
    assert v >= 0
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def zigzag(v):
    return varint((v << 1) ^ (v >> 63) if v >= 0 else ((-v) << 1) - 1)


def pstr(s):
    b = s if isinstance(s, bytes) else s.encode()
    return varint(len(b)) + b


def option(v, enc):
    return b"\x00" if v is None else b"\x01" + enc(v)


def socket_addr(a):
    """a = ("v4", (a,b,c,d), port) | ("v6", 16 octets, port)"""
    kind, octets, port = a
    return varint(0 if kind == "v4" else 1) + bytes(octets) + varint(port)


def enc_id(i):
    """i: dict node_id, lid, local_addr, remote_addr, client_id, username, create_time (types.rs:1899-1911)"""
    return (varint(i["node_id"]) + varint(i["lid"]) + option(i.get("local_addr"), socket_addr) + option(i.get("remote_addr"), socket_addr) +
            pstr(i["client_id"]) + option(i.get("username"), pstr) + zigzag(i["create_time"]))


def enc_opts(o, features):
    """o: dict v5, qos, shared_group, limit_subs, no_local, rap, rh, sub_ident"""
    out = varint(1 if o["v5"] else 0) + bytes([o["qos"]])
    if features & FEAT_SHARED:
        out += option(o.get("shared_group"), pstr)
    if features & FEAT_LIMIT:
        out += option(o.get("limit_subs"), varint)
    if o["v5"]:
        out += bytes([int(o["no_local"]), int(o["rap"]), o["rh"]]) + option(o.get("sub_ident"), varint)
    return out


def enc_counter(c):
    return zigzag(c[0]) + zigzag(c[1]) + varint(c[2])


def enc_relations(relations, features):
    """relations: list of (filter, [(client_key, id, opts), ...]) — router.rs:428-432"""
    out = bytearray(varint(len(relations)))
    for f, rels in relations:
        out += pstr(f) + varint(len(rels))
        for key, i, o in rels:
            out += pstr(key) + enc_id(i) + enc_opts(o, features)
    return bytes(out)


def enc_client_states(states):
    """states: list of (client_key, id, online, handshaking, handshak_duration) — router.rs:434-438"""
    out = bytearray(varint(len(states)))
    for key, i, online, hs, dur in states:
        out += pstr(key) + enc_id(i) + bytes([int(online), int(hs)]) + zigzag(dur)
    return bytes(out)


def encode_snapshot(relations, client_states, topics_count, relations_count, compression=NONE, features=FEAT_SHARED | FEAT_LIMIT):
    """router.rs:440-450: four sections, each behind its length as 8 LE bytes; the counters are never compressed."""
    secs = [compress(compression, enc_relations(relations, features)), compress(compression, enc_client_states(client_states)),
            enc_counter(topics_count), enc_counter(relations_count)]
    return b"".join(struct.pack("<Q", len(s)) + s for s in secs)


# ---------------------------------------------------------------------------------------------
# postcard reader (the checker's own; the product's is C++)
# ---------------------------------------------------------------------------------------------
class Rd:
    def __init__(self, b):
        self.b, self.i = b, 0

    def byte(self):
        v = self.b[self.i]
        self.i += 1
        return v

    def varint(self, bits=64):
        v, max_bytes = 0, (bits + 6) // 7
        for k in range(max_bytes):
            b = self.byte()
            v |= (b & 0x7F) << (7 * k)
            if not b & 0x80:
                if v >> bits:
                    raise ValueError("varint overflows its type")
                return v
        raise ValueError("varint too long")

    def zigzag(self):
        v = self.varint()
        return (v >> 1) ^ -(v & 1)

    def flag(self, what):
        v = self.byte()
        if v > 1:
            raise ValueError(f"bad {what}")
        return bool(v)

    def str(self):
        n = self.varint()
        if n > len(self.b) - self.i:
            raise ValueError("length exceeds the data")
        s = bytes(self.b[self.i:self.i + n])
        self.i += n
        s.decode("utf-8")            # strict, like core::str::from_utf8
        return s

    def option(self, dec):
        return dec() if self.flag("Option tag") else None

    def socket_addr(self):
        v = self.varint(32)
        if v > 1:
            raise ValueError("SocketAddr variant")
        n = 4 if v == 0 else 16
        if n > len(self.b) - self.i:
            raise IndexError
        o = tuple(self.b[self.i:self.i + n])
        self.i += n
        return ("v4" if v == 0 else "v6", o, self.varint(16))

    def id(self):
        return dict(node_id=self.varint(), lid=self.varint(16), local_addr=self.option(self.socket_addr), remote_addr=self.option(self.socket_addr),
                    client_id=self.str(), username=self.option(self.str), create_time=self.zigzag())

    def opts(self, features):
        v = self.varint(32)
        if v > 1:
            raise ValueError("SubscriptionOptions variant")
        o = dict(v5=bool(v), qos=self.byte(), shared_group=None, limit_subs=None, no_local=False, rap=False, rh=0, sub_ident=None)
        if o["qos"] > 2:
            raise ValueError("invalid QoS value")
        if features & FEAT_SHARED:
            o["shared_group"] = self.option(self.str)
        if features & FEAT_LIMIT:
            o["limit_subs"] = self.option(self.varint)
        if o["v5"]:
            o["no_local"], o["rap"], o["rh"] = self.flag("bool"), self.flag("bool"), self.byte()
            if o["rh"] > 2:
                raise ValueError("invalid RetainHandling value")
            o["sub_ident"] = self.option(lambda: self.varint(32))
            if o["sub_ident"] == 0:
                raise ValueError("NonZeroU32 is 0")
        return o

    def counter(self):
        c = (self.zigzag(), self.zigzag(), self.varint(32))
        if c[2] > 4:
            raise ValueError("StatsMergeMode variant")
        return c


def decode_snapshot(snap, compression=NONE, features=FEAT_SHARED | FEAT_LIMIT):
    """-> (relations, client_states, topics_count, relations_count) in the shapes encode_snapshot takes (strings as bytes)."""
    secs, pos = [], 0
    for _ in range(4):
        (n,) = struct.unpack_from("<Q", snap, pos)
        pos += 8
        if n > len(snap) - pos:
            raise ValueError("length prefix runs past the snapshot")
        secs.append(snap[pos:pos + n])
        pos += n
    r = Rd(uncompress(compression, secs[0]))
    relations = []
    for _ in range(r.varint()):
        f = r.str()
        relations.append((f, [(r.str(), r.id(), r.opts(features)) for _ in range(r.varint())]))
    r = Rd(uncompress(compression, secs[1]))
    states = [(r.str(), r.id(), r.flag("bool"), r.flag("bool"), r.zigzag()) for _ in range(r.varint())]
    return relations, states, Rd(secs[2]).counter(), Rd(secs[3]).counter()


# ---------------------------------------------------------------------------------------------
# canonical text (the same rows rs_decode_dump of rmqtt_amd/host/router_capi.cpp prints)
# ---------------------------------------------------------------------------------------------
def _hex(b):
    if b is None:
        return "-"
    b = b if isinstance(b, bytes) else b.encode()
    return b.hex() or "-"


def ipv6_text(o):
    """std::net::Ipv6Addr's Display"""
    g = [o[2 * i] << 8 | o[2 * i + 1] for i in range(8)]
    if g[:5] == [0] * 5 and g[5] == 0xFFFF:
        return "::ffff:%d.%d.%d.%d" % tuple(o[12:16])
    best, best_len, i = -1, 0, 0
    while i < 8:
        if g[i]:
            i += 1
            continue
        j = i
        while j < 8 and not g[j]:
            j += 1
        if j - i > best_len:
            best, best_len = i, j - i
        i = j
    if best_len < 2:
        return ":".join("%x" % x for x in g)
    return ":".join("%x" % x for x in g[:best]) + "::" + ":".join("%x" % x for x in g[best + best_len:])


def addr_text(a):
    if a is None:
        return "-"
    kind, o, port = a
    return ("%d.%d.%d.%d:%d" % (*o, port)) if kind == "v4" else "[%s]:%d" % (ipv6_text(o), port)


def _id_text(i):
    return "\t".join([str(i["node_id"]), str(i["lid"]), addr_text(i.get("local_addr")), addr_text(i.get("remote_addr")), _hex(i["client_id"]),
                      _hex(i.get("username")), str(i["create_time"])])


def dump(relations, states, topics_count, relations_count, features=FEAT_SHARED | FEAT_LIMIT):
    rows = [f"F\t{len(relations)}"]
    for f, rels in relations:
        for key, i, o in rels:
            g = o.get("shared_group") if features & FEAT_SHARED else None
            lim = o.get("limit_subs") if features & FEAT_LIMIT else None
            v5 = o["v5"]
            rows.append("\t".join(["R", _hex(f), _hex(key), _id_text(i), "5" if v5 else "3", str(o["qos"]), "-" if g is None else _hex(g) + ".",
                                   "-" if lim is None else str(lim), str(int(v5 and o["no_local"])), str(int(v5 and o["rap"])), str(o["rh"] if v5 else 0),
                                   str(o["sub_ident"]) if v5 and o.get("sub_ident") else "-"]))
    for key, i, online, hs, dur in states:
        rows.append("\t".join(["C", _hex(key), _id_text(i), str(int(online)), str(int(hs)), str(dur)]))
    rows.append("T\t%d\t%d\t%d" % tuple(topics_count))
    rows.append("N\t%d\t%d\t%d" % tuple(relations_count))
    return "\n".join(rows) + "\n"


# ---------------------------------------------------------------------------------------------
# compression (router.rs:392-413 / :470-493)
# ---------------------------------------------------------------------------------------------
def _lib(name, soname):
    for cand in (soname, ctypes.util.find_library(name)):
        if cand:
            try:
                return C.CDLL(cand)
            except OSError:
                pass
    return None


_lz4 = _lib("lz4", "liblz4.so.1")
_zstd = _lib("zstd", "libzstd.so.1")


def have(compression):
    return {NONE: True, ZLIB: True, SNAPPY: True, LZ4: True, ZSTD: _zstd is not None}[compression]


def lz4_block_compress(data):
    """One LZ4 block (no size prefix).  liblz4 when installed, else a greedy hash matcher."""
    if _lz4 is not None and data:
        _lz4.LZ4_compressBound.restype = C.c_int
        cap = _lz4.LZ4_compressBound(C.c_int(len(data)))
        buf = C.create_string_buffer(cap)
        _lz4.LZ4_compress_default.restype = C.c_int
        n = _lz4.LZ4_compress_default(C.c_char_p(bytes(data)), buf, C.c_int(len(data)), C.c_int(cap))
        assert n > 0
        return buf.raw[:n]
    return lz4_block_compress_py(data)


def lz4_block_compress_py(data):
    data = bytes(data)
    n, out, anchor, i, table = len(data), bytearray(), 0, 0, {}

    def emit(lit, mlen, off):
        tok_l = min(len(lit), 15)
        tok_m = 0 if mlen is None else min(mlen - 4, 15)
        out.append(tok_l << 4 | tok_m)
        if tok_l == 15:
            r = len(lit) - 15
            while r >= 255:
                out.append(255)
                r -= 255
            out.append(r)
        out.extend(lit)
        if mlen is not None:
            out.extend(struct.pack("<H", off))
            if tok_m == 15:
                r = mlen - 19
                while r >= 255:
                    out.append(255)
                    r -= 255
                out.append(r)

    while i + 12 < n:                               # the format's end-of-block margins
        key = data[i:i + 4]
        j = table.get(key)
        table[key] = i
        if j is not None and i - j <= 0xFFFF:
            m = 4
            while i + m < n - 5 and data[j + m] == data[i + m]:
                m += 1
            emit(data[anchor:i], m, i - j)
            i += m
            anchor = i
        else:
            i += 1
    emit(data[anchor:], None, 0)
    return bytes(out)


def lz4_block_decompress(block, want):
    out, i, n = bytearray(), 0, len(block)
    while i < n:
        tok = block[i]
        i += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = block[i]
                i += 1
                lit += b
                if b != 255:
                    break
        out += block[i:i + lit]
        i += lit
        if i >= n:
            break
        off = block[i] | block[i + 1] << 8
        i += 2
        m = tok & 15
        if m == 15:
            while True:
                b = block[i]
                i += 1
                m += b
                if b != 255:
                    break
        m += 4
        if off == 0 or off > len(out):
            raise ValueError("lz4: bad offset")
        for _ in range(m):
            out.append(out[-off])
    if len(out) != want:
        raise ValueError("lz4: size differs from the prefix")
    return bytes(out)


def crc32c(data):
    t = crc32c.table
    c = 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


crc32c.table = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    crc32c.table.append(_c)


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def snappy_raw_compress(data):
    """Raw Snappy block: varint length, then literals and 2-byte-offset copies (greedy hash matcher)."""
    data = bytes(data)
    n, out, anchor, i, table = len(data), bytearray(varint(len(data))), 0, 0, {}

    def literal(lit):
        if not lit:
            return
        k = len(lit) - 1
        if k < 60:
            out.append(k << 2)
        else:
            nb = (k.bit_length() + 7) // 8
            out.append((59 + nb) << 2)
            out.extend(k.to_bytes(nb, "little"))
        out.extend(lit)

    while i + 4 <= n:
        key = data[i:i + 4]
        j = table.get(key)
        table[key] = i
        if j is not None and i - j <= 0xFFFF:
            m = 4
            while i + m < n and data[j + m] == data[i + m]:
                m += 1
            literal(data[anchor:i])
            off, left = i - j, m
            while left:
                k = min(left, 64)                    # 2-byte-offset copies carry 1..64 bytes, 1-byte-offset ones 4..11
                if 4 <= k <= 11 and off < 2048:
                    out.append(1 | (k - 4) << 2 | (off >> 8) << 5)
                    out.append(off & 0xFF)
                else:
                    out.append(2 | (k - 1) << 2)
                    out.extend(struct.pack("<H", off))
                left -= k
            i += m
            anchor = i
        else:
            i += 1
    literal(data[anchor:])
    return bytes(out)


def snappy_raw_decompress(block):
    i, want, shift = 0, 0, 0
    while True:
        b = block[i]
        i += 1
        want |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            break
    out, n = bytearray(), len(block)
    while i < n:
        tag = block[i]
        i += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(block[i:i + nb], "little")
                i += nb
            ln += 1
            out += block[i:i + ln]
            i += ln
            continue
        if kind == 1:
            ln, off = 4 + ((tag >> 2) & 7), (tag >> 5) << 8 | block[i]
            i += 1
        elif kind == 2:
            ln, off = (tag >> 2) + 1, block[i] | block[i + 1] << 8
            i += 2
        else:
            ln, off = (tag >> 2) + 1, int.from_bytes(block[i:i + 4], "little")
            i += 4
        if off == 0 or off > len(out):
            raise ValueError("snappy: bad offset")
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != want:
        raise ValueError("snappy: length differs from the header")
    return bytes(out)


def snappy_frame_compress(data, mix=True):
    """snap::write::FrameEncoder: stream identifier, then one chunk per 65536 plain bytes — compressed (type 0)
    when that is smaller, else uncompressed (type 1)."""
    out = bytearray(b"\xff\x06\x00\x00sNaPpY")
    for k in range(0, len(data), 65536):
        plain = data[k:k + 65536]
        comp = snappy_raw_compress(plain)
        kind, body = (0, comp) if (len(comp) < len(plain) - len(plain) // 8 or not mix) else (1, plain)
        body = struct.pack("<I", _mask(crc32c(plain))) + body
        out += bytes([kind]) + len(body).to_bytes(3, "little") + body
    return bytes(out)


def snappy_frame_decompress(data):
    out, i, seen = bytearray(), 0, False
    while i < len(data):
        kind, ln = data[i], int.from_bytes(data[i + 1:i + 4], "little")
        body = data[i + 4:i + 4 + ln]
        if len(body) != ln:
            raise ValueError("snappy: truncated chunk")
        i += 4 + ln
        if kind == 0xFF:
            if body != b"sNaPpY":
                raise ValueError("snappy: bad stream identifier")
            seen = True
            continue
        if not seen:
            raise ValueError("snappy: no stream identifier")
        if kind >= 0x80:
            continue
        if kind > 1:
            raise ValueError("snappy: reserved chunk type")
        plain = snappy_raw_decompress(body[4:]) if kind == 0 else bytes(body[4:])
        if _mask(crc32c(plain)) != struct.unpack("<I", body[:4])[0]:
            raise ValueError("snappy: checksum mismatch")
        out += plain
    return bytes(out)


def zstd_compress(data, streaming=True):
    """zstd::encode_all(data, 1): a streaming encoder, so the frame header carries no content size."""
    assert _zstd is not None
    z = _zstd
    if not streaming:
        z.ZSTD_compressBound.restype = C.c_size_t
        z.ZSTD_compressBound.argtypes = [C.c_size_t]
        cap = z.ZSTD_compressBound(len(data))
        buf = C.create_string_buffer(cap)
        z.ZSTD_compress.restype = C.c_size_t
        z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
        n = z.ZSTD_compress(buf, cap, bytes(data), len(data), 1)
        assert not z.ZSTD_isError(C.c_size_t(n))
        return buf.raw[:n]

    class Buf(C.Structure):
        _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    z.ZSTD_createCStream.restype = C.c_void_p
    z.ZSTD_initCStream.argtypes = [C.c_void_p, C.c_int]
    z.ZSTD_initCStream.restype = C.c_size_t
    for f in ("ZSTD_compressStream", "ZSTD_endStream"):
        getattr(z, f).restype = C.c_size_t
    z.ZSTD_compressStream.argtypes = [C.c_void_p, C.POINTER(Buf), C.POINTER(Buf)]
    z.ZSTD_endStream.argtypes = [C.c_void_p, C.POINTER(Buf)]
    z.ZSTD_freeCStream.argtypes = [C.c_void_p]
    z.ZSTD_isError.argtypes = [C.c_size_t]
    cs = z.ZSTD_createCStream()
    assert not z.ZSTD_isError(z.ZSTD_initCStream(cs, 1))
    src = C.create_string_buffer(bytes(data), len(data))
    dst = C.create_string_buffer(1 << 17)
    inb = Buf(C.cast(src, C.c_void_p), len(data), 0)
    out = bytearray()
    while inb.pos < inb.size:
        ob = Buf(C.cast(dst, C.c_void_p), len(dst), 0)
        assert not z.ZSTD_isError(z.ZSTD_compressStream(cs, C.byref(ob), C.byref(inb)))
        out += dst.raw[:ob.pos]
    while True:
        ob = Buf(C.cast(dst, C.c_void_p), len(dst), 0)
        left = z.ZSTD_endStream(cs, C.byref(ob))
        assert not z.ZSTD_isError(left)
        out += dst.raw[:ob.pos]
        if left == 0:
            break
    z.ZSTD_freeCStream(cs)
    return bytes(out)


def zstd_decompress(data):
    assert _zstd is not None
    z = _zstd

    class Buf(C.Structure):
        _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    z.ZSTD_createDStream.restype = C.c_void_p
    z.ZSTD_decompressStream.restype = C.c_size_t
    z.ZSTD_decompressStream.argtypes = [C.c_void_p, C.POINTER(Buf), C.POINTER(Buf)]
    z.ZSTD_freeDStream.argtypes = [C.c_void_p]
    z.ZSTD_isError.argtypes = [C.c_size_t]
    ds = z.ZSTD_createDStream()
    src = C.create_string_buffer(bytes(data), len(data))
    dst = C.create_string_buffer(1 << 17)
    inb, out, hint = Buf(C.cast(src, C.c_void_p), len(data), 0), bytearray(), 0 if not data else 1
    while inb.pos < inb.size or hint:
        ob = Buf(C.cast(dst, C.c_void_p), len(dst), 0)
        before = inb.pos
        hint = z.ZSTD_decompressStream(ds, C.byref(ob), C.byref(inb))
        if z.ZSTD_isError(hint):
            z.ZSTD_freeDStream(ds)
            raise ValueError("zstd: corrupt frame")
        out += dst.raw[:ob.pos]
        if inb.pos == before and ob.pos == 0:
            break
    z.ZSTD_freeDStream(ds)
    if hint:
        raise ValueError("zstd: truncated frame")
    return bytes(out)


def compress(compression, data):
    if compression == NONE:
        return bytes(data)
    if compression == ZLIB:
        return zlib.compress(bytes(data), 1)                      # flate2::Compression::fast()
    if compression == LZ4:
        return struct.pack("<I", len(data)) + lz4_block_compress(data)     # compress_prepend_size
    if compression == SNAPPY:
        return snappy_frame_compress(data)
    if compression == ZSTD:
        return zstd_compress(data)
    raise ValueError(compression)


def uncompress(compression, data):
    if compression == NONE:
        return bytes(data)
    if compression == ZLIB:
        return zlib.decompressobj().decompress(bytes(data))
    if compression == LZ4:
        return lz4_block_decompress(data[4:], struct.unpack("<I", data[:4])[0])
    if compression == SNAPPY:
        return snappy_frame_decompress(data)
    if compression == ZSTD:
        return zstd_decompress(data)
    raise ValueError(compression)
