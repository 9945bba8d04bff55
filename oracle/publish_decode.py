"""Oracle for the PUBLISH-packet scan — TEST INFRASTRUCTURE, never imported by the product.

Pure-Python restatement of what rmqtt's codec does to reach the topic of a PUBLISH packet:
  framing          rmqtt-codec/src/v3/codec.rs:63-97 (first byte, remaining length, body)
  varint           rmqtt-codec/src/utils.rs:142-155 (at most 4 bytes; a 5th continuation byte = InvalidLength)
  u16 / bytes      rmqtt-codec/src/utils.rs:76-81, 102-108; ByteString = UTF-8 checked (utils.rs:110-114)
  v3 PUBLISH       rmqtt-codec/src/v3/decode.rs:110-128
  v5 PUBLISH       rmqtt-codec/src/v5/packet/publish.rs:31-52, properties :64-101 (ids: packet/mod.rs:160-188;
                   single-valued properties may appear once: utils.rs:58-64)
Pinned on the reference's own decode vectors (v3/decode.rs:279-303), see tests/test_publish_packets.py.
Error classes: 1 not a PUBLISH frame, 2 InvalidLength, 3 MalformedPacket, 4 Utf8Error (= rgr_publish_info.error).
"""
NOT_PUBLISH, LENGTH, MALFORMED, UTF8 = 1, 2, 3, 4


class DecodeError(Exception):
    def __init__(self, code):
        super().__init__(code)
        self.code = code


def _varint(p, pos, end):
    shift = val = 0
    while True:
        if pos >= end:
            raise DecodeError(MALFORMED)
        b = p[pos]; pos += 1
        val += (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        if shift >= 21:
            raise DecodeError(LENGTH)
        shift += 7


def _u16(p, pos, end):
    if end - pos < 2:
        raise DecodeError(LENGTH)
    return (p[pos] << 8) | p[pos + 1], pos + 2


def _bytes(p, pos, end, utf8):
    n, pos = _u16(p, pos, end)
    if end - pos < n:
        raise DecodeError(LENGTH)
    if utf8:
        try:
            bytes(p[pos:pos + n]).decode("utf-8")       # strict, like std::str::from_utf8
        except UnicodeDecodeError:
            raise DecodeError(UTF8)
    return pos + n


def decode_publish(pkt, version=4):
    """-> dict(topic bytes, topic_pos, qos, retain, dup, packet_id, payload_off); raises DecodeError."""
    p = bytes(pkt)
    if len(p) < 2 or (p[0] >> 4) != 3:
        raise DecodeError(NOT_PUBLISH)
    flags = p[0] & 0x0F
    rem, pos = _varint(p, 1, len(p))
    if rem != len(p) - pos:
        raise DecodeError(LENGTH)                      # not exactly one frame
    end = len(p)
    tl, pos = _u16(p, pos, end)
    if end - pos < tl:
        raise DecodeError(LENGTH)
    try:
        p[pos:pos + tl].decode("utf-8")
    except UnicodeDecodeError:
        raise DecodeError(UTF8)
    topic_pos = pos
    pos += tl
    qos = (flags >> 1) & 3
    if qos == 3:
        raise DecodeError(MALFORMED)
    pid = 0
    if qos:
        pid, pos = _u16(p, pos, end)
        if pid == 0:
            raise DecodeError(MALFORMED)
    if version >= 5:
        plen, pos = _varint(p, pos, end)
        if end - pos < plen:
            raise DecodeError(LENGTH)
        pend = pos + plen
        seen = set()

        def once(i):
            if i in seen:
                raise DecodeError(MALFORMED)
            seen.add(i)
        while pos < pend:
            pid_ = p[pos]; pos += 1
            if pid_ == 0x01:
                once(pid_)
                if pend - pos < 1:
                    raise DecodeError(LENGTH)
                if p[pos] > 1:
                    raise DecodeError(MALFORMED)
                pos += 1
            elif pid_ == 0x02:
                once(pid_)
                if pend - pos < 4:
                    raise DecodeError(LENGTH)
                if int.from_bytes(p[pos:pos + 4], "big") == 0:
                    raise DecodeError(MALFORMED)
                pos += 4
            elif pid_ in (0x03, 0x08):
                once(pid_); pos = _bytes(p, pos, pend, True)
            elif pid_ == 0x09:
                once(pid_); pos = _bytes(p, pos, pend, False)
            elif pid_ == 0x0B:
                v, pos = _varint(p, pos, pend)
                if v == 0:
                    raise DecodeError(MALFORMED)
            elif pid_ == 0x23:
                once(pid_)
                v, pos = _u16(p, pos, pend)
                if v == 0:
                    raise DecodeError(MALFORMED)
            elif pid_ == 0x26:
                pos = _bytes(p, pos, pend, True); pos = _bytes(p, pos, pend, True)
            else:
                raise DecodeError(MALFORMED)
    return dict(topic=p[topic_pos:topic_pos + tl], topic_pos=topic_pos, qos=qos, retain=flags & 1, dup=(flags >> 3) & 1,
                packet_id=pid, payload_off=pos)


def encode_varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def encode_publish(topic, payload=b"", qos=0, retain=False, dup=False, packet_id=1, version=4, properties=b""):
    """A well-formed PUBLISH frame (test generator; the MQTT wire format, not reference code)."""
    t = topic if isinstance(topic, bytes) else topic.encode()
    body = len(t).to_bytes(2, "big") + t
    if qos:
        body += packet_id.to_bytes(2, "big")
    if version >= 5:
        body += encode_varint(len(properties)) + properties
    body += payload
    return bytes([0x30 | (8 if dup else 0) | (qos << 1) | (1 if retain else 0)]) + encode_varint(len(body)) + body
