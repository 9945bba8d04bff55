"""Maximum sizes the wire format allows (an MQTT UTF-8 string is at most 65 535 bytes), through the product's host
code and the kernels' per-lane functions on the CPU emulator, against the oracle: one 65 535-byte level, 32 768
levels in one topic, blank levels only, 16-level-deep '+' chains that fork at every level (the 2^L case of
trie.rs:301-409), and the same shapes on the retained path.  Every case runs on both backends: `emu` in the CPU
suite, `hip` (through the C ABI) under `-m gpu`."""
import threading

import numpy as np
import pytest

from oracle import oracle as orc
from tests.parity import Pair, make_backend, pack

BACKENDS = ["emu", pytest.param("hip", marks=pytest.mark.gpu)]

MAXLEN = 65535


def on_big_stack(fn):
    """The ORACLE recurses once per topic level (the reference does too: Node::_insert / _remove, trie.rs:113-149, would
    need more than a tokio worker's 2 MiB stack for such a filter), so its calls run on a thread with a 1 GiB stack.
    The product's walk is iterative and does not care."""
    def deco(kind):
        box = {}

        def run():
            try:
                fn(kind)
            except BaseException as e:      # noqa: BLE001 - re-raised on the test's thread
                box["e"] = e
        old = threading.stack_size(1 << 30)
        try:
            th = threading.Thread(target=run)
            th.start()
            th.join()
        finally:
            threading.stack_size(old)
        if "e" in box:
            raise box["e"]
    deco.__name__ = fn.__name__
    deco.__doc__ = fn.__doc__
    return pytest.mark.parametrize("kind", BACKENDS)(deco)


@on_big_stack
def test_router_maximum_topic_and_filter_sizes(kind):
    p = Pair(kind)
    one_level = "x" * MAXLEN                                      # a single level of the maximum length
    many_levels = "/".join(["a"] * 32768)                         # 32 768 levels, 65 535 bytes
    blanks = "/" * (MAXLEN)                                       # 65 536 blank levels
    long_prefix = "/".join(["a"] * 32767)                         # parent of many_levels
    filters = [one_level, many_levels, blanks, long_prefix + "/#", long_prefix + "/+", "#", "+", "/".join(["+"] * 32768), "a/#",
               "/".join(["a"] * 32766) + "/+/a", "/".join(["a"] * 32760) + "/#"]
    for i, f in enumerate(filters):
        assert p.add(f, f"c{i}", i, qos=i % 3), f[:20]
    p.commit()
    topics = [one_level, many_levels, blanks, long_prefix, long_prefix + "/b", "a", "x" * (MAXLEN - 1), "/" * (MAXLEN - 1), "/".join(["a"] * 32768)[:-1] + "b"]
    exp, got = p.check(*pack(topics))
    hits = np.diff(got["hit_offsets"]).tolist()
    assert hits[0] == 3            # '#', '+', the level itself
    assert hits[1] >= 6            # '#', a/#, the filter itself, prefix/#, prefix/+, the all-'+' filter, .../+/a, the deep '#'
    assert hits[2] == 2            # '#' and the all-blank filter
    # remove the deep filters again: pruning walks 32 768 levels bottom-up (trie.rs:129-149)
    for i, f in enumerate(filters):
        if len(f) > 1000:
            p.remove(f, f"c{i}", i)
    p.commit()
    exp, got = p.check(*pack(topics))
    assert np.diff(got["hit_offsets"]).tolist()[1] == 2           # '#' and 'a/#' are what is left for the deep topic


@pytest.mark.parametrize("kind", BACKENDS)
def test_router_every_level_forks(kind):
    """'+' and the literal at every level: the walk visits 2^L nodes (L = 14: 16 384 matched filters for one topic)."""
    p = Pair(kind)
    L = 14
    n = 0
    for mask in range(1 << L):
        f = "/".join("+" if mask >> k & 1 else f"l{k}" for k in range(L))
        assert p.add(f, "c", n, qos=n % 3)
        n += 1
    p.commit()
    topic = "/".join(f"l{k}" for k in range(L))
    exp, got = p.check(*pack([topic, topic + "/x", "/".join(f"l{k}" for k in range(L - 1)), "$" + topic]))
    assert int(got["hit_offsets"][1]) == 1 << L
    assert np.diff(got["hit_offsets"]).tolist()[1:] == [0, 0, 0]


@on_big_stack
def test_retain_maximum_sizes(kind):
    b = make_backend(kind)
    t = orc.RetainTree()
    topics = ["x" * MAXLEN, "/".join(["a"] * 32768), "/" * MAXLEN, "a", "/".join(["a"] * 32767), "$SYS/" + "y" * (MAXLEN - 5)]
    for i, tp in enumerate(topics):
        b.retain_add(tp, i)
        t.insert(tp, i)
    b.retain_commit()
    filters = ["#", "+", "a/#", "/".join(["+"] * 32768), "/".join(["a"] * 32767) + "/+", "/".join(["a"] * 32767) + "/#", "x" * MAXLEN, "/" * MAXLEN,
               "/".join(["+"] * 32767) + "/#", "$SYS/#", "$SYS/+", "/".join(["a"] * 16000) + "/#"]
    got = b.retain_match_batch(*pack(filters))
    blob, offs = pack(filters)
    st, eo, ev, _ = t.match_batch(blob, offs)
    assert np.array_equal(got["status"] < 0, st < 0)
    ho = got["hit_offsets"]
    for k in range(len(filters)):
        assert sorted(got["topic_ids"][int(ho[k]):int(ho[k + 1])].tolist()) == sorted(ev[int(eo[k]):int(eo[k + 1])].tolist()), filters[k][:30]
    assert int(ho[1] - ho[0]) == 5                                 # '#': everything outside '$'
    for tp in topics:                                              # the deep chains are unlinked here, on the big stack
        t.remove(tp)


@pytest.mark.parametrize("kind", BACKENDS)
def test_product_host_code_does_not_recurse_per_level(kind):
    """The same deep filters through the product's table compiler and walk on a 256 KiB stack (a tokio worker has 2 MiB):
    insert, commit, match, remove + prune, retained add / match / remove — no per-level recursion anywhere."""
    out = {}

    def body():
        b = make_backend(kind)
        deep, blanks = "/".join(["a"] * 32768), "/" * MAXLEN
        fids = []
        for i, f in enumerate([deep, blanks, "/".join(["+"] * 32768), "/".join(["a"] * 32767) + "/#"]):
            fid = b.filter_add(f)
            b.sub_add(fid, i, 0, 0)
            fids.append(fid)
        b.commit()
        out["hits"] = np.diff(b.match_batch(*pack([deep, blanks]))["hit_offsets"]).tolist()
        for i, fid in enumerate(fids):
            b.sub_remove(fid, i)
            b.filter_remove(fid)
        b.commit()
        out["after"] = np.diff(b.match_batch(*pack([deep, blanks]))["hit_offsets"]).tolist()
        b.retain_add(deep, 1)
        b.retain_add(blanks, 2)
        b.retain_commit()
        out["retain"] = np.diff(b.retain_match_batch(*pack(["#", "/".join(["+"] * 32768), deep]))["hit_offsets"]).tolist()
        b.retain_remove(deep)
        b.retain_remove(blanks)
        b.retain_commit()

    old = threading.stack_size(256 * 1024)
    try:
        th = threading.Thread(target=body)
        th.start()
        th.join()
    finally:
        threading.stack_size(old)
    assert out == {"hits": [3, 1], "after": [0, 0], "retain": [2, 1, 1]}
