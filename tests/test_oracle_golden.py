"""Pins the CPU oracle against every known-answer vector the reference's own unit tests
hold for this path:
  rmqtt/src/trie.rs:443-541   TopicTree  (topic_nodeid, topic)
  rmqtt/src/retain.rs:608-641 RetainTree (retain)
  rmqtt/src/topic.rs:460-617  Level / Topic parsing + the pairwise expectations that the
                              trie also satisfies
plus the end-to-end semantic checklist of rmqtt-test (App. B.4) and an independent
brute-force matcher (oracle/brute.py).
"""
import random

import pytest

from oracle import brute
from oracle import oracle as orc


def match_one(tree, topic, vs):
    """Port of the reference test helper (trie.rs:417-441): every yielded value must be in
    `vs`, and the total number of yielded values must equal len(vs)."""
    total = 0
    for _f, matched in tree.matches(topic):
        if any(v not in vs for v in matched):
            return False
        total += len(matched)
    return total == len(vs)


def test_trie_topic_nodeid_first_tree():   # trie.rs:445-477
    t = orc.TopicTree()
    for f, v in [("/iot/b/x", 1), ("/iot/b/x", 2), ("/iot/b/y", 3), ("/iot/cc/dd", 4), ("/ddl/22/#", 5), ("/ddl/+/+", 6),
                 ("/ddl/+/1", 7), ("/ddl/#", 8), ("/xyz/yy/zz", 7), ("/xyz", 8)]:
        t.insert(f, v)
    assert match_one(t, "/iot/b/x", [1, 2])
    assert match_one(t, "/iot/b/y", [3])
    assert match_one(t, "/iot/cc/dd", [4])
    assert not match_one(t, "/iot/cc/dd", [0])
    assert match_one(t, "/xyz/yy/zz", [7])
    assert match_one(t, "/ddl/22/1/2", [5, 8])
    assert match_one(t, "/ddl/22/1", [5, 6, 7, 8])
    assert match_one(t, "/ddl/22/", [5, 6, 8])
    assert match_one(t, "/ddl/22", [5, 8])
    assert t.remove("/iot/b/x", 2) == 1
    assert t.remove("/xyz/yy/zz", 7) == 1
    assert t.remove("/xyz", 123) == 0
    assert not match_one(t, "/xyz/yy/zz", [7])


def test_trie_topic_nodeid_second_tree():   # trie.rs:480-526
    t = orc.TopicTree()
    for f, v in [("/a/b/c", 1), ("/a/+", 2), ("/iot/b/c", 1), ("/iot/b", 2), ("/iot/#", 3), ("/iot/10", 10), ("/iot/11", 11)]:
        t.insert(f, v)
    for v in range(1, 10000):
        t.insert(f"/iot/{v}", v)
    for v in range(1, 10000):
        t.insert("/iot/x", v)
    # values_size survives (the reference round-trips through postcard here, trie.rs:497-502)
    assert t.values_size() == 5 + 9999 + 9999
    assert match_one(t, "/a/b/c", [1])
    assert match_one(t, "/a/b", [2])
    assert match_one(t, "/a/1", [2])
    assert t.is_match("/iot/x") == 1
    items = t.matches("/iot/x")
    assert [f for f, _ in items] == ["/iot/#", "/iot/x"] and len(items[1][1]) == 9999
    t.insert("/x/y/z/#", 1)
    t.insert("/x/y/z/#", 2)
    t.insert("/x/y/z/", 3)
    assert match_one(t, "/x/y/z/", [1, 2, 3])
    for v in (1, 2, 3):
        t.insert("/x/y/z/+", v)
    assert match_one(t, "/x/y/z/2", [1, 2, 1, 2, 3])   # multiset {1,2,1,2,3}, trie.rs:526


def test_trie_unit_values():   # trie.rs:530-541
    t = orc.TopicTree()
    for f in ["/iot/b/x", "/iot/b/x", "/iot/b/y", "/iot/cc/dd", "/ddl/22/#"]:
        t.insert(f, 0)
    assert t.values_size() == 4


def test_trie_iteration_order():   # SURVEY App. A.2, trie.rs:313-375
    t = orc.TopicTree()
    for i, f in enumerate(["a/b", "a/#", "a/+", "+/b", "#", "+/#", "a/b/#", "+/+", "a/+/#"]):
        t.insert(f, i)
    got = [f for f, _ in t.matches("a/b")]
    # root: '#', then '+' subtree (its '#', '+/+' then '+/b' : inside the '+' node: hash item, plus child, exact child),
    # then exact 'a' subtree.
    assert got == ["#", "+/#", "+/+", "+/b", "a/#", "a/+", "a/+/#", "a/b", "a/b/#"]


def test_trie_prune():   # trie.rs:134-149
    t = orc.TopicTree()
    t.insert("a/b/c", 1)
    t.insert("a/b", 2)
    assert t.nodes_size() == 3
    assert t.remove("a/b/c", 1) == 1
    assert t.nodes_size() == 2          # 'c' pruned, 'b' kept (has a value)
    assert t.remove("a/b", 2) == 1
    assert t.nodes_size() == 0          # whole chain pruned
    assert t.remove("a/b", 2) == 0


def test_retain_golden():   # retain.rs:609-641
    def rmatch(tree, f, vs):
        got = tree.matches(f)
        return all(v in vs for _, v in got) and len(got) == len(vs)

    t = orc.RetainTree()
    for s, v in [("/iot/b/x", 1), ("/iot/b/y", 2), ("/iot/b/z", 3), ("/iot/b", 123), ("/x/y/z", 4)]:
        t.insert(s, v)
    assert rmatch(t, "/iot/b/y", [2])
    assert rmatch(t, "/iot/b/+", [1, 2, 3])
    assert rmatch(t, "/x/y/z", [4])
    assert not rmatch(t, "/x/y/z", [1])
    for s, v in [("/xx/yy", -1), ("/xx/yy/", 0), ("/xx/yy/1", 1), ("/xx/yy/2", 2), ("/xx/yy/3", 3), ("/xx/yy/3/4", 4),
                 ("/xx/yy/3/4/5", 5)]:
        t.insert(s, v)
    assert rmatch(t, "/xx/yy/+", [0, 1, 2, 3])
    assert rmatch(t, "/xx/yy/3/+", [4])
    assert rmatch(t, "/xx/yy/3/4/+", [5])
    assert rmatch(t, "/xx/yy/1/+", [])
    n = t.values_size()
    assert t.retain_ge(10**9) == n       # retain(usize::MAX, |_| false)
    assert t.values_size() == 0 and t.nodes_size() == 0


def test_retain_hash_semantics():   # retain.rs:476-481, 502-524; SURVEY App. A.6
    t = orc.RetainTree()
    for i, s in enumerate(["a", "a/b", "a/b/c", "a/x", "b", "$SYS/up", "$SYS", "/lead", "a/"]):
        t.insert(s, i)
    assert [s for s, _ in t.matches("a/#")] == ["a", "a/", "a/b", "a/b/c", "a/x"]       # parent + descendants
    assert [s for s, _ in t.matches("#")] == ["/lead", "a", "a/", "a/b", "a/b/c", "a/x", "b"]   # no '$' topics
    assert [s for s, _ in t.matches("+")] == ["a", "b"]            # blank child "" has no value; '$SYS' skipped
    assert [s for s, _ in t.matches("$SYS/#")] == ["$SYS", "$SYS/up"]
    assert [s for s, _ in t.matches("+/#")] == ["/lead", "a", "a/", "a/b", "a/b/c", "a/x", "b"]
    assert [s for s, _ in t.matches("a/+")] == ["a/", "a/b", "a/x"]
    assert t.remove("a/b") == (1, 1)
    assert [s for s, _ in t.matches("a/+")] == ["a/", "a/x"]
    assert [s for s, _ in t.matches("a/b/c")] == ["a/b/c"]
    assert t.matches("a/#/b") is None


VALID = ["sport/tennis/player1", "sport/tennis/#", "$SYS/tennis/#", "sport/+/player1", "", "/finance", "$SYS", "#", "+",
         "+/tennis/#"]
INVALID = ["sport/#/player1", "sport/$SYS/player1", "sport/$SYS", "sport/tennis#", "sport/tennis/#/ranking", "sport+"]


def test_parse_validity():   # topic.rs:481-571
    for s in VALID:
        assert orc.parse_topic(s) is not None, s
    for s in INVALID:
        assert orc.parse_topic(s) is None, s
    K = dict(normal=0, meta=1, blank=2, plus=3, hash=4)
    assert orc.parse_topic("") == [K["blank"]]
    assert orc.parse_topic("/finance") == [K["blank"], K["normal"]]
    assert orc.parse_topic("$SYS") == [K["meta"]]
    assert orc.parse_topic("a/") == [K["normal"], K["blank"]]
    assert orc.parse_topic("/") == [K["blank"], K["blank"]]
    assert orc.parse_topic("+/tennis/#") == [K["plus"], K["normal"], K["hash"]]


PAIRWISE = [   # (filter, topic, expected) — topic.rs:586-617, the expectations the trie shares
    ("sport/tennis/player1/#", "sport/tennis/player1", True),
    ("sport/tennis/player1/#", "sport/tennis/player1/ranking", True),
    ("sport/tennis/player1/#", "sport/tennis/player1/score/wimbledon", True),
    ("sport/#", "sport", True),
    ("sport/tennis/+", "sport/tennis/player1", True),
    ("sport/tennis/+", "sport/tennis/player2", True),
    ("sport/tennis/+", "sport/tennis/player1/ranking", False),
    ("sport/+", "sport", False),
    ("sport/+", "sport/", True),
    ("+/+", "/finance", True),
    ("/+", "/finance", True),
    ("+", "/finance", False),
    ("#", "$SYS", False),
    ("+/monitor/Clients", "$SYS/monitor/Clients", False),
    ("$SYS/#", "$SYS/", True),
    ("$SYS/monitor/+", "$SYS/monitor/Clients", True),
]


@pytest.mark.parametrize("f,t,exp", PAIRWISE)
def test_pairwise_expectations_on_trie(f, t, exp):
    tree = orc.TopicTree()
    tree.insert(f, 1)
    assert bool(tree.is_match(t)) == exp
    assert brute.filter_matches(f, t) == exp


def test_harness_semantics():   # rmqtt-test functional cases, SURVEY App. B.4
    r = orc.DefaultRouter()
    subs = ["test/wildcard/+/message", "test/wildcard/#", "Case/Topic", "/t", "t", "#", "$SYS/#", "test/overlap/#", "test/overlap/foo"]
    for i, f in enumerate(subs):
        assert r.add(f, orc.mk_id(1, f"c{i}"), orc.mk_opts(qos=1), rel_id=i) == 0
    assert r.add("sport/#/x", orc.mk_id(1, "bad"), orc.mk_opts()) == -1          # wildcard.rs:271-289

    def filters(topic):
        res = r.match_flat(*orc.pack_strings([topic]))
        return sorted(subs[s] for s in res["sub_ids"])

    assert filters("test/wildcard/foo/message") == sorted(["test/wildcard/+/message", "test/wildcard/#", "#"])
    assert filters("test/wildcard/foo/bar/message") == sorted(["test/wildcard/#", "#"])
    assert filters("test/wildcard/a/b/c") == sorted(["test/wildcard/#", "#"])
    assert filters("case/topic") == ["#"]                                         # case-sensitive
    assert filters("Case/Topic") == sorted(["#", "Case/Topic"])
    assert filters("/t") == sorted(["#", "/t"]) and filters("t") == sorted(["#", "t"])   # leading slash matters
    assert filters("$SYS/broker/version") == ["$SYS/#"]                           # dollar_topics.rs:41-95
    assert filters("test/overlap/foo") == sorted(["#", "test/overlap/#", "test/overlap/foo"])   # v3: one copy per filter


def test_router_v5_collector_and_no_local():   # router.rs:196-201, types.rs:513-540
    r = orc.DefaultRouter()
    me = orc.mk_id(1, "pub", create_time=7)
    r.add("a/#", me, orc.mk_opts(qos=1, v5=True, no_local=True, sub_ident=11), rel_id=0)
    r.add("a/b", orc.mk_id(2, "v5c"), orc.mk_opts(qos=2, v5=True, sub_ident=5), rel_id=1)
    r.add("a/+", orc.mk_id(2, "v5c"), orc.mk_opts(qos=0, v5=True, sub_ident=6), rel_id=2)
    r.add("a/+", orc.mk_id(2, "v3c"), orc.mk_opts(qos=1), rel_id=3)
    r.add("a/b", orc.mk_id(2, "v3c"), orc.mk_opts(qos=0), rel_id=4)
    out = r.matches(me, "a/b")
    # No-Local drops the publisher's own v5 subscription; v3 keeps one row per matching filter;
    # v5 keeps the FIRST matched filter (a/+ precedes a/b in iteration order) and collects both ids.
    assert out == "N 2\n3 a/+\tv3c\t1\t3\n3 a/b\tv3c\t0\t4\n5 v5c\ta/+\t0\t0\t5,6\n"
    other = orc.mk_id(1, "pub", create_time=8)      # different create_time => different Id => delivered
    assert r.matches(other, "a/b").startswith("N 1\n5 pub\ta/#\t1\t1\t11\nN 2\n")
    assert r.matches(me, "a/$bad") is None
    assert r.routes() == 5 and r.topics() == 3
    assert r.remove("a/+", orc.mk_id(2, "v3c", create_time=1)) == 1     # Id mismatch -> not removed (router.rs:460-467)
    assert r.remove("a/+", orc.mk_id(2, "v3c")) == 0
    assert r.remove("a/+", orc.mk_id(2, "v5c")) == 0
    assert r.topics() == 2 and r.topics_tree() == 2


def test_wildcard_in_publish_topic_quirk():   # SURVEY App. A.4, trie.rs:358-370
    t = orc.TopicTree()
    t.insert("test/+", 1)
    t.insert("test/#", 2)
    assert [f for f, _ in t.matches("test/+")] == ["test/#", "test/+", "test/+"]     # '+' child reached twice
    assert [f for f, _ in t.matches("test/#")] == ["test/#", "test/+", "test/#"]


def test_brute_force_agrees_with_oracle_random():
    rng = random.Random(1234)
    alpha = ["a", "b", "c", "", "$s"]

    def rand_topic(wild):
        n = rng.randint(1, 5)
        lv = []
        for i in range(n):
            x = rng.random()
            if wild and x < 0.2:
                lv.append("+")
            elif wild and x < 0.3 and i == n - 1:
                lv.append("#")
            else:
                lv.append(rng.choice(alpha if i == 0 else alpha[:4]))
        return "/".join(lv)

    filters = sorted({rand_topic(True) for _ in range(400)})
    tree = orc.TopicTree()
    for i, f in enumerate(filters):
        assert tree.insert(f, i) == 1
    for _ in range(600):
        t = rand_topic(False)
        got = sorted(v for _f, vs in tree.matches(t) for v in vs)
        exp = sorted(i for i, f in enumerate(filters) if brute.filter_matches(f, t))
        assert got == exp, (t, [filters[i] for i in got], [filters[i] for i in exp])
