#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.

The reference is Rust and cannot be imported or run here, so these fixtures are NOT reference
outputs; they pin (1) the reference's own known-answer vectors transcribed from its unit tests
(rmqtt/src/trie.rs:443-541, rmqtt/src/retain.rs:608-641, rmqtt/src/topic.rs:460-617) in a
machine-readable form the GPU tests replay, and (2) the seeded workload generator + oracle on a
small instance of every BASELINE.json config, so that a different libm / compiler on the GPU box
cannot silently change the synthetic inputs or the expected outputs.
    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402
from rmqtt_amd import workload as wl  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def reference_vectors():
    return {
        "_source": "transcribed from the reference's unit tests; values are the test's node ids / retained values",
        "trie_tree1": {
            "insert": [["/iot/b/x", 1], ["/iot/b/x", 2], ["/iot/b/y", 3], ["/iot/cc/dd", 4], ["/ddl/22/#", 5], ["/ddl/+/+", 6],
                       ["/ddl/+/1", 7], ["/ddl/#", 8], ["/xyz/yy/zz", 7], ["/xyz", 8]],
            "expect_multiset": {"/iot/b/x": [1, 2], "/iot/b/y": [3], "/iot/cc/dd": [4], "/xyz/yy/zz": [7], "/ddl/22/1/2": [5, 8],
                                "/ddl/22/1": [5, 6, 7, 8], "/ddl/22/": [5, 6, 8], "/ddl/22": [5, 8]},
            "ref": "rmqtt/src/trie.rs:445-469"},
        "trie_tree2_tail": {
            "insert": [["/x/y/z/#", 1], ["/x/y/z/#", 2], ["/x/y/z/", 3], ["/x/y/z/+", 1], ["/x/y/z/+", 2], ["/x/y/z/+", 3]],
            "expect_multiset": {"/x/y/z/2": [1, 2, 1, 2, 3]}, "ref": "rmqtt/src/trie.rs:518-526"},
        "retain": {
            "insert": [["/iot/b/x", 1], ["/iot/b/y", 2], ["/iot/b/z", 3], ["/iot/b", 123], ["/x/y/z", 4], ["/xx/yy", -1], ["/xx/yy/", 0],
                       ["/xx/yy/1", 1], ["/xx/yy/2", 2], ["/xx/yy/3", 3], ["/xx/yy/3/4", 4], ["/xx/yy/3/4/5", 5]],
            "expect_after_first_5": {"/iot/b/y": [2], "/iot/b/+": [1, 2, 3], "/x/y/z": [4]},
            "expect_after_all": {"/xx/yy/+": [0, 1, 2, 3], "/xx/yy/3/+": [4], "/xx/yy/3/4/+": [5], "/xx/yy/1/+": []},
            "ref": "rmqtt/src/retain.rs:609-634"},
        "parse_valid": ["sport/tennis/player1", "sport/tennis/#", "$SYS/tennis/#", "sport/+/player1", "", "/finance", "$SYS", "#", "+", "+/tennis/#"],
        "parse_invalid": ["sport/#/player1", "sport/$SYS/player1", "sport/$SYS", "sport/tennis#", "sport/tennis/#/ranking", "sport+"],
    }


def seeded(cfg, n_sub, n_pub):
    c = wl.CONFIGS[cfg]
    if cfg == 5:
        blob, offs = wl.gen_topics(n_sub, wl.PUB_SEED + cfg, 0.01, c["p_blank"], c["fixed_depth"], distinct=True)
        fb, fo, _, _ = wl.gen_subs(n_pub, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"], force_wildcard=True)
        t = orc.RetainTree()
        rejected = t.insert_bulk(blob, offs)
        st, eo, ev, _ = t.match_batch(fb, fo)
        per = [np.sort(ev[int(a):int(b)]) for a, b in zip(eo[:-1], eo[1:])]
        return {"config": cfg, "n_table": n_sub, "n_query": n_pub, "table_sha256": sha(blob), "query_sha256": sha(fb),
                "first_table": wl.strings(blob, offs, 0, 3), "first_query": wl.strings(fb, fo, 0, 3), "rejected": rejected,
                "hit_offsets_sha256": sha(eo), "sorted_hits_sha256": sha(np.concatenate(per) if per else np.zeros(0, np.int64)),
                "n_hits": int(eo[-1])}
    blob, offs, client, qos = wl.gen_subs(n_sub, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"])
    tb, to = wl.gen_topics(n_pub, wl.PUB_SEED + cfg, 0.01 if cfg != 1 else 0.0, c["p_blank"], c["fixed_depth"])
    o = orc.DefaultRouter()
    o.add_bulk(blob, offs, client, qos)
    exp = o.match_flat(tb, to)
    return {"config": cfg, "n_table": n_sub, "n_query": n_pub, "table_sha256": sha(blob), "client_sha256": sha(client), "qos_sha256": sha(qos),
            "query_sha256": sha(tb), "first_table": wl.strings(blob, offs, 0, 3), "first_query": wl.strings(tb, to, 0, 3),
            "hit_offsets_sha256": sha(exp["hit_offsets"]), "sub_ids_sha256": sha(exp["sub_ids"]), "qos_hits_sha256": sha(exp["qos"]),
            "n_hits": int(exp["hit_offsets"][-1]), "stats": exp["stats"]}


def delivery():
    """Delivery stage: the oracle's forwards() dumps of a fixed world of 400 subscriptions
    (3 nodes, v3/v5 mix, No Local, RAP, subscription identifiers, re-subscribes) x 120 publishes."""
    from tests.test_deliver_parity import golden_dumps, golden_world
    w, pubs = golden_world("emu")
    dumps = golden_dumps(w, pubs, use_backend=False)
    return {"seed": 77, "n_subs": 400, "n_pub": 120, "dumps_sha256": hashlib.sha256("\x00".join(dumps).encode()).hexdigest(),
            "n_rows": sum(d.count("\n") for d in dumps), "first_nonempty": next(d for d in dumps if d)[:200]}


def main():
    json.dump(delivery(), open(os.path.join(HERE, "delivery_small.json"), "w"), indent=1)
    json.dump(reference_vectors(), open(os.path.join(HERE, "reference_vectors.json"), "w"), indent=1)
    out = [seeded(1, 2000, 3000), seeded(2, 20000, 5000), seeded(3, 20000, 3000), seeded(5, 20000, 1500)]
    json.dump(out, open(os.path.join(HERE, "seeded_small.json"), "w"), indent=1)
    print("wrote", HERE)


if __name__ == "__main__":
    main()
