"""Randomised long soak of the property-based parity suites through the host emulator (CPU).
The committed tests run a fixed (derandomised) example set so that CI is reproducible; this script
is how new counter-examples are hunted:  python tools/hyp_soak.py 20000"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypothesis import HealthCheck, given, settings, strategies as st
from tests import test_hypothesis_parity as H
from tests import test_deliver_parity as D
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
S = settings(max_examples=N, deadline=None, suppress_health_check=list(HealthCheck), database=None)
def run(name, fn):
    t = time.time()
    try:
        fn()
        print(name, 'ok', round(time.time() - t, 1), 's', flush=True)
    except Exception as e:
        print(name, 'FAILED', repr(e)[:2000], flush=True)
@S
@given(filters=st.lists(H.WILD, min_size=0, max_size=30), topics=st.lists(H.WILD, min_size=1, max_size=25),
       slot_cap=st.sampled_from([0, 1, 2]), window=st.sampled_from([0, 1, 7]), lds=st.sampled_from([0, 3, 2560]))
def router_wild(filters, topics, slot_cap, window, lds):
    H._router_property(filters, topics, slot_cap, window, lds)
@S
@given(filters=st.lists(H.TOPIC, min_size=0, max_size=25), topics=st.lists(H.TOPIC, min_size=1, max_size=25),
       slot_cap=st.sampled_from([0, 1, 2]), window=st.sampled_from([0, 1, 7]), lds=st.sampled_from([0, 3, 2560]))
def router_plain(filters, topics, slot_cap, window, lds):
    H._router_property(filters, topics, slot_cap, window, lds)
@S
@given(topics=st.lists(H.WILD_TOPIC, min_size=0, max_size=30, unique=True), filters=st.lists(H.WILD_TOPIC, min_size=1, max_size=20),
       removes=st.lists(st.integers(0, 29), max_size=6))
def retain_wild(topics, filters, removes):
    H._retain_property(topics, filters, removes)
@S
@given(topics=st.lists(H.TOPIC, min_size=0, max_size=25, unique=True), filters=st.lists(H.TOPIC, min_size=1, max_size=20),
       removes=st.lists(st.integers(0, 24), max_size=6))
def retain_plain(topics, filters, removes):
    H._retain_property(topics, filters, removes)
@S
@given(ops=st.lists(H._OP, min_size=1, max_size=80), topics=st.lists(H.WILD, min_size=1, max_size=20), slot_cap=st.sampled_from([0, 1, 2]))
def router_churn(ops, topics, slot_cap):
    H._router_churn(ops, topics, slot_cap)
run('router_churn', router_churn)
run('retain_wild', retain_wild)
run('retain_plain', retain_plain)
run('router_wild', router_wild)
run('router_plain', router_plain)
inner = D.test_delivery_stage_property.hypothesis.inner_test
@S
@given(subs=st.lists(D._SUB, min_size=0, max_size=30), pubs=st.lists(D._PUB, min_size=1, max_size=12),
       window=st.sampled_from([0, 1, 5]), slot_cap=st.sampled_from([0, 1, 2]))
def deliver(subs, pubs, window, slot_cap):
    inner(subs, pubs, window, slot_cap)
run('deliver', deliver)
