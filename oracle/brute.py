"""Independent brute-force pairwise matcher — TEST INFRASTRUCTURE ONLY.

Written from the MQTT matching rules as restated in SURVEY.md App. A.1/A.2 (not from the
trie code), so that a bug shared by the oracle's trie walk and the GPU walk cannot hide:
O(N*M) over small samples only.  Publish topics / retained topic names must not contain
wildcard levels here (that quirk is pinned separately, App. A.4).
"""


def valid(topic: str) -> bool:
    levels = topic.split("/")
    for i, l in enumerate(levels):
        if l in ("+", "#", ""):
            if l == "#" and i != len(levels) - 1:
                return False
            continue
        if "+" in l or "#" in l:
            return False
        if l.startswith("$") and i != 0:
            return False
    return True


def filter_matches(filt: str, topic: str) -> bool:
    """True iff MQTT filter `filt` matches the concrete topic `topic` under rmqtt's rules:
    levels compare byte-exact; '+' matches exactly one level (blank included); '#' matches
    the remaining levels including none ("a/#" matches "a"); a topic whose first level
    starts with '$' is not matched by a filter whose first level is a wildcard."""
    if not valid(filt) or not valid(topic):
        return False
    f, t = filt.split("/"), topic.split("/")
    if t[0].startswith("$") and f[0] in ("+", "#"):
        return False
    for i, fl in enumerate(f):
        if fl == "#":
            return True
        if i >= len(t):
            return False
        if fl == "+":
            continue
        if fl != t[i]:
            return False
    return len(f) == len(t)
