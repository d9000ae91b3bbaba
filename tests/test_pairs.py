"""CPU: make_pairs is bit-exact (edge list and order) against known answers produced by the reference
under CPython 3.10 (tests/golden/pairs.json), including SURVEY.md 8(a-1)'s hashes."""
import hashlib
import json
import os

import pytest

from conftest import GOLDEN
from align3r_amd.dust3r.image_pairs import make_pairs, shard_pairs

CASES = json.load(open(os.path.join(GOLDEN, "pairs.json")))["cases"]


def _edges(n, sg, sym, pref):
    pairs = make_pairs([dict(idx=i) for i in range(n)], scene_graph=sg, prefilter=pref, symmetrize=sym)
    return [(a["idx"], b["idx"]) for a, b in pairs]


@pytest.mark.parametrize("c", CASES, ids=[f"{c['n']}-{c['scene_graph']}-{c['symmetrize']}-{c['prefilter']}" for c in CASES])
def test_known_answers(c):
    if c.get("error"):
        with pytest.raises(ValueError):
            _edges(c["n"], c["scene_graph"], c["symmetrize"], c["prefilter"])
        return
    e = _edges(c["n"], c["scene_graph"], c["symmetrize"], c["prefilter"])
    assert len(e) == c["n_edges"]
    assert hashlib.sha256(repr(e).encode()).hexdigest()[:16] == c["sha"]
    if "edges" in c:
        assert e == [tuple(x) for x in c["edges"]]


def test_survey_hashes():
    h = lambda e: hashlib.sha256(repr(e).encode()).hexdigest()[:16]
    assert _edges(2, "complete", True, None) == [(1, 0), (0, 1)]
    e = _edges(16, "swin-3-noncyclic", True, None)
    assert len(e) == 84 and e[:8] == [(3, 4), (4, 6), (12, 13), (5, 7), (0, 2), (8, 9), (9, 11), (2, 5)]
    assert h(e) == "949c38699779d17b"
    assert h(_edges(64, "complete", True, None)) == "bd8d3fb59647e774"
    assert h(_edges(128, "swinstride-5-noncyclic", True, None)) == "818851d29a482ddd"
    assert h(_edges(256, "swin2stride-5-noncyclic", True, None)) == "89eb0d523307be50"


def test_shard_pairs_partition():
    for n in (0, 1, 7, 84, 4032):
        for ws in (1, 2, 3, 8):
            spans = [shard_pairs(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
