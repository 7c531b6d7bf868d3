"""The oracle against the committed digests (tests/golden/oracle_digests.json, written by
tests/golden/gen_digests.py): 390 streams over every fixture of the reference, every level, raw / zlib /
gzip.  Guards the oracle against drift; the same file is the expectation of the GPU path in
tests/test_gpu_parity.py::test_golden_digests."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen_digests
import oracle_binding as ob

GOLD = json.load(open(os.path.join(HERE, "golden", "oracle_digests.json")))


def test_oracle_reproduces_the_digests():
    seen = 0
    for name, data in gen_digests.cases():
        for lvl, (c, l, m) in gen_digests.LV.items():
            for wname, w in (("raw", 0), ("zlib", 1), ("gzip", 2)):
                key = "%s|%s|%s" % (name, lvl, wname)
                if key not in GOLD["digests"]:
                    continue
                z = (ob.encode_gzip(data, gen_digests.BLANK, opts=ob.make_opts(c, l, m)) if w == 2
                     else ob.encode(data, opts=ob.make_opts(c, l, m, w)))
                assert [len(z), hashlib.sha256(z).hexdigest()] == GOLD["digests"][key], key
                if key in GOLD["streams"]:
                    assert z.hex() == GOLD["streams"][key]
                seen += 1
    assert seen == len(GOLD["digests"]) == 390


# the reference's own byte vector (src/compress.rs:333-345) is among the committed streams
def test_deflate_late_vector_is_in_the_goldens():
    assert GOLD["streams"]["synthetic/deflate_late|best|raw"] == "73494dcb492c4955001100"
    assert GOLD["streams"]["synthetic/empty|default|raw"] == "0300"


# the full-size BASELINE workloads (config 3, 4 and the config-5 input): the digests the GPU tests hold the
# HIP path to are the oracle's, re-derived here
def test_big_digests_are_the_oracles():
    import json
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import gen_big_digests
    gold = json.load(open(os.path.join(HERE, "golden", "big_digests.json")))["digests"]
    seen = 0
    for name, make, level in gen_big_digests.workloads():
        assert gen_big_digests.digest(name, make, level) == gold[name], name
        seen += 1
    assert seen == 3
