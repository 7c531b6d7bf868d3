"""Writes tests/golden/long_digest.json: length and SHA-256 of the ORACLE's Default stream of the first 2^31 + 2^20 bytes
of the web-text input (tests/datagen.py::webtext) -- the input the GPU test of the range-walking long encode
(deflate-rs_amd/csrc/deflate_long.inc) crosses 2^31 with.  Not re-derived by the CPU suite: it takes ~4 minutes.
    python tests/golden/gen_long_digest.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datagen
import oracle_binding as ob

N = (1 << 31) + (1 << 20)
if __name__ == "__main__":
    data = datagen.webtext(N)
    z = ob.encode(data, level=ob.DEFAULT)
    out = {"in_len": len(data), "in_sha256": hashlib.sha256(data).hexdigest(), "out_len": len(z), "out_sha256": hashlib.sha256(z).hexdigest()}
    json.dump({"note": "oracle stream of webtext(2^31 + 2^20), Compression::Default; regenerate with gen_long_digest.py", "digest": out},
              open(os.path.join(HERE, "long_digest.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out))
