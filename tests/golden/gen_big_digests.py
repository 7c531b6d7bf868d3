"""Writes tests/golden/big_digests.json: length and SHA-256 of the ORACLE's stream for the BASELINE
workloads at full size -- config 3 (100 000 000 bytes of enwik8-like text, Default),
config 4 (the Silesia-like mix, Compression::Best), config 5 (the first 256 MiB of the web-text input,
Default).  The GPU tests hold the HIP path to these digests; tests/test_oracle_golden.py re-derives them.
    python tests/golden/gen_big_digests.py            (about a minute of CPU)"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datagen
import oracle_binding as ob


def workloads():
    yield "config3_enwik8_like_100MB_default", lambda: datagen.text_like(100_000_000, 0x656E77696B38), ob.DEFAULT
    yield "config4_silesia_like_best", lambda: datagen.silesia_like(0x53494C45), ob.BEST
    yield "config5_webtext_256MiB_default", lambda: datagen.webtext(256 << 20), ob.DEFAULT


def digest(name, make, level):
    data = make()
    z = ob.encode(data, level=level)
    return {"in_len": len(data), "in_sha256": hashlib.sha256(data).hexdigest(), "out_len": len(z),
            "out_sha256": hashlib.sha256(z).hexdigest()}


if __name__ == "__main__":
    out = {name: digest(name, make, level) for name, make, level in workloads()}
    json.dump({"note": "oracle streams of the BASELINE workloads at full size; regenerate with gen_big_digests.py",
               "digests": out}, open(os.path.join(HERE, "big_digests.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))
