"""Holds the oracle to whole streams of the REAL deflate-rs: diffs every file rust/reference-dump wrote
(tests/golden/rust_streams/<fixture>.<level>.<raw|zlib|gzip>) with oracle/deflref.cpp's stream for the same
fixture and level.  Run where a Rust toolchain exists (the build image has none):
    (cd rust/reference-dump && cargo run --release -- ../../tests/golden/ref_inputs ../../tests/golden/rust_streams)
    python tests/golden/check_against_rust.py
Any difference is a bug in the oracle; the GPU path is held to the oracle byte for byte."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob

LEVELS = {"fast": ob.FAST, "default": ob.DEFAULT, "best": ob.BEST, "rle": ob.RLE, "huffman_only": ob.HUFFMAN_ONLY}
BLANK = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff])


def main():
    src = os.path.join(HERE, "ref_inputs")
    dumps = os.path.join(HERE, "rust_streams")
    if not os.path.isdir(dumps):
        raise SystemExit("no tests/golden/rust_streams: run rust/reference-dump first (needs cargo)")
    bad = seen = 0
    for base, _, files in os.walk(src):
        for f in files:
            path = os.path.join(base, f)
            rel = os.path.relpath(path, src)
            data = open(path, "rb").read()
            for lname, lvl in LEVELS.items():
                for ext, w in (("raw", 0), ("zlib", 1), ("gzip", 2)):
                    dump = os.path.join(dumps, "%s.%s.%s" % (rel, lname, ext))
                    if not os.path.exists(dump):
                        continue
                    ref = open(dump, "rb").read()
                    got = ob.encode_gzip(data, BLANK, level=lvl) if w == 2 else ob.encode(data, level=lvl, wrapper=w)
                    seen += 1
                    if got != ref:
                        bad += 1
                        i = next((k for k, (x, y) in enumerate(zip(got, ref)) if x != y), min(len(got), len(ref)))
                        print("DIFF %s %s %s: oracle %d bytes, deflate-rs %d bytes, first difference at %d" % (
                            rel, lname, ext, len(got), len(ref), i))
    print("%d streams compared, %d differ" % (seen, bad))
    sys.exit(1 if bad or not seen else 0)


if __name__ == "__main__":
    main()
