"""Writes tests/golden/oracle_digests.json: length + sha256 of the ORACLE's output (oracle/deflref.cpp,
pinned by the reference's known-answer tests) for every fixture of the reference at every level and
wrapper, plus the full streams of a few tiny inputs.  The real crate cannot be built here (no Rust), so
these are regression anchors for oracle and GPU path alike, not vectors of the reference itself.
Run from the repo root:  python tests/golden/gen_digests.py"""
import glob, hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob

LV = {"fast": (1, 0, 0), "default": (128, 32, 1), "best": (1768, 128, 1), "rle": (0, 0, 1), "huffman_only": (0, 0, 0)}
BLANK = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff])


def cases():
    fix = os.path.join(HERE, "ref_inputs")
    files = ["pg11.txt", "short.bin", "issue_18_201911.bin", "dump.bin"] + sorted(
        os.path.join("afl", os.path.basename(f)) for f in glob.glob(os.path.join(fix, "afl", "*")))
    for f in files:
        yield f, open(os.path.join(fix, f), "rb").read()
    yield "synthetic/empty", b""
    yield "synthetic/deflate_late", b"Deflate late"
    yield "synthetic/short_run", bytes([10, 10, 10, 10, 10, 55])
    yield "synthetic/zeros_65537", bytes(65537)


def main():
    out = {"_about": __doc__.strip().split("\n")[0], "streams": {}, "digests": {}}
    for name, data in cases():
        for lvl, (c, l, m) in LV.items():
            if name.startswith("afl") and lvl not in ("default", "fast"):
                continue
            for wname, w in (("raw", 0), ("zlib", 1), ("gzip", 2)):
                if w == 2:
                    z = ob.encode_gzip(data, BLANK, opts=ob.make_opts(c, l, m))
                else:
                    z = ob.encode(data, opts=ob.make_opts(c, l, m, w))
                key = "%s|%s|%s" % (name, lvl, wname)
                out["digests"][key] = [len(z), hashlib.sha256(z).hexdigest()]
                if name.startswith("synthetic") or name == "short.bin":
                    if len(z) <= 200:
                        out["streams"][key] = z.hex()
    json.dump(out, open(os.path.join(HERE, "oracle_digests.json"), "w"), indent=0, sort_keys=True)
    print(len(out["digests"]), "digests,", len(out["streams"]), "full streams")


if __name__ == "__main__":
    main()
