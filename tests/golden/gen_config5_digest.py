"""Writes tests/golden/config5_digest.json: length and SHA-256 of the ORACLE's Default stream of BASELINE config 5 at its stated
size -- 8 GiB of the web-text input (tests/datagen.py::webtext) -- raw and zlib.  The GPU tests
(test_config5_at_8_gib_on_one_gpu, ..._over_8_virtual_ranks) hold the range walk of one GPU and the multi-GPU call to it.
The input is generated a MiB segment at a time on all cores and fed to the oracle's streaming encoder in 64 MiB writes (the
reference's write_all in pieces is its write_all of the whole: nothing is flushed; checked here on 200 MB against the one-shot
call).  About six minutes on eight cores; not re-derived by the CPU suite.
    python tests/golden/gen_config5_digest.py [bytes]"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datagen  # noqa: E402
import oracle_binding as ob  # noqa: E402

N = 8 << 30
CHUNK = 64 << 20


_seg = datagen.webtext_segment_bytes


def chunks(n, pool):
    """the input in CHUNK pieces, generated ahead on the pool"""
    per = CHUNK // datagen.WEB_SEGMENT
    n_seg = (n + datagen.WEB_SEGMENT - 1) // datagen.WEB_SEGMENT
    pending = []
    nxt = 0
    done = 0
    while done < n:
        while len(pending) < 3 and nxt < n_seg:
            hi = min(nxt + per, n_seg)
            pending.append(pool.map_async(_seg, range(nxt, hi), chunksize=4))
            nxt = hi
        b = b"".join(pending.pop(0).get())
        if done + len(b) > n:
            b = b[:n - done]
        done += len(b)
        yield b


def digest(n, pool, wrapper):
    s = ob.Stream(ob.make_opts(128, 32, 1, wrapper))
    hin = hashlib.sha256()
    for b in chunks(n, pool):
        hin.update(b)
        s.write_all(b)
    s._chk(ob.lib().deflref_stream_finish(s._s))
    out_len, out_sha = s.output_sha256()
    return {"in_len": n, "in_sha256": hin.hexdigest(), "out_len": out_len, "out_sha256": out_sha}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else N
    with mp.Pool(max(1, min(32, (os.cpu_count() or 2) - 1))) as pool:
        # the streamed form is the one-shot form
        small = b"".join(chunks(200_000_000, pool))
        assert small == datagen.webtext(200_000_000)
        s = ob.Stream(ob.make_opts(128, 32, 1, 0))
        for i in range(0, len(small), CHUNK):
            s.write_all(small[i:i + CHUNK])
        assert s.finish() == ob.encode(small, level=ob.DEFAULT), "chunked write_all differs from the one-shot call"
        del small, s
        out = {}
        for name, wrapper in (("raw", 0), ("zlib", 1)):
            t0 = time.time()
            out[name] = digest(n, pool, wrapper)
            print(name, out[name], round(time.time() - t0, 1), "s", flush=True)
    json.dump({"note": "oracle streams of webtext(%d), Compression::Default; regenerate with gen_config5_digest.py" % n, "digests": out},
              open(os.path.join(HERE, "config5_digest.json" if n == N else "config5_digest_%d.json" % n), "w"), indent=1, sort_keys=True)
