"""Deterministic synthetic inputs shared by the CPU and GPU parity tests (and bench.py)."""
import random

import numpy as np


def rng_bytes(n, seed):
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8).tobytes()


_WORDS = None


def _vocab():
    global _WORDS
    if _WORDS is None:
        r = random.Random(0x656E77696B38)
        letters = "etaoinshrdlcumwfgypbvkjxqz"
        weights = [12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2,
                   2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07]
        words = []
        for _ in range(50000):
            ln = r.randint(2, 12)
            words.append("".join(r.choices(letters, weights, k=ln)))
        markup = ["[[", "]]", "<page>", "</page>", "<title>", "&quot;", "==", "{{", "}}", "|"]
        _WORDS = (words, markup)
    return _WORDS


_TABLE = None


def _table():
    """fixed-width byte rows: one per vocabulary word, markup token and separator"""
    global _TABLE
    if _TABLE is None:
        words, markup = _vocab()
        seps = [" ", " ", " ", " ", " ", " ", ", ", ". ", ".\n", " "]
        rows = [w.encode() for w in words] + [m.encode() for m in markup] + [x.encode() for x in seps]
        width = max(len(r) for r in rows)
        mat = np.zeros((len(rows), width), dtype=np.uint8)
        lens = np.zeros(len(rows), dtype=np.int64)
        for i, r in enumerate(rows):
            mat[i, :len(r)] = np.frombuffer(r, dtype=np.uint8)
            lens[i] = len(r)
        ranks = np.arange(1, len(words) + 1, dtype=np.float64)
        p = ranks ** -1.07
        p /= p.sum()
        _TABLE = (mat, lens, len(words), len(markup), len(seps), np.cumsum(p))
    return _TABLE


def text_like(n, seed):
    """enwik8-like text: Zipf(1.07) over a 50 000-word vocabulary, sentences, 5 % markup tokens
    (SURVEY.md 8d config 3).  Fully vectorised: 100 MB generate in a few seconds."""
    mat, lens, nw, nm, ns, cdf = _table()
    rng = np.random.default_rng(seed)
    out = []
    have = 0
    col = np.arange(mat.shape[1])
    while have < n:
        k = 1 << 20
        # (the generator is drawn from a million tokens at a time whatever n is -- the bytes of a seed must not depend on how
        # they are made -- but only the tokens that can be needed are built: a word and its separator are three bytes or more.
        # mixed() asks for pieces of 7 to 40 000 bytes by the thousand: 133 s for 20 MB before, 3 s now)
        u1, u2 = rng.random(k), rng.random(k)
        mi = rng.integers(0, nm, size=k) + nw
        sp = rng.integers(0, ns, size=k) + nw + nm
        k = min(k, (n - have) // 3 + 2)
        u1, u2, mi, sp = u1[:k], u2[:k], mi[:k], sp[:k]
        w = np.searchsorted(cdf, u1, side="right").clip(0, nw - 1)
        mk = u2 < 0.05
        tok = np.where(mk, mi, w)
        rows = np.empty(2 * k, dtype=np.int64)
        rows[0::2] = tok
        rows[1::2] = sp
        mask = col[None, :] < lens[rows][:, None]
        chunk = mat[rows][mask]
        out.append(chunk)
        have += chunk.size
    return np.concatenate(out)[:n].tobytes()


def mixed(n, seed):
    """runs, periodic pieces, text and noise glued together: exercises every block type."""
    r = random.Random(seed)
    out = bytearray()
    while len(out) < n:
        kind = r.randint(0, 5)
        ln = r.choice([7, 100, 258, 259, 1000, 5000, 40000])
        if kind == 0:
            out += rng_bytes(ln, r.getrandbits(32))
        elif kind == 1:
            out += bytes([r.getrandbits(8)]) * ln
        elif kind == 2:
            per = rng_bytes(r.choice([2, 3, 5, 17, 300, 4000]), r.getrandbits(32))
            out += (per * (ln // len(per) + 1))[:ln]
        elif kind == 3:
            out += text_like(ln, r.getrandbits(32))
        elif kind == 4:
            out += bytes(r.choice([0, 0, 0, 1, 255]) for _ in range(min(ln, 3000)))
        else:
            if len(out) > 40000:
                d = r.randint(1, 33000)
                s = len(out) - d
                out += out[s:s + min(ln, d)]
    return bytes(out[:n])


def silesia_like(seed=0x53494C45, scale=1.0):
    """about 211.9 MB in twelve pieces by entropy class, with the sizes of
    the Silesia corpus files: text (10.2, 6.6, 41.5 MB), structured binary records (51.2, 6.2, 21.6 MB),
    repetitive database rows (33.6, 10.1 MB), smooth 16-bit samples (10.0, 8.5 MB), near-incompressible
    (7.3 MB), markup-heavy text (5.3 MB).  `scale` shrinks every piece (tests)."""
    rng = np.random.default_rng(seed)
    mb = lambda x: max(1000, int(x * 1e6 * scale))

    def records(n, width, seed2):
        r = np.random.default_rng(seed2)
        rows = n // width + 1
        tpl = r.integers(0, 256, size=width, dtype=np.uint8)
        a = np.tile(tpl, (rows, 1))
        cnt = np.arange(rows, dtype=np.uint32)
        a[:, 4:8] = cnt.view(np.uint8).reshape(rows, 4)                      # a running counter
        k = max(1, width // 8)
        cols = r.choice(np.arange(8, width), size=k, replace=False)
        a[:, cols] = r.integers(0, 16, size=(rows, k), dtype=np.uint8)        # a few low-entropy fields
        hot = r.random(rows) < 0.02
        a[hot, 8:] = r.integers(0, 256, size=(int(hot.sum()), width - 8), dtype=np.uint8)  # the odd noisy record
        return a.reshape(-1)[:n].tobytes()

    def rows_db(n, seed2):
        r = np.random.default_rng(seed2)
        pool = text_like(400 * 160, int(r.integers(1 << 30)))
        lens = r.integers(60, 160, size=400)
        base = [pool[i * 160:i * 160 + int(lens[i])] for i in range(400)]
        idx = r.integers(0, len(base), size=n // 100 + 1)
        rows = [base[j] + (b"|%08d\n" % (i * 7)) for i, j in enumerate(idx.tolist())]
        return b"".join(rows)[:n].ljust(n, b"\n")

    def samples16(n, seed2):
        r = np.random.default_rng(seed2)
        d = r.integers(-24, 25, size=n // 2 + 1, dtype=np.int32)
        v = (np.cumsum(d) + 30000).astype(np.uint16)
        return v.view(np.uint8)[:n].tobytes()

    parts = [text_like(mb(10.2), int(rng.integers(1 << 30))), text_like(mb(6.6), int(rng.integers(1 << 30))),
             text_like(mb(41.5), int(rng.integers(1 << 30))),
             records(mb(51.2), 96, int(rng.integers(1 << 30))), records(mb(6.2), 40, int(rng.integers(1 << 30))),
             records(mb(21.6), 256, int(rng.integers(1 << 30))),
             rows_db(mb(33.6), int(rng.integers(1 << 30))), rows_db(mb(10.1), int(rng.integers(1 << 30))),
             samples16(mb(10.0), int(rng.integers(1 << 30))), samples16(mb(8.5), int(rng.integers(1 << 30))),
             rng_bytes(mb(7.3), int(rng.integers(1 << 30))),
             text_like(mb(5.3), int(rng.integers(1 << 30)))]
    return b"".join(parts)


_WEB = None


def _web_table():
    """webtext(): vocabulary words, HTML-ish template pieces and separators as one flat byte array"""
    global _WEB
    if _WEB is None:
        words, _ = _vocab()
        tags = ["<div class=\"post\">", "</div>", "<p>", "</p>", "<a href=\"http://www.", ".com/", ".html\">", "</a>",
                "<span>", "</span>", "<li>", "</li>", "<ul>", "</ul>", "<br/>", "<h2>", "</h2>", "&nbsp;", "&amp;",
                "<img src=\"/img/", ".png\" alt=\"", "\"/>", "<td>", "</td>", "<tr>", "</tr>", "\n", "\n\n"]
        seps = [" ", " ", " ", " ", " ", ", ", ". ", "? ", " - ", " ", ""]  # (the empty one follows a tag)
        rows = [w.encode() for w in words] + [t.encode() for t in tags] + [x.encode() for x in seps]
        lens = np.array([len(r) for r in rows], dtype=np.int64)
        starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
        flat = np.frombuffer(b"".join(rows), dtype=np.uint8)
        ranks = np.arange(1, len(words) + 1, dtype=np.float64)
        p = ranks ** -1.05
        p /= p.sum()
        _WEB = (flat, starts, lens, len(words), len(tags), len(seps), np.cumsum(p))
    return _WEB


WEB_SEGMENT = 1 << 20


def webtext_segment(index, seed=0x57454254):
    """One 1 MiB segment of the config-5 workload: HTML-ish templates around Zipf(1.05)
    words, generated from seed ^ index alone -- any rank regenerates exactly its own part of the 8 GiB
    input (plus the halo it needs) without anybody shipping the whole."""
    flat, starts, lens, nw, nt, ns, cdf = _web_table()
    rng = np.random.default_rng((seed ^ index) & 0xFFFFFFFFFFFF)
    k = 190_000  # tokens: about 1.1 MiB (mean token + separator is 6.1 bytes); topped up below if short
    out = []
    have = 0
    while have < WEB_SEGMENT:
        w = np.searchsorted(cdf, rng.random(k), side="right").clip(0, nw - 1)
        tg = rng.random(k) < 0.12
        ti = rng.integers(0, nt, size=k) + nw
        sp = rng.integers(0, ns - 1, size=k) + nw + nt
        rows = np.empty(2 * k, dtype=np.int64)
        rows[0::2] = np.where(tg, ti, w)
        rows[1::2] = np.where(tg, nw + nt + ns - 1, sp)
        ln = lens[rows]
        total = int(ln.sum())
        src0 = np.repeat(starts[rows] - (np.cumsum(ln) - ln), ln)
        chunk = flat[src0 + np.arange(total)]
        out.append(chunk)
        have += total
    return (np.concatenate(out) if len(out) > 1 else out[0])[:WEB_SEGMENT]


def webtext_segment_bytes(index):
    """(a top-level function for process pools: the 8 GiB of config 5 are generated on all cores)"""
    return webtext_segment(index).tobytes()


def webtext(n, seed=0x57454254, start=0):
    """bytes [start, start + n) of the web-text workload"""
    first, last = start // WEB_SEGMENT, (start + n + WEB_SEGMENT - 1) // WEB_SEGMENT
    parts = [webtext_segment(i, seed) for i in range(first, max(last, first + 1))]
    a = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    off = start - first * WEB_SEGMENT
    return a[off:off + n].tobytes()
