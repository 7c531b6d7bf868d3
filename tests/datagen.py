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


def text_like(n, seed):
    """enwik8-like text: Zipf(1.07) over a 50 000-word vocabulary, sentences, 5 % markup
    (SURVEY.md 8d config 3).  Vectorised so that 100 MB generate in seconds."""
    words, markup = _vocab()
    rng = np.random.default_rng(seed)
    ranks = np.arange(1, len(words) + 1, dtype=np.float64)
    p = ranks ** -1.07
    p /= p.sum()
    enc = [w.encode() for w in words]
    out = bytearray()
    seps = [b" ", b" ", b" ", b" ", b" ", b" ", b", ", b". ", b".\n", b" "]
    m_enc = [m.encode() for m in markup]
    while len(out) < n:
        k = 200000
        idx = rng.choice(len(words), size=k, p=p)
        sp = rng.integers(0, len(seps), size=k)
        mk = rng.random(k) < 0.05
        mi = rng.integers(0, len(m_enc), size=k)
        parts = []
        for i in range(k):
            parts.append(m_enc[mi[i]] if mk[i] else enc[idx[i]])
            parts.append(seps[sp[i]])
        out += b"".join(parts)
    return bytes(out[:n])


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
