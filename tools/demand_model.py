"""CPU model behind the go / no-go of a demand-driven match search.
k_match3 searches every position; the reference's parser consults about a third of them.  Two ways to search less:
 (a) proposal -> parse -> exact search of the positions the proposed path reads -> parse again ..., to a fix-point:
     a cheap table (budget 8) everywhere, the exact table (budget 128) only where some parse has looked;
 (b) frontier search from every possible entry of a segment's entry zone, for several segment / zone sizes.
Printed: share of positions searched exactly, passes over the data (each pass = one parse + one sparse search launch),
and for (b) the longest chain of dependent searches per segment (what a lane-serial walker would have to wait for).
usage: demand_model.py [bytes] [text|silesia]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen, hostsim_binding as hs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
data = datagen.text_like(n, 0x656E) if kind == "text" else datagen.silesia_like(scale=n / 212.1e6)
n = len(data)
LAZY = 32


def table(checks):
    M = np.array(hs.match_table(data, checks), dtype=np.uint32)
    L = (M & 0xffff).astype(np.int64)
    D = (M >> 16).astype(np.int64)
    L[~((L >= 3) & ~((L == 3) & (D > 8192)))] = 0  # lz77.rs:275-278 too_far; shorter than 3: no match
    return L


LX, LP = table(128), table(8)
print("%s, %d bytes: positions where the budget-8 table differs from the exact one: %.1f %%" % (kind, n, 100.0 * (LX != LP).mean()))


def step(L, j):
    """one restart step of the lazy parser from j over length table L: (next restart, entries of the table it read)"""
    if j + 2 >= n or L[j] == 0:
        return j + 1, 1
    a, l, k = j, L[j], 1
    while l < LAZY and a + 3 < n:
        k += 1
        if L[a + 1] > l:
            a += 1
            l = L[a]
        else:
            break
    return a + l, k


# ---- (a) fix-point ------------------------------------------------------------------------------------------------
known = np.zeros(n, dtype=bool)
Lmix = LP.copy()
it = 0
while True:
    it += 1
    read = np.zeros(n, dtype=bool)
    j = 0
    while j < n:
        nx, k = step(Lmix, j)
        read[j:j + k] = True
        j = nx
    new = read & ~known
    print("(a) pass %d: the path reads %.1f %% of the entries, %.1f %% of them not exact yet" % (it, 100.0 * read.mean(), 100.0 * new.sum() / max(1, read.sum())))
    if not new.any():
        break
    known |= new
    Lmix[new] = LX[new]
    if it >= 40:
        print("(a) ... stopped after 40 passes")
        break
print("(a) searched exactly: %.1f %% of the positions in %d passes (plus the budget-8 table everywhere)" % (100.0 * known.mean(), it))

# ---- (b) frontier from the entry zone -----------------------------------------------------------------------------------
adv = np.ones(n, dtype=np.int64)
looked = np.ones(n, dtype=np.int64)
for j in range(n):
    nx, k = step(LX, j)
    adv[j] = nx - j
    looked[j] = k
seen = np.zeros(n, dtype=bool)
j = 0
while j < n:
    seen[j:j + looked[j]] = True
    j += adv[j]
print("(b) the one real path reads %.1f %% of the entries" % (100.0 * seen.mean()))
for SEG, ZONE in ((1024, 576), (4096, 576), (16384, 576), (1024, 64), (4096, 64), (256, 64)):
    need = np.zeros(n, dtype=bool)
    depth = []
    for s0 in range(0, n, SEG):
        end = min(n, s0 + SEG)
        mark = np.zeros(end - s0 + 600, dtype=bool)
        dmax = 0
        for e in range(s0, min(end, s0 + ZONE)):
            j, d = e, 0
            while j < end and not mark[j - s0]:
                mark[j - s0] = True
                j += adv[j]
                d += 1
            dmax = max(dmax, d)
        depth.append(dmax)
        idx = np.nonzero(mark[:end - s0])[0] + s0
        for j in idx:
            need[j:min(n, j + looked[j])] = True
    print("(b) segments of %5d, entry zone %3d: %.1f %% of the entries lie on a path from some entry; longest chain of dependent "
          "steps per segment: mean %.0f, max %d" % (SEG, ZONE, 100.0 * need.mean(), float(np.mean(depth)), max(depth)))
