"""CPU aid: what share of the match table does the parser look at?  
For `Compression::Default` on the bench text: the positions the reference's lazy parser searches on its one real
path, and the positions that lie on the path from ANY possible entry of a 1 KiB segment's entry zone (what a
data-parallel parse that does not know the entry would have to have).  usage: path_fraction.py [bytes]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen, hostsim_binding as hs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
LAZY, SEG, ZONE = 32, 1024, 576
data = datagen.text_like(n, 0x656E)
M = np.array(hs.match_table(data, 128), dtype=np.uint32)
L = (M & 0xffff).astype(np.int64)
D = (M >> 16).astype(np.int64)
ok = (L >= 3) & ~((L == 3) & (D > 8192))  # lz77.rs:275-278 too_far
# one restart step from j: the searched positions and the next restart (stages.h parse_step, lazy)
adv = np.ones(n, dtype=np.int64)
looked = np.ones(n, dtype=np.int64)  # entries of M the step reads: j, then one per deferral look-ahead
for j in range(n):
    if j + 2 >= n or not ok[j]:
        continue
    a, l, k = j, L[j], 1
    while l < LAZY and a + 3 < n:
        k += 1
        if L[a + 1] > l:
            a += 1
            l = L[a]
        else:
            break
    adv[j] = (a - j) + l
    looked[j] = k
# the real path
seen = np.zeros(n, dtype=bool)
j = 0
while j < n:
    seen[j:j + looked[j]] = True
    j += adv[j]
print("entries the one real path reads: %.1f %%" % (100.0 * seen.mean()))
# union over all entries of each segment's entry zone
need = np.zeros(n, dtype=bool)
for s0 in range(0, n, SEG):
    end = min(n, s0 + SEG)
    mark = np.zeros(end - s0 + 600, dtype=bool)
    for e in range(s0, min(end, s0 + ZONE)):
        j = e
        while j < end and not mark[j - s0]:
            mark[j - s0] = True
            j += adv[j]
    idx = np.nonzero(mark[:end - s0])[0] + s0
    for j in idx:
        need[j:min(n, j + looked[j])] = True
print("entries some path from a segment's entry zone reads: %.1f %%" % (100.0 * need.mean()))
