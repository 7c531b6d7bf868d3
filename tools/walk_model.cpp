// walk_model.cpp -- CPU model of the sorted walk as k_match3 runs it (replaces band_sim{,2,3}.cpp, whose results are
// in DESIGN.md): the positions of an epoch in (hash, position) order, 64 per wave; a lane asks about four candidates
// per group and parks at the first probe hit or at the end of its segment; a service settles the parked lanes.
// What is explored here is WHEN to service: after every block of G groups (the kernel as first built), or only
// when fewer than MINW lanes still walk / at least MINP lanes wait -- a parked lane costs nothing while the others
// walk, a service pass costs its ~80 vector instructions whatever its occupancy.
//   g++ -O2 -o /tmp/walk_model tools/walk_model.cpp && /tmp/walk_model file [checks] [max_bytes] [G] [MINW] [MINP] [REFILL]
// Prints groups, services, set-ups per 64 positions, the occupancies, and a price in vector wave-instructions
// (11 per group, 80 per service, 150 per set-up) and in serial latency events (a group, a service).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

static inline uint32_t hash3(const uint8_t* d) { return ((d[0] & 31u) << 10) ^ ((uint32_t)d[1] << 5) ^ d[2]; }

struct Lane {
    int state;  // 0 walking, 1 parked on a hit, 2 parked at the end of a segment, 3 done / idle
    size_t i;   // own index in the epoch-pair sorted array
    uint32_t k, n1, n2, seg;  // candidates visited so far, own-epoch candidates, previous-epoch ones, segment
    uint32_t best, maxlen, hitk;
};

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    uint32_t checks = argc > 2 ? atoi(argv[2]) : 128;
    size_t maxb = argc > 3 ? strtoull(argv[3], 0, 10) : (size_t)8000000;
    uint32_t G = argc > 4 ? atoi(argv[4]) : 3;
    uint32_t MINW = argc > 5 ? atoi(argv[5]) : 65;  // service when fewer lanes walk (65: after every block)
    uint32_t MINP = argc > 6 ? atoi(argv[6]) : 0;   // ... or when at least this many wait (0: off)
    uint32_t REFILL = argc > 7 ? atoi(argv[7]) : 0; // idle lanes take new positions at a service once this many are idle (0: never)
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> d(maxb + 300);
    size_t n = fread(d.data(), 1, maxb, f);
    fclose(f);
    const uint32_t W = 32768;
    uint64_t visits_prev = 0, pos_prev = 0, positions = 0, visits = 0, groups = 0, services = 0, setups = 0, setup_lanes = 0, walk_slots = 0, serv_lanes = 0, asked = 0;
    std::vector<uint32_t> ss;
    for (size_t e = 0; e * W < n; e++) {
        size_t lo = e ? (e - 1) * W : 0, hi = std::min(n, (e + 1) * W);
        ss.clear();
        for (size_t p = lo; p < hi; p++)
            if (p + 2 < n) ss.push_back((uint32_t)p);
        // (hash, epoch, position): the own epoch's entries of a bucket behind the previous epoch's
        std::stable_sort(ss.begin(), ss.end(), [&](uint32_t a, uint32_t b) { return hash3(&d[a]) < hash3(&d[b]); });
        size_t m = ss.size();
        std::vector<uint32_t> bstart(m, 0);
        for (size_t i = 0; i < m; i++) bstart[i] = (i && hash3(&d[ss[i]]) == hash3(&d[ss[i - 1]])) ? bstart[i - 1] : (uint32_t)i;
        std::vector<uint32_t> own;
        for (size_t i = 0; i < m; i++)
            if (ss[i] >= e * W) own.push_back((uint32_t)i);
        // candidate c-th of entry i is ss[i - c] (position order inside a bucket = nearest first), while within the window
        size_t q = 0;
        while (q < own.size()) {
            Lane L[64];
            for (int l = 0; l < 64; l++) L[l].state = 3;
            auto setup = [&](int l) {
                size_t i = own[q++];
                Lane& a = L[l];
                a.i = i;
                uint32_t p = ss[i], c = 0;
                for (size_t j = i; j > bstart[i] && c < checks; j--) {
                    if (p - ss[j - 1] > W) break;
                    c++;
                }
                a.n1 = c;  // (own and previous epoch are not told apart here: one list, one possible switch event)
                uint32_t ownc = 0;
                for (size_t j = i; j > bstart[i] && ownc < c && ss[j - 1] >= e * W; j--) ownc++;
                a.n2 = c - ownc;
                a.n1 = ownc;
                a.seg = a.n1 ? 0 : 1;
                a.k = 0;
                a.best = 1;
                a.maxlen = (uint32_t)std::min<size_t>(n - p, 258);
                a.state = c ? 1 : 3;  // the first candidate goes straight to the service
                a.hitk = 1;
                positions++;
                visits += c;
                visits_prev += c - ownc;
                pos_prev += (c - ownc) ? 1 : 0;
            };
            uint32_t nl = (uint32_t)std::min<size_t>(64, own.size() - q);
            for (uint32_t l = 0; l < nl; l++) setup((int)l);
            setups++;
            setup_lanes += nl;
            uint32_t since = 0;
            for (;;) {
                uint32_t nw = 0, np = 0, idle = 0;
                for (int l = 0; l < 64; l++) {
                    nw += L[l].state == 0;
                    np += L[l].state == 1 || L[l].state == 2;
                    idle += L[l].state == 3;
                }
                if (!nw && !np) break;
                bool serv = np && (nw == 0 || (since >= G && (nw < MINW || (MINP && np >= MINP))));
                if (serv) {
                    services++;
                    serv_lanes += np;
                    since = 0;
                    for (int l = 0; l < 64; l++) {
                        Lane& a = L[l];
                        if (a.state == 1) {
                            uint32_t p = ss[a.i], c = ss[a.i - a.hitk], len = 0;
                            while (len < a.maxlen && d[p + len] == d[c + len]) len++;
                            if (len > a.best) a.best = len;
                            a.k = a.hitk;
                            uint32_t segend = a.seg == 0 ? a.n1 : a.n1 + a.n2;
                            if (len == a.maxlen)
                                a.state = 3;
                            else if (a.k < segend)
                                a.state = 0;
                            else
                                a.state = 2;
                        }
                        if (a.state == 2) {
                            if (a.seg == 0 && a.n2) {
                                a.seg = 1;
                                a.state = 0;
                            } else
                                a.state = 3;
                        }
                    }
                    if (REFILL) {
                        idle = 0;
                        for (int l = 0; l < 64; l++) idle += L[l].state == 3;
                        if (idle >= REFILL && q < own.size()) {
                            uint32_t took = 0;
                            for (int l = 0; l < 64 && q < own.size(); l++)
                                if (L[l].state == 3) {
                                    setup(l);
                                    took++;
                                }
                            setups++;
                            setup_lanes += took;
                        }
                    }
                    continue;
                }
                // one group of four steps
                groups++;
                since++;
                for (int l = 0; l < 64; l++) {
                    Lane& a = L[l];
                    if (a.state != 0) continue;
                    walk_slots++;
                    uint32_t p = ss[a.i], b = a.best, segend = a.seg == 0 ? a.n1 : a.n1 + a.n2;
                    bool hit = false;
                    for (uint32_t s = 1; s <= 4 && a.k + s <= segend; s++) {
                        asked++;
                        uint32_t c = ss[a.i - (a.k + s)];
                        if (d[c + b - 1] == d[p + b - 1] && d[c + b] == d[p + b]) {
                            a.hitk = a.k + s;
                            a.state = 1;
                            hit = true;
                            break;
                        }
                    }
                    if (hit) continue;
                    a.k = std::min(a.k + 4, segend);
                    if (a.k >= segend) a.state = (a.seg == 0 && a.n2) ? 2 : 3;  // (the end of the last segment needs no service)
                }
            }
        }
    }
    printf("candidates in the previous epoch: %.3f of all visits, %.3f of the positions have some\n", (double)visits_prev / visits, (double)pos_prev / positions);
    double per = 64.0 / positions;
    printf("positions %llu visits/pos %.2f  G=%u MINW=%u MINP=%u REFILL=%u\n", (unsigned long long)positions, (double)visits / positions, G,
           MINW, MINP, REFILL);
    printf("per 64 positions: groups %.2f (walking lanes %.1f, %.2f probes asked per visit), services %.2f (%.1f lanes each), set-ups %.2f (%.1f lanes)\n",
           groups * per, (double)walk_slots / groups, (double)asked / visits, services * per, (double)serv_lanes / services, setups * per,
           (double)setup_lanes / setups);
    double valu = groups * 11.0 + services * 80.0 + setups * 150.0;
    printf("price: %.0f vector wave-instructions per 64 positions (11/group, 80/service, 150/set-up); %.1f serial events (groups + services)\n",
           valu * per, (groups + services) * per);
    return 0;
}
