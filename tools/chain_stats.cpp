// chain_stats.cpp -- CPU statistics of the hash-chain walk (matching.rs:87-166 with prev_length = 0)
// over a file, used to size the k_match designs.  Not part of the product or the oracle.
//   g++ -O2 -o /tmp/chain_stats tools/chain_stats.cpp && /tmp/chain_stats file [checks] [max_bytes]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

static inline uint32_t hash3(const uint8_t* d) { return ((d[0] & 31u) << 10) ^ ((uint32_t)d[1] << 5) ^ d[2]; }

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    uint32_t checks = argc > 2 ? atoi(argv[2]) : 128;
    size_t maxb = argc > 3 ? strtoull(argv[3], 0, 10) : (size_t)20000000;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> d(maxb + 8);
    size_t n = fread(d.data(), 1, maxb, f);
    fclose(f);
    std::vector<uint32_t> link(n, 0);
    {
        std::vector<int64_t> head(32768, -1);
        for (size_t p = 0; p + 2 < n; p++) {
            uint32_t h = hash3(&d[p]);
            if (head[h] >= 0 && p - head[h] <= 32768) link[p] = (uint32_t)(p - head[h]);
            head[h] = p;
        }
    }
    // per position: visits, compares, improvements, whether compare #1 is on visit #1, the in-window
    // chain length (uncapped count up to 4096)
    const int NB = 12;  // visit buckets: 0,1,2,3-4,5-8,9-16,17-32,33-64,65-127,128(=cap),...
    uint64_t pos_in[NB] = {0}, vis_in[NB] = {0};
    uint64_t tv = 0, tc = 0, ti = 0, first_hit = 0, npos = 0, later_cmp = 0, later_imp = 0;
    uint64_t lenhist[40] = {0};
    // predictor: the number of same-hash positions in [p - 32768, p) (exact chain population)
    uint64_t est_err_under = 0, est_err_over = 0;
    std::vector<uint32_t> cnt(32768, 0);
    for (size_t p = 0; p + 2 < n; p++) {
        // sliding population: add p-1 ... handled below by explicit add/remove
        if (p >= 1) cnt[hash3(&d[p - 1])]++;
        if (p >= 32769) cnt[hash3(&d[p - 32769])]--;
        uint32_t pop = cnt[hash3(&d[p])];
        uint32_t maxlen = (uint32_t)std::min<size_t>(n - p, 258);
        uint32_t best = 1, visits = 0, cmps = 0, imps = 0;
        size_t cand = p;
        bool fh = false;
        for (uint32_t i = 0; i < checks; i++) {
            uint32_t l = link[cand];
            if (!l) break;
            cand -= l;
            if (p - cand > 32768) break;
            visits++;
            if (d[cand + best - 1] == d[p + best - 1] && d[cand + best] == d[p + best]) {
                cmps++;
                if (visits == 1) fh = true;
                uint32_t len = 0;
                while (len < maxlen && d[p + len] == d[cand + len]) len++;
                if (len > best) {
                    best = len;
                    imps++;
                    if (len == maxlen) break;
                }
            }
        }
        npos++;
        tv += visits;
        tc += cmps;
        ti += imps;
        if (fh) {
            first_hit++;
            later_cmp += cmps - 1;
        } else
            later_cmp += cmps;
        int b = visits == 0 ? 0 : visits == 1 ? 1 : visits == 2 ? 2 : visits <= 4 ? 3 : visits <= 8 ? 4 : visits <= 16 ? 5 : visits <= 32 ? 6 : visits <= 64 ? 7 : visits < checks ? 8 : 9;
        pos_in[b]++;
        vis_in[b] += visits;
        uint32_t capped = std::min(pop, checks);
        if (capped < visits) est_err_under += visits - capped;
        else est_err_over += capped - visits;
        lenhist[std::min<uint32_t>(best, 39)]++;
    }
    printf("positions %llu  visits/pos %.2f  compares/pos %.3f  improvements/pos %.3f\n", (unsigned long long)npos,
           (double)tv / npos, (double)tc / npos, (double)ti / npos);
    printf("first visit is a compare: %.3f of positions; other compares/pos %.3f\n", (double)first_hit / npos,
           (double)later_cmp / npos);
    const char* names[NB] = {"0", "1", "2", "3-4", "5-8", "9-16", "17-32", "33-64", "65..cap-1", "cap", "", ""};
    for (int b = 0; b < 10; b++)
        printf("  visits %-10s positions %6.3f  share of visits %6.3f\n", names[b], (double)pos_in[b] / npos,
               (double)vis_in[b] / tv);
    printf("population predictor (exact same-hash count in window, capped): over %.3f under %.3f visits/pos\n",
           (double)est_err_over / npos, (double)est_err_under / npos);
    printf("best length: ");
    for (int i = 1; i < 40; i++) printf("%d:%.3f ", i, (double)lenhist[i] / npos);
    printf("\n");
    return 0;
}
