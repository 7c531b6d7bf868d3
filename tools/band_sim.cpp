// band_sim.cpp -- CPU model of the "sorted band" match kernel (k_match2): positions of an
// epoch pair sorted by (hash, position); a wave takes 64 consecutive sorted entries as its positions and
// every lane visits the entries before its own, all lanes in lockstep; probe hits are recorded under the
// probe of the round start and resolved every R steps.  Prints lane utilisation and resolve work.
//   g++ -O2 -o /tmp/band_sim tools/band_sim.cpp && /tmp/band_sim file [checks] [max_bytes] [R]
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
    size_t maxb = argc > 3 ? strtoull(argv[3], 0, 10) : (size_t)8000000;
    uint32_t R = argc > 4 ? atoi(argv[4]) : 32;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> d(maxb + 8);
    size_t n = fread(d.data(), 1, maxb, f);
    fclose(f);
    const uint32_t W = 32768;
    uint64_t positions = 0, visits = 0, wave_steps = 0, lane_slots = 0, batches = 0, halo_lanes = 0;
    uint64_t rounds = 0, resolve_iters = 0, hits_total = 0, hits_pass = 0, compares = 0, cmp_rounds_wave = 0;
    uint64_t first_cmp_rounds_wave = 0;
    std::vector<uint32_t> ss;
    for (size_t e = 0; e * W < n; e++) {
        size_t lo = e ? (e - 1) * W : 0, hi = std::min(n, (e + 1) * W);
        ss.clear();
        for (size_t p = lo; p < hi; p++)
            if (p + 2 < n) ss.push_back((uint32_t)p);
        std::stable_sort(ss.begin(), ss.end(), [&](uint32_t a, uint32_t b) { return hash3(&d[a]) < hash3(&d[b]); });
        // exact cnt per entry of epoch e
        size_t m = ss.size();
        std::vector<uint32_t> cnt(m, 0), bstart(m, 0);
        for (size_t i = 0; i < m; i++) bstart[i] = (i && hash3(&d[ss[i]]) == hash3(&d[ss[i - 1]])) ? bstart[i - 1] : (uint32_t)i;
        for (size_t i = 0; i < m; i++) {
            uint32_t p = ss[i];
            if (p < e * W) continue;
            uint32_t c = 0;
            for (size_t j = i; j > bstart[i] && c < checks; j--) {
                if (p - ss[j - 1] > W) break;
                c++;
            }
            cnt[i] = c;
        }
        for (size_t j0 = 0; j0 < m; j0 += 64) {
            uint32_t mx = 0, act = 0;
            for (size_t l = j0; l < std::min(m, j0 + 64); l++)
                if (ss[l] >= e * W) {
                    act++;
                    mx = std::max(mx, cnt[l]);
                    visits += cnt[l];
                }
            if (!act) continue;
            batches++;
            positions += act;
            halo_lanes += 64 - act;
            wave_steps += mx;
            lane_slots += 64ull * mx;
            // hit model per lane
            uint32_t best[64], bestd[64];
            bool done[64];
            uint32_t fr = 0;
            for (uint32_t l = 0; l < 64; l++) {
                best[l] = 1;
                bestd[l] = 0;
                done[l] = true;
                size_t i = j0 + l;
                if (i >= m || ss[i] < e * W || cnt[i] == 0) continue;
                done[l] = false;
                // first compare at setup
                uint32_t p = ss[i], c = ss[i - 1];
                uint32_t maxlen = (uint32_t)std::min<size_t>(n - p, 258), len = 0;
                while (len < maxlen && d[p + len] == d[c + len]) len++;
                fr = std::max(fr, len / 8 + 1);
                if (len > 1) {
                    best[l] = len;
                    bestd[l] = p - c;
                    if (len == maxlen) done[l] = true;
                }
            }
            first_cmp_rounds_wave += fr;
            for (uint32_t k0 = 2; k0 <= mx; k0 += R) {
                rounds++;
                uint32_t mxh = 0, mxcr = 0;
                for (uint32_t l = 0; l < 64; l++) {
                    size_t i = j0 + l;
                    if (done[l] || i >= m) continue;
                    uint32_t p = ss[i];
                    uint32_t maxlen = (uint32_t)std::min<size_t>(n - p, 258);
                    uint32_t b0 = best[l], h = 0, cr = 0;
                    for (uint32_t k = k0; k < k0 + R && k <= cnt[i]; k++) {
                        uint32_t c = ss[i - k];
                        if (d[c + b0 - 1] == d[p + b0 - 1] && d[c + b0] == d[p + b0]) {  // recorded under the round's probe
                            h++;
                            hits_total++;
                            uint32_t b = best[l];
                            if (d[c + b - 1] == d[p + b - 1] && d[c + b] == d[p + b]) {  // exact probe in the resolve
                                hits_pass++;
                                compares++;
                                uint32_t len = 0;
                                while (len < maxlen && d[p + len] == d[c + len]) len++;
                                cr = std::max(cr, len / 8 + 1);
                                if (len > best[l]) {
                                    best[l] = len;
                                    bestd[l] = p - c;
                                    if (len == maxlen) {
                                        done[l] = true;
                                        break;
                                    }
                                }
                            }
                        }
                    }
                    mxh = std::max(mxh, h);
                    mxcr += cr;
                }
                resolve_iters += mxh;
                cmp_rounds_wave += mxcr;
            }
        }
    }
    printf("positions %llu visits/pos %.2f  batches %llu  lanes/batch %.1f\n", (unsigned long long)positions,
           (double)visits / positions, (unsigned long long)batches, (double)positions / batches);
    printf("wave-steps/pos %.3f (x64 = %.1f lane slots/pos), utilisation %.3f\n", (double)wave_steps / positions,
           64.0 * wave_steps / positions, (double)visits / lane_slots);
    printf("R=%u: rounds/batch %.2f, resolve iterations/round %.2f, recorded hits/pos %.3f, exact-probe passes/pos %.3f\n", R,
           (double)rounds / batches, (double)resolve_iters / rounds, (double)hits_total / positions,
           (double)hits_pass / positions);
    printf("first-compare 8-byte rounds per batch (max over lanes) %.2f\n", (double)first_cmp_rounds_wave / batches);
    return 0;
}
