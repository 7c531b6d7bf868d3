// band_sim2.cpp -- CPU model of the sorted-order match kernel with parking (k_match2):
// positions of an epoch, taken in (hash, position) order 64 at a time; each lane walks its own candidates
// S[i-1], S[i-2], ... (exact count known), a lane whose probe hits parks until the wave's next service
// (every R steps, or at once when no lane walks); the first candidate is compared at setup.
//   g++ -O2 -o /tmp/band_sim2 tools/band_sim2.cpp && /tmp/band_sim2 file [checks] [max_bytes] [R]
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
    uint32_t R = argc > 4 ? atoi(argv[4]) : 8;
    int sort_by_cnt = argc > 5 ? atoi(argv[5]) : 0;
    uint32_t TH = argc > 6 ? atoi(argv[6]) : 1;  // service only when at least TH lanes wait (or none walks)
    uint32_t MINW = argc > 7 ? atoi(argv[7]) : 0;  // ... or when fewer than MINW lanes still walk
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> d(maxb + 8);
    size_t n = fread(d.data(), 1, maxb, f);
    fclose(f);
    const uint32_t W = 32768;
    uint64_t positions = 0, visits = 0, wave_steps = 0, batches = 0, services = 0, cmp_rounds = 0, parked_serv = 0;
    uint64_t first_rounds = 0, walking_slots = 0;
    std::vector<uint32_t> ss;
    for (size_t e = 0; e * W < n; e++) {
        size_t lo = e ? (e - 1) * W : 0, hi = std::min(n, (e + 1) * W);
        ss.clear();
        for (size_t p = lo; p < hi; p++)
            if (p + 2 < n) ss.push_back((uint32_t)p);
        std::stable_sort(ss.begin(), ss.end(), [&](uint32_t a, uint32_t b) { return hash3(&d[a]) < hash3(&d[b]); });
        size_t m = ss.size();
        std::vector<uint32_t> cnt(m, 0), bstart(m, 0), own;
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
            own.push_back((uint32_t)i);
        }
        if (sort_by_cnt == 1) std::stable_sort(own.begin(), own.end(), [&](uint32_t a, uint32_t b) { return cnt[a] > cnt[b]; });
        if (sort_by_cnt == 2) {  // by the size class of the bucket within the own epoch (what k_sort can know)
            std::vector<uint32_t> bsz(m, 0);
            for (size_t i = 0; i < m;) {
                size_t j = i;
                uint32_t c = 0;
                while (j < m && bstart[j] == bstart[i]) {
                    c += ss[j] >= e * W;
                    j++;
                }
                for (size_t k = i; k < j; k++) bsz[k] = c;
                i = j;
            }
            auto cls = [&](uint32_t z) { return z >= 65 ? 0 : z >= 49 ? 1 : z >= 33 ? 2 : z >= 25 ? 3 : z >= 17 ? 4 : z >= 13 ? 5 : z >= 9 ? 6 : z >= 7 ? 7 : z >= 5 ? 8 : 13 - (int)z; };
            std::stable_sort(own.begin(), own.end(), [&](uint32_t a, uint32_t b) { return cls(bsz[a]) < cls(bsz[b]); });
        }
        for (size_t b0 = 0; b0 < own.size(); b0 += 64) {
            uint32_t nl = (uint32_t)std::min<size_t>(64, own.size() - b0);
            batches++;
            positions += nl;
            uint32_t k[64], best[64], maxlen[64];
            int state[64];  // 0 walking, 1 parked, 2 done
            uint32_t fr = 0;
            for (uint32_t l = 0; l < nl; l++) {
                size_t i = own[b0 + l];
                uint32_t p = ss[i];
                visits += cnt[i];
                maxlen[l] = (uint32_t)std::min<size_t>(n - p, 258);
                best[l] = 1;
                k[l] = 1;
                state[l] = 2;
                if (cnt[i] == 0) continue;
                uint32_t c = ss[i - 1], len = 0;
                while (len < maxlen[l] && d[p + len] == d[c + len]) len++;
                fr = std::max(fr, len / 8 + 1);
                if (len > 1) best[l] = len;
                k[l] = 2;
                state[l] = (len == maxlen[l] || cnt[i] < 2) ? 2 : 0;
            }
            first_rounds += fr;
            uint32_t step = 0;
            for (;;) {
                bool anyw = false, anyp = false;
                for (uint32_t l = 0; l < nl; l++) {
                    anyw |= state[l] == 0;
                    anyp |= state[l] == 1;
                }
                if (!anyw && !anyp) break;
                if (anyw) {
                    wave_steps++;
                    step++;
                    for (uint32_t l = 0; l < nl; l++) {
                        if (state[l] != 0) continue;
                        walking_slots++;
                        size_t i = own[b0 + l];
                        uint32_t p = ss[i], c = ss[i - k[l]], b = best[l];
                        if (d[c + b - 1] == d[p + b - 1] && d[c + b] == d[p + b])
                            state[l] = 1;
                        else {
                            k[l]++;
                            if (k[l] > cnt[i]) state[l] = 2;
                        }
                    }
                }
                uint32_t npend = 0;
                for (uint32_t l = 0; l < nl; l++) npend += state[l] == 1;
                uint32_t nwalk = 0;
                for (uint32_t l = 0; l < nl; l++) nwalk += state[l] == 0;
                bool serv = !anyw || ((step % R) == 0 && npend >= TH) || (npend > 0 && nwalk < MINW);
                if (serv) step = 0;
                if (!serv) continue;
                uint32_t np = 0, mr = 0;
                for (uint32_t l = 0; l < nl; l++) {
                    if (state[l] != 1) continue;
                    np++;
                    size_t i = own[b0 + l];
                    uint32_t p = ss[i], c = ss[i - k[l]], len = 0;
                    while (len < maxlen[l] && d[p + len] == d[c + len]) len++;
                    mr = std::max(mr, len / 8 + 1);
                    if (len > best[l]) best[l] = len;
                    k[l]++;
                    state[l] = (len == maxlen[l] || k[l] > cnt[i]) ? 2 : 0;
                }
                if (np) {
                    services++;
                    parked_serv += np;
                    cmp_rounds += mr;
                }
            }
        }
    }
    printf("positions %llu visits/pos %.2f batches %llu\n", (unsigned long long)positions, (double)visits / positions,
           (unsigned long long)batches);
    printf("R=%u TH=%u sort=%d: wave-steps/batch %.2f (walking-lane share %.3f), services/batch %.2f, parked lanes/service %.1f, "
           "8-byte compare rounds/service %.2f, first-compare rounds/batch %.2f\n",
           R, TH, sort_by_cnt, (double)wave_steps / batches, (double)walking_slots / (64.0 * wave_steps), (double)services / batches,
           (double)parked_serv / services, (double)cmp_rounds / services, (double)first_rounds / batches);
    double instr = (double)wave_steps * 8 + (double)services * (25 + 0) + (double)cmp_rounds * 14 + batches * (60.0) + first_rounds * 14.0;
    printf("model: %.1f wave-instructions per position (8/step, 25+14/round per service, 60+14/round setup)\n", instr / positions);
    return 0;
}
