// band_sim3.cpp -- CPU model of k_match2 with lane refill: a wave keeps walking while lanes that are done take
// the next entries of the epoch's sorted array at a service point, once at least T lanes are idle
// (T = 65: never, the kernel as it is).  Counts wave-steps, services, set-ups; prices them like band_sim2.
//   g++ -O2 -o /tmp/band_sim3 tools/band_sim3.cpp && /tmp/band_sim3 file [checks] [max_bytes] [R] [T]
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
    uint32_t R = argc > 4 ? atoi(argv[4]) : 12;
    uint32_t T = argc > 5 ? atoi(argv[5]) : 65;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> d(maxb + 8);
    size_t n = fread(d.data(), 1, maxb, f);
    fclose(f);
    const uint32_t W = 32768;
    uint64_t positions = 0, visits = 0, wave_steps = 0, services = 0, setups = 0, setup_lanes = 0, walking_slots = 0, cmp_rounds = 0;
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
        // 16 waves share the epoch's queue; model them one after the other (a wave = one stream of batches)
        size_t q = 0;
        while (q < own.size()) {
            uint32_t k[64], best[64], maxlen[64];
            size_t idx[64];
            int state[64];  // 0 walking, 1 parked, 2 done/idle
            for (int l = 0; l < 64; l++) state[l] = 2;
            auto setup = [&](int l) {
                size_t i = own[q++];
                idx[l] = i;
                positions++;
                visits += cnt[i];
                uint32_t p = ss[i];
                maxlen[l] = (uint32_t)std::min<size_t>(n - p, 258);
                best[l] = 1;
                k[l] = 1;
                state[l] = cnt[i] ? 0 : 2;
            };
            uint32_t nl = (uint32_t)std::min<size_t>(64, own.size() - q);
            for (uint32_t l = 0; l < nl; l++) setup((int)l);
            setups++;
            setup_lanes += nl;
            uint32_t step = 0;
            for (;;) {
                bool anyw = false, anyp = false;
                uint32_t idle = 0;
                for (int l = 0; l < 64; l++) {
                    anyw |= state[l] == 0;
                    anyp |= state[l] == 1;
                    idle += state[l] == 2;
                }
                if (!anyw && !anyp) break;
                if (anyw) {
                    wave_steps++;
                    step++;
                    for (int l = 0; l < 64; l++) {
                        if (state[l] != 0) continue;
                        walking_slots++;
                        size_t i = idx[l];
                        uint32_t p = ss[i], c = ss[i - k[l]], b = best[l];
                        if (d[c + b - 1] == d[p + b - 1] && d[c + b] == d[p + b])
                            state[l] = 1;
                        else {
                            k[l]++;
                            if (k[l] > cnt[i]) state[l] = 2;
                        }
                    }
                }
                bool serv = !anyw || (step % R) == 0;
                if (!serv) continue;
                step = 0;
                uint32_t np = 0, mr = 0;
                for (int l = 0; l < 64; l++) {
                    if (state[l] != 1) continue;
                    np++;
                    size_t i = idx[l];
                    uint32_t p = ss[i], c = ss[i - k[l]], len = 0;
                    while (len < maxlen[l] && d[p + len] == d[c + len]) len++;
                    mr = std::max(mr, len / 16 + 1);
                    if (len > best[l]) best[l] = len;
                    k[l]++;
                    state[l] = (len == maxlen[l] || k[l] > cnt[i]) ? 2 : 0;
                }
                services++;
                cmp_rounds += mr;
                idle = 0;
                for (int l = 0; l < 64; l++) idle += state[l] == 2;
                if (idle >= T && idle < 64 && q < own.size()) {  // refill (a full set-up pass under the mask)
                    uint32_t took = 0;
                    for (int l = 0; l < 64 && q < own.size(); l++)
                        if (state[l] == 2) {
                            setup(l);
                            took++;
                        }
                    setups++;
                    setup_lanes += took;
                }
            }
        }
    }
    printf("positions %llu visits/pos %.2f\n", (unsigned long long)positions, (double)visits / positions);
    printf("R=%u T=%u: wave-steps %llu (walking-lane share %.3f), services %llu, set-ups %llu (%.1f lanes each)\n", R, T,
           (unsigned long long)wave_steps, (double)walking_slots / (64.0 * wave_steps), (unsigned long long)services,
           (unsigned long long)setups, (double)setup_lanes / setups);
    double instr = (double)wave_steps * 9.5 + (double)services * 60 + (double)setups * 150;
    printf("model: %.2f wave-instructions per position (9.5/step, 60/service, 150/set-up)\n", instr / positions);
    return 0;
}
