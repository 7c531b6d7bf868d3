// deflate_kernels.hip -- gfx950 (MI355X) kernels of the DEFLATE encode path.
//
// One kernel per stage of stages.h; every kernel names the reference code it stands in for.
// Layout of the per-encode workspace in HBM (n = input bytes, K = ceil(n / SEG) segments,
// nb <= n / 31744 + 1 blocks):
//   S      u16[n]      per 32 KiB epoch: positions sorted by (filing hash, position)  2 B/byte   (k_sort -> k_match3; stored as 2 x position)
//   B      u16[32776 per epoch]  start of every hash bucket in the epoch's S       2 B/byte
//   link   u16[n]      distance to the previous position with the same hash (k_links -> k_match: the first two
//                      epochs of a re-warmed stream, streams with late-filed positions)
//   M, Mq  u32[n]      longest_match(prev_length = 0) at full / quarter budget   4 (+4) B/byte
//   adv    u16[n]      restart step length from every position                   2 B/byte
//   J      u16[n]      scratch of the per-segment exit sweep                     2 B/byte
//   X[l]   u16[K_l*ZONE]  exit tables per level (an exit lies less than MAX_JUMP beyond its unit), E[l] u32[K_l] entry positions
//   tokbuf u32[K*SEG]  tokens per segment, dtok u32[T] tokens in stream order    4 + 4 B/byte
//   per block: ll_freq u32[4][288], d_freq u32[4][32] (a histogram per quarter of the block), BlockHeader, BlockPlan, bstart
// All integer work; the bound is vector-ALU issue / LDS in k_match3 and latency or HBM elsewhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#ifdef MI355_MATCH_STATS
// instrumented build (tools/variants.sh): per-lane counters of match_walk_park, summed over the launch
__device__ unsigned long long g_mstats[16];
__device__ unsigned long long g_sstats[16];  // k_sort's phase clocks (KS_T)
#define MI355_STAT_DECL uint32_t stat_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MI355_STAT(i, v) stat_[i] += (v);
#define MI355_STAT_FLUSH(policy)                                                          \
    for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_mstats[i_], (unsigned long long)stat_[i_]); \
    atomicMax((policy).wgmax, stat_[0]);                                                  \
    if ((threadIdx.x & 63) == 0) {                                                        \
        unsigned long long t_ = wall_clock64();                                           \
        atomicMax((policy).wgend, t_);                                                    \
        atomicAdd((policy).wgsum, t_ - (policy).t0);                                      \
    }
#endif
#include "stages.h"

namespace mi355 {

// ---- device-side accessors -----------------------------------------------------------------
struct GBytes {  // bounds-checked global bytes (reads past the end give 0)
    const uint8_t* d;
    uint64_t n;
    __device__ uint32_t operator()(uint64_t i) const { return i < n ? d[i] : 0u; }
};
struct GM {  // match / run table in global memory
    const uint32_t* m;
    __device__ uint32_t operator()(uint64_t i) const { return m[i]; }
};
struct LdsWin {  // the window of k_match: bytes and links in one window coordinate system
    const uint8_t* by;
    const uint16_t* lk;
    // 4 bytes at any byte offset from two aligned dwords (one ds_read2_b32) + v_alignbyte: an
    // unaligned ds_read_b32 works on gfx950 but is replayed lane by lane
    __device__ uint32_t load32(uint32_t i) const {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(by) + (i >> 2);
        uint32_t lo = w[0], hi = w[1];
        return __builtin_amdgcn_alignbyte(hi, lo, i);  // v_alignbyte_b32 shifts by bits 1:0 of i
    }
    // the window keeps "no earlier position" as 0xFFFF (k_match converts while staging), so the
    // common step needs no separate test for it
    __device__ uint32_t link_far(uint32_t i) const { return lk[i]; }
    __device__ uint32_t link(uint32_t i) const {
        uint32_t d = lk[i];
        return d == 0xFFFFu ? 0u : d;
    }
};

// Lanes of ONE wave handing data to each other through LDS: the hardware executes a wave's LDS
// operations in order, so only the compiler has to be kept from moving or forwarding the accesses.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr uint32_t PIECES_MAX = 12;  // (deflate_host.inc ARRIVE_MAX)
// scalars shared between kernels of one encode
struct DevScalars {
    uint32_t T;            // tokens in the stream
    uint32_t nb;           // blocks (T / 31744 + 1)
    uint64_t total_bits;   // raw deflate bits
    uint32_t b0_full;      // block 0 holds 31744 tokens
    uint32_t b0_last_tok;  // its last token
    uint32_t b0_last_pos;  // and where that token starts
    uint32_t q13_hits;
    uint32_t ref_panic;
    uint32_t n_stored, n_fixed, n_dynamic;
    uint32_t adler;
    uint32_t crc;          // CRC-32 of the input (gzip trailer), XOR-accumulated by k_crc_fold
    uint64_t adler_a, adler_b;  // sums of the chunk contributions (k_adler_part)
    uint32_t n_fix[PIECES_MAX];    // segments whose speculative entry did not check out (k_spec_check), per piece
    uint32_t q1_cancel;    // a small call's first pass: block 0 ended inside the first window and the hashes will be re-warmed (Q1) -- the
                           // block stages of this pass have nothing to do (the host runs the call again; run_encode `speculate`)
    // A host call whose input is still arriving works on the stream piece by piece (deflate_host.inc run_streamed): what a
    // piece hands to the next one -- tokens and complete blocks so far; total_bits above is the bit position so far
    uint32_t Tcum[PIECES_MAX + 1];
    uint32_t nbcum[PIECES_MAX + 1];
};
// which part of the stream a launch of the token / block kernels works on (a whole encode: piece 0 of one, `last`)
struct Piece {
    uint32_t seg_lo;  // its first segment; its last is the kernel's K
    uint32_t p;       // its number
    uint32_t last;    // the stream ends with it
};

// ... and what outlives the clearing of the scalars at the start of an encode
struct DevState {
    DevScalars sc;
    uint32_t sort_bad;  // k_match3 met a bucket whose entries do not ascend: k_sort's ranks from LDS atomics cannot be trusted
    uint32_t spec_bad;  // a segment's speculative entry (k_emit<true>) is not where the segment before it was left
    uint32_t pad[2];
};

// (the kernels behind a speculative parse that failed its check have nothing to do: the host parses again)
// (... and so have the block stages of a first pass that will be run again with re-warmed hashes: q1_cancel)
__device__ __forceinline__ bool spec_failed(const DevScalars* sc) { return (reinterpret_cast<const DevState*>(sc)->spec_bad | sc->q1_cancel) != 0; }

constexpr uint32_t SEG = 1024;  // positions per level-0 segment
#ifndef MI355_ADV_STRAIGHT
#define MI355_ADV_STRAIGHT 1  // (k_adv 0.22 -> 0.14 ms: parse 0.577 -> 0.494 ms)
#endif
#ifndef MI355_STEPS_IN_EMIT
#define MI355_STEPS_IN_EMIT 1
#endif
#ifndef MI355_FAN
#define MI355_FAN 16  // (measured: parse 0.789 / 0.770 / 0.767 / 0.778 / 0.811 ms with 4 / 8 / 16 / 32 / 64 -- more, shorter launches win)
#endif
constexpr uint32_t FAN = MI355_FAN;    // children per unit in the table tree

// ---------------------------------------------------------------------------------------------
// k_links_a / k_links_b: chained_hash_table.rs:118-158 (add_hash_value) for every position, as
// "distance to the most recent earlier position with the same hash".  The instruction-heavy part
// (hash, equal-hash lanes inside a 64-position batch) needs no table and runs on the whole chip;
// only the cross-batch part walks a table in order.
// ---------------------------------------------------------------------------------------------
// 4 input bytes at p (zeros past the end), branch free so that the compiler can count the loads it
// has in flight: one unaligned dword load from min(p, n-4), shifted.  Needs n >= 4.
__device__ __forceinline__ uint32_t load_u32_clamped(const uint8_t* in, uint64_t p, uint64_t n) {
    uint64_t q = p + 4 <= n ? p : n - 4;
    uint32_t v;
    __builtin_memcpy(&v, in + q, 4);
    uint32_t sh = (uint32_t)(p - q) * 8;
    return sh >= 32 ? 0u : v >> sh;
}

// Phase A (fully parallel, no table): per aligned batch of 64 positions the hashes, and by 15
// ballots the mask of lanes with the same hash.  A lane with an equal hash below it gets its link
// at once; `last` marks the lane that must publish its position to the table.
__global__ __launch_bounds__(256) void k_links_a(const uint8_t* __restrict__ in, uint32_t n, uint16_t* __restrict__ link,
                                                 uint16_t* __restrict__ hl, HashOverride ov) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (n < 4) {
        if (p < n) {
            link[p] = 0;
            hl[p] = 0;
        }
        return;
    }
    bool active = p + 2 < n;
    uint32_t v = load_u32_clamped(in, p < n ? p : n - 1, n);
    uint32_t a = v & 0xff, b1 = (v >> 8) & 0xff, c = (v >> 16) & 0xff;
    uint32_t ab;
    if (ov.ns | ov.nh) {  // write patterns around a sync flush that leave positions out or file them a byte late
        ab = skewed_ab(ov, p, (p > 0 && p <= n) ? (uint32_t)in[p - 1] : 0u, a, b1);
        active = active && !hash_hole(ov, p);
    } else {
        ab = rewarm_ab(ov, p, a, b1);
    }
    uint32_t h = active ? hash3(ab & 0xff, ab >> 8, c) : 0;
    uint64_t peers = __ballot(active);
#pragma unroll
    for (int b = 0; b < 15; b++) {
        uint64_t bal = __ballot(active && ((h >> b) & 1));
        peers &= ((h >> b) & 1) ? bal : ~bal;
    }
    uint64_t lower = peers & ((1ull << lane) - 1ull);
    uint32_t l = (active && lower) ? lane - (63u - (uint32_t)__builtin_clzll(lower)) : 0u;
    uint32_t last = (active && ((peers >> lane) >> 1) == 0) ? 1u : 0u;
    if (p < n) {
        link[p] = (uint16_t)l;
        hl[p] = (uint16_t)(h | (last << 15));
    }
}

// Phase B: chained_hash_table.rs:148-158 across batches.  One wave owns one 32 KiB epoch and a
// 32768-entry last-occurrence table in LDS (u16 window-relative positions); it replays the previous
// epoch to warm the table, then resolves the lanes phase A left open (no equal hash below them in
// their batch).  Per group of LG batches the table traffic is issued back to back in batch order --
// read(g), write(g), read(g+1), ...: the LDS executes one wave's operations in order, so read(g+1)
// sees write(g), the writes do not depend on the reads, and the wave waits once per group.
#ifndef MI355_LINKS_LG
#define MI355_LINKS_LG 8
#endif
constexpr int LG = MI355_LINKS_LG;

//
// `ident`: the reference's head table starts as head[h] = h (chained_hash_table.rs:64-69), so until
// the first slide an unused bucket sends a chain on to the position numbered like the hash value.
// With true hashes such a candidate (and everything behind it) differs within the first three
// bytes and cannot change a result, so it is left out; after a hash re-warm (HashOverride) two
// positions are filed under hashes unrelated to their bytes, these hops become real candidates,
// and the first two epochs reproduce them.  ident == 2: in every epoch (the table goes back to head[h] = h for
// every bucket whose last entry slides out, chained_hash_table.rs:197-219) -- for the positions that a one-byte
// write after a flush files a byte late (HashOverride::skw), wherever they are in the stream.
__global__ __launch_bounds__(64) void k_links_b(uint32_t n, uint16_t* __restrict__ link,
                                                const uint16_t* __restrict__ hl, uint32_t ident) {
    __shared__ uint16_t head[32768];
    const uint32_t lane = threadIdx.x;
    const uint32_t c0 = blockIdx.x * (uint32_t)WINDOW_SIZE;
    const uint32_t start = c0 >= (uint32_t)WINDOW_SIZE ? c0 - WINDOW_SIZE : 0;
    const uint32_t bias = c0 >= (uint32_t)WINDOW_SIZE ? 0 : WINDOW_SIZE;  // table value of position p: p - start + bias
    const bool id = ident == 2 || (ident && blockIdx.x < 2);
    // (identity: bucket h points at the position h bytes into the buffer, i.e. start + h: table value h + bias)
    for (uint32_t i = lane; i < 32768; i += 64) head[i] = id ? (uint16_t)(i + bias) : (uint16_t)0xFFFF;
    __syncthreads();
    const uint32_t stop = c0 + WINDOW_SIZE < n ? c0 + WINDOW_SIZE : n;
    // One wave, nothing else on its SIMD: every instruction counts.  Both arrays are padded by
    // LINKS_PAD entries, so the loads need no clamp, and they are addressed as base + 32-bit index.
    // hl/link of the groups one and two ahead are in flight while a group is worked on.
    // Groups of LG batches run through four register stages: a group is worked on from its stage
    // and the stage is then refilled with the group four ahead, so loaded values are touched only
    // where they are used (no hand-over copies that would make the wave wait for loads it has just
    // issued).  Running pointers with constant offsets: one 64-bit add per four groups.
    constexpr int ST = 4;
    constexpr uint32_t GP = 64 * LG;  // positions per group
    const uint16_t* ph = hl + start + lane;
    uint16_t* pl = link + start + lane;
    const uint32_t count = stop - start;
    const uint32_t own = c0 - start;  // the previous epoch comes first: it only warms the table
    uint32_t H[ST][LG], K[ST][LG];
#pragma unroll
    for (int r = 0; r < ST; r++)
#pragma unroll
        for (int g = 0; g < LG; g++) H[r][g] = ph[GP * r + 64 * g];
    for (uint32_t i0 = 0; i0 < own; i0 += GP * ST) {  // (an epoch is 64 groups)
#pragma unroll
        for (int r = 0; r < ST; r++) {
            wave_lds_fence();
#pragma unroll
            for (int g = 0; g < LG; g++) {
                const uint32_t i = i0 + GP * r + 64 * g + lane;
                const uint32_t v = H[r][g];
                if (v >> 15) head[v & 0x7fff] = (uint16_t)(i + bias);  // (start + i + 2 < n holds before c0 <= n)
                wave_lds_fence();
            }
#pragma unroll
            for (int g = 0; g < LG; g++) H[r][g] = ph[GP * (ST + r) + 64 * g];
        }
        ph += GP * ST;
    }
    pl += own;
#pragma unroll
    for (int r = 0; r < ST; r++)
#pragma unroll
        for (int g = 0; g < LG; g++) K[r][g] = pl[GP * r + 64 * g];
    for (uint32_t i0 = own; i0 < count; i0 += GP * ST) {
#pragma unroll
        for (int r = 0; r < ST; r++) {
            uint32_t stored[LG], rel[LG];
            bool need[LG];
            wave_lds_fence();
#pragma unroll
            for (int g = 0; g < LG; g++) {
                const uint32_t i = i0 + GP * r + 64 * g + lane;
                const bool active = start + i + 2 < n && i < count;
                const uint32_t v = H[r][g];
                const uint32_t h = v & 0x7fff;
                rel[g] = i + bias;  // 0..65535
                need[g] = active && K[r][g] == 0;
                const bool last = active && (v >> 15);
                stored[g] = need[g] ? (uint32_t)head[h] : 0xFFFFu;
                wave_lds_fence();
                if (last) head[h] = (uint16_t)rel[g];
                wave_lds_fence();
            }
#pragma unroll
            for (int g = 0; g < LG; g++)
                if (need[g] && stored[g] < rel[g] && rel[g] - stored[g] <= WINDOW_SIZE)
                    pl[GP * r + 64 * g] = (uint16_t)(rel[g] - stored[g]);
#pragma unroll
            for (int g = 0; g < LG; g++) {
                H[r][g] = ph[GP * (ST + r) + 64 * g];
                K[r][g] = pl[GP * (ST + r) + 64 * g];
            }
        }
        ph += GP * ST;
        pl += GP * ST;
    }
}
constexpr uint32_t LINKS_PAD = 64 * 8 * LG + 64;  // what k_links_b reads past the last position (stages ahead)

// ---------------------------------------------------------------------------------------------
// k_match: matching.rs:87-166 longest_match (prev_length = 0) for every position.  A workgroup
// stages the 32 KiB history + its 21 KiB tile + 258 lookahead bytes and the links of the same
// range in LDS (159 of 160 KiB, one workgroup of 16 waves per CU); each lane keeps two hash chains
// in flight and takes positions from a counter of the tile (match_walk_park in stages.h).
// ---------------------------------------------------------------------------------------------
#ifndef MI355_MATCH_TILE
#define MI355_MATCH_TILE 21504  // the largest multiple of 1024 whose window + links fit the 160 KiB of LDS
#endif
constexpr uint32_t MT = MI355_MATCH_TILE;                    // most positions per tile (a multiple of 64)
// The tile size of a launch is chosen on the host (match_tile): the largest that fits, shrunk so that
// the tiles come in whole rounds over the CUs, and smaller still for inputs that would not give
// every CU a tile.
#ifndef MI355_MATCH_THREADS
#define MI355_MATCH_THREADS 1024
#endif
constexpr uint32_t MTHREADS = MI355_MATCH_THREADS;
#ifndef MI355_MATCH_U
#define MI355_MATCH_U 2
#endif
#ifndef MI355_MATCH_R
#define MI355_MATCH_R 8
#endif
#ifndef MI355_EXT_ROUNDS
#define MI355_EXT_ROUNDS 100
#endif
#ifndef MI355_EXT_DENSE
#define MI355_EXT_DENSE 1
#endif
constexpr uint32_t MCHAINS = MI355_MATCH_U;                  // chains in flight per lane
constexpr uint32_t MW_BYTES = WINDOW_SIZE + MT + 258 + 14;   // a multiple of 16
constexpr uint32_t MW_LINKS = WINDOW_SIZE + MT;

struct MatchEmit {
    uint32_t* M;
    uint32_t* Mq;
    uint64_t wstart;
    __device__ void operator()(uint32_t idx, uint32_t m, uint32_t mq) const {
        M[wstart + idx] = m;
        if (Mq) Mq[wstart + idx] = mq;
    }
};

// positions of the tile are handed out through one LDS counter, so lanes stay busy until the tile is
// done whatever their chain lengths were
struct TileNext {
    uint32_t* counter;
    uint32_t base, count;
    __device__ uint32_t operator()() {
        uint32_t i = atomicAdd(counter, 1u);
        return i < count ? base + i : (uint32_t)NO_POS;
    }
};
// when to service parked / finished slots: every 8th step, or at once when no lane of the wave can
// walk on (all lanes of a wave iterate in lockstep, so `iter` is uniform)
struct WavePolicy {
#ifdef MI355_MATCH_STATS
    uint32_t* wgmax;
    unsigned long long* wgend;
    unsigned long long* wgsum;
    unsigned long long t0;
#endif
    __device__ bool operator()(bool pending, bool walking, uint32_t iter) const {
        if ((iter % MI355_MATCH_R) == MI355_MATCH_R - 1) return true;
        return __builtin_amdgcn_ballot_w64(walking) == 0;
    }
    // the compare loop is worth its instructions while many lanes take part; a few long matches go
    // on at the next service instead of holding the whole wave
    __device__ bool keep_extending(bool any, uint32_t round) const {
        uint32_t cnt = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(any));
        return cnt >= MI355_EXT_DENSE || (round < MI355_EXT_ROUNDS && cnt > 0);
    }
};

// end of the visible data for a window-relative position (segment end of the absolute position)
struct TileLimit {
    SegEnds sg;
    uint64_t wstart;
    uint32_t one;  // window-relative end of the input when there are no flush points (read once), else 0
    __device__ uint32_t operator()(uint32_t idx) const {
        if (sg.m == 1) return one;
        uint64_t e = seg_end(sg, wstart + idx);
        uint64_t r = e - wstart;
        return r > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)r;
    }
};

template <bool HAS_Q>
__global__ __launch_bounds__(MTHREADS) void k_match(const uint8_t* __restrict__ in, uint32_t n,
                                                    const uint16_t* __restrict__ link, uint32_t* __restrict__ M,
                                                    uint32_t* __restrict__ Mq, uint32_t checks, uint32_t checks_q,
                                                    int in_aligned4, SegEnds sg, uint32_t mt) {
    // one block, bytes first: both arrays then start below 64 KiB and their base folds into the
    // 16-bit offset field of the ds_read instructions
    __shared__ __attribute__((aligned(16))) uint8_t s_win[MW_BYTES + 2 * MW_LINKS];
    uint8_t* const s_bytes = s_win;
    uint16_t* const s_link = reinterpret_cast<uint16_t*>(s_win + MW_BYTES);
    __shared__ uint32_t s_next;
    const uint32_t tid = threadIdx.x;
    const uint64_t E = (uint64_t)blockIdx.x * mt;
    const uint32_t wbytes = WINDOW_SIZE + mt + 258 + 14, wlinks = WINDOW_SIZE + mt;  // multiples of 16 / 8
    const uint64_t wstart = E >= WINDOW_SIZE ? E - WINDOW_SIZE : 0;
    uint32_t* sb32 = reinterpret_cast<uint32_t*>(s_bytes);
    if (in_aligned4) {
        for (uint32_t w = tid; w < wbytes / 16; w += MTHREADS) {
            uint64_t g = wstart + 16ull * w;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (g + 16 <= n) {
                v = *reinterpret_cast<const uint4*>(in + g);  // wstart is a multiple of 16
            } else {
                uint32_t t[4] = {0, 0, 0, 0};
                for (int b = 0; b < 16; b++)
                    if (g + b < n) t[b >> 2] |= (uint32_t)in[g + b] << (8 * (b & 3));
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
            reinterpret_cast<uint4*>(s_bytes)[w] = v;
        }
    } else {
        for (uint32_t w = tid; w < wbytes / 4; w += MTHREADS) {
            uint64_t g = wstart + 4ull * w;
            uint32_t v = 0;
            for (int b = 0; b < 4; b++)
                if (g + b < n) v |= (uint32_t)in[g + b] << (8 * b);
            sb32[w] = v;
        }
    }
    for (uint32_t w = tid; w < wlinks / 8; w += MTHREADS) {
        uint64_t g = wstart + 8ull * w;  // link is 256-byte aligned and wstart a multiple of 8
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g + 8 <= n) {
            v = *reinterpret_cast<const uint4*>(link + g);
        } else {
            uint32_t t[4] = {0, 0, 0, 0};
            for (int b = 0; b < 8; b++)
                if (g + b < n) t[b >> 1] |= (uint32_t)link[g + b] << (16 * (b & 1));
            v = make_uint4(t[0], t[1], t[2], t[3]);
        }
        // 0 (none) -> 0xFFFF per halfword
        uint32_t t4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t lo = t4[q] & 0xffffu, hi = t4[q] >> 16;
            t4[q] = (lo ? lo : 0xffffu) | ((hi ? hi : 0xffffu) << 16);
        }
        reinterpret_cast<uint4*>(s_link)[w] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
    }
    if (tid == 0) s_next = 0;
#ifdef MI355_MATCH_STATS
    __shared__ uint32_t s_wgmax;
    __shared__ unsigned long long s_wgend, s_wgsum;
    if (tid == 0) {
        s_wgmax = 0;
        s_wgend = 0;
        s_wgsum = 0;
    }
    const unsigned long long t_enter = wall_clock64();
#endif
    __syncthreads();
    LdsWin win{s_bytes, s_link};
    MatchEmit emit{M, HAS_Q ? Mq : nullptr, wstart};
    TileNext next{&s_next, (uint32_t)(E - wstart), mt};
#ifdef MI355_MATCH_STATS
    const unsigned long long t_start = wall_clock64();
    WavePolicy pol{&s_wgmax, &s_wgend, &s_wgsum, t_start};
#else
    WavePolicy pol;
#endif
    TileLimit lim{sg, wstart, sg.m == 1 ? (uint32_t)(sg.ends[0] - wstart) : 0u};
    match_walk_park<MCHAINS, HAS_Q>(win, next, lim, checks, checks_q, emit, pol);
#ifdef MI355_MATCH_STATS
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&g_mstats[8], (unsigned long long)s_wgmax);  // the slowest lane of the workgroup
        atomicAdd(&g_mstats[9], 1ull);
        atomicAdd(&g_mstats[10], s_wgend - t_start);   // wall clock (100 MHz) until the last wave ends
        atomicAdd(&g_mstats[11], s_wgsum);             // sum over the 16 waves of their own end times
        atomicAdd(&g_mstats[12], t_start - t_enter);   // staging
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// k_sort + k_match3: the same longest_match table over hash-SORTED positions (stages.h SwG).
//
// k_sort (one workgroup per 32 KiB epoch) stands in for chained_hash_table.rs:118-158 as a whole: it
// files every position of the epoch under its 15-bit hash, in position order -- S_e = the epoch's
// positions sorted by (hash, position), B_e[h] = start of bucket h -- by a histogram (bucket starts)
// and two stable counting-sort passes over the hash (low 8 bits, high 7 bits).  A pass ranks 64 keys
// at a time: the lanes of a wave with the same digit find each other by ballots, the wave's running
// digit offsets live in LDS, and every wave owns a contiguous sixteenth of the keys, so the order of
// equal digits is kept.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t BSTRIDE = WINDOW_SIZE + 8;  // u16 entries per epoch in the bucket-start table (B[32768] = J)
constexpr uint32_t SORT_CHUNK = WINDOW_SIZE / 16;  // keys per wave

__device__ __forceinline__ uint32_t epoch_active(uint32_t n, uint64_t E) {  // positions of the epoch with a hash byte
    const uint64_t act = n >= 2 ? (uint64_t)n - 2 : 0;
    return act > E ? (uint32_t)((act - E) < (uint64_t)WINDOW_SIZE ? (act - E) : (uint64_t)WINDOW_SIZE) : 0u;
}

// exclusive prefix of one value per thread over a 1024-thread workgroup (two barriers inside)
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* red /*16*/, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(x, off, 64);
        if (lane >= (uint32_t)off) x += y;
    }
    __syncthreads();  // (red may still be read from a previous call)
    if (lane == 63) red[wv] = x;
    __syncthreads();
    uint32_t add = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        const uint32_t t = red[k];
        add += k < wv ? t : 0u;
        all += t;
    }
    if (total) *total = all;
    return add + x - v;
}

// lanes of the wave that hold the same NB-bit digit as this lane (among `valid` lanes), as two 32-bit
// halves.  Per bit: the lanes with the bit set (one compare), and "agrees with me" = not (set XOR mine)
// folded into the running mask -- four vector instructions.
template <int NB>
__device__ __forceinline__ void wave_match(uint32_t d, bool valid, uint32_t* plo, uint32_t* phi) {
    const uint64_t all = __builtin_amdgcn_ballot_w64(valid);
    uint32_t lo = (uint32_t)all, hi = (uint32_t)(all >> 32);
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int32_t mine = __builtin_amdgcn_sbfe((int32_t)d, (uint32_t)b, 1u);  // 0 or -1
        const uint64_t bal = __builtin_amdgcn_ballot_w64(mine != 0);
        lo &= ~((uint32_t)bal ^ (uint32_t)mine);
        hi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)mine);
    }
    *plo = lo;
    *phi = hi;
}

#ifdef MI355_MATCH_STATS
#define KS_T(i)                                                                        \
    if (threadIdx.x == 0) {                                                            \
        unsigned long long t_ = __builtin_readcyclecounter();                          \
        atomicAdd(&g_sstats[i], t_ - ks_t);                                            \
        ks_t = t_;                                                                     \
    }
#else
#define KS_T(i)
#endif
// one stable counting-sort pass of the workgroup: key i (i < J) has digit dig(i) < 2^NB and payload
// pay(i); put(dest, payload) stores it at its rank
template <int NB, class Dig, class Pay, class Put>
__device__ __forceinline__ void sort_pass(uint32_t J, uint32_t* cnt /*16 * 256*/, uint32_t* red, Dig dig, Pay pay, Put put,
                                          unsigned long long& ks_t, int stat0) {
    constexpr uint32_t ND = 1u << NB;
    constexpr int NBAT = SORT_CHUNK / 64;  // batches per wave
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* mine = cnt + wv * 256;
    for (uint32_t k = lane; k < 256; k += 64) mine[k] = 0;
    wave_lds_fence();
    const uint32_t cb = wv * SORT_CHUNK;
    // the digits of the wave's keys stay in registers (four per register) between the count and the scatter:
    // the second pass reads them through an indirection it then does not have to repeat
    uint32_t dc[NBAT / 4];
    // ... and so do a key's rank among the wave's keys of the same digit and whether it is their last (four
    // keys per register): the lanes of a digit are found once, here, and the digit's first lane counts for all of them
    // -- 64 atomics on one counter (a hot digit; a run of one byte) would be served one after the other
    uint32_t bt[NBAT / 4];
    uint32_t Jc = J;
    asm volatile("" : "+s"(Jc));  // (a copy the compiler cannot match with the other pass's: it kept all 32 `i < J` of a pass alive as 0/1 registers, spilling them)
#pragma unroll
    for (int b = 0; b < NBAT; b++) {
        const uint32_t i = cb + 64 * b + lane;
        const bool valid = i < Jc;
        const uint32_t d = valid ? dig(i) : 0u;
        uint32_t plo, phi;
        wave_match<NB>(d, valid, &plo, &phi);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        const uint32_t total = (uint32_t)__builtin_popcount(plo) + (uint32_t)__builtin_popcount(phi);
        if (valid && below == 0) atomicAdd(&mine[d], total);
        if ((b & 3) == 0)
            dc[b >> 2] = d;
        else
            dc[b >> 2] |= d << (8 * (b & 3));
        const uint32_t pk = below | (below + 1 == total ? 64u : 0u) | (valid ? 128u : 0u);  // below < 64
        if ((b & 3) == 0)
            bt[b >> 2] = pk;
        else
            bt[b >> 2] |= pk << (8 * (b & 3));
        if ((b & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // (keeps the 32 batches' digits from being fetched all at once)
    }
    __syncthreads();
    KS_T(stat0)
    // offsets in (digit, wave) order: thread t takes digit t / 4, waves 4 * (t % 4) ..+3
    uint32_t v[4] = {0, 0, 0, 0}, sum = 0;
    const uint32_t d4 = tid >> 2, w4 = (tid & 3) * 4;
    if (d4 < ND) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = cnt[(w4 + k) * 256 + d4];
            sum += v[k];
        }
    }
    uint32_t base = block_excl_scan_1024(sum, red, nullptr);
    if (d4 < ND) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            cnt[(w4 + k) * 256 + d4] = base;
            base += v[k];
        }
    }
    __syncthreads();
    KS_T(stat0 + 1)
#pragma unroll
    for (int b = 0; b < NBAT; b++) {
        const uint32_t i = cb + 64 * b + lane;
        const uint32_t d = (dc[b >> 2] >> (8 * (b & 3))) & 0xffu;
        const uint32_t pk = (bt[b >> 2] >> (8 * (b & 3))) & 0xffu;
        const bool valid = (pk & 128u) != 0;  // (not `i < J` again: the compiler would keep 32 of those alive from the count)
        const uint32_t py = valid ? pay(i) : 0u;
        const uint32_t below = pk & 63u;
        uint32_t at = 0;
        if (valid) at = mine[d];
        wave_lds_fence();  // every lane has read its offset before the last lane of each digit moves it on
        if (valid) {
            put(at + below, py);
            if (pk & 64u) mine[d] = at + below + 1;  // the digit's last lane moves the offset on
        }
        wave_lds_fence();
    }
    __syncthreads();
    KS_T(stat0 + 2)
}

#ifndef MI355_SORT_HOT
#define MI355_SORT_HOT 4     // lanes on the first lane's digit from which they are ranked by lane order instead of queueing on the counter
#endif
#ifndef MI355_SORT_ROUNDS
#define MI355_SORT_ROUNDS 1  // (measured on the 100 MB text: k_sort 0.440 ms without, 0.427 with one digit from 4 / 8 lanes on, 0.43 from 16;
#endif                       //  two / three digits that way 0.47 / 0.50 -- every round is a dependent read and write of the counter)
// (... in the pass over the high digit, and in the pass over the LOW digit only for an epoch with runs of one byte -- zero fill: all
// of a batch on one counter, 0.98 against 1.68 ms for 256 MiB -- : its batches are keys in position order and meet on no counter in
// particular otherwise, where the look for a first digit only cost: 0.427 -> 0.413 ms under the profiler on the text)
#ifndef MI355_SORT_SB
#define MI355_SORT_SB 4  // batches whose digits are fetched together in the count phase of sort_pass_rtn
#endif
// The same pass with the rank of a key taken from the LDS itself: ds_add_rtn on the (wave, digit) counter returns how
// many of the wave's keys with that digit came before -- batches are instructions in program order, and within one
// instruction the LDS serves the lanes that hit one address in lane order.  That order is what the hardware does, not
// what the ISA manual promises: k_lds_order_test checks it on the device when a context is made, and a device that fails
// it sorts with sort_pass (ranks from ballots: 8 + 7 ballots per batch of 64 keys were half of k_sort's instructions).
// A batch whose keys all share one digit (a run of one byte) takes its ranks from the lane number: 64 atomics on one
// counter would be served one after the other.
template <int NB, bool LOW, class Dig, class Pay, class Put>
__device__ __forceinline__ void sort_pass_rtn(uint32_t J, uint32_t* cnt /*16 * 256*/, uint32_t* red, Dig dig, Pay pay, Put put,
                                              unsigned long long& ks_t, int stat0, bool hot_low) {
    constexpr uint32_t ND = 1u << NB;
    constexpr int NBAT = SORT_CHUNK / 64;  // batches per wave
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* mine = cnt + wv * 256;
    for (uint32_t k = lane; k < 256; k += 64) mine[k] = 0;
    wave_lds_fence();
    const uint32_t cb = wv * SORT_CHUNK;
    uint32_t dc[NBAT / 4];  // digits, four per register
    uint32_t rk[NBAT / 2];  // rank among the wave's keys of the digit (< 2048) | valid << 15, two per register
    uint32_t Jc = J;
    asm volatile("" : "+s"(Jc));  // (see sort_pass)
#pragma unroll
    for (int b = 0; b < NBAT; b++) {
        const uint32_t i = cb + 64 * b + lane;
        const bool valid = i < Jc;
        const uint32_t d = valid ? dig(i) : 0u;
        const uint64_t vm = __builtin_amdgcn_ballot_w64(valid);
        uint32_t r = 0;
        // The lanes that share the FIRST lane's digit take their ranks from the lane order (one read and one write of the
        // counter for all of them); only the others queue on their counters.  In the second pass a batch's keys share their
        // low digit, text has few trigrams per low digit, and half a batch meets on one counter: served one lane after the other.
        // (MI355_SORT_ROUNDS digits that way, each the digit of the first lane not yet ranked)
        uint64_t left = vm;
        bool mine_done = !valid;
#pragma unroll
        for (int round = 0; round < MI355_SORT_ROUNDS; round++) {
            if (LOW && !hot_low) break;
            if (left == 0) break;
            const uint32_t first = (uint32_t)__builtin_ctzll(left);
            const uint32_t dk = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)first);
            const uint64_t mk = __builtin_amdgcn_ballot_w64(!mine_done && d == dk);
            const uint32_t nk = (uint32_t)__popcll(mk);
            if (nk < MI355_SORT_HOT) break;
            const uint32_t at = mine[dk];
            wave_lds_fence();
            const bool hot = !mine_done && d == dk;
            if (hot) r = at + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (lane == first) mine[dk] = at + nk;
            wave_lds_fence();
            mine_done = mine_done || hot;
            left &= ~mk;
        }
        if (!mine_done) r = atomicAdd(&mine[d], 1u);
        if ((b & 3) == 0)
            dc[b >> 2] = d;
        else
            dc[b >> 2] |= d << (8 * (b & 3));
        const uint32_t pk = r | (valid ? 0x8000u : 0u);
        if ((b & 1) == 0)
            rk[b >> 1] = pk;
        else
            rk[b >> 1] |= pk << 16;
        if ((b % MI355_SORT_SB) == MI355_SORT_SB - 1) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    KS_T(stat0)
    // offsets in (digit, wave) order: thread t takes digit t / 4, waves 4 * (t % 4) ..+3
    uint32_t v[4] = {0, 0, 0, 0}, sum = 0;
    const uint32_t d4 = tid >> 2, w4 = (tid & 3) * 4;
    if (d4 < ND) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = cnt[(w4 + k) * 256 + d4];
            sum += v[k];
        }
    }
    uint32_t base = block_excl_scan_1024(sum, red, nullptr);
    if (d4 < ND) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            cnt[(w4 + k) * 256 + d4] = base;
            base += v[k];
        }
    }
    __syncthreads();
    KS_T(stat0 + 1)
#pragma unroll
    for (int b = 0; b < NBAT; b++) {
        const uint32_t i = cb + 64 * b + lane;
        const uint32_t d = (dc[b >> 2] >> (8 * (b & 3))) & 0xffu;
        const uint32_t pk = (rk[b >> 1] >> (16 * (b & 1))) & 0xffffu;
        if (pk & 0x8000u) put(mine[d] + (pk & 0x7fffu), pay(i));
    }
    __syncthreads();
    KS_T(stat0 + 2)
}

// Does the LDS serve the lanes of one ds_add_rtn that hit the same address in lane order?  (sort_pass_rtn)  Sixteen waves of
// a workgroup hammer their counters with digit patterns of every kind at once -- one digit, two, few, many, random -- and
// compare what comes back with the rank counted from ballots.  *bad counts the lanes that differ.
__global__ __launch_bounds__(1024) void k_lds_order_test(uint32_t rounds, uint32_t* bad) {
    __shared__ uint32_t sCnt[16 * 256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* mine = sCnt + wv * 256;
    for (uint32_t k = lane; k < 256; k += 64) mine[k] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * 1024 + tid) * 2654435761u + 12345u, wrong = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t kind = (r + wv) & 7u;
        const uint32_t mask = kind == 0 ? 0u : kind == 1 ? 1u : kind == 2 ? 3u : kind == 3 ? 7u : kind == 4 ? 15u : kind == 5 ? 63u : 255u;
        uint32_t d = (x >> 13) & mask;
        if (kind == 7) d = (lane >> 2) & 255u;  // runs of four
        const bool valid = ((x >> 5) & 15u) != 0 || kind < 2;
        uint32_t plo, phi;
        wave_match<8>(d, valid, &plo, &phi);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        uint32_t before = 0;
        if (valid) before = mine[d];
        wave_lds_fence();
        uint32_t got = 0;
        if (valid) got = atomicAdd(&mine[d], 1u);
        wave_lds_fence();
        if (valid && got != before + below) wrong++;
    }
    if (wrong) atomicAdd(bad, wrong);
}

// MODE 0: ranks from ballots (sort_pass), 1: from returning atomics (sort_pass_rtn).  (A third form -- ONE pass in
// position order with the bucket starts as cursors, the sixteen waves taking turns -- was measured at 0.51 ms against
// 0.45 ms and is in the history of this file: DESIGN.md section 5.)
#ifndef MI355_SWZ_BANKS
#define MI355_SWZ_BANKS 14  // an epoch of whose sampled neighbours 64 sit on this many LDS banks or fewer is walked with the permuted
                            // pair table (k_match3_swz); 0: never
#endif
#ifndef MI355_SWZ_PERSISTENT
#define MI355_SWZ_PERSISTENT 0  // k_match3_swz: 1 = a workgroup per compute unit takes the marked epochs in turn, 0 = a workgroup per epoch
#endif
#ifndef MI355_SORT_P1
#define MI355_SORT_P1 8
#endif
// What the first kernel of a small call does on the side (run_encode): the call's scalars cleared and the one segment end set,
// instead of two fills of the runtime in front of it.
struct SortInit {
    uint32_t* sc;
    uint32_t sc_words;
    uint32_t* seg_end;
    uint32_t seg_val;
};
template <int MODE>
__global__ __launch_bounds__(1024) void k_sort(const uint8_t* __restrict__ in, uint32_t n, HashOverride ov,
                                               uint16_t* __restrict__ Sg, uint16_t* __restrict__ Bg, uint32_t e0, uint32_t dbl,
                                               SortInit init) {
    __shared__ __attribute__((aligned(16))) uint16_t sH[WINDOW_SIZE];  // hashes; the sorted array at the end
    __shared__ __attribute__((aligned(16))) uint32_t sBuf[WINDOW_SIZE / 2];  // histogram (u16 pairs), then pass-1 output (u16) / cursors
    __shared__ uint32_t sCnt[16 * 256];
    __shared__ uint32_t sRed[16];
    __shared__ uint32_t s_runs;  // pieces of 512 positions that lie in a run of one byte
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_runs = 0;
    if (init.sc && blockIdx.x == 0) {
        for (uint32_t i = tid; i < init.sc_words; i += 1024) init.sc[i] = 0;
        if (tid == 0 && init.seg_end) *init.seg_end = init.seg_val;
    }
    const uint32_t e = e0 + blockIdx.x;
    const uint64_t E = (uint64_t)e * WINDOW_SIZE;
    const uint32_t J = epoch_active(n, E);
    unsigned long long ks_t = __builtin_readcyclecounter();
    (void)ks_t;
    for (uint32_t k = tid; k < WINDOW_SIZE / 2; k += 1024) sBuf[k] = 0;
    __syncthreads();
    // hashes (chained_hash_table.rs:55-62) and their histogram.  A full epoch of a 16-byte aligned input: a thread takes
    // eight consecutive positions from two aligned dwords and the one behind them (one byte-granular dword load per
    // position kept the address unit busy for a quarter of the kernel); else eight positions per thread and round,
    // one clamped load each.
    const bool whole = J == WINDOW_SIZE && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && E + WINDOW_SIZE + 4 <= (uint64_t)n &&
                       !(ov.on | ov.m);
    if (whole) {
        // (the four rounds' bytes fetched before the first is worked on: with a load at the head of each round -- the LDS
        // atomics keep the compiler from moving it up -- a thread waited for memory four times in a row)
        uint2 wr[WINDOW_SIZE / (8 * 1024)];
        uint32_t nr[WINDOW_SIZE / (8 * 1024)];
#pragma unroll
        for (uint32_t r = 0; r < WINDOW_SIZE / (8 * 1024); r++) {
            const uint32_t i = (r * 1024 + tid) * 8;
            wr[r] = *reinterpret_cast<const uint2*>(in + E + i);
            nr[r] = *reinterpret_cast<const uint32_t*>(in + E + i + 8);
        }
#pragma unroll
        for (uint32_t r = 0; r < WINDOW_SIZE / (8 * 1024); r++) {
            const uint32_t i = (r * 1024 + tid) * 8;
            const uint2 w = wr[r];
            const uint32_t nx = nr[r];
            const uint32_t d[3] = {w.x, w.y, nx};
            uint32_t hs[8];
            // (a wave whose 512 positions lie in a run of one byte counts them with one add: 64 lanes on one counter,
            // eight times over, are served one after the other -- zero-filled input spent half of the kernel here)
            const bool run1 = w.x == w.y && w.y == nx && w.x == __builtin_amdgcn_alignbyte(w.x, w.x, 1u);
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.x);
            const bool wave_run = __builtin_amdgcn_ballot_w64(!run1 || w.x != w0) == 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t v = (k & 3) ? __builtin_amdgcn_alignbyte(d[(k >> 2) + 1], d[k >> 2], (uint32_t)(k & 3)) : d[k >> 2];
                hs[k] = hash3(v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff);
                if (!wave_run) atomicAdd(&sBuf[hs[k] >> 1], (hs[k] & 1) ? 0x10000u : 1u);
            }
            *reinterpret_cast<uint4*>(&sH[i]) = make_uint4(hs[0] | (hs[1] << 16), hs[2] | (hs[3] << 16), hs[4] | (hs[5] << 16), hs[6] | (hs[7] << 16));
            if (wave_run && (tid & 63) == 0) {
                atomicAdd(&sBuf[hs[0] >> 1], (hs[0] & 1) ? 512u << 16 : 512u);
                atomicAdd(&s_runs, 1u);
            }
        }
    } else
    for (uint32_t i0 = 0; i0 < J; i0 += 8 * 1024) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = i0 + 1024 * k + tid;
            const uint64_t p = E + (i < J ? i : 0u);
            if (n >= 4) {
                v[k] = load_u32_clamped(in, p, n);
            } else {
                v[k] = (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16);  // (J > 0: n == 3, p == 0)
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = i0 + 1024 * k + tid;
            if (i < J) {
                const uint32_t ab = rewarm_ab(ov, E + i, v[k] & 0xff, (v[k] >> 8) & 0xff);
                const uint32_t h = hash3(ab & 0xff, ab >> 8, (v[k] >> 16) & 0xff);
                sH[i] = (uint16_t)h;
                atomicAdd(&sBuf[h >> 1], (h & 1) ? 0x10000u : 1u);
            }
        }
    }
    __syncthreads();
    KS_T(0)
    {   // bucket starts: thread t owns bins 32 t .. 32 t + 31
        uint32_t wds[16], sum = 0;
        const uint4* src = reinterpret_cast<const uint4*>(sBuf + tid * 16);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 x = src[q];
            wds[4 * q] = x.x;
            wds[4 * q + 1] = x.y;
            wds[4 * q + 2] = x.z;
            wds[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) sum += (wds[q] & 0xffffu) + (wds[q] >> 16);
        uint32_t run = block_excl_scan_1024(sum, sRed, nullptr);
        uint32_t outw[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint32_t lo = run;
            run += wds[q] & 0xffffu;
            const uint32_t hi = run;
            run += wds[q] >> 16;
            outw[q] = lo | (hi << 16);
        }
        if (tid == 1023) {
            Bg[(size_t)e * BSTRIDE + WINDOW_SIZE] = (uint16_t)J;
            // (k_match3 walks such an epoch with the one-sided long compare: more than half of it in runs of one byte)
            Bg[(size_t)e * BSTRIDE + WINDOW_SIZE + 1] = s_runs > WINDOW_SIZE / 512 / 2 ? 1 : 0;
        }
        // (back into their LDS words: they leave as whole lines below -- a thread's own sixteen words were 64-byte pieces
        // 64 bytes apart: 446 -> 435 us)
        uint4* cur = reinterpret_cast<uint4*>(sBuf + tid * 16);
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = make_uint4(outw[4 * q], outw[4 * q + 1], outw[4 * q + 2], outw[4 * q + 3]);
    }
    __syncthreads();
    {
        uint4* dst = reinterpret_cast<uint4*>(Bg + (size_t)e * BSTRIDE);
        const uint4* src = reinterpret_cast<const uint4*>(sBuf);
#pragma unroll
        for (uint32_t k = 0; k < WINDOW_SIZE / 8 / 1024; k++) dst[k * 1024 + tid] = src[k * 1024 + tid];
    }
    KS_T(1)
    uint16_t* buf16 = reinterpret_cast<uint16_t*>(sBuf);
    if (MODE == 1) {
        constexpr int P1 = MI355_SORT_P1;  // bits of the first digit (the per-wave counter tables hold 256: 7 or 8)
        const bool runs_here = __builtin_amdgcn_readfirstlane((int)s_runs) != 0;  // (final since the barrier behind the hashes)
        sort_pass_rtn<P1, true>(
            J, sCnt, sRed, [&](uint32_t i) { return (uint32_t)sH[i] & ((1u << P1) - 1u); }, [&](uint32_t i) { return i; },
            [&](uint32_t at, uint32_t v) { buf16[at] = (uint16_t)v; }, ks_t, 2, runs_here);
        sort_pass_rtn<15 - P1, false>(
            J, sCnt, sRed, [&](uint32_t i) { return (uint32_t)sH[buf16[i]] >> P1; }, [&](uint32_t i) { return (uint32_t)buf16[i]; },
            [&](uint32_t at, uint32_t v) { sH[at] = (uint16_t)v; }, ks_t, 5, false);
    } else {
    sort_pass<8>(
        J, sCnt, sRed, [&](uint32_t i) { return (uint32_t)sH[i] & 255u; }, [&](uint32_t i) { return i; },
        [&](uint32_t at, uint32_t v) { buf16[at] = (uint16_t)v; }, ks_t, 2);
    // (the second pass has read every hash it needs -- the digits sit in registers -- before its first store,
    // a workgroup barrier lies in between: the sorted array can take the place of the hashes)
    sort_pass<7>(
        J, sCnt, sRed, [&](uint32_t i) { return (uint32_t)sH[buf16[i]] >> 8; }, [&](uint32_t i) { return (uint32_t)buf16[i]; },
        [&](uint32_t at, uint32_t v) { sH[at] = (uint16_t)v; }, ks_t, 5);
    }
#ifdef MI355_DEBUG_HOOKS
    // test build: what a device would do whose LDS served the lanes of an atomic out of order -- two neighbours of one
    // bucket change places (dbl bit 1; mi355_debug_break_sort)
    if ((dbl & 2u) && tid == 0) {
        for (uint32_t i = 1; i < J; i++) {
            const uint32_t a = sH[i - 1], b = sH[i];
            const uint8_t* pa = in + E + a;
            const uint8_t* pb = in + E + b;
            if (E + b + 2 < n && hash3(pa[0], pa[1], pa[2]) == hash3(pb[0], pb[1], pb[2])) {
                sH[i - 1] = (uint16_t)b;
                sH[i] = (uint16_t)a;
                break;
            }
        }
    }
    __syncthreads();
    dbl &= 1u;
#endif
    // Do the epoch's sorted positions sit on few LDS banks?  Four times 64 neighbours of the sorted array -- the lanes of four of
    // the walk's batches -- and the banks their pair-table entries fall in (address = 2 * position: bits 2..7): rows of records
    // put them on sixteen, four, or one (PairWinT<true>); two such samples, or one on half as many, mark the epoch for k_match3_swz.
    if (MI355_SWZ_BANKS) {
        __syncthreads();
        if (tid < 8) sRed[tid] = 0;
        __syncthreads();
        if (tid < 256 && J >= 1024) {  // at 1/8, 3/8, 5/8, 7/8 of the sorted array
            const uint32_t q = tid >> 6;
            const uint32_t bank = ((uint32_t)sH[(2 * q + 1) * (J / 8) + (tid & 63)] >> 1) & 63u;
            atomicOr(&sRed[2 * q + (bank >> 5)], 1u << (bank & 31));
        }
        __syncthreads();
        if (tid == 0) {
            // (two samples on few banks, or one on very few: an epoch that is left to k_match3 among marked ones is a workgroup
            // that runs a millisecond by itself while the other kernel waits)
            uint32_t few = 0, very = 0;
            for (int q = 0; q < 4; q++) {
                const uint32_t nb = (uint32_t)(__popc(sRed[2 * q]) + __popc(sRed[2 * q + 1]));
                few += nb <= MI355_SWZ_BANKS ? 1u : 0u;
                very += nb <= MI355_SWZ_BANKS / 2 ? 1u : 0u;
            }
            Bg[(size_t)e * BSTRIDE + WINDOW_SIZE + 2] = (J >= 1024 && (few >= 2 || very >= 1) && dbl) ? 1 : 0;
        }
    }
    uint4* out = reinterpret_cast<uint4*>(Sg + (size_t)e * WINDOW_SIZE);
    const uint4* fin = reinterpret_cast<const uint4*>(sH);
    // dbl = 1: entries as 2 * position (k_match3 adds them to a pair-table address); positions are below 32768,
    // so the shift of a whole word moves nothing across its halves
    for (uint32_t k = tid; k < (J + 7) / 8; k += 1024) {
        const uint4 v = fin[k];
        out[k] = make_uint4(v.x << dbl, v.y << dbl, v.z << dbl, v.w << dbl);
    }
}

#ifdef MI355_MATCH_STATS
// (MI355_MATCH_STATS = 1: the clocks and the number of batches; 2: the counters.  Together -- and as sixteen 64-bit values --
// they took the walk's scalar registers, the instrumented kernel spilled 60 bytes a lane and its clock shares were those of
// another kernel.  32 bits hold a wave's totals, and differences survive the wrap.)
#define M2_CNT(i, v) \
    if (((MI355_MATCH_STATS) & 2) || (i) == 0 || (i) == 4) m2c[i] += (uint32_t)(v);
#define M2_T0 uint32_t m2t = ((MI355_MATCH_STATS) & 1) ? (uint32_t)__builtin_readcyclecounter() : 0u;
#define M2_T(i)                                                      \
    if ((MI355_MATCH_STATS) & 1) {                                   \
        const uint32_t t_ = (uint32_t)__builtin_readcyclecounter();  \
        m2c[i] += t_ - m2t;                                          \
        m2t = t_;                                                    \
    }
#else
#define M2_CNT(i, v)
#define M2_T0
#define M2_T(i)
#endif

// ---------------------------------------------------------------------------------------------
// k_match3: matching.rs:87-166 for the positions of one epoch, taken 64 at a time in the order of S_e (so the
// lanes of a wave are neighbours in a hash bucket: their candidate lists are the same array shifted by one and
// their reads of it coalesce), with the window held in LDS as a table of byte PAIRS (stages.h SwG): the probe of
// a candidate is ONE aligned two-byte read, and the step block asks about eight candidates at a time: eight
// address adds, eight reads in flight, eight compares that shrink EXEC, and the bookkeeping (entry offset,
// window, entries left) once per group instead of once per step.  The table is 2 B per position of the
// previous and the own epoch: 131.6 KB, one workgroup of 16 waves per CU.  (The byte-image form of round 2,
// k_match2, and the forms that were measured and lost -- k_match4, k_match5, k_match_coop, refill -- are in
// the history of this file and in DESIGN.md section 5.)
// ---------------------------------------------------------------------------------------------
constexpr uint32_t M3T = 1024;
constexpr uint32_t M3_PAIRS = 2 * WINDOW_SIZE + 258 + 14;  // a multiple of 16: pairs in the table (= bytes staged)

// SWZ: the table with its 8-byte words permuted inside every 256 bytes -- word index XOR bits 8..12 of the address.  Rows of
// records put the lanes of a wave (neighbours in a hash bucket = one row apart) on addresses a multiple of the row length
// apart: 192 bytes between lanes is four of the LDS's 64 banks, 512 bytes one -- 0.84 of the LDS's cycles on such data were
// bank conflicts, at 0.86 of the LDS busy and 0.35 of the vector ALU (tools/probes/lds_conflicts.sh).  Every address the walk
// computes stays the plain one; only what is handed to the LDS is permuted (three vector instructions an address, which is
// why a workgroup takes this form only for an epoch whose sorted positions sit on few banks: k_sort's mark).  The permutation
// itself is stages.h m3_swz (pinned on the CPU: tests/test_stages_vs_oracle.py).
template <bool SWZ>
struct PairWinT {
    const uint16_t* sb;  // global: index 0 = entry 0 of the previous epoch's sorted array (entries are 2 * position)
    uint32_t tbase;      // LDS address of T[0] (a multiple of 256)
    typedef __attribute__((address_space(3))) const uint16_t* lds_u16;
    enum : uint32_t { SH = 1 };
    __device__ uint32_t key_at(uint32_t a) const { return *(lds_u16)(SWZ ? m3_swz(a) : a); }
    // 16 bytes from position pos on: the 40 aligned table bytes that hold T[pos & ~3 ...] are twenty pairs, every
    // other one of them -- the low half of each dword -- new bytes; five 8-byte reads (the 64-bank form), the
    // halves packed by byte selects, then the shift by pos & 3
    __device__ void load16(uint32_t pos, uint32_t* q) const {
        // (written out: the compiler pairs neighbouring 8-byte reads into ds_read2_b64, which takes four times the
        // LDS cycles of two ds_read_b64 -- MI355X_MICROARCH.md, LDS table)
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2 d0, d1, d2, d3, d4;
        const uint32_t a8 = (tbase + 2 * pos) & ~7u;
        if (SWZ) {  // (five words, each where the permutation puts it: they may lie in two 256-byte blocks)
            asm volatile(
                "ds_read_b64 %0, %5\n\t"
                "ds_read_b64 %1, %6\n\t"
                "ds_read_b64 %2, %7\n\t"
                "ds_read_b64 %3, %8\n\t"
                "ds_read_b64 %4, %9\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4)
                : "v"(m3_swz(a8)), "v"(m3_swz(a8 + 8)), "v"(m3_swz(a8 + 16)), "v"(m3_swz(a8 + 24)), "v"(m3_swz(a8 + 32))
                : "memory");
        } else
        asm volatile(
            "ds_read_b64 %0, %5\n\t"
            "ds_read_b64 %1, %5 offset:8\n\t"
            "ds_read_b64 %2, %5 offset:16\n\t"
            "ds_read_b64 %3, %5 offset:24\n\t"
            "ds_read_b64 %4, %5 offset:32\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4)
            : "v"(a8)
            : "memory");
        const uint32_t w0 = __builtin_amdgcn_perm(d0.y, d0.x, 0x05040100u), w1 = __builtin_amdgcn_perm(d1.y, d1.x, 0x05040100u),
                       w2 = __builtin_amdgcn_perm(d2.y, d2.x, 0x05040100u), w3 = __builtin_amdgcn_perm(d3.y, d3.x, 0x05040100u),
                       w4 = __builtin_amdgcn_perm(d4.y, d4.x, 0x05040100u);
        q[0] = __builtin_amdgcn_alignbyte(w1, w0, pos);
        q[1] = __builtin_amdgcn_alignbyte(w2, w1, pos);
        q[2] = __builtin_amdgcn_alignbyte(w3, w2, pos);
        q[3] = __builtin_amdgcn_alignbyte(w4, w3, pos);
    }
    __device__ uint32_t load32(uint32_t pos) const {
        if (SWZ) return key_at(tbase + 2 * pos) | (key_at(tbase + 2 * pos + 4) << 16);
        lds_u16 t = (lds_u16)(tbase + 2 * pos);
        return (uint32_t)t[0] | ((uint32_t)t[2] << 16);
    }
    __device__ uint32_t sidx(uint32_t i) const { return sb[(int64_t)(int32_t)i]; }
};
typedef PairWinT<false> PairWin;

// Where the lanes of `dropped` stopped in their last group of eight steps: at the first probe that equals their key
// (the compares of the step block shrank EXEC there) -> *any, the probe's address and how far offb is past its
// entry; the other lanes left at the end of the group.
// (probe keys and addresses BY VALUE: selects among the fields of a struct behind a reference become a load through a
// selected pointer, and the struct then lives in scratch memory)
struct MsGroup {
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7, a0, a1, a2, a3, a4, a5, a6, a7, probe;
};
__device__ __forceinline__ void ms_decode(const MsGroup st, uint64_t* any_out, uint32_t* asel_out, uint32_t* back_out) {
    // The first probe that equals the key wins: the probes are looked at from the last to the first and every hit overrides what
    // the later ones left -- eight compares, sixteen selects and no lane-mask arithmetic at all.  (Rounds 3-4 built one-hot "first hit" masks with a chain of
    // scalar and-nots and selected through them: fourteen selects, seventeen scalar instructions.  Scalar instructions of a wave are
    // latency of that wave: without them 3.60 -> 3.54 ms, Best 5.54 -> 5.44; a tree of ten selects with nine scalar ORs MORE was slower,
    // 3.63.  DESIGN.md section 5.)
    uint32_t asel_o = st.a0, back_o = 0;  // (back = 0: no hit; a hit at probe i: 16 - 2 i)
#define MS_STEP(T, A, OFF)                 \
    {                                      \
        const bool h_ = st.T == st.probe;  \
        asel_o = h_ ? st.A : asel_o;          \
        back_o = h_ ? OFF : back_o;          \
    }
    MS_STEP(t7, a7, 2u) MS_STEP(t6, a6, 4u) MS_STEP(t5, a5, 6u) MS_STEP(t4, a4, 8u)
    MS_STEP(t3, a3, 10u) MS_STEP(t2, a2, 12u) MS_STEP(t1, a1, 14u) MS_STEP(t0, a0, 16u)
#undef MS_STEP
    *any_out = __builtin_amdgcn_ballot_w64(back_o != 0);
    *asel_out = asel_o;
    *back_out = back_o ? back_o : 16u;  // (a lane without a hit left at the end of its group)
}

// ---------------------------------------------------------------------------------------------
// Two batches per wave: the pair table leaves one workgroup of 16
// waves per CU -- four per SIMD, with more than half of the 128 vector registers each may use idle -- and at
// four waves the kernel is bound by the latency of its dependent steps (table read -> compare -> next group;
// entry load -> first group), not by any unit: vector ALU 72 %, LDS 50 %, TA 52 % busy.  So a wave walks TWO
// batches of 64 positions at once, fibre x and fibre y, each with its own SwG state: the step block issues the
// entry loads and the eight table reads of both fibres before it looks at the first answer (16 reads in flight
// per wave), the services and set-ups of the two follow each other.  Twice the work per latency.
// ---------------------------------------------------------------------------------------------
#define MF_ADD(F, A, REG, HALF) \
    "v_add_u32_sdwa %[" F A "], " REG ", %[" F "bb] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" HALF " src1_sel:DWORD\n\t"
#ifndef MI355_STEP_WAITS
#define MI355_STEP_WAITS 2  // waits per group of eight answers: 8 = one in front of every compare, 2 = after four and after eight, 1 = one for
                            // all (a wait is an instruction of the wave: 3.50 / 3.485 / 3.478 / 3.484 ms with 8 / 4 / 2 / 1)
#endif
#define MF_CMPW(F, N, T) "s_waitcnt lgkmcnt(" N ")\n\tv_cmpx_ne_u32_e32 vcc, %[" F T "], %[" F "probe]\n\t"
#define MF_CMPN(F, N, T) "v_cmpx_ne_u32_e32 vcc, %[" F T "], %[" F "probe]\n\t"
#define MF_CMP(F, N, T) MF_CMPW(F, N, T)
#define MF_ADDS8(F, R0, R1, R2, R3)                                                                                  \
    MF_ADD(F, "a0", R3, "WORD_1") MF_ADD(F, "a1", R3, "WORD_0") MF_ADD(F, "a2", R2, "WORD_1") MF_ADD(F, "a3", R2, "WORD_0") \
    MF_ADD(F, "a4", R1, "WORD_1") MF_ADD(F, "a5", R1, "WORD_0") MF_ADD(F, "a6", R0, "WORD_1") MF_ADD(F, "a7", R0, "WORD_0")
// the eight reads: plain, or each at its permuted address (PairWinT<true>; worked out in the answer's own register)
#define MF_RD8_0(F)                                                                                                  \
    "ds_read_u16 %[" F "t0], %[" F "a0]\n\tds_read_u16 %[" F "t1], %[" F "a1]\n\tds_read_u16 %[" F "t2], %[" F "a2]\n\t"    \
    "ds_read_u16 %[" F "t3], %[" F "a3]\n\tds_read_u16 %[" F "t4], %[" F "a4]\n\tds_read_u16 %[" F "t5], %[" F "a5]\n\t"    \
    "ds_read_u16 %[" F "t6], %[" F "a6]\n\tds_read_u16 %[" F "t7], %[" F "a7]\n\t"
#define MF_SW1(F, T, A)                                                                                   \
    "v_lshrrev_b32_e32 %[" F T "], 5, %[" F A "]\n\tv_and_b32_e32 %[" F T "], 0xf8, %[" F T "]\n\t"       \
    "v_xor_b32_e32 %[" F T "], %[" F A "], %[" F T "]\n\tds_read_u16 %[" F T "], %[" F T "]\n\t"
#define MF_RD8_1(F)                                                                                           \
    MF_SW1(F, "t0", "a0") MF_SW1(F, "t1", "a1") MF_SW1(F, "t2", "a2") MF_SW1(F, "t3", "a3") MF_SW1(F, "t4", "a4") \
    MF_SW1(F, "t5", "a5") MF_SW1(F, "t6", "a6") MF_SW1(F, "t7", "a7")
#define MF_ISSUE8X(RD8, F, R0, R1, R2, R3) MF_ADDS8(F, R0, R1, R2, R3) RD8(F) "v_add_u32_e32 %[" F "offb], -16, %[" F "offb]\n\t"
// the eight answers of a fibre in turn, then window and entries left; N0 = reads of the other fibre still behind them
#if MI355_STEP_WAITS == 8
#define MF_TEST8(F, N7, N6, N5, N4, N3, N2, N1, N0)                                                               \
    MF_CMP(F, N7, "t0") MF_CMP(F, N6, "t1") MF_CMP(F, N5, "t2") MF_CMP(F, N4, "t3")                                \
    MF_CMP(F, N3, "t4") MF_CMP(F, N2, "t5") MF_CMP(F, N1, "t6") MF_CMP(F, N0, "t7")                                \
    "v_cmpx_ge_u32_e32 vcc, %[" F "a7], %[" F "lowa]\n\t"                                                         \
    "v_cmpx_ge_i32_e32 vcc, %[" F "offb], %[" F "endb]\n\t"
#elif MI355_STEP_WAITS == 4
#define MF_TEST8(F, N7, N6, N5, N4, N3, N2, N1, N0)                                                               \
    MF_CMPW(F, N6, "t0") MF_CMPN(F, N6, "t1") MF_CMPW(F, N4, "t2") MF_CMPN(F, N4, "t3")                            \
    MF_CMPW(F, N2, "t4") MF_CMPN(F, N2, "t5") MF_CMPW(F, N0, "t6") MF_CMPN(F, N0, "t7")                            \
    "v_cmpx_ge_u32_e32 vcc, %[" F "a7], %[" F "lowa]\n\t"                                                         \
    "v_cmpx_ge_i32_e32 vcc, %[" F "offb], %[" F "endb]\n\t"
#elif MI355_STEP_WAITS == 2
#define MF_TEST8(F, N7, N6, N5, N4, N3, N2, N1, N0)                                                               \
    MF_CMPW(F, N4, "t0") MF_CMPN(F, N4, "t1") MF_CMPN(F, N4, "t2") MF_CMPN(F, N4, "t3")                            \
    MF_CMPW(F, N0, "t4") MF_CMPN(F, N0, "t5") MF_CMPN(F, N0, "t6") MF_CMPN(F, N0, "t7")                            \
    "v_cmpx_ge_u32_e32 vcc, %[" F "a7], %[" F "lowa]\n\t"                                                         \
    "v_cmpx_ge_i32_e32 vcc, %[" F "offb], %[" F "endb]\n\t"
#else
#define MF_TEST8(F, N7, N6, N5, N4, N3, N2, N1, N0)                                                               \
    MF_CMPW(F, N0, "t0") MF_CMPN(F, N0, "t1") MF_CMPN(F, N0, "t2") MF_CMPN(F, N0, "t3")                            \
    MF_CMPN(F, N0, "t4") MF_CMPN(F, N0, "t5") MF_CMPN(F, N0, "t6") MF_CMPN(F, N0, "t7")                            \
    "v_cmpx_ge_u32_e32 vcc, %[" F "a7], %[" F "lowa]\n\t"                                                         \
    "v_cmpx_ge_i32_e32 vcc, %[" F "offb], %[" F "endb]\n\t"
#endif
// (addresses and answers are written before they are read by every lane that walks: plain outputs, so none of a fibre's
// sixteen stays live between its service and its next step block)
#define MF_OPS(F, S)                                                                                                        \
    [F##offb] "+v"(S.offb), [F##a0] "=&v"(S.a0), [F##a1] "=&v"(S.a1), [F##a2] "=&v"(S.a2), [F##a3] "=&v"(S.a3), [F##a4] "=&v"(S.a4), \
    [F##a5] "=&v"(S.a5), [F##a6] "=&v"(S.a6), [F##a7] "=&v"(S.a7), [F##t0] "=&v"(S.t0), [F##t1] "=&v"(S.t1), [F##t2] "=&v"(S.t2), \
    [F##t3] "=&v"(S.t3), [F##t4] "=&v"(S.t4), [F##t5] "=&v"(S.t5), [F##t6] "=&v"(S.t6), [F##t7] "=&v"(S.t7)
#define MF_INS(F, S) [F##bb] "v"(S.bb2), [F##lowa] "v"(S.lowa2), [F##probe] "v"(S.probe), [F##endb] "v"(S.endb)

#define MF_ONE(RD8, F, W, C, RA, RB, R0, R1, R2, R3, R4, R5, R6, R7)                           \
        "s_mov_b64 exec, %[" W "]\n\t"                                                      \
        "global_load_dwordx4 " RA ", %[" F "offb], %[sb] offset:-14\n\t"                    \
        "global_load_dwordx4 " RB ", %[" F "offb], %[sb] offset:-30\n\t"                    \
        "s_waitcnt vmcnt(1)\n\t"                                                            \
        MF_ISSUE8X(RD8, F, R0, R1, R2, R3)                                                        \
        MF_TEST8(F, "7", "6", "5", "4", "3", "2", "1", "0")                                 \
        "s_cbranch_execz .Lmf_one" F "%=\n\t"                                               \
        "s_waitcnt vmcnt(0)\n\t"                                                            \
        MF_ISSUE8X(RD8, F, R4, R5, R6, R7)                                                        \
        MF_TEST8(F, "7", "6", "5", "4", "3", "2", "1", "0")                                 \
        ".Lmf_one" F "%=:\n\t"                                                              \
        "s_mov_b64 %[" C "], exec\n\t"

// (towards the end of a pair of batches one fibre has often finished while the other still walks -- a quarter of the blocks on
// text: its vector instructions would issue all the same, an empty EXEC skips memory instructions, not vector ALU ones -- so
// a block for the fibre that is left: MF_ONE.  x: entries off-7 .. off in v[56:59], off-15 .. off-8 in v[60:63]; y: v[64:71].
// Two instantiations: the plain table and the permuted one, PairWinT<true>.)
#define MS_STEPS_DUAL_DEF(NAME, RD8) \
template <bool HAS_Q> \
__device__ __forceinline__ void NAME(SwG<HAS_Q>& x, SwG<HAS_Q>& y, const uint16_t* sb8, uint64_t walkx, uint64_t walky, \
                                              uint64_t* stillx, uint64_t* stilly) { \
    uint64_t save, cx, cy; \
    asm volatile( \
        "s_mov_b64 %[save], exec\n\t" \
 \
 \
 \
        "s_mov_b64 %[cx], 0\n\t" \
        "s_mov_b64 %[cy], 0\n\t" \
        "s_cmp_eq_u64 %[wy], 0\n\t" \
        "s_cbranch_scc1 .Lmf_xonly%=\n\t" \
        "s_cmp_eq_u64 %[wx], 0\n\t" \
        "s_cbranch_scc1 .Lmf_yonly%=\n\t" \
        "s_mov_b64 exec, %[wx]\n\t" \
        "global_load_dwordx4 v[56:59], %[xoffb], %[sb] offset:-14\n\t" \
        "global_load_dwordx4 v[60:63], %[xoffb], %[sb] offset:-30\n\t" \
        "s_mov_b64 exec, %[wy]\n\t" \
        "global_load_dwordx4 v[64:67], %[yoffb], %[sb] offset:-14\n\t" \
        "global_load_dwordx4 v[68:71], %[yoffb], %[sb] offset:-30\n\t" \
        "s_mov_b64 exec, %[wx]\n\t" \
        "s_waitcnt vmcnt(3)\n\t" \
        MF_ISSUE8X(RD8, "x", "v56", "v57", "v58", "v59") \
        "s_mov_b64 exec, %[wy]\n\t" \
        "s_waitcnt vmcnt(1)\n\t" \
        MF_ISSUE8X(RD8, "y", "v64", "v65", "v66", "v67") \
        "s_mov_b64 exec, %[wx]\n\t" \
        MF_TEST8("x", "15", "14", "13", "12", "11", "10", "9", "8") \
        "s_mov_b64 %[cx], exec\n\t" \
        "s_mov_b64 exec, %[wy]\n\t" \
        MF_TEST8("y", "7", "6", "5", "4", "3", "2", "1", "0") \
        "s_mov_b64 %[cy], exec\n\t" \
        "s_or_b64 vcc, %[cx], %[cy]\n\t" \
        "s_cbranch_scc0 .Lmf_end%=\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        "s_mov_b64 exec, %[cx]\n\t" \
        MF_ISSUE8X(RD8, "x", "v60", "v61", "v62", "v63") \
        "s_mov_b64 exec, %[cy]\n\t" \
        MF_ISSUE8X(RD8, "y", "v68", "v69", "v70", "v71") \
        "s_mov_b64 exec, %[cx]\n\t" \
        MF_TEST8("x", "15", "14", "13", "12", "11", "10", "9", "8") \
        "s_mov_b64 %[cx], exec\n\t" \
        "s_mov_b64 exec, %[cy]\n\t" \
        MF_TEST8("y", "7", "6", "5", "4", "3", "2", "1", "0") \
        "s_mov_b64 %[cy], exec\n\t" \
        "s_branch .Lmf_end%=\n\t" \
        ".Lmf_xonly%=:\n\t" \
        MF_ONE(RD8, "x", "wx", "cx", "v[56:59]", "v[60:63]", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63") \
        "s_branch .Lmf_end%=\n\t" \
        ".Lmf_yonly%=:\n\t" \
        MF_ONE(RD8, "y", "wy", "cy", "v[64:67]", "v[68:71]", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71") \
        ".Lmf_end%=:\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        "s_mov_b64 exec, %[save]\n\t" \
        : MF_OPS(x, x), MF_OPS(y, y), [save] "=&s"(save), [cx] "=&s"(cx), [cy] "=&s"(cy) \
        : MF_INS(x, x), MF_INS(y, y), [sb] "s"(sb8), [wx] "s"(walkx), [wy] "s"(walky) \
        : "vcc", "scc", "memory", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", \
          "v70", "v71"); \
    *stillx = cx; \
    *stilly = cy; \
}
MS_STEPS_DUAL_DEF(ms_steps_dual, MF_RD8_0)
MS_STEPS_DUAL_DEF(ms_steps_dual_swz, MF_RD8_1)
#undef MF_ONE

constexpr uint32_t ADV_TILE = 1024, ADV_HALO = 260;
struct TileM {
    const uint32_t* t;  // the staged tile
    __device__ uint32_t operator()(uint32_t r) const { return t[r]; }
};
// the end of the data the encoder had at a position, relative to a base and clipped to 32 bits
__device__ __forceinline__ uint32_t rel_end(const SegEnds& sg, uint64_t base, uint32_t r) {
    const uint64_t e = (uint64_t)seg_end(sg, base + r) - base;
    return e > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)e;
}
// One epoch (or one part of it) by one workgroup: the body of both kernels below.  SWZ: the permuted table (PairWinT<true>).
template <bool HAS_Q, bool SWZ, bool SINGLE = false>
__device__ __forceinline__ void m3_epoch(uint4* s_T, uint32_t& s_next, const uint32_t e, const uint32_t part, const uint8_t* __restrict__ in, uint32_t n,
                                         const uint16_t* __restrict__ Sg, const uint16_t* __restrict__ Bg, uint32_t* __restrict__ M,
                                         uint32_t* __restrict__ Mq, uint32_t checks, uint32_t checks_q, int in_aligned16, SegEnds sg,
                                         HashOverride ov, uint32_t split, uint32_t* __restrict__ Ms, uint32_t* __restrict__ Mqs,
                                         uint32_t* __restrict__ sort_bad) {
    constexpr uint32_t pair = SINGLE ? 1u : 2u;  // batches a wave takes at a time (SINGLE: a small call, see walk_all)
    const uint32_t tid = threadIdx.x, lane = tid & 63;
#ifdef MI355_MATCH_STATS
    uint32_t m2c[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    M2_T0
    const uint64_t E = (uint64_t)e * WINDOW_SIZE;
    const uint64_t wbase = e ? E - WINDOW_SIZE : 0;
    const uint32_t wbytes = (uint32_t)(E - wbase) + WINDOW_SIZE + 258 + 14;
    const uint32_t tb0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)s_T;
    // stage the window: 16 bytes and the byte behind them make 16 pairs = two 16-byte stores
    for (uint32_t w = tid; w < wbytes / 16; w += M3T) {
        const uint64_t g = wbase + 16ull * w;
        uint32_t t[5] = {0, 0, 0, 0, 0};
        if (in_aligned16 && g + 20 <= n) {
            const uint4 v = *reinterpret_cast<const uint4*>(in + g);
            t[0] = v.x;
            t[1] = v.y;
            t[2] = v.z;
            t[3] = v.w;
            t[4] = *reinterpret_cast<const uint32_t*>(in + g + 16);
        } else {
            for (int b = 0; b < 17; b++)
                if (g + b < n) t[b >> 2] |= (uint32_t)in[g + b] << (8 * (b & 3));
        }
        uint32_t o[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            o[2 * k] = __builtin_amdgcn_perm(t[k], t[k], 0x02010100u);          // b0 b1 | b1 b2
            o[2 * k + 1] = __builtin_amdgcn_perm(t[k + 1], t[k], 0x04030302u);  // b2 b3 | b3 b4
        }
        if (SWZ) {  // (8-byte words, each where the permutation puts it)
            typedef __attribute__((address_space(3))) uint64_t* lds_u64;
            const uint32_t a = tb0 + 32u * w;
#pragma unroll
            for (int k = 0; k < 4; k++) *(lds_u64)m3_swz(a + 8u * k) = (uint64_t)o[2 * k] | ((uint64_t)o[2 * k + 1] << 32);
        } else {
            s_T[2 * w] = make_uint4(o[0], o[1], o[2], o[3]);
            s_T[2 * w + 1] = make_uint4(o[4], o[5], o[6], o[7]);
        }
    }
    const uint32_t J = epoch_active(n, E);
    const uint32_t nbat = (J + 63) / 64;
    const uint32_t b_lo = (uint32_t)((uint64_t)nbat * part / split), b_hi = (uint32_t)((uint64_t)nbat * (part + 1) / split);
    if (tid == 0) s_next = b_lo;
    if (part == 0 && tid < 2) {  // the positions without a hash byte (the last two of the input) are never searched
        const uint64_t p = E + J + tid;
        if (p < n && p < E + WINDOW_SIZE) {
            M[p] = 0;
            if (HAS_Q) Mq[p] = 0;
        }
    }
    __syncthreads();
    const uint32_t tbase = tb0;
    const uint32_t bias = (uint32_t)(E - wbase);  // position of the own epoch's first byte in the window
    // (the array pointer may lie before the array for epoch 0; only indices >= 32768 - 7 are read then, and
    // the array has a pad in front)
    const uint16_t* sbase = Sg + (size_t)e * WINDOW_SIZE - WINDOW_SIZE;
    const uint16_t* own = Sg + (size_t)e * WINDOW_SIZE;
    const uint16_t* Bown = Bg + (size_t)e * BSTRIDE;
    const uint16_t* Bprev = Bown - BSTRIDE;
    TileLimit lim{sg, wbase, sg.m == 1 ? (uint32_t)(sg.ends[0] - wbase) : 0u};
    M2_T(10)  // (the window staged, the barrier)
    uint32_t unordered = 0;
    // set a fibre up for batch b (all lanes call it: the lane masks it sets must be ballots of the whole wave)
    // (before0: the entry in front of the batch's first one, for the order check below)
    // (skip / budget / budget_q: a small call's second fibre walks the SAME batch from `skip` candidates down -- see walk_all)
    auto set_up = [&](const auto& win, SwG<HAS_Q>& st, uint32_t b, bool have, uint32_t* srel_out, uint32_t before0, uint32_t* last_out,
                      uint32_t skip, uint32_t budget, uint32_t budget_q) -> bool {
        const uint32_t j = b * 64 + lane;
        const bool valid = have && j < J;
        uint32_t raw = 0, srel = 0, ob = 0, pb0 = 0, pb1 = 0, prel = bias, nrel = bias;  // (a lane without a position: nothing to search)
        if (valid) {
            raw = (uint32_t)own[j];
            srel = raw >> 1;
            prel = bias + srel;
            const uint32_t v = win.load32(prel);
            const uint32_t ab = rewarm_ab(ov, E + srel, v & 0xff, (v >> 8) & 0xff);
            const uint32_t h = hash3(ab & 0xff, ab >> 8, (v >> 16) & 0xff);
            ob = Bown[h];
            if (e) {
                pb0 = Bprev[h];
                pb1 = Bprev[h + 1];
            }
            nrel = lim(prel);
        }
        // The chain order IS the order of the bucket's entries (nearest first = descending positions).  k_sort<1> takes
        // a key's rank from the order in which the LDS serves the lanes of one atomic -- what the hardware does, not what
        // the ISA promises (k_lds_order_test samples it when a context is made) -- so the property is checked on the data
        // itself, for every entry and the one before it in its bucket: a violation could never corrupt a stream, but it
        // would silently change which of two equally long matches wins.  The host then sorts again with ballot ranks.
        // (The entry before a lane's own is the lane below's -- one shift of the wave; a load of own[j - 1] here cost 2.3 %
        // of the kernel.)
        const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp((int)before0, (int)raw, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        unordered |= (valid && j > ob && before >= raw) ? 1u : 0u;
        *last_out = (uint32_t)__builtin_amdgcn_readlane((int)raw, 63);
        // (the candidates from rank skip + 1 on are those of the entry `skip` places down its bucket -- the same array, shifted --
        // and where the own epoch's part of the bucket is shorter than that, what is left of `skip` comes off the previous epoch's)
        uint32_t je = valid ? j : 0u, pe1 = pb1;
        if (skip) {
            const uint32_t n_own = je - ob;
            if (n_own >= skip) {
                je -= skip;
            } else {
                const uint32_t rest = skip - n_own, n_prev = pb1 - pb0;
                je = ob;
                pe1 = pb1 - (rest < n_prev ? rest : n_prev);
            }
        }
        (void)swg_setup(st, win, je, ob, pb0, pe1, prel, nrel, tbase, bias, budget, budget_q);
        *srel_out = srel;
        return valid;
    };
    // settle the lanes of a fibre that left the last block
    auto settle = [&](auto run1, const auto& win, SwG<HAS_Q>& st, uint64_t dropped) {
        uint64_t any;
        uint32_t asel, back;
        ms_decode(MsGroup{st.t0, st.t1, st.t2, st.t3, st.t4, st.t5, st.t6, st.t7, st.a0, st.a1, st.a2, st.a3, st.a4, st.a5,
                                      st.a6, st.a7, st.probe},
                              &any, &asel, &back);
        // a lane that used up its last segment without a hit has its result: only hits and the move to the
        // previous epoch's bucket need the service
        if ((any & dropped) | (dropped & st.has2)) {
            M2_CNT(2, 1)
            M2_CNT(3, __popcll(dropped))
            swg_service<decltype(run1)::value>(st, win, tbase, checks_q, dropped, any & dropped, asel, st.offb + back);
        }
    };
    uint32_t* const Mo = Ms ? Ms : M;      // where a batch's results go: in the order of S_e (turned round at the end), else by position
    uint32_t* const Mqo = Ms ? Mqs : Mq;
    // The walk of the workgroup's batches, in two instantiations: RUN1 is the service with the one-sided long compare
    // (stages.h swg_service) for epochs that k_sort found to consist of runs of one byte -- zero fill and the like, where
    // every position's first candidate is its neighbour and matches to the end -- and costs the other epochs nothing.
    // (the step block's base of the sorted arrays, made scalar once: inside the loop it was two read-first-lanes per round)
    const uint16_t* sb8u;
    {
        const uint64_t v = (uint64_t)(uintptr_t)(sbase - 4);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        sb8u = (const uint16_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    }
    auto walk_all = [&](auto run1) {
        constexpr bool RUN1 = decltype(run1)::value;
        const PairWinT<SWZ> win{sbase, tbase};
        uint32_t pxat = ~0u, pxm = 0, pxq = 0, pyat = ~0u, pym = 0, pyq = 0;
        for (;;) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(&s_next, pair);  // (pair == 1: a small call's waves take one batch each -- the second fibre stays empty)
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            if (b >= b_hi) break;
            SwG<HAS_Q> sx, sy;
            uint32_t srx, sry;
            // (the entry in front of the pair's first one: a scalar load -- b is the same for the whole wave)
            uint32_t lastx, lasty;
            const uint32_t front = b ? (uint32_t)own[(uint32_t)__builtin_amdgcn_readfirstlane((int)(b * 64 - 1))] : 0u;
            // A small call (pair == 1: a batch a wave) gives its second fibre the far half of the SAME batch's candidates: what
            // longest_match computes is a function of the candidate set (the longest common prefix, the nearest among equals:
            // stages.h, the head of the sorted walk), so the first `half` candidates and the rest can be walked side by side
            // and the far half's result taken only where it is strictly longer.  The quarter-budget result (checks >> 2 <=
            // half) is the near half's.  A batch's time is its slowest lane's chain of step blocks: half as long this way.
            const uint32_t half = (pair == 1u && checks >= 16u) ? checks / 2u : 0u;
            const bool vx = set_up(win, sx, b, true, &srx, front, &lastx, 0u, half ? half : checks, checks_q);
            const bool vy = half ? set_up(win, sy, b, true, &sry, front, &lasty, half, checks - half, 0u)
                                 : set_up(win, sy, b + 1, pair == 2u && b + 1 < b_hi, &sry, lastx, &lasty, 0u, checks, checks_q);
            (void)lasty;
            M2_CNT(0, 2)
            M2_T(8)
            // (the results of the pair before go out here, behind this pair's set-up loads)
            if (pxat != ~0u) {
                Mo[E + pxat] = pxm;
                if (HAS_Q) Mqo[E + pxat] = pxq;
            }
            if (pyat != ~0u) {
                Mo[E + pyat] = pym;
                if (HAS_Q) Mqo[E + pyat] = pyq;
            }
            // the first candidate of every lane goes straight to the service
            // (no finding out where they stopped: every lane concerned "left at the first probe of its group")
            {
                const uint64_t dx = swg_first(sx, win, 8u), dy = swg_first(sy, win, 8u);
                if (dx) swg_service<RUN1>(sx, win, tbase, checks_q, dx, dx, sx.a0, sx.offb + 16u);
                if (dy) swg_service<RUN1>(sy, win, tbase, checks_q, dy, dy, sy.a0, sy.offb + 16u);
                M2_CNT(2, 2)
                M2_CNT(3, __popcll(dx) + __popcll(dy))
                M2_T(9)
            }
            for (;;) {
                const uint64_t wx = sx.walk, wy = sy.walk;
                if ((wx | wy) == 0) break;
                M2_CNT(1, 1)
                M2_CNT(7, __popcll(wx) + __popcll(wy))
                uint64_t cx, cy;
                if (SWZ)
                    ms_steps_dual_swz(sx, sy, sb8u, wx, wy, &cx, &cy);
                else
                    ms_steps_dual(sx, sy, sb8u, wx, wy, &cx, &cy);
                M2_T(12)
                sx.walk = cx;
                sy.walk = cy;
                // (the lanes that left the block, as scalar instructions: left to itself the compiler takes the second fibre's masks
                // for lane values -- two moves, two bit-selects, a 64-bit compare and two read-first-lanes per round)
                uint64_t dx, dy;
                asm("s_andn2_b64 %0, %2, %3\n\ts_andn2_b64 %1, %4, %5" : "=&s"(dx), "=&s"(dy) : "s"(wx), "s"(cx), "s"(wy), "s"(cy) : "scc");
                if (dx) settle(run1, win, sx, dx);
                if (dy) settle(run1, win, sy, dy);
                M2_T(9)
            }
            swg_result(sx, &pxm, &pxq);
            swg_result(sy, &pym, &pyq);
            pxat = vx ? (Ms ? b * 64 + lane : srx) : ~0u;
            pyat = vy ? (Ms ? b * 64 + 64 + lane : sry) : ~0u;
            if (half) {  // (the far half of the same positions: only a strictly longer match counts)
                pxm = m_len(pym) > m_len(pxm) ? pym : pxm;
                pyat = ~0u;
            }
            M2_T(13)
        }
        if (pxat != ~0u) {
            Mo[E + pxat] = pxm;
            if (HAS_Q) Mqo[E + pxat] = pxq;
        }
        if (pyat != ~0u) {
            Mo[E + pyat] = pym;
            if (HAS_Q) Mqo[E + pyat] = pyq;
        }
    };
    if (!SWZ && __builtin_amdgcn_readfirstlane((int)Bown[WINDOW_SIZE + 1]))  // (rows of records are no runs of one byte)
        walk_all(std::true_type{});
    else
        walk_all(std::false_type{});
    if (__builtin_amdgcn_ballot_w64(unordered != 0) != 0 && lane == 0) atomicOr(sort_bad, 1u);
    M2_T(11)  // (a wave's last results, and the wait for nothing: the walk's loop ended at its last M2_T)
    if (Ms) {
        // The results went out in the order of S_e -- a batch's 64 results are 256 consecutive bytes; stored by position
        // they were 64 stores into 64 lines, which left the L2 as partial lines over and over (WRITE_SIZE 4.6 GB for
        // 0.4 GB of M).  Now that the walk is over the pair table is not needed any more: its LDS takes the epoch's
        // results by position, and they leave as whole lines.
        uint32_t* const lm = reinterpret_cast<uint32_t*>(s_T);
        const uint32_t cnt = (uint32_t)((uint64_t)n - E < (uint64_t)WINDOW_SIZE ? (uint64_t)n - E : (uint64_t)WINDOW_SIZE);
        for (int pass = 0; pass < (HAS_Q ? 2 : 1); pass++) {
            const uint32_t* src = pass ? Mqs : Ms;
            uint32_t* dst = pass ? Mq : M;
            __syncthreads();
            for (uint32_t j = tid; j < J; j += M3T) lm[(uint32_t)own[j] >> 1] = src[E + j];
            if (tid < 2 && J + tid < cnt) lm[J + tid] = 0;  // the positions without a hash byte
            __syncthreads();
            for (uint32_t i = tid * 4; i < cnt; i += M3T * 4) {
                if (i + 4 <= cnt) {
                    *reinterpret_cast<uint4*>(dst + E + i) = *reinterpret_cast<const uint4*>(lm + i);
                } else {
                    for (uint32_t k = i; k < cnt; k++) dst[E + k] = lm[k];
                }
            }
        }
    }
    M2_T(14)  // (the results turned round: whole epochs only)
    M2_CNT(4, 1)  // waves
#ifdef MI355_MATCH_STATS
    if (lane == 0)
        for (int i = 0; i < 16; i++) atomicAdd(&g_mstats[i], (unsigned long long)m2c[i]);
#endif
}

constexpr uint32_t M3_TABLE_U4 = (M3_PAIRS * 2 + 255) / 256 * 16;  // the table in uint4: whole 256-byte blocks (PairWinT<true>)
// a workgroup per epoch (or per part of one: small inputs).  Epochs that k_sort marked for the permuted table are left to
// k_match3_swz.
template <bool HAS_Q>
__global__ __launch_bounds__(M3T) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_match3(const uint8_t* __restrict__ in, uint32_t n, const uint16_t* __restrict__ Sg,
                                                const uint16_t* __restrict__ Bg, uint32_t* __restrict__ M,
                                                uint32_t* __restrict__ Mq, uint32_t checks, uint32_t checks_q, int in_aligned16,
                                                SegEnds sg, HashOverride ov, uint32_t e0, uint32_t split, uint32_t* __restrict__ Ms,
                                                uint32_t* __restrict__ Mqs, uint32_t* __restrict__ sort_bad) {
    __shared__ __attribute__((aligned(256))) uint4 s_T[M3_TABLE_U4];  // T[k] = byte k | byte k+1 << 8, k from the window's start
    __shared__ uint32_t s_next;
    const uint32_t e = e0 + blockIdx.x / split, part = blockIdx.x % split;
    if (MI355_SWZ_BANKS && __builtin_amdgcn_readfirstlane((int)Bg[(size_t)e * BSTRIDE + WINDOW_SIZE + 2])) return;
    m3_epoch<HAS_Q, false>(s_T, s_next, e, part, in, n, Sg, Bg, M, Mq, checks, checks_q, in_aligned16, sg, ov, split, Ms, Mqs, sort_bad);
}
// The epochs of [e0, e0 + ne) that k_sort marked -- their sorted positions sit on few LDS banks: rows of records -- with the
// permuted table; a workgroup per compute unit takes the marked epochs in turn (text has none: the launch is 256 workgroups
// that read a dozen flags each and leave).
template <bool HAS_Q>
__global__ __launch_bounds__(M3T) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_match3_swz(const uint8_t* __restrict__ in, uint32_t n, const uint16_t* __restrict__ Sg,
                                                const uint16_t* __restrict__ Bg, uint32_t* __restrict__ M,
                                                uint32_t* __restrict__ Mq, uint32_t checks, uint32_t checks_q, int in_aligned16,
                                                SegEnds sg, HashOverride ov, uint32_t e0, uint32_t units, uint32_t split,
                                                uint32_t* __restrict__ Ms, uint32_t* __restrict__ Mqs, uint32_t* __restrict__ sort_bad) {
    __shared__ __attribute__((aligned(256))) uint4 s_T[M3_TABLE_U4];
    __shared__ uint32_t s_next;
#if MI355_SWZ_PERSISTENT
    for (uint32_t u = blockIdx.x; u < units; u += gridDim.x) {  // units: epochs x parts
        const uint32_t e = e0 + u / split;
        if (!__builtin_amdgcn_readfirstlane((int)Bg[(size_t)e * BSTRIDE + WINDOW_SIZE + 2])) continue;
        __syncthreads();  // (the unit before is done with the LDS)
        m3_epoch<HAS_Q, true>(s_T, s_next, e, u % split, in, n, Sg, Bg, M, Mq, checks, checks_q, in_aligned16, sg, ov, split, Ms, Mqs, sort_bad);
    }
#else
    (void)units;
    const uint32_t e = e0 + blockIdx.x / split;
    if (!__builtin_amdgcn_readfirstlane((int)Bg[(size_t)e * BSTRIDE + WINDOW_SIZE + 2])) return;
    m3_epoch<HAS_Q, true>(s_T, s_next, e, blockIdx.x % split, in, n, Sg, Bg, M, Mq, checks, checks_q, in_aligned16, sg, ov, split, Ms, Mqs, sort_bad);
#endif
}

// Small inputs (at most M3_BOTH_UNITS workgroups): one launch whose workgroups take the table their epoch is marked for --
// the second launch's 5 us are 2 % of a 167 KB file's time.  (Not for large inputs: the two bodies in one kernel cost the
// plain one registers, + 2 % on text.)
#ifndef MI355_M3_BOTH_UNITS
#define MI355_M3_BOTH_UNITS 256
#endif
constexpr uint32_t M3_BOTH_UNITS = MI355_M3_BOTH_UNITS;
template <bool HAS_Q, bool SINGLE>
__global__ __launch_bounds__(M3T) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_match3_both(const uint8_t* __restrict__ in, uint32_t n, const uint16_t* __restrict__ Sg,
                                                const uint16_t* __restrict__ Bg, uint32_t* __restrict__ M,
                                                uint32_t* __restrict__ Mq, uint32_t checks, uint32_t checks_q, int in_aligned16,
                                                SegEnds sg, HashOverride ov, uint32_t e0, uint32_t split, uint32_t* __restrict__ Ms,
                                                uint32_t* __restrict__ Mqs, uint32_t* __restrict__ sort_bad) {
    __shared__ __attribute__((aligned(256))) uint4 s_T[M3_TABLE_U4];
    __shared__ uint32_t s_next;
    const uint32_t e = e0 + blockIdx.x / split, part = blockIdx.x % split;
    if (MI355_SWZ_BANKS && __builtin_amdgcn_readfirstlane((int)Bg[(size_t)e * BSTRIDE + WINDOW_SIZE + 2]))
        m3_epoch<HAS_Q, true, SINGLE>(s_T, s_next, e, part, in, n, Sg, Bg, M, Mq, checks, checks_q, in_aligned16, sg, ov, split, Ms, Mqs, sort_bad);
    else
        m3_epoch<HAS_Q, false, SINGLE>(s_T, s_next, e, part, in, n, Sg, Bg, M, Mq, checks, checks_q, in_aligned16, sg, ov, split, Ms, Mqs, sort_bad);
}

// ---------------------------------------------------------------------------------------------
// k_rle: rle.rs:13-18 get_match_length_rle for every position: R[p] = run of data[p-1]
// starting at p, capped at 258 and at the end of input.
// A workgroup takes a tile of 4096 positions.  "Position i continues the run" (data[i] == data[i-1]) is one bit; a lane
// works out the sixteen bits of its sixteen positions from five aligned dwords of the staged tile (byte-wise zero test of
// the words XOR-ed with themselves shifted by one byte) and leaves them in LDS; the run that begins behind its chunk is the
// number of set bits from there on -- at most seventeen 16-bit words, not 258 byte compares in a row -- and the sixteen
// results follow from one backward pass over its own bits.  They leave through LDS as whole lines.
// (Before: bytes staged one by one, the forward scan a loop of up to 258 dependent LDS reads per lane -- on zero fill,
// BASELINE config 2, every lane ran all of it -- and sixteen 4-byte stores per lane, 64 bytes apart: 3.16 ms for 256 MiB.)
// ---------------------------------------------------------------------------------------------
constexpr uint32_t RT = 4096;                       // positions of a tile
constexpr uint32_t RLE_UNITS = RT / 16 + 17;        // 16-position units with bits: the tile and 272 >= MAX_MATCH positions behind it
constexpr uint32_t RLE_BYTES = 16 + RLE_UNITS * 16; // staged bytes: s[j] = in[E - 16 + j]
// (adv: the restart step of the RLE level -- rle.rs:46-69: a run of three or more is taken whole, else one literal -- needs
// nothing but the run at the position, so it is written here and k_adv is not launched for this level.)
__global__ __launch_bounds__(256) void k_rle(const uint8_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ R,
                                             uint16_t* __restrict__ adv, SegEnds sg) {
    __shared__ __attribute__((aligned(16))) uint8_t s[RLE_BYTES];
    __shared__ uint32_t eb[RLE_UNITS + 1];
    __shared__ __attribute__((aligned(16))) uint32_t so[RT];
    __shared__ __attribute__((aligned(16))) uint16_t sa[RT];
    const uint32_t tid = threadIdx.x;
    const uint64_t E = (uint64_t)blockIdx.x * RT;
    const bool al = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    for (uint32_t k = tid; k < RLE_BYTES / 16; k += 256) {
        const int64_t g0 = (int64_t)E - 16 + 16 * (int64_t)k;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (al && g0 >= 0 && (uint64_t)g0 + 16 <= n) {
            v = *reinterpret_cast<const uint4*>(in + g0);
        } else {
            uint32_t t[4] = {0, 0, 0, 0};
            for (int b = 0; b < 16; b++) {
                const int64_t g = g0 + b;
                if (g >= 0 && (uint64_t)g < n) t[b >> 2] |= (uint32_t)in[g] << (8 * (b & 3));
            }
            v = make_uint4(t[0], t[1], t[2], t[3]);
        }
        *reinterpret_cast<uint4*>(s + 16 * k) = v;
    }
    __syncthreads();
    // the bits of unit u: position i = 16 u + j (tile relative) sits at s[i + 16], the byte before it at s[i + 15]
    auto unit_bits = [&](uint32_t u) -> uint32_t {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(s + 16 * u + 12);
        const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4];
        const uint32_t z[4] = {d1 ^ __builtin_amdgcn_alignbyte(d1, d0, 3u), d2 ^ __builtin_amdgcn_alignbyte(d2, d1, 3u),
                               d3 ^ __builtin_amdgcn_alignbyte(d3, d2, 3u), d4 ^ __builtin_amdgcn_alignbyte(d4, d3, 3u)};
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t m = ~(((z[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z[k]) & 0x80808080u;  // bit 7 of every zero byte
            bits |= ((((m >> 7) * 0x00204081u) >> 21) & 15u) << (4 * k);
        }
        // positions 0 and >= n continue nothing (the bytes staged for them are zeros, which may well be equal)
        const uint64_t g0 = E + 16ull * u;
        if (g0 == 0) bits &= ~1u;
        if (g0 + 16 > n) bits &= g0 >= n ? 0u : (1u << (uint32_t)((uint64_t)n - g0)) - 1u;
        return bits;
    };
    const uint32_t mine = unit_bits(tid);
    eb[tid] = mine;
    if (tid < RLE_UNITS - RT / 16) eb[RT / 16 + tid] = unit_bits(RT / 16 + tid);
    __syncthreads();
    // the run that begins right behind my sixteen positions
    uint32_t c = 0;
    for (uint32_t k = tid + 1; k < tid + 18; k++) {
        const uint32_t x = ~eb[k] & 0xffffu;
        if (x) {
            c += (uint32_t)__builtin_ctz(x);
            break;
        }
        c += 16;
    }
    c = c < MAX_MATCH ? c : (uint32_t)MAX_MATCH;
    uint32_t r16[16];
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        c = ((mine >> i) & 1u) ? c + 1 : 0u;
        uint32_t r = c < MAX_MATCH ? c : (uint32_t)MAX_MATCH;
        if (sg.m > 1) {  // a run is cut where the data the encoder had ended (sync flush)
            const uint64_t g = E + tid * 16 + (uint32_t)i;
            if (g < n) {
                const uint64_t left = (uint64_t)seg_end(sg, g) - g;
                if (r > left) r = (uint32_t)left;
            }
        }
        r16[i] = r;
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
        *reinterpret_cast<uint4*>(so + tid * 16 + 4 * q) = make_uint4(r16[4 * q], r16[4 * q + 1], r16[4 * q + 2], r16[4 * q + 3]);
    {
        uint32_t a16[16];
#pragma unroll
        for (int i = 0; i < 16; i++) a16[i] = r16[i] >= MIN_MATCH ? r16[i] : 1u;
#pragma unroll
        for (int q = 0; q < 2; q++)
            *reinterpret_cast<uint4*>(sa + tid * 16 + 8 * q) =
                make_uint4(a16[8 * q] | (a16[8 * q + 1] << 16), a16[8 * q + 2] | (a16[8 * q + 3] << 16),
                           a16[8 * q + 4] | (a16[8 * q + 5] << 16), a16[8 * q + 6] | (a16[8 * q + 7] << 16));
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < RT / (8 * 256); k++) {
        const uint32_t i = (k * 256 + tid) * 8;
        const uint64_t g = E + i;
        if (g + 8 <= n) {
            *reinterpret_cast<uint4*>(adv + g) = *reinterpret_cast<const uint4*>(sa + i);  // (adv is 256-byte aligned, g a multiple of 8)
        } else {
            for (uint32_t j = 0; j < 8 && g + j < n; j++) adv[g + j] = sa[i + j];
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < RT / (4 * 256); k++) {
        const uint32_t i = (k * 256 + tid) * 4;
        const uint64_t g = E + i;
        if (g + 4 <= n) {
            *reinterpret_cast<uint4*>(R + g) = *reinterpret_cast<const uint4*>(so + i);  // (R is 256-byte aligned, g a multiple of 4)
        } else {
            for (uint32_t j = 0; j < 4 && g + j < n; j++) R[g + j] = so[i + j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_adv: lz77.rs:305-547 / rle.rs:23-71 seen from a restart position: how far does the parser
// get before it is again in a state that depends on the position only.
// ---------------------------------------------------------------------------------------------
// M of a workgroup's 1024 positions plus a halo, staged in LDS: the lazy step looks one entry ahead per deferral
// (lz77.rs:351-355), a chain of dependent reads that is cheap from LDS.  The halo covers the longest chain there is: a
// deferral needs a strictly longer match (3 .. 258), so a step reads at most 256 entries beyond its position -- no step
// leaves the tile, the reads are plain LDS reads at 32-bit tile-relative positions (with a 64-entry halo and a fall-back
// to global memory they were generic loads behind 64-bit selects, and the kernel was bound by its vector instructions).
__global__ __launch_bounds__(256) void k_adv(uint32_t n, const uint32_t* __restrict__ M, const uint32_t* __restrict__ Mq,
                                             ParseCfg cfg, uint16_t* __restrict__ adv, SegEnds sg, uint32_t blk0) {
    __shared__ __attribute__((aligned(16))) uint32_t sM[ADV_TILE + ADV_HALO], sQ[ADV_TILE + ADV_HALO];
    const uint64_t t0 = ((uint64_t)blockIdx.x + blk0) * ADV_TILE;  // (blk0: a launch may cover a range of tiles)
    const bool useq = Mq != nullptr;
    // (the tables are padded by 64 entries and 256-byte aligned, a tile starts at a multiple of 1024 entries:
    // sixteen bytes per lane; entries beyond the padding read as "no match")
    for (uint32_t i = threadIdx.x * 4; i < ADV_TILE + ADV_HALO; i += 1024) {
        const uint64_t g = t0 + i;
        uint4 v = make_uint4(0, 0, 0, 0), q = make_uint4(0, 0, 0, 0);
        if (g + 4 <= (uint64_t)n + 64) {
            v = *reinterpret_cast<const uint4*>(M + g);
            if (useq) q = *reinterpret_cast<const uint4*>(Mq + g);
        } else {
            uint32_t t[4] = {0, 0, 0, 0}, u[4] = {0, 0, 0, 0};
            for (int k = 0; k < 4; k++)
                if (g + k < (uint64_t)n + 64) {
                    t[k] = M[g + k];
                    if (useq) u[k] = Mq[g + k];
                }
            v = make_uint4(t[0], t[1], t[2], t[3]);
            q = make_uint4(u[0], u[1], u[2], u[3]);
        }
        *reinterpret_cast<uint4*>(sM + i) = v;
        if (useq) *reinterpret_cast<uint4*>(sQ + i) = q;
    }
    __syncthreads();
    // four consecutive positions per lane, one 8-byte store of adv
    const uint32_t r0 = threadIdx.x * 4;
    if (t0 + r0 >= n) return;
    const uint32_t left = (uint64_t)n - t0 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)((uint64_t)n - t0);  // positions from t0
    const uint32_t one = sg.m == 1 ? rel_end(sg, t0, 0) : 0u;  // without flush points: one end for all
    TileM m{sM}, mq{useq ? sQ : sM};
    uint16_t a[4] = {0, 0, 0, 0};
#if MI355_ADV_STRAIGHT
    // The lazy step without its loop (one end for all, full-budget table only): "a+1 beats a" is a property of the
    // position -- len(M[a]) < lazy_lt, a+1 has a hash byte, len(M[a+1]) > len(M[a]) -- so the deferral chain from j is
    // the run of such positions starting at j.  Eight of these bits from the lane's own nine entries cover the chains
    // of its four positions unless one is longer than four deferrals (then the loop below takes that lane).
    bool loop_it = !(cfg.mode == MODE_LAZY && !cfg.use_quarter && sg.m == 1);
    if (!loop_it) {
        const uint4 e0 = *reinterpret_cast<const uint4*>(sM + r0), e1 = *reinterpret_cast<const uint4*>(sM + r0 + 4);
        const uint32_t e[9] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, sM[r0 + 8]};
        uint32_t ups = 0;
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t L = m_len(e[i]), L2 = m_len(e[i + 1]);
            ups |= (uint32_t)(L < cfg.lazy_lt && L2 > L) << i;
        }
        // a + 1 + 2 < one  for a = r0 + i:  i < one - r0 - 3
        const uint32_t room = one > r0 + 3 ? one - r0 - 3 : 0u;
        ups &= room >= 8 ? 0xFFu : (1u << room) - 1u;
        uint32_t runs[4];
        bool deep = false;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            runs[q] = (uint32_t)__builtin_ctz(~(ups >> q));
            deep = deep || q + runs[q] >= 8;
        }
        if (!deep) {
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                const uint32_t L = m_len(e[q]);
                const bool ok = r0 + q + 2 < one && L >= MIN_MATCH && !match_too_far(L, m_dist(e[q]));
                const uint32_t Le = m_len(sM[r0 + q + runs[q]]);
                a[q] = r0 + q < left ? (uint16_t)(ok ? (runs[q] + Le) | (runs[q] << ADV_RUN_SHIFT) : 1u) : (uint16_t)0;
            }
        }
        loop_it = deep;
    }
    if (loop_it)
#endif
#pragma unroll
    for (uint32_t q = 0; q < 4; q++)
        if (r0 + q < left) a[q] = (uint16_t)adv_pack(parse_step(m, mq, r0 + q, sg.m == 1 ? one : rel_end(sg, t0, r0 + q), cfg));
    if (r0 + 4 <= left) {
        uint2 v = make_uint2((uint32_t)a[0] | ((uint32_t)a[1] << 16), (uint32_t)a[2] | ((uint32_t)a[3] << 16));
        *reinterpret_cast<uint2*>(adv + t0 + r0) = v;  // adv is 256-byte aligned, r0 a multiple of 4
    } else {
        for (uint32_t q = 0; q < 4 && r0 + q < left; q++) adv[t0 + r0 + q] = a[q];
    }
}

// ---------------------------------------------------------------------------------------------
// k_seg_exit: for every position of a segment the first path position at or beyond the end of
// the segment (right-to-left sweep), and the level-0 table over the segment's entry zone.
// ---------------------------------------------------------------------------------------------
// One wave per segment, right to left in chunks of 64 positions: a lane whose jump leaves the chunk
// is resolved at once (from the table of the chunks already done, kept in LDS); jumps that stay
// inside the chunk are resolved by pointer jumping across lanes (<= 7 rounds).
// the sweep of one segment by one wave: J[r] = the exit of position r (relative to the segment's end)
__device__ __forceinline__ void seg_sweep(const uint16_t* __restrict__ adv, uint64_t a, uint32_t len, uint16_t* J, uint32_t lane) {
    // (all of the segment's jumps are fetched before the sweep: one load per chunk inside it was sixteen memory latencies
    // in a row per wave, and those were the kernel's time)
    uint16_t av[SEG / 64];
#pragma unroll
    for (uint32_t q = 0; q < SEG / 64; q++) av[q] = q * 64 + lane < len ? adv[a + q * 64 + lane] : (uint16_t)0;
#pragma unroll
    for (int32_t q = (int32_t)(SEG / 64) - 1; q >= 0; q--) {
        const int32_t c = q * 64;
        if ((uint32_t)c >= len) continue;
        uint32_t r = (uint32_t)c + lane;  // segment-relative position
        bool valid = r < len;
        // (k_adv's entries: adv_pack.  The mask is taken here, behind an opaque move: left to itself the compiler masks every
        // entry right behind its load, with a wait for that load -- sixteen memory latencies in a row again, 0.2 -> 0.4 ms)
        uint32_t step = av[q];
        asm volatile("" : "+v"(step));
        uint32_t t = valid ? r + (step & ADV_LEN_MASK) : 0;
        // one word per lane: bit 31 set = resolved, the low bits the exit; clear = the lane (of this chunk) it jumps to, times 4.
        // A round of pointer jumping is then ONE cross-lane read -- the word of the target is either its answer or the
        // lane two jumps on -- where value, flag and target were three (the kernel's time was their trips through the LDS
        // crossbar)
        uint32_t w = 0x80000000u;
        if (valid) {
            if (t >= len) {
                w = 0x80000000u | (t - len);
            } else if (t >= (uint32_t)c + 64) {
                w = 0x80000000u | J[t];
            } else {
                w = (t - (uint32_t)c) << 2;  // (kept as the byte address ds_bpermute wants)
            }
        }
        while (__any((int32_t)w >= 0)) {
            const uint32_t tw = (uint32_t)__builtin_amdgcn_ds_bpermute((int)w, (int)w);  // (a resolved lane's read is not used)
            w = (int32_t)w >= 0 ? tw : w;
        }
        const uint32_t val = w & 0x7fffffffu;
        if (valid) J[r] = (uint16_t)val;
        wave_lds_fence();
    }
}
__global__ __launch_bounds__(256) void k_seg_exit(uint32_t n, uint32_t K, const uint16_t* __restrict__ adv,
                                                  uint16_t* __restrict__ X0, uint32_t seg0) {
    __shared__ uint16_t sJ[4][SEG];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t k = (uint64_t)blockIdx.x * 4 + wv + seg0;  // (seg0: a launch may cover a range of segments)
    if (k >= K) return;  // whole wave; no workgroup barrier is used below
    uint16_t* J = sJ[wv];
    const uint64_t a = k * SEG, b = a + SEG < n ? a + SEG : n;
    const uint32_t len = (uint32_t)(b - a);
    seg_sweep(adv, a, len, J, lane);
    uint16_t* x = X0 + k * ZONE;
    for (uint32_t e = lane; e < ZONE; e += 64) x[e] = e < len ? J[e] : (uint16_t)(e - len);
}
// k_level_up: compose FAN child tables into one parent table.
// A workgroup per parent unit, a thread per entry of its zone.  The FAN child tables (one contiguous piece of C) are
// staged in LDS first, every thread fetching its FAN entries at once: an entry's way through the children is a chain of
// FAN dependent reads, which from global memory was FAN memory latencies per launch (52 us at the widest level).
// Positions are 32-bit and relative to the child at hand.
__global__ __launch_bounds__(ZONE) void k_level_up(uint32_t n, uint32_t nc, uint64_t csize,
                                                   const uint16_t* __restrict__ C, uint32_t nu,
                                                   uint16_t* __restrict__ X) {
    __shared__ uint16_t sC[FAN][ZONE];
    const uint32_t u = blockIdx.x, e = threadIdx.x;
    const uint32_t c0 = u * FAN, c1 = c0 + FAN < nc ? c0 + FAN : nc;
    uint16_t v[FAN];
#pragma unroll
    for (uint32_t k = 0; k < FAN; k++) v[k] = c0 + k < c1 ? C[(uint64_t)(c0 + k) * ZONE + e] : (uint16_t)0;
#pragma unroll
    for (uint32_t k = 0; k < FAN; k++) sC[k][e] = v[k];
    __syncthreads();
    uint32_t rel = e;  // the entry's position, relative to the start of child k
    for (uint32_t k = 0; k < c1 - c0; k++) {
        const uint64_t cstart = (uint64_t)(c0 + k) * csize;
        const uint32_t sz = (uint64_t)n - cstart < csize ? (uint32_t)((uint64_t)n - cstart) : (uint32_t)csize;  // (csize <= 2^32: a level's units tile a 32-bit input)
        if (rel < sz) rel = sz + sC[k][rel];
        rel -= sz;
    }
    X[(uint64_t)u * ZONE + e] = (uint16_t)rel;  // relative to the end of the unit
}

// k_level_down: given the entry position of every parent unit, the entry of each child.
__global__ __launch_bounds__(64) void k_level_down(uint32_t n, uint32_t nc, uint64_t csize,
                                                   const uint16_t* __restrict__ C, uint32_t nu,
                                                   const uint32_t* __restrict__ Eparent, uint32_t* __restrict__ Echild) {
    uint64_t u = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (u >= nu) return;
    uint64_t pos = Eparent ? Eparent[u] : 0;
    uint64_t c1 = (u + 1) * FAN < nc ? (u + 1) * FAN : nc;
    for (uint64_t c = u * FAN; c < c1; c++) {
        uint64_t cstart = c * csize;
        uint64_t cend = cstart + csize < n ? cstart + csize : n;
        Echild[c] = (uint32_t)pos;
        if (pos < cend) pos = cend + C[c * ZONE + (pos - cstart)];
    }
}

// k_tree_top: the levels of the table tree that have at most TOP_UNITS units, up AND down, in ONE launch by one workgroup:
// their tables fit the LDS together, and as launches of their own each of them was a kernel of a few workgroups that
// took a launch's latency (five levels up and five down for 100 MB: ten launches, six of them for 27 units).  In:
// the tables of level `lt` (global, written by the last wide k_level_up) and the entry of the root unit; out: the entries of
// the units of every level from the root down to lt - 1 (the level below the staged ones: its tables are read from
// global memory, one entry per unit), and the tables above lt (the exit table of the whole range is the root's).
constexpr uint32_t TOP_UNITS = 64;
constexpr uint32_t TOP_LEVELS = 8;
struct TopLevels {
    uint32_t nl;                 // levels lt .. root
    uint32_t count[TOP_LEVELS];  // units per level
    uint64_t usize[TOP_LEVELS];  // positions per unit
    uint16_t* X[TOP_LEVELS];
    uint32_t* E[TOP_LEVELS];
    // the level below lt (count_below == 0: lt is level 0)
    uint32_t count_below;
    uint64_t usize_below;
    const uint16_t* X_below;
    uint32_t* E_below;
};
__global__ __launch_bounds__(1024) void k_tree_top(uint32_t n, TopLevels t) {
    __shared__ uint16_t sX[(TOP_UNITS + TOP_UNITS / FAN + 4) * ZONE];
    __shared__ uint32_t sE[TOP_UNITS + TOP_UNITS / FAN + 4];
    const uint32_t tid = threadIdx.x;
    uint32_t off[TOP_LEVELS + 1];
    off[0] = 0;
    for (uint32_t l = 0; l < t.nl; l++) off[l + 1] = off[l] + t.count[l];
    for (uint32_t i = tid; i < t.count[0] * ZONE; i += 1024) sX[i] = t.X[0][i];
    __syncthreads();
    // up: a unit's table = its children's tables composed (k_level_up)
    for (uint32_t l = 0; l + 1 < t.nl; l++) {
        const uint32_t nc = t.count[l], np = t.count[l + 1];
        const uint64_t csize = t.usize[l];
        for (uint32_t i = tid; i < np * ZONE; i += 1024) {
            const uint32_t u = i / ZONE, e = i % ZONE;
            const uint32_t c0 = u * FAN, c1 = c0 + FAN < nc ? c0 + FAN : nc;
            uint32_t rel = e;
            for (uint32_t c = c0; c < c1; c++) {
                const uint64_t cstart = (uint64_t)c * csize;
                const uint32_t sz = (uint64_t)n - cstart < csize ? (uint32_t)((uint64_t)n - cstart) : (uint32_t)csize;
                if (rel < sz) rel = sz + sX[(off[l] + c) * ZONE + rel];
                rel -= sz;
            }
            sX[(off[l + 1] + u) * ZONE + e] = (uint16_t)rel;
            t.X[l + 1][i] = (uint16_t)rel;
        }
        __syncthreads();
    }
    // down: the entry of every child from the entry of its parent (k_level_down)
    if (tid == 0) sE[off[t.nl - 1]] = t.E[t.nl - 1][0];
    __syncthreads();
    for (uint32_t l = t.nl - 1; l-- > 0;) {
        const uint32_t nc = t.count[l], np = t.count[l + 1];
        const uint64_t csize = t.usize[l];
        if (tid < np) {
            uint64_t pos = sE[off[l + 1] + tid];
            const uint32_t c1 = (tid + 1) * FAN < nc ? (tid + 1) * FAN : nc;
            for (uint32_t c = tid * FAN; c < c1; c++) {
                const uint64_t cstart = (uint64_t)c * csize;
                const uint64_t cend = cstart + csize < n ? cstart + csize : n;
                sE[off[l] + c] = (uint32_t)pos;
                t.E[l][c] = (uint32_t)pos;
                if (pos < cend) pos = cend + sX[(off[l] + c) * ZONE + (uint32_t)(pos - cstart)];
            }
        }
        __syncthreads();
    }
    if (t.count_below && tid < t.count[0]) {
        const uint32_t nc = t.count_below;
        const uint64_t csize = t.usize_below;
        uint64_t pos = sE[tid];
        const uint32_t c1 = (tid + 1) * FAN < nc ? (tid + 1) * FAN : nc;
        for (uint32_t c = tid * FAN; c < c1; c++) {
            const uint64_t cstart = (uint64_t)c * csize;
            const uint64_t cend = cstart + csize < n ? cstart + csize : n;
            t.E_below[c] = (uint32_t)pos;
            if (pos < cend) pos = cend + t.X_below[(uint64_t)c * ZONE + (pos - cstart)];
        }
    }
}

// the table with the two entries at and behind position j already in registers (positions relative to `g`)
struct NearM {
    const uint32_t* g;
    uint32_t j;
    uint32_t v0, v1;
    __device__ uint32_t operator()(uint32_t i) const { return i == j ? v0 : i == j + 1 ? v1 : g[i]; }
};
// ---------------------------------------------------------------------------------------------
// k_emit: walk each segment from its entry and write its tokens (output_writer.rs:47-65).
// ---------------------------------------------------------------------------------------------
// One wave per segment.  The chain from the segment's entry is serial, so it is made short first: all lanes
// turn adv[] into two-step and then four-step jumps (in LDS, stopping at the end of the segment), lane 0
// follows the four-step jumps and records every fourth path position, and every lane then replays the
// (up to) four steps behind one recorded position -- parse_step gives both the tokens and the way on -- a
// wave scan places the tokens.
// (Segments are relative to pos0: the sharded path parses a sub-range [pos0, pos0 + n) of a buffer of
// n_total bytes; the whole-buffer path has pos0 = 0, n = n_total.)
//
// SPEC: the segment's entry is not given but FOUND -- the wave starts SPEC_W positions in front of its segment, at a
// position that need not lie on the path at all, and follows the restart steps: paths merge for good as soon as they
// share one restart position (a step depends on nothing but its position), which on anything but long periodic data
// happens within a few dozen bytes.  The first restart position at or behind the segment's start is taken as its entry
// (written to E0), the parse goes on from there as usual, and where it leaves the segment is written to Xs: segment 0
// starts at the stream's true entry, so if every segment's entry equals the exit of the segment before it (k_scan_a
// checks) all of them are the true ones -- by induction, no probability involved -- and the exit tables of every
// segment, the table tree above them and the way down (k_seg_exit, k_level_up, k_tree_top, k_level_down) were not
// needed.  If one differs the host parses again the exact way.
// MODE 2, the repair of a handful of entries that did not check out (k_spec_check lists them; a seam of pieces of
// different kinds, a short periodic stretch): a wave per listed segment parses it again from where the segment before it
// was left, and goes on into the next segments for as long as its exit differs from their entry (at most FIX_HOPS of them,
// never into a segment that is listed itself: that one has a wave of its own).  What is still inconsistent afterwards --
// chains that met, long periodic data where every boundary fails -- is seen by the final check in k_scan_a.
constexpr uint32_t SPEC_W = 128;  // (measured with the repair in place, parse stage of the 100 MB text: 64 / 128 / 256 positions 0.59 / 0.58 / 0.63 ms;
                                  // the Silesia-like mix 1.64 / 1.33 / 1.40 ms -- at 64 a thousand of its entries need the repair)
constexpr uint32_t FIX_MAX = 1024;  // listed segments the repair takes on (more: the data is periodic at large, the exact parse is due)
constexpr uint32_t FIX_HOPS = 24;
constexpr uint32_t STEP_CHUNKS = 5;  // k_emit's steps from M: rounds of 256 positions of a wave
struct SpecFix {
    uint32_t* list;          // segments whose entry is not the exit of the segment before, each with that exit
    uint32_t* badmap;        // the same as a bit per segment
    const uint32_t* n;       // how many
};
// (the rows of a wave: STEP_CHUNKS * 256 + 8 entries where the steps are worked out here -- STEPS: they go through 256 positions
// at a time and look eight entries beyond a position --, else the positions the wave holds)
template <int MODE, bool STEPS>
struct EmitRows {
    static constexpr bool SPEC = MODE == 1;
    static constexpr uint32_t REG = SEG + (SPEC ? SPEC_W : 0u);  // positions a wave holds: its segment and the run-up in front of it
    static constexpr uint32_t ROW = STEPS ? STEP_CHUNKS * 256 + 8 : REG;
    static_assert(REG + 12 <= STEP_CHUNKS * 256 && ROW % 4 == 0, "the chunks cover the region and what a step looks at behind it");
};
// One wave, one segment (MODE 2: and the segments behind it while the repair goes on).  A, P: the wave's two rows in LDS;
// np_w, exit_w: a word each.  No workgroup barrier inside.
template <int MODE, bool STEPS>
__device__ __forceinline__ void emit_wave(const uint8_t* __restrict__ in, uint32_t n, uint32_t K,
                                          const uint32_t* __restrict__ M, const uint32_t* __restrict__ Mq,
                                          const ParseCfg& cfg, const uint16_t* __restrict__ adv,
                                          uint32_t* E0, uint32_t* __restrict__ tokbuf,
                                          uint32_t* cnt, uint32_t pos0, uint32_t n_total, const SegEnds& sg,
                                          uint32_t* Xs, const uint32_t* badmap, uint32_t runup0,
                                          uint16_t* A, uint16_t* P, uint32_t* np_w, uint32_t* exit_w, uint64_t k, uint32_t given,
                                          uint32_t lane) {
    constexpr bool SPEC = MODE == 1;
    constexpr uint32_t REG = EmitRows<MODE, STEPS>::REG;
#define EMIT_NP (*np_w)
#define EMIT_EXIT (*exit_w)
#define EMIT_BADMAP badmap
#include "emit_body.inc"
#undef EMIT_NP
#undef EMIT_EXIT
#undef EMIT_BADMAP
}

// (STEPS: the wave works the restart steps out itself, from M -- adv is not read; else they come from k_adv / k_rle through adv)
template <int MODE, bool STEPS>
__global__ __launch_bounds__(256) void k_emit(const uint8_t* __restrict__ in, uint32_t n, uint32_t K,
                                              const uint32_t* __restrict__ M, const uint32_t* __restrict__ Mq,
                                              ParseCfg cfg, const uint16_t* __restrict__ adv,
                                              uint32_t* __restrict__ E0, uint32_t* __restrict__ tokbuf,
                                              uint32_t* __restrict__ cnt, uint32_t pos0, uint32_t n_total, SegEnds sg,
                                              uint32_t* __restrict__ Xs, SpecFix fix, uint32_t runup0, uint32_t seg0) {
    constexpr uint32_t ROW = EmitRows<MODE, STEPS>::ROW;
    __shared__ __attribute__((aligned(8))) uint16_t s_adv[4][ROW];
    __shared__ __attribute__((aligned(8))) uint16_t s_pp[4][ROW];
    __shared__ uint32_t s_np[4], s_exit[4];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint64_t k = (uint64_t)blockIdx.x * 4 + wv + seg0;  // (seg0: a launch may cover a range of segments)
    uint32_t given = 0;  // MODE 2: the entry the segment is parsed from
    if (MODE == 2) {
        const uint32_t nf = *fix.n;
        if (nf > FIX_MAX || k >= nf) return;
        // (the exit of the segment before it as k_spec_check saw it -- not Xs[k - 1] as it is now, which a wave that repairs the
        // segments before this one may be rewriting: what a repair is based on must not depend on which wave runs first)
        given = fix.list[2 * k + 1];
        k = fix.list[2 * k];
    }
    if (k >= K) return;  // whole wave; no workgroup barrier is used below
    constexpr bool SPEC = MODE == 1;
    constexpr uint32_t REG = EmitRows<MODE, STEPS>::REG;
    uint16_t* A = s_adv[wv];
    uint16_t* P = s_pp[wv];
#define EMIT_NP s_np[wv]
#define EMIT_EXIT s_exit[wv]
#define EMIT_BADMAP fix.badmap
#include "emit_body.inc"
#undef EMIT_NP
#undef EMIT_EXIT
#undef EMIT_BADMAP
}

// Quirk Q1 (lz77.rs:628-638): block 0 holds 31744 tokens and its last token -- tk, at tp < WINDOW_SIZE -- leaves the encoder inside
// the first window: two hashes are re-warmed at *wpos and the tables change.  One predicate for the host (which runs the call
// again) and the device (which stops the first pass's block stages).
__host__ __device__ inline bool q1_rewarm(uint32_t tk, uint64_t tp, uint32_t mode, uint64_t n, uint64_t* wpos) {
    uint64_t lp;
    if (mode == MODE_LAZY) {
        lp = tp + 1;
        if (tk >> 16)
            *wpos = tp + tok_cover(tk);
        else
            *wpos = ((tp + 1) + 2 < n) ? tp + 2 : tp + 1;
    } else {
        lp = tp;
        *wpos = tp + tok_cover(tk);
    }
    return lp < WINDOW_SIZE && *wpos <= WINDOW_SIZE;
}
// segment i was entered somewhere else than segment i - 1 was left, and segment i - 1 was not (a run's head)
__device__ __forceinline__ bool spec_run_head(uint32_t i, uint32_t K, const uint32_t* E0, const uint32_t* Xs) {
    if (i == 0 || i >= K || E0[i] == Xs[i - 1]) return false;
    return i == 1 || E0[i - 1] == Xs[i - 2];
}
// k_spec_check: after the speculative k_emit, which segments were entered somewhere else than the segment before them was
// left?  A bit per segment and a list (in no particular order; segment and the exit before it) for the repair; sc->n_fix counts them.
// (lo: the first segment of the range looked at -- a multiple of 64 -- K its end)
__global__ __launch_bounds__(256) void k_spec_check(uint32_t K, const uint32_t* __restrict__ E0, const uint32_t* __restrict__ Xs,
                                                    uint32_t* __restrict__ badmap, uint32_t* __restrict__ list, uint32_t* __restrict__ n_fix,
                                                    uint32_t lo) {
    const uint32_t i = lo + blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    // (listed: the FIRST of a run of such segments -- a periodic stretch of a few kilobytes fails at every boundary inside it, and
    // only its first segment's predecessor was left where the check sees it: the repair's wave goes on through the rest of the
    // run, hop by hop.  With a wave per failed boundary, each parsing from an exit that was itself wrong, a zero run of 1.5 KB
    // somewhere in the input sent the whole call to the exact parse.)
    const bool off = spec_run_head(i, K, E0, Xs);
    const uint64_t m = __builtin_amdgcn_ballot_w64(off);
    if (lane == 0) {
        badmap[(i >> 5)] = (uint32_t)m;
        badmap[(i >> 5) + 1] = (uint32_t)(m >> 32);
    }
    if (m == 0) return;
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(n_fix, (uint32_t)__popcll(m));
    at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (off && at + rank < FIX_MAX) {
        list[2 * (at + rank)] = i;
        list[2 * (at + rank) + 1] = Xs[i - 1];  // what the repair parses the segment from
    }
}

// k_scan_a / k_scan_b: exclusive scan of the per-segment token counts.  A workgroup takes 1024
// counts: (a) its sum; (b) the sums of the workgroups before it, added to a scan of its own counts
// (wave scans in registers, the sixteen wave totals through LDS).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(v, off, 64);
        if (lane >= (uint32_t)off) v += y;
    }
    return v;
}
// (E0 / Xs / spec_bad: after a speculative k_emit -- every segment's entry must be the exit of the segment before it)
__global__ __launch_bounds__(1024) void k_scan_a(uint32_t K, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ part,
                                                 const uint32_t* __restrict__ E0, const uint32_t* __restrict__ Xs,
                                                 uint32_t* __restrict__ spec_bad, uint32_t lo) {
    __shared__ uint32_t wtot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t i = lo + blockIdx.x * 1024 + tid;
    if (Xs) {
        const bool off = i > 0 && i < K && E0[i] != Xs[i - 1];
        const uint64_t offm = __builtin_amdgcn_ballot_w64(off);
        if (offm != 0 && lane == 0) atomicAdd(spec_bad, (uint32_t)__popcll(offm));
    }
    uint32_t v = i < K ? cnt[i] : 0;
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) wtot[wv] = v;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (int k = 0; k < 16; k++) t += wtot[k];
        part[blockIdx.x] = t;
    }
}
// (tend / pb: a one-shot call has one segment, whose token count and block count are the totals -- written here, and
// k_seg_tokens / k_block_count are not launched)
// (pc: the segments [pc.seg_lo, K) are one piece of the stream; the tokens before it are sc->Tcum[pc.p])
__global__ __launch_bounds__(1024) void k_scan_b(uint32_t K, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ part,
                                                 uint32_t* __restrict__ base, DevScalars* sc, uint32_t* __restrict__ tend,
                                                 uint32_t* __restrict__ pb, Piece pc) {
    __shared__ uint32_t wtot[16], red[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // sum of the workgroups before this one
    uint32_t pre = 0;
    for (uint32_t j = tid; j < blockIdx.x; j += 1024) pre += part[j];
#pragma unroll
    for (int off = 32; off; off >>= 1) pre += __shfl_xor(pre, off, 64);
    if (lane == 0) red[wv] = pre;
    const uint32_t i = pc.seg_lo + blockIdx.x * 1024 + tid;
    const uint32_t v = i < K ? cnt[i] : 0;
    const uint32_t x = wave_incl_scan(v, lane);
    if (lane == 63) wtot[wv] = x;
    __syncthreads();
    uint32_t add = sc->Tcum[pc.p], all = 0;
    for (uint32_t k = 0; k < 16; k++) {
        add += red[k];
        add += k < wv ? wtot[k] : 0;
        all += wtot[k];
    }
    if (i < K) base[i] = add + x - v;
    if (tid == 0 && blockIdx.x == gridDim.x - 1) {
        uint32_t T = sc->Tcum[pc.p];
        for (uint32_t k = 0; k < 16; k++) T += red[k];
        T += all;
        // (the blocks that are complete with this piece; the stream's last block may be short or empty)
        const uint32_t nb = T / MAX_BUFFER_LENGTH + (pc.last ? 1u : 0u);
        sc->T = T;
        sc->nb = nb;
        sc->Tcum[pc.p + 1] = T;
        sc->nbcum[pc.p + 1] = nb;
        if (tend) {
            tend[0] = T;
            pb[0] = 0;
            pb[1] = nb;
        }
    }
}
// The call's scalars for the host, written by the device into page-locked host memory: the runtime's copy of 240 bytes was a
// 3.5 us kernel of its own behind 5.8 us of idle queue.
__global__ __launch_bounds__(64) void k_state_out(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t words) {
    for (uint32_t i = threadIdx.x; i < words; i += 64) dst[i] = src[i];
}
// K == 0: no tokens
__global__ void k_scan_zero(DevScalars* sc) {
    if (threadIdx.x || blockIdx.x) return;
    sc->T = 0;
    sc->nb = 1;
}

// k_compact: tokens of all segments into one dense stream.
__global__ __launch_bounds__(256) void k_compact(uint32_t K, const uint32_t* __restrict__ tokbuf,
                                                 const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ base,
                                                 uint32_t* __restrict__ dtok, const DevScalars* sc, uint32_t seg0) {
    uint64_t k = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6) + seg0;
    if (k >= K || spec_failed(sc)) return;
    uint32_t lane = threadIdx.x & 63;
    uint32_t c = cnt[k], b = base[k];
    const uint32_t* src = tokbuf + k * SEG;
    // (sixteen bytes a lane -- the slot is 4 KiB aligned, the dense array takes them at any dword --: with four bytes a lane the copy
    // ran at 4.4 TB/s and looked bound by memory; 53 -> 38 us)
    const uint32_t c4 = c & ~3u;
    for (uint32_t i = 4 * lane; i < c4; i += 256) *reinterpret_cast<uint4*>(dtok + b + i) = *reinterpret_cast<const uint4*>(src + i);
    if (lane < (c & 3u)) dtok[b + c4 + lane] = src[c4 + lane];
}

// start position of token t (t < T); the whole wave calls it.  The segment that holds the token -- the last k with
// base[k] <= t -- by a search that looks at 64 places per round (three rounds for 2^18 segments: a thread's binary
// search was seventeen memory latencies in a row), the covers of the segment's tokens before t summed across the lanes.
__device__ uint32_t token_start(uint32_t t, uint32_t K, const uint32_t* base, const uint32_t* E0,
                                const uint32_t* tokbuf, uint32_t lane, uint32_t* seg = nullptr) {
    uint32_t lo = 0, hi = K;
    while (hi - lo > 1) {
        const uint32_t span1 = hi - lo - 1;
        const uint32_t p = lo + 1 + (uint32_t)(((uint64_t)span1 * lane) >> 6);  // ascending with the lane, inside (lo, hi)
        const uint32_t c = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(base[p] <= t));  // (true for a prefix of the lanes)
        const uint32_t nlo = c ? lo + 1 + (uint32_t)(((uint64_t)span1 * (c - 1)) >> 6) : lo;
        const uint32_t nhi = c < 64 ? lo + 1 + (uint32_t)(((uint64_t)span1 * c) >> 6) : hi;
        lo = nlo;
        hi = nhi;
    }
    if (seg) *seg = lo;
    const uint32_t m = t - base[lo];
    const uint32_t* tk = tokbuf + (uint64_t)lo * SEG;
    uint32_t cov = 0;
    // (a segment's tokens before t -- up to 1023 on noise --, fetched together: a load a round was a memory latency a round,
    // sixteen in a row where every byte is a token)
    uint32_t tv[SEG / 64];
#pragma unroll
    for (uint32_t r = 0; r < SEG / 64; r++) tv[r] = lane + 64 * r < m ? tk[lane + 64 * r] : 0u;
#pragma unroll
    for (uint32_t r = 0; r < SEG / 64; r++) cov += lane + 64 * r < m ? tok_cover(tv[r]) : 0u;
#pragma unroll
    for (int off = 32; off; off >>= 1) cov += __shfl_xor(cov, off, 64);
    return E0[lo] + cov;
}

// ---------------------------------------------------------------------------------------------
// Block table.  Blocks are "every 31744 tokens" (output_writer.rs:19,38-44) inside each segment of
// the input (a sync flush ends the current block, compress.rs:256-261); a segment whose token count
// is a multiple of 31744 -- including an empty one -- still ends with one (empty) block, because the
// reference only learns that the data is over on the next call (lz77.rs:709-742, A.4 Q8).
//   k_seg_tokens   tokens that start before each segment end (the parse passes through every end)
//   k_block_count  blocks per segment, prefix sum, total
//   k_block_bounds per block: token range, start position in the input, sync-marker flag, the data
//                  for the Q1 decision and the Q13 condition
// ---------------------------------------------------------------------------------------------
struct BlockTab {
    uint32_t* t0;    // first token
    uint32_t* nt;    // token count (0..31744)
    uint32_t* sync;  // 1 = the empty stored block 00 00 FF FF follows (compress.rs:258-261)
};

__global__ __launch_bounds__(64) void k_seg_tokens(SegEnds sg, uint32_t K, const uint32_t* __restrict__ base,
                                                   const uint32_t* __restrict__ E0, const uint32_t* __restrict__ tokbuf,
                                                   const uint32_t* __restrict__ cnt, const DevScalars* sc,
                                                   uint32_t* __restrict__ tend) {
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= sg.m) return;
    uint32_t e = sg.ends[i];
    if (i + 1 == sg.m || K == 0) {
        tend[i] = sc->T;
        return;
    }
    uint32_t k = e / SEG;
    if (k >= K) {
        tend[i] = sc->T;
        return;
    }
    uint32_t pos = E0[k], c = 0, nk = cnt[k];
    const uint32_t* tk = tokbuf + (uint64_t)k * SEG;
    while (pos < e && c < nk) pos += tok_cover(tk[c++]);
    tend[i] = base[k] + c;
}

__global__ void k_block_count(SegEnds sg, const uint32_t* __restrict__ tend, uint32_t* __restrict__ pb, DevScalars* sc) {
    if (threadIdx.x || blockIdx.x) return;
    uint32_t acc = 0, prev = 0;
    for (uint32_t i = 0; i < sg.m; i++) {
        pb[i] = acc;
        acc += (tend[i] - prev) / MAX_BUFFER_LENGTH + 1;
        prev = tend[i];
    }
    pb[sg.m] = acc;
    sc->nb = acc;
}

// (a wave per block: token_start is a wave's work)
__global__ __launch_bounds__(256) void k_block_bounds(uint32_t n, uint32_t K, uint32_t nb_max, uint32_t mode, SegEnds sg,
                                                      uint32_t sync_final, const uint32_t* __restrict__ tend,
                                                      const uint32_t* __restrict__ pb, const uint32_t* __restrict__ base,
                                                      const uint32_t* __restrict__ E0, const uint32_t* __restrict__ tokbuf,
                                                      const uint32_t* __restrict__ dtok, DevScalars* sc,
                                                      uint32_t* __restrict__ bstart, uint32_t* __restrict__ q13,
                                                      BlockTab tab, Piece pc, const uint32_t* __restrict__ xs_last, uint32_t q1_cancel) {
    // (a piece of a stream that is still arriving: the blocks that became complete with it, sc->nbcum[pc.p] .. sc->nb - 1)
    const uint32_t b = sc->nbcum[pc.p] + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b > nb_max) return;
    const uint32_t nb = sc->nb;
    if (b > nb) return;
    if (b == nb) {
        // where the block behind the last one begins: the end of the data -- or, while the stream goes on, where the first
        // token that is not in a complete block starts (behind the last token so far if there is none yet: xs_last)
        uint32_t bs = n;
        if (!pc.last) {
            const uint32_t t = nb * (uint32_t)MAX_BUFFER_LENGTH;
            bs = t < sc->T ? token_start(t, K, base, E0, tokbuf, lane) : *xs_last;
        }
        if (lane == 0) bstart[b] = bs;
        return;
    }
    uint32_t lo = 0, hi = sg.m;  // segment i with pb[i] <= b < pb[i+1]
    while (hi - lo > 1) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (pb[mid] <= b)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t i = lo, j = b - pb[i], nbi = pb[i + 1] - pb[i];
    const uint32_t tprev = i ? tend[i - 1] : 0;
    const uint32_t t0 = tprev + j * MAX_BUFFER_LENGTH;
    const uint32_t left = tend[i] - t0;
    const uint32_t nt = left < MAX_BUFFER_LENGTH ? left : (uint32_t)MAX_BUFFER_LENGTH;
    const uint32_t bs = nt ? token_start(t0, K, base, E0, tokbuf, lane) : sg.ends[i];
    uint32_t flag = 0;
    if (nt == MAX_BUFFER_LENGTH) {  // a full block: look at its last token
        uint32_t t1 = t0 + MAX_BUFFER_LENGTH - 1;
        uint32_t tk = dtok[t1];
        uint32_t tp = token_start(t1, K, base, E0, tokbuf, lane);
        if (tp < WINDOW_SIZE && lane == 0) {  // the one block that can fill inside the first window (Q1, lz77.rs:628-638)
            sc->b0_full = 1;
            sc->b0_last_tok = tk;
            sc->b0_last_pos = tp;
            uint64_t wpos;
            if (q1_cancel && q1_rewarm(tk, tp, mode, n, &wpos)) sc->q1_cancel = 1;
        }
        if (tk >> 16) {  // lz77.rs:679-695
            uint64_t lp = (mode == MODE_LAZY) ? (uint64_t)tp + 1 : tp;
            uint64_t wdx = lp / WINDOW_SIZE;
            uint64_t wend = (wdx + 1) * (uint64_t)WINDOW_SIZE;
            uint64_t mend = (uint64_t)tp + tok_cover(tk);
            if (wdx >= 1 && mend > wend) {
                uint64_t buf_end = wdx * (uint64_t)WINDOW_SIZE + 65794;
                uint64_t have = sg.ends[i];  // the data the encoder held when this block ended
                if (buf_end > have) buf_end = have;
                flag = (mend + WINDOW_SIZE > buf_end) ? 2u : 1u;
            }
        }
    }
    if (lane == 0) {
        tab.t0[b] = t0;
        tab.nt[b] = nt;
        tab.sync[b] = (j + 1 == nbi && (i + 1 < sg.m || sync_final)) ? 1u : 0u;
        bstart[b] = bs;
        q13[b] = flag;
    }
}

// k_small_fix: what lies between the speculative parse and the dense tokens of a one-shot call (one segment end) of at most
// 2048 token segments -- 2 MiB -- in ONE workgroup (beyond that the block table, a wave a block, is better off with workgroups of its own): k_spec_check, the repair (k_emit<2>), k_scan_a, k_scan_b and k_block_bounds
// were five launches of 4.5 us each, of one to four workgroups, with (on anything but periodic data) nothing to repair.
//   1  which segments were entered somewhere else than the one before them was left: the bits and the list of k_spec_check;
//   2  a wave per listed segment parses it again from where the segment before it was left (emit_wave<2>, as k_emit<2>);
//   3  the chain of entries and exits once more (what is still off fails the call's speculation: spec_bad), the scan of
//      the token counts, the block table.  The last token of a full block is read from its segment's slot: the dense
//      array is made by k_compact, behind this kernel.
// (Xs == nullptr: the exact parse -- entries from the table tree -- has nothing to check or repair: step 3 only.)
#ifndef MI355_SMALL_TAIL
#define MI355_SMALL_TAIL 1
#endif
#ifndef MI355_SMALL_SEGS
#define MI355_SMALL_SEGS 2048
#endif
constexpr uint32_t SMALL_TAIL_SEGS = MI355_SMALL_SEGS, SMALL_FIX_T = 512, SMALL_FIX_PER = SMALL_TAIL_SEGS / SMALL_FIX_T;  // (2 MiB; segments a thread scans)
template <bool STEPS>
__global__ __launch_bounds__(SMALL_FIX_T) void k_small_fix(const uint8_t* __restrict__ in, uint32_t n, uint32_t K,
                                                           const uint32_t* __restrict__ M, const uint32_t* __restrict__ Mq, ParseCfg cfg,
                                                           const uint16_t* __restrict__ adv, uint32_t* E0, uint32_t* __restrict__ tokbuf,
                                                           uint32_t* cnt, SegEnds sg, uint32_t* Xs, uint32_t* badmap, uint32_t* list,
                                                           uint32_t nb_max, uint32_t sync_final, uint32_t* __restrict__ spec_bad,
                                                           uint32_t* __restrict__ base, DevScalars* sc, uint32_t* __restrict__ tend,
                                                           uint32_t* __restrict__ pb, uint32_t* __restrict__ bstart,
                                                           uint32_t* __restrict__ q13, BlockTab tab, uint32_t q1_cancel) {
    constexpr uint32_t ROW = EmitRows<2, STEPS>::ROW, NW = SMALL_FIX_T / 64;
    __shared__ __attribute__((aligned(8))) uint16_t s_adv[NW][ROW];
    __shared__ __attribute__((aligned(8))) uint16_t s_pp[NW][ROW];
    __shared__ uint32_t s_np[NW], s_exit[NW], wtot[NW], wbad[NW], s_nfix, s_base[SMALL_TAIL_SEGS];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (Xs) {
        // ---- 1: the check (k_spec_check) ----
        if (tid == 0) s_nfix = 0;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < SMALL_TAIL_SEGS; i0 += SMALL_FIX_T) {
            const uint32_t i = i0 + tid;
            if ((i & ~63u) >= K) break;  // (whole waves)
            const bool off = spec_run_head(i, K, E0, Xs);  // (the first of a run of them: k_spec_check)
            const uint64_t m = __builtin_amdgcn_ballot_w64(off);
            if (lane == 0) {
                badmap[i >> 5] = (uint32_t)m;
                badmap[(i >> 5) + 1] = (uint32_t)(m >> 32);
            }
            if (m == 0) continue;
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&s_nfix, (uint32_t)__popcll(m));
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (off && at + rank < FIX_MAX) {
                list[2 * (at + rank)] = i;
                list[2 * (at + rank) + 1] = Xs[i - 1];  // what the repair parses the segment from
            }
        }
        __syncthreads();
        const uint32_t nf = s_nfix;
        if (tid == 0) sc->n_fix[0] = nf;
        // ---- 2: the repair (k_emit<2>) ----
        if (nf != 0 && nf <= FIX_MAX) {
            for (uint32_t u = wv; u < nf; u += NW) {
                const uint32_t k = list[2 * u], given = list[2 * u + 1];
                if (k < K)
                    emit_wave<2, STEPS>(in, n, K, M, Mq, cfg, adv, E0, tokbuf, cnt, 0u, n, sg, Xs, badmap, 0u, s_adv[wv], s_pp[wv], &s_np[wv],
                                        &s_exit[wv], k, given, lane);
            }
        }
        __syncthreads();
    }
    // ---- 3: the chain of entries and exits (k_scan_a), the scan of the token counts (k_scan_b): a thread takes `per`
    // consecutive segments (two for a call of up to 1 MiB, four for 2 MiB) ----
    const uint32_t per = (K + SMALL_FIX_T - 1) / SMALL_FIX_T;  // <= SMALL_FIX_PER
    const uint32_t s0 = tid * per;
    uint32_t nbad = 0, v = 0;
    uint32_t cs[SMALL_FIX_PER];
#pragma unroll
    for (uint32_t q = 0; q < SMALL_FIX_PER; q++) {
        const uint32_t i = s0 + q;
        const bool in = q < per && i < K;
        if (Xs) nbad += (in && i > 0 && E0[i] != Xs[i - 1]) ? 1u : 0u;
        cs[q] = in ? cnt[i] : 0u;
        v += cs[q];
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) nbad += __shfl_xor(nbad, off, 64);
    const uint32_t x = wave_incl_scan(v, lane);
    if (lane == 63) wtot[wv] = x;
    if (lane == 0) wbad[wv] = nbad;
    __syncthreads();
    const uint32_t T0 = sc->Tcum[0];
    uint32_t add = T0, all = 0, bad = 0;
    for (uint32_t k = 0; k < NW; k++) {
        add += k < wv ? wtot[k] : 0;
        all += wtot[k];
        bad += wbad[k];
    }
    uint32_t run = add + x - v;
#pragma unroll
    for (uint32_t q = 0; q < SMALL_FIX_PER; q++) {
        const uint32_t i = s0 + q;
        if (q < per && i < K) {
            s_base[i] = run;
            base[i] = run;
        }
        run += cs[q];
    }
    const uint32_t T = T0 + all, nb = T / (uint32_t)MAX_BUFFER_LENGTH + 1u;
    if (tid == 0) {
        if (bad) atomicAdd(spec_bad, bad);
        sc->T = T;
        sc->nb = nb;
        sc->Tcum[1] = T;
        sc->nbcum[1] = nb;
        tend[0] = T;
        pb[0] = 0;
        pb[1] = nb;
    }
    __syncthreads();
    // ---- k_block_bounds: a wave per block, and one for the end behind the last (sc->nbcum[0] is 0: one piece) ----
    for (uint32_t b = wv; b <= nb && b <= nb_max; b += NW) {
        if (b == nb) {
            if (lane == 0) bstart[b] = n;
            continue;
        }
        const uint32_t t0 = b * (uint32_t)MAX_BUFFER_LENGTH;
        const uint32_t left = T - t0;
        const uint32_t nt = left < MAX_BUFFER_LENGTH ? left : (uint32_t)MAX_BUFFER_LENGTH;
        const uint32_t bs = nt ? token_start(t0, K, s_base, E0, tokbuf, lane) : sg.ends[0];  // (the bases from LDS: this workgroup wrote them)
        uint32_t flag = 0;
        if (nt == MAX_BUFFER_LENGTH) {  // a full block: look at its last token
            const uint32_t t1 = t0 + MAX_BUFFER_LENGTH - 1;
            uint32_t seg = 0;
            const uint32_t tp = token_start(t1, K, s_base, E0, tokbuf, lane, &seg);
            const uint32_t tk = tokbuf[(uint64_t)seg * SEG + (t1 - s_base[seg])];
            if (tp < WINDOW_SIZE && lane == 0) {  // (Q1, lz77.rs:628-638)
                sc->b0_full = 1;
                sc->b0_last_tok = tk;
                sc->b0_last_pos = tp;
                uint64_t wpos;
                if (q1_cancel && q1_rewarm(tk, tp, cfg.mode, n, &wpos)) sc->q1_cancel = 1;
            }
            if (tk >> 16) {  // lz77.rs:679-695
                const uint64_t lp = (cfg.mode == MODE_LAZY) ? (uint64_t)tp + 1 : tp;
                const uint64_t wdx = lp / WINDOW_SIZE;
                const uint64_t wend = (wdx + 1) * (uint64_t)WINDOW_SIZE;
                const uint64_t mend = (uint64_t)tp + tok_cover(tk);
                if (wdx >= 1 && mend > wend) {
                    uint64_t buf_end = wdx * (uint64_t)WINDOW_SIZE + 65794;
                    const uint64_t have = sg.ends[0];
                    if (buf_end > have) buf_end = have;
                    flag = (mend + WINDOW_SIZE > buf_end) ? 2u : 1u;
                }
            }
        }
        if (lane == 0) {
            tab.t0[b] = t0;
            tab.nt[b] = nt;
            tab.sync[b] = (b + 1 == nb && sync_final) ? 1u : 0u;
            bstart[b] = bs;
            q13[b] = flag;
        }
    }
}

// the implicit table of the sharded path: blocks of 31744 tokens, no sync markers
__global__ __launch_bounds__(256) void k_block_table_uniform(uint64_t T2, uint32_t nb2, BlockTab tab) {
    uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb2) return;
    uint64_t t0 = (uint64_t)b * MAX_BUFFER_LENGTH;
    uint64_t left = T2 > t0 ? T2 - t0 : 0;
    tab.t0[b] = (uint32_t)t0;
    tab.nt[b] = left < MAX_BUFFER_LENGTH ? (uint32_t)left : (uint32_t)MAX_BUFFER_LENGTH;
    tab.sync[b] = 0;
}

// ---------------------------------------------------------------------------------------------
// k_block_hist: output_writer.rs:47-65,75-85 -- literal/length and distance frequencies, reduced in
// LDS.  A block's tokens are cut into PSPLIT parts of PQ tokens; every part gets a workgroup and its own
// histogram (the end-of-block symbol, output_writer.rs:83, is added by the reader): the block's
// frequencies are the sum of the parts, and the parts' sums of code lengths tell k_pack where each
// part's bits begin.
// ---------------------------------------------------------------------------------------------
#ifndef MI355_PACK_SPLIT
#define MI355_PACK_SPLIT 4
#endif
constexpr uint32_t PSPLIT = MI355_PACK_SPLIT;
constexpr uint32_t PQ = MAX_BUFFER_LENGTH / PSPLIT;  // tokens of a part
static_assert(MAX_BUFFER_LENGTH % PSPLIT == 0, "parts of equal size");

// (HT threads a workgroup: 256, or 1024 for a call with fewer quarter blocks than compute units -- a quarter block in one round of loads)
template <uint32_t HT>
__global__ __launch_bounds__(HT) void k_block_hist(const uint32_t* __restrict__ dtok, const DevScalars* sc,
                                                    uint32_t* __restrict__ ll_freq, uint32_t* __restrict__ d_freq,
                                                    BlockTab tab, uint32_t piece, uint32_t* __restrict__ clear, uint32_t clear_words) {
    __shared__ uint32_t h[320];
    // (a small call's output buffer is cleared here -- k_pack ORs its bits in -- instead of by a fill of its own in front of
    // k_plan: two launches of the runtime on the way of a 0.3 ms call)
    if (clear)
        for (uint32_t i = blockIdx.x * HT + threadIdx.x; i < clear_words; i += gridDim.x * HT) clear[i] = 0;
    const uint32_t b = sc->nbcum[piece] + blockIdx.x / PSPLIT, q = blockIdx.x % PSPLIT;
    if (b >= sc->nb || spec_failed(sc)) return;
    for (uint32_t i = threadIdx.x; i < 320; i += HT) h[i] = 0;
    __syncthreads();
    const uint32_t nt = tab.nt[b];
    const uint64_t t0 = (uint64_t)tab.t0[b] + (uint64_t)q * PQ;
    const uint64_t t1 = (uint64_t)tab.t0[b] + ((q + 1) * PQ < nt ? (q + 1) * PQ : nt);
    // (eight tokens a thread fetched together: one load per round in front of its LDS atomics was a memory latency per
    // round, thirty-one in a row -- the kernel's whole time)
    constexpr uint32_t HB = 8;
    for (uint64_t tb = t0 + threadIdx.x; tb < t1; tb += HT * HB) {
        uint32_t tks[HB];
#pragma unroll
        for (uint32_t k = 0; k < HB; k++) tks[k] = tb + HT * k < t1 ? dtok[tb + HT * k] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < HB; k++) {
            if (tb + HT * k >= t1) break;
            const uint32_t tk = tks[k];
            if (tk >> 16) {
                uint32_t c, eb, ev;
                length_symbol(tk & 0xff, &c, &eb, &ev);
                atomicAdd(&h[257 + c], 1u);
                distance_symbol(tk >> 16, &c, &eb, &ev);
                atomicAdd(&h[288 + c], 1u);
            } else {
                atomicAdd(&h[tk & 0xff], 1u);
            }
        }
    }
    __syncthreads();
    const uint64_t slot = (uint64_t)b * PSPLIT + q;
    for (uint32_t i = threadIdx.x; i < 288; i += HT) ll_freq[slot * 288 + i] = i < NUM_LL ? h[i] : 0;
    for (uint32_t i = threadIdx.x; i < 32; i += HT) d_freq[slot * 32 + i] = i < NUM_DIST ? h[288 + i] : 0;
}

// ---------------------------------------------------------------------------------------------
// k_block_header: huffman_lengths.rs:167-266 for one block per wave: three length-limited Huffman codes
// (length_encode.rs:347-415), the run-length coded table (length_encode.rs:82-155) and the cost figures.
// All blocks run at once, so the kernel takes as long as one block does: what can be spread over the
// wave is (stages.h huff_lengths_sorted is the serial form the host twin runs) --
//   used symbols: ballot compaction; sort: 64-lane rank sort on (freq << 9 | symbol), which equals the
//   reference's stable sort by freq; Moffat-Katajainen phase 1 (the two-queue merge, :218-247): one lane,
//   the heads of both queues in registers; depths of the internal nodes (:249-252): pointer jumping over
//   (depth, parent) words instead of the serial sweep; leaves per depth (:253-278): from the histogram of
//   the internal depths, level by level; limiter: limit_code_lengths on the LDS histogram; hand-out
//   (:402-408): every lane finds the length of its rank by a walk over the 15 counts; costs: a wave sum.
// Local arrays with a dynamic index would live in scratch memory (one round trip to L2 per access): all
// tables are in LDS.
// ---------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) HuffScratch {  // what one wave needs to build one code
    uint32_t key[292];   // freq << 9 | symbol of the used symbols (read four at a time: padded behind the last)
    uint32_t val[288];   // sorted frequencies -> Moffat-Katajainen working array
    uint32_t sym[288];   // symbol of every sorted leaf
    uint32_t pj[288];    // depth << 16 | ancestor of every internal node
    uint32_t icnt[288];  // internal nodes per depth
    uint32_t num[40];    // leaves per depth (num_codes[33])
};
struct HdrLds {
    uint32_t llf[288];
    uint32_t df[32];
    uint32_t clf[20];
    HuffScratch w[2];    // wave 0: literal/length code, then the code-length code; wave 1: distance code
    uint8_t ll_len[288];
    uint8_t d_len[32];
    uint8_t cl_len[20];
    uint8_t chain[320];
    uint16_t enc[320];
    uint32_t n_ll, n_d, n_enc, used;
};

// Moffat-Katajainen phase 1 on one lane: val[0..n) ascending; afterwards val[t] = parent of internal node t
// for t < n - 2 (length_encode.rs:218-247).  The two items a round pairs are the two smallest of the next two of the
// internal-node queue (r0, r1) and the next two leaves (l0, l1): all four are fetched before the round looks at any of them
// and the round itself has no branch -- a queue that has run out (no leaf left; no internal node made yet that is not taken)
// reads as the largest value, which is what the reference's conditions `leaf >= n ||` and `root < next &&` come to: ties go
// to the leaf, an internal node is made in every round, so the queue is never empty when a round begins.  (Before: the next
// weight of whichever queue a pick took from was read when the pick was made -- two dependent LDS reads and two branches a
// round, 474 cycles a round on the one lane, half of k_block_header on a block of text.)
#ifndef MI355_MK_WINDOW
#define MI355_MK_WINDOW 1
#endif
__device__ void mk_phase1(uint32_t* val, uint32_t n) {
    constexpr uint32_t INF = 0xFFFFFFFFu;
#if MI355_MK_WINDOW
    uint32_t root = 0, leaf = 2;
    uint32_t r0 = val[0] + val[1], r1 = INF;
    val[0] = r0;
    uint32_t l0 = leaf < n ? val[leaf] : INF, l1 = leaf + 1 < n ? val[leaf + 1] : INF;
    for (uint32_t next = 1; next + 1 < n; next++) {
        const bool t1 = r0 < l0;  // first pick: the internal node unless the leaf is at most as heavy
        const uint32_t a = t1 ? r0 : l0;
        const uint32_t rr = t1 ? r1 : r0, ll = t1 ? l0 : l1;
        const bool t2 = rr < ll;  // second pick
        const uint32_t v = a + (t2 ? rr : ll);
        const uint32_t cr = (t1 ? 1u : 0u) + (t2 ? 1u : 0u);
        // the nodes taken get their parent (they are root, root + 1: below next)
        if (cr >= 1) val[root] = next;
        if (cr == 2) val[root + 1] = next;
        val[next] = v;
        root += cr;
        leaf += 2 - cr;
        // the heads of both queues for the next round (node `next` is in the queue now)
        r0 = val[root <= next ? root : next];  // (root <= next always: the node just made is there)
        const uint32_t r1v = val[root + 1 <= next ? root + 1 : next];
        const uint32_t l0v = val[leaf < n ? leaf : n - 1], l1v = val[leaf + 1 < n ? leaf + 1 : n - 1];
        r1 = root + 1 <= next ? r1v : INF;
        l0 = leaf < n ? l0v : INF;
        l1 = leaf + 1 < n ? l1v : INF;
    }
#else
    uint32_t root = 0, leaf = 2;
    uint32_t rv = val[0] + val[1];
    val[0] = rv;
    uint32_t lv = leaf < n ? val[leaf] : 0u;
    for (uint32_t next = 1; next + 1 < n; next++) {
        uint32_t v;
        if (leaf >= n || rv < lv) {
            v = rv;
            val[root] = next;
            root++;
            rv = root < next ? val[root] : 0u;
        } else {
            v = lv;
            leaf++;
            lv = leaf < n ? val[leaf] : 0u;
        }
        if (leaf >= n || (root < next && rv < lv)) {
            v += rv;
            val[root] = next;
            root++;
            rv = root < next ? val[root] : 0u;
        } else {
            v += lv;
            leaf++;
            lv = leaf < n ? val[leaf] : 0u;
        }
        val[next] = v;
        if (root == next) rv = v;
    }
#endif
}

// One code, built by ONE wave (the two waves of the workgroup build the literal/length and the distance
// code side by side): the lanes hand data to each other through the wave's scratch, so wave_lds_fence
// stands where a workgroup would need a barrier.
#ifdef MI355_HDR_TIMERS
#define HT(i) { unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0 && blockIdx.x == 0) ht[i] += t_ - ht0; ht0 = __builtin_readcyclecounter(); }  // (wave 0 of block 0: the literal/length code's wave -- the kernel's critical path)
#define HT_DECL unsigned long long ht0 = __builtin_readcyclecounter();
__device__ unsigned long long ht[16];
#else
#define HT(i)
#define HT_DECL
#endif
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <uint32_t NQ>
__device__ __forceinline__ void huff_rank_sort(HuffScratch& s, uint32_t m, uint32_t lane) {
    uint32_t k[NQ], rank[NQ];
#pragma unroll
    for (uint32_t q = 0; q < NQ; q++) {
        k[q] = lane + 64 * q < m ? s.key[lane + 64 * q] : 0u;
        rank[q] = 0;
    }
    const uint4* kp = reinterpret_cast<const uint4*>(s.key);
    uint4 kk = kp[0];
    for (uint32_t j = 0; j < m; j += 4) {
        const uint4 cur = kk;
        kk = kp[(j >> 2) + 1];  // (at most one read behind the pad: inside the array)
#pragma unroll
        for (uint32_t q = 0; q < NQ; q++)
            rank[q] += (cur.x < k[q] ? 1u : 0u) + (cur.y < k[q] ? 1u : 0u) + (cur.z < k[q] ? 1u : 0u) + (cur.w < k[q] ? 1u : 0u);
    }
#pragma unroll
    for (uint32_t q = 0; q < NQ; q++)
        if (lane + 64 * q < m) {
            s.val[rank[q]] = k[q] >> 9;
            s.sym[rank[q]] = k[q] & 511u;
        }
}
template <class LenArr>
__device__ void wave_huff(HuffScratch& s, const uint32_t* freqs, uint32_t n, uint32_t n_total, uint32_t max_len,
                          LenArr& lengths, uint32_t lane, uint32_t htb = 0) {
    (void)htb;
    HT_DECL
    for (uint32_t i = lane; i < n_total; i += 64) lengths[i] = 0;
    uint32_t m = 0;  // gather_nodes
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t i = c0 + lane;
        const uint32_t f = i < n ? freqs[i] : 0u;
        const uint64_t mask = __builtin_amdgcn_ballot_w64(f > 0);
        if (f > 0) s.key[m + (uint32_t)__popcll(mask & ((1ull << lane) - 1))] = (f << 9) | i;
        m += (uint32_t)__popcll(mask);
    }
    wave_lds_fence();
    if (m == 0) return;
    if (m == 1) {  // length_encode.rs:377-382
        if (lane == 0) lengths[s.key[0] & 511u] = 1;
        wave_lds_fence();
        return;
    }
    // rank sort: a lane's keys (one for every 64 symbols in use) against every key, four keys a read, the next four fetched
    // before these are compared (one key a round and five registers whatever m is: 11 of the literal/length code's 80 kcycles)
    if (lane < 4) s.key[m + lane] = 0xFFFFFFFFu;  // (behind the last: below no key)
    wave_lds_fence();
    switch ((m + 63) / 64) {
    case 1: huff_rank_sort<1>(s, m, lane); break;
    case 2: huff_rank_sort<2>(s, m, lane); break;
    case 3: huff_rank_sort<3>(s, m, lane); break;
    case 4: huff_rank_sort<4>(s, m, lane); break;
    default: huff_rank_sort<5>(s, m, lane); break;
    }
    wave_lds_fence();
    HT(0 + htb)
    if (lane == 0) mk_phase1(s.val, m);
    wave_lds_fence();
    HT(1 + htb)
    // internal nodes 0 .. m-2, the root is m-2: depth = number of parent hops to the root
    const uint32_t rootn = m - 2;
    for (uint32_t t = lane; t + 1 < m; t += 64) {
        s.pj[t] = t < rootn ? (1u << 16) | s.val[t] : rootn;
        s.icnt[t] = 0;
    }
    if (lane < 40) s.num[lane] = 0;
    wave_lds_fence();
    for (uint32_t span = 1; span < m; span <<= 1) {  // after k rounds every word spans 2^k hops or ends at the root
        uint32_t nw[5];
#pragma unroll
        for (uint32_t k = 0; k < 5; k++) {
            const uint32_t t = lane + 64 * k;
            if (t + 1 < m) {
                const uint32_t w = s.pj[t], pw = s.pj[w & 0xffff];
                nw[k] = (((w >> 16) + (pw >> 16)) << 16) | (pw & 0xffff);
            }
        }
        wave_lds_fence();
#pragma unroll
        for (uint32_t k = 0; k < 5; k++) {
            const uint32_t t = lane + 64 * k;
            if (t + 1 < m) s.pj[t] = nw[k];
        }
        wave_lds_fence();
    }
    for (uint32_t t = lane; t + 1 < m; t += 64) atomicAdd(&s.icnt[s.pj[t] >> 16], 1u);
    wave_lds_fence();
    HT(2 + htb)
    {
        // :253-278 level by level: of the `available` slots of a depth -- one at the root, else two for every internal node a
        // level up -- the internal nodes take theirs, the leaves the rest: a lane per depth (a level without internal nodes has
        // none below it, so the loop's end needs no looking for)
        for (uint32_t d = lane; d < m; d += 64) {
            const uint32_t used = d + 1 < m ? s.icnt[d] : 0u;
            const uint32_t available = d == 0 ? 1u : 2u * s.icnt[d - 1];  // (d - 1 + 1 < m)
            const uint32_t leaves = available > used ? available - used : 0u;
            if (d < 32)
                s.num[d] = leaves;
            else if (leaves)
                atomicAdd(&s.num[32], leaves);
        }
        wave_lds_fence();
        // enforce_max_code_lengths :290-327 (stages.h limit_code_lengths): the Kraft sum by the wave; the loop that moves
        // codes only where it is off (a depth beyond max_len, or more than a full tree after the fold)
        const uint32_t nd = lane < 33 ? s.num[lane] : 0u;
        const uint32_t above = wave_sum(lane > max_len ? nd : 0u);
        const uint32_t mine = lane == max_len ? nd + above : nd;
        const uint32_t total = wave_sum(lane >= 1 && lane <= max_len ? mine << (max_len - lane) : 0u);
        if (above != 0 || total != (1u << max_len)) {
            if (lane == 0) limit_code_lengths(s.num, max_len);
        }
    }
    wave_lds_fence();
    HT(3 + htb)
    for (uint32_t idx = lane; idx < m; idx += 64) {  // :402-408: the idx-th leaf from the end
        uint32_t acc = 0, len = 0;
        for (uint32_t i = 1; i <= max_len; i++) {
            acc += s.num[i];
            if (len == 0 && idx < acc) len = i;
        }
        lengths[s.sym[m - 1 - idx]] = (uint8_t)len;
    }
    wave_lds_fence();
    HT(4 + htb)
}

__global__ __launch_bounds__(128) void k_block_header(const DevScalars* sc, const uint32_t* __restrict__ ll_freq,
                                                      const uint32_t* __restrict__ d_freq, BlockHeader* __restrict__ hdr, uint32_t piece) {
    __shared__ HdrLds s;
    const uint32_t b = sc->nbcum[piece] + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (b >= sc->nb || spec_failed(sc)) return;
    HT_DECL
    for (uint32_t i = tid; i < 288; i += 128) {
        uint32_t f = i == END_OF_BLOCK ? 1u : 0u;  // output_writer.rs:83
        for (uint32_t q = 0; q < PSPLIT; q++) f += ll_freq[((uint64_t)b * PSPLIT + q) * 288 + i];
        s.llf[i] = f;
    }
    if (tid < 32) {
        uint32_t f = 0;
        for (uint32_t q = 0; q < PSPLIT; q++) f += d_freq[((uint64_t)b * PSPLIT + q) * 32 + tid];
        s.df[tid] = f;
    }
    if (tid < 20) s.clf[tid] = 0;
    __syncthreads();
    if (wv == 0) { HT(5) }
    if (wv == 0) {
        // remove_trailing_zeroes huffman_lengths.rs:44-47 (stages.h trimmed_count): the last used symbol
        uint32_t last_ll = 0;
        for (uint32_t i = lane; i < NUM_LL; i += 64)
            if (s.llf[i]) last_ll = i + 1;
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const uint32_t a = __shfl_xor(last_ll, off);
            last_ll = a > last_ll ? a : last_ll;
        }
        const uint32_t n_ll = last_ll > 257 ? last_ll : 257;
        if (lane == 0) s.n_ll = n_ll;
        wave_huff(s.w[0], s.llf, n_ll, 288, 15, s.ll_len, lane);
    } else {
        uint32_t last_d = (lane < NUM_DIST && s.df[lane]) ? lane + 1 : 0;
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const uint32_t a = __shfl_xor(last_d, off);
            last_d = a > last_d ? a : last_d;
        }
        const uint32_t n_d = last_d > 1 ? last_d : 1;
        if (lane == 0) s.n_d = n_d;
        wave_huff(s.w[1], s.df, n_d, 32, 15, s.d_len, lane);
    }
    __syncthreads();
    for (uint32_t i = tid; i < s.n_ll; i += 128) s.chain[i] = s.ll_len[i];
    for (uint32_t i = tid; i < s.n_d; i += 128) s.chain[s.n_ll + i] = s.d_len[i];
    __syncthreads();
    if (wv == 0) { HT(6) }
    if (wv == 0) {
        // length_encode.rs:82-155 run by run (stages.h el_run_count / el_run_emit): a lane per run start
        const uint32_t n_len = s.n_ll + s.n_d;
        uint64_t starts[5];
#pragma unroll
        for (uint32_t q = 0; q < 5; q++) {
            const uint32_t i = lane + 64 * q;
            starts[q] = __builtin_amdgcn_ballot_w64(i < n_len && (i == 0 || s.chain[i] != s.chain[i - 1]));
        }
        uint32_t base = 0;
#pragma unroll
        for (uint32_t q = 0; q < 5; q++) {
            const uint32_t i = lane + 64 * q;
            const bool st = (starts[q] >> lane) & 1;
            uint32_t e = n_len;  // where the run ends: the next start, in this chunk or a later one
#pragma unroll
            for (uint32_t q2 = 4; q2 > q; q2--)
                if (starts[q2]) e = 64 * q2 + (uint32_t)__builtin_ctzll(starts[q2]);
            const uint64_t rest = (starts[q] >> lane) >> 1;
            if (rest) e = i + 1 + (uint32_t)__builtin_ctzll(rest);
            const uint32_t v = st ? s.chain[i] : 0u;
            const uint32_t cnt = st ? el_run_count(v, e - i) : 0u;
            const uint32_t incl = wave_incl_scan(cnt, lane);
            if (st) el_run_emit(v, e - i, s.enc, base + incl - cnt);
            base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) s.n_enc = base;
    }
    __syncthreads();
    for (uint32_t i = tid; i < s.n_enc; i += 128) atomicAdd(&s.clf[el_symbol_index(s.enc[i])], 1u);
    __syncthreads();
    if (wv == 0) { HT(7) }
    BlockHeader* h = hdr + b;
    if (wv == 0) {
        wave_huff(s.w[0], s.clf, 19, 19, 7, s.cl_len, lane, 9);
        if (lane == 0) s.used = count_used_hclens(s.cl_len);
    }
    __syncthreads();
    if (wv == 1) {
        for (uint32_t i = lane; i < 288; i += 64) h->ll_len[i] = s.ll_len[i];
        if (lane < 32) h->d_len[lane] = s.d_len[lane];
        if (lane < 19) h->cl_len[lane] = s.cl_len[lane];
        for (uint32_t i = lane; i < s.n_enc; i += 64) h->enc[i] = s.enc[i];
        return;
    }
    {  // stages.h block_costs, summed over the wave
        uint32_t d_ll = 0, s_ll = 0, d_d = 0, s_d = 0, f_d = 0, table = 0, table_real = 0;
        for (uint32_t c = lane; c < s.n_ll; c += 64) {
            const uint32_t f = s.llf[c], extra = c >= 257 ? length_extra_bits_of_code(c - 257) : 0u;
            d_ll += f * (s.ll_len[c] + extra);
            s_ll += f * (fixed_ll_length(c) + extra);
        }
        if (lane < s.n_d) {
            const uint32_t f = s.df[lane], extra = distance_extra_bits_of_code(lane);
            d_d = f * (s.d_len[lane] + extra);
            s_d = f * (fixed_ll_length(lane) + extra);  // Q12: the ll table is used for distances too
            f_d = f * (5 + extra);
        }
        if (lane < 19) {
            const uint32_t extra = (lane == 16 || lane == 17) ? 3u : (lane == 18 ? 7u : 0u);
            const uint32_t extra_real = lane == 16 ? 2u : extra;  // write_huffman_lengths :343 writes 2 bits
            table = s.clf[lane] * (s.cl_len[lane] + extra);
            table_real = s.clf[lane] * (s.cl_len[lane] + extra_real);
        }
        d_ll = wave_sum(d_ll);
        s_ll = wave_sum(s_ll);
        d_d = wave_sum(d_d);
        s_d = wave_sum(s_d);
        f_d = wave_sum(f_d);
        table = wave_sum(table);
        table_real = wave_sum(table_real);
        if (lane == 0) {
            h->n_enc = s.n_enc;
            h->n_ll = s.n_ll;
            h->n_d = s.n_d;
            h->used_hclens = s.used;
            h->dyn_est = (uint64_t)d_ll + d_d + table + (uint64_t)s.used * 3 + 5 + 5 + 4;
            h->dyn_bits = (uint64_t)d_ll + d_d + table_real + (uint64_t)s.used * 3 + 5 + 5 + 4;
            h->static_est = (uint64_t)s_ll + s_d;
            h->fixed_bits = (uint64_t)s_ll + f_d;
        }
    }
    HT(8)
#ifdef MI355_HDR_TIMERS
    if (lane == 0 && b == 0) {
        printf("hdr timers (cycles, the literal/length wave of block 0): load %llu | ll code: gather+sort %llu phase1 %llu depths %llu levels+limit %llu handout %llu (the whole, with the wait for the distance wave and the chain: %llu) | rle + count %llu | cl code: gather+sort %llu phase1 %llu depths %llu levels+limit %llu handout %llu (with the costs: %llu)\n",
               ht[5], ht[0], ht[1], ht[2], ht[3], ht[4], ht[6], ht[7], ht[9], ht[10], ht[11], ht[12], ht[13], ht[8]);
        for (int i = 0; i < 16; i++) ht[i] = 0;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// k_plan: compress.rs:157-246 -- the one strictly serial step: block type (needs the bit phase,
// huffman_lengths.rs:269), bit offsets, BFINAL.  One wave; each lane loads one block, the wave
// then steps through its 64 blocks in order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void put_bits(uint32_t* out32, uint64_t bitpos, uint64_t bits, uint32_t nbits);

__device__ __forceinline__ void plan_blocks(uint32_t* s_red, uint32_t* s_flag, DevScalars* sc, const BlockHeader* __restrict__ hdr,
                                            const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ q13,
                                            BlockPlan* __restrict__ plan, uint64_t bit_base, uint32_t compat,
                                            const uint32_t* __restrict__ blk_sync, uint32_t* __restrict__ out32, Piece pc) {
    const uint32_t lane = threadIdx.x & 63;
    if (spec_failed(sc)) return;
    // (a piece: the blocks from sc->nbcum[pc.p] on, behind the bits planned so far; `fin` = the block that ends the stream)
    const uint32_t nb = sc->nb, fin = pc.last ? nb : 0xFFFFFFFFu;
    uint64_t bitpos = bit_base + sc->total_bits;
    // A thread per block where the piece has at most 1024 of them (110 MB of text) and none depends on the bit phase it
    // starts at -- no Stored block, no sync marker: type and length the same for all eight phases --: the bit offsets are a
    // prefix sum over the workgroup.  (One wave, 64 blocks a round with the eight phases of each worked out by its lane, took
    // 22 us for the 937 blocks of the 100 MB text: fifteen rounds of arithmetic, not of memory.)  Anything else is the
    // one wave's walk below.
    {
        const uint32_t first = sc->nbcum[pc.p], nbp = nb > first ? nb - first : 0u;
        const uint32_t b = first + threadIdx.x;
        const bool have = threadIdx.x < nbp;
        bool plain = nbp <= 1024;
        BlockPlan p0;
        p0.btype = BT_FIXED;
        p0.bfinal = 0;
        p0.bit_start = 0;
        p0.bit_len = 0;
        if (plain && have) {
            const uint64_t dyn_bits = hdr[b].dyn_bits, dyn_est = hdr[b].dyn_est, static_est = hdr[b].static_est,
                           fixed_bits = hdr[b].fixed_bits, in_bytes = (uint64_t)bstart[b + 1] - bstart[b];
            plain = blk_sync[b] == 0;
            plan_block(dyn_bits, dyn_est, static_est, fixed_bits, in_bytes, b + 1 == fin, 0, &p0);
            plain = plain && p0.btype != BT_STORED;
#pragma unroll
            for (uint32_t ph = 1; ph < 8; ph++) {
                BlockPlan q;
                plan_block(dyn_bits, dyn_est, static_est, fixed_bits, in_bytes, b + 1 == fin, ph, &q);
                plain = plain && q.btype == p0.btype && q.bit_len == p0.bit_len;
            }
        }
        if (threadIdx.x < 3) s_flag[threadIdx.x] = 0;
        __syncthreads();
        if (!plain) s_flag[0] = 1;
        if (plain && have) atomicAdd(&s_flag[p0.btype == BT_FIXED ? 1 : 2], 1u);
        __syncthreads();
        if (s_flag[0] == 0) {
            uint32_t total = 0;
            const uint32_t mylen = have ? (uint32_t)p0.bit_len : 0u;  // < 2^21 for a block that is not stored
            const uint32_t before = block_excl_scan_1024(mylen, s_red, &total);
            if (have) {
                p0.bit_start = bitpos + before;
                plan[b] = p0;
            }
            if (threadIdx.x == 0) {
                sc->total_bits = bitpos + total - bit_base;
                sc->n_fixed += s_flag[1];
                sc->n_dynamic += s_flag[2];
            }
            return;
        }
        if (threadIdx.x >= 64) return;  // (whole waves; no barrier below)
    }
    uint32_t n_st = 0, n_fx = 0, n_dy = 0, hits = 0, panic = 0;
    for (uint32_t b0 = sc->nbcum[pc.p]; b0 < nb; b0 += 64) {
        uint32_t b = b0 + lane;
        bool have = b < nb;
        uint64_t dyn_bits = 0, dyn_est = 0, static_est = 0, fixed_bits = 0, in_bytes = 0;
        uint32_t my_sync = 0, my_q13 = 0;  // everything the serial walk needs is loaded here, 64 blocks at a time
        if (have) {
            dyn_bits = hdr[b].dyn_bits;
            dyn_est = hdr[b].dyn_est;
            static_est = hdr[b].static_est;
            fixed_bits = hdr[b].fixed_bits;
            in_bytes = (uint64_t)bstart[b + 1] - bstart[b];
            my_sync = blk_sync[b];
            my_q13 = q13[b];
        }
        uint32_t cnt = nb - b0 < 64 ? nb - b0 : 64;
        // Only the cost of a Stored block (and the pad of a sync marker) depends on the bit phase a block
        // starts at.  A block whose type and length come out the same for all eight phases needs no
        // walk: when that holds for all 64 blocks of the group, their bit offsets are a prefix sum.
        bool fixed_len = have && my_sync == 0, fixed_type = fixed_len;
        BlockPlan p0;
        p0.btype = BT_FIXED;
        p0.bfinal = 0;
        p0.bit_start = 0;
        p0.bit_len = 0;
        if (have) plan_block(dyn_bits, dyn_est, static_est, fixed_bits, in_bytes, b + 1 == fin && my_sync == 0, 0, &p0);
        if (fixed_len) {
#pragma unroll
            for (uint32_t ph = 1; ph < 8; ph++) {
                BlockPlan q;
                plan_block(dyn_bits, dyn_est, static_est, fixed_bits, in_bytes, false, ph, &q);
                fixed_type = fixed_type && q.btype == p0.btype;
                fixed_len = fixed_len && q.btype == p0.btype && q.bit_len == p0.bit_len;
            }
        }
        // A group of blocks that are Stored whatever the phase (incompressible data): a stored block ends on a
        // byte boundary, so every block but the group's first starts at phase 0 -- the length p0 was computed
        // for -- and the first one takes the real phase.
        if (__builtin_amdgcn_ballot_w64(have && !(fixed_type && p0.btype == BT_STORED)) == 0) {
            if (lane == 0) plan_block(dyn_bits, dyn_est, static_est, fixed_bits, in_bytes, b + 1 == fin, bitpos, &p0);
            const uint64_t mylen = have ? p0.bit_len : 0ull;
            uint64_t incl = mylen;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t ylo = __shfl_up((uint32_t)incl, off), yhi = __shfl_up((uint32_t)(incl >> 32), off);
                if (lane >= (uint32_t)off) incl += ((uint64_t)yhi << 32) | ylo;
            }
            if (have) {
                p0.bit_start = bitpos + (incl - mylen);
                plan[b] = p0;
                n_st++;
                if (my_q13) {
                    hits++;
                    if (my_q13 == 2 && (compat & 1)) panic = 1;
                }
            }
            bitpos += ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(incl >> 32), 63) << 32) |
                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)incl, 63);
            continue;
        }
        if (__builtin_amdgcn_ballot_w64(have && !fixed_len) == 0) {
            const uint32_t mylen = have ? (uint32_t)p0.bit_len : 0u;  // < 2^21 for a non-stored block
            uint32_t incl = mylen;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t y = __shfl_up(incl, off);
                if (lane >= (uint32_t)off) incl += y;
            }
            if (have) {
                p0.bit_start = bitpos + (incl - mylen);
                plan[b] = p0;
                if (p0.btype == BT_STORED) {  // (cannot happen here: a stored length depends on the phase)
                    n_st++;
                } else if (p0.btype == BT_FIXED) {
                    n_fx++;
                } else {
                    n_dy++;
                }
            }
            bitpos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            continue;
        }
        for (uint32_t sidx = 0; sidx < cnt; sidx++) {
            uint64_t len = 0;
            if (lane == sidx) {
                BlockPlan p;
                const bool sync = my_sync != 0;
                plan_block(dyn_bits, dyn_est, static_est, fixed_bits, in_bytes, (b + 1 == fin) && !sync, bitpos, &p);
                plan[b] = p;
                len = p.bit_len;
                if (sync) {  // compress.rs:256-261: empty stored block = 3 zero bits, pad, 00 00 FF FF
                    uint64_t e = bitpos + len;
                    uint64_t hb = (e + 3 + 7) & ~7ull;
                    put_bits(out32, hb, 0xFFFF0000ull, 32);
                    len = hb + 32 - bitpos;
                }
                if (p.btype == BT_STORED) {
                    n_st++;
                    if (my_q13) {
                        hits++;
                        if (my_q13 == 2 && (compat & 1)) panic = 1;
                    }
                } else if (p.btype == BT_FIXED) {
                    n_fx++;
                } else {
                    n_dy++;
                }
            }
            // (sidx is uniform: a scalar read of the lane, no trip through the LDS crossbar)
            uint32_t len_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)len, (int)sidx),
                     len_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(len >> 32), (int)sidx);
            bitpos += ((uint64_t)len_hi << 32) | len_lo;
        }
    }
    // reduce the per-lane counters
    for (int off = 32; off; off >>= 1) {
        n_st += __shfl_down(n_st, off);
        n_fx += __shfl_down(n_fx, off);
        n_dy += __shfl_down(n_dy, off);
        hits += __shfl_down(hits, off);
        panic |= __shfl_down(panic, off);
    }
    if (lane == 0) {
        sc->total_bits = bitpos - bit_base;
        sc->n_stored += n_st;
        sc->n_fixed += n_fx;
        sc->n_dynamic += n_dy;
        sc->q13_hits += hits;
        sc->ref_panic |= panic;
    }
}

// (host_state: a small one-shot call's scalars are final with the plan -- k_pack writes none -- and leave for the host's
// page-locked copy from here: the 240 bytes were a launch of their own)
__global__ __launch_bounds__(1024) void k_plan(DevScalars* sc, const BlockHeader* __restrict__ hdr,
                                               const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ q13,
                                               BlockPlan* __restrict__ plan, uint64_t bit_base, uint32_t compat,
                                               const uint32_t* __restrict__ blk_sync, uint32_t* __restrict__ out32, Piece pc,
                                               uint32_t* __restrict__ host_state) {
    __shared__ uint32_t s_red[16], s_flag[3];
    plan_blocks(s_red, s_flag, sc, hdr, bstart, q13, plan, bit_base, compat, blk_sync, out32, pc);
    if (host_state) {
        __syncthreads();  // (what one thread wrote into *sc above)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(sc);
        if (threadIdx.x < sizeof(DevState) / 4) host_state[threadIdx.x] = src[threadIdx.x];
    }
}

// ---------------------------------------------------------------------------------------------
// k_pack: encoder_state.rs:58-105 + bitstream.rs:76-86 + stored_block.rs:13-40 for one block per
// workgroup.  Tokens are coded 256 at a time: per-lane bit string, workgroup exclusive scan of
// the lengths, then OR into the zeroed output words (seams between lanes and blocks share
// words, hence atomicOr).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void put_bits(uint32_t* out32, uint64_t bitpos, uint64_t bits, uint32_t nbits) {
    if (nbits == 0) return;
    uint64_t w = bitpos >> 5;
    uint32_t sh = (uint32_t)(bitpos & 31);
    uint64_t lo = bits << sh;                    // bits 0..63 of the shifted value
    uint32_t hi = sh ? (uint32_t)(bits >> (64 - sh)) : 0u;
    uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32);
    if (w0) atomicOr(out32 + w, w0);
    if (w1) atomicOr(out32 + w + 1, w1);
    if (hi) atomicOr(out32 + w + 2, hi);
}

// A wave's 256 tokens of a round take at most 256 * 48 bits: they are OR-ed together in LDS and go out as
// whole words with plain stores; only the first and the last word of the wave's bit range, which it
// shares with its neighbours, are OR-ed into the output.
constexpr uint32_t PACK_WORDS = 256 * 48 / 32 + 4;
#ifndef MI355_PACK_THREADS
#define MI355_PACK_THREADS 256
#endif
constexpr uint32_t PKT_LARGE = MI355_PACK_THREADS;  // threads of a k_pack workgroup (a part of a block, 7936 tokens, in 7936 / (4 PKT) rounds)
constexpr uint32_t PKT_SMALL = 1024;                // ... of a call with fewer parts than compute units: a part's rounds are its time
__device__ __forceinline__ void put_bits_lds(uint32_t* buf, uint32_t bitpos, uint64_t bits, uint32_t nbits) {
    if (nbits == 0) return;
    const uint32_t w = bitpos >> 5, sh = bitpos & 31;
    const uint64_t lo = bits << sh;
    const uint32_t hi = sh ? (uint32_t)(bits >> (64 - sh)) : 0u;
    const uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32);
    if (w0) atomicOr(buf + w, w0);
    if (w1) atomicOr(buf + w + 1, w1);
    if (hi) atomicOr(buf + w + 2, hi);
}

template <uint32_t PKW>
struct alignas(4) PackLds {
    uint16_t llc[288];
    uint16_t dc[32];
    uint16_t clc[20];
    uint8_t lll[288];  // (the three length arrays start at multiples of four bytes: k_pack reads them four at a time)
    uint8_t dl[32];
    uint8_t cll[20];   // code-length code lengths of a dynamic block
    uint32_t cnt[48];   // canonical codes: symbols per code length, then first code per length; [table][16]
    uint32_t scan[256];
    uint32_t carry;
    uint32_t wbuf[PKW][PACK_WORDS];  // per wave: the bits of its 256 tokens of a round, zero between rounds
};

template <uint32_t PKT>
__global__ __launch_bounds__(PKT) void k_pack(const uint8_t* __restrict__ in, uint32_t n,
                                              const uint32_t* __restrict__ dtok, const DevScalars* sc,
                                              const BlockHeader* __restrict__ hdr, const BlockPlan* __restrict__ plan,
                                              const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ q13,
                                              uint32_t compat, uint32_t* __restrict__ out32, BlockTab tab,
                                              const uint32_t* __restrict__ ll_freq, const uint32_t* __restrict__ d_freq, uint32_t piece) {
    constexpr uint32_t PKW = PKT / 64;
    __shared__ PackLds<PKW> s;
    const uint32_t b = sc->nbcum[piece] + blockIdx.x / PSPLIT, part = blockIdx.x % PSPLIT, tid = threadIdx.x;
    if (b >= sc->nb || spec_failed(sc)) return;
    const uint32_t gtid = part * PKT + tid;  // a stored block's bytes are spread over all parts' threads
    constexpr uint32_t GT = PKT * PSPLIT;
    const BlockPlan pl = plan[b];
    const BlockHeader* h = hdr + b;
    uint64_t bp = pl.bit_start;
    if (pl.btype == BT_STORED) {
        // compress.rs:59-77, stored_block.rs:13-40
        uint64_t src = bstart[b];
        if (q13[b] && (compat & 1)) src += WINDOW_SIZE;  // bug-for-bug (A.4 Q13)
        uint64_t left = (uint64_t)bstart[b + 1] - bstart[b];
        do {
            uint64_t piece = left < (uint64_t)MAX_STORED_BLOCK_LENGTH ? left : (uint64_t)MAX_STORED_BLOCK_LENGTH;
            bool last_piece = piece == left;
            uint64_t hb = (bp + 3 + 7) & ~7ull;  // header bits then pad to a byte
            if (gtid == 0) {
                put_bits(out32, bp, (pl.bfinal && last_piece) ? 1u : 0u, 3);
                put_bits(out32, hb, (piece & 0xffff) | (((~piece) & 0xffff) << 16), 32);
            }
            uint64_t ob = (hb >> 3) + 4;  // first payload byte
            // The output words that lie wholly inside the payload belong to this piece alone: plain
            // 4-byte stores (the source is read byte-wise, it has no alignment to speak of).  The up
            // to three bytes before the first and after the last whole word share their words with
            // the header or with the next block: OR.
            const uint64_t w0 = (ob + 3) >> 2, w1 = (ob + piece) >> 2;  // whole words [w0, w1)
            if (w1 > w0) {
                for (uint64_t w = w0 + gtid; w < w1; w += GT) {
                    const uint64_t i = (w << 2) - ob;  // payload offset of the word's first byte
                    uint32_t v = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) v |= ((src + i + k < n) ? (uint32_t)in[src + i + k] : 0u) << (8 * k);
                    out32[w] = v;
                }
                const uint64_t headn = (w0 << 2) - ob, tail0 = (w1 << 2) - ob;
                for (uint64_t i = gtid; i < headn + (piece - tail0); i += GT) {
                    const uint64_t j = i < headn ? i : tail0 + (i - headn);
                    const uint64_t o = ob + j;
                    uint32_t v = (src + j < n) ? in[src + j] : 0u;
                    if (v) atomicOr(out32 + (o >> 2), v << (8 * (o & 3)));
                }
            } else {
                for (uint64_t i = gtid; i < piece; i += GT) {
                    uint64_t o = ob + i;
                    uint32_t v = (src + i < n) ? in[src + i] : 0u;
                    if (v) atomicOr(out32 + (o >> 2), v << (8 * (o & 3)));
                }
            }
            bp = (ob + piece) * 8;
            src += piece;
            left -= piece;
        } while (left > 0);
        return;
    }
    const uint32_t nt = tab.nt[b];
    if (part > 0 && part * PQ >= nt) return;  // (an empty block is part 0's)
    // code tables
    if (pl.btype == BT_FIXED) {
        for (uint32_t i = tid; i < 288; i += PKT) s.lll[i] = (uint8_t)fixed_ll_length(i);
        if (tid < 32) s.dl[tid] = 5;
    } else {
        for (uint32_t i = tid; i < 288; i += PKT) s.lll[i] = h->ll_len[i];
        if (tid < 32) s.dl[tid] = h->d_len[tid];
    }
    if (tid < 20) s.cll[tid] = (pl.btype == BT_DYNAMIC && tid < 19) ? h->cl_len[tid] : 0;
    if (tid < 48) s.cnt[tid] = 0;
    for (uint32_t i = tid; i < PKW * PACK_WORDS; i += PKT) (&s.wbuf[0][0])[i] = 0;
    __syncthreads();
    // Canonical codes (huffman_table.rs:253-278; stages.h canonical_codes is the serial form) for the three
    // tables at once: symbols per length by LDS atomics, first code of every length by one thread per table,
    // then symbol i takes the first code of its length plus the number of symbols before it with that length.
    for (uint32_t i = tid; i < 288; i += PKT)
        if (s.lll[i]) atomicAdd(&s.cnt[s.lll[i]], 1u);
    if (tid < 32 && s.dl[tid]) atomicAdd(&s.cnt[16 + s.dl[tid]], 1u);
    if (tid < 19 && s.cll[tid]) atomicAdd(&s.cnt[32 + s.cll[tid]], 1u);
    __syncthreads();
    if (tid < 3) {
        uint32_t* c = s.cnt + 16 * tid;
        uint32_t code = 0, before = 0;  // (no symbol is counted under length 0)
        for (uint32_t bits = 1; bits < 16; bits++) {
            code = ((code + before) << 1) & 0xffff;
            before = c[bits];
            c[bits] = code;
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < 288 + 32 + 20; i += PKT) {
        const uint8_t* len = i < 288 ? s.lll : (i < 320 ? s.dl : s.cll);
        uint16_t* codes = i < 288 ? s.llc : (i < 320 ? s.dc : s.clc);
        const uint32_t* first = s.cnt + (i < 288 ? 0 : (i < 320 ? 16 : 32));
        const uint32_t k = i < 288 ? i : (i < 320 ? i - 288 : i - 320);
        const uint32_t l = len[k];
        uint32_t r = 0;
        if (l) {
            // (the symbols before k with its length, four lengths a read: byte by byte this loop was up to 287 dependent LDS
            // reads a thread, a third of the kernel's table building; the three arrays are 4-byte aligned)
            const uint32_t* lw = reinterpret_cast<const uint32_t*>(len);
            const uint32_t pat = l * 0x01010101u, whole = k >> 2, rem = k & 3u;
#pragma unroll 4
            for (uint32_t w = 0; w < whole; w++) {
                const uint32_t z = lw[w] ^ pat;
                r += (uint32_t)__builtin_popcount(~(((z & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z) & 0x80808080u);  // its zero bytes
            }
            if (rem) {
                const uint32_t z = lw[whole] ^ pat;
                r += (uint32_t)__builtin_popcount(~(((z & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z) & 0x80808080u & ((1u << (8 * rem)) - 1u));
            }
        }
        codes[k] = l ? (uint16_t)reverse_bits16((first[l] + r) & 0xffff, l) : (uint16_t)0;
    }
    __syncthreads();
    // block header
    uint32_t hdr_bits = 3;
    if (part > 0) {
        // Where this part's bits begin: behind the header and the tokens of the parts before it, whose sizes
        // follow from their histograms (code length + extra bits per symbol).  A dynamic header is what is
        // left of dyn_bits (stages.h block_costs) after all the symbols.
        uint32_t pre = 0, all = 0;
        for (uint32_t i = tid; i < 320; i += PKT) {
            uint32_t len = 0;
            if (i < NUM_LL) len = s.lll[i] + (i >= 257 ? length_extra_bits_of_code(i - 257) : 0u);
            if (i >= 288 && i - 288 < NUM_DIST) len = s.dl[i - 288] + distance_extra_bits_of_code(i - 288);
            for (uint32_t k = 0; k < PSPLIT; k++) {
                const uint64_t slot = (uint64_t)b * PSPLIT + k;
                const uint32_t f = i < 288 ? ll_freq[slot * 288 + i] : d_freq[slot * 32 + (i - 288)];
                all += f * len;
                if (k < part) pre += f * len;
            }
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            pre += __shfl_xor(pre, off);
            all += __shfl_xor(all, off);
        }
        if ((tid & 63) == 0) {
            s.scan[tid >> 6] = pre;
            s.scan[PKW + (tid >> 6)] = all;
        }
        __syncthreads();
        pre = 0;
        all = 0;
        for (uint32_t k = 0; k < PKW; k++) {
            pre += s.scan[k];
            all += s.scan[PKW + k];
        }
        __syncthreads();
        hdr_bits = 3 + pre;
        if (pl.btype == BT_DYNAMIC) hdr_bits += (uint32_t)(h->dyn_bits - all - s.lll[END_OF_BLOCK]);
    } else if (pl.btype == BT_DYNAMIC) {
        // 3 + 14 bits, the code-length code lengths (huffman_lengths.rs:329-331), then the run-length
        // coded lengths (:338-368), one symbol per thread: bit strings, a scan of their lengths over
        // the workgroup, OR into the output.  (One lane walking the list would wait for two dependent
        // loads from the header in global memory per symbol.)
        const uint32_t used = h->used_hclens, n_enc = h->n_enc;
        if (tid == 0) {
            uint64_t p = bp;
            put_bits(out32, p, pl.bfinal ? 5u : 4u, 3);  // encoder_state.rs:12-13
            p += 3;
            put_bits(out32, p, (h->n_ll - 257) | ((h->n_d - 1) << 5) | ((used >= 4 ? used - 4 : 0) << 10), 14);
            p += 14;
            for (uint32_t i = 0; i < used; i++) {
                put_bits(out32, p, s.cll[hclen_order(i)], 3);
                p += 3;
            }
        }
        uint64_t hp = bp + 17 + 3ull * used;
        const uint32_t lane0 = tid & 63, wv0 = tid >> 6;
        for (uint32_t i0 = 0; i0 < n_enc; i0 += PKT) {
            const uint32_t i = i0 + tid;
            uint64_t bits = 0;
            uint32_t nb2 = 0;
            if (i < n_enc) {
                const uint32_t e = h->enc[i], kind = e >> 8, v = e & 0xff;
                const uint32_t sym = el_symbol_index(e);
                nb2 = s.cll[sym];
                bits = s.clc[sym];
                if (kind == 1) {
                    bits |= (uint64_t)(v - 3) << nb2;
                    nb2 += 2;
                } else if (kind == 2) {
                    bits |= (uint64_t)(v - 3) << nb2;
                    nb2 += 3;
                } else if (kind == 3) {
                    bits |= (uint64_t)(v - 11) << nb2;
                    nb2 += 7;
                }
            }
            uint32_t incl = nb2;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t y = __shfl_up(incl, off);
                if (lane0 >= (uint32_t)off) incl += y;
            }
            if (lane0 == 63) s.scan[wv0] = incl;
            __syncthreads();
            uint32_t wbase = 0, total = 0;
#pragma unroll
            for (uint32_t k = 0; k < PKW; k++) {
                uint32_t y = s.scan[k];
                if (k < wv0) wbase += y;
                total += y;
            }
            put_bits(out32, hp + wbase + (incl - nb2), bits, nb2);
            hp += total;
            __syncthreads();
        }
        hdr_bits = (uint32_t)(hp - bp);
    } else if (tid == 0) {
        put_bits(out32, bp, pl.bfinal ? 3u : 2u, 3);  // encoder_state.rs:10-11
    }
    bp += hdr_bits;
    // tokens: 4 consecutive tokens per lane and round; lengths are scanned inside the wave with
    // shuffles and across the 4 waves through LDS (two barriers per 1024 tokens)
    const uint64_t t0 = (uint64_t)tab.t0[b] + (uint64_t)part * PQ;
    const uint64_t t1 = (uint64_t)tab.t0[b] + ((part + 1) * PQ < nt ? (part + 1) * PQ : nt);
    const uint32_t lane = tid & 63, wv = tid >> 6;
    // (a round's four tokens are fetched a round ahead, as one 16-byte load where all four exist: at the head of the round
    // they were a memory latency per round, between two barriers)
    auto fetch4 = [&](uint64_t tq, uint32_t* tk) {
        if (tq + 4 <= t1) {
            const uint4 v = *reinterpret_cast<const uint4*>(dtok + tq);  // (dword aligned is all a global load asks for)
            tk[0] = v.x;
            tk[1] = v.y;
            tk[2] = v.z;
            tk[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) tk[q] = tq + q < t1 ? dtok[tq + q] : 0u;
        }
    };
    uint32_t nxt[4];
    fetch4(t0 + 4ull * tid, nxt);
    uint32_t* const buf = &s.wbuf[0][0];
    bool first_shared = true;
    for (uint64_t tb = t0; tb < t1; tb += 4 * PKT) {
        uint64_t tq = tb + 4ull * tid;
        uint32_t nb4[4];
        uint64_t bits4[4];
        uint32_t mine = 0;
        const uint32_t cur[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
        if (tb + 4 * PKT < t1) fetch4(tq + 4 * PKT, nxt);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            nb4[q] = 0;
            bits4[q] = 0;
            if (tq + q < t1) bits4[q] = token_bits(cur[q], s.llc, s.lll, s.dc, s.dl, &nb4[q]);
            mine += nb4[q];
        }
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t v = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += v;
        }
        if (lane == 63) s.scan[wv] = incl;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (uint32_t k = 0; k < PKW; k++) {
            uint32_t v = s.scan[k];
            if (k < wv) wbase += v;
            total += v;
        }
        // The round's bits go into ONE buffer of the workgroup, word 0 = the output word the round begins in: the seams between
        // the four waves close in LDS, the round's whole words leave by plain stores, and the word it ends in stays behind as word 0
        // of the next round.  Only the first word of the part (shared with the header or the part before) and its last one
        // (behind the loop) are OR-ed into the output.  (Before: a buffer per wave, its first and last word OR-ed into the output
        // every round -- sixty-four atomics a part among the plain stores to the same lines: 101 -> 80 us without them.)
        const uint32_t rel0 = (uint32_t)(bp & 31);
        uint32_t rel = rel0 + wbase + (incl - mine);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            put_bits_lds(buf, rel, bits4[q], nb4[q]);
            rel += nb4[q];
        }
        __syncthreads();
        const uint64_t word0 = bp >> 5;
        const uint32_t endbit = rel0 + total, nfull = endbit >> 5;  // whole words of the round
        for (uint32_t w = tid; w < nfull; w += PKT) {
            const uint32_t v = buf[w];
            buf[w] = 0;
            if (w == 0 && first_shared) {
                if (v) atomicOr(out32 + word0, v);
            } else {
                out32[word0 + w] = v;
            }
        }
        if (tid == 0 && nfull) {  // (thread 0 has done word 0 above; nobody else touches word nfull)
            buf[0] = buf[nfull];
            buf[nfull] = 0;
        }
        if (nfull) first_shared = false;
        bp += total;
        // (the next round writes its sums behind this round's reads of them, and into the buffer behind its own first barrier)
    }
    if (tid == 0 && (bp & 31)) {  // the word the part ends in: the next part's, the next block's or the end-of-block code's as well
        const uint32_t v = buf[0];
        if (v) atomicOr(out32 + (bp >> 5), v);
    }
    if (tid == 0 && (part + 1) * PQ >= nt) put_bits(out32, bp, s.llc[END_OF_BLOCK], s.lll[END_OF_BLOCK]);  // encoder_state.rs:102-105
}

// ---------------------------------------------------------------------------------------------
// Sharded (multi-GPU, stream-exact) path: the blocks a rank owns are cut out of its own tokens plus
// the tokens its right neighbour sent, so block geometry comes from token covers alone.
// k_cover: per block the bytes its tokens cover and its last token; slot 0 = the `skip` tokens that
// belong to the left neighbour's last block.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cover(const uint32_t* __restrict__ dtok, uint64_t skip, uint64_t T2,
                                               uint32_t nb2, uint64_t* __restrict__ cover,
                                               uint32_t* __restrict__ last_tok) {
    __shared__ uint64_t red[256];
    const uint32_t slot = blockIdx.x;  // 0 = head, 1 + b = block b
    if (slot > nb2) return;
    uint64_t t0, t1;
    if (slot == 0) {
        t0 = 0;
        t1 = skip;
    } else {
        t0 = skip + (uint64_t)(slot - 1) * MAX_BUFFER_LENGTH;
        t1 = t0 + MAX_BUFFER_LENGTH < skip + T2 ? t0 + MAX_BUFFER_LENGTH : skip + T2;
    }
    uint64_t s = 0;
    for (uint64_t t = t0 + threadIdx.x; t < t1; t += 256) s += tok_cover(dtok[t]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 128; off; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cover[slot] = red[0];
        if (slot) last_tok[slot - 1] = t1 > t0 ? dtok[t1 - 1] : 0u;
    }
}

// block start positions (buffer coordinates) and the Q13 condition (global window geometry)
__global__ void k_shard_bounds(uint64_t first_pos, const uint64_t* __restrict__ cover,
                               const uint32_t* __restrict__ last_tok, uint32_t nb2, uint64_t T2, uint32_t mode,
                               uint64_t global_lo, uint64_t n_global, uint32_t* __restrict__ bstart,
                               uint32_t* __restrict__ q13) {
    if (threadIdx.x || blockIdx.x) return;
    uint64_t pos = first_pos + cover[0];
    for (uint32_t b = 0; b < nb2; b++) {
        bstart[b] = (uint32_t)pos;
        uint64_t t0 = (uint64_t)b * MAX_BUFFER_LENGTH;
        uint32_t flag = 0;
        if (t0 + MAX_BUFFER_LENGTH <= T2) {
            uint32_t tk = last_tok[b];
            if (tk >> 16) {
                uint64_t mend = global_lo + pos + cover[1 + b];  // absolute end of the last token
                uint64_t tp = mend - tok_cover(tk);
                uint64_t lp = (mode == MODE_LAZY) ? tp + 1 : tp;
                uint64_t wdx = lp / WINDOW_SIZE;
                uint64_t wend = (wdx + 1) * (uint64_t)WINDOW_SIZE;
                if (wdx >= 1 && mend > wend) {
                    uint64_t buf_end = wdx * (uint64_t)WINDOW_SIZE + 65794;
                    if (buf_end > n_global) buf_end = n_global;
                    flag = (mend + WINDOW_SIZE > buf_end) ? 2u : 1u;
                }
            }
        }
        q13[b] = flag;
        pos += cover[1 + b];
    }
    bstart[nb2] = (uint32_t)pos;
}

struct ShardCost {  // what the global, serial plan needs from each block
    uint64_t dyn_bits, dyn_est, static_est, fixed_bits, in_bytes;
    uint32_t q13, pad;
};
__global__ __launch_bounds__(256) void k_shard_costs(const BlockHeader* __restrict__ hdr,
                                                     const uint32_t* __restrict__ bstart,
                                                     const uint32_t* __restrict__ q13, uint32_t nb2,
                                                     ShardCost* __restrict__ out) {
    uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb2) return;
    ShardCost c;
    c.dyn_bits = hdr[b].dyn_bits;
    c.dyn_est = hdr[b].dyn_est;
    c.static_est = hdr[b].static_est;
    c.fixed_bits = hdr[b].fixed_bits;
    c.in_bytes = (uint64_t)bstart[b + 1] - bstart[b];
    c.q13 = q13[b];
    c.pad = 0;
    out[b] = c;
}

// ---------------------------------------------------------------------------------------------
// Adler-32 (RFC 1950; crate adler32 as used by checksum.rs:33-57).  With a = 1 + sum d_j and
// b = n + sum (n - j) d_j, a 16 KiB chunk c contributes a_c = sum d and b_c + after_c * a_c, where
// b_c = sum (len_c - k) d_k inside the chunk and after_c = bytes behind it: no order between chunks,
// so every workgroup adds its two numbers (reduced mod 65521) into 64-bit sums and one lane
// finishes.  A thread takes 16-byte pieces (coalesced): piece at chunk offset o gives s = sum d,
// w = sum k d_k, and (len - o) s - w to b.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t ADLER_CHUNK = 16384;
__global__ __launch_bounds__(256) void k_adler_part(const uint8_t* __restrict__ in, uint32_t n, DevScalars* sc) {
    __shared__ uint32_t sa[4], sb[4];
    const uint32_t tid = threadIdx.x;
    const uint64_t c0 = (uint64_t)blockIdx.x * ADLER_CHUNK;
    const uint32_t len = n - c0 < ADLER_CHUNK ? (uint32_t)(n - c0) : ADLER_CHUNK;
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    uint32_t a = 0, b = 0;
#pragma unroll
    for (uint32_t i = 0; i < ADLER_CHUNK / (256 * 16); i++) {
        const uint32_t o = (i * 256 + tid) * 16;
        if (o >= len) break;
        uint32_t v[4] = {0, 0, 0, 0};
        if (o + 16 <= len && aligned) {
            const uint4 q = *reinterpret_cast<const uint4*>(in + c0 + o);
            v[0] = q.x;
            v[1] = q.y;
            v[2] = q.z;
            v[3] = q.w;
        } else {
            for (uint32_t k = 0; k < 16; k++)
                if (o + k < len) v[k >> 2] |= (uint32_t)in[c0 + o + k] << (8 * (k & 3));
        }
        uint32_t s = 0, w = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t d = (v[k >> 2] >> (8 * (k & 3))) & 0xff;  // (bytes past the end are 0)
            s += d;
            w += k * d;
        }
        a += s;
        b += (len - o) * s - w;  // <= 4 * 16384 * 4080 < 2^32
    }
    b %= 65521u;
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    if ((tid & 63) == 0) {
        sa[tid >> 6] = a;
        sb[tid >> 6] = b;
    }
    __syncthreads();
    if (tid == 0) {
        const uint64_t A = ((uint64_t)sa[0] + sa[1] + sa[2] + sa[3]) % 65521u;
        const uint64_t B = ((uint64_t)sb[0] + sb[1] + sb[2] + sb[3]) % 65521u;
        const uint64_t after = (uint64_t)n - c0 - len;
        atomicAdd(reinterpret_cast<unsigned long long*>(&sc->adler_a), (unsigned long long)A);
        atomicAdd(reinterpret_cast<unsigned long long*>(&sc->adler_b), (unsigned long long)((B + (after % 65521u) * A) % 65521u));
    }
}
__global__ void k_adler_fold(uint32_t n, DevScalars* sc) {
    if (threadIdx.x || blockIdx.x) return;
    const uint64_t a = (1 + sc->adler_a) % 65521u, b = (n + sc->adler_b) % 65521u;
    sc->adler = (uint32_t)((b << 16) | a);
}

// zlib framing written on the device (lib.rs:182-198, zlib.rs:59-62): 78 9C, Adler-32 BE.
__global__ void k_zlib_frame(DevScalars* sc, uint8_t* out, uint32_t trailer) {
    if (threadIdx.x || blockIdx.x) return;
    uint64_t nbytes = (sc->total_bits + 7) / 8;
    out[0] = 0x78;
    out[1] = 0x9C;
    if (!trailer) return;
    uint32_t a = sc->adler;
    out[2 + nbytes + 0] = (uint8_t)(a >> 24);
    out[2 + nbytes + 1] = (uint8_t)(a >> 16);
    out[2 + nbytes + 2] = (uint8_t)(a >> 8);
    out[2 + nbytes + 3] = (uint8_t)a;
}

// ---------------------------------------------------------------------------------------------
// CRC-32 (RFC 1952 section 8; crate gzip-header 1.0 `Crc`, as used by lib.rs:258-266 and
// writer.rs:408-444).  k_crc_part: every thread runs the table-driven CRC (four bytes per step, four
// 256-entry tables in LDS) over its own 512-byte chunk; the chunks of a workgroup are staged through
// LDS in 128-byte pieces so that eight lanes read one full 128-byte line of HBM and a thread then
// reads its piece bank-conflict free (row stride 33 words).  k_crc_fold: CRCs are linear in the
// sense of zlib's crc32_combine, crc(A||B) = x^(8|B|) * crc(A) + crc(B) over GF(2)[x] mod P, so
// crc(input) = sum_i x^(8 * bytes after chunk i) * crc(chunk_i): every thread raises its own factor
// by squaring and the products are XOR-ed together.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t CRC_CHUNK = 512;   // bytes per thread
constexpr uint32_t CRC_PIECE = 128;   // bytes per thread per staging round
constexpr uint32_t CRC_POLY = 0xEDB88320u;

// a(x) * b(x) mod P, reflected representation (bit 31 = x^0)
__device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
        p ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
    }
    return p;
}
// x^(8 * nbytes) mod P
__device__ uint32_t crc_xpow8(uint64_t nbytes) {
    uint32_t r = 0x80000000u;  // x^0
    uint32_t sq = 0x00800000u; // x^8
    while (nbytes) {
        if (nbytes & 1) r = crc_mulmod(sq, r);
        sq = crc_mulmod(sq, sq);
        nbytes >>= 1;
    }
    return r;
}

__global__ __launch_bounds__(256) void k_crc_part(const uint8_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ part) {
    __shared__ uint32_t T[4][256];
    __shared__ uint32_t stage[256 * 33];
    const uint32_t tid = threadIdx.x;
    {
        uint32_t c = tid;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
        T[0][tid] = c;
    }
    __syncthreads();
    for (int t = 1; t < 4; t++) {
        uint32_t v = T[t - 1][tid];
        T[t][tid] = (v >> 8) ^ T[0][v & 0xff];
        __syncthreads();
    }
    const uint64_t tile = (uint64_t)blockIdx.x * 256 * CRC_CHUNK;
    const uint64_t my0 = tile + (uint64_t)tid * CRC_CHUNK;
    const uint32_t mylen = my0 >= n ? 0u : (n - my0 < CRC_CHUNK ? (uint32_t)(n - my0) : CRC_CHUNK);
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    uint32_t crc = 0xFFFFFFFFu;
    for (uint32_t piece = 0; piece < CRC_CHUNK / CRC_PIECE; piece++) {
        // stage: lane j of round r loads 16 bytes of chunk (r * 32 + j / 8), part j % 8
        for (uint32_t r = 0; r < 8; r++) {
            const uint32_t c = r * 32 + tid / 8, part16 = tid % 8;
            const uint64_t g = tile + (uint64_t)c * CRC_CHUNK + piece * CRC_PIECE + part16 * 16;
            uint32_t v[4] = {0, 0, 0, 0};
            if (g + 16 <= n && aligned) {
                const uint4 q = *reinterpret_cast<const uint4*>(in + g);
                v[0] = q.x;
                v[1] = q.y;
                v[2] = q.z;
                v[3] = q.w;
            } else {
                for (uint32_t b = 0; b < 16; b++)
                    if (g + b < n) v[b >> 2] |= (uint32_t)in[g + b] << (8 * (b & 3));
            }
            uint32_t* dst = stage + c * 33 + part16 * 4;
            dst[0] = v[0];
            dst[1] = v[1];
            dst[2] = v[2];
            dst[3] = v[3];
        }
        __syncthreads();
        const uint32_t done = piece * CRC_PIECE;
        const uint32_t here = mylen > done ? (mylen - done < CRC_PIECE ? mylen - done : CRC_PIECE) : 0u;
        const uint32_t* src = stage + tid * 33;
        uint32_t w = 0;
        for (; w * 4 + 4 <= here; w++) {
            crc ^= src[w];
            crc = T[3][crc & 0xff] ^ T[2][(crc >> 8) & 0xff] ^ T[1][(crc >> 16) & 0xff] ^ T[0][crc >> 24];
        }
        for (uint32_t b = w * 4; b < here; b++) {
            const uint32_t d = (src[b >> 2] >> (8 * (b & 3))) & 0xff;
            crc = T[0][(crc ^ d) & 0xff] ^ (crc >> 8);
        }
        __syncthreads();
    }
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + tid;
    if (mylen) part[idx] = ~crc;
}

__global__ __launch_bounds__(256) void k_crc_fold(uint32_t n, uint32_t nchunks, const uint32_t* __restrict__ part,
                                                  DevScalars* sc) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t v = 0;
    if (i < nchunks) {
        const uint64_t end = (uint64_t)(i + 1) * CRC_CHUNK;
        const uint64_t after = end < n ? n - end : 0;
        v = part[i];
        if (after) v = crc_mulmod(crc_xpow8(after), v);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) v ^= __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicXor(&sc->crc, v);
}

// gzip framing written on the device (lib.rs:250-266): the caller's header bytes, then after the
// stream CRC-32 and the input length mod 2^32, both little endian.
__global__ void k_gzip_frame(DevScalars* sc, uint8_t* out, const uint8_t* hdr, uint32_t hdr_len, uint32_t in_len,
                             uint32_t trailer) {
    if (blockIdx.x) return;
    for (uint32_t i = threadIdx.x; i < hdr_len; i += blockDim.x) out[i] = hdr[i];
    if (threadIdx.x || !trailer) return;
    uint64_t nbytes = (sc->total_bits + 7) / 8;
    uint8_t* t = out + hdr_len + nbytes;
    uint32_t c = sc->crc;
    for (int k = 0; k < 4; k++) t[k] = (uint8_t)(c >> (8 * k));
    for (int k = 0; k < 4; k++) t[4 + k] = (uint8_t)(in_len >> (8 * k));
}

}  // namespace mi355

// k_emit with the steps from adv, or worked out by the kernel itself when `steps` -- what goes into its adv argument -- is nullptr
#define MI355_LAUNCH_EMIT(MODE, steps, grid, st, ...)                                                    \
    do {                                                                                                 \
        if (steps)                                                                                       \
            hipLaunchKernelGGL((k_emit<MODE, false>), grid, dim3(256), 0, st, __VA_ARGS__);              \
        else                                                                                             \
            hipLaunchKernelGGL((k_emit<MODE, true>), grid, dim3(256), 0, st, __VA_ARGS__);               \
    } while (0)
// k_pack with workgroups of 1024 threads where the call has fewer parts than the device compute units (a part's rounds are
// its time then: a 167 KB file 24.6 -> us), of MI355_PACK_THREADS otherwise (more workgroups a unit)
#define MI355_LAUNCH_HIST(units, n_cu, st, ...)                                                              \
    do {                                                                                                     \
        if ((units) <= ((n_cu) ? (n_cu) : 256u))                                                             \
            hipLaunchKernelGGL((k_block_hist<1024>), dim3(units), dim3(1024), 0, st, __VA_ARGS__);           \
        else                                                                                                 \
            hipLaunchKernelGGL((k_block_hist<256>), dim3(units), dim3(256), 0, st, __VA_ARGS__);             \
    } while (0)
#define MI355_LAUNCH_PACK(units, n_cu, st, ...)                                                              \
    do {                                                                                                     \
        if ((units) <= ((n_cu) ? (n_cu) : 256u))                                                             \
            hipLaunchKernelGGL((k_pack<PKT_SMALL>), dim3(units), dim3(PKT_SMALL), 0, st, __VA_ARGS__);       \
        else                                                                                                 \
            hipLaunchKernelGGL((k_pack<PKT_LARGE>), dim3(units), dim3(PKT_LARGE), 0, st, __VA_ARGS__);       \
    } while (0)
#include "deflate_host.inc"
#include "deflate_shard.inc"
#include "deflate_long.inc"
#include "deflate_multi.inc"
