// stages.h -- the per-element stage functions of the MI355X DEFLATE encode path.
//
// The encode path of deflate-rs (one serial sliding-window loop, /root/reference/src/lz77.rs,
// matching.rs, huffman_lengths.rs, ...) is re-cut here into data-parallel stages whose
// per-element work lives in this header as inline functions.  The HIP kernels
// (deflate_kernels.hip) call them from device code; tests/hostsim/ compiles the same
// functions for the host to diff every intermediate against the CPU oracle without a GPU.
// Nothing in here is a CPU fallback of the product: the product path only ever runs the
// kernels.
//
// Stage map (DESIGN.md has the derivations):
//   links   link[p]  = distance to the previous position with the same 15-bit 3-byte hash
//                      (chained_hash_table.rs:55-62,118-158), 0 if none within 32768
//   match   M[p]     = result of matching.rs:87-166 longest_match for prev_length = 0; the
//                      parser applies the prev_length filter (it is only a filter, see
//                      DESIGN.md "M is a pure function of the data")
//   step    step(j)  = what the lazy / greedy / rle parser (lz77.rs:305-547, rle.rs:23-71)
//                      emits when it is at position j with no pending match: a run of
//                      literals, at most one match, and the next such "restart" position
//   path    the true parse is the chain 0 -> next(0) -> next(next(0)) ...
//   blocks  every 31744 tokens (output_writer.rs:19) -> histogram -> Huffman lengths
//           (length_encode.rs:347-415) -> block type (huffman_lengths.rs:167-287) -> bits
#ifndef MI355_DEFLATE_STAGES_H
#define MI355_DEFLATE_STAGES_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MI355_HD __host__ __device__ __forceinline__
#else
#define MI355_HD inline
#endif

namespace mi355 {

// ---- constants (reference file:line in comments) --------------------------
enum : uint32_t {
    WINDOW_SIZE = 32768,        // chained_hash_table.rs:1
    MIN_MATCH = 3,              // huffman_table.rs:20
    MAX_MATCH = 258,            // huffman_table.rs:21
    TOO_FAR = 8192,             // lz77.rs:276
    MAX_BUFFER_LENGTH = 31744,  // output_writer.rs:19
    MAX_STORED_BLOCK_LENGTH = 32767,  // stored_block.rs:11
    NUM_LL = 286,               // huffman_table.rs:14
    NUM_DIST = 30,              // huffman_table.rs:10
    END_OF_BLOCK = 256,         // huffman_table.rs:28
    MAX_JUMP = 520,             // bound on next(j)-j: <=257 lazy deferrals + 258 match bytes
    ZONE = 576,                 // entry zone of a segment (>= MAX_JUMP, multiple of 64)
};

enum ParseMode : uint32_t { MODE_GREEDY = 0, MODE_LAZY = 1, MODE_RLE = 2 };
enum BlockType : uint32_t { BT_STORED = 0, BT_FIXED = 1, BT_DYNAMIC = 2 };

// Token = LZValue (lzvalue.rs:42-76): litlen | distance << 16; distance 0 => literal.
MI355_HD uint32_t tok_literal(uint32_t byte) { return byte; }
MI355_HD uint32_t tok_match(uint32_t len, uint32_t dist) { return (len - MIN_MATCH) | (dist << 16); }
MI355_HD uint32_t tok_cover(uint32_t t) { return (t >> 16) ? (t & 0xff) + MIN_MATCH : 1; }

// M entry: len | dist << 16 (len 0 = nothing found; len may be 2, which the parsers ignore).
MI355_HD uint32_t m_pack(uint32_t len, uint32_t dist) { return len | (dist << 16); }
MI355_HD uint32_t m_len(uint32_t m) { return m & 0xffff; }
MI355_HD uint32_t m_dist(uint32_t m) { return m >> 16; }

// ---- hash (chained_hash_table.rs:55-62) --------------------------------------------------
// Three rolling updates ((h << 5) ^ b) & 0x7fff leave exactly this function of 3 bytes.
MI355_HD uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) { return ((a & 31u) << 10) ^ (b << 5) ^ c; }

// Quirk Q1 (lz77.rs:628-638): when the first block fills inside the first window, the rolling
// hash is re-warmed with data[0], data[1]; the next two inserted positions w, w+1 then hash
// (b0,b1,d[w+2]) and (b1,d[w+2],d[w+3]).  `on` = 0 in the common case.
// The same re-warm happens at a sync-flush point F <= 32768 when the write that follows the flush
// does not itself start processing (add_initial is per call, lz77.rs:601-614): `pts` holds those
// points, ascending, `m` of them (streaming API only).
struct HashOverride {
    uint64_t pos;
    uint32_t b0, b1;
    uint32_t on;
    uint32_t m;
    const uint32_t* pts;
    // Two more effects of write calls around a sync flush (streaming API only; lz77.rs:601-614).  The
    // first write after a flush point F > 2 re-adds the positions F-2 and F-1 with its first two bytes; a
    // write of ONE byte re-adds only F-2: F-1 is never filed (`hol`, ascending), and the rolling hash is a
    // byte behind when F and F+1 are filed -- under (d[F-1], d[F], d[F+2]) and (d[F], d[F+2], d[F+3])
    // (`skw` = those F, ascending).  After a flush at F <= 2 nothing is re-added: the positions before F
    // are holes as well, and F is a re-warm point (`pts`).
    uint32_t ns;
    const uint32_t* skw;
    uint32_t nh;
    const uint32_t* hol;
};

MI355_HD bool list_has(const uint32_t* v, uint32_t n, uint32_t p) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        uint32_t x = v[mid];
        if (x == p) return true;
        if (x < p)
            lo = mid + 1;
        else
            hi = mid;
    }
    return false;
}
MI355_HD bool rewarm_listed(const HashOverride& ov, uint32_t p) { return list_has(ov.pts, ov.m, p); }
MI355_HD bool hash_hole(const HashOverride& ov, uint64_t p) { return ov.nh && p <= 0xFFFFFFFFull && list_has(ov.hol, ov.nh, (uint32_t)p); }

// the first two bytes of position p's 3-byte hash input, replaced where a re-warm applies; returned
// packed (a | b << 8) -- by value, so that nothing has to live in memory
MI355_HD uint32_t rewarm_ab(const HashOverride& ov, uint64_t p, uint32_t a, uint32_t b) {
    if ((ov.on | ov.m) && p <= WINDOW_SIZE + 1) {
        bool first = (ov.on && p == ov.pos) || (ov.m && rewarm_listed(ov, (uint32_t)p));
        bool second = (ov.on && p == ov.pos + 1) || (ov.m && p > 0 && rewarm_listed(ov, (uint32_t)p - 1));
        if (first) {
            a = ov.b0;
            b = ov.b1;
        } else if (second) {
            a = ov.b1;
        }
    }
    return a | (b << 8);
}
// the same with the one-byte-write skew in front of it (a re-warm at the same point replaces the rolling
// hash and wins); prev = d[p-1]
MI355_HD uint32_t skewed_ab(const HashOverride& ov, uint64_t p, uint32_t prev, uint32_t a, uint32_t b) {
    if (ov.ns && p <= 0xFFFFFFFFull) {
        if (list_has(ov.skw, ov.ns, (uint32_t)p)) {
            b = a;
            a = prev;
        } else if (p > 0 && list_has(ov.skw, ov.ns, (uint32_t)p - 1)) {
            a = prev;
        }
    }
    return rewarm_ab(ov, p, a, b);
}

template <class Bytes>
MI355_HD uint32_t position_hash(const Bytes& by, uint64_t p, const HashOverride& ov) {
    const uint32_t ab = skewed_ab(ov, p, p ? by(p - 1) : 0u, by(p), by(p + 1));
    return hash3(ab & 0xff, ab >> 8, by(p + 2));
}

// index of the lowest set bit, all ones for zero -- which an OR keeps all ones, so that a zero word drops out of
// a minimum (v_ffbl_b32 does exactly this; __builtin_ffs - 1 costs a compare and a select on top)
MI355_HD uint32_t first_bit_or_ones(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return x ? (uint32_t)__builtin_ctz(x) : 0xFFFFFFFFu;
#endif
}

MI355_HD uint32_t ctz32(uint32_t x) { return (uint32_t)__builtin_ctz(x); }

// ---- match (matching.rs:87-166 with prev_length = 0) -------------------------------------
// `W` gives unaligned little-endian 4-byte loads and u16 links in one index space (window
// coordinates on the GPU, absolute on the host).  p = index of the searched position,
// max_len = min(N - P, 258) (matching.rs:112), checks = max_hash_checks.  If checks_q != 0 the
// state after checks_q iterations is also reported (lz77.rs:351-355: the reduced budget used
// when prev_length >= 32).  Iterations that end the chain count like the reference's do.
template <class W>
MI355_HD void match_walk(const W& w, uint32_t p, uint32_t max_len, uint32_t checks, uint32_t checks_q,
                         uint32_t* out_m, uint32_t* out_mq) {
    uint32_t best = 1, best_d = 0;
    uint32_t cand = p;
    uint32_t probe = w.load32(p) & 0xffffu;  // bytes best-1, best of P
    uint32_t mq = 0;
    bool have_q = (checks_q == 0);
    for (uint32_t i = 0; i < checks; i++) {
        if (!have_q && i == checks_q) {
            mq = m_pack(best_d ? best : 0, best_d);
            have_q = true;
        }
        uint32_t d = w.link(cand);
        if (d == 0) break;                     // chain end (matching.rs:127, self loop / older)
        if (d > cand) break;                   // cannot happen in a well-formed table
        cand -= d;
        if (p - cand > WINDOW_SIZE) break;     // current_head < limit (matching.rs:102-106,127)
        if ((w.load32(cand + best - 1) & 0xffffu) == probe) {  // matching.rs:141-143
            uint32_t len = 0;                  // get_match_length matching.rs:67-72
            while (len < max_len) {
                uint32_t x = w.load32(p + len) ^ w.load32(cand + len);
                if (x) {
                    len += ctz32(x) >> 3;
                    break;
                }
                len += 4;
            }
            if (len > max_len) len = max_len;
            if (len > best) {                  // matching.rs:149-156
                best = len;
                best_d = p - cand;
                if (len == max_len) break;
                probe = w.load32(p + best - 1) & 0xffffu;
            }
        }
    }
    uint32_t m = m_pack(best_d ? best : 0, best_d);
    if (!have_q) mq = m;
    *out_m = m;
    *out_mq = mq;
}

// The same walk for a lane that owns `count` positions first, first+stride, ...: U chains are
// kept in flight at once (their LDS reads are independent, which hides the read latency a single
// pointer chase exposes) and a finished chain is replaced at once by the lane's next position, so
// lanes of a wave stay busy although chain lengths differ.  Results are identical to match_walk;
// emit(idx, m, mq) is called once per position idx < nrel (nrel = end of input in W's index space).
template <int U, bool HAS_Q, class W, class Emit>
MI355_HD void match_walk_multi(const W& w, uint32_t first, uint32_t stride, uint32_t count, uint32_t nrel,
                               uint32_t checks, uint32_t checks_q, Emit& emit) {
    uint32_t p[U], cand[U], best[U], bestd[U], probe[U], it[U], maxlen[U], mq[U], d[U], pv[U];
    bool act[U], hq[U], cmp[U];
    uint32_t knext = 0;
    auto start = [&](int s) {
        act[s] = false;
        while (knext < count) {
            uint32_t idx = first + knext * stride;
            knext++;
            if (idx >= nrel) continue;
            if (idx + 2 >= nrel) {  // no hash byte: never searched (lz77.rs:294-301)
                emit(idx, 0u, 0u);
                continue;
            }
            p[s] = idx;
            cand[s] = idx;
            best[s] = 1;
            bestd[s] = 0;
            probe[s] = w.load32(idx) & 0xffffu;
            it[s] = 0;
            maxlen[s] = nrel - idx < (uint32_t)MAX_MATCH ? nrel - idx : (uint32_t)MAX_MATCH;
            mq[s] = 0;
            hq[s] = !HAS_Q;
            act[s] = true;
            return;
        }
    };
    auto finish = [&](int s) {
        uint32_t m = m_pack(bestd[s] ? best[s] : 0, bestd[s]);
        emit(p[s], m, hq[s] ? (HAS_Q ? mq[s] : m) : m);
        start(s);
    };
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < U; s++) start(s);
    for (;;) {
        bool any = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int s = 0; s < U; s++) any = any || act[s];
        if (!any) break;
        // A: the U independent link reads
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int s = 0; s < U; s++) d[s] = act[s] ? w.link(cand[s]) : 0u;
        // B: chain end tests, then the U independent probe reads
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int s = 0; s < U; s++) {
            cmp[s] = false;
            if (!act[s]) continue;
            if (HAS_Q && !hq[s] && it[s] == checks_q) {
                mq[s] = m_pack(bestd[s] ? best[s] : 0, bestd[s]);
                hq[s] = true;
            }
            if (it[s] >= checks || d[s] == 0 || d[s] > cand[s]) {
                finish(s);
                continue;
            }
            cand[s] -= d[s];
            if (p[s] - cand[s] > WINDOW_SIZE) {
                finish(s);
                continue;
            }
            pv[s] = w.load32(cand[s] + best[s] - 1) & 0xffffu;
            cmp[s] = true;
        }
        // C: compare, extend on a hit
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int s = 0; s < U; s++) {
            if (!cmp[s]) continue;
            it[s]++;
            if (pv[s] != probe[s]) continue;
            uint32_t len = 0;
            while (len < maxlen[s]) {
                uint32_t x = w.load32(p[s] + len) ^ w.load32(cand[s] + len);
                if (x) {
                    len += ctz32(x) >> 3;
                    break;
                }
                len += 4;
            }
            if (len > maxlen[s]) len = maxlen[s];
            if (len > best[s]) {
                best[s] = len;
                bestd[s] = p[s] - cand[s];
                if (len == maxlen[s]) {
                    finish(s);
                    continue;
                }
                probe[s] = w.load32(p[s] + len - 1) & 0xffffu;
            }
        }
    }
}

// Third formulation of the same walk, shaped for gfx950: per chain step a slot issues BOTH reads
// that depend on the current candidate together -- its link (the next candidate) and its probe
// bytes -- so a step costs one LDS round trip, and U slots per lane overlap theirs.  Everything
// that is rare per step is taken out of the step loop: a slot whose probe hits is PARKED, a slot
// whose chain ended is FIN; only when `policy` says so (every 8th step on the GPU) the parked
// slots are extended (get_match_length) in one dense loop and the finished ones report and take
// the lane's next position from `next()`.  The common step stays ~25 VALU instructions per slot.
// Results are identical to match_walk, including tie-breaks.
enum : uint32_t { NO_POS = 0xFFFFFFFFu };

struct ServiceAlways {  // host policy: service parked / finished slots at once, extend to the end
    MI355_HD bool operator()(bool pending, bool /*walking*/, uint32_t /*iter*/) const { return pending; }
    MI355_HD bool keep_extending(bool any, uint32_t /*round*/) const { return any; }
};

// `lim(idx)` = end of the data the encoder had when it searched position idx (W's index space): the
// end of input, or the next flush point (a sync flush makes the reference parse what it has, so
// matches and hash bytes stop there: lz77.rs:593,627).
struct ConstLimit {
    uint32_t n;
    MI355_HD uint32_t operator()(uint32_t) const { return n; }
};

#ifndef MI355_STAT
#define MI355_STAT_DECL
#define MI355_STAT(i, v)
#define MI355_STAT_FLUSH(policy)
#endif

// A per-lane flag that lives across the iterations of match_walk_park.  On the GPU it is a lane mask
// (one bit per lane, the same 64-bit value in every lane, i.e. a pair of scalar registers): the
// state bookkeeping of the common step then runs on the scalar unit and the vector ALU, which bounds
// k_match, only does the arithmetic.  (As plain bools the compiler keeps loop-carried flags as 0/1
// in vector registers and spends three vector instructions per flag and step on them.)  On the
// host it is 0 or 1.
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint64_t lane_flag;
__device__ __forceinline__ lane_flag lf_of(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool lf_me(lane_flag f) { return __builtin_amdgcn_inverse_ballot_w64(f); }
__device__ __forceinline__ bool lf_any(lane_flag f) { return f != 0; }
__device__ __forceinline__ lane_flag lf_not(lane_flag f) { return ~f; }
#else
typedef uint32_t lane_flag;  // 0 or 1
inline lane_flag lf_of(bool b) { return b ? 1u : 0u; }
inline bool lf_me(lane_flag f) { return f != 0; }
inline bool lf_any(lane_flag f) { return f != 0; }
inline lane_flag lf_not(lane_flag f) { return f ^ 1u; }
#endif

template <int U, bool HAS_Q, class W, class Emit, class Next, class Policy, class Lim>
MI355_HD void match_walk_park(const W& w, Next& next, const Lim& lim, uint32_t checks, uint32_t checks_q, Emit& emit,
                              const Policy& policy) {
    // A slot is walking (on a chain), parked (the current candidate passed the probe and waits for its
    // compare), finished (result ready -- p == NO_POS: nothing to report -- waits to report and to take
    // a new position) or idle (none of the three).  All lanes of a wave stay in the loop until the
    // whole wave is out of work (an idle slot does nothing but take part in the reads).
    uint32_t p[U], cand[U], best[U], bestd[U], probe[U], it[U], maxlen[U], dsave[U], len[U], mq[U];
    uint32_t rd[U], rv[U], ra[U], rb[U], ra2[U], rb2[U];
    int32_t low[U];  // p - 32768: the lowest candidate in reach (matching.rs:102-106)
    lane_flag walk[U], park[U], fin[U], ext[U], hq[U];  // hq: the quarter-budget result is taken
#if defined(__HIP_DEVICE_COMPILE__)
#define MI355_UNROLL _Pragma("unroll")
#else
#define MI355_UNROLL
#endif
    MI355_STAT_DECL
    MI355_UNROLL
    for (int s = 0; s < U; s++) {
        walk[s] = lf_of(false);
        park[s] = lf_of(false);
        fin[s] = lf_of(true);  // nothing to report yet: the first service hands out the first positions
        p[s] = NO_POS;
        cand[s] = 0;  // (every slot takes part in the reads of the common step: keep them in range)
        low[s] = 0;
        best[s] = 1;
        bestd[s] = 0;
        probe[s] = 0;
        it[s] = 0;
        maxlen[s] = 0;
        dsave[s] = 0;
        len[s] = 0;
        mq[s] = 0;
        hq[s] = lf_of(true);
    }
    for (uint32_t iter = 0;; iter++) {
        lane_flag walking = lf_of(false), pending = lf_of(false);
        MI355_UNROLL
        for (int s = 0; s < U; s++) {
            walking = walking | walk[s];
            pending = pending | park[s] | fin[s];
            MI355_STAT(1, lf_me(walk[s]) ? 1u : 0u)
            MI355_STAT(4, lf_me(park[s]) ? 1u : 0u)
            MI355_STAT(5, lf_me(fin[s]) ? 1u : 0u)
            MI355_STAT(6, (!lf_me(walk[s]) && !lf_me(park[s]) && !lf_me(fin[s])) ? 1u : 0u)
        }
        MI355_STAT(0, 1u)
        if (!lf_any(walking) && !lf_any(pending)) break;
        // The common step, branch free: link and probe of the current candidate are read together
        // (for a slot that is not walking the reads are harmless and their results unused), then
        // matching.rs:124-143 as selects.
        MI355_UNROLL
        for (int s = 0; s < U; s++) {
            rd[s] = w.link_far(cand[s]);  // 0xFFFF = no earlier position: fails the distance test below
            rv[s] = w.load32(cand[s] + best[s] - 1);
        }
        MI355_UNROLL
        for (int s = 0; s < U; s++) {
            const lane_flag hit = walk[s] & lf_of((rv[s] & 0xffffu) == probe[s]);   // :141-143, compare deferred
            const uint32_t c = cand[s] - rd[s];
            // (no link: rd = 0xFFFF and c falls below p - 65535 < low, as c <= cand <= p)
            const lane_flag ok = lf_of(it[s] < checks) & lf_of((int32_t)c >= low[s]);
            const lane_flag miss = walk[s] & lf_not(hit);
            const lane_flag adv = miss & ok;
            if (HAS_Q) {
                const lane_flag cap = miss & lf_not(hq[s]) & lf_of(it[s] == checks_q && it[s] < checks);
                mq[s] = lf_me(cap) ? m_pack(bestd[s] ? best[s] : 0, bestd[s]) : mq[s];
                hq[s] = hq[s] | cap;
            }
            dsave[s] = lf_me(hit) ? rd[s] : dsave[s];
            park[s] = park[s] | hit;
            fin[s] = fin[s] | (miss & lf_not(ok));
            walk[s] = adv;
            cand[s] = lf_me(adv) ? c : cand[s];
            it[s] += lf_me(adv) ? 1u : 0u;
        }
        walking = lf_of(false);
        pending = lf_of(false);
        MI355_UNROLL
        for (int s = 0; s < U; s++) {
            walking = walking | walk[s];
            pending = pending | park[s] | fin[s];
        }
        if (!policy(lf_me(pending), lf_me(walking), iter)) continue;
        MI355_STAT(2, 1u)
        // ---- service, written as selects as well: the few lanes that need a part of it are spread
        // over the wave, so every part runs for the whole wave anyway ----
        // (1) get_match_length (matching.rs:67-72) for the parked slots, eight bytes per round (most
        // matches on text end inside the first round); a slot whose compare the policy cuts short
        // stays parked and goes on next time
        MI355_UNROLL
        for (int s = 0; s < U; s++) ext[s] = park[s];
        for (uint32_t round = 0;; round++) {
            lane_flag any = lf_of(false);
            MI355_UNROLL
            for (int s = 0; s < U; s++) any = any | ext[s];
            if (!policy.keep_extending(lf_me(any), round)) break;
            MI355_STAT(3, 1u)
            MI355_UNROLL
            for (int s = 0; s < U; s++) {
                const uint32_t a = (p[s] == NO_POS ? 0u : p[s]) + len[s], b = cand[s] + len[s];
                ra[s] = w.load32(a);
                rb[s] = w.load32(b);
                ra2[s] = w.load32(a + 4);
                rb2[s] = w.load32(b + 4);
            }
            MI355_UNROLL
            for (int s = 0; s < U; s++) {
                const uint64_t z = ((uint64_t)(ra2[s] ^ rb2[s]) << 32) | (ra[s] ^ rb[s]);
                const uint32_t n8 = z ? ((uint32_t)__builtin_ctzll(z) >> 3) : 8u;
                uint32_t nl = len[s] + n8;
                nl = nl < maxlen[s] ? nl : maxlen[s];
                const lane_flag stop = lf_of(n8 < 8 || nl == maxlen[s]);
                len[s] = lf_me(ext[s]) ? nl : len[s];
                ext[s] = ext[s] & lf_not(stop);
            }
        }
        // (2) matching.rs:149-156 for the compares that are through, then :124-132 (loop header, link,
        // the two chain-end tests) for their slots
        MI355_UNROLL
        for (int s = 0; s < U; s++) {
            const bool done = lf_me(park[s] & lf_not(ext[s]));
            const bool up = done && len[s] > best[s];
            best[s] = up ? len[s] : best[s];
            bestd[s] = up ? p[s] - cand[s] : bestd[s];
            rv[s] = w.load32((p[s] == NO_POS ? 0u : p[s]) + best[s] - 1);
        }
        MI355_UNROLL
        for (int s = 0; s < U; s++) {
            const lane_flag done = park[s] & lf_not(ext[s]);
            const bool dn = lf_me(done);
            const bool full = dn && best[s] == maxlen[s] && bestd[s] == p[s] - cand[s];  // this compare hit max_length
            probe[s] = dn ? (rv[s] & 0xffffu) : probe[s];
            const bool more = dn && !full && it[s] < checks;
            if (HAS_Q) {
                const lane_flag cap = lf_of(more && it[s] == checks_q) & lf_not(hq[s]);
                mq[s] = lf_me(cap) ? m_pack(bestd[s] ? best[s] : 0, bestd[s]) : mq[s];
                hq[s] = hq[s] | cap;
            }
            const uint32_t c = cand[s] - dsave[s];
            const lane_flag go = lf_of(more && (int32_t)c >= low[s]);  // dsave == 0xFFFF ("none") fails here
            cand[s] = lf_me(go) ? c : cand[s];
            it[s] += lf_me(go) ? 1u : 0u;
            walk[s] = walk[s] | go;
            fin[s] = fin[s] | (done & lf_not(go));
            park[s] = park[s] & lf_not(done);
            len[s] = dn ? 0u : len[s];  // the next compare of this slot starts from byte 0 again
        }
        // (3) finished slots report and take one new position each; a position without a candidate is
        // set up as finished (result 0) and reported at the next service
        MI355_UNROLL
        for (int s = 0; s < U; s++) {
            const bool f = lf_me(fin[s]);
            uint32_t idx = NO_POS;
            if (f) {
                if (p[s] != NO_POS) {
                    uint32_t m = m_pack(bestd[s] ? best[s] : 0, bestd[s]);
                    emit(p[s], m, (HAS_Q && lf_me(hq[s])) ? mq[s] : m);
                }
                idx = next();
            }
            const bool have = idx != NO_POS;
            const uint32_t ix = have ? idx : 0u;
            const uint32_t nrel = lim(ix);
            const bool inr = have && ix < nrel;
            const bool search = inr && ix + 2 < nrel;  // else no hash byte: never searched (lz77.rs:294-301)
            const uint32_t p0 = w.load32(ix);
            const uint32_t d = w.link_far(ix);
            const bool ok = search && checks > 0 && d <= (uint32_t)WINDOW_SIZE;
            if (HAS_Q) {
                mq[s] = f ? 0u : mq[s];
                hq[s] = (hq[s] & lf_not(fin[s])) | lf_of(f && ok && checks_q == 0);
            }
            p[s] = f ? (inr ? ix : (uint32_t)NO_POS) : p[s];
            low[s] = f ? (int32_t)ix - (int32_t)WINDOW_SIZE : low[s];
            cand[s] = f ? (ok ? ix - d : ix) : cand[s];
            best[s] = f ? 1u : best[s];
            bestd[s] = f ? 0u : bestd[s];
            probe[s] = f ? (p0 & 0xffffu) : probe[s];  // bytes 0,1 of P (matching.rs:110,141)
            it[s] = f ? (ok ? 1u : 0u) : it[s];
            const uint32_t left = nrel - ix;
            maxlen[s] = f ? (search ? (left < (uint32_t)MAX_MATCH ? left : (uint32_t)MAX_MATCH) : 0u) : maxlen[s];
            len[s] = f ? 0u : len[s];
            walk[s] = walk[s] | lf_of(f && ok);
            fin[s] = lf_of(f && have && !ok);
        }
    }
    MI355_STAT_FLUSH(policy)
#undef MI355_UNROLL
}

// ---- the same walk over hash-sorted positions (k_sort + k_match3) --------------------------------
// The positions of every 32 KiB epoch are sorted by (hash, position) -- S_e[0..J), bucket h =
// S_e[B_e[h] .. B_e[h+1]) -- so the chain of matching.rs:124-132 for the entry j of bucket h (position p)
// is S_e[j-1], S_e[j-2], ... down to B_e[h], then the bucket's entries of the previous epoch from
// B_{e-1}[h+1]-1 downwards while they lie within 32768 of p: plain descending array reads instead of a
// pointer chase, consecutive entries of a bucket (= consecutive lanes of a wave) read consecutive addresses
// and have chains of nearly the same length.
// Index space of `W::sidx(i)`: i < 32768 = entry i of the previous epoch's array, i >= 32768 = entry
// i - 32768 of the own epoch's; values are positions relative to their epoch.
// (Two earlier formulations of this walk -- SortedLane: runs with a park state; SwLean: the predicated form
// k_match2 ran over a byte image -- are in the history of this file.)
enum : uint32_t { SW_OWN = 32768 };
MI355_HD lane_flag lf_and_not(lane_flag a, lane_flag b) { return a & lf_not(b); }

// ---- the sorted walk as the GPU runs it (k_match3): pair table, probes eight at a time -------------------------
// What longest_match computes for prev_length = 0 is a pure function of the candidate list: among the first
// K candidates of the chain that lie within 32768 bytes (matching.rs:119-132), the one with the longest
// common prefix with P, the nearest among equals, if that prefix is at least 2 bytes (matching.rs:149-156
// only ever replaces the best by a strictly longer one).  The two-byte probe at best-1, best
// (matching.rs:141-143) never rejects a candidate that would improve the result, so ANY test that passes
// every candidate the probe passes gives the same answer: a candidate looked at in vain is compared and
// dropped.  k_match3 uses that freedom in one place only -- it reads the probe bytes of EIGHT consecutive
// candidates before it looks at the first answer, so a lane that finds a hit has asked about up to seven
// candidates it does not need yet (and, at the end of a run, about up to seven entries beyond it, which the
// service discards by index).  How the walk is laid out:
//   * the window's bytes sit in LDS as a table of PAIRS, T[k] = byte k | byte k+1 << 8: the probe is one
//     aligned two-byte read at any k (the byte image needed two one-byte reads and a shift per candidate),
//     and the sorted arrays hold 2 * position, so that entry + bb2 IS the address of the probe;
//   * a lane's candidates are at most two SEGMENTS fixed at set-up -- its own epoch's bucket below it, cut
//     to the budget, then the previous epoch's bucket, cut to what is left -- so the step only tests
//     "entries left", once per group, and the window (matching.rs:102-106), once per group;
//   * the quarter result of lz77.rs:351-355 is taken when the first hit beyond the quarter budget is
//     settled (or the final result, when there is none): no run ends there.
// Coordinates: positions count from the window's first byte (`org` = 0); probe addresses are LDS addresses
// tbase + (position << W::SH) (tbase = 0 on the host), and the sorted arrays hold position << W::SH.

template <bool HAS_Q>
struct SwG {
    uint32_t offb, endb, bb2, lowa2, probe;  // the registers of the step block (offb = 2 * index + 8 of the next entry)
    // probe addresses and probe bytes of the last group -- of four or of eight steps (named: an array selected by
    // lane masks ends up in scratch memory on the GPU)
    uint32_t a0, a1, a2, a3, a4, a5, a6, a7, t0, t1, t2, t3, t4, t5, t6, t7;
    uint32_t prel, maxlen, p16[4], bm1, bestd, low;
    uint32_t offb2, endb2;                   // the segment in the previous epoch's bucket (none: offb2 < endb2)
    uint32_t hoq, hoq2, mq;                  // HAS_Q: the entry offset below which a hit lies beyond the quarter budget, in the
                                             // lane's segment and in its second one (both known at set-up)
    lane_flag walk, hq;
    lane_flag has2;  // a segment in the previous epoch's bucket is still to come
};

// Set a lane up for entry j of its epoch's array.  own_b0 = B_e[h]; [pb0, pb1) = the bucket in the previous
// epoch (pb0 == pb1 for epoch 0).  prel / nrel: position and end of the visible data, `bias` = position of the
// own epoch's first byte.  Returns whether the position is searched at all (lz77.rs:294-301); s.walk = it has
// candidates.
template <bool HAS_Q, class W>
MI355_HD bool swg_setup(SwG<HAS_Q>& s, const W& w, uint32_t j, uint32_t own_b0, uint32_t pb0, uint32_t pb1, uint32_t prel,
                        uint32_t nrel, uint32_t tbase, uint32_t bias, uint32_t checks, uint32_t checks_q) {
    s.prel = prel;
    s.low = prel > (uint32_t)WINDOW_SIZE ? prel - (uint32_t)WINDOW_SIZE : 0u;
    s.bm1 = 0;
    s.bestd = 0;
    s.mq = 0;
    s.hq = lf_of(HAS_Q && checks_q == 0);  // a quarter budget of zero iterations: empty result
    s.a0 = s.a1 = s.a2 = s.a3 = s.a4 = s.a5 = s.a6 = s.a7 = tbase;
    s.t0 = s.t1 = s.t2 = s.t3 = s.t4 = s.t5 = s.t6 = s.t7 = 0;
    const bool search = prel + 2 < nrel && checks > 0;
    const uint32_t left = nrel - prel;
    s.maxlen = search ? (left < (uint32_t)MAX_MATCH ? left : (uint32_t)MAX_MATCH) : 0u;
    uint32_t n1 = j - own_b0;
    n1 = n1 < checks ? n1 : checks;
    uint32_t n2 = pb1 - pb0;
    n2 = n2 < checks - n1 ? n2 : checks - n1;
    s.offb2 = 2 * (pb1 - 1) + 8;
    s.endb2 = s.offb2 + 2 - 2 * n2;  // (n2 == 0: one beyond the first entry)
    const bool own = n1 > 0;
    s.offb = own ? 2 * (SW_OWN + j - 1) + 8 : s.offb2;
    s.endb = own ? s.offb + 2 - 2 * n1 : s.endb2;
    s.has2 = lf_of(own) & lf_of(n2 > 0);
    if (HAS_Q) {
        // rank of the entry at offset ho = (candidates of the segment before) + (seg0 - ho) / 2 + 1; it exceeds the quarter budget
        // exactly when ho < seg0 - 2 (checks_q - before - 1): one compare per service instead of the rank's arithmetic
        s.hoq = s.offb - 2u * (checks_q - 1u);
        s.hoq2 = s.offb2 - 2u * (checks_q - n1 - 1u);  // (the second segment, if there is one, has the n1 of the first before it)
    }
    s.bb2 = tbase + ((own ? bias : 0u) << W::SH);
    s.lowa2 = tbase + (s.low << W::SH);
    w.load16(prel, s.p16);
    s.probe = w.key_at(tbase + (prel << W::SH));
    s.walk = lf_of(prel + 2 < nrel) & lf_of(checks > 0) & lf_of(n1 + n2 > 0);
    return search;
}

// The first candidate of every lane that walks is settled without a step: same-bucket entries nearly always
// share the first two bytes, so the lanes would all leave their first group at its first probe.  Pretends
// that group: a0 / t0 of a hit, offb a group (of `width` steps) further.  Returns the lanes concerned.
template <bool HAS_Q, class W>
MI355_HD lane_flag swg_first(SwG<HAS_Q>& s, const W& w, uint32_t width) {
    const lane_flag st = s.walk;
    uint32_t e2 = 0;
    if (lf_me(st)) e2 = w.sidx((uint32_t)((int32_t)(s.offb - 8) >> 1));
    s.a0 = lf_me(st) ? e2 + s.bb2 : s.a0;
    s.t0 = lf_me(st) ? s.probe : s.t0;
    s.offb -= lf_me(st) ? 2 * width : 0u;
    s.walk = lf_of(false);
    return st;
}

// One group of `width` (four or eight) steps of a walking lane (host twin of the GPU's step block): the entries
// from offb downwards, their probe reads, then the tests in the GPU's order -- the probes in turn (the first hit
// ends it: `d` = its number), window of the last, entries left.  A lane that leaves keeps a / t of this group.
template <bool HAS_Q, class W>
MI355_HD void swg_group_ref(SwG<HAS_Q>& s, const W& w, int* d, uint32_t width) {
    *d = -1;
    if (!lf_me(s.walk)) return;
    const int32_t idx = (int32_t)(s.offb - 8) >> 1;
    uint32_t a[8], t[8];
    for (uint32_t i = 0; i < width; i++) {
        a[i] = w.sidx((uint32_t)(idx - (int32_t)i)) + s.bb2;
        t[i] = w.key_at(a[i]);
    }
    for (uint32_t i = width; i < 8; i++) {
        a[i] = s.a7;
        t[i] = s.t7;
    }
    s.a0 = a[0]; s.a1 = a[1]; s.a2 = a[2]; s.a3 = a[3]; s.a4 = a[4]; s.a5 = a[5]; s.a6 = a[6]; s.a7 = a[7];
    s.t0 = t[0]; s.t1 = t[1]; s.t2 = t[2]; s.t3 = t[3]; s.t4 = t[4]; s.t5 = t[5]; s.t6 = t[6]; s.t7 = t[7];
    s.offb -= 2 * width;
    for (uint32_t i = 0; i < width; i++)
        if (t[i] == s.probe) {
            *d = (int)i;
            s.walk = lf_of(false);
            return;
        }
    if (a[width - 1] < s.lowa2 || (int32_t)s.offb < (int32_t)s.endb) s.walk = lf_of(false);
}

// Settle the lanes that left the last block (`dropped`).  `dany` = those that left at a probe, `asel` = that
// probe's address, `ho` = the offset of its entry (the caller picks them from a / t and offb: which step it was
// is read off the probe bytes; the others left at the end of their group).  Straight-line selects; only a match
// longer than 16 bytes loops.
// RUN1: the variant for inputs made of long runs of one byte (see the long compare below); chosen per launch.
template <bool RUN1, bool HAS_Q, class W>
MI355_HD void swg_service(SwG<HAS_Q>& s, const W& w, uint32_t tbase, uint32_t checks_q, lane_flag dropped, lane_flag dany,
                          uint32_t asel, uint32_t ho) {
    // a probe that "hit" beyond the segment's last entry, or behind a candidate that is out of the window
    // (positions fall along a segment, so the hit's own address tells), is no hit
    // (one comparison per ballot: a ballot of `a && b` makes the compiler turn a lane mask into 0 / 1 values and back)
    const lane_flag hit = dany & lf_of((int32_t)ho >= (int32_t)s.endb) & lf_of(asel >= s.lowa2);  // matching.rs:102-106,127,141-143
    // get_match_length (matching.rs:67-72) against the 16 bytes of P kept in registers
    const uint32_t cpos = ((asel - tbase) >> W::SH) - s.bm1;
    uint32_t q[4];
    w.load16(cpos, q);
    const uint32_t b0 = first_bit_or_ones(q[0] ^ s.p16[0]);
    const uint32_t b1 = first_bit_or_ones(q[1] ^ s.p16[1]) | 32u;
    const uint32_t b2 = first_bit_or_ones(q[2] ^ s.p16[2]) | 64u;
    const uint32_t b3 = first_bit_or_ones(q[3] ^ s.p16[3]) | 96u;
    uint32_t bits = b0 < b1 ? b0 : b1;
    const uint32_t bh = b2 < b3 ? b2 : b3;
    bits = bits < bh ? bits : bh;
    uint32_t len = (bits < 128u ? bits : 128u) >> 3;
    // (a lane whose sixteen bytes are all there is -- maxlen <= 16, the end of the input -- comes along and leaves the loop at once:
    // asking for maxlen > 16 here was two vector and two scalar instructions per service)
    const lane_flag lng = hit & lf_of(len == 16);
    if (lf_any(lng)) {
        // A candidate ONE byte back matches for as long as the position's bytes repeat the candidate's first one
        // (data[c + k] == data[c + k + 1] for every k below the length): when that is so for every lane that goes on --
        // a run of one byte, where each position's first candidate is its neighbour -- the rounds read the position's
        // side only and compare it with that byte four times over (zero fill at Default was all this loop).
        // (RUN1; measured: zero fill at Default 30.5 -> 35.9 GB/s, but the text loses 1 % -- 3.63 -> 3.67 ms -- to the longer
        // service, so it is an instantiation of its own that the host picks for inputs of that kind)
        if (RUN1 && !lf_any(lng & lf_of(s.prel - cpos != 1u))) {
            if (lf_me(lng)) {
                const uint32_t b4 = (q[0] & 0xffu) * 0x01010101u;
                while (len < s.maxlen) {
                    uint32_t pa[4];
                    w.load16(s.prel + len, pa);
                    const uint32_t c0 = first_bit_or_ones(pa[0] ^ b4);
                    const uint32_t c1 = first_bit_or_ones(pa[1] ^ b4) | 32u;
                    const uint32_t c2 = first_bit_or_ones(pa[2] ^ b4) | 64u;
                    const uint32_t c3 = first_bit_or_ones(pa[3] ^ b4) | 96u;
                    uint32_t cb = c0 < c1 ? c0 : c1;
                    const uint32_t ch = c2 < c3 ? c2 : c3;
                    cb = cb < ch ? cb : ch;
                    if (cb < 128u) {
                        len += cb >> 3;
                        break;
                    }
                    len += 16;
                }
            }
        } else if (lf_me(lng)) {  // sixteen more bytes per round (runs of one byte take sixteen rounds to 258)
            while (len < s.maxlen) {
                uint32_t pa[4], ca[4];
                w.load16(s.prel + len, pa);
                w.load16(cpos + len, ca);
                const uint32_t c0 = first_bit_or_ones(pa[0] ^ ca[0]);
                const uint32_t c1 = first_bit_or_ones(pa[1] ^ ca[1]) | 32u;
                const uint32_t c2 = first_bit_or_ones(pa[2] ^ ca[2]) | 64u;
                const uint32_t c3 = first_bit_or_ones(pa[3] ^ ca[3]) | 96u;
                uint32_t cb = c0 < c1 ? c0 : c1;
                const uint32_t ch = c2 < c3 ? c2 : c3;
                cb = cb < ch ? cb : ch;
                if (cb < 128u) {
                    len += cb >> 3;
                    break;
                }
                len += 16;
            }
        }
    }
    len = len < s.maxlen ? len : s.maxlen;
    if (HAS_Q) {  // lz77.rs:351-355: the state after max_hash_checks >> 2 iterations, taken before the first later hit counts
        const lane_flag cap = lf_and_not(hit, s.hq) & lf_of((int32_t)ho < (int32_t)s.hoq);
        s.mq = lf_me(cap) ? m_pack(s.bestd ? s.bm1 + 1 : 0, s.bestd) : s.mq;
        s.hq = s.hq | cap;
    }
    const lane_flag imp = hit & lf_of(len > s.bm1 + 1);  // matching.rs:149-156
    const uint32_t delta = lf_me(imp) ? len - 1 - s.bm1 : 0u;
    s.bestd = lf_me(imp) ? s.prel - cpos : s.bestd;
    s.bm1 += delta;
    s.bb2 += delta << W::SH;
    s.lowa2 += delta << W::SH;
    const lane_flag full = imp & lf_of(len == s.maxlen);
    const uint32_t pr = w.key_at(tbase + ((s.prel + s.bm1) << W::SH));
    s.probe = lf_me(imp) ? pr : s.probe;
    s.offb = lf_me(hit) ? ho - 2 : s.offb;
    const lane_flag more = lf_of((int32_t)s.offb >= (int32_t)s.endb);
    // A settled lane goes on in its segment (a hit that is not a full match, with a candidate left), or its segment is used up --
    // then on to the previous epoch's bucket if the budget reaches it (`has2`), else it is finished --, or its match is full and it
    // is finished.  (Few masks on purpose, and no branch around the move: every scalar instruction of the service is latency of
    // the wave)
    const lane_flag resume = lf_and_not(hit & more, full);
    const lane_flag sw = lf_and_not(lf_and_not(dropped, resume), full) & s.has2;
    if (lf_any(sw)) {  // (measured without the branch: 3.44 against 3.41 ms)
        if (HAS_Q) s.hoq = lf_me(sw) ? s.hoq2 : s.hoq;
        s.offb = lf_me(sw) ? s.offb2 : s.offb;
        s.endb = lf_me(sw) ? s.endb2 : s.endb;
        s.bb2 = lf_me(sw) ? tbase + (s.bm1 << W::SH) : s.bb2;
        s.has2 = lf_and_not(s.has2, sw);
    }
    // (the lanes that go on: those that still walk -- none of them was settled here -- and of the settled ones those with a
    // candidate left in their segment or a second segment to move to.  No "done" mask is kept.)
    s.walk = s.walk | resume | sw;
}

template <bool HAS_Q>
MI355_HD void swg_result(const SwG<HAS_Q>& s, uint32_t* m, uint32_t* mq) {
    const uint32_t r = m_pack(s.bestd ? s.bm1 + 1 : 0, s.bestd);
    *m = r;
    *mq = (HAS_Q && lf_me(s.hq)) ? s.mq : r;
}

// ---- the permuted pair table (k_match3_swz; deflate_kernels.hip PairWinT<true>) --------------------------------------------
// LDS address -> where the permuted table keeps that byte: the index of the 8-byte word inside its 256-byte block (address
// bits 3..7) XOR address bits 8..12.  A bijection of every 256-byte block onto itself that moves whole 8-byte words, so an
// aligned read of up to eight bytes finds its bytes together; lanes whose addresses differ by a multiple of 128, 192, 256 or
// 512 bytes -- rows of records -- land on different banks.  (The table starts at a multiple of 256 and is whole blocks long.
// Bits 11..15 instead -- byte 1 of the address, masked: one SDWA instruction less per read -- spread rows of 64 and 128 bytes
// over eight banks only.)
MI355_HD uint32_t m3_swz(uint32_t a) { return a ^ ((a >> 5) & 0xF8u); }

// ---- rle (rle.rs:13-18, 46-53) ------------------------------------------------------------
// R[p] = number of bytes from p equal to data[p-1], capped at 258 and at the end of input; 0 if
// p == 0 or data[p] != data[p-1].
template <class Bytes>
MI355_HD uint32_t rle_run(const Bytes& by, uint64_t p, uint64_t n) {
    if (p == 0) return 0;
    uint32_t prev = by(p - 1);
    uint64_t cap = n - p < (uint64_t)MAX_MATCH ? n - p : (uint64_t)MAX_MATCH;
    uint32_t c = 0;
    while (c < cap && by(p + c) == prev) c++;
    return c;
}

// ---- segment ends (sync flush points) -------------------------------------------------------------
// ends[0] <= ends[1] <= ... <= ends[m-1] = n: the input as the encoder saw it arrive.  A position p
// belongs to the segment that ends at the first ends[i] > p.  m == 1 is the one-shot case.
struct SegEnds {
    const uint32_t* ends;
    uint32_t m;
};
MI355_HD uint32_t seg_end(const SegEnds& sg, uint64_t p) {
    if (sg.m == 1) return sg.ends[0];
    uint32_t lo = 0, hi = sg.m - 1;  // first i with ends[i] > p (p < ends[m-1] for every real position)
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (sg.ends[mid] > p)
            hi = mid;
        else
            lo = mid + 1;
    }
    return sg.ends[lo];
}

// ---- the parser as a restart transducer ---------------------------------------------------
struct ParseCfg {
    uint32_t mode;           // ParseMode
    uint32_t checks;         // max_hash_checks
    uint32_t lazy_lt;        // lazy_if_less_than (already clamped to 32768, deflate_state.rs:105)
    uint32_t use_quarter;    // 1 if a second table with checks>>2 exists (checks>>2 != checks
                             // can matter only when lazy_lt > 32)
};

struct Step {
    uint32_t nlit;   // literals at j, j+1, ..., j+nlit-1
    uint32_t mlen;   // 0 = no match in this step; else match at j+nlit
    uint32_t mdist;
    uint32_t adv;    // next restart position - j  (1..MAX_JUMP)
    uint32_t fromq;  // the match's entry is the quarter-budget table's (lz77.rs:351-355)
};
// A step as k_adv files it for k_emit, 16 bits: the length of the step (1 .. 255 + 258) in bits 0-9, the literals in front of
// its match (the deferrals) in bits 10-14 -- 31: more than 30, not kept -- and in bit 15 which table the match's entry is in.
// With these the replay of a step is one read (the distance at j + deferrals) instead of the step worked out again.
constexpr uint32_t ADV_LEN_MASK = 1023, ADV_RUN_SHIFT = 10, ADV_RUN_MASK = 31, ADV_RUN_MANY = 31, ADV_FROMQ_SHIFT = 15;
MI355_HD uint32_t adv_pack(const Step& s) {
    const uint32_t run = s.mlen ? (s.nlit < ADV_RUN_MANY ? s.nlit : ADV_RUN_MANY) : 0u;
    return s.adv | (run << ADV_RUN_SHIFT) | (s.fromq << ADV_FROMQ_SHIFT);
}

// lz77.rs:275-278
MI355_HD bool match_too_far(uint32_t len, uint32_t dist) { return len == MIN_MATCH && dist > TOO_FAR; }

// One restart step.  `M(p)` / `Mq(p)` read the match tables (full budget / quarter budget);
// in MODE_RLE `M(p)` reads the run table.  j < n.
//
// Lazy (lz77.rs:305-486): at a position with no usable pending match the parser's future
// depends only on the position.  It emits literal(j) if nothing (>= 3, not too far) is found
// at j; otherwise it defers: as long as the match at a+1 is strictly longer (the prev_length
// filter matching.rs:110,161), a becomes a literal; the chain ends by emitting the pending
// match when it is not beaten, when it is >= lazy_if_less_than (lz77.rs:374-377: no lookahead),
// or when a+1 has no hash byte (lz77.rs:442-468).
// (I: the type of a position -- uint64_t for absolute ones; the kernels use 32-bit positions relative to a tile or
// segment, with `n` relative to the same origin and the tables read through an accessor that knows it)
template <class MT, class I>
MI355_HD Step parse_step(const MT& M, const MT& Mq, I j, I n, const ParseCfg& cfg) {
    Step s;
    s.nlit = 0;
    s.mlen = 0;
    s.mdist = 0;
    s.adv = 1;
    s.fromq = 0;
    if (cfg.mode == MODE_RLE) {  // rle.rs:46-69
        uint32_t r = (uint32_t)M(j);
        if (r >= MIN_MATCH) {
            s.mlen = r;
            s.mdist = 1;
            s.adv = r;
        } else {
            s.nlit = 1;
        }
        return s;
    }
    bool hashable = j + 2 < n;  // lz77.rs:294-301: a position has a hash byte iff p+2 < len
    if (!hashable) {            // lz77.rs:470-482 / :539-544: the last two bytes are literals
        s.nlit = 1;
        return s;
    }
    uint32_t m = (uint32_t)M(j);
    uint32_t L = m_len(m), D = m_dist(m);
    if (L < MIN_MATCH || match_too_far(L, D)) {  // lz77.rs:370-372, :512
        s.nlit = 1;
        return s;
    }
    if (cfg.mode == MODE_GREEDY) {  // lz77.rs:512-534
        s.mlen = L;
        s.mdist = D;
        s.adv = L;
        return s;
    }
    I a = j;
    for (;;) {
        if (L >= cfg.lazy_lt) break;   // ignore_next (lz77.rs:374-377,380-386)
        if (a + 1 + 2 >= n) break;     // a+1 has no hash byte (lz77.rs:442-468)
        const bool quarter = cfg.use_quarter && L >= 32;  // lz77.rs:351-355
        uint32_t m2 = quarter ? (uint32_t)Mq(a + 1) : (uint32_t)M(a + 1);
        uint32_t L2 = m_len(m2);
        if (L2 > L) {                  // strictly better (matching.rs:161); cannot be too_far (L2 >= 4)
            s.fromq = quarter ? 1u : 0u;
            s.nlit++;                  // lz77.rs:430-434: the previous byte becomes a literal
            a++;
            L = L2;
            D = m_dist(m2);
            continue;
        }
        break;                         // lz77.rs:388-429: the pending match wins
    }
    s.mlen = L;
    s.mdist = D;
    s.adv = s.nlit + L;
    return s;
}

// ---- symbol coding (huffman_table.rs:45-194), arithmetic instead of lookup tables ---------
MI355_HD uint32_t ilog2(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x); }

// stored length s = len-3 -> code index 0..28, extra bit count and value (RFC 1951 3.2.5)
MI355_HD void length_symbol(uint32_t s, uint32_t* code, uint32_t* nbits, uint32_t* value) {
    if (s < 8) {
        *code = s;
        *nbits = 0;
        *value = 0;
    } else if (s == 255) {
        *code = 28;
        *nbits = 0;
        *value = 0;
    } else {
        uint32_t k = ilog2(s);  // 3..7
        *code = 4 * (k - 1) + ((s >> (k - 2)) & 3);
        *nbits = k - 2;
        *value = s & ((1u << (k - 2)) - 1);
    }
}
// distance 1..32768 -> code 0..29, extra bit count and value
MI355_HD void distance_symbol(uint32_t dist, uint32_t* code, uint32_t* nbits, uint32_t* value) {
    uint32_t d = dist - 1;
    if (d < 4) {
        *code = d;
        *nbits = 0;
        *value = 0;
    } else {
        uint32_t k = ilog2(d);  // 2..14
        *code = 2 * k + ((d >> (k - 1)) & 1);
        *nbits = k - 1;
        *value = d & ((1u << (k - 1)) - 1);
    }
}
MI355_HD uint32_t length_extra_bits_of_code(uint32_t c) {  // LENGTH_EXTRA_BITS_LENGTH huffman_table.rs:45-47
    return (c < 8 || c == 28) ? 0 : (c - 4) >> 2;
}
MI355_HD uint32_t distance_extra_bits_of_code(uint32_t c) {  // huffman_table.rs:120-126
    uint32_t k = c >> 1;
    return k ? k - 1 : 0;
}
MI355_HD uint32_t fixed_ll_length(uint32_t sym) {  // FIXED_CODE_LENGTHS huffman_table.rs:32-42
    return sym < 144 ? 8 : sym < 256 ? 9 : sym < 280 ? 7 : 8;
}

// bit_reverse.rs:3-10
MI355_HD uint32_t reverse_bits16(uint32_t n, uint32_t length) {
    n = ((n & 0xaaaa) >> 1) | ((n & 0x5555) << 1);
    n = ((n & 0xcccc) >> 2) | ((n & 0x3333) << 2);
    n = ((n & 0xf0f0) >> 4) | ((n & 0x0f0f) << 4);
    n = ((n & 0xff00) >> 8) | ((n & 0x00ff) << 8);
    return (n & 0xffff) >> (16 - length);
}

// ---- Huffman code lengths (length_encode.rs:218-415) --------------------------------------
// `nodes` holds the used symbols sorted ascending by (freq, symbol) -- identical to the
// reference's stable sort by freq (length_encode.rs:386).  value[] is overwritten.
struct HuffNode {
    uint32_t value;
    uint32_t symbol;
};

// enforce_max_code_lengths length_encode.rs:290-327 on the depth histogram num_codes[0..32] (depths of 32
// and more counted under 32).
template <class NumArr>
MI355_HD void limit_code_lengths(NumArr& num_codes, uint32_t max_len) {
    uint32_t above = 0;
    for (uint32_t i = max_len + 1; i < 33; i++) above += num_codes[i];
    num_codes[max_len] += above;
    uint32_t total = 0;
    for (uint32_t i = max_len; i >= 1; i--) total += num_codes[i] << (max_len - i);
    while (total != (1u << max_len)) {
        num_codes[max_len]--;
        for (uint32_t i = max_len - 1; i >= 1; i--) {
            if (num_codes[i] != 0) {
                num_codes[i]--;
                num_codes[i + 1] += 2;
                break;
            }
        }
        total--;
    }
}

// In-place Moffat-Katajainen, the miniz length limiter and the reversed hand-out
// (length_encode.rs:218-278, 290-327, 392-408).  lengths[] must be zeroed by the caller for
// all symbols; n >= 2.
template <class NodeArr, class LenArr>
MI355_HD void huff_lengths_sorted(NodeArr& leaves, uint32_t n, uint32_t max_len, LenArr& lengths) {
    // step_1 :218-247
    {
        uint32_t root = 0, leaf = 2;
        leaves[0].value += leaves[1].value;
        for (uint32_t next = 1; next + 1 < n; next++) {
            if (leaf >= n || leaves[root].value < leaves[leaf].value) {
                leaves[next].value = leaves[root].value;
                leaves[root].value = next;
                root++;
            } else {
                leaves[next].value = leaves[leaf].value;
                leaf++;
            }
            if (leaf >= n || (root < next && leaves[root].value < leaves[leaf].value)) {
                leaves[next].value += leaves[root].value;
                leaves[root].value = next;
                root++;
            } else {
                leaves[next].value += leaves[leaf].value;
                leaf++;
            }
        }
    }
    // step_2 :249-278
    {
        leaves[n - 2].value = 0;
        for (uint32_t t = n - 2; t-- > 0;) leaves[t].value = leaves[leaves[t].value].value + 1;
        uint32_t available = 1, used = 0, depth = 0;
        int32_t root = (int32_t)n - 2, next = (int32_t)n - 1;
        while (available > 0) {
            while (root >= 0 && leaves[root].value == depth) {
                used++;
                root--;
            }
            while (available > used) {
                leaves[next].value = depth;
                next--;
                available--;
            }
            available = 2 * used;
            depth++;
            used = 0;
        }
    }
    // depth histogram :392-395 and enforce_max_code_lengths :290-327
    uint32_t num_codes[33];
    for (int i = 0; i < 33; i++) num_codes[i] = 0;
    for (uint32_t i = 0; i < n; i++) num_codes[leaves[i].value < 32 ? leaves[i].value : 32]++;
    limit_code_lengths(num_codes, max_len);
    // hand out lengths shortest first, walking the sorted leaves from the end :402-408
    uint32_t li = n;
    for (uint32_t i = 1; i <= max_len; i++)
        for (uint32_t k = 0; k < num_codes[i]; k++) {
            li--;
            lengths[leaves[li].symbol] = (uint8_t)i;
        }
}

// ---- run-length coding of the code lengths (length_encode.rs:82-155) ----------------------
// Output symbol = kind << 8 | value; kind 0 literal length, 1 copy-previous (sym 16),
// 2 zeros 3..10 (sym 17), 3 zeros 11..138 (sym 18).  freqs[19] must be zeroed by the caller.
MI355_HD uint32_t el_symbol_index(uint32_t e) {
    uint32_t k = e >> 8;
    return k == 0 ? (e & 0xff) : 15 + k;
}
MI355_HD bool not_max_repetitions(uint32_t l, uint32_t repeats) { return (l == 0 && repeats < 138) || repeats < 6; }

// The values pushed by the short-run arm (:135-152 re-reads lengths[skip..]) are known without a read: the
// run's elements equal `prev`, the one at the current index is `l`.  COUNT = false leaves the symbol
// frequencies to the caller (the block-header kernel counts them from `out` with the whole wave); the next
// length is read one round ahead so that a single lane does not wait for it.
template <bool COUNT, class LenArr, class OutArr, class FreqArr>
MI355_HD uint32_t encode_lengths_rle_impl(const LenArr& lengths, uint32_t n_len, OutArr& out, FreqArr& freqs) {
    uint32_t n_out = 0;
    uint32_t repeat = 0;
    uint32_t nextl = lengths[0];
    uint32_t prev = (~nextl) & 0xff;
    uint32_t idx = 0;
#define MI355_EL_PUSH(e)                           \
    do {                                           \
        uint32_t e__ = (e);                        \
        if (COUNT) freqs[el_symbol_index(e__)]++;  \
        out[n_out++] = (uint16_t)e__;              \
    } while (0)
    while (idx < n_len) {
        uint32_t n = idx;
        uint32_t l = nextl;
        idx++;
        bool peek_none = idx >= n_len;
        if (!peek_none) nextl = lengths[idx];
        if (l == prev && not_max_repetitions(l, repeat)) repeat++;
        if (l != prev || peek_none || !not_max_repetitions(l, repeat)) {
            if (repeat >= 3) {
                uint32_t kind = prev == 0 ? (repeat <= 10 ? 2u : 3u) : 1u;  // from_prev_and_repeat :19-31
                MI355_EL_PUSH((kind << 8) | repeat);
                repeat = 0;
                if (l != prev) {
                    if (l != 0 || peek_none) {
                        MI355_EL_PUSH(l);
                        repeat = 0;
                    } else {
                        repeat = 1;
                    }
                }
            } else {
                uint32_t extra_skip = (peek_none && l == prev) ? 1 : 0;
                uint32_t skip = n + extra_skip - repeat;
                uint32_t extra = (l != 0 || peek_none) ? 1 : 0;
                uint32_t take = repeat + extra;
                for (uint32_t k = 0; k < take && skip + k < n_len; k++) MI355_EL_PUSH(skip + k < n ? prev : l);
                repeat = 1 - extra;
            }
        }
        prev = l;
    }
#undef MI355_EL_PUSH
    return n_out;
}
template <class LenArr, class OutArr, class FreqArr>
MI355_HD uint32_t encode_lengths_rle(const LenArr& lengths, uint32_t n_len, OutArr& out, FreqArr& freqs) {
    return encode_lengths_rle_impl<true>(lengths, n_len, out, freqs);
}

// The same coding run by run -- what the state machine above comes to, and what the block-header kernel does
// with one lane per run.  A run of c equal lengths v:
//   v != 0: the first one as itself, the other c - 1 as "copy previous" symbols of 6 (the counter is flushed
//           when it reaches 6), then the rest r < 6 as one copy symbol if r >= 3, else as r literals;
//   v == 0: every zero counts (the first one too): zero runs of 138, then the rest r < 138 as symbol 18
//           (r >= 11), symbol 17 (3..10) or r literal zeros.
// tests/test_stages_vs_oracle.py holds the two forms against each other.
MI355_HD uint32_t el_run_count(uint32_t v, uint32_t c) {
    if (v) {
        const uint32_t R = c - 1, r = R % 6;
        return 1 + R / 6 + (r >= 3 ? 1u : r);
    }
    const uint32_t r = c % 138;
    return c / 138 + (r >= 3 ? 1u : r);
}
template <class OutArr>
MI355_HD void el_run_emit(uint32_t v, uint32_t c, OutArr& out, uint32_t at) {
    if (v) {
        const uint32_t R = c - 1, r = R % 6;
        out[at++] = (uint16_t)v;
        for (uint32_t k = 0; k < R / 6; k++) out[at++] = (uint16_t)((1u << 8) | 6u);
        if (r >= 3)
            out[at++] = (uint16_t)((1u << 8) | r);
        else
            for (uint32_t k = 0; k < r; k++) out[at++] = (uint16_t)v;
    } else {
        const uint32_t r = c % 138;
        for (uint32_t k = 0; k < c / 138; k++) out[at++] = (uint16_t)((3u << 8) | 138u);
        if (r >= 3)
            out[at++] = (uint16_t)(((r <= 10 ? 2u : 3u) << 8) | r);
        else
            for (uint32_t k = 0; k < r; k++) out[at++] = (uint16_t)0;
    }
}
template <class LenArr, class OutArr>
MI355_HD uint32_t encode_lengths_runs(const LenArr& lengths, uint32_t n_len, OutArr& out) {
    uint32_t n_out = 0;
    for (uint32_t i = 0; i < n_len;) {
        uint32_t e = i + 1;
        while (e < n_len && lengths[e] == lengths[i]) e++;
        el_run_emit((uint32_t)lengths[i], e - i, out, n_out);
        n_out += el_run_count((uint32_t)lengths[i], e - i);
        i = e;
    }
    return n_out;
}

// ---- canonical codes (huffman_table.rs:232-278) -------------------------------------------
// codes[i] for lengths[i] != 0, bit-reversed for LSB-first emission.
template <class LenArr, class CodeArr>
MI355_HD void canonical_codes(const LenArr& lengths, uint32_t n, CodeArr& codes) {
    uint32_t counts[16];
    for (int i = 0; i < 16; i++) counts[i] = 0;
    uint32_t max_length = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t l = lengths[i];
        if (l) counts[l]++;
        if (l > max_length) max_length = l;
    }
    uint32_t next_code[17];
    uint32_t code = 0;
    next_code[0] = 0;
    for (uint32_t bits = 1; bits <= max_length; bits++) {
        code = ((code + counts[bits - 1]) << 1) & 0xffff;
        next_code[bits] = code;
    }
    for (uint32_t i = 0; i < n; i++) {
        uint32_t l = lengths[i];
        if (l) {
            codes[i] = (uint16_t)reverse_bits16(next_code[l], l);
            next_code[l] = (next_code[l] + 1) & 0xffff;
        }
    }
}

// ---- per-block header (what one wave computes for one block) ------------------------------
// HUFFMAN_LENGTH_ORDER huffman_lengths.rs:27-29
MI355_HD uint32_t hclen_order(uint32_t i) {
    const uint8_t o[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    return o[i];
}

struct BlockHeader {
    uint8_t ll_len[288];
    uint8_t d_len[32];
    uint8_t cl_len[19];
    uint8_t pad0;
    uint16_t enc[320];      // run-length coded ll ++ dist lengths
    uint32_t n_enc;
    uint32_t n_ll;          // HLIT + 257
    uint32_t n_d;           // HDIST + 1
    uint32_t used_hclens;   // HCLEN + 4
    uint64_t dyn_bits;      // what a dynamic block really takes, without the 3 block bits
    uint64_t dyn_est;       // the reference's estimate (huffman_lengths.rs:257-263); it prices
                            // symbol 16 at 3 extra bits (:50-56) although 2 are written (:343)
    uint64_t static_est;    // the reference's estimate, Q12 bias included (:244-266)
    uint64_t fixed_bits;    // what a fixed block really takes, without the 3 block bits
};

// stored_padding huffman_lengths.rs:113-124 and stored_length :132-143
MI355_HD uint64_t stored_padding(uint32_t pending_bits) {
    uint32_t free_space = 8 - pending_bits;
    return free_space >= 3 ? free_space - 3 : 8 - (3 - free_space);
}
MI355_HD uint64_t stored_length_bits(uint64_t input_bytes) {
    uint64_t num_blocks = (input_bytes - 1) / MAX_STORED_BLOCK_LENGTH + 1;
    return (input_bytes + 4 * num_blocks + (num_blocks - 1)) * 8;
}

// Sort helper for the host / single-lane path: ascending by (value, symbol).
template <class NodeArr>
MI355_HD void sort_nodes(NodeArr& a, uint32_t n) {
    for (uint32_t i = 1; i < n; i++) {
        HuffNode x = a[i];
        uint32_t j = i;
        while (j > 0 && (a[j - 1].value > x.value || (a[j - 1].value == x.value && a[j - 1].symbol > x.symbol))) {
            a[j] = a[j - 1];
            j--;
        }
        a[j] = x;
    }
}

// in_place_lengths length_encode.rs:347-415, split into the pieces the block-header kernel
// runs: gather the used symbols, sort them by (freq, symbol), then lengths_from_sorted.
template <class FreqArr, class NodeArr>
MI355_HD uint32_t gather_nodes(const FreqArr& freqs, uint32_t n, NodeArr& nodes) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; i++)
        if (freqs[i] > 0) {
            nodes[m].value = freqs[i];
            nodes[m].symbol = i;
            m++;
        }
    return m;
}
// lengths[] already zeroed for the whole output slice (:355-357); special cases :377-382
template <class NodeArr, class LenArr>
MI355_HD void lengths_from_sorted(NodeArr& nodes, uint32_t m, uint32_t max_len, LenArr& lengths) {
    if (m == 0) return;
    if (m == 1) {
        lengths[nodes[0].symbol] = 1;
        return;
    }
    huff_lengths_sorted(nodes, m, max_len, lengths);
}
template <class FreqArr, class LenArr, class NodeArr>
MI355_HD void huff_lengths(const FreqArr& freqs, uint32_t n, uint32_t n_total, uint32_t max_len, LenArr& lengths,
                           NodeArr& scratch) {
    for (uint32_t i = 0; i < n_total; i++) lengths[i] = 0;
    uint32_t m = gather_nodes(freqs, n, scratch);
    if (m >= 2) sort_nodes(scratch, m);
    lengths_from_sorted(scratch, m, max_len, lengths);
}

// remove_trailing_zeroes huffman_lengths.rs:44-47
template <class FreqArr>
MI355_HD uint32_t trimmed_count(const FreqArr& f, uint32_t n, uint32_t min_len) {
    while (n > min_len && f[n - 1] == 0) n--;
    return n;
}
// used_hclens huffman_lengths.rs:230-235
template <class LenArr>
MI355_HD uint32_t count_used_hclens(const LenArr& cl_len) {
    uint32_t used = 19;
    while (used > 0 && cl_len[hclen_order(used - 1)] == 0) used--;
    return used;
}
// The bit costs of huffman_lengths.rs:241-266 plus the real sizes (see BlockHeader).
template <class FreqArr, class CFreqArr, class LLLen, class DLen, class CLLen>
MI355_HD void block_costs(const FreqArr& ll_freq, const FreqArr& d_freq, const CFreqArr& cl_freq, const LLLen& ll_len,
                          const DLen& d_len, const CLLen& cl_len, uint32_t n_ll, uint32_t n_d, uint32_t used,
                          uint64_t* dyn_bits, uint64_t* dyn_est, uint64_t* static_est, uint64_t* fixed_bits) {
    uint64_t d_ll = 0, s_ll = 0;
    for (uint32_t c = 0; c < n_ll; c++) {
        uint64_t f = ll_freq[c];
        uint64_t extra = c >= 257 ? length_extra_bits_of_code(c - 257) : 0;
        d_ll += f * (ll_len[c] + extra);
        s_ll += f * (fixed_ll_length(c) + extra);
    }
    uint64_t d_d = 0, s_d = 0, f_d = 0;
    for (uint32_t c = 0; c < n_d; c++) {
        uint64_t f = d_freq[c];
        uint64_t extra = distance_extra_bits_of_code(c);
        d_d += f * (d_len[c] + extra);
        s_d += f * (fixed_ll_length(c) + extra);  // Q12: the ll table is used for distances too
        f_d += f * (5 + extra);
    }
    uint64_t table = 0, table_real = 0;  // calculate_huffman_length :59-68
    for (uint32_t i = 0; i < 19; i++) {
        uint64_t extra = (i == 16 || i == 17) ? 3 : (i == 18 ? 7 : 0);
        uint64_t extra_real = i == 16 ? 2 : extra;  // write_huffman_lengths :343 writes 2 bits
        table += (uint64_t)cl_freq[i] * (cl_len[i] + extra);
        table_real += (uint64_t)cl_freq[i] * (cl_len[i] + extra_real);
    }
    *dyn_est = d_ll + d_d + table + (uint64_t)used * 3 + 5 + 5 + 4;
    *dyn_bits = d_ll + d_d + table_real + (uint64_t)used * 3 + 5 + 5 + 4;
    *static_est = s_ll + s_d;
    *fixed_bits = s_ll + f_d;
}

// gen_huffman_lengths huffman_lengths.rs:167-287 minus the block-type choice (which needs the
// bit phase and is made by plan_block).  ll_freq[286] (EOB already counted), d_freq[30].
// Host composite of the pieces above; the kernel k_block_header runs the same pieces.
template <class FreqArr, class NodeArr>
MI355_HD void build_block_header(const FreqArr& ll_freq, const FreqArr& d_freq, BlockHeader& h, NodeArr& scratch) {
    uint32_t n_ll = trimmed_count(ll_freq, NUM_LL, 257);
    uint32_t n_d = trimmed_count(d_freq, NUM_DIST, 1);
    huff_lengths(ll_freq, n_ll, 288, 15, h.ll_len, scratch);
    huff_lengths(d_freq, n_d, 32, 15, h.d_len, scratch);
    h.n_ll = n_ll;
    h.n_d = n_d;
    uint8_t chain[320];  // chained lengths :212-218
    for (uint32_t i = 0; i < n_ll; i++) chain[i] = h.ll_len[i];
    for (uint32_t i = 0; i < n_d; i++) chain[n_ll + i] = h.d_len[i];
    uint32_t cl_freq[19];
    for (int i = 0; i < 19; i++) cl_freq[i] = 0;
    h.n_enc = encode_lengths_rle(chain, n_ll + n_d, h.enc, cl_freq);
    huff_lengths(cl_freq, 19, 19, 7, h.cl_len, scratch);
    h.used_hclens = count_used_hclens(h.cl_len);
    block_costs(ll_freq, d_freq, cl_freq, h.ll_len, h.d_len, h.cl_len, n_ll, n_d, h.used_hclens, &h.dyn_bits,
                &h.dyn_est, &h.static_est, &h.fixed_bits);
}

// ---- block plan (compress.rs:157-246, huffman_lengths.rs:179,269-286) ---------------------
struct BlockPlan {
    uint32_t btype;
    uint32_t bfinal;
    uint64_t bit_start;   // first header bit
    uint64_t bit_len;     // total bits of the block including the 3 header bits and padding
};

// Bits a stored block sequence occupies when it starts at bit phase `phase` (0..7):
// stored_block.rs:13-40, compress.rs:59-77.  Every piece: 3 header bits, pad to a byte, LEN,
// NLEN, payload.
MI355_HD uint64_t stored_total_bits(uint64_t nbytes, uint32_t phase) {
    uint64_t bits = 0;
    uint64_t left = nbytes;
    uint32_t ph = phase;
    do {
        uint64_t piece = left < (uint64_t)MAX_STORED_BLOCK_LENGTH ? left : (uint64_t)MAX_STORED_BLOCK_LENGTH;
        uint64_t hdr = 3 + ((8 - ((ph + 3) & 7)) & 7);
        bits += hdr + 32 + piece * 8;
        ph = 0;
        left -= piece;
    } while (left > 0);
    return bits;
}

// Decide one block from its four cost figures (BlockHeader), the bytes its tokens cover and the
// bit position it starts at.
MI355_HD void plan_block(uint64_t dyn_bits, uint64_t dyn_est, uint64_t static_est, uint64_t fixed_bits,
                         uint64_t in_bytes, bool is_last, uint64_t bit_pos, BlockPlan* out) {
    uint32_t btype;
    if (in_bytes <= 4) {  // huffman_lengths.rs:179-181
        btype = BT_FIXED;
    } else {
        uint64_t stored = stored_length_bits(in_bytes) + stored_padding((uint32_t)(bit_pos & 7));  // :269
        uint64_t used = dyn_est < static_est ? dyn_est : static_est;
        if (stored < used) used = stored;
        if (used == static_est)
            btype = BT_FIXED;  // :277-286 (Q5)
        else if (used == stored)
            btype = BT_STORED;
        else
            btype = BT_DYNAMIC;
    }
    out->btype = btype;
    out->bfinal = is_last ? 1 : 0;
    out->bit_start = bit_pos;
    if (btype == BT_FIXED)
        out->bit_len = 3 + fixed_bits;
    else if (btype == BT_DYNAMIC)
        out->bit_len = 3 + dyn_bits;
    else
        out->bit_len = stored_total_bits(in_bytes, (uint32_t)(bit_pos & 7));
}

// Bits of one token under a code table: returns the concatenated LSB-first bit string (<= 48
// bits) and its length (encoder_state.rs:58-83).
template <class LLCodes, class LLLens, class DCodes, class DLens>
MI355_HD uint64_t token_bits(uint32_t tok, const LLCodes& llc, const LLLens& lll, const DCodes& dc, const DLens& dl,
                             uint32_t* nbits) {
    uint32_t dist = tok >> 16;
    if (dist == 0) {
        uint32_t b = tok & 0xff;
        *nbits = lll[b];
        return llc[b];
    }
    uint32_t code, eb, ev;
    length_symbol(tok & 0xff, &code, &eb, &ev);
    uint32_t sym = 257 + code;
    uint64_t bits = llc[sym];
    uint32_t n = lll[sym];
    bits |= (uint64_t)ev << n;
    n += eb;
    uint32_t dcode, deb, dev;
    distance_symbol(dist, &dcode, &deb, &dev);
    bits |= (uint64_t)dc[dcode] << n;
    n += dl[dcode];
    bits |= (uint64_t)dev << n;
    n += deb;
    *nbits = n;
    return bits;
}

}  // namespace mi355
#endif
