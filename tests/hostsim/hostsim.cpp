// hostsim.cpp -- TEST INFRASTRUCTURE.  Runs the stage functions of
// deflate-rs_amd/csrc/stages.h sequentially on the host, stage by stage and array by array
// exactly as the HIP kernels are organised, so that the re-cut of the reference's serial loop
// (pure match table -> restart path -> 31744-token blocks -> per-block Huffman -> bit plan)
// can be diffed against the CPU oracle without a GPU.  It is never linked into the product
// library.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../deflate-rs_amd/csrc/stages.h"

using namespace mi355;

namespace {

struct HostBytes {
    const uint8_t* d;
    uint32_t operator()(uint64_t i) const { return d[i]; }
};

struct HostWin {  // absolute index space
    const uint8_t* d;
    const uint16_t* lk;
    uint32_t load32(uint32_t i) const {
        uint32_t v;
        memcpy(&v, d + i, 4);
        return v;
    }
    uint32_t link(uint32_t i) const { return lk[i]; }
    uint32_t link_far(uint32_t i) const { return lk[i] ? lk[i] : 0xFFFFu; }
};

struct MAcc {
    const uint32_t* m;
    uint32_t operator()(uint64_t i) const { return m[i]; }
};

int g_force_ident = 0;  // walk the identity hops of the head table even without a re-warm
int g_multi = 0;  // use match_walk_multi (the k_match formulation) instead of match_walk

struct Sim {
    std::vector<uint8_t> in;  // padded with 16 zero bytes
    uint64_t n;
    ParseCfg cfg;
    std::vector<uint16_t> link;
    std::vector<uint32_t> M, Mq;
    std::vector<uint16_t> adv;
    std::vector<uint32_t> tokens;
    std::vector<uint64_t> tok_pos;
    int hier_mismatch = 0;
    HashOverride ov = {0, 0, 0, 0, 0, nullptr, 0, nullptr, 0, nullptr};  // what stage_links was given (mode 4 sorts by the same hashes)
};

void stage_links(Sim& s, const HashOverride& ov) {
    s.ov = ov;
    s.link.assign(s.n + 1, 0);
    std::vector<int64_t> head(32768, -1);
    // identity-initialised head table (chained_hash_table.rs:64-69): matters only after a re-warm
    if (ov.on | ov.m | (uint32_t)g_force_ident)
        for (int64_t h = 0; h < 32768; h++) head[h] = h;
    HostBytes by{s.in.data()};
    for (uint64_t p = 0; p + 2 < s.n; p++) {
        uint32_t h = position_hash(by, p, ov);
        int64_t q = head[h];
        s.link[p] = (q >= 0 && (uint64_t)q < p && p - (uint64_t)q <= WINDOW_SIZE) ? (uint16_t)(p - (uint64_t)q) : 0;
        head[h] = (int64_t)p;
    }
}

void stage_match(Sim& s) {
    s.M.assign(s.n + 1, 0);
    s.Mq.assign(s.n + 1, 0);
    if (s.cfg.mode == MODE_RLE) {
        HostBytes by{s.in.data()};
        for (uint64_t p = 0; p < s.n; p++) s.M[p] = rle_run(by, p, s.n);
        return;
    }
    if (s.cfg.checks == 0) return;
    HostWin w{s.in.data(), s.link.data()};
    uint32_t cq = s.cfg.use_quarter ? (s.cfg.checks >> 2) : 0;
    if (g_multi == 6 || g_multi == 7 || g_multi == 8) {
        // (8: a small call's walk -- the near half of a position's candidates and the far half walked apart, m3_epoch<.., SINGLE>)
        // the k_sort + k_match3 formulation: epochs sorted by (hash, position), lanes walk their bucket's
        // entries (stages.h SwG); groups of steps and services alternate as on the GPU (7: the RUN1 service)
        const uint32_t W = WINDOW_SIZE;
        HostBytes by{s.in.data()};
        bool hasq = s.cfg.use_quarter && cq != 0;
        std::vector<uint16_t> prevS, curS;
        std::vector<uint32_t> prevB(32769, 0), curB(32769, 0);
        for (uint64_t E = 0; E < s.n; E += W) {
            prevS.swap(curS);
            prevB.swap(curB);
            curS.clear();
            std::vector<uint32_t> hs;
            uint64_t hi = std::min<uint64_t>(s.n, E + W);
            std::vector<uint32_t> order;
            for (uint64_t p = E; p < hi; p++)
                if (p + 2 < s.n) order.push_back((uint32_t)(p - E));
            std::vector<uint32_t> hh(W, 0);
            for (uint32_t r : order) hh[r] = position_hash(by, E + r, s.ov);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hh[a] < hh[b]; });
            std::fill(curB.begin(), curB.end(), 0);
            for (uint32_t r : order) curB[hh[r] + 1]++;
            for (uint32_t h = 0; h < 32768; h++) curB[h + 1] += curB[h];
            curS.assign(order.begin(), order.end());
            const uint64_t wbase = E >= W ? E - W : 0;
            const uint32_t bias = (uint32_t)(E - wbase);
            for (uint32_t j = 0; j < curS.size(); j++) {
                const uint32_t r = curS[j], h = hh[r];
                const uint32_t prel = bias + r, nrel = (uint32_t)(s.n - wbase);
                uint32_t pb0 = 0, pb1 = 0;
                if (E) {
                    pb0 = prevB[h];
                    pb1 = prevB[h + 1];
                }
                uint32_t m = 0, mq = 0;
                // the k_match3 form: pair table, groups of four probes, the service decodes where a lane stopped
                struct PWin {  // pair table: the key is the pair, entries and addresses count two bytes per position
                    enum : uint32_t { SH = 1 };
                    const uint8_t* d;  // position 0 of the window
                    uint64_t nb;       // readable bytes from d
                    const uint16_t* ps;
                    const uint16_t* cs;
                    uint32_t np, nc;
                    uint32_t byte(uint64_t k) const { return k < nb ? d[k] : 0u; }
                    uint32_t key_at(uint32_t a) const { return byte(a >> 1) | (byte((a >> 1) + 1) << 8); }  // (tbase = 0)
                    uint32_t load32(uint32_t k) const { return byte(k) | (byte(k + 1) << 8) | (byte(k + 2) << 16) | (byte(k + 3) << 24); }
                    void load16(uint32_t k, uint32_t* q) const {
                        for (int i = 0; i < 4; i++) q[i] = load32(k + 4 * i);
                    }
                    uint32_t sidx(uint32_t i) const {  // entries are 2 * position (k_sort with dbl = 1)
                        if (i >= SW_OWN) return i - SW_OWN < nc ? 2u * cs[i - SW_OWN] : 0u;
                        return i < np ? 2u * ps[i] : 0u;
                    }
                };
                // (skip / budget / budget_q: as k_match3's set_up -- the candidates from rank skip + 1 on are those of the entry
                // `skip` places down its bucket, and what the own epoch's part is short of comes off the previous epoch's)
                uint32_t skip = 0, budget = s.cfg.checks, budget_q = cq;
                auto run6 = [&](auto& ln, auto pw) {
                    pw.d = s.in.data() + wbase;
                    pw.nb = (uint64_t)s.in.size() - wbase;
                    pw.ps = prevS.data();
                    pw.cs = curS.data();
                    pw.np = (uint32_t)prevS.size();
                    pw.nc = (uint32_t)curS.size();
                    uint32_t je = j, pe1 = pb1;
                    if (skip) {
                        const uint32_t n_own = je - curB[h];
                        if (n_own >= skip) {
                            je -= skip;
                        } else {
                            const uint32_t rest = skip - n_own, n_prev = pb1 - pb0;
                            je = curB[h];
                            pe1 = pb1 - (rest < n_prev ? rest : n_prev);
                        }
                    }
                    (void)swg_setup(ln, pw, je, curB[h], pb0, pe1, prel, nrel, 0u, bias, budget, budget_q);
                    const uint32_t width = (j & 4) ? 8u : 4u;  // (groups of four and of eight steps)
                    lane_flag dropped = (j % 3) ? swg_first(ln, pw, width) : lf_of(false);  // (with and without the short cut)
                    int d = dropped ? 0 : -1;
                    uint32_t guard = 0;
                    for (;;) {
                        if (lf_me(dropped)) {
                            const uint32_t av[8] = {ln.a0, ln.a1, ln.a2, ln.a3, ln.a4, ln.a5, ln.a6, ln.a7};
                            if (g_multi == 7)
                                swg_service<true>(ln, pw, 0u, cq, dropped, lf_of(d >= 0), d >= 0 ? av[d] : 0u, ln.offb + 2 * (width - (d >= 0 ? d : 0)));
                            else
                                swg_service<false>(ln, pw, 0u, cq, dropped, lf_of(d >= 0), d >= 0 ? av[d] : 0u, ln.offb + 2 * (width - (d >= 0 ? d : 0)));
                        }
                        if (!lf_me(ln.walk)) break;
                        const uint32_t groups = 1 + (guard % 3);
                        d = -1;
                        for (uint32_t g = 0; g < groups && lf_me(ln.walk); g++) swg_group_ref(ln, pw, &d, width);
                        dropped = lf_not(ln.walk);
                        if (++guard > 100000) break;
                    }
                    swg_result(ln, &m, &mq);
                };
                auto run_one = [&]() {
                    if (hasq) {
                        SwG<true> ln;
                        run6(ln, PWin());
                    } else {
                        SwG<false> ln;
                        run6(ln, PWin());
                    }
                };
                const uint32_t half = (g_multi == 8 && s.cfg.checks >= 16) ? s.cfg.checks / 2 : 0;
                if (half) {
                    // the near half (it also has the quarter-budget result), then the far half: only a strictly longer match counts
                    budget = half;
                    run_one();
                    const uint32_t mx = m, mqx = mq;
                    skip = half;
                    budget = s.cfg.checks - half;
                    budget_q = 0;
                    run_one();
                    m = m_len(m) > m_len(mx) ? m : mx;
                    mq = mqx;
                } else {
                    run_one();
                }
                s.M[E + r] = m;
                s.Mq[E + r] = mq;
            }
        }
        if (s.cfg.use_quarter && cq == 0)
            for (uint64_t p = 0; p < s.n; p++) s.Mq[p] = 0;
        return;
    }
    if (g_multi >= 2) {
        // the k_match formulation: parked extension, positions handed out by a shared counter
        struct Emit {
            Sim* s;
            void operator()(uint32_t idx, uint32_t m, uint32_t mq) const {
                s->M[idx] = m;
                s->Mq[idx] = mq;
            }
        } emit{&s};
        bool hasq = s.cfg.use_quarter && cq != 0;
        const uint32_t T = 16384;
        for (uint64_t E = 0; E < s.n; E += T) {
            // one "lane" takes every position of the tile in turn (the hand-out order is free)
            struct Next {
                uint32_t cur, end;
                uint32_t operator()() { return cur < end ? cur++ : (uint32_t)NO_POS; }
            } next{(uint32_t)E, (uint32_t)(E + T)};
            if (g_multi == 3) {
                // service only every 3rd step and cut every compare after one round: exercises the
                // "stay parked, go on next time" path the GPU policy takes for long matches
                struct Cut {
                    bool operator()(bool pending, bool walking, uint32_t iter) const { return pending && (iter % 3 == 2 || !walking); }
                    bool keep_extending(bool any, uint32_t round) const { return any && round < 1; }
                } pol;
                if (hasq)
                    match_walk_park<4, true>(w, next, ConstLimit{(uint32_t)s.n}, s.cfg.checks, cq, emit, pol);
                else
                    match_walk_park<4, false>(w, next, ConstLimit{(uint32_t)s.n}, s.cfg.checks, 0, emit, pol);
                continue;
            }
            ServiceAlways pol;
            if (hasq)
                match_walk_park<4, true>(w, next, ConstLimit{(uint32_t)s.n}, s.cfg.checks, cq, emit, pol);
            else
                match_walk_park<4, false>(w, next, ConstLimit{(uint32_t)s.n}, s.cfg.checks, 0, emit, pol);
        }
        if (s.cfg.use_quarter && cq == 0)
            for (uint64_t p = 0; p < s.n; p++) s.Mq[p] = 0;
        return;
    }
    if (g_multi) {
        // the k_match geometry: tiles of 16384 positions, 1024 lanes x 16 positions, 4 chains in flight
        struct Emit {
            Sim* s;
            void operator()(uint32_t idx, uint32_t m, uint32_t mq) const {
                s->M[idx] = m;
                s->Mq[idx] = mq;
            }
        } emit{&s};
        const uint32_t T = 16384, TH = 1024;
        bool hasq = s.cfg.use_quarter && cq != 0;
        for (uint64_t E = 0; E < s.n; E += T)
            for (uint32_t tid = 0; tid < TH; tid++) {
                if (hasq)
                    match_walk_multi<4, true>(w, (uint32_t)(E + tid), TH, T / TH, (uint32_t)s.n, s.cfg.checks, cq, emit);
                else
                    match_walk_multi<4, false>(w, (uint32_t)(E + tid), TH, T / TH, (uint32_t)s.n, s.cfg.checks, 0, emit);
            }
        if (s.cfg.use_quarter && cq == 0)
            for (uint64_t p = 0; p < s.n; p++) s.Mq[p] = 0;
        return;
    }
    for (uint64_t p = 0; p + 2 < s.n; p++) {
        uint32_t max_len = (uint32_t)std::min<uint64_t>(s.n - p, MAX_MATCH);
        uint32_t m, mq;
        if (s.cfg.use_quarter && cq == 0) {
            // budget >> 2 == 0: zero iterations -> nothing found
            match_walk(w, (uint32_t)p, max_len, s.cfg.checks, 0, &m, &mq);
            mq = 0;
        } else {
            match_walk(w, (uint32_t)p, max_len, s.cfg.checks, cq, &m, &mq);
        }
        s.M[p] = m;
        s.Mq[p] = mq;
    }
}

// direct serial walk of the restart path
void stage_parse_direct(Sim& s) {
    s.tokens.clear();
    s.tok_pos.clear();
    MAcc M{s.M.data()}, Mq{s.Mq.data()};
    uint64_t j = 0;
    while (j < s.n) {
        Step st = parse_step(M, Mq, j, s.n, s.cfg);
        for (uint32_t k = 0; k < st.nlit; k++) {
            s.tokens.push_back(tok_literal(s.in[j + k]));
            s.tok_pos.push_back(j + k);
        }
        if (st.mlen) {
            s.tokens.push_back(tok_match(st.mlen, st.mdist));
            s.tok_pos.push_back(j + st.nlit);
        }
        j += st.adv;
    }
}

// The same path found the way the GPU finds it: adv[j] for every j, per-segment exit tables by
// a right-to-left sweep, a tree of composed tables, then top-down entry resolution.
void stage_parse_hier(Sim& s, uint32_t S, uint32_t G) {
    MAcc M{s.M.data()}, Mq{s.Mq.data()};
    uint64_t n = s.n;
    s.adv.assign(n, 0);
    for (uint64_t j = 0; j < n; j++) {
        const Step st = parse_step(M, Mq, j, n, s.cfg);
        s.adv[j] = (uint16_t)st.adv;
        // What the kernels make of a step instead of running it (reported like a path mismatch):
        // (a) k_emit's replay reads a step off the entry k_adv files for it (adv_pack): a length of 1 is a literal, else the
        //     deferrals are literals and the match is the rest of the length at the distance of the entry behind them
        const uint32_t w = adv_pack(st);
        const uint32_t len = w & ADV_LEN_MASK, run = (w >> ADV_RUN_SHIFT) & ADV_RUN_MASK, fromq = w >> ADV_FROMQ_SHIFT;
        if (len != st.adv) s.hier_mismatch = true;
        if (run != ADV_RUN_MANY) {
            const uint32_t nl = len == 1 ? 1u : run;
            uint32_t ml = 0, md = 0;
            if (len > 1) {
                ml = len - run;
                md = s.cfg.mode == MODE_RLE ? 1u : m_dist(fromq ? (uint32_t)Mq(j + run) : (uint32_t)M(j + run));
            }
            if (nl != st.nlit || ml != st.mlen || md != st.mdist) s.hier_mismatch = true;
        }
        // (b) the lazy step without its loop (k_adv, k_emit): "a + 1 beats a" is a property of the position, the chain of
        //     deferrals from j the run of such positions from j on (full-budget table only)
        if (s.cfg.mode == MODE_LAZY && !s.cfg.use_quarter) {
            const uint32_t L0 = m_len((uint32_t)M(j));
            const bool ok = j + 2 < n && L0 >= MIN_MATCH && !match_too_far(L0, m_dist((uint32_t)M(j)));
            uint64_t a = j;
            if (ok)
                while (m_len((uint32_t)M(a)) < s.cfg.lazy_lt && a + 3 < n && m_len((uint32_t)M(a + 1)) > m_len((uint32_t)M(a))) a++;
            const uint32_t want = ok ? (uint32_t)(a - j) + m_len((uint32_t)M(a)) : 1u;
            if (want != st.adv || (ok && (uint32_t)(a - j) != st.nlit)) s.hier_mismatch = true;
        }
    }
    uint64_t K = (n + S - 1) / S;
    if (K == 0) return;
    // level 0: X[k][e] = first path position >= seg_end starting from seg_start + e
    std::vector<std::vector<uint64_t>> levels_start;  // unit start per level
    std::vector<std::vector<uint32_t>> tables;        // [unit * ZONE + e] = exit - unit_end
    std::vector<uint64_t> unit_size;
    {
        std::vector<uint32_t> X(K * ZONE, 0);
        std::vector<uint16_t> J(S);
        for (uint64_t k = 0; k < K; k++) {
            uint64_t a = k * S, b = std::min<uint64_t>(n, a + S);
            for (uint64_t j = b; j-- > a;) {
                uint64_t t = j + s.adv[j];
                J[j - a] = (uint16_t)(t >= b ? t - b : J[t - a]);
            }
            for (uint32_t e = 0; e < ZONE; e++) X[k * ZONE + e] = (a + e < b) ? J[e] : (uint32_t)(a + e - b);
        }
        tables.push_back(X);
        unit_size.push_back(S);
    }
    // level up
    while (tables.back().size() / ZONE > 1) {
        const std::vector<uint32_t>& C = tables.back();
        uint64_t csize = unit_size.back();
        uint64_t nc = C.size() / ZONE;
        uint64_t nu = (nc + G - 1) / G;
        uint64_t usize = csize * G;
        std::vector<uint32_t> X(nu * ZONE, 0);
        for (uint64_t u = 0; u < nu; u++) {
            uint64_t ustart = u * usize, uend = std::min<uint64_t>(n, ustart + usize);
            for (uint32_t e = 0; e < ZONE; e++) {
                uint64_t pos = ustart + e;
                for (uint64_t c = u * G; c < std::min<uint64_t>(nc, (u + 1) * G); c++) {
                    uint64_t cstart = c * csize, cend = std::min<uint64_t>(n, cstart + csize);
                    if (pos < cend) pos = cend + C[c * ZONE + (pos - cstart)];
                }
                X[u * ZONE + e] = (uint32_t)(pos - uend);
            }
        }
        tables.push_back(X);
        unit_size.push_back(usize);
    }
    // top-down: E[level][unit] = first path position >= unit start
    std::vector<uint64_t> Ecur(1, 0);
    for (size_t lv = tables.size() - 1; lv-- > 0;) {
        const std::vector<uint32_t>& C = tables[lv];
        uint64_t csize = unit_size[lv];
        uint64_t nc = C.size() / ZONE;
        std::vector<uint64_t> En(nc, 0);
        for (uint64_t u = 0; u < Ecur.size(); u++) {
            uint64_t pos = Ecur[u];
            for (uint64_t c = u * G; c < std::min<uint64_t>(nc, (u + 1) * G); c++) {
                uint64_t cstart = c * csize, cend = std::min<uint64_t>(n, cstart + csize);
                if (pos < cstart) pos = cstart;  // cannot happen; path positions only grow
                En[c] = pos;
                if (pos < cend) pos = cend + C[c * ZONE + (pos - cstart)];
            }
        }
        Ecur = En;
    }
    // emit per segment from its entry, compare with the direct walk
    std::vector<uint32_t> toks;
    for (uint64_t k = 0; k < K; k++) {
        uint64_t b = std::min<uint64_t>(n, (k + 1) * S);
        uint64_t j = Ecur[k];
        while (j < b) {
            Step st = parse_step(M, Mq, j, n, s.cfg);
            for (uint32_t q = 0; q < st.nlit; q++) toks.push_back(tok_literal(s.in[j + q]));
            if (st.mlen) toks.push_back(tok_match(st.mlen, st.mdist));
            j += st.adv;
        }
    }
    if (toks != s.tokens) s.hier_mismatch = 1;
}

struct BitSink {
    std::vector<uint8_t>& out;
    void put(uint64_t bitpos, uint64_t bits, uint32_t nbits) {
        for (uint32_t i = 0; i < nbits; i++)
            if ((bits >> i) & 1) out[(bitpos + i) >> 3] |= (uint8_t)(1u << ((bitpos + i) & 7));
    }
};

}  // namespace

extern "C" {

// encode_lengths_rle (the reference's state machine) against encode_lengths_runs (run by run, what
// k_block_header does): 0 = same symbols, else 1 + the index of the first difference
int hostsim_rle_forms_agree(const uint8_t* lens, uint32_t n) {
    std::vector<uint16_t> a(n + 8), b(n + 8);
    uint32_t freqs[19] = {0};
    const uint32_t na = encode_lengths_rle(lens, n, a, freqs);
    const uint32_t nb = encode_lengths_runs(lens, n, b);
    if (na != nb) return 1 + (int)std::min(na, nb);
    for (uint32_t i = 0; i < na; i++)
        if (a[i] != b[i]) return 1 + (int)i;
    return 0;
}

struct hostsim_block {
    uint32_t btype, bfinal, ntok;
    uint64_t in_bytes, bit_start;
};

// returns 0 ok; -3 = reference would panic (Q13 slice out of range); fills *flags with bit0 =
// Q1 override used, bit1 = Q13 shifted stored source used, bit2 = hierarchical path mismatch
int hostsim_encode(const uint8_t* in, uint64_t n, uint32_t checks, uint32_t lazy_lt, uint32_t matching_type,
                   uint8_t* out, uint64_t cap, uint64_t* out_len, uint32_t* flags, hostsim_block* blocks,
                   uint64_t blocks_cap, uint64_t* n_blocks, uint32_t seg, uint32_t fan) {
    Sim s;
    s.n = n;
    s.in.assign(in, in + n);
    s.in.resize(n + 16, 0);
    s.cfg.checks = checks;
    s.cfg.lazy_lt = lazy_lt < 32768 ? lazy_lt : 32768;
    s.cfg.mode = matching_type == 0 ? MODE_GREEDY : (checks == 0 ? MODE_RLE : MODE_LAZY);
    s.cfg.use_quarter = (s.cfg.mode == MODE_LAZY && s.cfg.lazy_lt > 32) ? 1 : 0;
    if (s.cfg.mode == MODE_LAZY && s.cfg.lazy_lt < 3) return -4;  // unsupported
    *flags = 0;

    HashOverride ov = {0, 0, 0, 0, 0, nullptr, 0, nullptr, 0, nullptr};
    for (int pass = 0; pass < 2; pass++) {
        if (s.cfg.mode != MODE_RLE && checks > 0) stage_links(s, ov);
        stage_match(s);
        stage_parse_direct(s);
        if (pass == 1 || s.cfg.mode == MODE_RLE || checks == 0) break;
        // Q1 (lz77.rs:628-638): did block 0 fill inside the first window with overlap == 0?
        if (s.tokens.size() < MAX_BUFFER_LENGTH || n < 2) break;
        uint32_t t = s.tokens[MAX_BUFFER_LENGTH - 1];
        uint64_t tp = s.tok_pos[MAX_BUFFER_LENGTH - 1];
        uint64_t lp, w;  // loop position at which the token was pushed; restart position
        if (s.cfg.mode == MODE_LAZY) {
            bool hashable_next = (tp + 1) + 2 < n;  // pushed in the normal branch of position tp+1
            if (t >> 16) {
                lp = tp + 1;
                w = tp + tok_cover(t);
            } else if (hashable_next) {
                lp = tp + 1;
                w = tp + 2;
            } else {
                lp = tp + 1;  // tail pushes: positions without hash byte; override is a no-op
                w = tp + 1;
            }
        } else {
            lp = tp;
            w = tp + tok_cover(t);
        }
        if (lp < WINDOW_SIZE && w <= WINDOW_SIZE) {
            ov.on = 1;
            ov.pos = w;
            ov.b0 = in[0];
            ov.b1 = in[1];
            *flags |= 1;
            continue;
        }
        break;
    }
    if (seg) stage_parse_hier(s, seg, fan ? fan : 4);
    if (s.hier_mismatch) *flags |= 4;

    // blocks
    uint64_t T = s.tokens.size();
    uint64_t nb = T / MAX_BUFFER_LENGTH + 1;
    *n_blocks = nb;
    std::vector<BlockHeader> hdr(nb);
    std::vector<BlockPlan> plan(nb);
    std::vector<uint64_t> bstart(nb + 1);
    for (uint64_t b = 0; b < nb; b++) {
        uint64_t t0 = b * MAX_BUFFER_LENGTH;
        bstart[b] = t0 < T ? s.tok_pos[t0] : n;
    }
    bstart[nb] = n;
    std::vector<HuffNode> scratch(288);
    uint64_t bitpos = 0;
    std::vector<uint64_t> src_shift(nb, 0);
    for (uint64_t b = 0; b < nb; b++) {
        uint64_t t0 = b * MAX_BUFFER_LENGTH, t1 = std::min<uint64_t>(T, t0 + MAX_BUFFER_LENGTH);
        uint32_t llf[NUM_LL] = {0}, df[NUM_DIST] = {0};
        llf[END_OF_BLOCK] = 1;
        for (uint64_t t = t0; t < t1; t++) {
            uint32_t tk = s.tokens[t];
            if (tk >> 16) {
                uint32_t c, eb, ev;
                length_symbol(tk & 0xff, &c, &eb, &ev);
                llf[257 + c]++;
                distance_symbol(tk >> 16, &c, &eb, &ev);
                df[c]++;
            } else {
                llf[tk & 0xff]++;
            }
        }
        const uint32_t* pl = llf;
        const uint32_t* pd = df;
        build_block_header(pl, pd, hdr[b], scratch);
        uint64_t in_bytes = bstart[b + 1] - bstart[b];
        plan_block(hdr[b].dyn_bits, hdr[b].dyn_est, hdr[b].static_est, hdr[b].fixed_bits, in_bytes, b + 1 == nb, bitpos,
                   &plan[b]);
        // Q13: a full block whose last value is a match crossing the end of a
        // non-first window makes the reference slide before it reads the stored bytes.
        if (plan[b].btype == BT_STORED && t1 - t0 == MAX_BUFFER_LENGTH) {
            uint32_t tk = s.tokens[t1 - 1];
            uint64_t tp = s.tok_pos[t1 - 1];
            if (tk >> 16) {
                uint64_t lp = (s.cfg.mode == MODE_LAZY) ? tp + 1 : tp;
                uint64_t wdx = lp / WINDOW_SIZE;
                uint64_t wend = (wdx + 1) * (uint64_t)WINDOW_SIZE;
                uint64_t mend = tp + tok_cover(tk);
                if (wdx >= 1 && mend > wend) {
                    // after the slide the buffer holds [wdx*32768, min(n, wdx*32768 + 65794))
                    uint64_t buf_end = std::min<uint64_t>(n, wdx * (uint64_t)WINDOW_SIZE + 65794);
                    if (mend + WINDOW_SIZE > buf_end) return -3;
                    src_shift[b] = WINDOW_SIZE;
                    *flags |= 2;
                }
            }
        }
        bitpos += plan[b].bit_len;
        if (b < blocks_cap) {
            blocks[b].btype = plan[b].btype;
            blocks[b].bfinal = plan[b].bfinal;
            blocks[b].ntok = (uint32_t)(t1 - t0);
            blocks[b].in_bytes = in_bytes;
            blocks[b].bit_start = plan[b].bit_start;
        }
    }
    uint64_t total_bytes = (bitpos + 7) / 8;
    *out_len = total_bytes;
    if (total_bytes > cap) return -2;
    std::vector<uint8_t> o(total_bytes + 8, 0);
    BitSink sink{o};
    for (uint64_t b = 0; b < nb; b++) {
        uint64_t t0 = b * MAX_BUFFER_LENGTH, t1 = std::min<uint64_t>(T, t0 + MAX_BUFFER_LENGTH);
        uint64_t bp = plan[b].bit_start;
        const BlockHeader& h = hdr[b];
        if (plan[b].btype == BT_STORED) {
            uint64_t src = bstart[b] + src_shift[b];
            uint64_t left = bstart[b + 1] - bstart[b];
            do {
                uint64_t piece = std::min<uint64_t>(left, MAX_STORED_BLOCK_LENGTH);
                bool last_piece = piece == left;
                sink.put(bp, (plan[b].bfinal && last_piece) ? 1 : 0, 3);
                bp += 3;
                bp = (bp + 7) & ~7ull;
                sink.put(bp, piece & 0xffff, 16);
                sink.put(bp + 16, (~piece) & 0xffff, 16);
                bp += 32;
                for (uint64_t i = 0; i < piece; i++) o[(bp >> 3) + i] |= s.in[src + i];
                bp += piece * 8;
                src += piece;
                left -= piece;
            } while (left > 0);
            continue;
        }
        uint16_t llc[288] = {0}, dc[32] = {0};
        uint8_t lll[288], dl[32];
        if (plan[b].btype == BT_FIXED) {
            for (int i = 0; i < 288; i++) lll[i] = (uint8_t)fixed_ll_length(i);
            for (int i = 0; i < 32; i++) dl[i] = 5;
            sink.put(bp, plan[b].bfinal ? 3 : 2, 3);
            bp += 3;
        } else {
            memcpy(lll, h.ll_len, 288);
            memcpy(dl, h.d_len, 32);
            sink.put(bp, plan[b].bfinal ? 5 : 4, 3);
            bp += 3;
            sink.put(bp, h.n_ll - 257, 5);
            sink.put(bp + 5, h.n_d - 1, 5);
            sink.put(bp + 10, h.used_hclens >= 4 ? h.used_hclens - 4 : 0, 4);
            bp += 14;
            for (uint32_t i = 0; i < h.used_hclens; i++) {
                sink.put(bp, h.cl_len[hclen_order(i)], 3);
                bp += 3;
            }
            uint16_t clc[19] = {0};
            canonical_codes(h.cl_len, 19, clc);
            for (uint32_t i = 0; i < h.n_enc; i++) {
                uint32_t e = h.enc[i], kind = e >> 8, v = e & 0xff;
                uint32_t sym = el_symbol_index(e);
                sink.put(bp, clc[sym], h.cl_len[sym]);
                bp += h.cl_len[sym];
                if (kind == 1) {
                    sink.put(bp, v - 3, 2);
                    bp += 2;
                } else if (kind == 2) {
                    sink.put(bp, v - 3, 3);
                    bp += 3;
                } else if (kind == 3) {
                    sink.put(bp, v - 11, 7);
                    bp += 7;
                }
            }
        }
        canonical_codes(lll, 288, llc);
        canonical_codes(dl, 32, dc);
        for (uint64_t t = t0; t < t1; t++) {
            uint32_t nb2;
            uint64_t bits = token_bits(s.tokens[t], llc, lll, dc, dl, &nb2);
            sink.put(bp, bits, nb2);
            bp += nb2;
        }
        sink.put(bp, llc[END_OF_BLOCK], lll[END_OF_BLOCK]);
        bp += lll[END_OF_BLOCK];
        if (bp != plan[b].bit_start + plan[b].bit_len) return -5;  // plan/emit disagreement
    }
    memcpy(out, o.data(), total_bytes);
    return 0;
}

void hostsim_use_multi(int on) { g_multi = on; }
uint32_t hostsim_swz(uint32_t a) { return m3_swz(a); }  // the permuted pair table's address map (k_match3_swz)
void hostsim_force_ident(int on) { g_force_ident = on; }

// expose M for diffing: longest_match(prev_length=0) for every position
int hostsim_match_table(const uint8_t* in, uint64_t n, uint32_t checks, uint32_t* m_out) {
    Sim s;
    s.n = n;
    s.in.assign(in, in + n);
    s.in.resize(n + 16, 0);
    s.cfg.checks = checks;
    s.cfg.lazy_lt = 32;
    s.cfg.mode = MODE_LAZY;
    s.cfg.use_quarter = 0;
    HashOverride ov = {0, 0, 0, 0, 0, nullptr, 0, nullptr, 0, nullptr};
    stage_links(s, ov);
    stage_match(s);
    for (uint64_t i = 0; i < n; i++) m_out[i] = s.M[i];
    return 0;
}
}
