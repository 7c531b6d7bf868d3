/*
 * deflref.cpp -- CPU ORACLE: literal C++ restatement of the DEFLATE encode hot path of
 * image-rs/deflate-rs v1.0.0.  TEST INFRASTRUCTURE ONLY (see deflref.h for the rules and
 * the pinning status).  Each section names the reference file it follows; every function
 * cites file:line.  Quirks are restated, never "fixed".
 *
 * All citations are relative to /root/reference/.
 */
#include "deflref.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

struct RefPanic {
    std::string msg;
};
thread_local std::string g_last_panic;
thread_local int g_hazards = 0;
[[noreturn]] void ref_panic(const char* m) { throw RefPanic{m}; }
#define REF_ASSERT(c, m)         \
    do {                         \
        if (!(c)) ref_panic(m);  \
    } while (0)

/* ------------------------------------------------------------------------------------------
 * src/huffman_table.rs -- constants and symbol tables
 * ---------------------------------------------------------------------------------------- */
const size_t NUM_LENGTH_CODES = 29;          /* :6  */
const size_t NUM_DISTANCE_CODES = 30;        /* :10 */
const size_t NUM_LITERALS_AND_LENGTHS = 286; /* :14 */
const size_t MAX_CODE_LENGTH = 15;           /* :17 */
const u16 MIN_MATCH = 3;                     /* :20 */
const u16 MAX_MATCH = 258;                   /* :21 */
const size_t END_OF_BLOCK_POSITION = 256;    /* :28 */
const u16 LENGTH_BITS_START = 257;           /* :71 */

/* FIXED_CODE_LENGTHS :32-42 (0-143: 8, 144-255: 9, 256-279: 7, 280-287: 8) */
struct FixedLens {
    u8 ll[288];
    u8 d[32];
    FixedLens() {
        for (int i = 0; i < 288; i++) ll[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        for (int i = 0; i < 32; i++) d[i] = 5; /* FIXED_CODE_LENGTHS_DISTANCE :75 */
    }
};
const FixedLens FIXED;

/* LENGTH_EXTRA_BITS_LENGTH :45-47 */
const u8 LENGTH_EXTRA_BITS_LENGTH[NUM_LENGTH_CODES] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                                       2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
/* BASE_LENGTH :65-68 */
const u8 BASE_LENGTH[NUM_LENGTH_CODES] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  10,
                                          12, 14, 16, 20, 24, 28, 32, 40, 48, 56,
                                          64, 80, 96, 112, 128, 160, 192, 224, 255};
/* DISTANCE_BASE :108-111 */
const u16 DISTANCE_BASE[NUM_DISTANCE_CODES] = {0,    1,    2,    3,    4,    6,     8,     12,
                                               16,   24,   32,   48,   64,   96,    128,   192,
                                               256,  384,  512,  768,  1024, 1536,  2048,  3072,
                                               4096, 6144, 8192, 12288, 16384, 24576};

/* LENGTH_CODE :50-62 and DISTANCE_CODES :77-99 are plain RFC 1951 lookup tables; they are
 * rebuilt here from the RFC ranges instead of being transcribed (same values, checked by the
 * KATs test_get_length_code / test_distance_code). */
struct SymTables {
    u8 length_code[256];
    u8 distance_codes[512];
    SymTables() {
        /* stored length s = length-3; code n covers [BASE_LENGTH[n], BASE_LENGTH[n+1]) and
         * code 28 is exactly 255 (length 258) */
        for (int s = 0; s < 256; s++) {
            int n = 0;
            for (int c = 0; c < 28; c++)
                if (s >= BASE_LENGTH[c]) n = c;
            if (s == 255) n = 28;
            length_code[s] = (u8)n;
        }
        /* first half: index = distance-1 for distances 1..256; second half: index =
         * 256 + ((distance-1)>>7) for distances 257..32768 (entries 256,257 unused = 0) */
        for (int i = 0; i < 512; i++) distance_codes[i] = 0;
        for (int d = 1; d <= 256; d++) distance_codes[d - 1] = code_of(d);
        for (int d = 257; d <= 32768; d++) distance_codes[256 + ((d - 1) >> 7)] = code_of(d);
    }
    static u8 code_of(int d) {
        int c = 0;
        for (int k = 0; k < 30; k++)
            if (d - 1 >= DISTANCE_BASE[k]) c = k;
        return (u8)c;
    }
};
const SymTables SYM;

/* :113-115 */
inline u8 num_extra_bits_for_length_code(u8 code) { return LENGTH_EXTRA_BITS_LENGTH[code]; }
/* :120-126 */
inline u8 num_extra_bits_for_distance_code(u8 code) {
    u8 c = code >> 1;
    c -= (c != 0) ? 1 : 0;
    return c;
}
struct ExtraBits { /* :131-139 */
    u16 code_number;
    u8 num_bits;
    u16 value;
};
/* :143-147 */
inline size_t get_length_code(u16 length) {
    return (size_t)SYM.length_code[(u8)(u16)(length - MIN_MATCH)] + LENGTH_BITS_START;
}
/* :150-164 */
inline ExtraBits get_length_code_and_extra_bits(u8 stored_length) {
    u8 n = SYM.length_code[stored_length];
    u8 base = BASE_LENGTH[n];
    u8 num_bits = num_extra_bits_for_length_code(n);
    return ExtraBits{(u16)(n + LENGTH_BITS_START), num_bits, (u16)(u8)(stored_length - base)};
}
/* :170-182 */
inline u8 get_distance_code(u16 distance) {
    size_t d = distance;
    if (d >= 1 && d <= 256) return SYM.distance_codes[d - 1];
    if (d >= 257 && d <= 32768) return SYM.distance_codes[256 + ((d - 1) >> 7)];
    return 0;
}
/* :184-194 */
inline ExtraBits get_distance_code_and_extra_bits(u16 distance) {
    u8 distance_code = get_distance_code(distance);
    u8 extra = num_extra_bits_for_distance_code(distance_code);
    u16 base = (u16)(DISTANCE_BASE[distance_code] + 1);
    return ExtraBits{distance_code, extra, (u16)(distance - base)};
}

/* src/bit_reverse.rs:3-10 */
inline u16 reverse_bits(u16 n, u8 length) {
    n = (u16)(((n & 0xaaaa) >> 1) | ((n & 0x5555) << 1));
    n = (u16)(((n & 0xcccc) >> 2) | ((n & 0x3333) << 2));
    n = (u16)(((n & 0xf0f0) >> 4) | ((n & 0x0f0f) << 4));
    n = (u16)(((n & 0xff00) >> 8) | ((n & 0x00ff) << 8));
    return (u16)(n >> (16 - length));
}

struct HuffmanCode { /* :196-200 */
    u16 code;
    u8 length;
};

/* build_length_count_table :232-249 */
void build_length_count_table(const u8* table, size_t n, u16 len_counts[16], size_t* max_length,
                              size_t* max_length_pos) {
    REF_ASSERT(n > 0, "BUG! Empty lengths!");
    size_t mx = 0;
    for (size_t i = 0; i < n; i++) mx = std::max<size_t>(mx, table[i]);
    REF_ASSERT(mx <= MAX_CODE_LENGTH, "assert max_length <= MAX_CODE_LENGTH");
    size_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        u8 length = table[i];
        if (length > 0) {
            len_counts[length] += 1;
            pos = i;
        }
    }
    *max_length = mx;
    *max_length_pos = pos;
}

/* create_codes_in_place :253-278 */
void create_codes_in_place(u16* code_table, const u8* length_table, size_t n) {
    u16 len_counts[16] = {0};
    size_t max_length, max_length_pos;
    build_length_count_table(length_table, n, len_counts, &max_length, &max_length_pos);
    u16 code = 0;
    std::vector<u16> next_code;
    next_code.push_back(code);
    for (size_t bits = 1; bits <= max_length; bits++) {
        code = (u16)((u16)(code + len_counts[bits - 1]) << 1);
        next_code.push_back(code);
    }
    for (size_t i = 0; i <= max_length_pos; i++) {
        size_t length = length_table[i];
        if (length != 0) {
            code_table[i] = reverse_bits(next_code[length], (u8)length);
            next_code[length] = (u16)(next_code[length] + 1); /* wrapping_add */
        }
    }
}

/* HuffmanTable :281-424 */
struct HuffmanTable {
    u16 codes[288];
    u8 code_lengths[288];
    u16 distance_codes[32];
    u8 distance_code_lengths[32];
    HuffmanTable() { /* empty() :291-298 */
        memset(codes, 0, sizeof codes);
        memset(code_lengths, 0, sizeof code_lengths);
        memset(distance_codes, 0, sizeof distance_codes);
        memset(distance_code_lengths, 0, sizeof distance_code_lengths);
    }
    void update_from_lengths() { /* :331-337 */
        create_codes_in_place(codes, code_lengths, 288);
        create_codes_in_place(distance_codes, distance_code_lengths, 32);
    }
    void set_to_fixed() { /* :339-343 */
        memcpy(code_lengths, FIXED.ll, 288);
        memcpy(distance_code_lengths, FIXED.d, 32);
        update_from_lengths();
    }
    HuffmanCode get_ll_huff(size_t v) const { return HuffmanCode{codes[v], code_lengths[v]}; }
    HuffmanCode get_literal(u8 v) const { return get_ll_huff(v); }                 /* :360-363 */
    HuffmanCode get_end_of_block() const { return get_ll_huff(END_OF_BLOCK_POSITION); } /* :367 */
    void get_length_huffman(u8 stored, HuffmanCode* c, HuffmanCode* e) const { /* :373-385 */
        ExtraBits d = get_length_code_and_extra_bits(stored);
        *c = get_ll_huff(d.code_number);
        *e = HuffmanCode{d.value, d.num_bits};
    }
    void get_distance_huffman(u16 distance, HuffmanCode* c, HuffmanCode* e) const { /* :391-410 */
        ExtraBits d = get_distance_code_and_extra_bits(distance);
        *c = HuffmanCode{distance_codes[d.code_number], distance_code_lengths[d.code_number]};
        *e = HuffmanCode{d.value, d.num_bits};
    }
};

/* ------------------------------------------------------------------------------------------
 * src/bitstream.rs -- LsbWriter (64-bit accumulator variant :8-35)
 * ---------------------------------------------------------------------------------------- */
struct LsbWriter {
    std::vector<u8> w;
    u8 bits = 0;
    u64 acc = 0;
    static const u8 FLUSH_AT = 48; /* :16 */
    u8 pending_bits() const { return bits; } /* :71-73 */
    void push() {                            /* :21-33 */
        for (int i = 0; i < 6; i++) w.push_back((u8)(acc >> (8 * i)));
    }
    void write_bits(u16 v, u8 n) { /* :76-86 */
        acc |= ((u64)v) << bits;
        bits = (u8)(bits + n);
        while (bits >= FLUSH_AT) {
            push();
            acc >>= FLUSH_AT;
            bits = (u8)(bits - FLUSH_AT);
        }
    }
    void write_bits_finish(u16 v, u8 n) { /* :88-97 */
        acc |= ((u64)v) << bits;
        bits = (u8)(bits + n % 8);
        while (bits >= 8) {
            w.push_back((u8)acc);
            acc >>= 8;
            bits = (u8)(bits - 8);
        }
    }
    void flush_raw() { /* :99-106 */
        u8 missing = (u8)(FLUSH_AT - bits);
        if (missing > 0 && bits > 0) write_bits_finish(0, missing);
    }
    size_t write(const u8* buf, size_t n) { /* impl Write :110-119 */
        if (acc == 0) {
            w.insert(w.end(), buf, buf + n);
        } else {
            for (size_t i = 0; i < n; i++) write_bits((u16)buf[i], 8);
        }
        return n;
    }
};

/* ------------------------------------------------------------------------------------------
 * src/lzvalue.rs :42-76 and src/output_writer.rs
 * ---------------------------------------------------------------------------------------- */
struct LZValue {
    u8 litlen;
    u16 distance;
    static LZValue literal(u8 v) { return LZValue{v, 0}; }          /* :50-55 */
    static LZValue length_distance(u16 length, u16 distance) {      /* :59-67 */
        return LZValue{(u8)(length - MIN_MATCH), distance};
    }
};

const size_t MAX_BUFFER_LENGTH = 1024 * 31; /* output_writer.rs:19 */

struct DynamicWriter { /* output_writer.rs:28-118 */
    std::vector<LZValue> buffer;
    u16 frequencies[NUM_LITERALS_AND_LENGTHS];
    u16 distance_frequencies[NUM_DISTANCE_CODES];
    DynamicWriter() { /* new :75-85 */
        buffer.reserve(MAX_BUFFER_LENGTH);
        clear_frequencies();
    }
    bool full() const { return buffer.size() >= MAX_BUFFER_LENGTH; } /* check_buffer_length :38-44 */
    bool write_literal(u8 literal) {                                 /* :47-52 */
        buffer.push_back(LZValue::literal(literal));
        frequencies[literal] = (u16)(frequencies[literal] + 1);
        return full();
    }
    bool write_length_distance(u16 length, u16 distance) { /* :55-65 */
        buffer.push_back(LZValue::length_distance(length, distance));
        size_t l = get_length_code(length);
        frequencies[l] = (u16)(frequencies[l] + 1);
        u8 d = get_distance_code(distance);
        distance_frequencies[d] = (u16)(distance_frequencies[d] + 1);
        return full();
    }
    bool write_length_rle(u16 length) { /* :90-98 */
        buffer.push_back(LZValue::length_distance(length, 1));
        size_t l = get_length_code(length);
        frequencies[l] = (u16)(frequencies[l] + 1);
        distance_frequencies[0] = (u16)(distance_frequencies[0] + 1);
        return full();
    }
    size_t buffer_length() const { return buffer.size(); }
    void clear_frequencies() { /* :104-108 */
        memset(frequencies, 0, sizeof frequencies);
        memset(distance_frequencies, 0, sizeof distance_frequencies);
        frequencies[END_OF_BLOCK_POSITION] = 1;
    }
    void clear() { /* :114-117 */
        clear_frequencies();
        buffer.clear();
    }
};

/* ------------------------------------------------------------------------------------------
 * src/chained_hash_table.rs
 * ---------------------------------------------------------------------------------------- */
const size_t WINDOW_SIZE = 32768;          /* :1 */
const size_t WINDOW_MASK = WINDOW_SIZE - 1; /* :2 */
const u16 HASH_SHIFT = 5;                   /* :5 */
const u16 HASH_MASK = (u16)WINDOW_MASK;     /* :6 */

/* update_hash :55-62 */
inline u16 update_hash(u16 current_hash, u8 to_insert) {
    return (u16)(((u16)(current_hash << HASH_SHIFT) ^ (u16)to_insert) & HASH_MASK);
}

struct ChainedHashTable {
    u16 current_hash;
    std::vector<u16> head, prev;
    ChainedHashTable() : current_hash(0), head(WINDOW_SIZE), prev(WINDOW_SIZE) { /* create_tables :34-51 */
        for (size_t n = 0; n < WINDOW_SIZE; n++) head[n] = (u16)n;
        prev = head;
    }
    void reset() { /* :98-109 -- Q11: the copy into `prev` goes to a temporary; prev stays stale */
        current_hash = 0;
        for (size_t n = 0; n < WINDOW_SIZE; n++) head[n] = (u16)n;
    }
    void add_initial_hash_values(u8 v1, u8 v2) { /* :111-114 */
        current_hash = update_hash(current_hash, v1);
        current_hash = update_hash(current_hash, v2);
    }
    void add_hash_value(size_t position, u8 value) { /* :118-138 */
        u16 new_hash = update_hash(current_hash, value);
        add_with_hash(position, new_hash);
        current_hash = new_hash;
    }
    void set_hash(u16 h) { current_hash = h; } /* :142-144 */
    void add_with_hash(size_t position, u16 hash) { /* :148-158 */
        prev[position & WINDOW_MASK] = head[hash];
        head[hash] = (u16)position;
    }
    u16 get_prev(size_t bytes) const { return prev[bytes & WINDOW_MASK]; } /* :173-175 */
    static u16 slide_value(u16 b, u16 pos, u16 bytes) { return b >= bytes ? (u16)(b - bytes) : pos; } /* :197-203 */
    void slide(size_t bytes) { /* :206-219 */
        for (size_t n = 0; n < WINDOW_SIZE; n++) head[n] = slide_value(head[n], (u16)n, (u16)bytes);
        for (size_t n = 0; n < WINDOW_SIZE; n++) prev[n] = slide_value(prev[n], (u16)n, (u16)bytes);
    }
};

/* ------------------------------------------------------------------------------------------
 * src/matching.rs
 * ---------------------------------------------------------------------------------------- */
/* get_match_length :13-73 (the live "naive" version :67-72) */
inline size_t get_match_length(const u8* data, size_t len, size_t current_pos, size_t pos_to_check) {
    size_t n = 0;
    size_t a = current_pos, b = pos_to_check;
    while (n < MAX_MATCH && a < len && b < len && data[a] == data[b]) {
        n++;
        a++;
        b++;
    }
    return n;
}

/* longest_match :87-166 */
void longest_match(const u8* data, size_t len, const ChainedHashTable& hash_table, size_t position,
                   size_t prev_length, u16 max_hash_checks, size_t* out_len, size_t* out_dist) {
    if (prev_length >= MAX_MATCH || position + prev_length >= len) { /* :98-100 */
        *out_len = 0;
        *out_dist = 0;
        return;
    }
    size_t limit = position > WINDOW_SIZE ? position - WINDOW_SIZE : 0; /* :102-106 */
    prev_length = std::max<size_t>(prev_length, 1);                     /* :110 */
    size_t max_length = std::min<size_t>(len - position, MAX_MATCH);    /* :112 */
    size_t current_head = position;
    size_t best_length = prev_length;
    size_t best_distance = 0;
    size_t prev_head;
    for (u32 i = 0; i < max_hash_checks; i++) { /* :124 */
        prev_head = current_head;
        current_head = hash_table.get_prev(current_head);
        if (current_head >= prev_head || current_head < limit) break; /* :127-132 */
        /* :141-143 two-byte probe at offsets best_length-1, best_length */
        REF_ASSERT(position + best_length < len, "index out of bounds in longest_match probe");
        if (data[position + best_length - 1] == data[current_head + best_length - 1] &&
            data[position + best_length] == data[current_head + best_length]) {
            size_t length = get_match_length(data, len, position, current_head); /* :148 */
            if (length > best_length) {
                best_length = length;
                best_distance = position - current_head;
                if (length == max_length) break; /* :152-156 */
            }
        }
    }
    if (best_length > prev_length) { /* :161-165 */
        *out_len = best_length;
        *out_dist = best_distance;
    } else {
        *out_len = 0;
        *out_dist = 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * src/input_buffer.rs
 * ---------------------------------------------------------------------------------------- */
const size_t BUFFER_SIZE = WINDOW_SIZE * 2 + MAX_MATCH; /* :8 */

struct Slice { /* Option<&[u8]> */
    const u8* p;
    size_t n;
    bool some;
};
inline Slice some(const u8* p, size_t n) { return Slice{p, n, true}; }
inline Slice none() { return Slice{nullptr, 0, false}; }
inline Slice unwrap_or_empty(Slice s) { return s.some ? s : Slice{nullptr, 0, true}; }

struct InputBuffer {
    std::vector<u8> buffer;
    InputBuffer() { buffer.reserve(BUFFER_SIZE); }
    size_t current_end() const { return buffer.size(); } /* :49-51 */
    Slice add_data(const u8* data, size_t n) {           /* :31-46 */
        if (current_end() + n > BUFFER_SIZE) {
            size_t space_left = BUFFER_SIZE - buffer.size();
            buffer.insert(buffer.end(), data, data + space_left);
            return some(data + space_left, n - space_left);
        }
        buffer.insert(buffer.end(), data, data + n);
        return none();
    }
    Slice slide(const u8* data, size_t n) { /* :56-91 */
        REF_ASSERT(buffer.size() > WINDOW_SIZE * 2, "assert buffer.len() > WINDOW_SIZE*2");
        u8* lower = buffer.data();
        u8* upper = buffer.data() + WINDOW_SIZE;
        size_t upper_total = buffer.size() - WINDOW_SIZE;
        memcpy(lower, upper, WINDOW_SIZE);
        size_t lookahead_len = upper_total - WINDOW_SIZE;
        memmove(upper, upper + WINDOW_SIZE, lookahead_len);
        size_t upper_len = upper_total - lookahead_len;
        size_t end = std::min(n, upper_len);
        /* upper[lookahead_len .. lookahead_len+end] = data[..end] (always fits: the slice
         * `upper` is upper_total = 32768+lookahead_len long) */
        memcpy(upper + lookahead_len, data, end);
        size_t final_len = WINDOW_SIZE + lookahead_len + end;
        buffer.resize(final_len); /* truncate */
        if (n > upper_len) return some(data + end, n - end);
        return none();
    }
};

/* ------------------------------------------------------------------------------------------
 * src/lz77.rs
 * ---------------------------------------------------------------------------------------- */
enum MatchingType { Greedy = 0, Lazy = 1 }; /* :27-37 */
enum Flush { FlushNone, FlushSync, FlushFinish }; /* compress.rs:18-30 */

struct ChunkState { /* :162-185 */
    u16 current_length = 0;
    u16 current_distance = 0;
    u8 prev_byte = 0;
    u8 cur_byte = 0;
    bool add = false;
};

struct LZ77State { /* :49-143 */
    ChainedHashTable hash_table;
    bool is_first_window = true;
    bool is_last_block = false;
    size_t overlap = 0;
    u64 current_block_input_bytes = 0;
    u16 max_hash_checks;
    u16 lazy_if_less_than;
    MatchingType matching_type;
    ChunkState match_state;
    size_t bytes_to_hash = 0;
    bool was_synced = false;
    LZ77State(u16 mhc, u16 lilt, MatchingType mt)
        : max_hash_checks(mhc), lazy_if_less_than(lilt), matching_type(mt) {}
    void reset() { /* :99-107 (was_synced is NOT reset) */
        hash_table.reset();
        is_first_window = true;
        is_last_block = false;
        overlap = 0;
        current_block_input_bytes = 0;
        match_state = ChunkState();
        bytes_to_hash = 0;
    }
    size_t pending_byte_as_num() const { return match_state.add ? 1 : 0; } /* :134-142 */
};

struct ProcessStatus { /* :149-156 */
    bool buffer_full;
    size_t position;
};
inline ProcessStatus ps_ok() { return ProcessStatus{false, 0}; }
inline ProcessStatus ps_full(size_t p) { return ProcessStatus{true, p}; }

/* The three cursors of create_iterators :281-303 */
struct Iters {
    size_t end;   /* min(data.len(), iterated_data.end) */
    size_t ipos;  /* next position insert_it yields */
    size_t hidx;  /* next index hash_it yields */
    size_t dlen;
};
Iters create_iterators(size_t data_len, size_t start, size_t range_end) {
    Iters it;
    it.end = std::min(data_len, range_end);
    REF_ASSERT(start <= it.end, "slice index starts after end in create_iterators");
    it.ipos = start;
    it.hidx = (data_len - start > 2) ? start + 2 : data_len;
    it.dlen = data_len;
    return it;
}

/* add_to_hash_table :236-256 */
void add_to_hash_table(size_t bytes_to_add, const u8* data, Iters& it, ChainedHashTable& hash_table) {
    u16 hash = hash_table.current_hash;
    size_t taken = 0;
    while (taken < bytes_to_add && it.ipos < it.end) {
        size_t ipos = it.ipos++;
        taken++;
        if (it.hidx < it.dlen) {
            u8 hb = data[it.hidx++];
            hash = update_hash(hash, hb);
            hash_table.add_with_hash(ipos, hash);
        }
    }
    hash_table.set_hash(hash);
}

/* match_too_far :275-278 */
inline bool match_too_far(size_t match_len, size_t match_dist) {
    return match_len == MIN_MATCH && match_dist > 8 * 1024;
}

struct ChunkResult {
    size_t overlap;
    ProcessStatus status;
};

/* process_chunk_lazy :305-486 */
ChunkResult process_chunk_lazy(const u8* data, size_t data_len, size_t r_start, size_t r_end,
                               ChunkState& state, ChainedHashTable& hash_table, DynamicWriter& writer,
                               u16 max_hash_checks, size_t lazy_if_less_than) {
    Iters it = create_iterators(data_len, r_start, r_end);
    const size_t end = it.end;
    u16 prev_length = state.current_length;
    u16 prev_distance = state.current_distance;
    state.current_length = 0;
    state.current_distance = 0;
    size_t overlap = 0;
    bool ignore_next = (size_t)prev_length >= lazy_if_less_than; /* :333 */
    state.prev_byte = state.cur_byte;                            /* :337 */

    while (it.ipos < it.end) { /* :340 */
        size_t position = it.ipos;
        u8 b = data[it.ipos++];
        state.cur_byte = b;
        if (it.hidx < it.dlen) { /* :342 */
            u8 hash_byte = data[it.hidx++];
            hash_table.add_hash_value(position, hash_byte);
            if (!ignore_next) { /* :347 */
                u16 checks = prev_length >= 32 ? (u16)(max_hash_checks >> 2) : max_hash_checks; /* :351-355 */
                size_t match_len, match_dist;
                longest_match(data, data_len, hash_table, position, prev_length, checks, &match_len, &match_dist);
                if (match_too_far(match_len, match_dist)) match_len = 0; /* :370-372 */
                if (match_len >= lazy_if_less_than) ignore_next = true;  /* :374-377 */
                state.current_length = (u16)match_len;
                state.current_distance = (u16)match_dist;
            } else { /* :380-386 */
                state.current_length = 0;
                state.current_distance = 0;
                ignore_next = false;
            }
            if (prev_length >= state.current_length && prev_length >= MIN_MATCH) { /* :388 */
                bool full = writer.write_length_distance(prev_length, prev_distance);
                u16 bytes_to_add = (u16)(prev_length - 2);
                add_to_hash_table(bytes_to_add, data, it, hash_table);
                if (position + prev_length > end) overlap = position + prev_length - end - 1; /* :413-416 */
                state.add = false;
                state.current_length = 0;
                state.current_distance = 0;
                if (full) return ChunkResult{overlap, ps_full(position + prev_length - 1)}; /* :424-427 */
                ignore_next = false;
            } else if (state.add) { /* :430-434 */
                if (writer.write_literal(state.prev_byte)) return ChunkResult{0, ps_full(position + 1)};
            } else {
                state.add = true;
            }
            prev_length = state.current_length;
            prev_distance = state.current_distance;
            state.prev_byte = b;
        } else { /* :442-483 */
            if (prev_length >= MIN_MATCH) {
                bool full = writer.write_length_distance(prev_length, prev_distance);
                state.current_length = 0;
                state.current_distance = 0;
                state.add = false;
                size_t o = position + prev_length; /* saturating_sub(end).saturating_sub(1) :455-457 */
                o = o > end ? o - end : 0;
                o = o > 1 ? o - 1 : 0;
                overlap = o;
                if (full) return ChunkResult{overlap, ps_full(end)};
                return ChunkResult{overlap, ps_ok()};
            }
            if (state.add) { /* :470-476 */
                state.add = false;
                if (writer.write_literal(state.prev_byte)) return ChunkResult{0, ps_full(position)};
            }
            if (writer.write_literal(b)) return ChunkResult{0, ps_full(position + 1)}; /* :482 */
        }
    }
    return ChunkResult{overlap, ps_ok()};
}

/* process_chunk_greedy :488-547 */
ChunkResult process_chunk_greedy(const u8* data, size_t data_len, size_t r_start, size_t r_end,
                                 ChainedHashTable& hash_table, DynamicWriter& writer, u16 max_hash_checks) {
    Iters it = create_iterators(data_len, r_start, r_end);
    const size_t end = it.end;
    size_t overlap = 0;
    while (it.ipos < it.end) {
        size_t position = it.ipos;
        u8 b = data[it.ipos++];
        if (it.hidx < it.dlen) {
            u8 hash_byte = data[it.hidx++];
            hash_table.add_hash_value(position, hash_byte);
            size_t match_len, match_dist;
            longest_match(data, data_len, hash_table, position, 0, max_hash_checks, &match_len, &match_dist);
            if (match_len >= MIN_MATCH && !match_too_far(match_len, match_dist)) { /* :512 */
                bool full = writer.write_length_distance((u16)match_len, (u16)match_dist);
                size_t bytes_to_add = match_len - 1;
                add_to_hash_table(bytes_to_add, data, it, hash_table);
                if (position + match_len > end) overlap = position + match_len - end; /* :526-529 */
                if (full) return ChunkResult{overlap, ps_full(position + match_len)};
            } else {
                if (writer.write_literal(b)) return ChunkResult{0, ps_full(position + 1)};
            }
        } else {
            if (writer.write_literal(b)) return ChunkResult{0, ps_full(position + 1)}; /* :543 */
        }
    }
    return ChunkResult{overlap, ps_ok()};
}

/* src/rle.rs: get_match_length_rle :13-18 */
inline size_t get_match_length_rle(const u8* data, size_t n, u8 prev) {
    size_t c = 0;
    while (c < n && c < MAX_MATCH && data[c] == prev) c++;
    return c;
}

/* src/rle.rs: process_chunk_greedy_rle :23-71 */
ChunkResult process_chunk_greedy_rle(const u8* data, size_t data_len, size_t r_start, size_t r_end,
                                     DynamicWriter& writer) {
    if (data_len == 0) return ChunkResult{0, ps_ok()};
    size_t end = std::min(data_len, r_end);
    size_t start = std::max<size_t>(r_start, 1);
    REF_ASSERT(start - 1 < data_len, "index out of bounds: data[start-1] in rle");
    u8 prev = data[start - 1];
    size_t chunk_begin = std::min(start, end); /* :37 */
    size_t chunk_len = end - chunk_begin;
    size_t n_it = 0; /* enumerate() index of the next element */
    size_t overlap = 0;
    if (r_start == 0 && data_len != 0) { /* :41-44 */
        if (writer.write_literal(data[0])) return ChunkResult{0, ps_full(1)};
    }
    while (n_it < chunk_len) { /* :46 */
        size_t n = n_it;
        u8 b = data[chunk_begin + n];
        n_it++;
        size_t position = n + start;
        size_t match_len = 0;
        if (prev == b) {
            REF_ASSERT(position <= data_len, "slice start out of range in rle");
            match_len = get_match_length_rle(data + position, data_len - position, prev);
        }
        if (match_len >= MIN_MATCH) {
            if (position + match_len > end) overlap = position + match_len - end;
            bool full = writer.write_length_rle((u16)match_len);
            if (full) return ChunkResult{overlap, ps_full(position + match_len)};
            n_it += (match_len - 2) + 1; /* insert_it.nth(match_len - 2) consumes match_len-1 items */
        } else {
            if (writer.write_literal(b)) return ChunkResult{0, ps_full(position + 1)};
        }
        prev = b;
    }
    return ChunkResult{overlap, ps_ok()};
}

/* process_chunk :192-232 (cfg!(test) NO_RLE switch omitted: not reachable from the public API) */
ChunkResult process_chunk(const u8* data, size_t data_len, size_t r_start, size_t r_end,
                          ChunkState& match_state, ChainedHashTable& hash_table, DynamicWriter& writer,
                          u16 max_hash_checks, size_t lazy_if_less_than, MatchingType matching_type,
                          bool avoid_rle) {
    if (matching_type == Greedy)
        return process_chunk_greedy(data, data_len, r_start, r_end, hash_table, writer, max_hash_checks);
    if (max_hash_checks > 0 || avoid_rle)
        return process_chunk_lazy(data, data_len, r_start, r_end, match_state, hash_table, writer,
                                  max_hash_checks, lazy_if_less_than);
    return process_chunk_greedy_rle(data, data_len, r_start, r_end, writer);
}

enum LZ77Status { NeedInput, EndBlock, Finished }; /* :550-558 */

struct BlockResult {
    size_t consumed;
    LZ77Status status;
    size_t position;
    bool slid_before_return; /* oracle-only: set when the BufferFull arm slid the buffer (Q13) */
};

const u16 NO_RLE = 43212; /* :23 */

/* lz77_compress_block :581-770 */
BlockResult lz77_compress_block(const u8* data, size_t data_n, LZ77State& state, InputBuffer& buffer,
                                DynamicWriter& writer, Flush flush, bool test_mode = false) {
    const size_t window_size = WINDOW_SIZE;
    const bool finish = flush == FlushFinish || flush == FlushSync;
    const bool sync = flush == FlushSync;
    size_t current_position = 0;
    LZ77Status status = EndBlock;
    bool add_initial = true;
    bool slid = false;

    if (state.was_synced) { /* :605-614 */
        if (buffer.current_end() > 2) {
            size_t pos_add = buffer.current_end() - 2;
            for (size_t n = 0; n < 2 && n < data_n; n++) state.hash_table.add_hash_value(n + pos_add, data[n]);
            add_initial = false;
        }
        state.was_synced = false;
    }

    Slice remaining_data = buffer.add_data(data, data_n); /* :617 */

    for (;;) {
        size_t pending_previous = state.pending_byte_as_num(); /* :622 */
        REF_ASSERT(writer.buffer_length() <= window_size * 2, "assert writer.buffer_length() <= window_size*2");
        if (buffer.current_end() >= window_size * 2 + MAX_MATCH || finish) { /* :627 */
            if (state.is_first_window) {
                if (buffer.buffer.size() >= 2 && add_initial && state.current_block_input_bytes == 0) { /* :629-638 (Q1) */
                    state.hash_table.add_initial_hash_values(buffer.buffer[0], buffer.buffer[1]);
                    add_initial = false;
                }
            } else if (buffer.current_end() >= window_size + 2) { /* :639-648 */
                size_t avail = buffer.buffer.size() - (window_size + 2);
                for (size_t n = 0; n < avail && n < state.bytes_to_hash; n++)
                    state.hash_table.add_hash_value(window_size + n, buffer.buffer[window_size + 2 + n]);
                state.bytes_to_hash = 0;
            }
            size_t window_start = state.is_first_window ? 0 : window_size;
            size_t start = state.overlap + window_start;
            size_t end = std::min(window_size + window_start, buffer.current_end());

            bool avoid_rle = test_mode && state.lazy_if_less_than == NO_RLE;
            ChunkResult cr = process_chunk(buffer.buffer.data(), buffer.buffer.size(), start, end,
                                           state.match_state, state.hash_table, writer, state.max_hash_checks,
                                           (size_t)state.lazy_if_less_than, state.matching_type, avoid_rle);
            size_t overlap = cr.overlap;
            state.bytes_to_hash = overlap; /* :669 */

            if (cr.status.buffer_full) { /* :671-700 */
                size_t written = cr.status.position;
                state.current_block_input_bytes +=
                    (u64)(written - start + pending_previous - state.pending_byte_as_num());
                if (overlap > 0) {
                    if (!state.is_first_window) {
                        if (state.max_hash_checks > 0) state.hash_table.slide(window_size);
                        Slice r = unwrap_or_empty(remaining_data);
                        remaining_data = buffer.slide(r.p, r.n);
                        slid = true;
                    } else {
                        state.is_first_window = false;
                    }
                    state.overlap = overlap;
                } else {
                    state.overlap = written - window_start;
                }
                current_position = written - state.pending_byte_as_num();
                break;
            }

            state.current_block_input_bytes +=
                (u64)(end - start + overlap + pending_previous - state.pending_byte_as_num()); /* :702-703 */
            state.overlap = overlap;

            if ((state.is_first_window || !remaining_data.some) && finish && end >= buffer.current_end()) { /* :709-712 */
                if (state.is_first_window) {
                    current_position = end - state.pending_byte_as_num();
                } else {
                    current_position = buffer.current_end();
                }
                if (!sync) {
                    state.is_last_block = true;
                    state.is_first_window = false;
                } else { /* :728-740 */
                    state.overlap = state.is_first_window ? end : buffer.current_end() - window_size;
                    state.was_synced = true;
                }
                status = Finished;
                break;
            } else if (state.is_first_window) {
                state.is_first_window = false;
            } else { /* :745-756 */
                if (state.max_hash_checks > 0) state.hash_table.slide(window_size);
                Slice r = unwrap_or_empty(remaining_data);
                remaining_data = buffer.slide(r.p, r.n);
            }
        } else {
            status = NeedInput; /* :757-762 */
            break;
        }
    }
    Slice r = unwrap_or_empty(remaining_data);
    return BlockResult{data_n - r.n, status, current_position, slid};
}

/* ------------------------------------------------------------------------------------------
 * src/length_encode.rs
 * ---------------------------------------------------------------------------------------- */
enum ELKind { EL_Length = 0, EL_CopyPrevious = 1, EL_RepeatZero3Bits = 2, EL_RepeatZero7Bits = 3 };
struct EncodedLength { /* :7-16 */
    u8 kind;
    u8 v;
};
const size_t COPY_PREVIOUS = 16, REPEAT_ZERO_3_BITS = 17, REPEAT_ZERO_7_BITS = 18; /* :34-36 */
const u8 MIN_REPEAT = 3;                                                            /* :38 */

/* from_prev_and_repeat :19-31 */
EncodedLength from_prev_and_repeat(u8 prev, u8 repeat) {
    if (prev == 0) return EncodedLength{(u8)(repeat <= 10 ? EL_RepeatZero3Bits : EL_RepeatZero7Bits), repeat};
    REF_ASSERT(prev >= 1 && prev <= 15, "panic!() in from_prev_and_repeat");
    return EncodedLength{EL_CopyPrevious, repeat};
}
/* update_out_and_freq :41-56 */
void update_out_and_freq(EncodedLength e, std::vector<EncodedLength>& out, u16 freqs[19]) {
    size_t index = e.kind == EL_Length ? e.v
                   : e.kind == EL_CopyPrevious ? COPY_PREVIOUS
                   : e.kind == EL_RepeatZero3Bits ? REPEAT_ZERO_3_BITS
                                                  : REPEAT_ZERO_7_BITS;
    REF_ASSERT(index < 19, "index out of bounds in update_out_and_freq");
    freqs[index] = (u16)(freqs[index] + 1);
    out.push_back(e);
}
/* not_max_repetitions :59-61 */
inline bool not_max_repetitions(u8 length_value, u8 repeats) {
    return (length_value == 0 && repeats < 138) || repeats < 6;
}

/* encode_lengths_m :82-155.  `lengths` is the concatenated sequence the caller chains. */
void encode_lengths_m(const u8* lengths, size_t n_len, std::vector<EncodedLength>& out, u16 frequencies[19]) {
    out.clear();
    u8 repeat = 0;
    REF_ASSERT(n_len > 0, "No length values!");
    size_t idx = 0;                 /* iter position */
    u8 prev = (u8)~lengths[0];      /* :94 */
    while (idx < n_len) {
        size_t n = idx;
        u8 l = lengths[idx++];
        bool peek_none = idx >= n_len;
        if (l == prev && not_max_repetitions(l, repeat)) repeat = (u8)(repeat + 1);
        if (l != prev || peek_none || !not_max_repetitions(l, repeat)) {
            if (repeat >= MIN_REPEAT) {
                EncodedLength val = from_prev_and_repeat(prev, repeat);
                update_out_and_freq(val, out, frequencies);
                repeat = 0;
                if (l != prev) {
                    if (l != 0 || peek_none) {
                        update_out_and_freq(EncodedLength{EL_Length, l}, out, frequencies);
                        repeat = 0;
                    } else {
                        repeat = 1;
                    }
                }
            } else {
                size_t extra_skip = (peek_none && l == prev) ? 1 : 0;
                /* lengths.clone().skip(n + extra_skip - repeat) */
                size_t skip = n + extra_skip - (size_t)repeat;
                size_t extra = (l != 0 || peek_none) ? 1 : 0;
                size_t take = (size_t)repeat + extra;
                for (size_t k = 0; k < take && skip + k < n_len; k++)
                    update_out_and_freq(EncodedLength{EL_Length, lengths[skip + k]}, out, frequencies);
                repeat = (u8)(1 - (u8)extra);
            }
        }
        prev = l;
    }
}

/* mod in_place :162-415 */
struct Node { /* :208-212 */
    u32 value;
    u16 symbol;
};

void step_1(std::vector<Node>& leaves) { /* :218-247 */
    size_t root = 0, leaf = 2;
    const size_t n = leaves.size();
    leaves[0].value += leaves[1].value;
    for (size_t next = 1; next + 1 < n; next++) {
        if (leaf >= n || leaves[root].value < leaves[leaf].value) {
            leaves[next].value = leaves[root].value;
            leaves[root].value = (u32)next;
            root++;
        } else {
            leaves[next].value = leaves[leaf].value;
            leaf++;
        }
        if (leaf >= n || (root < next && leaves[root].value < leaves[leaf].value)) {
            leaves[next].value += leaves[root].value;
            leaves[root].value = (u32)next;
            root++;
        } else {
            leaves[next].value += leaves[leaf].value;
            leaf++;
        }
    }
}

void step_2(std::vector<Node>& leaves) { /* :249-278 */
    const size_t n = leaves.size();
    leaves[n - 2].value = 0;
    for (size_t t = n + 1 - 3; t-- > 0;) leaves[t].value = leaves[leaves[t].value].value + 1;
    size_t available = 1, used = 0;
    u32 depth = 0;
    long root = (long)n - 2, next = (long)n - 1;
    while (available > 0) {
        while (root >= 0 && leaves[(size_t)root].value == depth) {
            used++;
            root--;
        }
        while (available > used) {
            REF_ASSERT(next >= 0, "index out of bounds in step_2");
            leaves[(size_t)next].value = depth;
            next--;
            available--;
        }
        available = 2 * used;
        depth++;
        used = 0;
    }
}

const size_t NUM_CODES_LENGTH = 33; /* :280-281 */

void enforce_max_code_lengths(u16 num_codes[NUM_CODES_LENGTH], size_t num_used, size_t max_len) { /* :290-327 */
    if (num_used > 1) {
        u16 num_above_max = 0;
        for (size_t i = max_len + 1; i < NUM_CODES_LENGTH; i++) num_above_max = (u16)(num_above_max + num_codes[i]);
        num_codes[max_len] = (u16)(num_codes[max_len] + num_above_max);
        u32 total = 0;
        for (size_t i = max_len; i >= 1; i--) total += ((u32)num_codes[i]) << (max_len - i);
        while (total != (1u << max_len)) {
            num_codes[max_len] = (u16)(num_codes[max_len] - 1);
            for (size_t i = max_len - 1; i >= 1; i--) {
                if (num_codes[i] != 0) {
                    num_codes[i] = (u16)(num_codes[i] - 1);
                    num_codes[i + 1] = (u16)(num_codes[i + 1] + 2);
                    break;
                }
            }
            total -= 1;
        }
    }
}

/* in_place_lengths :347-415.  `lengths_n` is the length of the caller's output slice (all of
 * it is zeroed first, :355-357). */
void in_place_lengths(const u16* frequencies, size_t freq_n, size_t max_len, std::vector<Node>& leaves,
                      u8* lengths, size_t lengths_n) {
    for (size_t i = 0; i < lengths_n; i++) lengths[i] = 0;
    leaves.clear();
    for (size_t n = 0; n < freq_n; n++)
        if (frequencies[n] > 0) leaves.push_back(Node{(u32)frequencies[n], (u16)n});
    if (leaves.size() == 1) {
        lengths[leaves[0].symbol] = 1;
        return;
    } else if (leaves.empty()) {
        return;
    }
    std::stable_sort(leaves.begin(), leaves.end(), [](const Node& a, const Node& b) { return a.value < b.value; }); /* :386 */
    step_1(leaves);
    step_2(leaves);
    u16 num_codes[NUM_CODES_LENGTH] = {0};
    for (const Node& l : leaves) {
        REF_ASSERT(l.value < NUM_CODES_LENGTH, "index out of bounds: num_codes[l.value]");
        num_codes[l.value] = (u16)(num_codes[l.value] + 1);
    }
    enforce_max_code_lengths(num_codes, leaves.size(), max_len);
    size_t li = leaves.size(); /* leaves.iter().rev() */
    for (size_t i = 1; i <= max_len; i++) {
        for (u16 k = 0; k < num_codes[i]; k++) {
            REF_ASSERT(li > 0, "unwrap on None: leaf_it.next()");
            li--;
            lengths[leaves[li].symbol] = (u8)i;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * src/huffman_lengths.rs
 * ---------------------------------------------------------------------------------------- */
const size_t MIN_NUM_LITERALS_AND_LENGTHS = 257; /* :18 */
const size_t MIN_NUM_DISTANCES = 1;              /* :20 */
const size_t NUM_HUFFMAN_LENGTHS = 19;           /* :22 */
const u8 HUFFMAN_LENGTH_ORDER[NUM_HUFFMAN_LENGTHS] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5,
                                                      11, 4, 12, 3, 13, 2, 14, 1, 15}; /* :27-29 */
const u8 HLIT_BITS = 5, HDIST_BITS = 5, HCLEN_BITS = 4; /* :32-34 */
const size_t MAX_HUFFMAN_CODE_LENGTH = 7;               /* :37 */
const u64 STORED_BLOCK_HEADER_LENGTH = 4;               /* :40 */
const u8 BLOCK_MARKER_LENGTH = 3;                       /* :41 */
const size_t MAX_STORED_BLOCK_LENGTH = 65535 / 2;       /* stored_block.rs:11 */

/* remove_trailing_zeroes :44-47 */
template <class T>
size_t remove_trailing_zeroes(const T* input, size_t n, size_t min_length) {
    size_t num_zeroes = 0;
    while (num_zeroes < n && input[n - 1 - num_zeroes] == 0) num_zeroes++;
    return std::max(n - num_zeroes, min_length);
}
/* extra_bits_for_huffman_length_code :50-56 */
inline u8 extra_bits_for_huffman_length_code(u8 code) { return (code == 16 || code == 17) ? 3 : code == 18 ? 7 : 0; }
/* calculate_huffman_length :59-68 */
u64 calculate_huffman_length(const u16* frequencies, const u8* code_lengths, size_t n) {
    u64 acc = 0;
    for (size_t i = 0; i < n; i++)
        acc += (u64)frequencies[i] * ((u64)code_lengths[i] + (u64)extra_bits_for_huffman_length_code((u8)i));
    return acc;
}
/* calculate_block_length :76-107.  NOTE (Q12): both callers are zipped against the
 * literal/length FIXED_CODE_LENGTHS, including the distance call. */
template <class F>
void calculate_block_length(const u16* frequencies, size_t n_freq, const u8* dyn_code_lengths, size_t n_dyn,
                            F get_num_extra_bits, u64* d_len, u64* s_len) {
    u64 d = 0, s = 0;
    size_t n = std::min(n_freq, std::min<size_t>(n_dyn, 288));
    for (size_t c = 0; c < n; c++) {
        u64 f = frequencies[c];
        u64 extra = get_num_extra_bits(c);
        d += f * ((u64)dyn_code_lengths[c] + extra);
        s += f * ((u64)FIXED.ll[c] + extra);
    }
    *d_len = d;
    *s_len = s;
}
/* stored_padding :113-124 */
u64 stored_padding(u8 pending_bits) {
    REF_ASSERT(pending_bits <= 8, "assert pending_bits <= 8");
    u8 free_space = (u8)(8 - pending_bits);
    if (free_space >= BLOCK_MARKER_LENGTH) return (u64)(free_space - BLOCK_MARKER_LENGTH);
    return (u64)(8 - (BLOCK_MARKER_LENGTH - free_space));
}
/* stored_length :132-143 */
u64 stored_length(u64 input_bytes) {
    REF_ASSERT(input_bytes >= 1, "Underflow calculating stored block length!");
    u64 num_blocks = (input_bytes - 1) / (u64)MAX_STORED_BLOCK_LENGTH + 1;
    return (input_bytes + STORED_BLOCK_HEADER_LENGTH * num_blocks + (num_blocks - 1)) * 8;
}

enum BlockType { BT_Stored = 0, BT_Fixed = 1, BT_Dynamic = 2 }; /* :145-149 */
struct DynamicBlockHeader {                                     /* :155-161 */
    std::vector<u8> huffman_table_lengths;
    size_t used_hclens;
};
struct LengthBuffers { /* deflate_state.rs:50-63 */
    std::vector<Node> leaf_buf;
    std::vector<EncodedLength> length_buf;
};

/* gen_huffman_lengths :167-287 */
BlockType gen_huffman_lengths(const u16* l_freqs_full, const u16* d_freqs_full, u64 num_input_bytes,
                              u8 pending_bits, u8 l_lengths[288], u8 d_lengths[32],
                              LengthBuffers& length_buffers, DynamicBlockHeader* header) {
    if (num_input_bytes <= 4) return BT_Fixed; /* :179-181 */
    size_t n_l = remove_trailing_zeroes(l_freqs_full, NUM_LITERALS_AND_LENGTHS, MIN_NUM_LITERALS_AND_LENGTHS);
    size_t n_d = remove_trailing_zeroes(d_freqs_full, NUM_DISTANCE_CODES, MIN_NUM_DISTANCES);
    in_place_lengths(l_freqs_full, n_l, MAX_CODE_LENGTH, length_buffers.leaf_buf, l_lengths, 288);
    in_place_lengths(d_freqs_full, n_d, MAX_CODE_LENGTH, length_buffers.leaf_buf, d_lengths, 32);
    size_t used_lengths = n_l, used_distances = n_d;
    u16 freqs[19] = {0};
    std::vector<u8> chain(l_lengths, l_lengths + used_lengths); /* :212-218 */
    chain.insert(chain.end(), d_lengths, d_lengths + used_distances);
    encode_lengths_m(chain.data(), chain.size(), length_buffers.length_buf, freqs);
    std::vector<u8> huffman_table_lengths(19, 0);
    in_place_lengths(freqs, 19, MAX_HUFFMAN_CODE_LENGTH, length_buffers.leaf_buf, huffman_table_lengths.data(), 19);
    size_t trailing = 0; /* :230-235 */
    while (trailing < NUM_HUFFMAN_LENGTHS &&
           huffman_table_lengths[HUFFMAN_LENGTH_ORDER[NUM_HUFFMAN_LENGTHS - 1 - trailing]] == 0)
        trailing++;
    size_t used_hclens = NUM_HUFFMAN_LENGTHS - trailing;
    u64 d_ll_length, s_ll_length, d_dist_length, s_dist_length;
    calculate_block_length(l_freqs_full, n_l, l_lengths, 288,
                           [](size_t c) -> u64 {
                               size_t k = c >= LENGTH_BITS_START ? c - LENGTH_BITS_START : 0; /* saturating_sub */
                               return num_extra_bits_for_length_code((u8)k);
                           },
                           &d_ll_length, &s_ll_length);
    calculate_block_length(d_freqs_full, n_d, d_lengths, 32,
                           [](size_t c) -> u64 { return num_extra_bits_for_distance_code((u8)c); },
                           &d_dist_length, &s_dist_length);
    u64 huff_table_length = calculate_huffman_length(freqs, huffman_table_lengths.data(), 19);
    u64 dynamic_length = d_ll_length + d_dist_length + huff_table_length + (u64)used_hclens * 3 +
                         HLIT_BITS + HDIST_BITS + HCLEN_BITS;
    u64 static_length = s_ll_length + s_dist_length;
    u64 stored_len = stored_length(num_input_bytes) + stored_padding((u8)(pending_bits % 8)); /* :269 */
    u64 used_length = std::min(std::min(dynamic_length, static_length), stored_len);
    if (used_length == static_length) return BT_Fixed; /* :277-286 (Q5) */
    if (used_length == stored_len) return BT_Stored;
    header->huffman_table_lengths = huffman_table_lengths;
    header->used_hclens = used_hclens;
    return BT_Dynamic;
}

/* write_huffman_lengths :290-369 */
void write_huffman_lengths(const DynamicBlockHeader& header, const HuffmanTable& huffman_table,
                           const std::vector<EncodedLength>& encoded_lengths, LsbWriter& writer) {
    size_t n_ll = remove_trailing_zeroes(huffman_table.code_lengths, 288, MIN_NUM_LITERALS_AND_LENGTHS);
    size_t n_d = remove_trailing_zeroes(huffman_table.distance_code_lengths, 32, MIN_NUM_DISTANCES);
    const std::vector<u8>& huffman_table_lengths = header.huffman_table_lengths;
    size_t used_hclens = header.used_hclens;
    REF_ASSERT(n_ll <= NUM_LITERALS_AND_LENGTHS, "assert literal_len_lengths.len() <= 286");
    REF_ASSERT(n_ll >= MIN_NUM_LITERALS_AND_LENGTHS, "assert literal_len_lengths.len() >= 257");
    REF_ASSERT(n_d <= NUM_DISTANCE_CODES, "assert distance_lengths.len() <= 30");
    REF_ASSERT(n_d >= MIN_NUM_DISTANCES, "assert distance_lengths.len() >= 1");
    u16 hlit = (u16)(n_ll - MIN_NUM_LITERALS_AND_LENGTHS);
    writer.write_bits(hlit, HLIT_BITS);
    u16 hdist = (u16)(n_d - MIN_NUM_DISTANCES);
    writer.write_bits(hdist, HDIST_BITS);
    size_t hclen = used_hclens >= 4 ? used_hclens - 4 : 0; /* saturating_sub */
    writer.write_bits((u16)hclen, HCLEN_BITS);
    for (size_t i = 0; i < used_hclens; i++) writer.write_bits((u16)huffman_table_lengths[HUFFMAN_LENGTH_ORDER[i]], 3);
    u16 codes[NUM_HUFFMAN_LENGTHS] = {0};
    create_codes_in_place(codes, huffman_table_lengths.data(), NUM_HUFFMAN_LENGTHS);
    for (const EncodedLength& v : encoded_lengths) {
        switch (v.kind) {
        case EL_Length:
            writer.write_bits(codes[v.v], huffman_table_lengths[v.v]);
            break;
        case EL_CopyPrevious:
            writer.write_bits(codes[COPY_PREVIOUS], huffman_table_lengths[COPY_PREVIOUS]);
            writer.write_bits((u16)(v.v - 3), 2);
            break;
        case EL_RepeatZero3Bits:
            writer.write_bits(codes[REPEAT_ZERO_3_BITS], huffman_table_lengths[REPEAT_ZERO_3_BITS]);
            writer.write_bits((u16)(v.v - 3), 3);
            break;
        default:
            writer.write_bits(codes[REPEAT_ZERO_7_BITS], huffman_table_lengths[REPEAT_ZERO_7_BITS]);
            writer.write_bits((u16)(v.v - 11), 7);
            break;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * src/encoder_state.rs
 * ---------------------------------------------------------------------------------------- */
const u16 FIXED_FIRST_BYTE = 0b010, FIXED_FIRST_BYTE_FINAL = 0b011;     /* :10-11 */
const u16 DYNAMIC_FIRST_BYTE = 0b100, DYNAMIC_FIRST_BYTE_FINAL = 0b101; /* :12-13 */

struct EncoderState { /* :23-26 */
    HuffmanTable huffman_table;
    LsbWriter writer;
    std::vector<u8>& inner_vec() { return writer.w; } /* :47-49 */
    void write_literal(u8 value) {                    /* :52-56 */
        HuffmanCode code = huffman_table.get_literal(value);
        writer.write_bits(code.code, code.length);
    }
    void write_lzvalue(LZValue v) { /* :58-83 */
        if (v.distance == 0) {      /* LZValue::value() lzvalue.rs:69-75 */
            write_literal(v.litlen);
        } else {
            HuffmanCode code, extra;
            huffman_table.get_length_huffman(v.litlen, &code, &extra);
            writer.write_bits(code.code, code.length);
            writer.write_bits(extra.code, extra.length);
            huffman_table.get_distance_huffman(v.distance, &code, &extra);
            writer.write_bits(code.code, code.length);
            writer.write_bits(extra.code, extra.length);
        }
    }
    void write_start_of_block(bool fixed, bool final_block) { /* :85-99 */
        if (final_block)
            writer.write_bits(fixed ? FIXED_FIRST_BYTE_FINAL : DYNAMIC_FIRST_BYTE_FINAL, 3);
        else
            writer.write_bits(fixed ? FIXED_FIRST_BYTE : DYNAMIC_FIRST_BYTE, 3);
    }
    void write_end_of_block() { /* :102-105 */
        HuffmanCode code = huffman_table.get_end_of_block();
        writer.write_bits(code.code, code.length);
    }
    void flush() { writer.flush_raw(); }                         /* :108-110 */
    void set_huffman_to_fixed() { huffman_table.set_to_fixed(); } /* :112-114 */
};

/* ------------------------------------------------------------------------------------------
 * src/stored_block.rs
 * ---------------------------------------------------------------------------------------- */
void write_stored_header(LsbWriter& writer, bool final_block) { /* :13-23 */
    writer.write_bits(final_block ? 1 : 0, 3);
    writer.flush_raw();
}
void compress_block_stored(const u8* input, size_t n, LsbWriter& writer) { /* :26-40 */
    REF_ASSERT(n <= 65535, "Stored block too long!");
    u8 l[2] = {(u8)(n & 0xff), (u8)((n >> 8) & 0xff)};
    writer.write(l, 2);
    u16 c = (u16)(~n);
    u8 nl[2] = {(u8)(c & 0xff), (u8)(c >> 8)};
    writer.write(nl, 2);
    writer.write(input, n);
}

/* ------------------------------------------------------------------------------------------
 * src/compress.rs
 * ---------------------------------------------------------------------------------------- */
const size_t LARGEST_OUTPUT_BUF_SIZE = 1024 * 32; /* :12 */

void flush_to_bitstream(const std::vector<LZValue>& buffer, EncoderState& state) { /* :34-39 */
    for (const LZValue& b : buffer) state.write_lzvalue(b);
    state.write_end_of_block();
}

void write_stored_block(const u8* input, size_t n, LsbWriter& writer, bool final_block) { /* :59-77 */
    if (n != 0) {
        size_t off = 0;
        while (off < n) {
            size_t chunk = std::min(MAX_STORED_BLOCK_LENGTH, n - off);
            bool last_chunk = off + chunk >= n;
            write_stored_header(writer, final_block && last_chunk);
            compress_block_stored(input + off, chunk, writer);
            off += chunk;
        }
    } else {
        write_stored_header(writer, final_block);
        compress_block_stored(nullptr, 0, writer);
    }
}

/* The wrapped writer W.  Vec<u8> accepts everything; max_write > 0 emulates a sink that takes
 * at most max_write bytes per call (tests/test.rs:175-199 SmallWriter). */
struct Sink {
    std::vector<u8> data;
    size_t max_write = 0;
    size_t write(const u8* p, size_t n) {
        size_t k = (max_write && n > max_write) ? max_write : n;
        data.insert(data.end(), p, p + k);
        return k;
    }
    void write_all(const u8* p, size_t n) {
        while (n) {
            size_t k = write(p, n);
            p += k;
            n -= k;
        }
    }
};

thread_local std::vector<deflref_block_info> g_trace;
thread_local bool g_trace_on = false;

struct DeflateState { /* deflate_state.rs:66-119 */
    LZ77State lz77_state;
    InputBuffer input_buffer;
    EncoderState encoder_state;
    DynamicWriter lz77_writer;
    LengthBuffers length_buffers;
    u64 bytes_written = 0;
    Sink* inner;
    size_t output_buf_pos = 0;
    Flush flush_mode = FlushNone;
    bool needs_flush = false;
    u64 bytes_written_control = 0;
    u64 flushed_bytes = 0; /* oracle-only: bytes already moved out of output_buf (for the trace) */
    DeflateState(const deflref_opts& o, Sink* w)
        : lz77_state(o.max_hash_checks, std::min<u16>(o.lazy_if_less_than, 32768 /* MAX_HASH_CHECKS */),
                     o.matching_type ? Lazy : Greedy),
          inner(w) {
        encoder_state.writer.w.reserve(1024 * 32);
    }
    std::vector<u8>& output_buf() { return encoder_state.inner_vec(); }
    u64 total_bits_now() { /* oracle-only helper for the trace */
        return (flushed_bytes + output_buf().size()) * 8 + encoder_state.writer.bits;
    }
    void clear_output_buf() {
        flushed_bytes += output_buf().size();
        output_buf().clear();
    }
    void reset(Sink* writer) { /* :133-152 */
        encoder_state.flush();
        inner->write_all(output_buf().data(), output_buf().size());
        output_buf().clear();
        input_buffer = InputBuffer();
        lz77_writer.clear();
        lz77_state.reset();
        bytes_written = 0;
        output_buf_pos = 0;
        flush_mode = FlushNone;
        needs_flush = false;
        bytes_written_control = 0;
        flushed_bytes = 0;
        inner = writer;
    }
};

struct IoResult {
    bool ok;
    bool interrupted;
    size_t n;
};

/* compress_data_dynamic_n :80-302 */
IoResult compress_data_dynamic_n(const u8* input, size_t input_n, DeflateState& ds, Flush flush) {
    size_t bytes_written = 0;
    const u8* slice = input;
    size_t slice_n = input_n;

    while (!ds.needs_flush) {
        size_t output_buf_len = ds.output_buf().size();
        size_t output_buf_pos = ds.output_buf_pos;
        if (output_buf_len > LARGEST_OUTPUT_BUF_SIZE) { /* :96-124 */
            size_t written = ds.inner->write(ds.output_buf().data() + output_buf_pos, output_buf_len - output_buf_pos);
            REF_ASSERT(output_buf_len >= output_buf_pos, "checked_sub unwrap");
            if (written < output_buf_len - output_buf_pos) {
                ds.output_buf_pos += written;
            } else {
                ds.needs_flush = false;
                ds.output_buf_pos = 0;
                ds.clear_output_buf();
            }
            if (bytes_written == 0) return IoResult{false, true, 0};
            return IoResult{true, false, bytes_written};
        }
        if (ds.lz77_state.is_last_block) break; /* :126-129 */

        BlockResult br = lz77_compress_block(slice, slice_n, ds.lz77_state, ds.input_buffer, ds.lz77_writer, flush);
        bytes_written += br.consumed;
        ds.bytes_written += br.consumed;
        if (br.status == NeedInput) return IoResult{true, false, bytes_written}; /* :145-150 */
        slice += br.consumed;
        slice_n -= br.consumed;

        bool last_block = ds.lz77_state.is_last_block;
        u64 current_block_input_bytes = ds.lz77_state.current_block_input_bytes;
        ds.bytes_written_control += current_block_input_bytes;
        u8 partial_bits = ds.encoder_state.writer.pending_bits();

        deflref_block_info bi;
        bi.n_lz = (u32)ds.lz77_writer.buffer.size();
        bi.in_bytes = current_block_input_bytes;
        bi.bfinal = last_block ? 1 : 0;
        bi.bit_start = ds.total_bits_now();

        DynamicBlockHeader header;
        BlockType res = gen_huffman_lengths(ds.lz77_writer.frequencies, ds.lz77_writer.distance_frequencies,
                                            current_block_input_bytes, partial_bits,
                                            ds.encoder_state.huffman_table.code_lengths,
                                            ds.encoder_state.huffman_table.distance_code_lengths,
                                            ds.length_buffers, &header);
        bi.btype = (u8)res;
        if (g_trace_on) g_trace.push_back(bi);

        switch (res) {
        case BT_Dynamic: /* :188-214 */
            ds.encoder_state.write_start_of_block(false, last_block);
            write_huffman_lengths(header, ds.encoder_state.huffman_table, ds.length_buffers.length_buf,
                                  ds.encoder_state.writer);
            ds.encoder_state.huffman_table.update_from_lengths();
            flush_to_bitstream(ds.lz77_writer.buffer, ds.encoder_state);
            break;
        case BT_Fixed: /* :215-229 */
            ds.encoder_state.write_start_of_block(true, last_block);
            ds.encoder_state.set_huffman_to_fixed();
            flush_to_bitstream(ds.lz77_writer.buffer, ds.encoder_state);
            break;
        case BT_Stored: { /* :230-246 */
            size_t position = br.position;
            size_t start_pos = position >= (size_t)current_block_input_bytes ? position - (size_t)current_block_input_bytes : 0;
            REF_ASSERT(position >= (size_t)current_block_input_bytes,
                       "Error! Trying to output a stored block with forgotten data!");
            if (br.slid_before_return) g_hazards++; /* Q13: position is a pre-slide coordinate */
            REF_ASSERT(start_pos <= position && position <= ds.input_buffer.buffer.size(),
                       "slice index out of range in stored arm (Q13)");
            write_stored_block(ds.input_buffer.buffer.data() + start_pos, position - start_pos,
                               ds.encoder_state.writer, flush == FlushFinish && last_block);
            break;
        }
        }
        ds.lz77_writer.clear();                       /* :250 */
        ds.lz77_state.current_block_input_bytes = 0;  /* reset_input_bytes :253 */

        if (br.status == Finished) { /* :256-273 */
            if (flush == FlushSync) {
                write_stored_block(nullptr, 0, ds.encoder_state.writer, false);
                ds.needs_flush = true;
            } else if (!ds.lz77_state.is_last_block) {
                ds.encoder_state.set_huffman_to_fixed();
                ds.encoder_state.write_start_of_block(true, true);
                ds.encoder_state.write_end_of_block();
            }
            break;
        }
    }

    ds.encoder_state.flush(); /* :277 */
    size_t output_buf_pos = ds.output_buf_pos;
    size_t avail = ds.output_buf().size() - output_buf_pos;
    size_t written_to_writer = ds.inner->write(ds.output_buf().data() + output_buf_pos, avail);
    if (written_to_writer < avail) {
        ds.output_buf_pos += written_to_writer;
    } else {
        ds.output_buf_pos = 0;
        ds.clear_output_buf();
        ds.needs_flush = false;
    }
    return IoResult{true, false, bytes_written};
}

/* writer.rs: compress_until_done :15-58 */
void compress_until_done(const u8* input, size_t input_n, DeflateState& ds, Flush flush_mode) {
    REF_ASSERT(flush_mode != FlushNone, "assert flush_mode != Flush::None");
    for (;;) {
        IoResult r = compress_data_dynamic_n(input, input_n, ds, flush_mode);
        if (r.ok && r.n == 0) {
            if (ds.output_buf().empty()) break;
            input_n = 0;
        } else if (r.ok) {
            if (r.n < input_n) {
                input += r.n;
                input_n -= r.n;
            } else {
                input_n = 0;
            }
        } else if (r.interrupted) {
            /* retry */
        }
    }
    REF_ASSERT(ds.bytes_written == ds.bytes_written_control, "debug_assert bytes_written == control");
}

/* zlib.rs :40-62 */
u8 add_fcheck(u8 cmf, u8 flg) {
    size_t rem = ((size_t)cmf * 256 + flg) % 31;
    flg = flg & 0xE0;
    return (u8)(flg + (31 - (u8)rem));
}
void get_zlib_header(u8 level_bits, u8 out[2]) {
    u8 cmf = 8 | (7 << 4);
    out[0] = cmf;
    out[1] = add_fcheck(cmf, level_bits);
}

/* Adler-32 per RFC 1950 (crate adler32 1.2.0 is not in the reference tree; call sites
 * checksum.rs:33-57) */
u32 adler32_update(u32 adler, const u8* data, size_t n) {
    u32 a = adler & 0xffff, b = adler >> 16;
    while (n) {
        size_t k = n < 5552 ? n : 5552;
        for (size_t i = 0; i < k; i++) {
            a += data[i];
            b += a;
        }
        a %= 65521;
        b %= 65521;
        data += k;
        n -= k;
    }
    return (b << 16) | a;
}

/* crate gzip-header 1.0 `Crc` (a wrapper of crc32fast): RFC 1952 section 8, bit by bit */
u32 crc32_update(u32 crc, const u8* data, size_t n) {
    crc = ~crc;
    for (size_t i = 0; i < n; i++) {
        crc ^= data[i];
        for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    }
    return ~crc;
}

} // namespace

/* ==========================================================================================
 * C interface
 * ======================================================================================== */
struct deflref_stream {
    deflref_opts opts;
    Sink sink;
    DeflateState* ds;
    u32 adler = 1;
    u32 crc = 0;      /* gzip: Crc::sum() */
    u32 amount = 0;   /* gzip: Crc::amt_as_u32() */
    std::vector<u8> gz_header;
    std::vector<u8> handed_out; /* what reset() returned */
    bool header_written = false;
    bool finished = false;
    void check_write_header() { /* writer.rs:226-232 (zlib), :360-368 (gzip) */
        if (opts.wrapper == 1 && !header_written) {
            u8 h[2];
            get_zlib_header(2 << 6, h);
            ds->output_buf().insert(ds->output_buf().end(), h, h + 2);
            header_written = true;
        }
        if (opts.wrapper == 2 && !gz_header.empty()) {
            ds->output_buf().insert(ds->output_buf().end(), gz_header.begin(), gz_header.end());
            gz_header.clear();
        }
    }
    void write_trailer() { /* writer.rs:235-245 (zlib), :408-425 (gzip) */
        if (opts.wrapper == 1) {
            u8 t[4] = {(u8)(adler >> 24), (u8)(adler >> 16), (u8)(adler >> 8), (u8)adler};
            sink.write_all(t, 4);
        }
        if (opts.wrapper == 2) {
            u8 t[8] = {(u8)crc, (u8)(crc >> 8), (u8)(crc >> 16), (u8)(crc >> 24),
                       (u8)amount, (u8)(amount >> 8), (u8)(amount >> 16), (u8)(amount >> 24)};
            sink.write_all(t, 8);
        }
    }
};

#define GUARD_BEGIN try {
#define GUARD_END                        \
    }                                    \
    catch (const RefPanic& p) {          \
        g_last_panic = p.msg;            \
        return DEFLREF_E_REF_PANIC;      \
    }

extern "C" {

void deflref_preset(int level, deflref_opts* out) {
    /* compression_options.rs:126-196 */
    deflref_opts o = {128, 32, 1, 0};
    switch (level) {
    case 0: o = {1, 0, 0, 0}; break;      /* fast :141 */
    case 1: o = {128, 32, 1, 0}; break;   /* default :67-72 */
    case 2: o = {1768, 128, 1, 0}; break; /* high :126-133 */
    case 3: o = {0, 0, 1, 0}; break;      /* rle :171-178 */
    case 4: o = {0, 0, 0, 0}; break;      /* huffman_only :155-162 */
    }
    *out = o;
}

size_t deflref_bound(size_t in_len) { return in_len + 5 * (in_len / 31744 + in_len / 32767 + 2) + 64; }

const char* deflref_last_panic(void) { return g_last_panic.c_str(); }
int deflref_last_hazards(void) { return g_hazards; }

int deflref_encode(const uint8_t* in, size_t in_len, const deflref_opts* opts, uint8_t* out, size_t out_cap,
                   size_t* out_len) {
    if (!opts || !out_len || (!in && in_len)) return DEFLREF_E_ARG;
    g_hazards = 0;
    g_trace.clear();
    g_trace_on = true;
    GUARD_BEGIN
    Sink sink;
    u32 adler = 1;
    if (opts->wrapper == 1) { /* lib.rs:182-198 */
        u8 h[2];
        get_zlib_header(2 << 6, h);
        sink.write_all(h, 2);
        adler = adler32_update(1, in, in_len);
    }
    {
        DeflateState ds(*opts, &sink); /* lib.rs:110-122 */
        compress_until_done(in, in_len, ds, FlushFinish);
    }
    if (opts->wrapper == 1) {
        u8 t[4] = {(u8)(adler >> 24), (u8)(adler >> 16), (u8)(adler >> 8), (u8)adler};
        sink.write_all(t, 4);
    }
    g_trace_on = false;
    *out_len = sink.data.size();
    if (sink.data.size() > out_cap) return DEFLREF_E_OUT_TOO_SMALL;
    memcpy(out, sink.data.data(), sink.data.size());
    return DEFLREF_OK;
    GUARD_END
}

uint32_t deflref_crc32(uint32_t crc, const uint8_t* data, size_t n) { return crc32_update(crc, data, n); }

int deflref_encode_gzip(const uint8_t* in, size_t in_len, const deflref_opts* opts, const uint8_t* hdr,
                        size_t hdr_len, uint8_t* out, size_t out_cap, size_t* out_len) {
    if (!opts || !out_len || (!in && in_len) || (!hdr && hdr_len)) return DEFLREF_E_ARG;
    g_hazards = 0;
    g_trace.clear();
    g_trace_on = true;
    GUARD_BEGIN
    Sink sink;
    sink.write_all(hdr, hdr_len); /* lib.rs:250-253 */
    {
        deflref_opts o = *opts;
        o.wrapper = 0;
        DeflateState ds(o, &sink); /* :254-256 */
        compress_until_done(in, in_len, ds, FlushFinish);
    }
    u32 crc = crc32_update(0, in, in_len); /* :258-259 */
    u32 amt = (u32)in_len;
    u8 t[8] = {(u8)crc, (u8)(crc >> 8), (u8)(crc >> 16), (u8)(crc >> 24),
               (u8)amt, (u8)(amt >> 8), (u8)(amt >> 16), (u8)(amt >> 24)};
    sink.write_all(t, 8); /* :261-266 */
    g_trace_on = false;
    *out_len = sink.data.size();
    if (sink.data.size() > out_cap) return DEFLREF_E_OUT_TOO_SMALL;
    memcpy(out, sink.data.data(), sink.data.size());
    return DEFLREF_OK;
    GUARD_END
}

size_t deflref_trace_blocks(deflref_block_info* out, size_t cap) {
    size_t n = g_trace.size();
    for (size_t i = 0; i < n && i < cap; i++) out[i] = g_trace[i];
    return n;
}

deflref_stream* deflref_stream_new(const deflref_opts* opts) {
    deflref_stream* s = new deflref_stream();
    s->opts = *opts;
    s->ds = new DeflateState(*opts, &s->sink);
    return s;
}

int deflref_stream_write(deflref_stream* s, const uint8_t* data, size_t n) {
    GUARD_BEGIN
    g_trace_on = false;
    /* io::Write::write_all over {Deflate,Zlib}Encoder::write (writer.rs:124-127, 254-267) */
    while (n > 0) {
        s->check_write_header();
        IoResult r = compress_data_dynamic_n(data, n, *s->ds, s->ds->flush_mode);
        if (r.ok) {
            size_t k = r.n == 0 ? n : r.n; /* writer.rs:258-265: Ok(0) checksums the whole buf */
            if (s->opts.wrapper == 1) s->adler = adler32_update(s->adler, data, k);
            if (s->opts.wrapper == 2) { /* writer.rs:436-444 */
                s->crc = crc32_update(s->crc, data, k);
                s->amount += (u32)k;
            }
            if (r.n == 0) return DEFLREF_E_REF_PANIC; /* write_all -> WriteZero error */
            data += r.n;
            n -= r.n;
        } else if (!r.interrupted) {
            return DEFLREF_E_ARG;
        }
    }
    return DEFLREF_OK;
    GUARD_END
}

int deflref_stream_flush(deflref_stream* s) {
    GUARD_BEGIN
    compress_until_done(nullptr, 0, *s->ds, FlushSync); /* writer.rs:134-137 */
    return DEFLREF_OK;
    GUARD_END
}

int deflref_stream_finish(deflref_stream* s) {
    GUARD_BEGIN
    s->check_write_header(); /* writer.rs:201-205 */
    compress_until_done(nullptr, 0, *s->ds, FlushFinish);
    s->write_trailer();
    s->finished = true;
    return DEFLREF_OK;
    GUARD_END
}

int deflref_stream_gzip_header(deflref_stream* s, const uint8_t* hdr, size_t hdr_len) {
    if (!s || s->opts.wrapper != 2 || (!hdr && hdr_len)) return DEFLREF_E_ARG;
    s->gz_header.assign(hdr, hdr + hdr_len);
    return DEFLREF_OK;
}

int deflref_stream_reset(deflref_stream* s, const uint8_t** data, size_t* n) {
    GUARD_BEGIN
    /* output_all + trailer, then DeflateState::reset with a new Vec (writer.rs:110-117, 216-223,
     * 383-402; deflate_state.rs:133-152).  The gzip header of the next stream is set again by the
     * caller (reset uses a blank GzBuilder, reset_with_builder the given one). */
    s->check_write_header();
    compress_until_done(nullptr, 0, *s->ds, FlushFinish);
    s->ds->reset(&s->sink);
    s->write_trailer();
    s->handed_out.swap(s->sink.data);
    s->sink.data.clear();
    s->adler = 1;
    s->crc = 0;
    s->amount = 0;
    s->header_written = false;
    if (data) *data = s->handed_out.data();
    if (n) *n = s->handed_out.size();
    return DEFLREF_OK;
    GUARD_END
}

size_t deflref_stream_output(deflref_stream* s, const uint8_t** data) {
    *data = s->sink.data.data();
    return s->sink.data.size();
}
uint32_t deflref_stream_checksum(deflref_stream* s) { return s->opts.wrapper == 2 ? s->crc : s->adler; }
void deflref_stream_free(deflref_stream* s) {
    if (!s) return;
    delete s->ds;
    delete s;
}

static u32 pack_lz(const LZValue& v) { return (u32)v.litlen | ((u32)v.distance << 16); }

long deflref_lz77(const uint8_t* in, size_t n, uint16_t max_hash_checks, uint16_t lazy_if_less_than,
                  int matching_type, uint32_t* out, size_t cap) {
    try {
        /* lz77_compress_conf lz77.rs:879-910 (TestStruct::with_config :833-845) */
        LZ77State state(max_hash_checks, lazy_if_less_than, matching_type ? Lazy : Greedy);
        InputBuffer buffer;
        DynamicWriter writer;
        std::vector<LZValue> all;
        const u8* slice = in;
        size_t slice_n = n;
        while (!state.is_last_block) {
            BlockResult br = lz77_compress_block(slice, slice_n, state, buffer, writer, FlushFinish, true);
            slice += br.consumed;
            slice_n -= br.consumed;
            all.insert(all.end(), writer.buffer.begin(), writer.buffer.end());
            writer.clear();
        }
        for (size_t i = 0; i < all.size() && i < cap; i++) out[i] = pack_lz(all[i]);
        return (long)all.size();
    } catch (const RefPanic& p) {
        g_last_panic = p.msg;
        return DEFLREF_E_REF_PANIC;
    }
}

int deflref_compress_fixed(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len) {
    GUARD_BEGIN
    /* compress_data_fixed compress.rs:44-57 with lz77_compress (1768/128/Lazy) lz77.rs:865-872 */
    std::vector<u32> lz(n + 16);
    long cnt = deflref_lz77(in, n, 1768, 128, 1, lz.data(), lz.size());
    if (cnt < 0) return (int)cnt;
    EncoderState st;
    st.huffman_table.set_to_fixed(); /* EncoderState::fixed encoder_state.rs:39-45 */
    st.write_start_of_block(true, true);
    std::vector<LZValue> buf;
    for (long i = 0; i < cnt; i++) buf.push_back(LZValue{(u8)(lz[i] & 0xff), (u16)(lz[i] >> 16)});
    flush_to_bitstream(buf, st);
    st.flush();
    *out_len = st.writer.w.size();
    if (*out_len > cap) return DEFLREF_E_OUT_TOO_SMALL;
    memcpy(out, st.writer.w.data(), *out_len);
    return DEFLREF_OK;
    GUARD_END
}

void deflref_longest_match(const uint8_t* data, size_t n, size_t fill_n, size_t position, size_t prev_length,
                           uint16_t max_hash_checks, uint32_t* len, uint32_t* dist) {
    /* filled_hash_table(&data[..fill_n]) chained_hash_table.rs:222-230 */
    ChainedHashTable t;
    t.current_hash = update_hash(t.current_hash, data[0]);
    t.current_hash = update_hash(t.current_hash, data[1]);
    for (size_t i = 2; i < fill_n; i++) t.add_hash_value(i - 2, data[i]);
    size_t l = 0, d = 0;
    try {
        longest_match(data, n, t, position, prev_length, max_hash_checks, &l, &d);
    } catch (const RefPanic&) {
        l = d = 0xffffffff;
    }
    *len = (u32)l;
    *dist = (u32)d;
}

size_t deflref_get_match_length(const uint8_t* data, size_t n, size_t cur, size_t check) {
    return get_match_length(data, n, cur, check);
}

void deflref_huffman_lengths(const uint16_t* freqs, size_t n, size_t max_len, uint8_t* lens) {
    std::vector<Node> leaves; /* gen_lengths length_encode.rs:331-336 */
    in_place_lengths(freqs, n, max_len, leaves, lens, n);
}

long deflref_encode_lengths(const uint8_t* lens, size_t n, uint16_t* out, size_t cap, uint16_t freqs[19]) {
    try {
        std::vector<EncodedLength> enc;
        for (int i = 0; i < 19; i++) freqs[i] = 0;
        encode_lengths_m(lens, n, enc, freqs);
        for (size_t i = 0; i < enc.size() && i < cap; i++) out[i] = (u16)((enc[i].kind << 8) | enc[i].v);
        return (long)enc.size();
    } catch (const RefPanic& p) {
        g_last_panic = p.msg;
        return DEFLREF_E_REF_PANIC;
    }
}

uint16_t deflref_reverse_bits(uint16_t n, uint8_t length) { return reverse_bits(n, length); }

long deflref_lsb_write(const uint16_t* v, const uint8_t* nbits, size_t n, uint8_t* out, size_t cap) {
    LsbWriter w;
    for (size_t i = 0; i < n; i++) w.write_bits(v[i], nbits[i]);
    w.flush_raw();
    for (size_t i = 0; i < w.w.size() && i < cap; i++) out[i] = w.w[i];
    return (long)w.w.size();
}

uint64_t deflref_stored_padding(uint8_t pending_bits) {
    try {
        return stored_padding(pending_bits);
    } catch (const RefPanic&) {
        return ~0ull;
    }
}
size_t deflref_get_length_code(uint16_t length) { return get_length_code(length); }
uint8_t deflref_get_distance_code(uint16_t distance) { return get_distance_code(distance); }
void deflref_length_extra(uint8_t stored_length, uint16_t* code, uint8_t* nbits, uint16_t* value) {
    ExtraBits e = get_length_code_and_extra_bits(stored_length);
    *code = e.code_number;
    *nbits = e.num_bits;
    *value = e.value;
}
void deflref_distance_extra(uint16_t distance, uint16_t* code, uint8_t* nbits, uint16_t* value) {
    ExtraBits e = get_distance_code_and_extra_bits(distance);
    *code = e.code_number;
    *nbits = e.num_bits;
    *value = e.value;
}
void deflref_fixed_code(int is_distance, unsigned symbol, uint16_t* code, uint8_t* length) {
    HuffmanTable t;
    t.set_to_fixed();
    if (is_distance) {
        *code = t.distance_codes[symbol];
        *length = t.distance_code_lengths[symbol];
    } else {
        *code = t.codes[symbol];
        *length = t.code_lengths[symbol];
    }
}
void deflref_zlib_header(uint8_t level_bits, uint8_t out[2]) { get_zlib_header(level_bits, out); }
uint32_t deflref_adler32(const uint8_t* data, size_t n) { return adler32_update(1, data, n); }

long deflref_rle_chunk(const uint8_t* data, size_t n, size_t start, size_t end, uint32_t* out, size_t cap,
                       size_t* overlap) {
    try {
        DynamicWriter w;
        ChunkResult cr = process_chunk_greedy_rle(data, n, start, end, w);
        *overlap = cr.overlap;
        for (size_t i = 0; i < w.buffer.size() && i < cap; i++) out[i] = pack_lz(w.buffer[i]);
        return (long)w.buffer.size();
    } catch (const RefPanic& p) {
        g_last_panic = p.msg;
        return DEFLREF_E_REF_PANIC;
    }
}

} /* extern "C" */
