/*
 * deflref.h -- C interface of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a line-by-line C++ restatement of the DEFLATE encode hot path of
 * image-rs/deflate-rs v1.0.0 (reference tree: /root/reference, Rust, cannot be built in
 * this image: no rustc/cargo).  Every function in deflref.cpp cites the reference
 * file:line it follows.
 *
 * Pinning status: the restatement is pinned against every known-answer test the
 * reference's own test-suite holds for this path (SURVEY.md Appendix B group K; see
 * tests/test_oracle_kat.py) and against inflate round trips through system zlib for all
 * reference fixtures.  The reference holds NO golden compressed streams, so whole-stream
 * bytes are pinned transitively (literal restatement + KATs + invariants), not against
 * output of the real crate.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call
 * this library.  The product path (libmi355deflate.so) never does.
 */
#ifndef DEFLREF_H
#define DEFLREF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/compression_options.rs:78-120 (CompressionOptions) + src/lz77.rs:27-37 (MatchingType) */
typedef struct {
    uint16_t max_hash_checks;
    uint16_t lazy_if_less_than;
    uint8_t matching_type; /* 0 = Greedy, 1 = Lazy */
    uint8_t wrapper;       /* 0 = raw deflate, 1 = zlib (78 9C + Adler-32 BE), 2 = gzip (header given by the
                              caller + CRC-32 LE + length mod 2^32 LE; the streaming form only) */
} deflref_opts;

/* error codes */
#define DEFLREF_OK 0
#define DEFLREF_E_ARG (-1)
#define DEFLREF_E_OUT_TOO_SMALL (-2)
#define DEFLREF_E_REF_PANIC (-100) /* the reference itself would panic on this input */

/* Compression::{Fast,Default,Best} and the two special profiles
 * (src/compression_options.rs:126-196).  level: 0 fast, 1 default, 2 best, 3 rle,
 * 4 huffman_only. */
void deflref_preset(int level, deflref_opts* out);

/* deflate_bytes_conf / deflate_bytes_zlib_conf (src/lib.rs:137-198). */
int deflref_encode(const uint8_t* in, size_t in_len, const deflref_opts* opts, uint8_t* out,
                   size_t out_cap, size_t* out_len);

/* deflate_bytes_gzip_conf (src/lib.rs:242-267, feature "gzip"): `hdr` = what GzBuilder::into_header()
 * returned (crate gzip-header 1.0, not in the tree: the header is taken as bytes), then the raw stream,
 * then Crc::sum() and Crc::amt_as_u32() little endian. */
int deflref_encode_gzip(const uint8_t* in, size_t in_len, const deflref_opts* opts, const uint8_t* hdr,
                        size_t hdr_len, uint8_t* out, size_t out_cap, size_t* out_len);
/* Crc::update / Crc::sum of gzip-header 1.0 = CRC-32 of RFC 1952 section 8 (IEEE 802.3, reflected). */
uint32_t deflref_crc32(uint32_t crc, const uint8_t* data, size_t n);

/* Worst-case output size for a given input size (stored blocks + framing). */
size_t deflref_bound(size_t in_len);

/* Last panic message (thread local), for DEFLREF_E_REF_PANIC. */
const char* deflref_last_panic(void);

/* Number of times the Q13 hazard (stored block emitted after a BufferFull-slide; SURVEY.md
 * A.4 Q13) was hit during the last encode on this thread. */
int deflref_last_hazards(void);

/* ---- streaming encoders: write::{DeflateEncoder,ZlibEncoder}<Vec<u8>> (src/writer.rs) ---- */
typedef struct deflref_stream deflref_stream;
deflref_stream* deflref_stream_new(const deflref_opts* opts);
/* io::Write::write_all */
int deflref_stream_write(deflref_stream* s, const uint8_t* data, size_t n);
/* io::Write::flush  (Flush::Sync) */
int deflref_stream_flush(deflref_stream* s);
/* finish(): after this only _output/_free are valid */
int deflref_stream_finish(deflref_stream* s);
/* bytes the wrapped Vec<u8> sink holds so far */
size_t deflref_stream_output(deflref_stream* s, const uint8_t** data);
uint32_t deflref_stream_checksum(deflref_stream* s);
/* write::gzip::GzEncoder::from_builder (src/writer.rs:346-358): the header bytes of a wrapper-2 stream */
int deflref_stream_gzip_header(deflref_stream* s, const uint8_t* hdr, size_t hdr_len);
/* reset(&mut self, W) (src/writer.rs:110-117, 216-223, 383-402): finishes the stream, hands its bytes
 * out (valid until the next reset / free) and starts a new one with the same options */
int deflref_stream_reset(deflref_stream* s, const uint8_t** data, size_t* n);
void deflref_stream_free(deflref_stream* s);

/* ---- block trace of the last deflref_encode on this thread (for diffing intermediates) ---- */
typedef struct {
    uint8_t btype;       /* 0 stored, 1 fixed, 2 dynamic */
    uint8_t bfinal;
    uint32_t n_lz;       /* LZ values in the block */
    uint64_t in_bytes;   /* current_block_input_bytes */
    uint64_t bit_start;  /* bit offset of the block's first header bit in the raw stream */
} deflref_block_info;
size_t deflref_trace_blocks(deflref_block_info* out, size_t cap);

/* ---- test hooks used by the KAT tests (mirror the reference's #[cfg(test)] helpers) ---- */
/* lz77::lz77_compress_conf (src/lz77.rs:879-910): out[i] = litlen | distance<<16
 * (distance 0 = literal, litlen = length-3 for matches).  Returns count or <0. */
long deflref_lz77(const uint8_t* in, size_t n, uint16_t max_hash_checks,
                  uint16_t lazy_if_less_than, int matching_type, uint32_t* out, size_t cap);
/* compress::compress_data_fixed (src/compress.rs:44-57) */
int deflref_compress_fixed(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len);
/* matching::longest_match(data, filled_hash_table(&data[..fill_n]), position, ...)
 * (src/matching.rs:87-166, src/chained_hash_table.rs:222-230) */
void deflref_longest_match(const uint8_t* data, size_t n, size_t fill_n, size_t position, size_t prev_length,
                           uint16_t max_hash_checks, uint32_t* len, uint32_t* dist);
size_t deflref_get_match_length(const uint8_t* data, size_t n, size_t cur, size_t check);
/* length_encode::huffman_lengths_from_frequency (src/length_encode.rs:157-160) */
void deflref_huffman_lengths(const uint16_t* freqs, size_t n, size_t max_len, uint8_t* lens);
/* length_encode::encode_lengths (src/length_encode.rs:63-73): out[i] = kind<<8 | value,
 * kind 0 Length, 1 CopyPrevious, 2 RepeatZero3Bits, 3 RepeatZero7Bits */
long deflref_encode_lengths(const uint8_t* lens, size_t n, uint16_t* out, size_t cap,
                            uint16_t freqs[19]);
uint16_t deflref_reverse_bits(uint16_t n, uint8_t length);
/* LsbWriter: write all (v[i], nbits[i]) then flush_raw (src/bitstream.rs:76-106) */
long deflref_lsb_write(const uint16_t* v, const uint8_t* nbits, size_t n, uint8_t* out,
                       size_t cap);
uint64_t deflref_stored_padding(uint8_t pending_bits);
size_t deflref_get_length_code(uint16_t length);
uint8_t deflref_get_distance_code(uint16_t distance);
/* get_length_code_and_extra_bits / get_distance_code_and_extra_bits: code_number, num_bits,
 * value (src/huffman_table.rs:150-194) */
void deflref_length_extra(uint8_t stored_length, uint16_t* code, uint8_t* nbits, uint16_t* value);
void deflref_distance_extra(uint16_t distance, uint16_t* code, uint8_t* nbits, uint16_t* value);
/* HuffmanTable::fixed_table(): code/length of ll symbol or distance symbol */
void deflref_fixed_code(int is_distance, unsigned symbol, uint16_t* code, uint8_t* length);
/* zlib::get_zlib_header(level_bits) (src/zlib.rs:59-62) */
void deflref_zlib_header(uint8_t level_bits, uint8_t out[2]);
uint32_t deflref_adler32(const uint8_t* data, size_t n);
/* rle::process_chunk_greedy_rle over data[0..n) into a fresh writer; same packing as
 * deflref_lz77; *overlap receives the returned overlap (src/rle.rs:23-71) */
long deflref_rle_chunk(const uint8_t* data, size_t n, size_t start, size_t end, uint32_t* out,
                       size_t cap, size_t* overlap);

#ifdef __cplusplus
}
#endif
#endif
