/*
 * mi355_deflate.h -- C ABI of the MI355X-native DEFLATE encode path (libmi355deflate.so).
 *
 * Drop-in boundary for the hot path of image-rs/deflate-rs v1.0.0.  The reference has no FFI
 * seam (src/lib.rs:50 forbids unsafe code); the natural seam is its block driver
 *     compress_data_dynamic_n(input, &mut DeflateState<W>, Flush)      src/compress.rs:80-84
 * as called by compress_until_done (src/writer.rs:15-23) from compress_data_dynamic
 * (src/lib.rs:110-122) and by the Write impls (src/writer.rs:124-127, 254-267).  Each entry
 * point below names the reference item it replaces.  INTEGRATION.md shows the Rust shim a
 * maintainer would add on the reference side.
 *
 * All functions are extern "C", take plain pointers and sizes, never throw, never abort, and
 * return 0 on success or a negative MI355_E_* code.  There is NO CPU fallback: every encode
 * runs the HIP kernels and fails with MI355_E_HIP if no gfx950 device is usable.
 */
#ifndef MI355_DEFLATE_H
#define MI355_DEFLATE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_DEFLATE_VERSION 100 /* 0.1.0 */

#define MI355_OK 0
#define MI355_E_ARG (-1)           /* null pointer / bad option */
#define MI355_E_OUT_TOO_SMALL (-2) /* *out_len holds the size needed */
#define MI355_E_HIP (-3)           /* HIP runtime error; see mi355_deflate_last_error */
#define MI355_E_UNSUPPORTED (-4)   /* lazy_if_less_than < 3 with Lazy matching -- nothing else: one-shot
                                      calls, sync-flush chunks and streams, flushed anywhere or not at all, take any length
                                      in ranges, in bounded memory */
#define MI355_E_REF_PANIC (-5)     /* the reference itself panics on this input (A.4 Q13, slice out
                                      of range) and MI355_COMPAT_Q13 was requested */
#define MI355_E_STATE (-6)         /* stream used after finish, or after a range of it failed; context busy with a shard */

/* CompressionOptions (src/compression_options.rs:78-120) + MatchingType (src/lz77.rs:27-37).
 * `special` has only the Normal variant reachable from the public API and is omitted. */
typedef struct {
    uint16_t max_hash_checks;   /* compression_options.rs:84 */
    uint16_t lazy_if_less_than; /* :101; clamped to 32768 like deflate_state.rs:105 */
    uint8_t matching_type;      /* 0 = MatchingType::Greedy, 1 = MatchingType::Lazy */
    uint8_t wrapper;            /* 0 = raw deflate (deflate_bytes_conf, DeflateEncoder),
                                   1 = zlib: 78 9C + Adler-32 BE (deflate_bytes_zlib_conf,
                                   ZlibEncoder; src/lib.rs:182-198, src/zlib.rs:59-62),
                                   2 = gzip: header + CRC-32 LE + length mod 2^32 LE (feature "gzip":
                                   deflate_bytes_gzip_conf src/lib.rs:242-267, GzEncoder
                                   src/writer.rs:293-467); the header is GzBuilder::new()'s unless
                                   one of the _gzip entry points supplies it */
    uint8_t compat;             /* MI355_COMPAT_* bits */
    uint8_t flush;              /* MI355_FLUSH_FINISH (0) or MI355_FLUSH_SYNC (1) */
} mi355_deflate_opts;

/* How the stream ends (src/compress.rs:18-30 Flush).  FINISH: the last block carries BFINAL
 * (compress_until_done(.., Flush::Finish), what deflate_bytes_conf and finish() do).  SYNC: what a
 * fresh encoder has written after write_all(input) + flush() (src/writer.rs:134-137): every block
 * non-final, then the empty stored block 00 00 FF FF (src/compress.rs:256-261).  The output is byte
 * aligned, so SYNC chunks followed by one FINISH chunk concatenate into one valid deflate stream --
 * the chunk-exact ("P2") stitch used to shard one input over several GPUs.  With wrapper = 1 the
 * zlib trailer is only written for FINISH. */
#define MI355_FLUSH_FINISH 0
#define MI355_FLUSH_SYNC 1

/* Bug-for-bug mode for SURVEY A.4 Q13: when a Stored block ends on a match that crosses the
 * end of a non-first window, the reference reads the stored bytes 32768 too far ahead
 * (src/lz77.rs:679-695 + src/compress.rs:230-245) and emits a stream that does NOT inflate to
 * the input.  Default (bit clear): emit the block's real bytes (valid stream; differs from the
 * reference only inside that block).  Bit set: reproduce the reference's bytes exactly. */
#define MI355_COMPAT_Q13 1u

/* Compression::{Fast, Default, Best} -> CompressionOptions (compression_options.rs:188-196)
 * and the two named profiles rle() :171-178, huffman_only() :155-162. */
#define MI355_LEVEL_FAST 0
#define MI355_LEVEL_DEFAULT 1
#define MI355_LEVEL_BEST 2
#define MI355_LEVEL_RLE 3
#define MI355_LEVEL_HUFFMAN_ONLY 4
int mi355_deflate_preset(int level, mi355_deflate_opts* out);

int mi355_deflate_version(void);

/* Upper bound of the output size for in_len input bytes: every block is at most its stored form
 * (src/huffman_lengths.rs:269-286 picks the cheapest, src/stored_block.rs:13-40), plus the framing of any
 * wrapper with the blank gzip header.  mi355_deflate_bound_ex is the exact requirement of the encode
 * entry points for a given wrapper (0 raw, 1 zlib, 2 gzip with hdr_len header bytes) and number of sync
 * flush points (each can add a block header and the marker 00 00 FF FF, src/compress.rs:256-261); a
 * buffer of mi355_deflate_bound(in_len) bytes is always enough for a one-shot call. */
size_t mi355_deflate_bound(size_t in_len);
size_t mi355_deflate_bound_ex(size_t in_len, int wrapper, size_t hdr_len, size_t n_flush);

/* A context owns one HIP device, its workspace and timing events.  Not thread safe; use one
 * context per thread (the reference's encoders are likewise single-owner: DeflateState,
 * src/deflate_state.rs:66-97).  Where an entry point accepts ctx == NULL it works on a process-wide
 * default context on device 0 and holds that context's lock for the whole call, so NULL-context calls
 * from several threads are safe (and serialised).
 * Creating a context runs one short self-test kernel on the device (about 0.1 ms): the hash sort takes its ranks
 * from returning LDS atomics when -- and only when -- the device serves the lanes that hit one LDS address in lane
 * order (MI355X does); otherwise it ranks with ballots.  The output is the same either way, and every encode
 * checks the order of the sorted buckets it walks (MI355_CFG_SORT_RANKS below).
 * A context that holds a sharded encode (between mi355_shard_begin and mi355_shard_end) refuses other encodes
 * with MI355_E_STATE: the shard's tokens and tables live in the context's workspace. */
typedef struct mi355_deflate_ctx mi355_deflate_ctx;
int mi355_deflate_ctx_create(int device, mi355_deflate_ctx** out);
void mi355_deflate_ctx_destroy(mi355_deflate_ctx* ctx);
const char* mi355_deflate_last_error(mi355_deflate_ctx* ctx);

/* Tuning of one context (ctx may be NULL: the default context).  No reference item: the reference has no tunables
 * beyond CompressionOptions; these set the memory / throughput trade of the GPU path, never the bytes it produces.
 *   MI355_CFG_RANGE_BYTES  bytes per range of a long input or a stream (default 512 MiB; at least 16 MiB,
 *                          at most 3 GiB, rounded down to a multiple of 32768).  Device memory of a long encode is two
 *                          workspaces of about 20 B per byte of a range, host memory of a stream one range + 16 MiB.
 *   MI355_CFG_LONG_FROM    one-shot inputs of at least this many bytes are walked in ranges (default 1 GiB + 1; at most
 *                          4 GiB - 64 KiB, what a single pass takes)
 *   MI355_CFG_SORT_RANKS   where the hash sort takes its ranks from: 1 = returning LDS atomics (the default on a device
 *                          that passed the self-test of mi355_deflate_ctx_create; MI355_E_UNSUPPORTED on one that did
 *                          not), 0 = ballots (any device; about 40 % more time in the sort).  Whatever the setting, the
 *                          match kernel checks every hash bucket's order on the data it walks and an encode that finds
 *                          one out of order is done again with ballot ranks, which the context then keeps.
 *   MI355_CFG_HOST_STREAMING  how a host-buffer call of 16 MiB or more (mi355_deflate_encode and its zlib / gzip
 *                          forms) hands its bytes back: 1 (default) = piece by piece -- the input arrives in pieces, each
 *                          is encoded as it arrives (its parse, block and pack stages beside the match stage of the next
 *                          one) and the finished bytes leave on the copy engine while the later pieces are worked on;
 *                          0 = one copy-out after the last kernel; 2 = pieces with their stages behind each other
 *                          (a measuring aid).  Same bytes every way; data the piecewise form does not take (a hash re-warm
 *                          in the first window, long periodic data) is encoded the other way without the caller noticing.
 *                          (Measuring aids, read once per process: the environment variables MI355_HOST_PLAN="1,2,3,3" -- rounds of
 *                          256 windows per piece -- and MI355_HOST_FIRST=<windows of the first piece, 0 = a whole round>.)
 *   MI355_CFG_HOST_BOUNCE  what a host-buffer call does with PAGEABLE caller memory (a plain &[u8] / Vec<u8>: what
 *                          deflate_bytes hands over, src/lib.rs:137-147) of 4 MiB or more: 1 (default) = the context's host
 *                          threads copy it through two page-locked rings of 64 MiB -- the input's pieces arrive while the
 *                          first ones are worked on, the finished bytes leave piece by piece, as for a caller that page-locked
 *                          its buffers; 0 = the runtime's own copies (one thread, in series with the kernels).
 *                          Page-locked buffers (hipHostMalloc / hipHostRegister) never take the threads.  The same threads
 *                          carry the ranges of a call of more than 1 GiB and of mi355_deflate_encode_multi.
 *   MI355_CFG_HOST_THREADS  how many such threads the context starts when the first pageable call comes (1..16; default 8 on
 *                          a host of 32 hardware threads or more, else 4 or 2; the environment variable MI355_HOST_THREADS
 *                          sets the default).  They stay on the memory node of the thread that made them, poll for a
 *                          millisecond after a call and sleep between calls.  MI355_E_STATE once they run.
 *   MI355_CFG_MULTI_STITCH  (of rank 0's context of a mi355_multi handle: mi355_multi_ctx(m, 0)) how the packed ranges of
 *                          mi355_deflate_encode_multi_device reach rank 0's device: 0 (default) = peer copies
 *                          (hipMemcpyPeerAsync: xGMI between the GPUs of a node), 1 = RCCL -- one ncclSend per rank, the
 *                          matching ncclRecv posted by rank 0's thread straight into the caller's buffer; librccl.so is
 *                          looked up when the first such call is made (MI355_E_UNSUPPORTED if it is not there).
 *   MI355_CFG_STEPS_IN_EMIT  where the parser's restart steps (lz77.rs:305-547 seen from a position) are worked out: 1 (default) =
 *                          by the kernel that writes the tokens, from the match table, where that is possible (the lazy and
 *                          greedy levels without a quarter-budget table, input without flush points); 0 = always by a kernel
 *                          of their own that leaves them in device memory (what the exact path search, `Best` and flushed
 *                          streams use anyway).  Same bytes either way: a testing and measuring aid.
 *   MI355_CFG_STAGE_CLOCKS  the per-stage clocks in mi355_deflate_info -- stage_ms, match_ms --: 0 (default) = never, 1 = for every
 *                          call, 2 = for calls of 32 MiB or more.  They are events between the kernels of a call, each 5.7 us of
 *                          idle queue -- a tenth of a 167 KB call, 0.7 % of a 100 MB one; without them stage_ms and match_ms read 0
 *                          and total_ms is the host's clock from the call's first launch to the return of its last wait.
 *                          (The environment variable MI355_STAGE_CLOCKS sets the default: a measuring aid.) */
#define MI355_CFG_RANGE_BYTES 1
#define MI355_CFG_LONG_FROM 2
#define MI355_CFG_SORT_RANKS 3
#define MI355_CFG_HOST_STREAMING 4
#define MI355_CFG_MULTI_STITCH 5
#define MI355_CFG_STEPS_IN_EMIT 6
#define MI355_CFG_HOST_BOUNCE 7
#define MI355_CFG_HOST_THREADS 8
#define MI355_CFG_STAGE_CLOCKS 9
int mi355_deflate_ctx_config(mi355_deflate_ctx* ctx, int key, uint64_t value);

/* deflate_bytes_conf / deflate_bytes_zlib_conf (src/lib.rs:137-147, 182-198): host buffers
 * in, host buffer out.  ctx may be NULL (the default context, see above).  An input of 16 MiB or more is
 * copied in pieces and worked on while the later ones are still on the bus, and its finished bytes leave piece by piece:
 * for page-locked buffers (hipHostMalloc / hipHostRegister) on the copy engines directly, for pageable memory -- what a
 * drop-in caller has -- through the context's host threads and their page-locked slots (MI355_CFG_HOST_BOUNCE).
 * Any in_len is taken, like src/lib.rs:137-147: an input of more than 1 GiB is walked as consecutive ranges of
 * 512 MiB (csrc/deflate_long.inc -- the phases of the sharded encode below, one range after the other on this
 * GPU; the same bytes as a single pass), which also bounds the device memory of a call: two workspaces of
 * about 20 B per byte of a range. */
int mi355_deflate_encode(mi355_deflate_ctx* ctx, const uint8_t* in, size_t in_len, const mi355_deflate_opts* opts,
                         uint8_t* out, size_t out_cap, size_t* out_len);

/* Allocate now what encodes of up to in_len bytes need (the workspace is about 20 bytes per input byte
 * and only ever grows; with host_api != 0 also the device staging buffers of mi355_deflate_encode), so
 * that the first call does not pay for it.  No reference item: the reference's encoders allocate their
 * 330 KiB in ::new (src/deflate_state.rs:82-119). */
int mi355_deflate_ctx_reserve(mi355_deflate_ctx* ctx, size_t in_len, int host_api);

/* Same computation with input and output resident in device memory (no PCIe in the path):
 * compress_data_dynamic + compress_until_done(.., Flush::Finish) (src/lib.rs:110-122,
 * src/writer.rs:15-58) over d_in[0..in_len).  d_out must be 4-byte aligned and hold
 * mi355_deflate_bound(in_len) bytes (exactly: mi355_deflate_bound_ex(in_len, opts->wrapper, 10, 0); a
 * smaller buffer is refused with MI355_E_OUT_TOO_SMALL and the size in *out_len).  `hip_stream` is a
 * hipStream_t (NULL = the context's own stream); the
 * call returns after the stream has drained, with *out_len set.  Always raw deflate unless
 * opts->wrapper == 1, in which case the 2-byte header and 4-byte trailer are written too.
 * Any in_len (more than 1 GiB: in ranges, see mi355_deflate_encode; the work then runs on the context's streams). */
int mi355_deflate_encode_device(mi355_deflate_ctx* ctx, const void* d_in, size_t in_len, const mi355_deflate_opts* opts,
                                void* d_out, size_t out_cap, size_t* out_len, void* hip_stream);

/* What the last encode on this context did (also the hook bench.py reads its HIP-event timings
 * from). */
#define MI355_STAGE_LINKS 0   /* k_links        (chained_hash_table.rs) */
#define MI355_STAGE_MATCH 1   /* k_match/k_rle  (matching.rs, rle.rs) */
#define MI355_STAGE_PARSE 2   /* k_adv .. k_compact (lz77.rs parsers as a restart path) */
#define MI355_STAGE_BLOCKS 3  /* bounds, histogram, header, plan (output_writer.rs, huffman_lengths.rs) */
#define MI355_STAGE_PACK 4    /* k_pack         (encoder_state.rs, bitstream.rs, stored_block.rs) */
#define MI355_STAGE_OTHER 5   /* memset, adler, copies */
#define MI355_N_STAGES 6
typedef struct {
    uint64_t in_len, out_len;
    uint64_t n_tokens;
    uint32_t n_blocks, n_stored, n_fixed, n_dynamic;
    uint32_t q1_rewarm;       /* 1 if the hash re-warm quirk (A.4 Q1) fired and was reproduced */
    uint32_t q13_hits;        /* stored blocks hit by A.4 Q13 */
    uint32_t passes;          /* 1, or 2 when Q1 forced a second pass */
    uint32_t spec_fallback;   /* 1: the speculative segment entries of the parse did not check out (long periodic data) and
                                 the call was parsed again the exact way -- same bytes, about a millisecond per 100 MB more */
    float stage_ms[MI355_N_STAGES]; /* HIP-event time per stage, summed over passes (0 without the stage clocks: MI355_CFG_STAGE_CLOCKS) */
    float total_ms;                 /* first kernel enqueued .. last kernel done (without the stage clocks: the host's clock over the same) */
    uint32_t match_launches;        /* launches of the dominant kernel in this encode */
    float match_ms;                 /* their summed duration */
    uint32_t spec_repaired;         /* segments whose speculative entry was wrong and that were parsed again in place */
    uint32_t host_path;             /* how the last host-buffer call (mi355_deflate_encode and its forms) moved its bytes:
                                       MI355_HOST_PATH_* bits; 0 after a device-buffer call */
} mi355_deflate_info;
#define MI355_HOST_PATH_PIECES 1u      /* worked on and handed back piece by piece (MI355_CFG_HOST_STREAMING) */
#define MI355_HOST_PATH_IN_THREADS 2u  /* pageable input: carried to the device by the context's host threads */
#define MI355_HOST_PATH_OUT_THREADS 4u /* pageable output: carried back by them */
int mi355_deflate_last_info(mi355_deflate_ctx* ctx, mi355_deflate_info* info);

/* Diagnostics: the block layout of the last encode (type, BFINAL, tokens, input bytes, bit offset
 * of the first header bit inside the raw deflate stream) -- the per-block facts of
 * compress_data_dynamic_n's loop (src/compress.rs:157-246), for diffing against an oracle. */
typedef struct {
    uint32_t btype; /* 0 stored, 1 fixed, 2 dynamic */
    uint32_t bfinal;
    uint32_t n_tokens;
    uint32_t reserved;
    uint64_t in_bytes;
    uint64_t bit_start;
} mi355_block_info;
int mi355_deflate_last_blocks(mi355_deflate_ctx* ctx, mi355_block_info* out, size_t cap, size_t* n_blocks);

/* ---- sharded encode: ONE input over several GPUs, stream-exact (P1) ---------------------------
 * Rank r holds in device memory the bytes [global_lo, global_lo + n_ext) of the input: its own range
 * [parse_lo, parse_hi) (buffer coordinates; parse_lo = 32768 of history except on the first rank,
 * global_lo a multiple of 32768) plus >= 128 KiB of look-ahead except on the last rank (258 bytes of match
 * look-ahead, and a Stored block that begins in the range -- never more than ~110 KB -- must end inside it).  What the
 * reference's single loop threads through the stream (src/lz77.rs parse state, the 31744-value block
 * counter src/output_writer.rs:19, the bit position src/compress.rs:167) is exchanged between the
 * phases: a 576-entry exit table, token counts, <= 31743 straddling tokens, per-block costs.
 *   1. begin      links, match table, restart steps of the buffer; the range is parsed by speculation (its first
 *                 segment, like every other, finds its entry by a run-up of 128 positions into the history)
 *   2. spec       did the chain of segment entries and exits inside the range hold, where was the range entered and
 *                 left?  all-gather: if every rank's entry is the exit of the rank before it (rank 0: position 0),
 *                 every entry is the true one, the tokens of step 3 are there already and step 2b is not needed
 *   2b. exit_table X[e] = where the parse leaves the range (offset beyond parse_hi) if it enters at
 *                 parse_lo + e (made when asked for: the exact way, for periodic data);  all-gather, then every
 *                 rank knows its entry position
 *   3. emit       tokens of the path positions in [entry, parse_hi) (returns at once with the speculation's tokens
 *                 if entry is where that entered); all-gather the counts;
 *                 rank r sends the first (-first_token_index mod 31744) of its tokens to rank r-1
 *   4. blocks     histogram + Huffman per owned block -> cost records; all-gather
 *   5. mi355_plan_blocks (host, every rank, identical result) -> block types and global bit offsets
 *   6. pack       the rank's blocks at their global bit offsets; byte ranges are OR-stitched on rank 0
 * deflate-rs_amd/shard.py drives this over torch.distributed (one process per GPU); mi355_deflate_encode_multi
 * below drives it inside the library (one process, one thread per GPU). */
typedef struct mi355_shard mi355_shard;
typedef struct {
    uint64_t dyn_bits, dyn_est, static_est, fixed_bits, in_bytes;
    uint32_t q13, reserved;
} mi355_block_cost;
int mi355_shard_begin(mi355_deflate_ctx* ctx, const void* d_ext, size_t n_ext, size_t parse_lo, size_t parse_hi,
                      uint64_t global_lo, uint64_t n_global, const mi355_deflate_opts* opts, void* hip_stream,
                      mi355_shard** out);
/* held != 0: the range was parsed from *entry to *exit_pos (buffer coordinates, *exit_pos >= parse_hi), consistently
 * inside; 0: the caller takes the exact way (exit_table, emit). */
int mi355_shard_spec(mi355_shard* s, int* held, uint64_t* entry, uint64_t* exit_pos);
int mi355_shard_exit_table(mi355_shard* s, uint32_t* table576);
int mi355_shard_emit(mi355_shard* s, uint64_t entry, uint64_t* n_tokens, const void** d_tokens);
int mi355_shard_blocks(mi355_shard* s, uint64_t skip_tokens, const void* d_tail_tokens, uint64_t n_tail,
                       uint64_t* n_blocks, mi355_block_cost* costs, size_t costs_cap);
/* The same with the owner of the stream's last block named by the caller (owns_final != 0: this rank's tokens
 * end in the last, possibly partial or empty, block; 0: it owns whole blocks only, possibly none).
 * mi355_shard_blocks decides that by "is the last rank", which holds whenever every rank's range reaches the
 * next block boundary; a range with fewer tokens than that belongs to a block that began several ranks to its
 * left, and the tail a rank needs is then made of the heads of several ranks to its right. */
int mi355_shard_blocks_ex(mi355_shard* s, uint64_t skip_tokens, const void* d_tail_tokens, uint64_t n_tail,
                          int owns_final, uint64_t* n_blocks, mi355_block_cost* costs, size_t costs_cap);
int mi355_plan_blocks(const mi355_block_cost* costs, size_t n, uint32_t compat, mi355_block_info* plans,
                      uint64_t* total_bits);
int mi355_shard_pack(mi355_shard* s, const mi355_block_info* plans, uint64_t end_bit, void* d_out, size_t out_cap,
                     uint64_t* first_byte, size_t* n_bytes);
void mi355_shard_end(mi355_shard* s);
/* A sharded zlib / gzip stream (deflate_bytes_zlib_conf, ZlibEncoder: src/lib.rs:182-198, src/checksum.rs:33-57):
 * every rank sums its own byte range (mi355_adler32_device / mi355_crc32_device), rank 0 folds the sums in
 * rank order with this and frames the stitched raw stream.  kind 1 = Adler-32, 2 = CRC-32;
 * returns checksum(A || B) from checksum(A), checksum(B) and |B|. */
uint32_t mi355_checksum_combine(int kind, uint32_t sum_a, uint32_t sum_b, uint64_t len_b);

/* ---- ONE input over the GPUs of a node in ONE call (stream-exact) -----------------------------------------
 * deflate_bytes_conf / deflate_bytes_zlib_conf / deflate_bytes_gzip_conf (src/lib.rs:137-147, 182-198, 242-267) with
 * several devices behind them: one process, one context and one host thread per device.  The input is cut into
 * contiguous ranges of whole 32 KiB windows, one per device (each device also holds 32 KiB of history and 128 KiB of
 * look-ahead: read from the host buffer, no GPU-to-GPU traffic); every device runs the phases of the sharded encode
 * above on its range, what they exchange -- exit tables, token counts and head tokens, block costs: kilobytes --
 * goes through host memory, and the packed byte ranges land at their offsets in `out`, the word two neighbours share
 * OR-ed in.  The bytes are those of ONE encoder over the whole input, i.e. the reference's.
 *   _create    devices[i] = HIP device of rank i (a device may be named more than once: ranks then share it --
 *              how a one-GPU box tests the N-rank path); n_devices <= 64; devices == NULL with n_devices == 0: every
 *              device of the node (mi355_device_count of them, in HIP's order)
 * A handle serves ONE call at a time: the encode calls hold its lock from entry to return, calls from several threads
 * are safe and serialised, and the calling thread's current HIP device is what it was when the call returns.
 * Size: a rank is one pass of the pipeline (32-bit positions, about 20 B of device memory per byte of its range), so an
 * input of more than 2 x MI355_CFG_RANGE_BYTES of rank 0's context per device (default: 1 GiB per device) is cut into
 * MORE ranks than devices -- rank r lives on device r % n_devices, the ranks of a device run one after the other in
 * every phase, each with a context of its own -- up to 64 ranks (64 GiB with the default).  Beyond that the host-buffer
 * call goes through the single-device call of rank 0, which walks any length in ranges; the device-resident call,
 * whose shards the caller lays out, returns MI355_E_UNSUPPORTED (raise MI355_CFG_RANGE_BYTES: a rank is clamped to what one
 * pass addresses, just under 4 GiB -- reached at 2 GiB of MI355_CFG_RANGE_BYTES -- so 64 ranks take about 255 GiB).
 *   _encode_multi         host buffers in and out (ctx-less: the handle owns its contexts); wrapper 0 / 1 / 2 as in
 *              mi355_deflate_opts, `gz_hdr` = GzBuilder::into_header() for wrapper 2 (NULL: the blank header);
 *              out_cap >= mi355_deflate_bound_ex(in_len, wrapper, gz_len, 0); an input of less than 1 MiB per
 *              device uses fewer devices (down to the plain single-device call)
 *   _layout    which bytes rank `rank` holds for an input of in_len bytes: [g_lo, g_hi) of the input, of which
 *              [lo, hi) is its own range; *n_ranks = the ranks that take part (rank >= *n_ranks: nothing) -- fewer than
 *              the devices for a short input, more for a long one (rank r on device r % n_devices)
 *   _encode_multi_device  the same with the input already resident: d_ext[r] = device pointer, on rank r's device
 *              (device r % n_devices), to the bytes [g_lo, g_hi) of _layout (+ 64 readable bytes behind them); d_out = device buffer on
 *              rank 0's device (4-byte aligned).  The packed ranges reach it by peer copies (hipMemcpyPeer: xGMI
 *              between the GPUs of a node) -- the only GPU-to-GPU traffic of the call
 *   _ctx       rank r's context (mi355_deflate_last_info of it: that rank's match-kernel time ...; NULL for a rank no
 *              call has needed yet)
 *   _last_trace  rank 0's wall clock of the last call in ms: [0] bytes + tables, [1] wait, [2] entry + tokens,
 *              [3] wait, [4] block costs, [5] wait, [6] plan + pack + copy-out, [7] wait, [8] seams + framing,
 *              [9] host work of the exchanges (entries, token plan, tails, block plan, seams), [10] the whole call
 *   _stitch_info  how the packed ranges of the last call reached rank 0: *by_rccl = 1 over ncclSend / ncclRecv
 *              (MI355_CFG_MULTI_STITCH = 1, resident output), 0 by copies; *rccl_ranks = the ranks the communicator of rank 0's
 *              device reports (one per distinct device of the handle), 0 when the handle holds none.  A stitch that fails on
 *              one device ends the call with an error inside a bounded wait, aborts the communicators, and later calls of
 *              the handle take the peer copies */
typedef struct mi355_multi mi355_multi;
int mi355_device_count(void); /* HIP devices this process sees (0: none, or no HIP runtime); for callers that do not link HIP */
int mi355_multi_create(const int* devices, int n_devices, mi355_multi** out);
void mi355_multi_destroy(mi355_multi* m);
int mi355_multi_devices(const mi355_multi* m);
mi355_deflate_ctx* mi355_multi_ctx(mi355_multi* m, int rank);
const char* mi355_multi_last_error(mi355_multi* m);
int mi355_multi_layout(const mi355_multi* m, size_t in_len, int rank, int* n_ranks, uint64_t* g_lo, uint64_t* g_hi,
                       uint64_t* lo, uint64_t* hi);
int mi355_deflate_encode_multi(mi355_multi* m, const uint8_t* in, size_t in_len, const mi355_deflate_opts* opts,
                               const uint8_t* gz_hdr, size_t gz_len, uint8_t* out, size_t out_cap, size_t* out_len);
int mi355_deflate_encode_multi_device(mi355_multi* m, const void* const* d_ext, size_t in_len,
                                      const mi355_deflate_opts* opts, const uint8_t* gz_hdr, size_t gz_len, void* d_out,
                                      size_t out_cap, size_t* out_len);
int mi355_multi_last_trace(const mi355_multi* m, double* ms, size_t cap);
int mi355_multi_stitch_info(const mi355_multi* m, int* by_rccl, int* rccl_ranks);

/* The gzip forms (cargo feature "gzip").  `hdr` = the bytes GzBuilder::into_header() returned: the
 * header comes from the crate gzip-header 1.0, which is not part of the reference tree, so the shim
 * builds it with the real crate and passes it through.  Written after the stream: Crc::sum() and
 * Crc::amt_as_u32() little endian (src/lib.rs:258-266), both computed on the GPU.
 *   mi355_deflate_encode_gzip        = deflate_bytes_gzip_conf(input, options, builder)  :242-267
 *   mi355_deflate_encode with wrapper 2 = deflate_bytes_gzip(input) (blank header)        :283-285 */
int mi355_deflate_encode_gzip(mi355_deflate_ctx* ctx, const uint8_t* in, size_t in_len, const mi355_deflate_opts* opts,
                              const uint8_t* hdr, size_t hdr_len, uint8_t* out, size_t out_cap, size_t* out_len);
int mi355_deflate_encode_device_gzip(mi355_deflate_ctx* ctx, const void* d_in, size_t in_len,
                                     const mi355_deflate_opts* opts, const uint8_t* hdr, size_t hdr_len, void* d_out,
                                     size_t out_cap, size_t* out_len, void* hip_stream);
/* CRC-32 (RFC 1952 section 8) of a device buffer: gzip_header::Crc::update + sum as used by
 * src/lib.rs:258-259 and src/writer.rs:436-444, computed on the GPU. */
int mi355_crc32_device(mi355_deflate_ctx* ctx, const void* d_in, size_t in_len, uint32_t* crc, void* hip_stream);

/* Adler-32 of a device buffer (crate adler32's RollingAdler32::update_buffer as used by
 * src/checksum.rs:33-57), computed on the GPU. */
int mi355_adler32_device(mi355_deflate_ctx* ctx, const void* d_in, size_t in_len, uint32_t* adler, void* hip_stream);

/* Streaming encoder mirroring write::{DeflateEncoder, ZlibEncoder, gzip::GzEncoder}<W> (src/writer.rs:89-467):
 *   _new         = ::new(W, options)                    :93-99 / :189-199
 *   _write       = io::Write::write_all                 :124-127 / :254-267
 *   _flush       = io::Write::flush (Flush::Sync)       :134-137 / :274-277
 *   _finish      = finish(self) -> W                    :103-108 / :209-214
 *   _output      = the bytes produced and not yet taken (what the inner writer W is owed)
 *   _take_output = hand over up to `cap` of them: the shim's loop over W::write, which may accept fewer
 *                  bytes than offered (src/compress.rs:96-124, 280-299; tests/test.rs:163-200 issue_47)
 *   _checksum    = {Zlib,Gz}Encoder::checksum()         :248-250 / :428-430
 * write() gathers (the reference likewise does nothing until its 64 KiB + 258 buffer is full,
 * src/lz77.rs:627, and its output does not depend on how the input is split: src/lib.rs:408-433); the GPU
 * encodes at flush() and finish(), and the bytes of a flush -- ending in 00 00 FF FF, as
 * src/writer.rs:570-595 asserts -- are available when flush() returns.  Between flushes the handle keeps
 * the bytes since the last flush plus the 32 KiB window before it (the whole stream while the flushed part
 * is shorter than three windows).  What gathers is bounded all the same (the reference's O(window) streaming,
 * src/compress.rs:96-124): whenever 512 MiB and a margin of 16 MiB have gathered -- since the start of a stream that
 * has not been flushed yet, or since the last flush point -- write() encodes that range and makes its bytes available
 * (_output / _take_output), keeping the rest, one window of history and the bytes not yet taken: a stream of any
 * length, e.g. 8 GiB into a ZlibEncoder with or without a flush() in the middle, holds about 0.55 GiB of host memory.
 * The first range behind a flush point begins AT it, with the hash side effects of the write calls around the flush;
 * the next flush() / finish() ends the last range (with the sync marker / the final block).  Behind a flush() within the
 * first 96 KiB of a stream -- where a stream's start has hash rules of its own, src/lz77.rs:601-638 -- the first range
 * holds the stream from its start and takes over what the flush calls found: the flush points that re-warm the hash and
 * the re-warm behind a block that fills inside the first window (Q1).  One _write call stands for one write_all call (n == 0: no call at all);
 * the size of the first write after a flush is remembered, because the reference's hash re-warm at a
 * flush point inside the first window depends on it (src/lz77.rs:601-638).  The shim's Drop calls _finish
 * and drains the output like the reference's (src/writer.rs:139-152). */
typedef struct mi355_deflate_stream mi355_deflate_stream;
int mi355_deflate_stream_new(mi355_deflate_ctx* ctx, const mi355_deflate_opts* opts, mi355_deflate_stream** out);
int mi355_deflate_stream_write(mi355_deflate_stream* s, const uint8_t* data, size_t n);
/* io::Write::flush (Flush::Sync, src/writer.rs:134-137, 274-277; src/compress.rs:256-261; src/lz77.rs:
 * 605-614, 728-740): what was written so far is emitted in non-final blocks + 00 00 FF FF, the window
 * is kept.  The hash-table side effects the reference has for writes of one or two bytes around a flush
 * (positions filed late, under a stale rolling hash, or not at all; the re-warm of the first window) are
 * reproduced for every pattern; no write / flush sequence is refused. */
int mi355_deflate_stream_flush(mi355_deflate_stream* s);
int mi355_deflate_stream_finish(mi355_deflate_stream* s);
/* GzEncoder::from_builder (src/writer.rs:346-358): header bytes of a wrapper-2 stream, before the first
 * write; _checksum of such a stream is GzEncoder::checksum() (:428-430), the CRC-32. */
int mi355_deflate_stream_gzip_header(mi355_deflate_stream* s, const uint8_t* hdr, size_t hdr_len);
/* reset(&mut self, W) -> io::Result<W> (src/writer.rs:110-117, 216-223, 383-402): finishes the stream,
 * hands its untaken bytes out (valid until the next reset / free) and starts a new one with the same options
 * (gzip: with the blank header again, as GzEncoder::reset does). */
int mi355_deflate_stream_reset(mi355_deflate_stream* s, const uint8_t** data, size_t* n);
int mi355_deflate_stream_output(mi355_deflate_stream* s, const uint8_t** data, size_t* n);
int mi355_deflate_stream_take_output(mi355_deflate_stream* s, uint8_t* dst, size_t cap, size_t* n);
int mi355_deflate_stream_checksum(mi355_deflate_stream* s, uint32_t* adler);
/* Host memory the handle holds right now: gathered input, the window, bytes produced and not yet taken (what the
 * reference bounds at about 330 KiB, src/deflate_state.rs:82-119; here one range for a stream that is never flushed). */
uint64_t mi355_deflate_stream_held_bytes(mi355_deflate_stream* s);
void mi355_deflate_stream_free(mi355_deflate_stream* s);

#ifdef MI355_DEBUG_HOOKS
/* Test build only (libmi355deflate_dbg.so, `make debug`; never part of the product library): make the hash sort of this
 * context hand out a wrong order, as a device would whose LDS did not serve same-address lanes in lane order, so that
 * a test can watch the match kernel's order check catch it.  _sort_ranks reports what the context sorts with now. */
int mi355_debug_break_sort(mi355_deflate_ctx* ctx, int on);
int mi355_debug_sort_ranks(mi355_deflate_ctx* ctx);
#endif

#ifdef __cplusplus
}
#endif
#endif
