/* A plain C program over the C ABI of include/mi355_deflate.h (what the Rust shim of INTEGRATION.md
 * binds): compresses a file on the GPU.
 *   mi355_deflate_cli [-raw|-zlib|-gzip] [-fast|-default|-best] [-chunk N] IN OUT
 * -chunk N drives the streaming handle (write N bytes at a time) instead of the one-shot call.
 * Build:  gcc -O2 -Iinclude examples/mi355_deflate_cli.c -Ldeflate-rs_amd -lmi355deflate \
 *             -Wl,-rpath,$PWD/deflate-rs_amd -o /tmp/mi355_deflate_cli */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355_deflate.h"

static int fail(const char* what, int rc, mi355_deflate_ctx* ctx) {
    fprintf(stderr, "%s: error %d (%s)\n", what, rc, ctx ? mi355_deflate_last_error(ctx) : "");
    return 1;
}

int main(int argc, char** argv) {
    int wrapper = 0, level = 1;
    size_t chunk = 0;
    int a = 1;
    for (; a < argc && argv[a][0] == '-'; a++) {
        if (!strcmp(argv[a], "-raw")) wrapper = 0;
        else if (!strcmp(argv[a], "-zlib")) wrapper = 1;
        else if (!strcmp(argv[a], "-gzip")) wrapper = 2;
        else if (!strcmp(argv[a], "-fast")) level = 0;
        else if (!strcmp(argv[a], "-default")) level = 1;
        else if (!strcmp(argv[a], "-best")) level = 2;
        else if (!strcmp(argv[a], "-chunk") && a + 1 < argc) chunk = strtoull(argv[++a], NULL, 10);
        else break;
    }
    if (argc - a != 2) {
        fprintf(stderr, "usage: %s [-raw|-zlib|-gzip] [-fast|-default|-best] [-chunk N] IN OUT\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(argv[a], "rb");
    if (!f) return fail("open input", -1, NULL);
    fseek(f, 0, SEEK_END);
    size_t n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* in = (uint8_t*)malloc(n ? n : 1);
    if (fread(in, 1, n, f) != n) return fail("read input", -1, NULL);
    fclose(f);

    mi355_deflate_ctx* ctx = NULL;
    int rc = mi355_deflate_ctx_create(0, &ctx);
    if (rc) return fail("mi355_deflate_ctx_create (no GPU? there is no CPU fallback)", rc, NULL);
    mi355_deflate_opts o;
    mi355_deflate_preset(level, &o); /* Compression::{Fast,Default,Best} */
    o.wrapper = (uint8_t)wrapper;

    const uint8_t* out = NULL;
    uint8_t* owned = NULL;
    size_t out_len = 0;
    mi355_deflate_stream* s = NULL;
    if (chunk) { /* write::{Deflate,Zlib,Gz}Encoder: new, write_all ..., finish */
        rc = mi355_deflate_stream_new(ctx, &o, &s);
        if (rc) return fail("stream_new", rc, ctx);
        for (size_t i = 0; i < n; i += chunk) {
            size_t k = n - i < chunk ? n - i : chunk;
            if ((rc = mi355_deflate_stream_write(s, in + i, k))) return fail("stream_write", rc, ctx);
        }
        if ((rc = mi355_deflate_stream_finish(s))) return fail("stream_finish", rc, ctx);
        mi355_deflate_stream_output(s, &out, &out_len);
    } else { /* deflate_bytes_conf / deflate_bytes_zlib_conf / deflate_bytes_gzip */
        size_t cap = mi355_deflate_bound(n) + 64;
        owned = (uint8_t*)malloc(cap);
        rc = mi355_deflate_encode(ctx, in, n, &o, owned, cap, &out_len);
        if (rc) return fail("mi355_deflate_encode", rc, ctx);
        out = owned;
    }
    f = fopen(argv[a + 1], "wb");
    if (!f || fwrite(out, 1, out_len, f) != out_len) return fail("write output", -1, NULL);
    fclose(f);
    mi355_deflate_info info;
    mi355_deflate_last_info(ctx, &info);
    fprintf(stderr, "%zu -> %zu bytes, %u blocks, %.3f ms on the GPU\n", n, out_len, info.n_blocks, info.total_ms);
    if (s) mi355_deflate_stream_free(s);
    free(owned);
    free(in);
    mi355_deflate_ctx_destroy(ctx);
    return 0;
}
