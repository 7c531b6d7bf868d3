#!/bin/bash
# development aid (GPU box): the host side of the library under AddressSanitizer
#   build first (CPU box): hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Xarch_host -fsanitize=address -shared \
#        -o deflate-rs_amd/variants/lib_asan.so deflate-rs_amd/csrc/deflate_kernels.hip
#        clang -g -O1 -fsanitize=address -Iinclude examples/mi355_deflate_cli.c deflate-rs_amd/variants/lib_asan.so -o deflate-rs_amd/variants/cli_asan
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:protect_shadow_gap=0
V=$PWD/deflate-rs_amd/variants
for args in "-raw -default" "-raw -default -chunk 5000" "-zlib -best" "-zlib -best -chunk 5000" "-gzip -fast" "-gzip -fast -chunk 5000" "-zlib -best -chunk 1"; do
  (cd deflate-rs_amd && variants/cli_asan $args ../tests/golden/ref_inputs/pg11.txt /tmp/out.bin 2>&1 | grep -E "ERROR|SUMMARY|bytes,|error" | head -5)
done
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
LD_PRELOAD=$RT MI355_DEFLATE_LIB=$V/lib_asan.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streaming or flush or writer or issue or drop or many_small or long_stream or reset or gzip or randomized or zlib or known or edge" 2>&1 | grep -E "ERROR: AddressSanitizer|SUMMARY|passed|failed|#[0-9] " | head -30
