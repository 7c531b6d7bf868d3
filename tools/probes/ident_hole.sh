#!/bin/bash
# development aid (GPU box): tests/golden/identity_hop.bin against a build that sends only TWO epochs through the link kernels
# after a re-warm (the tree before round 5) -- the stream must differ from the oracle's there, and agree in the product build
cd $GRAFT_REPO_ROOT/deflate-rs_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DMI355_IDENT_EPOCHS=2 -shared -o /tmp/libident2.so csrc/deflate_kernels.hip 2>/dev/null
cd $GRAFT_REPO_ROOT && python - <<'PY'
import sys, os
sys.path.insert(0, "deflate-rs_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import torch, deflate_amd as da, oracle_binding as ob, tokdump
data = open("tests/golden/identity_hop.bin", "rb").read()
want = ob.encode(data, opts=ob.make_opts(128, 32, 1))
for lib in ("/tmp/libident2.so", da.LIB_PATH):
    da.LIB_PATH = lib; da._lib = None
    ctx = da.Context(0)
    got = ctx.encode(data, da.Compression.Default)
    toks = [t for b in tokdump.tokens(got) for t in b["toks"] if 65534 <= t[0] <= 65540]
    print(os.path.basename(lib), "same as oracle:", got == want, len(got), len(want), "tokens around 65536:", toks)
    ctx.close()
PY
