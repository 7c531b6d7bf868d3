"""CPU-only: the product library loads and exports every symbol include/mi355_deflate.h declares;
without a GPU it refuses to work instead of falling back to anything."""
import os
import re
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd"))


def test_header_symbols_are_exported():
    import deflate_amd
    if not os.path.exists(deflate_amd.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "deflate-rs_amd"), "-s"])
    L = deflate_amd.load()
    hdr = open(os.path.join(ROOT, "include", "mi355_deflate.h")).read()
    # (the hooks under MI355_DEBUG_HOOKS belong to the test build libmi355deflate_dbg.so only)
    product_hdr, n_dbg = re.subn(r"#ifdef MI355_DEBUG_HOOKS.*?#endif", "", hdr, flags=re.S)
    assert n_dbg == 1
    declared = set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", product_hdr))
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(deflate_amd.EXPORTED)
    # ... and nothing else: every mi355_* symbol the product library exports is declared in the header
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", deflate_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\b[TDB] (mi355_[a-z0-9_]+)$", nm, flags=re.M))
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    assert not any(x.startswith("mi355_debug") for x in exported)
    assert L.mi355_deflate_version() >= 100
    assert L.mi355_deflate_bound(0) >= 16


def test_presets_match_reference_levels():
    import deflate_amd
    L = deflate_amd.load()
    exp = {0: (1, 0, 0), 1: (128, 32, 1), 2: (1768, 128, 1), 3: (0, 0, 1), 4: (0, 0, 0)}
    for lvl, (c, l, m) in exp.items():
        o = deflate_amd.Opts()
        assert L.mi355_deflate_preset(lvl, o) == 0
        assert (o.max_hash_checks, o.lazy_if_less_than, o.matching_type) == (c, l, m)


def test_no_cpu_fallback_without_gpu():
    import deflate_amd
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(deflate_amd.DeflateError):
        deflate_amd.Context(0)


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "deflate-rs_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp")):
                src = open(os.path.join(base, f), errors="ignore").read()
                assert "deflref" not in src and "oracle_binding" not in src and "hostsim" not in src.replace(
                    "tests/hostsim", ""), f


# the header is plain C: a C program over it (examples/mi355_deflate_cli.c) compiles with gcc and links
# against the library -- no C++ and no torch types anywhere on the boundary
def test_header_is_plain_c_and_the_example_links(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cli")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
           os.path.join(root, "examples", "mi355_deflate_cli.c"), "-L", os.path.join(root, "deflate-rs_amd"),
           "-lmi355deflate", "-Wl,-rpath," + os.path.join(root, "deflate-rs_amd"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


# what a host shim in another language does (rust/deflate-mi355: build.rs emits -lmi355deflate and nothing else): a program that
# names EVERY entry of the header -- and the symbols the Rust shim declares in its extern blocks -- links against the library
# alone.  (Round 4's shim named hipGetDeviceCount, which lives in libamdhip64: "DSO missing from command line".)
def test_every_entry_links_against_the_library_alone(tmp_path):
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "mi355_deflate.h")).read()
    product_hdr = re.sub(r"#ifdef MI355_DEBUG_HOOKS.*?#endif", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", product_hdr)))
    rs = open(os.path.join(ROOT, "rust", "deflate-mi355", "src", "lib.rs")).read()
    rust_externs = sorted(set(re.findall(r"^\s*fn ([A-Za-z0-9_]+)\s*\(", "\n".join(re.findall(r'extern "C" \{(.*?)^\}', rs, flags=re.S | re.M)),
                                         flags=re.M)))
    assert rust_externs and all(n.startswith("mi355_") for n in rust_externs), rust_externs
    assert set(rust_externs) <= set(declared), sorted(set(rust_externs) - set(declared))
    src = tmp_path / "all.c"
    src.write_text('#include "mi355_deflate.h"\n#include <stdio.h>\nint main(void) {\n  void* p[] = {\n' +
                   "".join("    (void*)%s,\n" % n for n in declared) + "  };\n  printf(\"%d\\n\", (int)(sizeof p / sizeof p[0]));\n  return p[0] == 0;\n}\n")
    exe = str(tmp_path / "all")
    cmd = ["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-L", os.path.join(ROOT, "deflate-rs_amd"),
           "-lmi355deflate", "-Wl,-rpath," + os.path.join(ROOT, "deflate-rs_amd"), "-Wl,--no-copy-dt-needed-entries", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
