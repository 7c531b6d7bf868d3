"""ctypes binding of tests/hostsim (host build of the stage functions).  TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "hostsim")
LIB = os.path.join(DIR, "libhostsim.so")


class Block(C.Structure):
    _fields_ = [("btype", C.c_uint32), ("bfinal", C.c_uint32), ("ntok", C.c_uint32),
                ("in_bytes", C.c_uint64), ("bit_start", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", DIR, "-s"])
        L = C.CDLL(LIB)
        L.hostsim_encode.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint32), C.POINTER(Block), C.c_uint64,
                                     C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32]
        L.hostsim_encode.restype = C.c_int
        L.hostsim_match_table.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32)]
        L.hostsim_rle_forms_agree.argtypes = [C.c_char_p, C.c_uint32]
        L.hostsim_rle_forms_agree.restype = C.c_int
        _lib = L
    return _lib


def encode(data, checks, lazy_lt, matching_type, seg=0, fan=4):
    """returns (rc, bytes, flags, blocks)"""
    n = len(data)
    cap = n + 5 * (n // 32767 + 2) + 64
    out = (C.c_uint8 * cap)()
    olen = C.c_uint64(0)
    flags = C.c_uint32(0)
    nb = C.c_uint64(0)
    bcap = n // 31744 + 2
    blocks = (Block * bcap)()
    rc = lib().hostsim_encode(bytes(data), n, checks, lazy_lt, matching_type, out, cap, C.byref(olen),
                              C.byref(flags), blocks, bcap, C.byref(nb), seg, fan)
    bl = [dict(btype=b.btype, bfinal=b.bfinal, n_lz=b.ntok, in_bytes=b.in_bytes, bit_start=b.bit_start)
          for b in blocks[: nb.value]]
    return rc, bytes(memoryview(out)[: olen.value]) if rc == 0 else b"", flags.value, bl


def swz(a):
    """stages.h m3_swz: where the permuted pair table keeps the byte of LDS address a"""
    L = lib()
    L.hostsim_swz.restype = C.c_uint32
    return int(L.hostsim_swz(C.c_uint32(a)))


def use_multi(on):
    """switch the match stage to match_walk_multi (the formulation k_match runs)"""
    lib().hostsim_use_multi(int(on))


def force_ident(on):
    """start from the reference's head[h] = h table and follow those hops (must never change a result)"""
    lib().hostsim_force_ident(int(on))


def match_table(data, checks):
    n = len(data)
    m = (C.c_uint32 * max(n, 1))()
    lib().hostsim_match_table(bytes(data), n, checks, m)
    return list(m[:n])


def rle_forms_agree(lengths: bytes) -> int:
    """0 if the run-by-run coding of a code-length list equals the reference's state machine"""
    return lib().hostsim_rle_forms_agree(bytes(lengths), len(lengths))
