"""The host-side scalar routines of the library (no GPU needed: they do no device work) against the reference's own tests."""
import ctypes as C

import numpy as np

import oracle_ffi as of


def test_is_passing_dual(pkg):
    """src/hla/caller.rs:1837-1845"""
    L = pkg.ffi.lib()
    f = lambda a, b: L.sp_hla_is_passing_dual(a, b, 0.10, 0.5, 0.001, None, None)
    assert (f(3, 20), f(20, 3), f(10, 20), f(20, 10)) == (0, 0, 1, 1)


def test_is_hemizygous_better(pkg, oracle):
    """src/hla/caller.rs:1884-1898 ; identical costs to the oracle restatement"""
    L = pkg.ffi.lib()
    for (c1, c2, norm, delta), want in (((20, 0, 20.0, 1), 1), ((40, 0, 20.0, 1), 0), ((18, 2, 20.0, 1), 1), ((18, 17, 20.0, 1), 0), ((15, 6, 20.0, 20), 0)):
        n = c1 + c2
        is_c1 = np.array([1] * c1 + [0] * c2, np.uint8)
        s1 = np.array([0] * c1 + [delta] * c2, np.int64)
        s2 = np.array([delta] * c1 + [0] * c2, np.int64)
        h, d, oh, od = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        got = L.sp_hla_is_hemizygous_better(s1.ctypes.data, s2.ctypes.data, is_c1.ctypes.data, n, 1 if c2 else 0, 20, norm, C.byref(h), C.byref(d))
        exp = oracle.L.osp_is_hemizygous_better(s1.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p), is_c1.ctypes.data_as(C.c_void_p),
                                                n, 1 if c2 else 0, 20, 1, norm, C.byref(oh), C.byref(od))
        assert got == want == exp
        assert abs(h.value - oh.value) <= 1e-9 and abs(d.value - od.value) <= 1e-9


def test_hpc(pkg):
    """src/util/homopolymers.rs:72-100"""
    L = pkg.ffi.lib()
    out = C.create_string_buffer(64)
    n = L.sp_hpc(b"AACAAAAAAGGGTAACAA", 18, out)
    assert out.raw[:n] == b"ACAGTACA"
    seq = b"AACCCGTTTT"
    assert [L.sp_hpc_pos(seq, len(seq), i) for i in range(len(seq))] == [0, 0, 1, 1, 1, 2, 3, 3, 3, 3]
    assert L.sp_hpc_pos(b"ATTGGGGGAACCCGTTTT", 18, 6) == 2


def test_chain_to_hap(pkg):
    """src/cyp2d6/caller.rs:972-1006"""
    L = pkg.ffi.lib()
    cfg = of.default_cyp_config()
    tk = (C.c_char_p * len(cfg["translate"]))(*[a.encode() for a, _ in cfg["translate"]])
    tv = (C.c_char_p * len(cfg["translate"]))(*[b.encode() for _, b in cfg["translate"]])
    types = np.array([6, 2, 2, 2, 2], np.int32)
    subs = (C.c_char_p * 5)(None, b"1.001", b"10", b"1.002", b"1.002")

    def hap(chain, detail):
        ch = np.array(chain, np.int32)
        out = C.create_string_buffer(256)
        L.sp_cyp_chain_to_hap(ch.ctypes.data, len(ch), types.ctypes.data, subs, len(cfg["translate"]), tk, tv, detail, out, 256)
        return out.value.decode()
    assert hap([2, 2, 1, 0], 1) == "*1.001 + *10x2"
    assert hap([3, 1, 0], 1) == "*1.001 + *1.002"
    assert hap([3, 1, 0], 0) == "*1x2"
    assert hap([3, 4], 1) == "*1.002x2"
