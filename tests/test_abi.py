"""The C-ABI library loads on a machine without a GPU and exports every symbol include/starphase_hip.h declares;
no compute call is made here.  Also checks the 'fail loudly' behaviour: no device => SP_ERR_NO_DEVICE, never a fallback."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "starphase_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(pkg):
    lib = C.CDLL(pkg.lib_path())
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/starphase_hip.h but not exported"
    assert lib.sp_abi_version() == pkg.ffi.SP_ABI_VERSION == 2


def test_binding_matches_header(pkg):
    pkg.ffi.lib()          # binds every function; raises AttributeError on a missing one
    pkg.database._lib()    # the database / result-file part of the header
    pkg.database._io()     # the BAM / VCF readers
    import inspect
    src = inspect.getsource(pkg.ffi) + inspect.getsource(pkg.database)
    for n in declared_symbols():
        assert n in src, f"{n} has no ctypes binding"


def test_rust_extern_block_is_complete_and_current():
    """include/starphase_hip.rs -- the `extern "C"` block a Rust host (the reference's language) binds, INTEGRATION.md -- is generated from the header:
    every declared function is in it, and it is the generator's output for the header as it stands"""
    import subprocess
    import sys
    rs = open(os.path.join(ROOT, "include", "starphase_hip.rs")).read()
    fns = set(re.findall(r"pub fn (sp_[a-z0-9_]+)\(", rs))
    decl = set(declared_symbols())
    assert decl <= fns, f"not in the Rust block: {sorted(decl - fns)}"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "scripts", "rust_externs.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_no_device_is_an_error_not_a_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = pkg.ffi.lib()
    h = C.c_void_p()
    rc = lib.sp_ctx_create(0, None, C.byref(h))
    assert rc == pkg.ffi.SP_ERR_NO_DEVICE and not h.value
    with pytest.raises(pkg.StarphaseError):
        pkg.Context(0)


def test_hardware_queue_default_is_asked_for_before_hip_starts(pkg):
    """sp_ctx_create exports GPU_MAX_HW_QUEUES=16 when the host has not set it (before its own first HIP call; include/starphase_hip.h, sp_ctx_info) and leaves a
    host's own value alone -- checked in child processes, without a device (the call then fails with SP_ERR_NO_DEVICE, after the variable was dealt with)"""
    import subprocess
    import sys
    code = ("import os, sys, ctypes as C; sys.path.insert(0, %r); import __graft_entry__ as ge; pkg = ge.load_package(); h = C.c_void_p(); "
            "pkg.ffi.lib().sp_ctx_create(0, None, C.byref(h)); g = C.CDLL(None).getenv; g.restype = C.c_char_p; print((g(b'GPU_MAX_HW_QUEUES') or b'None').decode())" % ROOT)
    for given, want in ((None, "16"), ("8", "8"), ("24", "24")):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        if given:
            env["GPU_MAX_HW_QUEUES"] = given
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-1000:]
        assert out.stdout.strip().splitlines()[-1] == want


def test_the_binding_refuses_a_library_of_another_abi(pkg, monkeypatch):
    """records the caller allocates have grown (round 4: mm2_* fields; round 5: k1_* fields): the version and the record sizes are checked when the library is bound"""
    L = pkg.ffi.lib()
    text = open(os.path.join(ROOT, "include", "starphase_hip.h")).read()
    assert int(re.search(r"#define SP_ABI_VERSION (\d+)", text).group(1)) == pkg.ffi.SP_ABI_VERSION
    rs = open(os.path.join(ROOT, "include", "starphase_hip.rs")).read()
    assert int(re.search(r"SP_ABI_VERSION: i64 = (\d+)", rs).group(1)) == pkg.ffi.SP_ABI_VERSION
    # every struct of the binding that the header names has the library's size
    for name in dir(pkg.ffi):
        t = getattr(pkg.ffi, name)
        if isinstance(t, type) and issubclass(t, C.Structure) and name.startswith("sp_"):
            size = L.sp_struct_size(name.encode())
            assert size in (-1, C.sizeof(t)), f"{name}: binding {C.sizeof(t)} bytes, library {size}"
    assert L.sp_struct_size(b"sp_hla_realign") == pkg.ffi.REALIGN_DTYPE.itemsize == 112
    assert L.sp_struct_size(b"no_such_record") == -1
    monkeypatch.setattr(pkg.ffi, "_lib", None)
    monkeypatch.setattr(pkg.ffi, "SP_ABI_VERSION", 1)
    with pytest.raises(ImportError, match="ABI version"):
        pkg.ffi.lib()
    monkeypatch.setattr(pkg.ffi, "SP_ABI_VERSION", 2)
    monkeypatch.setattr(pkg.ffi, "_lib", None)
    pkg.ffi.lib()


def test_struct_layouts(pkg):
    assert C.sizeof(pkg.ffi.sp_ctx_info) == 16 + 256
    assert C.sizeof(pkg.ffi.sp_aln) == 32
    assert C.sizeof(pkg.ffi.sp_pair) == 16
    assert pkg.ffi.ALN_DTYPE.itemsize == 32
    assert pkg.ffi.REALIGN_DTYPE.itemsize == C.sizeof(pkg.ffi.sp_hla_realign)
