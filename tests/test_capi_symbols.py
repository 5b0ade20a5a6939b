"""CPU: the C-ABI library loads and exports every symbol include/mfc.h declares."""
import ctypes

from meanflow_audio_codec_amd import _build, _lib


def test_library_builds_and_exports_header_symbols():
    _build.build(verbose=False)
    l = _lib.lib()
    declared = _lib.header_symbols()
    assert declared, "no declarations parsed from include/mfc.h"
    for name in declared:
        assert hasattr(l, name), f"{name} declared in include/mfc.h but not exported by libmfc.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.SIGNATURES"
    assert set(_lib.SIGNATURES) == set(declared)


def test_header_and_ctypes_signatures_agree():
    """Argument count and C type class (pointer / integer width / float) of every declaration in include/mfc.h
    against the ctypes signature the Python side binds it with."""
    import re
    txt = re.sub(r"/\*.*?\*/", "", _lib.HEADER.read_text(), flags=re.S)
    kinds = {ctypes.c_void_p: "ptr", ctypes.c_char_p: "ptr", ctypes.c_int: "int", ctypes.c_int64: "int64_t",
             ctypes.c_uint64: "uint64_t", ctypes.c_float: "float", ctypes.c_double: "double"}
    seen = 0
    for m in re.finditer(r"\b(mfc_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        name, args = m.group(1), " ".join(m.group(2).split())
        decl = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        sig = _lib.SIGNATURES[name][1]
        assert len(sig) == len(decl), (name, decl, sig)
        for d, ty in zip(decl, sig):
            want = kinds[ty]
            if "*" in d:
                assert want == "ptr", (name, d, ty)
            else:
                assert d.split()[-2] == want or (want == "int" and d.split()[-2] in ("int", "unsigned")), (name, d, ty)
        seen += 1
    assert seen == len(_lib.SIGNATURES)


def test_no_gpu_entry_points():
    l = _lib.lib()
    assert l.mfc_abi_version() == 3          # MFC_ABI_VERSION of include/mfc.h
    assert b"gfx950" in l.mfc_build_info()
    # reference frame-count rule, preprocessing/mdct.py:491
    assert l.mfc_mdct_num_frames(196608, 512, 256) == 767
    assert l.mfc_mdct_num_frames(100, 256, 128) == 1
    assert l.mfc_mdct_num_frames(784, 512, 256) == 2
    assert l.mfc_mdct_out_len(767, 512, 256) == 197120
    assert l.mfc_mdct_num_frames(10, 0, 1) < 0


def test_argument_checks_fail_loudly_without_launching():
    l = _lib.lib()
    assert l.mfc_mdct_fwd(None, 1, 10, 10, 8, 4, None, None) == -14
    assert l.mfc_gemm(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, ctypes.c_float(1.0), None, 0,
                      ctypes.c_float(0.0), 1, None, None, None) == -14


def test_adamw_multi_argument_checks():
    l = _lib.lib()
    f = ctypes.c_float
    assert l.mfc_adamw_multi(-1, None, f(1), f(1e-3), f(.9), f(.999), f(1e-8), f(0), 1, None) == -22
    assert l.mfc_adamw_multi(2, None, f(1), f(1e-3), f(.9), f(.999), f(1e-8), f(0), 1, None) == -14
    items = (_lib.AdamwItem * 1)()            # null pointers inside a descriptor
    assert l.mfc_adamw_multi(1, ctypes.addressof(items), f(1), f(1e-3), f(.9), f(.999), f(1e-8), f(0), 1, None) == -14
    assert ctypes.sizeof(_lib.AdamwItem) == 56
