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


def test_no_gpu_entry_points():
    l = _lib.lib()
    assert l.mfc_abi_version() >= 1
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
