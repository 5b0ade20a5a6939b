"""CPU: the bench's work model and symbol mapping (no GPU needed)."""
import json
import math
import pathlib

import bench


def test_flop_model_matches_survey():
    # 48.16 GFLOP fwd per sample at D = 392704; CI shape 3.94 GFLOP at D = 32256
    assert abs(bench.conv_flow_flops_fwd(767 * 512) / 1e9 - 48.16) < 0.05
    assert abs(bench.conv_flow_flops_fwd(63 * 512) / 1e9 - 3.94) < 0.02


def test_algorithmic_bytes():
    # MDCT forward per clip: 4T + 4 n_frames N = 2 357 248 B
    nb, fl, _ = bench.algorithmic_work("mfc_mdct_fwd", (1, 196608, 196608, 512, 256), ())
    assert nb == 2357248
    nb, _, _ = bench.algorithmic_work("mfc_mdct_inv", (1, 767, 512, 256, 197120), ())
    assert nb == 2359296
    # AdamW with bf16 gradient and bf16 working copy: 28 B / parameter
    nb, _, _ = bench.algorithmic_work("mfc_adamw", (1, 1000, 3), (True, True, True, True, True))
    assert nb == 28000
    nb, fl, dt = bench.algorithmic_work("mfc_gemm", (1, 0, 128, 6270016, 128) + (0,) * 7, (True,) * 9)
    assert fl == 2.0 * 128 * 6270016 * 128 and dt == 1


def test_symbols_and_traffic_table():
    s = bench.symbol_of("mfc_adamw", (1, 802562048, 3), (True,) * 5)
    assert s == "adamw_vec_kernel<unsigned short, false>"
    tab = json.loads((pathlib.Path(bench.ROOT) / "profiles" / bench.PMC_TABLE).read_text())
    assert s in tab and tab[s]["avg_hbm_bytes_per_launch"] > 1e9
    assert bench.measured_traffic(s) == tab[s]["avg_hbm_bytes_per_launch"]
    assert bench.symbol_of("mfc_cnx_bwd_main", (1, 128, 626), (True,) * 8) == "cnx_bwd_kernel<unsigned short, 1>"
    assert bench.symbol_of("mfc_gemm", (0, 3, 4, 4, 16) + (0,) * 7, ()) == "gemm_kernel<float, 32, true, true, 64>"
    assert bench.symbol_of("mfc_gemm", (1, 16, 192, 6270016, 128) + (0,) * 7, ()) == "gemm_nstream_kernel<3, false>"


def test_auto_splitk_is_a_pure_function_of_the_shape():
    """models/common.py::auto_splitk: the slab count decides the (fixed) summation order of a split-K product, so it
    must depend on the shape only; one- and two-tile outputs take the measured optimum of 512 slices."""
    from meanflow_audio_codec_amd.models.common import auto_splitk
    S, D = 6270016, 392704
    assert auto_splitk(128, 128, S) == 512 and auto_splitk(192, 128, S) == 512 and auto_splitk(64, 128, D) == 512
    assert auto_splitk(128, 128, 400) == 1                       # short K: no split
    assert auto_splitk(128, 1040, 1040) == 8 and auto_splitk(128, 128, 1000) == 7   # small nets: >= 128 deep per slice
    assert auto_splitk(128, 128, 4096) == 16                     # never more slices than 256-deep chunks
    assert auto_splitk(128, S, 128) == 1 and auto_splitk(4096, 8192, 100000) == 1   # plenty of output tiles already
    assert auto_splitk(384, 128, S) == (1024 + 2) // 3          # three tiles: ~1024 workgroups in total
