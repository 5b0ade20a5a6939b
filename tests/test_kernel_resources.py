"""CPU: register / scratch budget of the built kernels, read from the code objects inside libmfc.so
(tools/kernel_resources.py).  A kernel that starts spilling, or whose accumulators fall into scratch because a loop
stopped unrolling, still passes every numerics test -- it is just several times slower (seen once in round 2: the
192-row GEMM tile went from 5.7 to 15.2 ms per step).  The occupancy tiers below are the ones DESIGN.md's numbers
were measured at (512 VGPRs per SIMD lane: <= 128 -> 4 waves, <= 168 -> 3, <= 256 -> 2)."""
import subprocess
import sys
import pathlib

import pytest

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tools.kernel_resources import kernel_resources  # noqa: E402
from meanflow_audio_codec_amd import _build  # noqa: E402


@pytest.fixture(scope="module")
def res():
    _build.build(verbose=False)
    raw = kernel_resources()
    names = subprocess.run(["c++filt"], input="\n".join(raw), capture_output=True, text=True, check=True).stdout.split("\n")
    return {n.replace("(anonymous namespace)::", "").replace("void ", "", 1).split("(")[0]: e for n, e in zip(names, raw.values())}


def test_no_vector_spills_and_no_scratch_in_bf16_kernels(res):
    assert len(res) > 100
    for k, e in res.items():
        assert e["vgpr_spill"] == 0, (k, e)
        if "float" not in k:              # fp32-storage variants (parity tests only) may spill a few SGPRs
            assert e["scratch"] == 0, (k, e)


@pytest.mark.parametrize("kernel,max_vgpr", [
    ("cnx_fwd_kernel<unsigned short, false, 0>", 128), ("cnx_fwd_kernel<unsigned short, false, 1>", 128),
    ("cnx_bwd_kernel<unsigned short, 0>", 128), ("cnx_fwd_kernel<unsigned short, true, 0>", 168),
    ("cnx_bwd_conv_kernel<unsigned short>", 168),      # + 51.7 KB of LDS: three workgroups per CU
    ("cnx_fwd_kernel<unsigned short, true, 1>", 256),
    ("cnx_bwd_kernel<unsigned short, 1>", 256),
    ("gemm_kernel<unsigned short, 64, true, false, 128>", 168),     # weight gradient + fused AdamW (the dominant kernel)
    ("gemm_kernel<unsigned short, 64, false, true, 128>", 168),
    ("gemm_kernel<unsigned short, 64, false, false, 64>", 128), ("gemm_kernel<unsigned short, 64, true, false, 64>", 128),
    ("gemm_kernel<unsigned short, 64, false, false, 192>", 256),
    ("gemm_nstream_kernel<3, false>", 256), ("gemm_nstream_kernel<1, false>", 128), ("gemm_nstream_kernel<2, true>", 168),
    ("m512::mdct512_fwd_kernel<true, 5>", 128), ("m512::mdct512_inv_kernel<true>", 168),
    # BASELINE config #3: fused channel MLP (forward: four 4-wave workgroups per CU; reverse: one 8-wave workgroup) and the
    # fp32 tiled GEMM with branch-free staging (two workgroups per CU)
    ("chanmlp_fwd_kernel<float>", 128), ("chanmlp_bwd_kernel<float, 8, 2>", 256), ("chanmlp_bwd_kernel<unsigned short, 8, 2>", 256),
    ("gemm_f32_fast_kernel<false, false, 128>", 256), ("gemm_f32_fast_kernel<true, false, 128>", 256),
    ("gemm_f32_fast_kernel<false, true, 128>", 256),
    # the bf16 products without an optimizer epilogue (K = S / K = D products; un-fused weight gradients of the DP schedule)
    ("gemm_bf16_fast_kernel<false, false, 192>", 256), ("gemm_bf16_fast_kernel<false, false, 128>", 168),
    ("gemm_bf16_fast_kernel<false, true, 128>", 168), ("gemm_bf16_fast_kernel<true, false, 128>", 168),
    ("gemm_bf16_fast_kernel<false, false, 64>", 128),
])
def test_occupancy_tier_of_the_hot_kernels(res, kernel, max_vgpr):
    assert kernel in res, sorted(res)[:5]
    e = res[kernel]
    assert e["vgpr"] + e["agpr"] <= max_vgpr, (kernel, e)
