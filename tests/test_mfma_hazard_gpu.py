"""GPU: the mixed-shape MFMA accumulate-chain hazard, as a regression test.

``tools/probe/mfma_mixed_shape.hip`` chains v_mfma_f32_16x16x32_bf16 and v_mfma_f32_16x16x16_bf16 on one accumulator
with exact integer data.  Measured on MI355X / ROCm 7.2: back to back (what hipcc emits) the chain returns wrong rows in
BOTH directions; with >= 5 wait states between the two shapes it is exact.  ``csrc/convnext.hip`` therefore fences its
one K16 -> K32 link (border tiles, and the folded-FiLM bias of ``RowW::set``) with ``mfma_shape_fence``.  The test pins
the part the kernels rely on -- fenced chains are exact -- and reports whether the toolchain still needs the fence."""
import pathlib
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_fenced_mixed_shape_chains_are_exact(tmp_path):
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if pathlib.Path("/opt/rocm/bin/hipcc").exists() else None)
    if hipcc is None:
        pytest.skip("hipcc not available on this box")
    exe = tmp_path / "mfma_mixed"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-w",
                    str(ROOT / "tools" / "probe" / "mfma_mixed_shape.hip"), "-o", str(exe)], check=True, timeout=600)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300).stdout
    rows = {}
    for line in out.splitlines():
        line = line.strip()
        if line.startswith("s_nop") or line.startswith("none"):
            key = line.split(":")[0].strip()
            nums = [int(t) for t in line.replace(":", " ").split() if t.isdigit()]
            rows[key] = nums[-3:]
    assert "none" in rows and any(k.startswith("s_nop") for k in rows), out
    for key, (a, b, both) in rows.items():
        if key.startswith("s_nop"):
            n = int(key.split()[1])
            if n >= 4:          # s_nop 4 = 5 wait states of our own (the asm boundary adds one more)
                assert (a, b, both) == (0, 0, 0), (key, a, b, both)
    needs_fence = rows["none"][2] != 0
    print(f"mixed-shape chain without wait states: {rows['none']} wrong elements -> fence "
          f"{'REQUIRED (toolchain emits no wait states)' if needs_fence else 'no longer required by this toolchain'}")


def test_border_tile_conv_path_matches_oracle():
    """The kernel-level consumer of the fence: images smaller than one 16 x 16 tile make every tile a border tile, and
    the bf16 forward (K = 32 conv steps after the K = 16 shift taps) must agree with the fp64 oracle."""
    import torch
    from meanflow_audio_codec_amd import ops
    from oracle import flow_oracle as fo
    torch.manual_seed(0)
    R, s = 3, 11
    g = torch.Generator().manual_seed(4)
    w64 = {"conv_w": torch.randn(3, 3, 16, 16, generator=g, dtype=torch.float64) * 0.2,
           "conv_b": torch.randn(16, generator=g, dtype=torch.float64) * 0.1,
           "exp_w": torch.randn(16, 32, generator=g, dtype=torch.float64) * 0.3,
           "exp_b": torch.randn(32, generator=g, dtype=torch.float64) * 0.1,
           "grn_gamma": torch.randn(32, generator=g, dtype=torch.float64) * 0.5,
           "grn_beta": torch.randn(32, generator=g, dtype=torch.float64) * 0.1,
           "con_w": torch.randn(32, 16, generator=g, dtype=torch.float64) * 0.3,
           "con_b": torch.randn(16, generator=g, dtype=torch.float64) * 0.1,
           "ls": torch.randn(16, generator=g, dtype=torch.float64) * 0.5}
    h0 = torch.randn(R, s, s, 16, generator=g, dtype=torch.float64)
    sc, sh = torch.randn(R, 16, generator=g, dtype=torch.float64) * 0.3, torch.randn(R, 16, generator=g, dtype=torch.float64) * 0.3
    wd = {k: (v.bfloat16() if k.endswith("_w") else v.float()).cuda().contiguous() for k, v in w64.items()}
    wq = {k: wd[k].double().cpu() for k in wd}
    h1, _ = ops.ln16(h0.bfloat16().cuda().contiguous())
    o, _, _, _ = ops.cnx_forward(h1, sc.float().cuda(), sh.float().cuda(), wd, s)
    # oracle on the same (bf16-rounded) inputs: FiLM(h1) -> ConvNeXt block
    h1q = h1.double().cpu()
    h2 = (1.0 + sc[:, None, None, :]) * h1q + sh[:, None, None, :]
    p = {"Conv_0": {"kernel": wq["conv_w"], "bias": wq["conv_b"]}, "Conv_1": {"kernel": wq["exp_w"].reshape(1, 1, 16, 32), "bias": wq["exp_b"]},
         "GlobalResponseNormalization_0": {"gamma": wq["grn_gamma"], "beta": wq["grn_beta"]},
         "Conv_2": {"kernel": wq["con_w"].reshape(1, 1, 32, 16), "bias": wq["con_b"]}, "layer_scale_gamma": wq["ls"]}
    ref = fo.convnext_block(p, h2)
    err = ((o.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < 4e-2, err
