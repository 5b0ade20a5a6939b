"""GPU parity of the AdaLN kernels (mfc_adaln_fwd / mfc_adaln_bwd; reference models/mlp_flow.py:96-110 and the AdaLN of
models/mlp_mixer.py:102-163) against an fp64 restatement built from the oracle's layer_norm: every kernel family -- one
workgroup per row (wide rows), the narrow-row kernels (a row held by 1..64 lanes: the Mixer's 16-channel tokens and its
encoder's 256-channel tokens) -- with per-row modulation (mod_div = 1) and per-sample modulation broadcast over tokens
(mod_div = tokens, incl. a ragged last group)."""
import pytest
import torch

from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu

# (rows, n_tan, W, mod_div)
CASES = [(300, 100, 784, 1),            # MLP flow: wide rows, per-row modulation, one workgroup per row
         (1500, 0, 1040, 1),            # wide, >= 1024 rows
         (4096, 2048, 16, 1024),        # Mixer tokens: narrow rows (one lane = 16 bytes), per-sample modulation
         (2176, 0, 256, 544),           # Mixer encoder: 1 KB rows held by a whole wave, 4 groups of 544 tokens
         (2000, 0, 256, 544),           # ragged last group
         (1088, 544, 256, 544),         # tangent rows
         (600, 0, 256, 200)]            # < 1024 rows: the per-row kernels with shared modulation


def _rel(a, b):
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _mods(rows, W, mod_div, g, dtype):
    groups = (rows + mod_div - 1) // mod_div
    sc = (0.3 * torch.randn(groups, W, generator=g)).to(dtype)
    sh = (0.3 * torch.randn(groups, W, generator=g)).to(dtype)
    return sc, sh


def _expand(m, rows, mod_div):
    return m.double().repeat_interleave(mod_div, 0)[:rows]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("R,n_tan,W,mod_div", CASES)
def test_adaln_forward_and_tangent(dtype, tol, R, n_tan, W, mod_div):
    from meanflow_audio_codec_amd import ops
    g = torch.Generator().manual_seed(R + W)
    x = torch.randn(R + n_tan, W, generator=g).to(dtype)
    sc, sh = _mods(R + n_tan, W, mod_div, g, dtype)       # rows >= R carry the tangents of (x, scale, shift)
    y = ops.adaln_fwd(x.cuda(), sc.cuda(), sh.cuda(), act_rows=R, mod_div=mod_div)
    xq, scq, shq = x.double(), _expand(sc, R + n_tan, mod_div), _expand(sh, R + n_tan, mod_div)
    f = lambda x_, sc_, sh_: (1.0 + sc_) * fo.layer_norm(x_) + sh_
    ref = f(xq[:R], scq[:R], shq[:R])
    assert _rel(y[:R], ref) < tol
    if n_tan:
        # tangent row R + i belongs to primal row i; its modulation rows follow the same (row / mod_div) rule
        _, jv = torch.func.jvp(f, (xq[:n_tan], scq[:n_tan], shq[:n_tan]), (xq[R:], scq[R:], shq[R:]))
        assert _rel(y[R:], jv) < tol * 3


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("R,n_tan,W,mod_div", CASES)
def test_adaln_reverse(dtype, tol, R, n_tan, W, mod_div):
    from meanflow_audio_codec_amd import ops
    g = torch.Generator().manual_seed(7 * R + W)
    x = torch.randn(R, W, generator=g).to(dtype)
    dy = torch.randn(R, W, generator=g).to(dtype)
    sc, _ = _mods(R, W, mod_div, g, dtype)
    groups = sc.shape[0]
    xq = x.double().requires_grad_(True)
    scg = sc.double().requires_grad_(True)
    shg = torch.zeros(groups, W, dtype=torch.float64, requires_grad=True)
    y = (1.0 + scg.repeat_interleave(mod_div, 0)[:R]) * fo.layer_norm(xq) + shg.repeat_interleave(mod_div, 0)[:R]
    (y * dy.double()).sum().backward()
    mdt = dtype if mod_div == 1 else torch.float32          # shared modulation: fp32 [groups, W] sums (mfc.h)
    dsc = torch.full((groups, W), 7.0, device="cuda", dtype=mdt)
    dsh = torch.full((groups, W), 7.0, device="cuda", dtype=mdt)
    dx = ops.adaln_bwd(x.cuda(), sc.cuda(), dy.cuda(), dsc, dsh, mod_div=mod_div)
    assert _rel(dx, xq.grad) < tol
    assert _rel(dsc, scg.grad) < tol and _rel(dsh, shg.grad) < tol
    dsc2, dsh2 = torch.empty_like(dsc), torch.empty_like(dsh)
    dx2 = ops.adaln_bwd(x.cuda(), sc.cuda(), dy.cuda(), dsc2, dsh2, mod_div=mod_div)
    assert torch.equal(dx, dx2) and torch.equal(dsc, dsc2) and torch.equal(dsh, dsh2)       # fixed-order sums
