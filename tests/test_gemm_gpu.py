"""GPU numerics: mfc_gemm (MFMA) vs torch fp64 matmul on the same inputs."""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(A, B, ta, tb, bias, bias_rows, alpha, R, beta):
    A64 = (A.t() if ta else A).double()
    B64 = (B.t() if tb else B).double()
    C = A64 @ B64
    if bias is not None:
        C[:bias_rows] += bias.double()[None, :]
    C = alpha * C
    if R is not None:
        C = C + beta * R.double()
    return C


SHAPES = [(128, 128, 32), (4, 16, 20), (130, 257, 100), (256, 384, 96), (37, 515, 1030), (300, 64, 4099),
          (1, 1, 1), (129, 129, 33), (192, 160, 72), (64, 128, 64), (448, 48, 200), (200, 32, 40)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ta,tb", list(itertools.product([False, True], repeat=2)))
def test_gemm_all_layouts(dtype, ta, tb):
    from meanflow_audio_codec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    for (M, N, K) in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), generator=g, device="cuda").to(dtype)
        B = torch.randn((N, K) if tb else (K, N), generator=g, device="cuda").to(dtype)
        bias = torch.randn(N, generator=g, device="cuda")
        R = torch.randn(M, N, generator=g, device="cuda").to(dtype)
        C = ops.gemm(A, B, trans_a=ta, trans_b=tb, bias=bias, bias_rows=max(1, M // 2), alpha=0.5,
                     residual=R, beta=2.0)
        ref = _ref(A, B, ta, tb, bias, max(1, M // 2), 0.5, R, 2.0)
        err = (C.double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        tol = (1e-5 if dtype == torch.float32 else 1e-2) * max(scale, 1.0)
        assert err <= tol, (M, N, K, ta, tb, dtype, err, scale)


FAST_SHAPES = [(130, 260, 128), (37, 516, 1024), (300, 64, 4096), (1, 4, 64), (129, 132, 192), (192, 160, 64),
               (448, 48, 256), (200, 40, 640), (256, 384, 128), (65, 1040, 832)]


@pytest.mark.parametrize("ta,tb", list(itertools.product([False, True], repeat=2)))
def test_gemm_f32_whole_ksteps_fast_staging(ta, tb):
    """fp32 products whose K is a whole number of 64-deep steps and whose operands are 16-byte aligned take the branch-free
    buffer-resource staging (gemm_f32_fast_kernel): ragged M / N tiles (rows past the operand read zeros or a neighbour's
    values and must not reach C), strided operands whose padding holds NaN, split-K, accumulate -- against fp64."""
    from meanflow_audio_codec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(2)
    for (M, N, K) in FAST_SHAPES:
        ash, bsh = ((K, M) if ta else (M, K)), ((N, K) if tb else (K, N))
        # operands are views into NaN-filled buffers with a wider (16-byte aligned) leading dimension
        Abig = torch.full((ash[0], ash[1] + 8), float("nan"), device="cuda")
        Bbig = torch.full((bsh[0], bsh[1] + 12), float("nan"), device="cuda")
        A, B = Abig[:, :ash[1]], Bbig[:, :bsh[1]]
        A.copy_(torch.randn(ash, generator=g, device="cuda"))
        B.copy_(torch.randn(bsh, generator=g, device="cuda"))
        bias = torch.randn(N, generator=g, device="cuda")
        R = torch.randn(M, N, generator=g, device="cuda")
        ref = _ref(A, B, ta, tb, bias, max(1, M // 2), 0.5, R, 2.0)
        for sk in (1, 2) if K >= 256 else (1,):
            C = ops.gemm(A, B, trans_a=ta, trans_b=tb, bias=bias, bias_rows=max(1, M // 2), alpha=0.5, residual=R, beta=2.0,
                         splitk=sk)
            assert torch.isfinite(C).all(), (M, N, K, ta, tb, sk)
            err = (C.double() - ref).abs().max().item()
            assert err <= 1e-5 * max(ref.abs().max().item(), 1.0), (M, N, K, ta, tb, sk, err)
        C0 = torch.randn(M, N, generator=g, device="cuda")
        C1 = ops.gemm(A, B, trans_a=ta, trans_b=tb, out=C0.clone(), accumulate=True)
        ref1 = (A.t() if ta else A).double() @ (B.t() if tb else B).double() + C0.double()
        assert (C1.double() - ref1).abs().max().item() <= 1e-5 * max(ref1.abs().max().item(), 1.0), (M, N, K, ta, tb)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_splitk_gelu_tangent_rows(dtype):
    """Row-stacked [x; xdot]: primal rows get bias + GELU, tangent rows t*gelu'(pre)."""
    from meanflow_audio_codec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    Rr, K, N = 24, 5000, 128
    X = torch.randn(2 * Rr, K, generator=g, device="cuda").to(dtype) * 0.05
    W = torch.randn(K, N, generator=g, device="cuda").to(dtype)
    b = torch.randn(N, generator=g, device="cuda")
    for splitk in (1, 7, 64):
        C = ops.gemm(X, W, bias=b, bias_rows=Rr, gelu=True, act_rows=Rr, splitk=splitk)
        x64 = X.double().requires_grad_(False)
        pre = x64[:Rr] @ W.double() + b.double()
        tan = x64[Rr:] @ W.double()
        prim = torch.nn.functional.gelu(pre, approximate="tanh")
        pre_g = pre.clone().requires_grad_(True)
        (torch.nn.functional.gelu(pre_g, approximate="tanh")).backward(torch.ones_like(pre_g))
        ref = torch.cat([prim, tan * pre_g.grad], 0)
        err = (C.double() - ref).abs().max().item()
        tol = 2e-5 if dtype == torch.float32 else 3e-2
        assert err <= tol * max(1.0, ref.abs().max().item()), (splitk, dtype, err)


def test_gemm_accumulate_and_strides():
    from meanflow_audio_codec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(2)
    A = torch.randn(70, 200, generator=g, device="cuda")[:, :150]   # lda=200, K=150
    B = torch.randn(150, 90, generator=g, device="cuda")[:, 3:80]   # unaligned base -> scalar loads
    B = B if B.stride(1) == 1 else B.contiguous()
    C0 = torch.randn(70, 77, generator=g, device="cuda")
    C = C0.clone()
    ops.gemm(A, B, out=C, accumulate=True)
    ref = C0.double() + A.double() @ B.double()
    assert (C.double() - ref).abs().max().item() < 1e-4


def test_gemm_rejects_cpu_tensors():
    from meanflow_audio_codec_amd import ops
    from meanflow_audio_codec_amd._lib import MfcError
    with pytest.raises(MfcError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_gemm_ln16_epilogue(dtype, tol):
    """MFC_GEMM_LN16: primal rows get LayerNorm over every 16-column group (+ 1/sigma out), tangent rows stay raw."""
    from meanflow_audio_codec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    R, nt, K, N = 20, 7, 128, 16 * 37
    X = torch.randn(R + nt, K, generator=g, device="cuda").to(dtype)
    W = torch.randn(K, N, generator=g, device="cuda").to(dtype)
    b = torch.randn(N, generator=g, device="cuda")
    rho = torch.zeros(R, N // 16, device="cuda")
    C = ops.gemm(X, W, bias=b, bias_rows=R, ln_rstd=rho)
    pre = X.double() @ W.double()
    pre[:R] += b.double()
    grp = pre[:R].reshape(R, N // 16, 16)
    mu, var = grp.mean(-1, keepdim=True), grp.var(-1, unbiased=False, keepdim=True)
    ref = torch.cat([((grp - mu) * torch.rsqrt(var + 1e-6)).reshape(R, N), pre[R:]], 0)
    assert (C.double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert ((rho.double() - torch.rsqrt(var + 1e-6).reshape(R, -1)).abs().max() /
            torch.rsqrt(var + 1e-6).max()).item() < (1e-4 if dtype == torch.float32 else 2e-2)


NS_SHAPES = [(192, 1040), (64, 64), (256, 3200), (250, 328), (16, 16), (130, 72)]


@pytest.mark.parametrize("M,N", NS_SHAPES)
def test_gemm_nstream_shapes(M, N):
    """bf16 NN products with K = 128 and M <= 256 take the N-streaming kernel (A in registers):
    bias on the first rows, alpha, residual, accumulate -- against fp64."""
    from meanflow_audio_codec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    dtype, K = torch.bfloat16, 128
    A = torch.randn(M, K, generator=g, device="cuda").to(dtype)
    B = torch.randn(K, N, generator=g, device="cuda").to(dtype)
    bias = torch.randn(N, generator=g, device="cuda")
    R = torch.randn(M, N, generator=g, device="cuda").to(dtype)
    br = max(1, (2 * M) // 3)
    ref = _ref(A, B, False, False, bias, br, 0.5, R, -1.5)
    C = ops.gemm(A, B, bias=bias, bias_rows=br, alpha=0.5, residual=R, beta=-1.5)
    assert (C.double() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    C0 = torch.randn(M, N, generator=g, device="cuda").to(dtype)
    C1 = ops.gemm(A, B, out=C0.clone(), accumulate=True)
    ref1 = A.double() @ B.double() + C0.double()
    assert (C1.double() - ref1).abs().max().item() <= 2e-2 * max(1.0, ref1.abs().max().item())


def _ln_ref(pre, R, N):
    grp = pre[:R].reshape(R, N // 16, 16)
    mu, var = grp.mean(-1, keepdim=True), grp.var(-1, unbiased=False, keepdim=True)
    rho = torch.rsqrt(var + 1e-6)
    return (grp - mu) * rho, rho


@pytest.mark.parametrize("M,N", NS_SHAPES + [(128, 64 * 37 + 48)])
def test_gemm_nstream_nt_shapes(M, N):
    """bf16 NT products with K = 128, M <= 256 and a dense [N, 128] B (the dX products of the ConvFlow block) take the NT
    form of the N-streaming kernel: alpha, residual, accumulate, ragged last tile -- against fp64; and bitwise against
    the tiled kernel (MFC_GEMM_NSTREAM=0 is read once per process, so that comparison uses a strided B, which the
    N-streaming dispatch declines)."""
    from meanflow_audio_codec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(13)
    dtype, K = torch.bfloat16, 128
    A = torch.randn(M, K, generator=g, device="cuda").to(dtype)
    B = torch.randn(N, K, generator=g, device="cuda").to(dtype)
    R = torch.randn(M, N, generator=g, device="cuda").to(dtype)
    ref = 0.5 * (A.double() @ B.double().T) - 1.5 * R.double()
    C = ops.gemm(A, B, trans_b=True, alpha=0.5, residual=R, beta=-1.5)
    assert (C.double() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    C0 = torch.randn(M, N, generator=g, device="cuda").to(dtype)
    C1 = ops.gemm(A, B, trans_b=True, out=C0.clone(), accumulate=True)
    ref1 = A.double() @ B.double().T + C0.double()
    assert (C1.double() - ref1).abs().max().item() <= 2e-2 * max(1.0, ref1.abs().max().item())
    # the same product through the tiled kernel (a B with leading dimension 136 is not dense: no N-streaming)
    Bs = torch.zeros(N, K + 8, device="cuda", dtype=dtype)[:, :K]
    Bs.copy_(B)
    C2 = ops.gemm(A, Bs, trans_b=True)
    C3 = ops.gemm(A, B, trans_b=True)
    assert (C2.double() - C3.double()).abs().max().item() <= 1e-2 * max(1.0, C3.double().abs().max().item())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("R,nt,N", [(32, 16, 16 * 37), (128, 64, 16 * 40), (48, 48, 16 * 9), (20, 7, 16 * 5)])
def test_gemm_ln16_tangent_rows(dtype, tol, R, nt, N):
    """MFC_GEMM_LN16 | MFC_GEMM_LN16T: primal rows LayerNorm'd, tangent rows (R + j <-> j) get the tangent of
    that LayerNorm.  (bf16 with 16-aligned R: fused in the N-streaming kernel; otherwise a second kernel.)"""
    from meanflow_audio_codec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    K = 128
    X = torch.randn(R + nt, K, generator=g, device="cuda").to(dtype)
    W = (torch.randn(K, N, generator=g, device="cuda") / 8).to(dtype)
    b = torch.randn(N, generator=g, device="cuda")
    rho = torch.zeros(R, N // 16, device="cuda")
    C = ops.gemm(X, W, bias=b, bias_rows=R, ln_rstd=rho, ln_tangent=True)
    pre = X.double() @ W.double()
    pre[:R] += b.double()
    n, rr = _ln_ref(pre, R, N)
    xd = pre[R:].reshape(nt, N // 16, 16)
    xc = xd - xd.mean(-1, keepdim=True)
    nd = rr[:nt] * (xc - n[:nt] * (n[:nt] * xc).mean(-1, keepdim=True))
    ref = torch.cat([n.reshape(R, N), nd.reshape(nt, N)], 0)
    assert (C[:R].double() - ref[:R]).abs().max().item() <= tol * max(1.0, ref[:R].abs().max().item())
    assert (C[R:].double() - ref[R:]).abs().max().item() <= tol * max(1.0, ref[R:].abs().max().item())
    assert ((rho.double() - rr.reshape(R, -1)).abs().max() / rr.max()).item() < (1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("M,N,K", [(300, 64, 128), (128, 4000, 128), (64, 32, 40), (1000, 128, 96)])
def test_gemm_adamw_fused_equals_gemm_then_adamw(M, N, K):
    """mfc_gemm_adamw (weight gradient with the AdamW update as its epilogue) == mfc_gemm -> bf16 gradient -> mfc_adamw,
    bit for bit, over several steps."""
    from meanflow_audio_codec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(21)
    X = (torch.randn(K, M, generator=g, device="cuda") * 0.3).bfloat16()      # dW = X^T dY
    pa = torch.randn(M, N, generator=g, device="cuda")
    ma, va = torch.zeros_like(pa), torch.zeros_like(pa)
    pb, mb, vb = pa.clone(), ma.clone(), va.clone()
    wa, wb = pa.bfloat16(), pb.bfloat16()
    for step in range(1, 4):
        dY = (torch.randn(K, N, generator=g, device="cuda") * 0.1).bfloat16()
        grad = ops.gemm(X, dY, trans_a=True, alpha=0.5)
        ops.adamw(pa, grad, ma, va, lr=1e-2, wd=1e-2, step=step, p_bf16=wa)
        ops.gemm_adamw(X, dY, trans_a=True, grad_scale=0.5, p=pb, m=mb, v=vb, p_bf16=wb, lr=1e-2, wd=1e-2, step=step)
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(wa, wb), step
    assert (pa - torch.randn(M, N, generator=torch.Generator(device="cuda").manual_seed(21), device="cuda")).abs().max() >= 0


@pytest.mark.parametrize("M,N,K", [(300, 64, 128), (128, 4000, 130), (1000, 128, 96), (256, 144, 33)])
def test_gemm_adamw_bias_gradient_output(M, N, K):
    """The ``colsum`` output of mfc_gemm_adamw (the Dense layer's bias gradient, scale x column sums of dY, taken from the
    B tiles the product stages): equals the standalone mfc_colsum to fp32 rounding, leaves the fused update bit-identical,
    and is itself reproducible bit for bit.  K <= 32 selects the 32-deep kernel, which has no such output: ENOSYS."""
    from meanflow_audio_codec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    X = (torch.randn(K, M, generator=g, device="cuda") * 0.3).bfloat16()
    dY = (torch.randn(K, N, generator=g, device="cuda") * 0.1).bfloat16()
    p0 = torch.randn(M, N, generator=g, device="cuda")
    outs = []
    for cs in (None, torch.full((N,), float("nan"), device="cuda"), torch.full((N,), 7.0, device="cuda")):
        p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        w = p.bfloat16()
        ops.gemm_adamw(X, dY, trans_a=True, grad_scale=0.25, p=p, m=m, v=v, p_bf16=w, lr=1e-2, wd=1e-2, step=1, colsum=cs,
                       colsum_scale=0.25)
        outs.append((p, m, v, w, cs))
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(a, b)
    assert torch.equal(outs[1][4], outs[2][4])                                   # overwritten, not accumulated; bitwise
    ref = 0.25 * dY.double().sum(0)
    assert (outs[1][4].double() - ref).abs().max().item() <= 1e-6 * max(1.0, dY.double().abs().sum(0).max().item())
    sep = ops.colsum(dY, scale=0.25)
    assert (outs[1][4] - sep).abs().max().item() <= 1e-6 * max(1.0, dY.double().abs().sum(0).max().item())


def test_gemm_adamw_bias_gradient_needs_the_deep_kernel():
    from meanflow_audio_codec_amd import ops
    X = torch.zeros(32, 64, device="cuda", dtype=torch.bfloat16)
    dY = torch.zeros(32, 64, device="cuda", dtype=torch.bfloat16)
    p = torch.zeros(64, 64, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemm_adamw(X, dY, trans_a=True, p=p, m=p.clone(), v=p.clone(), p_bf16=p.bfloat16(), lr=1e-3, wd=0.0, step=1,
                       colsum=torch.zeros(64, device="cuda"))
