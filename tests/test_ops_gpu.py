"""Per-kernel parity: every C-ABI op against torch CPU fp32 ops on the same seeded inputs.

Tolerances: forward ops  |got-ref| <= 1e-4*max|ref| + 1e-5,
gradients 1e-3 relative to the tensor max-norm (we hold them to 2e-4 in practice).
"""
import math

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

from discogan_modernized_amd import _lib, ops  # noqa: E402

DEV = "cuda"


def experiments_built():
    """The kernels that lost their same-box A/B (window forward kernel, register-staged plane reader, persistent window input-grad,
    paired-plane 16x16x32 body) live in the experiments build of the library since round 4 (csrc/Makefile EXPERIMENTS=1); their tests
    run when that build is the one loaded: DG_LIB=discogan_modernized_amd/libdiscogan_hip_experiments.so pytest -m gpu"""
    return bool(_lib.load().dg_build_flags() & 1)


needs_experiments = pytest.mark.skipif("not experiments_built()", reason="experiments library not loaded (make -C csrc EXPERIMENTS=1; DG_LIB=...)")


def close(got, ref, rtol=1e-4, atol=1e-5, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs().max().item()
    bound = rtol * ref.abs().max().item() + atol
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e} (max|ref| {ref.abs().max().item():.3e})"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def nhwc(t):
    """CPU logical NCHW -> GPU tensor with NHWC memory."""
    return t.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def krsc(w):
    return ops.krsc_param(w.to(DEV))


# (N, C, K, H) interior stride-2 layers: small, ragged-M, split-K, C=64 path, wide
S2_SHAPES = [
    (2, 64, 128, 8),      # M=32 (one masked tile)
    (3, 64, 128, 16),     # M=192, non-power-of-two batch
    (2, 128, 256, 8),     # split-K in fwd
    (4, 256, 512, 8),
    (2, 512, 1024, 4),    # 4x4 -> 2x2
    (1, 2048, 2048, 8),   # deepest reference layer, N=1
    (8, 64, 128, 32),     # M=2048
    (2, 128, 64, 16),     # K=64 (dgrad of a 64->128 convT role)
]


@pytest.mark.parametrize("N,C,K,H", S2_SHAPES)
def test_conv_s2_fwd_dgrad_wgrad(N, C, K, H):
    x = rnd(N, C, H, H, seed=1)
    w = rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C))
    dy = rnd(N, K, H // 2, H // 2, seed=3)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = TF.conv2d(xr, wr, stride=2, padding=1)
    yr.backward(dy)
    xg, wg, dyg = nhwc(x), krsc(w), nhwc(dy)
    y = ops.conv_fwd(xg, wg, 2, 1)
    close(y, yr, what="conv_fwd")
    dx = ops.conv_dgrad(dyg, wg, (H, H), 2, 1)
    close(dx, xr.grad, rtol=2e-4, what="conv_dgrad")
    dw = ops.conv_wgrad(dyg, xg, 2, 1)
    close(dw, wr.grad, rtol=2e-4, what="conv_wgrad")
    # accumulate path
    dw2 = ops.conv_wgrad(dyg, xg, 2, 1, out=dw.clone(), accumulate=True)
    close(dw2, 2 * wr.grad, rtol=2e-4, what="conv_wgrad accumulate")


@pytest.mark.parametrize("kt,splitk", [(16, 0), (32, 3), (32, 1)])
def test_conv_s2_options(kt, splitk):
    """Force the other K-tile / split-K configurations through the same checks."""
    _lib.set_option("kt", kt)
    _lib.set_option("splitk", splitk)
    try:
        test_conv_s2_fwd_dgrad_wgrad(2, 128, 256, 8)
        test_conv_s2_fwd_dgrad_wgrad(3, 64, 128, 16)
    finally:
        _lib.set_option("kt", 0)
        _lib.set_option("splitk", 0)


@pytest.mark.parametrize("N,C,K,H", [(2, 128, 256, 8), (3, 64, 128, 16), (4, 256, 64, 16), (2, 512, 512, 4), (5, 64, 128, 64),
                                     (2, 1024, 2048, 8)])
def test_conv_bf16_operands(N, C, K, H):
    """Option "bf16": operands rounded to bf16 (RNE), bf16 MFMA, fp32 accumulate.  bf16 x bf16 products are exact in
    fp32, so the result must equal the fp32 op on the ROUNDED operands up to summation order (tolerance as fp32)."""
    r = lambda t_: t_.bfloat16().float()
    x, w, dy = rnd(N, C, H, H, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C)), rnd(N, K, H // 2, H // 2, seed=3)
    yr = TF.conv2d(r(x).double(), r(w).double(), stride=2, padding=1).float()
    dxr = TF.conv_transpose2d(r(dy).double(), r(w).double(), stride=2, padding=1).float()
    dwr = torch.nn.grad.conv2d_weight(r(x).double(), w.shape, r(dy).double(), stride=2, padding=1).float()
    _lib.set_option("bf16", 1)
    try:
        xg, wg, dyg = nhwc(x), krsc(w), nhwc(dy)
        close(ops.conv_fwd(xg, wg, 2, 1), yr, what="bf16 conv fwd")
        close(ops.conv_dgrad(dyg, wg, (H, H), 2, 1), dxr, rtol=2e-4, what="bf16 conv dgrad")
        close(ops.conv_wgrad(dyg, xg, 2, 1), dwr, rtol=2e-4, what="bf16 conv wgrad")
        # and it is NOT the fp32 result (the rounding is really applied)
        yf = TF.conv2d(x, w, stride=2, padding=1)
        assert (ops.conv_fwd(xg, wg, 2, 1).cpu() - yf).abs().max() > 1e-4 * yf.abs().max()
    finally:
        _lib.set_option("bf16", 0)


def _with_shadow(t):
    """Register a bf16 (RNE) shadow of t the way the producers do (ops.shadow_put / the Adam-owned weight shadow)."""
    t16 = torch.empty_like(t, dtype=torch.bfloat16, memory_format=torch.preserve_format)
    ops.f32_to_bf16(t, t16)
    ops.shadow_put(t, t16)
    t._dg_bf16, t._dg_bf16_ver = t16, t._version
    return t


# (N, C, K, H): shapes that reach the LDS-DMA kernel (igemm_dma.hip: both operands bf16 in HBM, GEMM rows and columns >= 192)
DMA_SHAPES = [
    (4, 128, 256, 16),     # fwd: one 256x256 tile, split-K 8; wgrad: 8 column tiles
    (3, 256, 320, 16),     # ragged column tile (320 = 256 + 64), ragged row tile (192); dgrad 4 parity classes, K = 5 chunks
    (13, 256, 256, 8),     # ragged rows (208) and a ragged reduction in the weight gradient (208 pixels = 3 K-tiles + 16)
    (2, 512, 512, 32),     # two row tiles, two column tiles
    (1, 1024, 1024, 32),   # long reduction (256 K-tiles), split-K
    (9, 192, 448, 16),     # C = 3 chunks of 64 (fwd), K = 7 chunks (dgrad), 576 rows = 2.25 tiles
]


@pytest.mark.parametrize("N,C,K,H", DMA_SHAPES)
@pytest.mark.parametrize("splitk,mfma", [(0, 16), (1, 16), (3, 16), (0, 32), (3, 32)])
def test_conv_bf16_lds_dma_kernel(N, C, K, H, splitk, mfma):
    """bf16 shadow operands -> igemm_dma.hip (LDS-DMA staging, XOR-swizzled LDS images, 256x256 tile).  bf16 x bf16 products
    are exact in fp32, so the result must equal the fp64 convolution of the ROUNDED operands at fp32 tolerance -- and agree
    with the register-staged bf16 tiles (option no_dma) to the same bound.  mfma: the 16x16x32 body (default) / the 32x32x16 body."""
    r = lambda t_: t_.bfloat16().float()
    x, w, dy = rnd(N, C, H, H, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C)), rnd(N, K, H // 2, H // 2, seed=3)
    yr = TF.conv2d(r(x).double(), r(w).double(), stride=2, padding=1).float()
    dxr = TF.conv_transpose2d(r(dy).double(), r(w).double(), stride=2, padding=1).float()
    dwr = torch.nn.grad.conv2d_weight(r(x).double(), w.shape, r(dy).double(), stride=2, padding=1).float()
    L = _lib.load()
    _lib.set_option("bf16", 1)
    _lib.set_option("splitk", splitk)
    _lib.set_option("dma_mfma", mfma)
    ops.SHADOW = True
    try:
        xg, wg, dyg = _with_shadow(nhwc(x)), _with_shadow(krsc(w)), _with_shadow(nhwc(dy))
        assert L.dg_conv_bf16_operands_ok(0, N, H, H, C, K, 2, 1) == 2 and L.dg_conv_bf16_operands_ok(2, N, H, H, C, K, 2, 1) == 2
        y = ops.conv_fwd(xg, wg, 2, 1)
        dx = ops.conv_dgrad(dyg, wg, (H, H), 2, 1)
        dw = ops.conv_wgrad(dyg, xg, 2, 1)
        close(y, yr, what="dma conv fwd")
        close(dx, dxr, rtol=2e-4, what="dma conv dgrad")
        close(dw, dwr, rtol=2e-4, what="dma conv wgrad")
        dw2 = ops.conv_wgrad(dyg, xg, 2, 1, out=dw.clone(), accumulate=True)
        close(dw2, 2 * dwr, rtol=2e-4, what="dma conv wgrad accumulate")
        torch.cuda.synchronize()
        _lib.set_option("no_dma", 1)
        close(ops.conv_fwd(xg, wg, 2, 1), y, what="register-staged vs dma fwd")
        close(ops.conv_dgrad(dyg, wg, (H, H), 2, 1), dx, rtol=2e-4, what="register-staged vs dma dgrad")
        close(ops.conv_wgrad(dyg, xg, 2, 1), dw, rtol=2e-4, what="register-staged vs dma wgrad")
    finally:
        ops.SHADOW = False
        ops.shadow_clear()
        _lib.set_option("no_dma", 0)
        _lib.set_option("dma_mfma", 0)
        _lib.set_option("splitk", 0)
        _lib.set_option("bf16", 0)


# (N, C, K, H): input-grad shapes of the bf16 window kernel (igemm_dma_dgw.hip): C <= 128, K % 64 == 0 (the bf16 tiles' rule), gradient maps 32..128 wide
BF16_DGW_SHAPES = [
    (2, 64, 128, 64),      # 4 classes x 64 columns, Wo = 32: 8 image rows per tile; 4 chunks
    (1, 64, 128, 256),     # Wo = 128: two image rows per tile, 520-row window (33 pieces)
    (3, 40, 64, 128),      # ragged: 40 of 64 columns per class; Wo = 64, 2 chunks
    (1, 128, 256, 128),    # 2 classes x 128 columns, ph from the block index; 8 chunks
    (2, 96, 192, 64),      # 2 classes, 96 of 128 columns; 6 chunks
    (1, 8, 64, 64),        # 8 columns per class, two chunks
    (3, 64, 320, 64),      # 10 chunks; three images, 12 tiles
]


@pytest.mark.parametrize("N,C,K,H", BF16_DGW_SHAPES)
@pytest.mark.parametrize("splitk,out16", [(0, False), (1, False), (2, False), (0, True)])
def test_conv_bf16_input_grad_window_kernel(N, C, K, H, splitk, out16):
    """bf16 matrix path, input-grad with few output channels: all parity classes of a pixel tile in one workgroup, the gradient
    window in LDS (csrc/igemm_dma_dgw.hip).  bf16 x bf16 products are exact in fp32: the result must equal the fp64 convolution
    of the ROUNDED operands at fp32 tolerance and agree with the register-staged bf16 tiles; bf16 output = RNE of that."""
    r = lambda t_: t_.bfloat16().float()
    w, dy = rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C)), rnd(N, K, H // 2, H // 2, seed=3)
    dxr = TF.conv_transpose2d(r(dy).double(), r(w).double(), stride=2, padding=1).float()
    L = _lib.load()
    _lib.set_option("bf16", 1)
    _lib.set_option("splitk", splitk)
    ops.SHADOW = True
    ops.ACT16 = out16
    try:
        wg = _with_shadow(krsc(w))
        dyg = nhwc16(dy) if out16 else _with_shadow(nhwc(dy))
        assert L.dg_conv_bf16_operands_ok(1, N, H, H, C, K, 2, 1) == 2
        dx = ops.conv_dgrad(dyg, wg, (H, H), 2, 1)
        assert dx.dtype == (torch.bfloat16 if out16 else torch.float32)
        _lib.set_option("no_dma", 1)
        dxreg = ops.conv_dgrad(dyg, wg, (H, H), 2, 1)
        torch.cuda.synchronize()
    finally:
        ops.SHADOW = False
        ops.ACT16 = False
        ops.shadow_clear()
        _lib.set_option("no_dma", 0)
        _lib.set_option("splitk", 0)
        _lib.set_option("bf16", 0)
    if out16:
        close16(dx, dxr, what="bf16 window input-grad, bf16 out")
        # (two roundings of fp32 sums that differ in their last bits: one bf16 step apart at most)
        close16(dx, dxreg.float(), extra=2e-3, what="window vs register-staged, bf16 out")
    else:
        close(dx, dxr, rtol=2e-4, what="bf16 window input-grad")
        close(dx, dxreg, rtol=2e-4, what="window vs register-staged input-grad")


@pytest.mark.parametrize("N,C,K", [(5, 512, 100), (32, 2048, 100), (3, 128, 100)])
def test_conv_head_bf16_operands(N, C, K):
    """The 4x4 heads (plain GEMMs: FWD / DGRAD_PLAIN / WGRAD with ragged K = 100) on the bf16 tile kernels."""
    r = lambda t_: t_.bfloat16().float()
    x, w, dy = rnd(N, C, 4, 4, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C)), rnd(N, K, 1, 1, seed=3)
    xr, wr = r(x).double().requires_grad_(True), r(w).double().requires_grad_(True)
    yr = TF.conv2d(xr, wr)
    yr.backward(r(dy).double())
    _lib.set_option("bf16", 1)
    try:
        xg, wg, dyg = nhwc(x), krsc(w), nhwc(dy)
        close(ops.conv_fwd(xg, wg, 1, 0), yr.float(), what="bf16 head fwd")
        close(ops.conv_dgrad(dyg, wg, (4, 4), 1, 0), xr.grad.float(), rtol=2e-4, what="bf16 head dgrad")
        close(ops.conv_wgrad(dyg, xg, 1, 0), wr.grad.float(), rtol=2e-4, what="bf16 head wgrad")
    finally:
        _lib.set_option("bf16", 0)


def _rel(got, ref64):
    got = got.detach().double().cpu()
    return ((got - ref64).norm() / ref64.norm().clamp_min(1e-300)).item()


@pytest.mark.parametrize("N,C,K,H", [(2, 64, 128, 8), (3, 64, 128, 16), (2, 128, 256, 8), (4, 256, 64, 16), (5, 64, 128, 64),
                                     (2, 1024, 2048, 8), (1, 2048, 2048, 8)])
def test_conv_f32x3_is_fp32_accurate(N, C, K, H):
    """Option "bf16" = 2 ("f32x3"): every fp32 operand is split into three bf16 planes (hi + mid + lo = 24 significand
    bits) and each product block is six bf16 MFMAs with fp32 accumulation (the dropped plane pairs are <= 2^-23 of a
    product).  The claim is fp32 accuracy, so the yardstick is the exact-fp32 MFMA path of the same library: against an
    fp64 reference the split path may be at most 2x as far as the fp32-FMA-chain path (it is usually closer: the
    products are exact and only the accumulation rounds), and it passes the plain fp32 tolerance of every op test."""
    x, w, dy = rnd(N, C, H, H, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C)), rnd(N, K, H // 2, H // 2, seed=3)
    y64 = TF.conv2d(x.double(), w.double(), stride=2, padding=1)
    dx64 = TF.conv_transpose2d(dy.double(), w.double(), stride=2, padding=1)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride=2, padding=1)
    xg, wg, dyg = nhwc(x), krsc(w), nhwc(dy)
    e32 = (_rel(ops.conv_fwd(xg, wg, 2, 1), y64), _rel(ops.conv_dgrad(dyg, wg, (H, H), 2, 1), dx64), _rel(ops.conv_wgrad(dyg, xg, 2, 1), dw64))
    _lib.set_option("bf16", 2)
    try:
        y, dx, dw = ops.conv_fwd(xg, wg, 2, 1), ops.conv_dgrad(dyg, wg, (H, H), 2, 1), ops.conv_wgrad(dyg, xg, 2, 1)
    finally:
        _lib.set_option("bf16", 0)
    close(y, y64.float(), what="f32x3 conv fwd")
    close(dx, dx64.float(), rtol=2e-4, what="f32x3 conv dgrad")
    close(dw, dw64.float(), rtol=2e-4, what="f32x3 conv wgrad")
    ex3 = (_rel(y, y64), _rel(dx, dx64), _rel(dw, dw64))
    print(f"[{N},{C},{K},{H}] rel L2 error vs fp64: exact-fp32 MFMA {e32[0]:.2e}/{e32[1]:.2e}/{e32[2]:.2e}  f32x3 {ex3[0]:.2e}/{ex3[1]:.2e}/{ex3[2]:.2e}")
    for a, b, what in zip(ex3, e32, ("fwd", "dgrad", "wgrad")):
        assert a <= 2.0 * b + 2e-7, f"f32x3 {what}: {a:.2e} vs exact-fp32 path {b:.2e}"


# shapes that reach the plane kernel (igemm_dma_x3.hip: GEMM rows and columns >= 192; C % 16 forward, K % 16 input-grad)
X3_SHAPES = DMA_SHAPES + [
    (5, 224, 288, 16),     # C = 14 chunks of 16, K = 18 chunks: neither is a multiple of 64 (the bf16 LDS-DMA kernel refuses these)
    (32, 192, 512, 8),     # 512 rows, 192 input-grad columns (a 3/4 used tile)
    (4, 64, 128, 32),      # weight gradient of 128 rows: the 128 x 256 tile (4 waves); forward / input-grad stay on the register-staged tiles
    (2, 96, 160, 32),      # 160 weight-grad rows (two ragged 128-row tiles), 1536 columns
]


@pytest.mark.parametrize("N,C,K,H", X3_SHAPES)
@pytest.mark.parametrize("splitk", [0, 1, 3])
@pytest.mark.parametrize("body", [32, pytest.param(16, marks=[needs_experiments, pytest.mark.slow])])
def test_conv_f32x3_plane_kernel(N, C, K, H, splitk, body):
    """body 16: the 256 x 256 tile on v_mfma_f32_16x16x32_bf16 with the planes PAIRED along k (option "x3_mfma" 16: the same six
    products, two per instruction, so a different summation order -- everything but the bit-identity with the register-staged
    kernels is asserted).
    mfma_dtype="f32x3" with PLANE operands: the three bf16 planes of both operands are written once (dg_f32_to_bf16x3) and
    igemm_dma_x3.hip stages them by LDS-DMA.  Same six MFMAs per product block and the same reduction order as the
    register-staged split of igemm.hip: bit-identical on an unsplit GEMM, fp32 tolerance against fp64 otherwise, and at most
    2x the exact-fp32 MFMA path's distance from fp64."""
    x, w, dy = rnd(N, C, H, H, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C)), rnd(N, K, H // 2, H // 2, seed=3)
    y64 = TF.conv2d(x.double(), w.double(), stride=2, padding=1)
    dx64 = TF.conv_transpose2d(dy.double(), w.double(), stride=2, padding=1)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride=2, padding=1)
    xg, wg, dyg = nhwc(x), krsc(w), nhwc(dy)
    e32 = (_rel(ops.conv_fwd(xg, wg, 2, 1), y64), _rel(ops.conv_dgrad(dyg, wg, (H, H), 2, 1), dx64), _rel(ops.conv_wgrad(dyg, xg, 2, 1), dw64))
    L = _lib.load()
    _lib.set_option("bf16", 2)
    _lib.set_option("splitk", splitk)
    if body != 32:
        _lib.set_option("x3_mfma", body)
    try:
        yreg, dxreg, dwreg = ops.conv_fwd(xg, wg, 2, 1), ops.conv_dgrad(dyg, wg, (H, H), 2, 1), ops.conv_wgrad(dyg, xg, 2, 1)
        M = N * (H // 2) ** 2
        # 4: register-staged tiles read the planes (experiments library; the product library: 0 = operands split in the kernel)
        assert L.dg_conv_x3_planes_ok(0, N, H, H, C, K, 2, 1) == (1 if (K >= 192 and M >= 192) else (4 if experiments_built() else 0))
        assert L.dg_conv_x3_planes_ok(1, N, H, H, C, K, 2, 1) == int(C >= 192 and M >= 192)
        assert L.dg_conv_x3_planes_ok(2, N, H, H, C, K, 2, 1) == 1
        ops.X3 = True
        y, dx, dw = ops.conv_fwd(xg, wg, 2, 1), ops.conv_dgrad(dyg, wg, (H, H), 2, 1), ops.conv_wgrad(dyg, xg, 2, 1)
        dw2 = ops.conv_wgrad(dyg, xg, 2, 1, out=dw.clone(), accumulate=True)
        torch.cuda.synchronize()
        assert len(ops._PLANE_TAB) == 2                                  # x and dy were split once each
        # the forward form on the TRANSPOSED weight planes (what a weight of a flat Adam group gets): the same MFMAs in the
        # same order, only the weight tile's path into LDS differs -> bit-identical
        if L.dg_conv_x3_planes_ok(0, N, H, H, C, K, 2, 1) == 1:
            buf = wg._dg_x3[0]
            wg._dg_x3, wg._dg_x3_ver = (buf, 0, torch.zeros_like(buf)), None
            yt = ops.conv_fwd(xg, wg, 2, 1)
            assert ops.weight_planes(wg, transposed=True)[2] == 1
            t_ref = buf.view(3, K, 16 * C).transpose(1, 2).reshape(3, -1)
            assert torch.equal(wg._dg_x3[2], t_ref), "dg_x3_transpose_planes"
            assert torch.equal(yt, y), "forward on transposed weight planes"
        # the triple reproduces the fp32 tensor exactly
        t3 = ops._PLANE_TAB[xg.data_ptr()][1].float().sum(0)
        assert torch.equal(t3, xg.permute(0, 2, 3, 1).reshape(-1))
    finally:
        ops.X3 = False
        ops.planes_clear()
        _lib.set_option("splitk", 0)
        _lib.set_option("bf16", 0)
        if body != 32:
            _lib.set_option("x3_mfma", ops.X3_MFMA)
    close(y, y64.float(), what="plane conv fwd")
    close(dx, dx64.float(), rtol=2e-4, what="plane conv dgrad")
    close(dw, dw64.float(), rtol=2e-4, what="plane conv wgrad")
    close(dw2, 2 * dw64.float(), rtol=2e-4, what="plane conv wgrad accumulate")
    ex3 = (_rel(y, y64), _rel(dx, dx64), _rel(dw, dw64))
    if splitk == 0:          # the default plans of both paths (a forced single slab has 8x longer accumulation chains)
        for a, b, what in zip(ex3, e32, ("fwd", "dgrad", "wgrad")):
            assert a <= 2.0 * b + 2e-7, f"plane {what}: {a:.2e} vs exact-fp32 path {b:.2e}"
    close(y, yreg, what="plane vs register-staged fwd")
    close(dx, dxreg, rtol=2e-4, what="plane vs register-staged dgrad")
    close(dw, dwreg, rtol=2e-4, what="plane vs register-staged wgrad")
    if splitk == 1 and body == 32:   # one slab and the same K walk: the same sequence of MFMAs per accumulator as the register-staged kernel
        assert torch.equal(dw, dwreg), "plane kernel vs register-staged split: weight gradient"
        # (forward: both kernels run the four 16-channel tiles of a 64-channel group back to back since round 3 -- the same walk on every shape)
        assert torch.equal(y, yreg), "plane kernel vs register-staged split: forward"
        if K % 64:
            assert torch.equal(dx, dxreg), "plane kernel vs register-staged split: input gradient"


# (N, C, K, H): input-grad shapes of the window kernel (igemm_dma_x3_dgw.hip): C <= 128 output channels, gradient maps 32..128 wide
X3_DGW_SHAPES = [
    (2, 64, 128, 64),      # 4 classes x 64 columns, Wo = 32: 8 image rows per tile, 8 tiles
    (1, 64, 128, 256),     # Wo = 128: two image rows per tile, 520-row window (17 pieces)
    (3, 40, 64, 128),      # ragged: 40 of 64 columns per class; Wo = 64, 4 chunks
    (1, 128, 256, 128),    # 2 classes x 128 columns, ph from the block index; 16 chunks
    (2, 96, 96, 64),       # 2 classes, 96 of 128 columns; K = 96 = 6 chunks
    (1, 8, 32, 64),        # 8 columns per class, two chunks
]


@pytest.mark.parametrize("N,C,K,H", X3_DGW_SHAPES)
@pytest.mark.parametrize("splitk", [0, 1, 2])
def test_conv_f32x3_input_grad_window_kernel(N, C, K, H, splitk):
    """Input-grad with few output channels on plane operands: all parity classes of a pixel tile in one workgroup, the gradient
    window in LDS (csrc/igemm_dma_x3_dgw.hip).  Same products and the same (chunk, tap) reduction order per output element as
    the per-class kernels: bit-identical to the register-staged f32x3 form on an unsplit GEMM, fp32 tolerance against fp64,
    borders (zero halo) and ragged channel counts included."""
    w, dy = rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C)), rnd(N, K, H // 2, H // 2, seed=3)
    dx64 = TF.conv_transpose2d(dy.double(), w.double(), stride=2, padding=1)
    wg, dyg = krsc(w), nhwc(dy)
    L = _lib.load()
    _lib.set_option("bf16", 2)
    _lib.set_option("splitk", splitk)
    try:
        dxreg = ops.conv_dgrad(dyg, wg, (H, H), 2, 1)
        assert L.dg_conv_x3_planes_ok(1, N, H, H, C, K, 2, 1) == (2 if K % 64 == 0 else 1)      # 2: prefers quad-chunk planes
        ops.X3 = True
        dx = ops.conv_dgrad(dyg, wg, (H, H), 2, 1)
        torch.cuda.synchronize()
        assert len(ops._PLANE_TAB) == 1
        if K % 64 == 0:
            # the same launch on QUAD-CHUNK gradient planes [pixels/4][K/16][4][16] (what the BatchNorm kernels write for this kernel):
            # same products, same order -> the same bits; and the plane weight-grad kernel reads that layout too
            xg = nhwc(rnd(N, C, H, H, seed=9))
            dw = ops.conv_wgrad(dyg, xg, 2, 1)
            M = N * (H // 2) ** 2
            t3 = ops._PLANE_TAB[dyg.data_ptr()][1]
            cm3 = t3.view(3, M // 4, 4, K // 16, 16).permute(0, 1, 3, 2, 4).contiguous().view(3, -1)
            ops.planes_put(dyg, cm3, cm=True)
            dx_cm = ops.conv_dgrad(dyg, wg, (H, H), 2, 1)
            dw_cm = ops.conv_wgrad(dyg, xg, 2, 1)
            torch.cuda.synchronize()
            assert ops._PLANE_TAB[dyg.data_ptr()][2] is True and ops._PLANE_TAB[dyg.data_ptr()][1] is cm3     # no silent re-split
            assert torch.equal(dx_cm, dx), "window input-grad: quad-chunk planes vs pixel-major planes"
            if L.dg_conv_x3_planes_ok(2, N, H, H, C, K, 2, 1) == 1:
                assert torch.equal(dw_cm, dw), "plane weight-grad: quad-chunk dy planes vs pixel-major"
            # a reader that cannot take the layout (the forward form) gets a pixel-major split of the fp32 tensor instead
            p0, _, cmflag = ops.planes_of(dyg, allow_cm=False)
            assert cmflag == 0 and p0 != cm3.data_ptr()
    finally:
        ops.X3 = False
        ops.planes_clear()
        _lib.set_option("splitk", 0)
        _lib.set_option("bf16", 0)
    close(dx, dx64.float(), rtol=2e-4, what="window input-grad")
    close(dx, dxreg, rtol=2e-4, what="window vs register-staged input-grad")
    if splitk == 1:
        assert torch.equal(dx, dxreg), "window kernel vs register-staged split"


def test_x3_transpose_planes_group():
    """dg_x3_transpose_planes over a flat plane buffer holding several weights: [K][J] images -> [J][K] at the same offsets
    (16-byte path for K % 8 == 0 and J % 8 == 0, element path otherwise, ragged 64 x 64 tiles); the rest of dst is untouched."""
    shapes = [(256, 16 * 64), (100, 16 * 32), (36, 80), (8, 16), (200, 16 * 24), (64, 64)]
    offs, total = [], 0
    for k, j in shapes:
        offs.append(total)
        total += (k * j + 63) // 64 * 64 + 64
    g = torch.Generator().manual_seed(5)
    src = torch.randn(3, total, generator=g).to(DEV).bfloat16()
    dst = torch.full_like(src, 7.0)
    ops.x3_transpose_planes(src, dst, ops.x3_transpose_table([(o, k, j) for o, (k, j) in zip(offs, shapes)]))
    ref = torch.full_like(src, 7.0)
    for o, (k, j) in zip(offs, shapes):
        ref[:, o:o + k * j] = src[:, o:o + k * j].view(3, k, j).transpose(1, 2).reshape(3, -1)
    assert torch.equal(dst, ref)


@pytest.mark.parametrize("N,C,K", [(5, 512, 100), (32, 2048, 100)])
def test_conv_head_f32x3(N, C, K):
    x, w, dy = rnd(N, C, 4, 4, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C)), rnd(N, K, 1, 1, seed=3)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = TF.conv2d(xr, wr)
    yr.backward(dy.double())
    _lib.set_option("bf16", 2)
    try:
        xg, wg, dyg = nhwc(x), krsc(w), nhwc(dy)
        close(ops.conv_fwd(xg, wg, 1, 0), yr.float(), what="f32x3 head fwd")
        close(ops.conv_dgrad(dyg, wg, (4, 4), 1, 0), xr.grad.float(), rtol=2e-4, what="f32x3 head dgrad")
        close(ops.conv_wgrad(dyg, xg, 1, 0), wr.grad.float(), rtol=2e-4, what="f32x3 head wgrad")
    finally:
        _lib.set_option("bf16", 0)


@pytest.mark.parametrize("N,H", [(2, 8), (3, 16), (2, 64), (1, 256), (5, 32)])
def test_c3_dgrad_scatter_and_gather_forms(N, H):
    """Last ConvTranspose2d(64,3,4,2,1) + Sigmoid: the scatter form (dense [pixels x 64].[64 x 48] GEMM + overlap-add in
    LDS; tiles of 6 x 14 pixels with halo, so ragged in both directions at every size here) and the older gather form
    (option kt=16) against F.conv_transpose2d, and against each other."""
    x = rnd(N, 64, H // 2, H // 2, seed=5)
    w = rnd(64, 3, 4, 4, seed=6, scale=0.1)
    ref = torch.sigmoid(TF.conv_transpose2d(x.double(), w.double(), stride=2, padding=1)).float()
    xg, wg = nhwc(x), w.to(DEV)
    a = ops.c3_dgrad(xg, wg, ops.ACT_SIGMOID)
    close(a, ref, rtol=1e-5, atol=1e-6, what="c3 dgrad scatter form")
    raw = ops.c3_dgrad(xg, wg, ops.ACT_NONE)
    close(raw, TF.conv_transpose2d(x.double(), w.double(), stride=2, padding=1).float(), what="c3 dgrad scatter form (no act)")
    _lib.set_option("kt", 16)
    try:
        b = ops.c3_dgrad(xg, wg, ops.ACT_SIGMOID)
    finally:
        _lib.set_option("kt", 0)
    close(b, ref, rtol=1e-5, atol=1e-6, what="c3 dgrad gather form")
    assert torch.equal(ops.c3_dgrad(xg, wg, ops.ACT_SIGMOID), a), "not deterministic"


def test_pointer_path_kernels():
    """Tensors of 2 GiB and more use the 64-bit addressing instantiations (no buffer descriptors); force them
    on small shapes so that path stays covered."""
    _lib.set_option("pointer_path", 1)
    try:
        test_conv_s2_fwd_dgrad_wgrad(2, 128, 256, 8)
        test_conv_s2_fwd_dgrad_wgrad(3, 64, 128, 16)
        test_conv_head_valid(5, 512, 100)
        test_convT_s2(2, 128, 64, 8)
        test_c3_edge(2, 16, 64)
    finally:
        _lib.set_option("pointer_path", 0)


@pytest.mark.parametrize("N,C,K", [(2, 128, 100), (5, 512, 100), (4, 2048, 100), (2, 128, 1), (7, 512, 1), (32, 2048, 1)])
def test_conv_head_valid(N, C, K):
    """Conv2d(C,K,4,1,0) on a 4x4 input (model.py:35,107): fwd / dgrad / wgrad."""
    x = rnd(N, C, 4, 4, seed=1)
    w = rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C))
    dy = rnd(N, K, 1, 1, seed=3)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = TF.conv2d(xr, wr)
    yr.backward(dy)
    xg, wg, dyg = nhwc(x), krsc(w), nhwc(dy)
    close(ops.conv_fwd(xg, wg, 1, 0), yr, what="head fwd")
    close(ops.conv_dgrad(dyg, wg, (4, 4), 1, 0), xr.grad, rtol=2e-4, what="head dgrad")
    close(ops.conv_wgrad(dyg, xg, 1, 0), wr.grad, rtol=2e-4, what="head wgrad")


@pytest.mark.parametrize("N,Cin,Cout,Hin", [(2, 128, 64, 4), (3, 256, 128, 8), (2, 2048, 2048, 4), (4, 512, 256, 4), (2, 128, 64, 16)])
def test_convT_s2(N, Cin, Cout, Hin):
    """ConvTranspose2d(Cin,Cout,4,2,1) (model.py:118-140) through the autograd Function."""
    from discogan_modernized_amd import functional as F
    x = rnd(N, Cin, Hin, Hin, seed=1)
    w = rnd(Cin, Cout, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * Cout))
    dy = rnd(N, Cout, 2 * Hin, 2 * Hin, seed=3)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = TF.conv_transpose2d(xr, wr, stride=2, padding=1)
    yr.backward(dy)
    xg = nhwc(x).requires_grad_(True)
    wg = krsc(w).requires_grad_(True)
    y = F.ConvTransposeFn.apply(xg, wg, 2, 1)
    close(y, yr, what="convT fwd")
    y.backward(nhwc(dy))
    close(xg.grad, xr.grad, rtol=2e-4, what="convT dgrad")
    close(wg.grad, wr.grad, rtol=2e-4, what="convT wgrad")


@pytest.mark.parametrize("N,Cout", [(2, 128), (5, 512), (3, 2048)])
def test_convT_head(N, Cout):
    """ConvTranspose2d(100,Cout,4,1,0) on a 1x1 input (model.py:114)."""
    from discogan_modernized_amd import functional as F
    x = rnd(N, 100, 1, 1, seed=1)
    w = rnd(100, Cout, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * Cout))
    dy = rnd(N, Cout, 4, 4, seed=3)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = TF.conv_transpose2d(xr, wr)
    yr.backward(dy)
    xg = nhwc(x).requires_grad_(True)
    wg = krsc(w).requires_grad_(True)
    y = F.ConvTransposeFn.apply(xg, wg, 1, 0)
    close(y, yr, what="convT head fwd")
    y.backward(nhwc(dy))
    close(xg.grad, xr.grad, rtol=2e-4, what="convT head dgrad")
    close(wg.grad, wr.grad, rtol=2e-4, what="convT head wgrad")


@pytest.mark.parametrize("N,H,K", [(2, 16, 64), (3, 8, 64), (1, 64, 64), (2, 32, 128), (5, 16, 64)])
def test_c3_edge(N, H, K):
    """3-channel image side: conv1 (+LeakyReLU), its dgrad / wgrad, last convT (+Sigmoid)."""
    x = torch.rand(N, 3, H, H, generator=torch.Generator().manual_seed(1))
    w = rnd(K, 3, 4, 4, seed=2, scale=1.0 / math.sqrt(48))
    dy = rnd(N, K, H // 2, H // 2, seed=3)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = TF.conv2d(xr, wr, stride=2, padding=1)
    yr.backward(dy)
    xg, wg, dyg = x.to(DEV), w.to(DEV), nhwc(dy)
    close(ops.c3_fwd(xg, wg, ops.ACT_NONE), yr, what="c3 fwd")
    close(ops.c3_fwd(xg, wg, ops.ACT_LEAKY, 0.2), TF.leaky_relu(yr, 0.2), what="c3 fwd + lrelu")
    close(ops.c3_dgrad(dyg, wg, ops.ACT_NONE), xr.grad, rtol=2e-4, what="c3 dgrad")
    close(ops.c3_wgrad(dyg, xg), wr.grad, rtol=2e-4, what="c3 wgrad")
    # weight gradient with the LeakyReLU backward fused into the dy loads (dg_conv4x4s2_c3_wgrad_act)
    xr2, wr2 = x.clone(), w.clone().requires_grad_(True)
    yl = TF.leaky_relu(TF.conv2d(xr2, wr2, stride=2, padding=1), 0.2)
    yl.backward(dy)
    yg = ops.c3_fwd(xg, wg, ops.ACT_LEAKY, 0.2)
    close(ops.c3_wgrad(dyg, xg, act_out=yg, act=ops.ACT_LEAKY, slope=0.2), wr2.grad, rtol=2e-4, what="c3 wgrad + lrelu bwd")
    acc = torch.ones(K, 3, 4, 4, device=DEV)
    ops.c3_wgrad(dyg, xg, out=acc, accumulate=True, act_out=yg, act=ops.ACT_LEAKY, slope=0.2)
    close(acc - 1.0, wr2.grad, rtol=2e-4, atol=2e-6, what="c3 wgrad + lrelu bwd, accumulate")
    # last ConvTranspose2d(K,3) + sigmoid == sigmoid(dgrad)
    ref = torch.sigmoid(TF.conv_transpose2d(dy, w, stride=2, padding=1))
    close(ops.c3_dgrad(dyg, wg, ops.ACT_SIGMOID), ref, what="c3 convT + sigmoid")


@pytest.mark.parametrize("N,C,H,act", [(2, 128, 8, "leaky"), (4, 64, 16, "relu"), (2, 100, 1, "leaky"),
                                       (3, 2048, 4, "relu"), (2, 256, 32, "leaky"), (6, 512, 2, "none")])
def test_batchnorm_act(N, C, H, act):
    y = rnd(N, C, H, H, seed=1, scale=2.0) + 0.3
    gamma = rnd(C, seed=2) + 1.5
    beta = rnd(C, seed=3)
    dz = rnd(N, C, H, H, seed=4)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    yr = y.clone().requires_grad_(True)
    u = bn(yr)
    zr = {"leaky": lambda t: TF.leaky_relu(t, 0.2), "relu": TF.relu, "none": lambda t: t}[act](u)
    zr.backward(dz)
    code = {"leaky": ops.ACT_LEAKY, "relu": ops.ACT_RELU, "none": ops.ACT_NONE}[act]
    yg = nhwc(y)
    rm = torch.zeros(C, device=DEV)
    rv = torch.ones(C, device=DEV)
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    gg, bg = gamma.to(DEV), beta.to(DEV)
    saved = ops.bn_train_stats(yg, rm, rv, nbt, 1e-5, 0.1)
    z = ops.bn_act_fwd(yg, saved, gg, bg, code, 0.2)
    close(z, zr, what="bn fwd")
    close(rm, bn.running_mean, what="running_mean")
    close(rv, bn.running_var, what="running_var")
    assert int(nbt) == 1
    dy, dgamma, dbeta = ops.bn_act_bwd(nhwc(dz), yg, saved, gg, bg, code, 0.2)
    close(dy, yr.grad, rtol=5e-4, atol=1e-6, what="bn dx")
    close(dgamma, bn.weight.grad, rtol=5e-4, what="bn dgamma")
    close(dbeta, bn.bias.grad, rtol=5e-4, what="bn dbeta")


@pytest.mark.parametrize("N,C,K,H", [(2, 64, 128, 8), (3, 64, 128, 16), (4, 256, 512, 8), (8, 64, 128, 32), (2, 128, 64, 16)])
def test_conv_fused_bn_statistics(N, C, K, H):
    """BN statistics emitted by the conv epilogue / split-K reduction == the stand-alone statistics pass."""
    x = rnd(N, C, H, H, seed=1) + 0.4
    w = rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C))
    xg, wg = nhwc(x), krsc(w)
    for name, fn in (("fwd", lambda: ops.conv_fwd(xg, wg, 2, 1, want_stats=True)),):
        y, st = fn()
        assert st is not None and st.shape[1] == 3 * y.shape[1] + 4
        rm1, rv1 = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
        rm2, rv2 = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
        n1, n2 = torch.zeros((), dtype=torch.long, device=DEV), torch.zeros((), dtype=torch.long, device=DEV)
        a = ops.bn_stats_from_partials(st, y, rm1, rv1, n1, 1e-5, 0.1)
        b = ops.bn_train_stats(y, rm2, rv2, n2, 1e-5, 0.1)
        close(a, b, rtol=2e-6, atol=1e-7, what=f"{name} fused stats")
        close(rv1, rv2, rtol=2e-6, atol=1e-7, what=f"{name} running_var")
        assert int(n1) == 1
    # ConvTranspose direction (dgrad kernel, 4 parity classes): dy [N,K,H/2,H/2] -> [N,C,H,H]
    dy = nhwc(rnd(N, K, H // 2, H // 2, seed=3) + 0.2)
    out, st = ops.conv_dgrad(dy, wg, (H, H), 2, 1, want_stats=True)
    rm1, rv1 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    a = ops.bn_stats_from_partials(st, out, rm1, rv1, None, 1e-5, 0.1)
    b = ops.bn_train_stats(out, None, None, None, 1e-5, 0.1)
    close(a, b, rtol=2e-6, atol=1e-7, what="dgrad fused stats")


@pytest.mark.parametrize("N,C,H,act", [(4, 64, 16, "leaky"), (3, 256, 8, "relu"), (2, 72, 4, "none")])
def test_batchnorm_writes_plane_triples(N, C, H, act):
    """f32x3 path: the BatchNorm apply kernels also write the three bf16 planes of what they produce (dg_bn_act_fwd_x3 /
    dg_bn_act_bwd_x3).  The fp32 results are those of the plain kernels bit for bit, the planes are those of dg_f32_to_bf16x3
    on the fp32 result, and hi + mid + lo reproduces it exactly."""
    code = {"leaky": ops.ACT_LEAKY, "relu": ops.ACT_RELU, "none": ops.ACT_NONE}[act]
    yg, dzg = nhwc(rnd(N, C, H, H, seed=1, scale=2.0) + 0.3), nhwc(rnd(N, C, H, H, seed=4))
    gg, bg = (rnd(C, seed=2) + 1.5).to(DEV), rnd(C, seed=3).to(DEV)
    saved = ops.bn_train_stats(yg, None, None, None, 1e-5, 0.1)
    z0 = ops.bn_act_fwd(yg, saved, gg, bg, code, 0.2)
    dy0, dg0, db0 = ops.bn_act_bwd(dzg, yg, saved, gg, bg, code, 0.2)
    ops.X3 = True
    try:
        z = ops.bn_act_fwd(yg, saved, gg, bg, code, 0.2)
        dy, dg, db = ops.bn_act_bwd(dzg, yg, saved, gg, bg, code, 0.2)
        z3, dy3 = ops._PLANE_TAB[z.data_ptr()][1], ops._PLANE_TAB[dy.data_ptr()][1]
        ref = lambda t: ops.f32_to_bf16x3(t, torch.empty((3, t.numel()), device=DEV, dtype=torch.bfloat16))
        assert torch.equal(z, z0) and torch.equal(dy, dy0) and torch.equal(dg, dg0) and torch.equal(db, db0)
        assert torch.equal(z3, ref(z)) and torch.equal(dy3, ref(dy))
        mem = lambda t: t.permute(0, 2, 3, 1).reshape(-1)
        assert torch.equal(z3.float().sum(0), mem(z)) and torch.equal(dy3.float().sum(0), mem(dy))
        if C % 64 == 0 and (N * H * H) % 4 == 0:
            # quad-chunk planes [pixels/4][C/16][4][16] (plane_layout 1: for a window input-grad kernel): the same values, rearranged
            zc = ops.bn_act_fwd(yg, saved, gg, bg, code, 0.2, planes_cm=True)
            dyc, _, _ = ops.bn_act_bwd(dzg, yg, saved, gg, bg, code, 0.2, planes_cm=True)
            M = N * H * H
            cm = lambda t3: t3.view(3, M // 4, 4, C // 16, 16).permute(0, 1, 3, 2, 4).contiguous().view(3, -1)
            ez, edy = ops._PLANE_TAB[zc.data_ptr()], ops._PLANE_TAB[dyc.data_ptr()]
            assert ez[2] and edy[2] and torch.equal(zc, z0) and torch.equal(dyc, dy0)
            assert torch.equal(ez[1], cm(z3)) and torch.equal(edy[1], cm(dy3))
    finally:
        ops.X3 = False
        ops.planes_clear()


@pytest.mark.parametrize("N,C,H", [(4, 64, 16), (3, 256, 8), (1, 128, 6), (5, 1024, 2), (2, 2048, 4), (32, 64, 64)])
def test_batchnorm_row_geometry_kernels_equal_item_kernels(N, C, H):
    """The fp32 BatchNorm apply passes run in the reduction passes' geometry (fixed channels per thread; csrc/norm_act.hip
    bn_act_fwd_rows_kernel / bn_bwd_apply_rows_kernel) wherever C % 64 == 0 and M % 4 == 0; option "bn_items" 1 puts them back on the
    per-item kernels.  Same expressions: every output -- fp32, bf16 shadow, plane triples in both layouts, grouped launches, ragged row
    chunks -- must be bitwise the same."""
    yg, dzg = nhwc(rnd(N, C, H, H, seed=1, scale=2.0) + 0.3), nhwc(rnd(N, C, H, H, seed=4))
    y2, dz2 = nhwc(rnd(N, C, H, H, seed=11, scale=0.5) - 0.1), nhwc(rnd(N, C, H, H, seed=14))
    gg, bg = (rnd(C, seed=2) + 1.5).to(DEV), rnd(C, seed=3).to(DEV)

    def run():
        out = {}
        saved = ops.bn_train_stats(yg, None, None, None, 1e-5, 0.1)
        saved2 = ops.bn_train_stats(y2, None, None, None, 1e-5, 0.1)
        for act in (ops.ACT_LEAKY, ops.ACT_RELU, ops.ACT_NONE):
            out[("z", act)] = ops.bn_act_fwd(yg, saved, gg, bg, act, 0.2)
            out[("bwd", act)] = ops.bn_act_bwd(dzg, yg, saved, gg, bg, act, 0.2)
        zs = ops.bn_act_fwd_g([yg, y2], [saved, saved2], [gg, gg], [bg, bg], ops.ACT_LEAKY, 0.2)
        out["zg"] = tuple(zs)
        dgs, dbs = [torch.zeros(C, device=DEV) for _ in range(2)], [torch.zeros(C, device=DEV) for _ in range(2)]
        out["bwdg"] = (ops.bn_act_bwd_g([dzg, dz2], [yg, y2], [saved, saved2], [gg, gg], [bg, bg], ops.ACT_LEAKY, 0.2, dgs, dbs, False), dgs, dbs)
        ops.SHADOW = True
        try:
            z = ops.bn_act_fwd(yg, saved, gg, bg, ops.ACT_LEAKY, 0.2)
            dy = ops.bn_act_bwd(dzg, yg, saved, gg, bg, ops.ACT_LEAKY, 0.2)[0]
            out["shadow"] = (z, ops._SHADOW_TAB[z.data_ptr()][1].clone(), dy, ops._SHADOW_TAB[dy.data_ptr()][1].clone())
        finally:
            ops.SHADOW = False
            ops.shadow_clear()
        ops.X3 = True
        try:
            for cm in ((False, True) if (N * H * H) % 4 == 0 else (False,)):
                z = ops.bn_act_fwd(yg, saved, gg, bg, ops.ACT_RELU, 0.2, planes_cm=cm, planes_only=True)
                dy = ops.bn_act_bwd(dzg, yg, saved, gg, bg, ops.ACT_RELU, 0.2, planes_cm=cm, planes_only=True)[0]
                out[("planes", cm)] = (ops._PLANE_TAB[z.data_ptr()][1].clone(), ops._PLANE_TAB[dy.data_ptr()][1].clone())
        finally:
            ops.X3 = False
            ops.planes_clear()
        return out

    def flat(v):
        return [t for x in (v if isinstance(v, (tuple, list)) else (v,)) for t in (flat(x) if isinstance(x, (tuple, list)) else (x,))]
    rows = run()
    _lib.set_option("bn_items", 1)
    try:
        items = run()
    finally:
        _lib.set_option("bn_items", 0)
    assert rows.keys() == items.keys()
    for k in rows:
        for a, b in zip(flat(rows[k]), flat(items[k])):
            assert torch.equal(a, b), k


def test_bn_needs_two_values():
    y = nhwc(rnd(1, 100, 1, 1))
    with pytest.raises(_lib.DiscoganHipError, match="more than 1 value"):
        ops.bn_train_stats(y, None, None, None, 1e-5, 0.1)


@pytest.mark.parametrize("act", ["leaky", "relu", "sigmoid"])
@pytest.mark.parametrize("n", [1, 7, 4096, 100003])
def test_activations(act, n):
    x = rnd(n, seed=1, scale=20.0)
    dy = rnd(n, seed=2)
    xr = x.clone().requires_grad_(True)
    fn = {"leaky": lambda t: TF.leaky_relu(t, 0.2), "relu": TF.relu, "sigmoid": torch.sigmoid}[act]
    yr = fn(xr)
    yr.backward(dy)
    code = {"leaky": ops.ACT_LEAKY, "relu": ops.ACT_RELU, "sigmoid": ops.ACT_SIGMOID}[act]
    yg = ops.act_fwd(x.to(DEV), code, 0.2)
    close(yg, yr, rtol=1e-6, atol=1e-7, what="act fwd")
    close(ops.act_bwd(dy.to(DEV), yg, code, 0.2), xr.grad, rtol=1e-5, atol=1e-7, what="act bwd")


def test_sigmoid_saturation():
    x = torch.tensor([-200.0, -100.0, -40.0, -17.0, 0.0, 17.0, 18.0, 40.0, 100.0])
    y = ops.act_fwd(x.to(DEV), ops.ACT_SIGMOID).cpu()
    ref = torch.sigmoid(x)
    assert y[-1] == 1.0 and y[-2] == 1.0 and y[0] == 0.0
    assert torch.allclose(y, ref, rtol=1e-6, atol=1e-30)


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (4, 3, 64, 64), (1, 3, 5, 7)])
def test_mse(shape):
    x, t = torch.rand(shape, generator=torch.Generator().manual_seed(1)), torch.rand(shape, generator=torch.Generator().manual_seed(2))
    xr = x.clone().requires_grad_(True)
    lr = TF.mse_loss(xr, t)
    (lr * 0.37).backward()
    loss, xd, td = ops.mse_fwd(x.to(DEV), t.to(DEV))
    close(loss, lr, rtol=1e-6, atol=1e-8, what="mse")
    g = torch.tensor(0.37, device=DEV)
    close(ops.mse_bwd(xd, td, g), xr.grad, rtol=1e-5, atol=1e-10, what="mse bwd")


@pytest.mark.parametrize("label", [1.0, 0.0])
def test_bce_including_saturation(label):
    """-100 log clamp (forward) and the 1e-12 guard (backward) are on the parity-critical path."""
    p = torch.tensor([0.3, 0.9, 1.0, 0.0, 1e-30, 4e-18, 0.5, 0.9999999])
    pr = p.clone().requires_grad_(True)
    lr = TF.binary_cross_entropy(pr.view(-1, 1), torch.full((8, 1), label))
    (lr * 0.1).backward()
    loss, pc = ops.bce_fwd(p.to(DEV), label)
    close(loss, lr, rtol=1e-6, atol=1e-7, what="bce")
    dp = ops.bce_bwd(pc, label, torch.tensor(0.1, device=DEV)).cpu()
    ref = pr.grad
    assert torch.allclose(dp, ref, rtol=1e-5, atol=0), (dp, ref)


@pytest.mark.parametrize("N,C,H", [(2, 128, 8), (4, 256, 4), (3, 64, 16)])
def test_feature_matching(N, C, H):
    real, fake = rnd(N, C, H, H, seed=1), rnd(N, C, H, H, seed=2)
    rr, fr = real.clone().requires_grad_(True), fake.clone().requires_grad_(True)
    l2 = (rr.mean(0) - fr.mean(0)) * (rr.mean(0) - fr.mean(0))
    lr = torch.nn.HingeEmbeddingLoss()(l2, torch.ones(l2.size()))
    (lr * 0.9).backward()
    loss, diff, rd, fd = ops.fm_fwd(nhwc(real), nhwc(fake))
    close(loss, lr, rtol=1e-5, atol=1e-9, what="fm")
    dreal, dfake = ops.fm_bwd(diff, rd, fd, torch.tensor(0.9, device=DEV), True, True)
    close(dreal, rr.grad, rtol=1e-4, atol=1e-10, what="fm dreal")
    close(dfake, fr.grad, rtol=1e-4, atol=1e-10, what="fm dfake")


@pytest.mark.parametrize("arch", [0, 1, 2])
@pytest.mark.parametrize("rate", [0.01, 0.5])
def test_loss_mix(arch, rate):
    """dg_loss_mix_fwd/bwd against the reference's scalar arithmetic (image_translation.py:162-166, 367-382),
    bit-exact forward (same fp32 operations in the same order), exact seeds."""
    nfm = 3
    lv = torch.rand(8 + 2 * nfm, dtype=torch.float32) + 0.1
    t = lv.clone().requires_grad_(True)
    fmA = 0
    fmB = 0
    for l in range(nfm):
        fmA = fmA + t[8 + l]
        fmB = fmB + t[8 + nfm + l]
    disA, disB = (t[2] + t[3]) * 0.5, (t[5] + t[6]) * 0.5
    genA, genB = t[4], t[7]
    totA = (fmB * 0.9 + genB * 0.1) * (1 - rate) + t[0] * rate
    totB = (fmA * 0.9 + genA * 0.1) * (1 - rate) + t[1] * rate
    if arch == 0:
        gen, dis = totA + totB, disA + disB
    elif arch == 1:
        gen, dis = totA, disB
    else:
        gen, dis = genB * 0.1 + fmB * 0.9, disB
    out = ops.loss_mix_fwd(lv.to(DEV), nfm, rate, arch).cpu()
    ref = torch.stack([genA, genB, fmA, fmB, disA, disB, gen, dis]).detach()
    assert torch.allclose(out, ref, rtol=2e-7, atol=0), (out, ref)
    for which, loss in ((6, gen), (7, dis)):
        (g,) = torch.autograd.grad(loss * 1.5, t, retain_graph=True)
        gv = ops.loss_mix_bwd(torch.tensor([1.5], device=DEV), lv.numel(), nfm, rate, arch, which).cpu()
        assert torch.allclose(gv, g, rtol=2e-7, atol=0), (which, gv, g)


def test_adam_flat_matches_torch():
    """Op-wise Adam parity with oracle gradients over several steps."""
    from discogan_modernized_amd import optim
    torch.manual_seed(0)
    shapes = [(64, 3, 4, 4), (128,), (100, 128, 4, 4), (7,)]
    ref_p = [torch.nn.Parameter(torch.randn(s) * 0.05) for s in shapes]
    my_p = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref_p]
    ro = torch.optim.Adam(ref_p, lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-5)
    mo = optim.Adam(my_p, lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-5)
    for step in range(5):
        for rp, mp in zip(ref_p, my_p):
            g = torch.randn(rp.shape) * (10.0 ** (-step))
            rp.grad = g.clone()
            mp.grad.copy_(g.to(DEV))
        ro.step()
        mo.step()
        for rp, mp in zip(ref_p, my_p):
            assert (mp.detach().cpu() - rp.detach()).abs().max().item() <= 2e-7, f"step {step}"
    assert float(mo.state[0]) == 5.0


@pytest.mark.parametrize("n,off", [(8 * 1000 + 5, 0), (4096, 4), (37, 0), (100003, 12), (1 << 20, 0)])
def test_adam_forms_agree(n, off):
    """The Adam kernel's three forms -- plain, + bf16 shadow, + three bf16 planes (8 parameters per trip with 16-byte plane stores where
    the range starts on a 16-byte plane granule, else 4) -- update p / m / v bit for bit alike on ranges of any length and offset; the
    shadow is the rounded parameter, the planes are dg_f32_to_bf16x3 of it."""
    tot = n + off + 16
    base = [torch.randn(tot, generator=torch.Generator().manual_seed(i)).to(DEV) * 0.05 for i in range(4)]
    base[3].abs_()
    state = torch.zeros(8, device=DEV, dtype=torch.float64)
    ops.adam_advance(state, 2e-4, 0.5, 0.999)
    runs = []
    for form in range(3):
        p, g, m, v = (t.clone() for t in base)
        p16 = torch.zeros(tot, device=DEV, dtype=torch.bfloat16)
        PE = (tot + 7) // 8 * 8
        p3 = torch.zeros((3, PE), device=DEV, dtype=torch.bfloat16)
        sl = slice(off, off + n)
        kw = {} if form == 0 else (dict(p16=p16[sl]) if form == 1 else dict(p3=(p3.data_ptr() + 2 * off, p3.stride(0))))
        ops.adam_step_flat(p[sl], g[sl], m[sl], v[sl], state, 0.5, 0.999, 1e-8, 1e-5, **kw)
        runs.append((p, m, v, p16, p3))
    for form in (1, 2):
        for a, b in zip(runs[0][:3], runs[form][:3]):
            assert torch.equal(a, b), form
    p = runs[0][0]
    assert torch.equal(p[:off], base[0][:off]) and torch.equal(p[off + n:], base[0][off + n:])            # nothing outside the range
    assert torch.equal(runs[1][3][off:off + n], p[off:off + n].bfloat16()) and not bool(runs[1][3][off + n:].float().any())
    ref3 = torch.zeros((3, n + 8 - n % 8 if n % 8 else n), device=DEV, dtype=torch.bfloat16)
    ops.f32_to_bf16x3(p[off:off + n].clone(), ref3)
    got3 = runs[2][4]
    assert torch.equal(got3[:, off:off + n], ref3[:, :n]) and not bool(got3[:, off + n:].float().any()) and not bool(got3[:, :off].float().any())


def test_layout_roundtrip():
    x = rnd(3, 20, 6, 10, seed=5).to(DEV)
    y = ops.as_nhwc(x)
    assert ops.is_nhwc(y) and torch.equal(y.cpu(), x.cpu())
    z = ops.to_nchw_contiguous(y)
    assert z.is_contiguous() and torch.equal(z.cpu(), x.cpu())


def test_errors_are_reported_not_fatal():
    x = nhwc(rnd(2, 48, 8, 8))                     # C=48 is not a multiple of 32
    w = krsc(rnd(64, 48, 4, 4))
    with pytest.raises(_lib.DiscoganHipError, match="multiple of 32"):
        ops.conv_fwd(x, w, 2, 1)
    with pytest.raises(_lib.DiscoganHipError, match="unsupported"):
        ops.conv_fwd(nhwc(rnd(2, 64, 8, 8)), krsc(rnd(64, 64, 4, 4)), 1, 1)
    with pytest.raises(_lib.DiscoganHipError, match="CPU"):
        ops.conv_fwd(rnd(2, 64, 8, 8), rnd(64, 64, 4, 4), 2, 1)


# ---- bf16 activation storage (the *_t entry points): bf16 in / bf16 out, fp32 arithmetic inside --------------------------
BF16_ULP = 2.0 ** -8            # RNE to bf16: relative error <= 2^-9 per rounding; one rounding of the output is allowed


def close16(got16, ref32, extra=0.0, what=""):
    """got16 (bf16) must be ref32 rounded once: |got - ref| <= 2^-8 |ref| (+ extra * max|ref| for inputs' own effect)."""
    g, r = got16.detach().float().cpu(), ref32.detach().float().cpu()
    assert g.shape == r.shape, f"{what}: shape {tuple(g.shape)} vs {tuple(r.shape)}"
    assert torch.isfinite(g).all(), f"{what}: non-finite"
    bound = BF16_ULP * r.abs() + (extra + 1e-6) * r.abs().max()
    bad = ((g - r).abs() > bound)
    assert not bad.any(), f"{what}: {int(bad.sum())} elements off, worst {((g - r).abs() - bound).max().item():.3e} over the bound"


def nhwc16(t):
    """CPU logical NCHW fp32 -> GPU bf16 tensor with NHWC memory."""
    return t.to(DEV).bfloat16().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


@pytest.mark.parametrize("N,C,H", [(4, 64, 32), (3, 128, 16), (2, 2048, 4), (5, 512, 8), (32, 64, 8)])
@pytest.mark.parametrize("act", [ops.ACT_LEAKY, ops.ACT_RELU])
def test_bn_bf16_storage_matches_fp32_kernels_on_rounded_input(N, C, H, act):
    """bf16 y / dz in, bf16 z / dy out: the same fp32 / fp64 arithmetic as the fp32 kernels applied to the rounded tensors
    -- statistics and parameter gradients equal at fp32 tolerance, z and dy equal after ONE rounding to bf16."""
    y, dz = rnd(N, C, H, H, seed=5) * 2 + 0.3, rnd(N, C, H, H, seed=6)
    gamma, beta = (rnd(C, seed=7) + 1.5).to(DEV), rnd(C, seed=8).to(DEV)
    y16, dz16 = nhwc16(y), nhwc16(dz)
    y32, dz32 = nhwc(y16.float().cpu()), nhwc(dz16.float().cpu())          # the SAME values in fp32 tensors
    rm32, rv32, nb32 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), dtype=torch.long, device=DEV)
    rm16, rv16, nb16 = rm32.clone(), rv32.clone(), nb32.clone()
    s32 = ops.bn_train_stats(y32, rm32, rv32, nb32, 1e-5, 0.1)
    s16 = ops.bn_train_stats(y16, rm16, rv16, nb16, 1e-5, 0.1)
    close(s16, s32, rtol=2e-6, what="bf16-storage BN statistics")
    close(rm16, rm32, rtol=2e-6, what="running_mean")
    close(rv16, rv32, rtol=2e-6, what="running_var")
    assert int(nb16) == 1
    z32 = ops.bn_act_fwd(y32, s32, gamma, beta, act, 0.2)
    z16 = ops.bn_act_fwd(y16, s32, gamma, beta, act, 0.2)
    assert z16.dtype == torch.bfloat16 and ops.is_nhwc(z16)
    close16(z16, z32, what="bf16-storage BN apply")
    dy32, dg32, db32 = ops.bn_act_bwd(dz32, y32, s32, gamma, beta, act, 0.2)
    dy16, dg16, db16 = ops.bn_act_bwd(dz16, y16, s32, gamma, beta, act, 0.2)
    assert dy16.dtype == torch.bfloat16
    close16(dy16, dy32, what="bf16-storage BN backward dy")
    close(dg16, dg32, rtol=1e-5, what="dgamma")
    close(db16, db32, rtol=1e-5, what="dbeta")


@pytest.mark.parametrize("N,C,K,H", [(2, 128, 256, 8), (4, 128, 256, 16), (3, 256, 320, 16), (2, 64, 128, 32), (2, 128, 64, 16)])
def test_conv_bf16_in_bf16_out(N, C, K, H):
    """bf16 activations in, bf16 activations out (register-staged and LDS-DMA kernels, split-K reduction included): the
    fp32 result of the same kernels, rounded once."""
    x, w, dy = rnd(N, C, H, H, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C)), rnd(N, K, H // 2, H // 2, seed=3)
    _lib.set_option("bf16", 1)
    ops.SHADOW = True
    try:
        x16, dy16, wg = nhwc16(x), nhwc16(dy), _with_shadow(krsc(w))
        x32, dy32 = _with_shadow(nhwc(x16.float().cpu())), _with_shadow(nhwc(dy16.float().cpu()))
        y_ref, dx_ref, dw_ref = ops.conv_fwd(x32, wg, 2, 1), ops.conv_dgrad(dy32, wg, (H, H), 2, 1), ops.conv_wgrad(dy32, x32, 2, 1)
        assert y_ref.dtype == torch.float32
        ops.ACT16 = True
        y, dx, dw = ops.conv_fwd(x16, wg, 2, 1), ops.conv_dgrad(dy16, wg, (H, H), 2, 1), ops.conv_wgrad(dy16, x16, 2, 1)
        assert y.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16 and dw.dtype == torch.float32
        close16(y, y_ref, what="conv fwd bf16 out")
        close16(dx, dx_ref, what="conv dgrad bf16 out")
        assert torch.equal(dw, dw_ref), "weight gradient from bf16 tensors must be bitwise the shadowed one"
    finally:
        ops.ACT16 = False
        ops.SHADOW = False
        ops.shadow_clear()
        _lib.set_option("bf16", 0)


def test_head_and_bottleneck_bf16_storage():
    """The K == 1 discriminator head with a bf16 feature map (plain reductions), and the generators' 100-channel bottleneck:
    bf16 map in -> fp32 [N,100] out -> bf16 map back (C = 100 is not a multiple of 8: that tensor stays fp32)."""
    N, C = 6, 512
    x, w1, dy1 = rnd(N, C, 4, 4, seed=1), rnd(1, C, 4, 4, seed=2, scale=0.02), rnd(N, 1, 1, 1, seed=3)
    w100, dy100 = rnd(100, C, 4, 4, seed=4, scale=0.02), rnd(N, 100, 1, 1, seed=5)
    _lib.set_option("bf16", 1)
    ops.SHADOW = True
    try:
        x16 = nhwc16(x)
        x32 = _with_shadow(nhwc(x16.float().cpu()))
        wg1, wg100 = krsc(w1), _with_shadow(krsc(w100))
        ref = (ops.conv_fwd(x32, wg1, 1, 0), ops.conv_dgrad(dy1.to(DEV), wg1, (4, 4), 1, 0), ops.conv_wgrad(dy1.to(DEV), x32, 1, 0),
               ops.conv_fwd(x32, wg100, 1, 0), ops.conv_dgrad(dy100.to(DEV), wg100, (4, 4), 1, 0), ops.conv_wgrad(dy100.to(DEV), x32, 1, 0))
        ops.ACT16 = True
        got = (ops.conv_fwd(x16, wg1, 1, 0), ops.conv_dgrad(dy1.to(DEV), wg1, (4, 4), 1, 0), ops.conv_wgrad(dy1.to(DEV), x16, 1, 0),
               ops.conv_fwd(x16, wg100, 1, 0), ops.conv_dgrad(dy100.to(DEV), wg100, (4, 4), 1, 0), ops.conv_wgrad(dy100.to(DEV), x16, 1, 0))
        assert got[0].dtype == torch.float32 and got[3].dtype == torch.float32
        assert got[1].dtype == torch.bfloat16 and got[4].dtype == torch.bfloat16
        close(got[0], ref[0], rtol=1e-5, what="head1 fwd")
        close16(got[1], ref[1], what="head1 dgrad")
        close(got[2], ref[2], rtol=1e-5, what="head1 wgrad")
        close(got[3], ref[3], rtol=1e-5, what="bottleneck fwd")
        close16(got[4], ref[4], what="bottleneck dgrad")
        close(got[5], ref[5], rtol=1e-5, what="bottleneck wgrad")
    finally:
        ops.ACT16 = False
        ops.SHADOW = False
        ops.shadow_clear()
        _lib.set_option("bf16", 0)


@pytest.mark.parametrize("N,S", [(2, 16), (3, 64), (1, 128)])
def test_edge_kernels_bf16_storage(N, S):
    """3-channel edge layers with the 64-channel side in bf16: forward output rounded once; input-grad / weight-grad equal
    the fp32 kernels on the rounded tensors at fp32 tolerance (with and without the fused LeakyReLU backward)."""
    x, w = torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(1)), rnd(64, 3, 4, 4, seed=2, scale=0.2)
    dy = rnd(N, 64, S // 2, S // 2, seed=3)
    xg, wg = x.to(DEV), w.to(DEV)
    y32 = ops.c3_fwd(xg, wg, ops.ACT_LEAKY, 0.2)
    ops.ACT16 = True
    try:
        y16 = ops.c3_fwd(xg, wg, ops.ACT_LEAKY, 0.2)
    finally:
        ops.ACT16 = False
    assert y16.dtype == torch.bfloat16 and ops.is_nhwc(y16)
    close16(y16, y32, what="c3_fwd bf16 out")
    dy16 = nhwc16(dy)
    dy32 = nhwc(dy16.float().cpu())
    yr32 = nhwc(y16.float().permute(0, 1, 2, 3).cpu())
    close(ops.c3_dgrad(dy16, wg, ops.ACT_SIGMOID), ops.c3_dgrad(dy32, wg, ops.ACT_SIGMOID), rtol=2e-5, what="c3_dgrad bf16 in")
    close(ops.c3_dgrad(dy16, wg, ops.ACT_NONE), ops.c3_dgrad(dy32, wg, ops.ACT_NONE), rtol=2e-5, what="c3_dgrad bf16 in (no act)")
    close(ops.c3_wgrad(dy16, xg), ops.c3_wgrad(dy32, xg), rtol=2e-5, what="c3_wgrad bf16 in")
    close(ops.c3_wgrad(dy16, xg, act_out=y16, act=ops.ACT_LEAKY, slope=0.2),
          ops.c3_wgrad(dy32, xg, act_out=yr32, act=ops.ACT_LEAKY, slope=0.2), rtol=2e-5, what="c3_wgrad_act bf16 in")
    g16 = ops.act_bwd(dy16, y16, ops.ACT_LEAKY, 0.2)
    g32 = ops.act_bwd(dy32, yr32, ops.ACT_LEAKY, 0.2)
    assert g16.dtype == torch.bfloat16
    close16(g16, g32, what="act_bwd bf16")


def test_fm_bf16_storage():
    N, C, H = 8, 128, 16
    r, f_ = rnd(N, C, H, H, seed=1), rnd(N, C, H, H, seed=2)
    r16, f16 = nhwc16(r), nhwc16(f_)
    r32, f32 = nhwc(r16.float().cpu()), nhwc(f16.float().cpu())
    l16, d16, _, _ = ops.fm_fwd(r16, f16)
    l32, d32, _, _ = ops.fm_fwd(r32, f32)
    close(l16.reshape(1), l32.reshape(1), rtol=1e-5, atol=1e-9, what="fm loss")
    close(d16, d32, rtol=1e-5, what="fm diff")
    gout = torch.tensor(0.7, device=DEV)
    a16, b16 = ops.fm_bwd(d16, r16, f16, gout, True, True)
    a32, b32 = ops.fm_bwd(d32, r32, f32, gout, True, True)
    assert a16.dtype == torch.bfloat16 and b16.dtype == torch.bfloat16
    close16(a16, a32, what="fm dreal")
    close16(b16, b32, what="fm dfake")


@pytest.mark.parametrize("N,S", [(2, 16), (3, 64), (1, 128), (5, 32)])
def test_edge_kernels_on_the_bf16_matrix_path(N, S):
    """Option "bf16" = 1: the 3-channel forward and (with a bf16 dy) the last-convT forward / conv1 input-grad run on the bf16
    MFMA with image / dy and weights rounded to bf16 (RNE) -- they must equal the fp64 convolution of the ROUNDED operands at
    fp32 tolerance (bf16 x bf16 products are exact in fp32), padding rows / columns and ragged groups included."""
    r = lambda t_: t_.bfloat16().float()
    x, w = torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(1)), rnd(64, 3, 4, 4, seed=2, scale=0.2)
    dy = rnd(N, 64, S // 2, S // 2, seed=3)
    yr = TF.leaky_relu(TF.conv2d(r(x).double(), r(w).double(), stride=2, padding=1), 0.2).float()
    yr_none = TF.conv2d(r(x).double(), r(w).double(), stride=2, padding=1).float()
    dxr = TF.conv_transpose2d(r(dy).double(), r(w).double(), stride=2, padding=1)
    xg, wg = x.to(DEV), w.to(DEV)
    _lib.set_option("bf16", 1)
    try:
        close(ops.c3_fwd(xg, wg, ops.ACT_LEAKY, 0.2), yr, what="c3_fwd bf16 MFMA, fp32 out")
        close(ops.c3_fwd(xg, wg, ops.ACT_NONE, 0.2), yr_none, what="c3_fwd bf16 MFMA, no act")
        ops.ACT16 = True
        y16 = ops.c3_fwd(xg, wg, ops.ACT_LEAKY, 0.2)
        ops.ACT16 = False
        assert y16.dtype == torch.bfloat16
        close16(y16, yr, extra=2e-5, what="c3_fwd bf16 MFMA, bf16 out")
        dy16 = nhwc16(dy)
        close(ops.c3_dgrad(dy16, wg, ops.ACT_NONE), dxr.float(), rtol=2e-5, what="c3_dgrad bf16 MFMA")
        close(ops.c3_dgrad(dy16, wg, ops.ACT_SIGMOID), torch.sigmoid(dxr).float(), rtol=2e-5, what="c3_dgrad bf16 MFMA + sigmoid")
        # weight gradient on the bf16 MFMA: rounded image x bf16 dy; with the fused LeakyReLU backward dy * act'(y) is rounded again
        dwr = torch.nn.grad.conv2d_weight(r(x).double(), w.shape, r(dy).double(), stride=2, padding=1).float()
        close(ops.c3_wgrad(dy16, xg), dwr, rtol=2e-5, what="c3_wgrad bf16 MFMA")
        gm = (dy16.float() * torch.where(y16.float() > 0, 1.0, 0.2)).bfloat16().float().cpu()
        dwr2 = torch.nn.grad.conv2d_weight(r(x).double(), w.shape, gm.double(), stride=2, padding=1).float()
        close(ops.c3_wgrad(dy16, xg, act_out=y16, act=ops.ACT_LEAKY, slope=0.2), dwr2, rtol=2e-5, what="c3_wgrad_act bf16 MFMA")
        acc0 = torch.ones(64, 3, 4, 4, device=DEV)
        close(ops.c3_wgrad(dy16, xg, out=acc0, accumulate=True), dwr + 1.0, rtol=2e-5, what="c3_wgrad bf16 MFMA accumulate")
        # the fp32-MFMA kernels stay reachable on this path (option kt = 16 for the forward)
        _lib.set_option("kt", 16)
        close(ops.c3_fwd(xg, wg, ops.ACT_LEAKY, 0.2), TF.leaky_relu(TF.conv2d(x.double(), w.double(), stride=2, padding=1), 0.2).float(), what="c3_fwd fp32 MFMA")
    finally:
        ops.ACT16 = False
        _lib.set_option("kt", 0)
        _lib.set_option("bf16", 0)


@pytest.mark.parametrize("N,S", [(2, 16), (3, 64), (1, 256)])
def test_edge_forward_on_the_f32x3_path(N, S):
    """Option "bf16" = 2: the 3-channel forward (conv1, and the last transposed conv's input-grad) splits image and weights into
    their three bf16 planes in registers and multiplies with six bf16 MFMAs per block -- fp32-accurate: at least as close to the
    fp64 convolution of the UNROUNDED operands as the fp32-MFMA kernel (<= 2x its distance), padding and ragged groups included."""
    x, w = torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(1)), rnd(64, 3, 4, 4, seed=2, scale=0.2)
    y64 = TF.conv2d(x.double(), w.double(), stride=2, padding=1)
    xg, wg = x.to(DEV), w.to(DEV)
    y32 = ops.c3_fwd(xg, wg, ops.ACT_NONE, 0.2)
    _lib.set_option("bf16", 2)
    try:
        y3 = ops.c3_fwd(xg, wg, ops.ACT_NONE, 0.2)
        y3l = ops.c3_fwd(xg, wg, ops.ACT_LEAKY, 0.2)
    finally:
        _lib.set_option("bf16", 0)
    e32 = ((y32.double().cpu() - y64).norm() / y64.norm()).item()
    e3 = ((y3.double().cpu() - y64).norm() / y64.norm()).item()
    assert e3 <= 2 * e32 + 2e-8, f"f32x3 edge forward: relative L2 {e3:.2e} vs fp64 (fp32 MFMA kernel: {e32:.2e})"
    close(y3, y64.float(), rtol=2e-6, what="c3_fwd f32x3")
    close(y3l, TF.leaky_relu(y64, 0.2).float(), rtol=2e-6, what="c3_fwd f32x3 + LeakyReLU")


@pytest.mark.parametrize("N,H", [(2, 8), (3, 16), (2, 64), (1, 256), (5, 32)])
def test_edge_dgrad_on_the_f32x3_path(N, H):
    """Option "bf16" = 2: the last ConvTranspose2d(64,3,4,2,1) (+ Sigmoid) / conv1 input-grad in the scatter form with fp32-accurate
    products on the bf16 MFMA (dy and weights split into three bf16 planes in registers, six MFMAs per block): at least as close to
    the fp64 result as the fp32-MFMA kernel (<= 2x its distance), ragged tiles and zero halo included."""
    x = rnd(N, 64, H // 2, H // 2, seed=5)
    w = rnd(64, 3, 4, 4, seed=6, scale=0.1)
    raw64 = TF.conv_transpose2d(x.double(), w.double(), stride=2, padding=1)
    xg, wg = nhwc(x), w.to(DEV)
    d32 = ops.c3_dgrad(xg, wg, ops.ACT_NONE)
    _lib.set_option("bf16", 2)
    try:
        d3 = ops.c3_dgrad(xg, wg, ops.ACT_NONE)
        d3s = ops.c3_dgrad(xg, wg, ops.ACT_SIGMOID)
    finally:
        _lib.set_option("bf16", 0)
    e32 = ((d32.double().cpu() - raw64).norm() / raw64.norm()).item()
    e3 = ((d3.double().cpu() - raw64).norm() / raw64.norm()).item()
    assert e3 <= 2 * e32 + 2e-8, f"f32x3 edge dgrad: relative L2 {e3:.2e} vs fp64 (fp32 MFMA kernel: {e32:.2e})"
    close(d3, raw64.float(), rtol=2e-6, what="c3_dgrad f32x3")
    close(d3s, torch.sigmoid(raw64).float(), rtol=2e-6, atol=1e-6, what="c3_dgrad f32x3 + sigmoid")


@pytest.mark.parametrize("N,S", [(2, 16), (3, 64), (1, 256)])
def test_edge_forward_writes_the_plane_triple_of_its_output(N, S):
    """f32x3 plane path: conv1's forward kernel also writes hi / mid / lo of its output (dg_conv4x4s2_c3_fwd_x3), so the next layer's
    weight-gradient needs no separate split pass.  The fp32 output is the plain f32x3 kernel's bit for bit (the two accumulator
    blocks hold even / odd channels instead of halves: same products, same order per element), the planes are dg_f32_to_bf16x3 of it."""
    x, w = torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(1)).to(DEV), rnd(64, 3, 4, 4, seed=2, scale=0.2).to(DEV)
    _lib.set_option("bf16", 2)
    ops.X3 = True
    try:
        for act in (ops.ACT_LEAKY, ops.ACT_NONE):
            y0 = ops.c3_fwd(x, w, act, 0.2)
            y1 = ops.c3_fwd(x, w, act, 0.2, want_planes=True)
            t3 = ops._PLANE_TAB[y1.data_ptr()][1]
            assert torch.equal(y0, y1)
            ref = ops.f32_to_bf16x3(y1, torch.empty((3, y1.numel()), device=DEV, dtype=torch.bfloat16))
            assert torch.equal(t3, ref) and torch.equal(t3.float().sum(0), y1.permute(0, 2, 3, 1).reshape(-1))
    finally:
        ops.X3 = False
        ops.planes_clear()
        _lib.set_option("bf16", 0)


X3_FWW_SHAPES = [
    (1, 64, 128, 256),     # the benchmark layer: Wo = 128, two output rows per tile, 16 super-chunks
    (3, 96, 104, 128),     # ragged: 104 of 128 columns, 6 chunks; Wo = 64
    (1, 32, 8, 64),        # two chunks, 8 columns; Wo = 32: eight output rows per tile
    (2, 128, 64, 64),      # 8 chunks, 64 columns
]


@needs_experiments
@pytest.mark.slow
@pytest.mark.parametrize("N,C,K,H", X3_FWW_SHAPES)
def test_conv_f32x3_forward_window_kernel(N, C, K, H, monkeypatch):
    """Forward with few output channels on plane operands (csrc/igemm_dma_x3_fww.hip): the input window of one (16-channel chunk,
    input-parity class) in LDS, re-used by the class's four taps; transposed weight planes.  fp32 tolerance against the fp64
    convolution, at most 2x the exact-fp32 MFMA kernel's distance from it, borders (the conv's zero padding = out-of-range window
    pixels) and ragged column counts included; without transposed weight planes the register-staged tiles take the shape."""
    x, w = rnd(N, C, H, H, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C))
    y64 = TF.conv2d(x.double(), w.double(), stride=2, padding=1)
    xg, wg = nhwc(x), krsc(w)
    L = _lib.load()
    _lib.set_option("splitk", 1)               # the window kernel has no split-K: compare with UNSPLIT reductions (slab sums are pairwise-like
    e32 = _rel(ops.conv_fwd(xg, wg, 2, 1), y64)   # and would flatter the other kernels on these small test shapes)
    _lib.set_option("bf16", 2)
    try:
        yreg = ops.conv_fwd(xg, wg, 2, 1)
        assert L.dg_conv_x3_planes_ok(0, N, H, H, C, K, 2, 1) == 3
        ops.X3 = True
        monkeypatch.setattr(ops, "X3_FWW", True)               # (off by default: not faster than the register-staged tiles at 512 px)
        y_nt = ops.conv_fwd(xg, wg, 2, 1)                      # no transposed planes: register-staged tiles, same bits
        assert torch.equal(y_nt, yreg)
        buf = torch.empty((3, wg.numel()), device=DEV, dtype=torch.bfloat16)
        wg._dg_x3, wg._dg_x3_ver = (buf, 0, torch.zeros_like(buf)), None
        ops.planes_clear()
        y = ops.conv_fwd(xg, wg, 2, 1)
        torch.cuda.synchronize()
        assert len(ops._PLANE_TAB) == 1                        # x was split into planes: the window kernel ran
    finally:
        ops.X3 = False
        ops.planes_clear()
        _lib.set_option("bf16", 0)
        _lib.set_option("splitk", 0)
    close(y, y64.float(), what="window forward")
    close(y, yreg, rtol=2e-5, atol=2e-6, what="window forward vs register-staged f32x3")
    e3, ereg = _rel(y, y64), _rel(yreg, y64)
    print(f"[{N},{C},{K},{H}] rel L2 vs fp64: exact-fp32 MFMA {e32:.2e}, register-staged f32x3 {ereg:.2e}, window forward {e3:.2e}")
    assert e3 <= 2.0 * max(e32, ereg) + 2e-7, f"window forward: {e3:.2e} vs fp64 (exact-fp32 MFMA kernel {e32:.2e}, register-staged f32x3 {ereg:.2e})"


@pytest.mark.parametrize("N,H", [(2, 16), (3, 64), (1, 256), (5, 32)])
@pytest.mark.parametrize("mode", ["f32", "f32x3", "bf16_mfma", "bf16_storage"])
def test_c3_dgrad_takes_the_activation_backward_in_its_load_path(N, H, mode):
    """dg_conv4x4s2_c3_dgrad_act_p: conv1's input-gradient with dy * (out > 0 ? 1 : slope) applied while dy is loaded (the saved output
    prefetched with dy's offsets) must be BITWISE the stand-alone act_bwd pass followed by the plain input-gradient -- on the fp32 MFMA,
    the f32x3 form, the bf16 MFMA with a bf16 dy and the fp32 MFMA with a bf16 dy; ragged tiles and zero padding included."""
    w = (rnd(64, 3, 4, 4, seed=2) * 0.1).to(DEV)
    dy, out = rnd(N, 64, H // 2, H // 2, seed=3), rnd(N, 64, H // 2, H // 2, seed=4)
    out[0, :, 0, 0] = 0.0                                  # out == 0 takes the slope, like leaky_relu_backward on the result
    if mode.startswith("bf16"):
        dyg, outg = nhwc16(dy), nhwc16(out)
    else:
        dyg, outg = nhwc(dy), nhwc(out)
    assert ops.c3_dgrad_act_ok(64)
    _lib.set_option("bf16", {"f32": 0, "f32x3": 2, "bf16_mfma": 1, "bf16_storage": 0}[mode])
    try:
        ref = ops.c3_dgrad(ops.act_bwd(dyg, outg, ops.ACT_LEAKY, 0.2), w, ops.ACT_NONE)
        got = ops.c3_dgrad(dyg, w, ops.ACT_NONE, act_out=outg, in_act=ops.ACT_LEAKY, slope=0.2)
        refs = ops.c3_dgrad(ops.act_bwd(dyg, outg, ops.ACT_LEAKY, 0.2), w, ops.ACT_SIGMOID)
        gots = ops.c3_dgrad(dyg, w, ops.ACT_SIGMOID, act_out=outg, in_act=ops.ACT_LEAKY, slope=0.2)
    finally:
        _lib.set_option("bf16", 0)
    assert torch.equal(got, ref) and torch.equal(gots, refs)
    assert float(ref.abs().max()) > 0
    _lib.set_option("kt", 16)                              # the gather form has no fused variant: the query says so and the call refuses
    try:
        assert not ops.c3_dgrad_act_ok(64)
        with pytest.raises(_lib.DiscoganHipError, match="scatter kernel"):
            ops.c3_dgrad(nhwc(dy), w, ops.ACT_NONE, act_out=nhwc(out), in_act=ops.ACT_LEAKY, slope=0.2)
    finally:
        _lib.set_option("kt", 0)


@pytest.mark.parametrize("N,S", [(2, 64), (3, 128), (2, 256), (1, 512)])
def test_edge_wgrad_on_the_f32x3_path(N, S):
    """(From 256-pixel rows on; shorter rows run the exact-fp32 MFMA kernel on this path too, which is faster there -- round 4.)
    Option "bf16" = 2: conv1's weight gradient (and, roles swapped, the last transposed conv's) with the image rows staged through
    LDS in fp32 and both operands split into three bf16 planes in front of the MFMAs (c3_wgrad_lds_kernel<., X3>): at least as close
    to the fp64 result as the fp32-MFMA kernel (<= 2x its distance), with the fused LeakyReLU backward and accumulation."""
    x, dy = torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(1)), rnd(N, 64, S // 2, S // 2, seed=3)
    y = rnd(N, 64, S // 2, S // 2, seed=4)                       # a saved activation output: its sign selects the slope
    dw64 = torch.nn.grad.conv2d_weight(x.double(), (64, 3, 4, 4), dy.double(), stride=2, padding=1)
    gm = dy.double() * torch.where(y > 0, 1.0, 0.2).double()
    dw64a = torch.nn.grad.conv2d_weight(x.double(), (64, 3, 4, 4), gm, stride=2, padding=1)
    xg, dyg, yg = x.to(DEV), nhwc(dy), nhwc(y)
    e32 = _rel(ops.c3_wgrad(dyg, xg), dw64)
    e32a = _rel(ops.c3_wgrad(dyg, xg, act_out=yg, act=ops.ACT_LEAKY, slope=0.2), dw64a)
    _lib.set_option("bf16", 2)
    try:
        d3 = ops.c3_wgrad(dyg, xg)
        d3a = ops.c3_wgrad(dyg, xg, act_out=yg, act=ops.ACT_LEAKY, slope=0.2)
        acc = ops.c3_wgrad(dyg, xg, out=torch.ones(64, 3, 4, 4, device=DEV), accumulate=True)
    finally:
        _lib.set_option("bf16", 0)
    e3, e3a = _rel(d3, dw64), _rel(d3a, dw64a)
    print(f"[{N},{S}] c3 wgrad rel L2 vs fp64: fp32 MFMA {e32:.2e} / {e32a:.2e} (fused act), f32x3 {e3:.2e} / {e3a:.2e}")
    assert e3 <= 2 * e32 + 2e-7 and e3a <= 2 * e32a + 2e-7
    close(d3, dw64.float(), rtol=2e-5, what="c3_wgrad f32x3")
    close(d3a, dw64a.float(), rtol=2e-5, what="c3_wgrad_act f32x3")
    close(acc, dw64.float() + 1.0, rtol=2e-5, what="c3_wgrad f32x3 accumulate")


@pytest.mark.parametrize("N,C,K,H,op", [(2, 64, 256, 64, 0), (1, 256, 512, 32, 0), (2, 256, 64, 64, 1), (1, 128, 256, 128, 1), (2, 512, 256, 32, 1)])
def test_plane_kernels_emit_batchnorm_partial_statistics(N, C, K, H, op):
    """f32x3 plane path: the plane kernels' epilogues (256 x 256 tile, the window input-grad kernel, or their split-K reduction) emit
    BatchNorm partial rows of the conv OUTPUT; merged by dg_bn_stats_from_partials they must give the statistics of the separate
    pass (dg_bn_train_stats) over the same output, and identical running-statistics updates."""
    x, w = rnd(N, C, H, H, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C))
    dy = rnd(N, K, H // 2, H // 2, seed=3)
    xg, wg, dyg = nhwc(x), krsc(w), nhwc(dy)
    L = _lib.load()
    _lib.set_option("bf16", 2)
    ops.X3 = True
    try:
        buf = torch.empty((3, wg.numel()), device=DEV, dtype=torch.bfloat16)
        wg._dg_x3, wg._dg_x3_ver = (buf, 0, torch.zeros_like(buf)), None
        if op == 0:
            assert L.dg_conv_x3_bnstats_rows(0, N, H, H, C, K, 2, 1) > 0
            y, stat = ops.conv_fwd(xg, wg, 2, 1, want_stats=True)
            y0 = ops.conv_fwd(xg, wg, 2, 1)
        else:
            assert L.dg_conv_x3_bnstats_rows(1, N, H, H, C, K, 2, 1) > 0
            y, stat = ops.conv_dgrad(dyg, wg, (H, H), 2, 1, want_stats=True)
            y0 = ops.conv_dgrad(dyg, wg, (H, H), 2, 1)
        assert stat is not None
        # unsplit reductions: the same launch, the same bits; with split-K the statistics come out of a reduction kernel that sums
        # the slabs in its own (fixed) order -- outputs agree to fp32 rounding
        close(y, y0, rtol=1e-6, atol=1e-7, what="conv output with / without fused statistics")
        ch = y.shape[1]
        rm1, rv1, nb1 = torch.zeros(ch, device=DEV), torch.ones(ch, device=DEV), torch.zeros((), device=DEV, dtype=torch.int64)
        rm2, rv2, nb2 = torch.zeros(ch, device=DEV), torch.ones(ch, device=DEV), torch.zeros((), device=DEV, dtype=torch.int64)
        s_sep = ops.bn_train_stats(y, rm1, rv1, nb1, 1e-5, 0.1)
        s_fus = ops.bn_stats_from_partials(stat, y, rm2, rv2, nb2, 1e-5, 0.1)
    finally:
        ops.X3 = False
        ops.planes_clear()
        _lib.set_option("bf16", 0)
    close(s_fus[0], s_sep[0], rtol=1e-5, atol=1e-6, what="fused mean")
    close(s_fus[1], s_sep[1], rtol=1e-5, atol=1e-6, what="fused invstd")
    close(rm2, rm1, rtol=1e-5, atol=1e-7, what="running mean")
    close(rv2, rv1, rtol=1e-5, atol=1e-7, what="running var")
    assert int(nb1) == int(nb2) == 1


@pytest.mark.parametrize("N,C,K,H,op", [(2, 64, 256, 64, 0), (1, 256, 512, 32, 0), (32, 512, 1024, 16, 0), (2, 64, 128, 64, 0),
                                        (2, 256, 64, 64, 1), (1, 128, 256, 128, 1), (2, 512, 256, 32, 1), (32, 1024, 512, 16, 1)])
def test_bf16_kernels_emit_batchnorm_partial_statistics(N, C, K, H, op):
    """bf16 matrix path with bf16-stored feature maps (BASELINE configs[4]): the bf16 conv kernels (LDS-DMA kernel, window input-grad
    kernel, register-staged tiles, split-K reduction) emit the BatchNorm partial rows of their output from the fp32 ACCUMULATORS,
    before the output is rounded to bf16 (stat argument of dg_conv_fwd_mixed / dg_conv_dgrad_mixed).  Merged, they must equal the
    statistics of the separate pass over the UNROUNDED (fp32) output of the same kernel to fp32 rounding, the stored bf16 output
    must not change, and against the statistics of the rounded output they differ by no more than the rounding carries."""
    x, w = rnd(N, C, H, H, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C))
    dy = rnd(N, K, H // 2, H // 2, seed=3)
    xg, wg, dyg = nhwc(x).bfloat16(), krsc(w), nhwc(dy).bfloat16()
    xg, dyg = xg.contiguous(memory_format=torch.channels_last), dyg.contiguous(memory_format=torch.channels_last)
    L = _lib.load()
    _lib.set_option("bf16", 1)
    ops.SHADOW = True
    try:
        wg._dg_bf16, wg._dg_bf16_ver = wg.detach().bfloat16(), wg._version
        run = (lambda **kw: ops.conv_fwd(xg, wg, 2, 1, **kw)) if op == 0 else (lambda **kw: ops.conv_dgrad(dyg, wg, (H, H), 2, 1, **kw))
        ops.ACT16 = True
        y16, stat = run(want_stats=True)
        y16_plain = run()
        ops.ACT16 = False
        y32 = run()                                            # the same kernel, fp32 output
        assert stat is not None and y16.dtype == torch.bfloat16 and y32.dtype == torch.float32
        ch = y16.shape[1]
        mk = lambda: (torch.zeros(ch, device=DEV), torch.ones(ch, device=DEV), torch.zeros((), device=DEV, dtype=torch.int64))
        (rm1, rv1, nb1), (rm2, rv2, nb2), (rm3, rv3, nb3) = mk(), mk(), mk()
        s_sep = ops.bn_train_stats(y32, rm1, rv1, nb1, 1e-5, 0.1)
        s_fus = ops.bn_stats_from_partials(stat, y16, rm2, rv2, nb2, 1e-5, 0.1)
        s_rnd = ops.bn_train_stats(y16, rm3, rv3, nb3, 1e-5, 0.1)
    finally:
        ops.ACT16 = False
        ops.SHADOW = False
        ops.shadow_clear()
        _lib.set_option("bf16", 0)
    assert torch.equal(y16, y16_plain), "the stored output must not depend on the statistics epilogue"
    close(s_fus[0], s_sep[0], rtol=1e-5, atol=1e-6, what="fused mean vs separate pass over the fp32 output")
    close(s_fus[1], s_sep[1], rtol=1e-5, atol=1e-6, what="fused invstd")
    close(rm2, rm1, rtol=1e-5, atol=1e-7, what="running mean")
    close(rv2, rv1, rtol=1e-5, atol=1e-7, what="running var")
    sd = 1.0 / s_sep[1]
    assert float(((s_fus[0] - s_rnd[0]).abs() / sd).max()) < 2e-3, "mean vs statistics of the bf16-rounded output"
    close(s_fus[1], s_rnd[1], rtol=5e-3, what="invstd vs statistics of the bf16-rounded output")
    assert int(nb2) == 1


X3_RSP_SHAPES = [
    (1, 64, 128, 256),     # the benchmark layer (window-forward shape: code 3)
    (3, 96, 96, 32),       # Wo = 16: no window kernel (code 4); 6 chunks, 96 of 128 columns
    (2, 128, 64, 64),      # one 128 x 128 tile column half used
    (5, 32, 160, 16),      # Wo = 8, two column tiles (160 = 128 + 32), ragged rows (5 x 64 = 320 = 2.5 tiles)
    (32, 512, 128, 8),     # deep reduction (8192): split-K
]


@needs_experiments
@pytest.mark.slow
@pytest.mark.parametrize("N,C,K,H", X3_RSP_SHAPES)
@pytest.mark.parametrize("splitk", [0, 1])
def test_conv_f32x3_register_staged_tiles_read_planes(N, C, K, H, splitk, monkeypatch):
    """Forward convs with fewer than 192 output channels on the f32x3 plane path (ops.X3_RSP): the register-staged 128 x 128 tiles
    load the operands' PLANE TRIPLES (igemm_kernel<.., PREC 2, A16, B16>) instead of splitting fp32 operands in the kernel.  The
    planes are exactly what the in-kernel split computes and the MFMA sequence is the same: BIT-IDENTICAL to the splitting kernel,
    borders, ragged tiles and split-K included."""
    x, w = rnd(N, C, H, H, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C))
    y64 = TF.conv2d(x.double(), w.double(), stride=2, padding=1)
    xg, wg = nhwc(x), krsc(w)
    L = _lib.load()
    _lib.set_option("bf16", 2)
    _lib.set_option("splitk", splitk)
    try:
        yreg = ops.conv_fwd(xg, wg, 2, 1)                     # fp32 operands, split in the kernel
        assert L.dg_conv_x3_planes_ok(0, N, H, H, C, K, 2, 1) in (3, 4)
        ops.X3 = True
        monkeypatch.setattr(ops, "X3_RSP", False)
        y_off = ops.conv_fwd(xg, wg, 2, 1)
        assert len(ops._PLANE_TAB) == 0 and torch.equal(y_off, yreg)       # switch off: nothing is split into planes
        monkeypatch.setattr(ops, "X3_RSP", True)
        y = ops.conv_fwd(xg, wg, 2, 1)
        torch.cuda.synchronize()
        assert len(ops._PLANE_TAB) == 1                        # x was split into planes once: the plane reader ran
    finally:
        ops.X3 = False
        ops.planes_clear()
        _lib.set_option("bf16", 0)
        _lib.set_option("splitk", 0)
    close(y, y64.float(), what="register-staged plane reader")
    assert torch.equal(y, yreg), "plane reader vs in-kernel split"


@needs_experiments
@pytest.mark.slow
@pytest.mark.parametrize("N,C,K,H,splitk", [(8, 64, 128, 256, 0), (16, 128, 256, 128, 0), (40, 64, 64, 64, 2), (5, 40, 64, 256, 0)])
def test_window_input_grad_persistent_workgroups(N, C, K, H, splitk):
    """Option "dgw_persist" 1, more tiles than CUs: the f32x3 window input-grad kernel runs ONE persistent workgroup per CU that walks
    the tiles and issues the next tile's first window / weight DMA in front of the finished tile's statistics and epilogue (which
    work in the other window stage).  Same products in the same order per tile: output and fused BatchNorm partial rows
    BIT-IDENTICAL to one workgroup per tile (the default: the persistent form measured no faster), 4- and 2-class forms, split-K, a
    ragged channel count."""
    w, dy = rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C)), rnd(N, K, H // 2, H // 2, seed=3)
    wg, dyg = krsc(w), nhwc(dy)
    L = _lib.load()
    _lib.set_option("bf16", 2)
    _lib.set_option("splitk", splitk)
    ops.X3 = True
    try:
        assert L.dg_conv_x3_planes_ok(1, N, H, H, C, K, 2, 1) == 2
        res = []
        for mode in (0, 1):
            _lib.set_option("dgw_persist", mode)
            dx, stat = ops.conv_dgrad(dyg, wg, (H, H), 2, 1, want_stats=True)
            dx2 = ops.conv_dgrad(dyg, wg, (H, H), 2, 1)
            torch.cuda.synchronize()
            res.append((dx.clone(), None if stat is None else stat.clone(), dx2.clone()))
    finally:
        _lib.set_option("dgw_persist", 0)
        ops.X3 = False
        ops.planes_clear()
        _lib.set_option("splitk", 0)
        _lib.set_option("bf16", 0)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2]), "persistent vs one workgroup per tile"
    assert (res[0][1] is None) == (res[1][1] is None)
    if res[0][1] is not None:
        # (columns 1..3 of a partial row are padding nobody writes)
        assert torch.equal(res[0][1][:, 0], res[1][1][:, 0]) and torch.equal(res[0][1][:, 4:], res[1][1][:, 4:]), "partial statistics rows"
    dx64 = TF.conv_transpose2d(dy.double(), w.double(), stride=2, padding=1)
    close(res[1][0], dx64.float(), rtol=2e-4, what="persistent window input-grad")
