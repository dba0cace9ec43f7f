"""Tensor-level wrappers over the C ABI (device pointers + sizes; torch only supplies memory/streams).

Layout conventions (see include/discogan_hip.h):
  * interior activations: logical [N,C,H,W] tensors whose MEMORY is NHWC (torch channels_last);
  * image side (3 channels): plain contiguous NCHW;
  * Conv2d weight logical [K,C,4,4] stored KRSC; ConvTranspose2d weight logical [Cin,Cout,4,4] stored
    [Cin,4,4,Cout]; 3-channel edge weights stay logically contiguous.
"""
from __future__ import annotations

import torch

from . import _lib

ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3
PREC_DEFAULT, PREC_F32, PREC_BF16, PREC_F32X3 = -1, 0, 1, 2      # include/discogan_hip.h DG_PREC_*


class Context:
    """The arithmetic and operand-form settings ONE caller (a trainer, a test) runs its ops under -- a per-caller object instead
    of process-wide switches ("dtype enum" per call, "no global mutable state").

      prec    arithmetic of the conv products, passed to the C ABI with EVERY call (DG_PREC_F32 | _BF16 | _F32X3); None = the
              library's process default (dg_set_option("bf16"): tools and op tests that flip it around single calls)
      shadow  bf16 path: conv kernels read bf16 shadows written by the producers (see "bf16 shadow operands" below)
      act16   bf16 path: feature maps and their gradients exist only as bf16
      x3      f32x3 path: conv kernels read plane triples written once per tensor
      shadow_tab / plane_tab: the derived copies of this caller's live activations

    ``with ops.use(ctx):`` makes it the ambient context of the forward calls issued inside; every autograd Function remembers the
    context of its forward and runs its backward under it (functional._fwd / _bwd), so two trainers with different arithmetic can be
    interleaved call by call in one process."""

    def __init__(self, prec=None, shadow=False, act16=False, x3=False, group_plan="launch"):
        self.prec = prec
        self.group_plan = group_plan           # grouped conv launches: split-K planned for the whole "launch" or per "single" problem
        self.shadow, self.act16, self.x3 = bool(shadow), bool(act16), bool(x3)
        self.shadow_tab, self.plane_tab = {}, {}

    @property
    def cprec(self):
        return PREC_DEFAULT if self.prec is None else int(self.prec)

    def clear(self):
        self.shadow_tab.clear()
        self.plane_tab.clear()


_AMBIENT = Context()          # what module-level ops.SHADOW / ops.ACT16 / ops.X3 read and write outside any ``use`` block
_CUR = _AMBIENT


def current():
    return _CUR


class use:
    """Context manager: run the ops issued inside under ``ctx``."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        global _CUR
        self.prev = _CUR
        _CUR = self.ctx
        return self.ctx

    def __exit__(self, *exc):
        global _CUR
        _CUR = self.prev
        return False

# bench.py's roofline leg sets this to a list: every launch of the MFMA implicit-GEMM family then
# appends (op, algorithmic FLOPs, start event, end event), recorded on the launch stream.
PROFILE = None


class _prof:
    def __init__(self, name, flops):
        self.on = PROFILE is not None
        if self.on:
            self.name, self.flops = name, flops
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            PROFILE.append((self.name, self.flops, self.e0, self.e1))
        return False


# Same hook for the HBM-bound kernel families: (family, algorithmic bytes, start event, end event).
PROFILE_HBM = None


class _hbm:
    def __init__(self, name, nbytes):
        self.on = PROFILE_HBM is not None
        if self.on:
            self.name, self.nbytes = name, nbytes
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            PROFILE_HBM.append((self.name, self.nbytes, self.e0, self.e1))
        return False


def _need_fp32(t, who):
    """Guard of the non-plane code paths: a plane-only tensor (X3_PLANES_ONLY) has no fp32 memory image to read."""
    if getattr(t, "_dg_planes_only", False):
        raise _lib.DiscoganHipError(f"{who}: the operand exists only as its plane triple, but this shape has no plane kernel")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _tab(ts):
    """Pointer table of a grouped C-ABI call: one device pointer per problem (None entries allowed)."""
    import ctypes as C
    if ts is None:
        return None
    return (C.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else None) for t in ts])


def _check_dev(*ts, allow16=False, planes_ok=False):
    """fp32 HIP tensors only.  allow16: the wrapper dispatches on the storage type itself (bf16 activation storage: the
    ``*_t`` / ``*_mixed`` entry points); every other wrapper calls an fp32-only kernel with numel() elements, where a bf16
    tensor would be read as floats past its end -- refused here instead."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.DiscoganHipError(
                "discogan_modernized_amd ops need CUDA/HIP tensors (no CPU fallback exists); got a CPU tensor")
        if t.dtype != torch.float32 and not (allow16 and t.dtype == torch.bfloat16):
            raise _lib.DiscoganHipError(("fp32 or bf16" if allow16 else "fp32") + f" tensors required, got {t.dtype}")
        if not planes_ok and getattr(t, "_dg_planes_only", False):
            raise _lib.DiscoganHipError("this tensor exists only as its plane triple (ops.X3_PLANES_ONLY): its fp32 memory was never "
                                        "written and this op would read it")


# ---- layout helpers -------------------------------------------------------------------------------------
def empty_nhwc(n, c, h, w, device, dtype=torch.float32):
    """Logical [n,c,h,w] tensor backed by NHWC memory."""
    return torch.empty((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


# bf16 ACTIVATION STORAGE (DiscoGANTrainer(mfma_dtype="bf16", act_dtype="bf16")): with ACT16 on, every feature map and
# feature-map gradient an op produces is a bf16 tensor (channel counts that are multiples of 8; the [N,100] bottleneck and
# everything image-shaped stay fp32), and the ops take bf16 tensors as they come -- there is no fp32 copy to shadow.
def _is16(t):
    return t is not None and t.dtype == torch.bfloat16


def _act_dtype(channels):
    return torch.bfloat16 if (_CUR.act16 and channels % 8 == 0) else torch.float32


def is_nhwc(x):
    return x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous()


def as_nhwc(x):
    """Return x (logical NCHW) with NHWC memory; converts with the library's own transpose kernel."""
    if is_nhwc(x):
        return x
    _check_dev(x)
    if x.dtype != torch.float32:
        raise _lib.DiscoganHipError("bf16 activations must already have NHWC memory")
    xc = x.contiguous()
    n, c, h, w = xc.shape
    y = empty_nhwc(n, c, h, w, x.device)
    _lib.check(_lib.load().dg_nchw_to_nhwc(_ptr(xc), _ptr(y), n, c, h, w, _stream()), "dg_nchw_to_nhwc")
    return y


def to_nchw_contiguous(x):
    """Logical NCHW tensor with plain contiguous memory (for export / comparisons)."""
    if x.is_contiguous():
        return x
    if is_nhwc(x):
        n, c, h, w = x.shape
        y = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().dg_nhwc_to_nchw(_ptr(x), _ptr(y), n, c, h, w, _stream()), "dg_nhwc_to_nchw")
        return y
    return x.contiguous()


def u8hwc_to_f32chw(x_u8, bgr=False):
    """uint8 [N,H,W,3] (decoded image rows) -> float32 contiguous NCHW [N,3,H,W] = pixel / 255 (dataset.py:65-66)."""
    if not x_u8.is_cuda or x_u8.dtype != torch.uint8 or x_u8.dim() != 4 or x_u8.shape[3] != 3:
        raise _lib.DiscoganHipError("u8hwc_to_f32chw needs a uint8 HIP tensor of shape [N,H,W,3]")
    x = x_u8.contiguous()
    n, h, w, _ = x.shape
    y = torch.empty((n, 3, h, w), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().dg_u8hwc_to_f32chw(_ptr(x), _ptr(y), n, h, w, int(bool(bgr)), _stream()), "dg_u8hwc_to_f32chw")
    return y


def krsc_param(w_logical):
    """[K,C,4,4] contiguous -> same logical tensor whose memory is [K,4,4,C] (dim 1 innermost)."""
    return w_logical.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def is_krsc(w):
    return w.dim() == 4 and w.permute(0, 2, 3, 1).is_contiguous()


def _krsc(w):
    return w if is_krsc(w) else krsc_param(w)


def empty_krsc(k, c, device):
    return torch.empty((k, 4, 4, c), device=device, dtype=torch.float32).permute(0, 3, 1, 2)


def _ws(nbytes, device):
    if nbytes == 0:
        return None, 0
    t = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
    return t, t.numel() * 4


# ---- bf16 shadow operands (mfma_dtype="bf16") -------------------------------------------------------------------
# A shadow is a bf16 (RNE) copy of an fp32 tensor in the same memory layout, written by the tensor's producer (Adam for
# weights, the BatchNorm kernels / first conv for activations and gradients).  With SHADOW on, the conv wrappers hand the
# shadows to dg_conv_*_mixed: half the operand bytes, no conversion in the conv kernel, bit-identical results.
#   weights    : ``param._dg_bf16`` (a view of optim.Adam's flat bf16 buffer), valid while ``param._version`` is unchanged
#   activations: side table keyed by the fp32 tensor's storage address; the entry holds the fp32 tensor, so the address
#                cannot be recycled while the entry lives; the trainer clears the table every iteration.
def shadow_clear():
    _CUR.shadow_tab.clear()


def shadow_put(t, t16):
    _CUR.shadow_tab[t.data_ptr()] = (t, t16)


def derived_release(*tensors):
    """Drop the bf16 shadow / plane triple of tensors whose conv consumers have all been issued (functional.py calls this
    after a layer's input-grad and weight-grad): the fp32 tensor and its copies then die with autograd's own references
    instead of living to the end of the iteration (or, under hipGraph capture, forever in the graph's pool)."""
    for t in tensors:
        if t is None:
            continue
        key = t.data_ptr()
        for tab in (_CUR.shadow_tab, _CUR.plane_tab):
            e = tab.get(key)
            if e is not None and e[0].shape == t.shape:
                del tab[key]


def shadow_get(t):
    if not _CUR.shadow:
        return None
    e = _CUR.shadow_tab.get(t.data_ptr())
    if e is None or e[0].shape != t.shape or e[0].stride() != t.stride():
        return None
    return e[1]


def weight_shadow(w):
    """bf16 shadow of a conv weight Parameter (None when shadows are off or the parameter is not in a flat Adam group)."""
    if not _CUR.shadow:
        return None
    w16 = getattr(w, "_dg_bf16", None)
    if w16 is None:
        return None
    if getattr(w, "_dg_bf16_ver", None) != w._version:          # e.g. load_state_dict wrote the fp32 weights
        f32_to_bf16(w, w16)
        w._dg_bf16_ver = w._version
    return w16


def f32_to_bf16(x, out):
    """out (bf16, same memory layout / numel as x) <- RNE(x)."""
    assert out.dtype == torch.bfloat16 and out.stride() == x.stride() and out.shape == x.shape
    _lib.check(_lib.load().dg_f32_to_bf16(_ptr(x), _ptr(out), x.numel(), _stream()), "dg_f32_to_bf16")
    return out


def empty_nhwc_bf16(n, c, h, w, device):
    return torch.empty((n, h, w, c), device=device, dtype=torch.bfloat16).permute(0, 3, 1, 2)


def _bf16_ok(op, n, h, wd, c, k, stride, pad):
    return _CUR.shadow and _lib.load().dg_conv_bf16_operands_ok(op, n, h, wd, c, k, stride, pad) >= 1


# ---- f32x3 plane operands (mfma_dtype="f32x3") -----------------------------------------------------------------
# A plane triple is three bf16 tensors hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid) of an fp32 tensor,
# plane-major ([3, numel] bf16).  With X3 on, the conv wrappers hand plane triples of BOTH operands to dg_conv_*_x3 where the
# shape has the plane kernel (csrc/igemm_dma_x3.hip); everything else keeps the fp32 tensors (register-staged f32x3 tiles).
#   weights    : ``param._dg_x3`` = (flat [3, P] bf16 buffer of optim.Adam, element offset), refreshed by the Adam kernel;
#                valid while ``param._version`` is unchanged (a foreign write re-splits the one weight)
#   activations: side table like the bf16 shadows; filled by the producers (BatchNorm kernels) or, for a tensor nobody has
#                split yet, by dg_f32_to_bf16x3 at its first use (the triple then serves the forward AND the weight gradient)
# X3_CM: the BatchNorm kernels write the plane triples that a WINDOW input-grad kernel will read (dy of the conv layers with
# <= 128 input channels, the input of the transposed convs with <= 128 output channels) in the QUAD-CHUNK layout
# ([pixels/4][C/16][4][16], include/discogan_hip.h "plane_layout"): the window kernel then uses every byte of the 128-byte lines
# it fetches instead of 32.  Bit-identical results; DG_X3_CM=0 keeps every triple pixel-major (A/B).
X3_CM = __import__("os").environ.get("DG_X3_CM", "1") != "0"
# X3_PLANES_ONLY: a BatchNorm output whose every reader is a plane kernel (the next conv's forward and weight-grad; the
# layer's own input-grad and weight-grad for a gradient) is written ONLY as its plane triple: the planes carry all 24 significand
# bits (hi + mid + lo == the fp32 value), the fp32 tensor stays allocated but unwritten and is flagged ``_dg_planes_only`` --
# every op wrapper that would read fp32 memory refuses such a tensor.  model.py decides per layer; DG_X3_PLANES_ONLY=0: off (A/B).
X3_PLANES_ONLY = __import__("os").environ.get("DG_X3_PLANES_ONLY", "1") != "0"
# POISON_PLANES_ONLY (debug aid, DG_POISON_PLANES_ONLY=1; round-3 advisor finding): the fp32 memory of a plane-only tensor is never written
# and its only protection is the ``_dg_planes_only`` attribute, which a view / a torch-native reader does not carry.  With the switch on that
# memory is filled with NaN, so anything that reads it (instead of the planes) turns the losses NaN instead of using stale allocator bytes;
# tests/test_model_gpu.py runs a training step under it and requires the results bitwise unchanged.
POISON_PLANES_ONLY = __import__("os").environ.get("DG_POISON_PLANES_ONLY", "0") == "1"
# X3_FWW: forward convolutions with <= 128 output channels on the window forward kernel (csrc/igemm_dma_x3_fww.hip; planner code 3:
# needs the transposed weight planes).  OFF by default: built, parity-tested and measured at the benchmark shape (64 -> 128 channels,
# 512 px / batch 32) at 0.923 ms against 0.927 ms for the register-staged f32x3 tiles -- its window pixels are every OTHER pixel of a
# pixel-major plane, 32 bytes of each 128-byte line per fetch, and those loads cost it 0.28 of its 0.92 ms.
# DG_X3_FWW=1 turns it on.
X3_FWW = __import__("os").environ.get("DG_X3_FWW", "0") == "1"
# X3_FUSE_STATS: on the plane path the BatchNorm batch statistics come out of the producing conv kernel's epilogue (or its split-K
# reduction) as partial rows, merged by dg_bn_stats_from_partials -- no separate read pass over the conv output.  (On the exact-fp32
# path of the 64 px network the same idea lengthened 140 us kernels by more than the pass it saved: model.FUSE_BN_STATS fuses only
# launches of 40 GFLOP and more there; the plane kernels run for hundreds of microseconds per tile and do not notice ~400 VALU instructions per wave.)
X3_FUSE_STATS = __import__("os").environ.get("DG_X3_FUSE_STATS", "1") != "0"
# FUSE_STATS16: the bf16 matrix path takes its BatchNorm statistics from the bf16 conv
# kernels' fp32 accumulators too (LDS-DMA kernel, window input-grad, register-staged tiles, split-K reduction: stat argument of
# dg_conv_fwd_mixed / _dgrad_mixed) instead of dropping to the exact-fp32 kernels for the fused layers.
FUSE_STATS16 = __import__("os").environ.get("DG_FUSE_BN16", "1") != "0"
# X3_MFMA: the MFMA shape of the plane kernel's 256 x 256 tile (library option "x3_mfma").  16 = v_mfma_f32_16x16x32_bf16 with the
# planes paired along k (three instructions per 16 x 16 block and K-tile instead of six 32x32x16 ones: the same products, a shape
# under which the chip holds a higher clock; csrc/igemm_dma_x3.hip); 32 / 0 = the 32x32x16 body (default: the paired body is faster
# per launch on the middle layers and not at all in the whole step).  The
# environment variable is applied by _lib.load() like every DG_OPT_*; this constant is what tests restore the option to.
X3_MFMA = int(__import__("os").environ.get("DG_OPT_X3_MFMA", "0"))


# X3_RSP (off: measured 45 % SLOWER): forward convs without an LDS-DMA plane kernel (fewer than 192 output channels: the 64 -> 128
# layer and, same GEMM, the input-grad of the 128 -> 64 transposed conv) on the register-staged 128 x 128 tiles READING the plane
# triples their producers wrote (dg_conv_x3_planes_ok codes 3 / 4: igemm_kernel<.., PREC 2, A16, B16>) instead of splitting fp32
# operands inside the kernel.  Bit-identical, a quarter fewer instructions -- and 1.36 instead of 0.94 ms at 512 px / batch 32: a
# 16-channel K-tile of a pixel-major plane is a 32-byte piece of a 128-byte line, three planes make it three times as many load
# instructions of half the useful width (the fp32 operand: 64-byte pieces).  DG_X3_RSP=1 switches it on.  DESIGN.md 3.1.
X3_RSP = __import__("os").environ.get("DG_X3_RSP", "0") == "1"


def _plane_code_ok(code):
    return code in (1, 2) or (code == 3 and (X3_FWW or X3_RSP)) or (code == 4 and X3_RSP)


def planes_clear():
    _CUR.plane_tab.clear()


def planes_put(t, t3, cm=False):
    _CUR.plane_tab[t.data_ptr()] = (t, t3, bool(cm))


def f32_to_bf16x3(x, out3):
    """out3 ([3, P] bf16, P >= x.numel()) <- the plane triple of x's memory image."""
    assert out3.dtype == torch.bfloat16 and out3.dim() == 2 and out3.shape[0] == 3 and out3.stride(1) == 1
    _lib.check(_lib.load().dg_f32_to_bf16x3(_ptr(x), _ptr(out3), x.numel(), out3.stride(0), _stream()), "dg_f32_to_bf16x3")
    return out3


def planes_of(t, allow_cm=False):
    """(plane-0 address, plane distance in bytes, chunk-major?) of an fp32 activation / gradient tensor; splits it on first
    use.  A chunk-major triple (written by a BatchNorm kernel for a window input-grad kernel) is only handed to callers that
    can read it (allow_cm); anybody else gets a pixel-major split of the fp32 tensor."""
    e = _CUR.plane_tab.get(t.data_ptr())
    if e is None or e[0].shape != t.shape or e[0].stride() != t.stride() or (e[2] and not allow_cm):
        if getattr(t, "_dg_planes_only", False):
            raise _lib.DiscoganHipError("plane-only tensor without a usable plane triple (released, or in a layout this reader cannot take)")
        t3 = torch.empty((3, t.numel()), device=t.device, dtype=torch.bfloat16)
        with _hbm("x3_split", 10.0 * t.numel()):
            f32_to_bf16x3(t, t3)
        if e is None or not e[2]:
            planes_put(t, t3)
        return t3.data_ptr(), t3.stride(0) * 2, 0
    return e[1].data_ptr(), e[1].stride(0) * 2, int(e[2])


def x3_all_plane_readers(n, h, wd, c, k, forward_is_dgrad=False, need_wgrad=True):
    """Are the conv kernels that will READ a tensor all plane kernels?  For a layer input x of Conv2d(c, k, 4, 2, 1) on [n, c, h, wd]:
    its forward (op 0) and weight-grad (op 2); with forward_is_dgrad (ConvTranspose2d: the same geometry read as an input-grad): op 1
    and op 2.  For a gradient dy: the layer's input-grad (op 1) and weight-grad -- the same question with forward_is_dgrad."""
    if not (_CUR.x3 and X3_PLANES_ONLY):
        return False
    L = _lib.load()
    first = 1 if forward_is_dgrad else 0
    return _plane_code_ok(L.dg_conv_x3_planes_ok(first, n, h, wd, c, k, 2, 1)) and (not need_wgrad or L.dg_conv_x3_planes_ok(2, n, h, wd, c, k, 2, 1) >= 1)


def x3_window_dgrad(n, h, wd, c, k):
    """Will the input-grad of Conv2d(c, k, 4, 2, 1) on an [n, c, h, wd] input (= the forward of the transposed conv with the same
    weight) run on the window kernel AND its weight gradient on the plane kernel?  Then the producer of the gradient operand
    writes chunk-major planes (X3_CM)."""
    if not (_CUR.x3 and X3_CM) or k % 64 != 0:
        return False
    L = _lib.load()
    return L.dg_conv_x3_planes_ok(1, n, h, wd, c, k, 2, 1) == 2 and L.dg_conv_x3_planes_ok(2, n, h, wd, c, k, 2, 1) == 1


def x3_transpose_table(convs):
    """Host-side table for x3_transpose_planes: convs = [(element offset inside a plane, K, J = 16 C), ...]."""
    import ctypes as C
    n = len(convs)
    return (n, (C.c_int64 * max(n, 1))(*[c[0] for c in convs]), (C.c_int * max(n, 1))(*[c[1] for c in convs]),
            (C.c_int * max(n, 1))(*[c[2] for c in convs]))


def x3_transpose_planes(src3, dst3, table):
    """dst3 <- for every conv weight of the table the [K][J] plane images of src3 transposed to [J][K] (same offsets)."""
    n, off, k, j = table
    assert src3.shape == dst3.shape and src3.stride(0) == dst3.stride(0)
    nbytes = 12.0 * sum(int(k[i]) * int(j[i]) for i in range(n))
    with _hbm("x3_split", nbytes):
        _lib.check(_lib.load().dg_x3_transpose_planes(_ptr(src3), _ptr(dst3), src3.stride(0), off, k, j, n, _stream()),
                   "dg_x3_transpose_planes")


def weight_planes(w, transposed=False):
    """(plane-0 address, plane distance in bytes, is the transposed copy) of a conv weight Parameter's triple.  transposed:
    prefer the [(r, s, c)][k] copy the forward form of the plane kernel reads (weights of a flat Adam group have one)."""
    e = getattr(w, "_dg_x3", None)
    if e is None:                                    # a parameter outside a flat Adam group
        e = (torch.empty((3, w.numel()), device=w.device, dtype=torch.bfloat16), 0, None)
        w._dg_x3, w._dg_x3_ver = e, None
    buf, off, buft = e
    if getattr(w, "_dg_x3_ver", None) != w._version:            # e.g. load_state_dict wrote the fp32 weights
        _lib.check(_lib.load().dg_f32_to_bf16x3(_ptr(w), buf.data_ptr() + 2 * off, w.numel(), buf.stride(0), _stream()),
                   "dg_f32_to_bf16x3")
        if buft is not None:
            x3_transpose_planes(buf, buft, x3_transpose_table([(off, w.shape[0], 16 * w.shape[1])]))
        w._dg_x3_ver = w._version
    if transposed and buft is not None:
        return buft.data_ptr() + 2 * off, buft.stride(0) * 2, 1
    return buf.data_ptr() + 2 * off, buf.stride(0) * 2, 0


def _x3_ok(op, n, h, wd, c, k, stride, pad):
    return _CUR.x3 and k > 1 and _plane_code_ok(_lib.load().dg_conv_x3_planes_ok(op, n, h, wd, c, k, stride, pad))


# ---- interior convolutions ------------------------------------------------------------------------------
def _out_hw(h, w, stride, pad):
    return (h + 2 * pad - 4) // stride + 1, (w + 2 * pad - 4) // stride + 1


def _mixed_operand(t, allow_shadow=True):
    """(pointer tensor, is_bf16) of a conv operand: a bf16 tensor as it is, an fp32 tensor's shadow if it has one."""
    if _is16(t):
        return t, 1
    t16 = shadow_get(t) if allow_shadow else None
    return (t16, 1) if t16 is not None else (t, 0)


def conv_fwd(x, w, stride, pad, want_stats=False):
    """nn.Conv2d(C,K,4,stride,pad,bias=False) forward. x NHWC-memory [N,C,H,W], w [K,C,4,4] KRSC.
    want_stats: also return the BatchNorm partial-statistics rows of y (None if the layer has no fused path)."""
    _check_dev(x, allow16=True, planes_ok=True)
    _check_dev(w)
    x = as_nhwc(x)
    w = _krsc(w)
    n, c, h, wd = x.shape
    k = w.shape[0]
    ho, wo = _out_hw(h, wd, stride, pad)
    L = _lib.load()
    prec = _CUR.cprec
    ws, wsb = _ws(L.dg_conv_workspace_bytes_p(0, n, h, wd, c, k, stride, pad, prec, 1), x.device)
    rows = L.dg_conv_bnstats_rows_p(0, n, h, wd, c, k, stride, pad, prec) if want_stats else 0
    if want_stats == "split" and L.dg_conv_plan_splits_p(0, n, h, wd, c, k, stride, pad, prec, 1) <= 1:
        rows = 0                      # statistics only where the split-K reduction kernel can emit them
    code = L.dg_conv_x3_planes_ok(0, n, h, wd, c, k, stride, pad) if (_CUR.x3 and k > 1) else 0
    if _plane_code_ok(code):
        wp, wdist, wt = weight_planes(w, transposed=True)
        if code == 4 or (code == 3 and not (wt and X3_FWW)):
            # the register-staged tiles read the PLAIN planes (3 with transposed planes and X3_FWW: the window forward kernel)
            wp, wdist, wt = (*weight_planes(w)[:2], 0) if X3_RSP else (None, 0, 0)
        if wp is not None:
            xp, xd, _ = planes_of(x)
            y = empty_nhwc(n, k, ho, wo, x.device)
            srows = L.dg_conv_x3_bnstats_rows(0, n, h, wd, c, k, stride, pad) if want_stats else 0
            stat = torch.empty((srows, 3 * k + 4), device=x.device, dtype=torch.float32) if srows > 0 else None
            with _prof("conv_fwd", 2.0 * n * ho * wo * k * c * 16):
                _lib.check(L.dg_conv_fwd_x3(xp, xd, wp, wdist, wt, _ptr(y), n, h, wd, c, k, stride, pad, _ptr(stat),
                                            stat.numel() if stat is not None else 0, _ptr(ws), wsb, _stream()), "dg_conv_fwd_x3")
            return (y, stat) if want_stats else y
    _need_fp32(x, "conv_fwd")
    mixed = False
    xa, x16, wa, w16, o16 = x, 0, w, 0, 0
    if k == 1:
        mixed = _is16(x)
        x16 = int(mixed)
    elif (rows == 0 or FUSE_STATS16) and _bf16_ok(0, n, h, wd, c, k, stride, pad):
        xa, x16 = _mixed_operand(x)
        wsh = weight_shadow(w)
        if wsh is not None:
            wa, w16 = wsh, 1
        o16 = int(_act_dtype(k) == torch.bfloat16)
        mixed = bool(x16 or w16 or o16)
    if _is16(x) and not mixed:
        raise _lib.DiscoganHipError(f"conv_fwd: no bf16 kernel for a bf16 input of shape {tuple(x.shape)} -> {k} channels")
    y = empty_nhwc(n, k, ho, wo, x.device, torch.bfloat16 if o16 else torch.float32)
    with _prof("conv_fwd" if k > 1 else "head1", 2.0 * n * ho * wo * k * c * 16):
        if mixed:
            mrows = L.dg_conv_mixed_bnstats_rows(0, n, h, wd, c, k, stride, pad, x16, w16) if (want_stats and FUSE_STATS16) else 0
            stat = torch.empty((mrows, 3 * k + 4), device=x.device, dtype=torch.float32) if mrows > 0 else None
            _lib.check(L.dg_conv_fwd_mixed(_ptr(xa), x16, _ptr(wa), w16, _ptr(y), o16, n, h, wd, c, k,
                                           stride, pad, _ptr(stat), stat.numel() if stat is not None else 0, _ptr(ws), wsb, _stream()),
                       "dg_conv_fwd_mixed")
        else:       # the fp32-tensor form: arithmetic = this context's, passed with the call
            stat = torch.empty((rows, 3 * k + 4), device=x.device, dtype=torch.float32) if rows > 0 else None
            _lib.check(L.dg_conv_fwd_g(1, _tab([x]), _tab([w]), _tab([y]), n, h, wd, c, k, stride, pad, prec, 1, _tab([stat]) if rows > 0 else None,
                                       stat.numel() if rows > 0 else 0, _tab([ws]), wsb, _stream()), "dg_conv_fwd_g")
    return (y, stat) if want_stats else y


def conv_dgrad(dy, w, x_hw, stride, pad, want_stats=False):
    """Input-gradient of that Conv2d == ConvTranspose2d forward. dy [N,K,Ho,Wo]; returns [N,C,H,W]
    (and, with want_stats, the BatchNorm partial-statistics rows of the result or None)."""
    _check_dev(dy, allow16=True, planes_ok=True)
    _check_dev(w)
    dy = as_nhwc(dy)
    w = _krsc(w)
    n, k = dy.shape[0], dy.shape[1]
    c = w.shape[1]
    h, wd = x_hw
    L = _lib.load()
    prec = _CUR.cprec
    ws, wsb = _ws(L.dg_conv_workspace_bytes_p(1, n, h, wd, c, k, stride, pad, prec, 1), dy.device)
    rows = L.dg_conv_bnstats_rows_p(1, n, h, wd, c, k, stride, pad, prec) if want_stats else 0
    if want_stats == "split" and L.dg_conv_plan_splits_p(1, n, h, wd, c, k, stride, pad, prec, 1) <= 1:
        rows = 0
    if _x3_ok(1, n, h, wd, c, k, stride, pad):
        dp, dd, dcm = planes_of(dy, allow_cm=L.dg_conv_x3_planes_ok(1, n, h, wd, c, k, stride, pad) == 2)
        wp, wdist, _ = weight_planes(w)
        dx = empty_nhwc(n, c, h, wd, dy.device)
        srows = L.dg_conv_x3_bnstats_rows(1, n, h, wd, c, k, stride, pad) if want_stats else 0
        stat = torch.empty((srows, 3 * c + 4), device=dy.device, dtype=torch.float32) if srows > 0 else None
        with _prof("conv_dgrad", 2.0 * n * dy.shape[2] * dy.shape[3] * k * c * 16):
            _lib.check(L.dg_conv_dgrad_x3(dp, dd, dcm, wp, wdist, _ptr(dx), n, h, wd, c, k, stride, pad, _ptr(stat),
                                          stat.numel() if stat is not None else 0, _ptr(ws), wsb, _stream()), "dg_conv_dgrad_x3")
        return (dx, stat) if want_stats else dx
    _need_fp32(dy, "conv_dgrad")
    mixed = False
    da, d16, wa, w16, o16 = dy, 0, w, 0, 0
    if k == 1:
        o16 = int(_act_dtype(c) == torch.bfloat16)
        mixed = bool(o16)
    elif (rows == 0 or FUSE_STATS16) and _bf16_ok(1, n, h, wd, c, k, stride, pad):
        if k % 8 == 0:
            da, d16 = _mixed_operand(dy)
        wsh = weight_shadow(w)
        if wsh is not None:
            wa, w16 = wsh, 1
        o16 = int(_act_dtype(c) == torch.bfloat16)
        mixed = bool(d16 or w16 or o16)
    if _is16(dy) and not d16:
        raise _lib.DiscoganHipError(f"conv_dgrad: no bf16 kernel for a bf16 gradient of shape {tuple(dy.shape)}")
    dx = empty_nhwc(n, c, h, wd, dy.device, torch.bfloat16 if o16 else torch.float32)
    with _prof("conv_dgrad" if k > 1 else "head1", 2.0 * n * dy.shape[2] * dy.shape[3] * k * c * 16):
        if mixed:
            mrows = L.dg_conv_mixed_bnstats_rows(1, n, h, wd, c, k, stride, pad, d16, w16) if (want_stats and FUSE_STATS16 and k > 1) else 0
            stat = torch.empty((mrows, 3 * c + 4), device=dy.device, dtype=torch.float32) if mrows > 0 else None
            _lib.check(L.dg_conv_dgrad_mixed(_ptr(da), d16, _ptr(wa), w16, _ptr(dx), o16, n, h, wd, c, k,
                                             stride, pad, _ptr(stat), stat.numel() if stat is not None else 0, _ptr(ws), wsb, _stream()),
                       "dg_conv_dgrad_mixed")
        else:
            stat = torch.empty((rows, 3 * c + 4), device=dy.device, dtype=torch.float32) if rows > 0 else None
            _lib.check(L.dg_conv_dgrad_g(1, _tab([dy]), _tab([w]), _tab([dx]), n, h, wd, c, k, stride, pad, prec, 1, _tab([stat]) if rows > 0 else None,
                                         stat.numel() if rows > 0 else 0, _tab([ws]), wsb, _stream()), "dg_conv_dgrad_g")
    return (dx, stat) if want_stats else dx


def conv_fwd_bias_act(x, w, bias, stride, pad, act=ACT_NONE, slope=0.2):
    """Inference: act(Conv2d(x; w) + bias) in one kernel (BatchNorm folded into w / bias by the caller)."""
    _check_dev(x, w, bias)
    x, w = as_nhwc(x), _krsc(w)
    n, c, h, wd = x.shape
    k = w.shape[0]
    ho, wo = _out_hw(h, wd, stride, pad)
    y = empty_nhwc(n, k, ho, wo, x.device)
    L = _lib.load()
    ws, wsb = _ws(L.dg_conv_workspace_bytes(0, n, h, wd, c, k, stride, pad), x.device)
    _lib.check(L.dg_conv_fwd_bias_act(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), n, h, wd, c, k, stride, pad, act, slope, _ptr(ws), wsb,
                                      _stream()), "dg_conv_fwd_bias_act")
    return y


def conv_dgrad_bias_act(x, w, bias, out_hw, stride, pad, act=ACT_NONE, slope=0.0):
    """Inference: act(ConvTranspose2d(x; w) + bias) in one kernel; w is the transposed conv's [Cin,Cout,4,4] (KRSC memory)."""
    _check_dev(x, w, bias)
    x, w = as_nhwc(x), _krsc(w)
    n, k = x.shape[0], x.shape[1]
    c = w.shape[1]
    h, wd = out_hw
    y = empty_nhwc(n, c, h, wd, x.device)
    L = _lib.load()
    ws, wsb = _ws(L.dg_conv_workspace_bytes(1, n, h, wd, c, k, stride, pad), x.device)
    _lib.check(L.dg_conv_dgrad_bias_act(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), n, h, wd, c, k, stride, pad, act, slope, _ptr(ws), wsb,
                                        _stream()), "dg_conv_dgrad_bias_act")
    return y


def conv_wgrad(dy, x, stride, pad, out=None, accumulate=False):
    """Weight-gradient of that Conv2d: returns logical [K,C,4,4] (KRSC memory)."""
    _check_dev(dy, x, allow16=True, planes_ok=True)
    dy = as_nhwc(dy)
    x = as_nhwc(x)
    n, c, h, wd = x.shape
    k = dy.shape[1]
    dw = out if out is not None else empty_krsc(k, c, x.device)
    L = _lib.load()
    prec = _CUR.cprec
    ws, wsb = _ws(L.dg_conv_workspace_bytes_p(2, n, h, wd, c, k, stride, pad, prec, 1), x.device)
    if _x3_ok(2, n, h, wd, c, k, stride, pad):
        dp, dd, dcm = planes_of(dy, allow_cm=True)
        xp, xd, _ = planes_of(x)
        with _prof("conv_wgrad", 2.0 * n * dy.shape[2] * dy.shape[3] * k * c * 16):
            _lib.check(L.dg_conv_wgrad_x3(dp, dd, dcm, xp, xd, _ptr(dw), n, h, wd, c, k, stride, pad, int(accumulate), _ptr(ws), wsb,
                                          _stream()), "dg_conv_wgrad_x3")
        return dw
    _need_fp32(dy, "conv_wgrad")
    _need_fp32(x, "conv_wgrad")
    da, d16, xa, x16 = dy, 0, x, 0
    if k == 1:
        x16 = int(_is16(x))
    elif _bf16_ok(2, n, h, wd, c, k, stride, pad) or _is16(x) or _is16(dy):
        xa, x16 = _mixed_operand(x)
        if k % 8 == 0:
            da, d16 = _mixed_operand(dy)
    if (_is16(dy) and not d16) or (_is16(x) and not x16):
        raise _lib.DiscoganHipError(f"conv_wgrad: no bf16 kernel for bf16 operands of shapes {tuple(dy.shape)}, {tuple(x.shape)}")
    with _prof("conv_wgrad" if k > 1 else "head1", 2.0 * n * dy.shape[2] * dy.shape[3] * k * c * 16):
        if d16 or x16:
            _lib.check(L.dg_conv_wgrad_mixed(_ptr(da), d16, _ptr(xa), x16, _ptr(dw), n, h, wd, c, k,
                                             stride, pad, int(accumulate), _ptr(ws), wsb, _stream()), "dg_conv_wgrad_mixed")
        else:
            _lib.check(L.dg_conv_wgrad_g(1, 1, _tab([dy]), _tab([x]), _tab([dw]), n, h, wd, c, k, stride, pad, prec, 1, int(accumulate),
                                         _tab([ws]), wsb, _stream()), "dg_conv_wgrad_g")
    return dw


# ---- 3-channel edge layers ------------------------------------------------------------------------------
def c3_fwd(x_nchw, w, act=ACT_NONE, slope=0.2, want_planes=False):
    """x contiguous NCHW [N,3,H,W], w contiguous [K,3,4,4] -> NHWC-memory [N,K,H/2,W/2] (act fused).
    want_planes (f32x3 plane path): the kernel also writes the plane triple of its output (the next layer's weight-gradient reads it)."""
    _check_dev(x_nchw, w)
    x = x_nchw.contiguous()
    w = w.contiguous()
    n, _, h, wd = x.shape
    k = w.shape[0]
    if want_planes and _CUR.x3 and k == 64 and x.numel() * 4 < (1 << 30) and n * (h // 2) * (wd // 2) < (1 << 30):
        y = empty_nhwc(n, k, h // 2, wd // 2, x.device)
        y3 = torch.empty((3, y.numel()), device=x.device, dtype=torch.bfloat16)
        with _prof("c3_fwd", 2.0 * n * (h // 2) * (wd // 2) * k * 48), _hbm("edge_c3_fwd", 4.0 * x.numel() + 10.0 * y.numel()):
            _lib.check(_lib.load().dg_conv4x4s2_c3_fwd_x3(_ptr(x), _ptr(w), _ptr(y), _ptr(y3), y3.stride(0), n, h, wd, k, act, slope, _stream()),
                       "dg_conv4x4s2_c3_fwd_x3")
        planes_put(y, y3)
        return y
    o16 = _CUR.act16 and k == 64 and x.numel() * 4 < (1 << 30)
    y = empty_nhwc(n, k, h // 2, wd // 2, x.device, torch.bfloat16 if o16 else torch.float32)
    with _prof("c3_fwd", 2.0 * n * (h // 2) * (wd // 2) * k * 48), \
            _hbm("edge_c3_fwd", 4.0 * x.numel() + y.numel() * y.element_size()):
        _lib.check(_lib.load().dg_conv4x4s2_c3_fwd_p(_ptr(x), _ptr(w), _ptr(y), int(o16), n, h, wd, k, act, slope, _CUR.cprec, _stream()),
                   "dg_conv4x4s2_c3_fwd_p")
    if _CUR.shadow and not o16 and k % 8 == 0:
        shadow_put(y, f32_to_bf16(y, empty_nhwc_bf16(n, k, h // 2, wd // 2, x.device)))
    return y


def c3_dgrad_act_ok(k):
    """Does the input-gradient kernel take the layer's activation backward in its load path (c3_dgrad(..., act_out=...))?"""
    return bool(_lib.load().dg_c3_dgrad_act_ok(int(k)))


def c3_dgrad(dy, w, act=ACT_NONE, act_out=None, in_act=ACT_NONE, slope=0.2):
    """dy NHWC-memory [N,K,Ho,Wo], w contiguous [K,3,4,4] -> contiguous NCHW [N,3,2Ho,2Wo] (act fused).
    With ``act_out`` (the saved output of the layer's fused LeakyReLU; needs c3_dgrad_act_ok(K)) dy is taken through the activation
    backward on the fly -- bitwise c3_dgrad(act_bwd(dy, act_out, in_act, slope), w, act) without the pass in front."""
    _check_dev(dy, act_out, allow16=True)
    _check_dev(w)
    dy = as_nhwc(dy)
    w = w.contiguous()
    n, k, ho, wo = dy.shape
    dx = torch.empty((n, 3, 2 * ho, 2 * wo), device=dy.device, dtype=torch.float32)
    L = _lib.load()
    ws, wsb = _ws(L.dg_c3_dgrad_workspace_bytes(k), dy.device)
    if act_out is not None and in_act != ACT_NONE:
        ao = as_nhwc(act_out)
        if ao.shape != dy.shape or ao.dtype != dy.dtype or ao.stride() != dy.stride():
            raise _lib.DiscoganHipError("c3_dgrad: act_out must have dy's shape, type and layout")
        with _hbm("edge_c3_dgrad", 2.0 * dy.numel() * dy.element_size() + 4.0 * dx.numel()):
            _lib.check(L.dg_conv4x4s2_c3_dgrad_act_p(_ptr(dy), int(_is16(dy)), _ptr(ao), in_act, float(slope), _ptr(w), _ptr(dx), n, 2 * ho, 2 * wo, k,
                                                     act, _CUR.cprec, _ptr(ws), wsb, _stream()), "dg_conv4x4s2_c3_dgrad_act_p")
        return dx
    with _hbm("edge_c3_dgrad", dy.numel() * dy.element_size() + 4.0 * dx.numel()):
        _lib.check(L.dg_conv4x4s2_c3_dgrad_p(_ptr(dy), int(_is16(dy)), _ptr(w), _ptr(dx), n, 2 * ho, 2 * wo, k, act, _CUR.cprec, _ptr(ws), wsb,
                                             _stream()), "dg_conv4x4s2_c3_dgrad_p")
    return dx


def c3_wgrad(dy, x_nchw, out=None, accumulate=False, act_out=None, act=ACT_NONE, slope=0.0):
    """dw [K,3,4,4] contiguous from dy NHWC-memory [N,K,Ho,Wo] and x NCHW [N,3,H,W].  With ``act_out`` (the saved
    output of the layer's fused LeakyReLU/ReLU) dy is taken through the activation backward on the fly."""
    _check_dev(dy, act_out, allow16=True)
    _check_dev(x_nchw)
    dy = as_nhwc(dy)
    x = x_nchw.contiguous()
    n, k, ho, wo = dy.shape
    h, wd = x.shape[2], x.shape[3]
    dw = out if out is not None else torch.empty((k, 3, 4, 4), device=dy.device, dtype=torch.float32)
    L = _lib.load()
    ws, wsb = _ws(L.dg_c3_wgrad_workspace_bytes(n, h, wd, k), dy.device)
    fuse = act_out is not None and act != ACT_NONE
    ao = None
    if fuse:
        ao = as_nhwc(act_out)
        assert ao.shape == dy.shape and ao.dtype == dy.dtype
    if _is16(dy):
        with _hbm("edge_c3_wgrad", 2.0 * ((2 if fuse else 1) * dy.numel()) + 4.0 * x.numel()):
            _lib.check(L.dg_conv4x4s2_c3_wgrad_p(_ptr(dy), _ptr(ao), 1, act if fuse else ACT_NONE, float(slope), _ptr(x), _ptr(dw),
                                                 n, h, wd, k, _CUR.cprec, int(accumulate), _ptr(ws), wsb, _stream()), "dg_conv4x4s2_c3_wgrad_p")
        return dw
    if fuse:
        with _hbm("edge_c3_wgrad", 4.0 * (2 * dy.numel() + x.numel())):
            _lib.check(L.dg_conv4x4s2_c3_wgrad_p(_ptr(dy), _ptr(ao), 0, act, float(slope), _ptr(x), _ptr(dw), n, h, wd, k, _CUR.cprec,
                                                 int(accumulate), _ptr(ws), wsb, _stream()), "dg_conv4x4s2_c3_wgrad_p")
        return dw
    with _hbm("edge_c3_wgrad", 4.0 * (dy.numel() + x.numel())):
        _lib.check(L.dg_conv4x4s2_c3_wgrad_p(_ptr(dy), None, 0, ACT_NONE, 0.0, _ptr(x), _ptr(dw), n, h, wd, k, _CUR.cprec, int(accumulate), _ptr(ws), wsb,
                                             _stream()), "dg_conv4x4s2_c3_wgrad_p")
    return dw


# ---- BatchNorm + activation -----------------------------------------------------------------------------
def bn_train_stats(y, running_mean, running_var, nbt, eps, momentum):
    _check_dev(y, allow16=True)
    y = as_nhwc(y)
    n, c, h, w = y.shape
    m = n * h * w
    saved = torch.empty((2, c), device=y.device, dtype=torch.float32)
    L = _lib.load()
    ws, wsb = _ws(L.dg_bn_workspace_bytes(m, c), y.device)
    with _hbm("bn_stats", float(y.element_size()) * m * c):
        _lib.check(L.dg_bn_train_stats_t(_ptr(y), int(_is16(y)), m, c, eps, momentum, _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                         _ptr(saved), _ptr(ws), wsb, _stream()), "dg_bn_train_stats")
    return saved


_PARTIALS_ONE_LAUNCH = __import__("os").environ.get("DG_BN_PARTIALS_ONE_LAUNCH", "0") == "1"     # same-box A/B of the two-level merge


def bn_stats_from_partials(stat, y, running_mean, running_var, nbt, eps, momentum):
    """Same result as bn_train_stats(y, ...) from the partial rows a conv kernel emitted for y."""
    _check_dev(stat)
    n, c, h, w = y.shape
    saved = torch.empty((2, c), device=y.device, dtype=torch.float32)
    L = _lib.load()
    ws, wsb = (None, 0) if _PARTIALS_ONE_LAUNCH else _ws(L.dg_bn_partials_workspace_bytes(stat.shape[0], c), y.device)
    _lib.check(L.dg_bn_stats_from_partials(_ptr(stat), stat.shape[0], n * h * w, c, eps, momentum,
                                           _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(saved),
                                           _ptr(ws), wsb, _stream()), "dg_bn_stats_from_partials")
    return saved


def bn_act_fwd(y, saved, gamma, beta, act, slope=0.2, planes_cm=False, planes_only=False):
    """planes_cm (f32x3 plane path): write z's plane triple in the quad-chunk layout (its reader is a window input-grad kernel);
    planes_only: every reader of z is a plane kernel -> the fp32 copy is not written (X3_PLANES_ONLY)."""
    _check_dev(y, allow16=True)
    y = as_nhwc(y)
    n, c, h, w = y.shape
    if _is16(y):                      # bf16 activation storage: bf16 in, bf16 out, nothing else is written
        z = empty_nhwc(n, c, h, w, y.device, torch.bfloat16)
        with _hbm("bn_apply", 4.0 * n * h * w * c):
            _lib.check(_lib.load().dg_bn_act_fwd_t(_ptr(y), _ptr(z), 1, n * h * w, c, _ptr(saved), _ptr(gamma), _ptr(beta),
                                                   act, slope, _stream()), "dg_bn_act_fwd_t")
        return z
    z = empty_nhwc(n, c, h, w, y.device)
    if _CUR.x3 and c % 8 == 0:             # f32x3 path: the next conv (forward and weight gradient) reads z as a plane triple
        z3 = torch.empty((3, z.numel()), device=y.device, dtype=torch.bfloat16)
        with _hbm("bn_apply", (10.0 if (planes_only and X3_PLANES_ONLY) else 14.0) * n * h * w * c):
            cm = int(bool(planes_cm) and c % 64 == 0 and (n * h * w) % 4 == 0)
            po = bool(planes_only) and X3_PLANES_ONLY
            _lib.check(_lib.load().dg_bn_act_fwd_x3(_ptr(y), None if po else _ptr(z), _ptr(z3), z3.stride(0), cm, n * h * w, c, _ptr(saved),
                                                    _ptr(gamma), _ptr(beta), act, slope, _stream()), "dg_bn_act_fwd_x3")
        planes_put(z, z3, cm)
        if po:
            z._dg_planes_only = True
            if POISON_PLANES_ONLY:
                z.fill_(float("nan"))
        return z
    if _CUR.shadow and c % 8 == 0:
        z16 = empty_nhwc_bf16(n, c, h, w, y.device)
        with _hbm("bn_apply", 10.0 * n * h * w * c):
            _lib.check(_lib.load().dg_bn_act_fwd_bf16(_ptr(y), _ptr(z), _ptr(z16), n * h * w, c, _ptr(saved), _ptr(gamma), _ptr(beta),
                                                      act, slope, _stream()), "dg_bn_act_fwd_bf16")
        shadow_put(z, z16)
        return z
    with _hbm("bn_apply", 8.0 * n * h * w * c):
        _lib.check(_lib.load().dg_bn_act_fwd(_ptr(y), _ptr(z), n * h * w, c, _ptr(saved), _ptr(gamma), _ptr(beta),
                                             act, slope, _stream()), "dg_bn_act_fwd")
    return z


def bn_act_bwd(dz, y, saved, gamma, beta, act, slope=0.2, need_param_grads=True, out_grads=None, planes_cm=False, planes_only=False):
    """out_grads=(dgamma_buf, dbeta_buf): accumulate the parameter gradients into those buffers in place.
    planes_cm (f32x3 plane path): write dy's plane triple chunk-major (its reader is a window input-grad kernel)."""
    _check_dev(dz, y, allow16=True)
    dz = as_nhwc(dz)
    y = as_nhwc(y)
    n, c, h, w = y.shape
    m = n * h * w
    acc = 0
    if out_grads is not None:
        dgamma, dbeta = out_grads
        acc = 1
    else:
        dgamma = torch.empty(c, device=y.device, dtype=torch.float32) if need_param_grads else None
        dbeta = torch.empty(c, device=y.device, dtype=torch.float32) if need_param_grads else None
    L = _lib.load()
    ws, wsb = _ws(L.dg_bn_workspace_bytes(m, c), y.device)
    if _is16(y):
        if not _is16(dz):
            raise _lib.DiscoganHipError("bn_act_bwd: bf16 y needs a bf16 dz")
        dy = empty_nhwc(n, c, h, w, y.device, torch.bfloat16)
        with _hbm("bn_backward", 10.0 * m * c):
            _lib.check(L.dg_bn_act_bwd_t(_ptr(dz), _ptr(y), _ptr(dy), 1, m, c, _ptr(saved), _ptr(gamma), _ptr(beta), act,
                                         slope, _ptr(dgamma), _ptr(dbeta), acc, _ptr(ws), wsb, _stream()), "dg_bn_act_bwd_t")
        return dy, dgamma, dbeta
    dy = empty_nhwc(n, c, h, w, y.device)
    if _CUR.x3 and c % 8 == 0:             # f32x3 path: dy goes to the layer's input-gradient and weight-gradient convs as a plane triple
        dy3 = torch.empty((3, dy.numel()), device=y.device, dtype=torch.bfloat16)
        po = bool(planes_only) and X3_PLANES_ONLY
        with _hbm("bn_backward", (22.0 if po else 26.0) * m * c):
            cm = int(bool(planes_cm) and c % 64 == 0 and m % 4 == 0)
            _lib.check(L.dg_bn_act_bwd_x3(_ptr(dz), _ptr(y), None if po else _ptr(dy), _ptr(dy3), dy3.stride(0), cm, m, c, _ptr(saved), _ptr(gamma),
                                          _ptr(beta), act, slope, _ptr(dgamma), _ptr(dbeta), acc, _ptr(ws), wsb, _stream()), "dg_bn_act_bwd_x3")
        planes_put(dy, dy3, cm)
        if po:
            dy._dg_planes_only = True
            if POISON_PLANES_ONLY:
                dy.fill_(float("nan"))
        return dy, dgamma, dbeta
    if _CUR.shadow and c % 8 == 0:
        dy16 = empty_nhwc_bf16(n, c, h, w, y.device)
        with _hbm("bn_backward", 22.0 * m * c):
            _lib.check(L.dg_bn_act_bwd_bf16(_ptr(dz), _ptr(y), _ptr(dy), _ptr(dy16), m, c, _ptr(saved), _ptr(gamma), _ptr(beta), act,
                                            slope, _ptr(dgamma), _ptr(dbeta), acc, _ptr(ws), wsb, _stream()), "dg_bn_act_bwd_bf16")
        shadow_put(dy, dy16)
        return dy, dgamma, dbeta
    with _hbm("bn_backward", 20.0 * m * c):
        _lib.check(L.dg_bn_act_bwd(_ptr(dz), _ptr(y), _ptr(dy), m, c, _ptr(saved), _ptr(gamma), _ptr(beta), act, slope,
                                   _ptr(dgamma), _ptr(dbeta), acc, _ptr(ws), wsb, _stream()), "dg_bn_act_bwd")
    return dy, dgamma, dbeta


def _dense_like(x):
    """empty tensor with x's shape AND memory layout (x must be dense)."""
    return torch.empty_like(x, memory_format=torch.preserve_format)


def _dense(x):
    if x.is_contiguous() or is_nhwc(x):
        return x
    return x.contiguous()


def act_fwd(x, act, slope=0.2):
    _check_dev(x)
    x = _dense(x)
    y = _dense_like(x)
    _lib.check(_lib.load().dg_act_fwd(_ptr(x), _ptr(y), x.numel(), act, slope, _stream()), "dg_act_fwd")
    return y


def act_bwd(dy, out, act, slope=0.2):
    """Elementwise; dy is brought to out's memory layout first."""
    _check_dev(dy, out, allow16=True)
    out = _dense(out)
    if dy.stride() != out.stride():
        dyl = _dense_like(out)
        dyl.copy_(dy)
        dy = dyl
    dx = _dense_like(out)
    if _is16(out) or _is16(dy):
        if dy.dtype != out.dtype:
            raise _lib.DiscoganHipError("act_bwd: dy and out must have the same dtype")
        _lib.check(_lib.load().dg_act_bwd_t(_ptr(dy), _ptr(out), _ptr(dx), 1, out.numel(), act, slope, _stream()), "dg_act_bwd_t")
        return dx
    _lib.check(_lib.load().dg_act_bwd(_ptr(dy), _ptr(out), _ptr(dx), out.numel(), act, slope, _stream()), "dg_act_bwd")
    return dx


# ---- losses ---------------------------------------------------------------------------------------------
def _loss_ws(device):
    L = _lib.load()
    return _ws(L.dg_loss_workspace_bytes(), device)


def same_layout_pair(a, b):
    """Bring b to a's dense memory layout (elementwise kernels index memory linearly)."""
    a = _dense(a)
    if b.stride() != a.stride():
        bl = _dense_like(a)
        bl.copy_(b)
        b = bl
    return a, b


def mse_fwd(x, t, out=None):
    _check_dev(x, t)
    x, t = same_layout_pair(x, t)
    loss = out if out is not None else torch.empty((), device=x.device, dtype=torch.float32)
    ws, wsb = _loss_ws(x.device)
    _lib.check(_lib.load().dg_mse_fwd(_ptr(x), _ptr(t), x.numel(), _ptr(loss), _ptr(ws), wsb, _stream()), "dg_mse_fwd")
    return loss, x, t


def mse_bwd(x, t, gout):
    _check_dev(x, t, gout)
    dx = _dense_like(x)
    _lib.check(_lib.load().dg_mse_bwd(_ptr(x), _ptr(t), x.numel(), _ptr(gout), _ptr(dx), _stream()), "dg_mse_bwd")
    return dx


def bce_fwd(p, label, out=None):
    _check_dev(p)
    p = p.contiguous()
    loss = out if out is not None else torch.empty((), device=p.device, dtype=torch.float32)
    _lib.check(_lib.load().dg_bce_fwd(_ptr(p), p.numel(), float(label), _ptr(loss), None, 0, _stream()), "dg_bce_fwd")
    return loss, p


def bce_bwd(p, label, gout):
    dp = torch.empty_like(p)
    _lib.check(_lib.load().dg_bce_bwd(_ptr(p), p.numel(), float(label), _ptr(gout), _ptr(dp), _stream()), "dg_bce_bwd")
    return dp


def bce_target_fwd(p, target, out=None):
    _check_dev(p, target)
    p, target = p.contiguous(), target.contiguous()
    if p.numel() != target.numel():
        raise ValueError(f"Using a target size ({tuple(target.shape)}) that is different to the input size ({tuple(p.shape)}) is deprecated. Please ensure they have the same size.")
    loss = out if out is not None else torch.empty((), device=p.device, dtype=torch.float32)
    _lib.check(_lib.load().dg_bce_target_fwd(_ptr(p), _ptr(target), p.numel(), _ptr(loss), _stream()), "dg_bce_target_fwd")
    return loss, p, target


def bce_target_bwd(p, target, gout):
    dp = torch.empty_like(p)
    _lib.check(_lib.load().dg_bce_target_bwd(_ptr(p), _ptr(target), p.numel(), _ptr(gout), _ptr(dp), _stream()), "dg_bce_target_bwd")
    return dp


def hinge_fwd(x, y, margin=1.0):
    _check_dev(x, y)
    x, y = same_layout_pair(x, y)
    loss = torch.empty((), device=x.device, dtype=torch.float32)
    ws, wsb = _loss_ws(x.device)
    _lib.check(_lib.load().dg_hinge_fwd(_ptr(x), _ptr(y), x.numel(), float(margin), _ptr(loss), _ptr(ws), wsb, _stream()), "dg_hinge_fwd")
    return loss, x, y


def hinge_bwd(x, y, margin, gout):
    dx = _dense_like(x)
    _lib.check(_lib.load().dg_hinge_bwd(_ptr(x), _ptr(y), x.numel(), float(margin), _ptr(gout), _ptr(dx), _stream()), "dg_hinge_bwd")
    return dx


def fm_fwd(real, fake, out=None):
    """One layer of get_fm_loss; real/fake logical [N,C,H,W] with identical dense layouts."""
    _check_dev(real, fake, allow16=True)
    real, fake = same_layout_pair(real, fake)
    n = real.shape[0]
    j = real.numel() // n
    diff = torch.empty(j, device=real.device, dtype=torch.float32)
    loss = out if out is not None else torch.empty((), device=real.device, dtype=torch.float32)
    ws, wsb = _ws(_lib.load().dg_fm_workspace_bytes(n, j), real.device)
    if real.dtype != fake.dtype:
        raise _lib.DiscoganHipError("fm_fwd: real and fake features must have the same dtype")
    _lib.check(_lib.load().dg_fm_fwd_t(_ptr(real), _ptr(fake), int(_is16(real)), n, j, _ptr(diff), _ptr(loss), _ptr(ws), wsb, _stream()),
               "dg_fm_fwd")
    return loss, diff, real, fake


def fm_bwd(diff, like_real, like_fake, gout, need_real, need_fake):
    n = like_fake.shape[0]
    j = diff.numel()
    dreal = _dense_like(like_real) if need_real else None
    dfake = _dense_like(like_fake) if need_fake else None
    _lib.check(_lib.load().dg_fm_bwd_t(_ptr(diff), n, j, _ptr(gout), _ptr(dreal), _ptr(dfake), int(_is16(like_fake)), _stream()), "dg_fm_bwd")
    return dreal, dfake


def loss_mix_fwd(lossvec, nfm, rate, arch):
    out = torch.empty(8, device=lossvec.device, dtype=torch.float32)
    _lib.check(_lib.load().dg_loss_mix_fwd(_ptr(lossvec), _ptr(out), nfm, float(rate), arch, _stream()), "dg_loss_mix_fwd")
    return out


def loss_mix_bwd(gout, nslots, nfm, rate, arch, which):
    gv = torch.empty(nslots, device=gout.device, dtype=torch.float32)
    _lib.check(_lib.load().dg_loss_mix_bwd(_ptr(gout), _ptr(gv), nfm, float(rate), arch, which, _stream()), "dg_loss_mix_bwd")
    return gv


# ---- Adam -----------------------------------------------------------------------------------------------
def adam_advance(state, lr, beta1, beta2):
    _lib.check(_lib.load().dg_adam_advance(_ptr(state), lr, beta1, beta2, _stream()), "dg_adam_advance")


def adam_step_flat(p, g, m, v, state, beta1, beta2, eps, weight_decay, grad_scale=1.0, p16=None, p3=None):
    if p3 is not None:      # (plane-0 address of this range, plane distance in elements): the f32x3 plane triples
        with _hbm("adam", 34.0 * p.numel()):
            _lib.check(_lib.load().dg_adam_step_flat_x3(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(state), beta1, beta2,
                                                        eps, weight_decay, grad_scale, p3[0], p3[1], _stream()), "dg_adam_step_flat_x3")
        return
    if p16 is not None:
        with _hbm("adam", 30.0 * p.numel()):
            _lib.check(_lib.load().dg_adam_step_flat_bf16(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(state), beta1, beta2,
                                                          eps, weight_decay, grad_scale, _ptr(p16), _stream()), "dg_adam_step_flat_bf16")
        return
    with _hbm("adam", 28.0 * p.numel()):
        _lib.check(_lib.load().dg_adam_step_flat(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(state), beta1,
                                                 beta2,
                                                 eps, weight_decay, grad_scale, _stream()), "dg_adam_step_flat")


# ---- grouped launches (round 4) -------------------------------------------------------------------------------------------
# The reference issues the passes of an iteration in independent pairs of identical shape -- G_B(A) | G_A(B), G_A(AB) | G_B(BA),
# D_A(A) | D_B(B), D_A(BA) | D_B(AB) (image_translation.py:342-361) -- and each discriminator sees real and fake images with the same
# weights (:353-354,360-361).  The *_g wrappers take one tensor per problem (lists) and issue ONE launch per kernel for all of them
# (dg_*_g: a block index picks the problem); every problem's result is bitwise what the one-problem wrapper computes.  fp32 tensors on
# the exact-fp32 and the register-staged f32x3 arithmetic only: no shadows, no plane triples, no bf16 storage, no fused statistics.
def group_ok():
    """May the current context run grouped launches?"""
    return not (_CUR.shadow or _CUR.act16 or _CUR.x3)


def _plan_groups(g):
    """Split-K plan of a grouped conv launch: sized for the whole launch (the group fills the chip: fewer K-slabs per problem), or --
    Context.group_plan == "single" -- for every problem as if launched alone (bitwise the ungrouped numbers; tests)."""
    return 1 if getattr(_CUR, "group_plan", "launch") == "single" else g


def _same_shape(ts, who):
    for t in ts[1:]:
        if t.shape != ts[0].shape:
            raise _lib.DiscoganHipError(f"{who}: the problems of a grouped launch must have identical shapes")


def _ws_g(nbytes, g, device):
    """One workspace per problem (slices of ONE allocation): (list of tensors or Nones, bytes each)."""
    if nbytes == 0:
        return [None] * g, 0
    per = (nbytes + 255) // 256 * 64                    # floats per problem, 256-byte aligned
    t = torch.empty(g * per, device=device, dtype=torch.float32)
    return [t[i * per:(i + 1) * per] for i in range(g)], per * 4


def _share_of(params):
    """Consecutive problems that name the SAME parameter (a discriminator's real and fake pass) accumulate into one gradient tensor:
    returns the run length when every run has the same length, else raises."""
    g = len(params)
    run = 1
    while run < g and params[run] is params[0]:
        run += 1
    if g % run != 0 or any(params[i] is not params[i // run * run] for i in range(g)) or \
            any(params[i] is params[i - run] for i in range(run, g, run)):
        raise _lib.DiscoganHipError("grouped launch: problems sharing a parameter must form runs of equal length")
    return run


def conv_fwd_g(xs, ws_, stride, pad):
    """Conv2d forward of ``len(xs)`` problems in one launch (plus one split-K reduction launch)."""
    g = len(xs)
    _check_dev(*xs, *ws_)
    xs = [as_nhwc(x) for x in xs]
    ws_ = [_krsc(w) for w in ws_]
    _same_shape(xs, "conv_fwd_g")
    _same_shape(ws_, "conv_fwd_g")
    n, c, h, wd = xs[0].shape
    k = ws_[0].shape[0]
    ho, wo = _out_hw(h, wd, stride, pad)
    L = _lib.load()
    prec = _CUR.cprec
    pg = _plan_groups(g)
    wsl, wsb = _ws_g(L.dg_conv_workspace_bytes_p(0, n, h, wd, c, k, stride, pad, prec, pg), g, xs[0].device)
    ys = [empty_nhwc(n, k, ho, wo, xs[0].device) for _ in range(g)]
    with _prof("conv_fwd" if k > 1 else "head1", 2.0 * g * n * ho * wo * k * c * 16):
        _lib.check(L.dg_conv_fwd_g(g, _tab(xs), _tab(ws_), _tab(ys), n, h, wd, c, k, stride, pad, prec, pg, None, 0, _tab(wsl), wsb, _stream()),
                   "dg_conv_fwd_g")
    return ys


def conv_dgrad_g(dys, ws_, x_hw, stride, pad):
    g = len(dys)
    _check_dev(*dys, *ws_)
    dys = [as_nhwc(d) for d in dys]
    ws_ = [_krsc(w) for w in ws_]
    _same_shape(dys, "conv_dgrad_g")
    _same_shape(ws_, "conv_dgrad_g")
    n, k = dys[0].shape[0], dys[0].shape[1]
    c = ws_[0].shape[1]
    h, wd = x_hw
    L = _lib.load()
    prec = _CUR.cprec
    pg = _plan_groups(g)
    wsl, wsb = _ws_g(L.dg_conv_workspace_bytes_p(1, n, h, wd, c, k, stride, pad, prec, pg), g, dys[0].device)
    dxs = [empty_nhwc(n, c, h, wd, dys[0].device) for _ in range(g)]
    with _prof("conv_dgrad" if k > 1 else "head1", 2.0 * g * n * dys[0].shape[2] * dys[0].shape[3] * k * c * 16):
        _lib.check(L.dg_conv_dgrad_g(g, _tab(dys), _tab(ws_), _tab(dxs), n, h, wd, c, k, stride, pad, prec, pg, None, 0, _tab(wsl), wsb, _stream()),
                   "dg_conv_dgrad_g")
    return dxs


def conv_wgrad_g(dys, xs, stride, pad, outs, accumulate, share=1):
    """Weight gradients of ``len(dys)`` problems into ``outs`` (accumulating views of the flat gradient buffer).  share > 1: every
    ``share`` consecutive problems name the same ``outs`` tensor and are summed into it in problem order."""
    g = len(dys)
    _check_dev(*dys, *xs)
    dys = [as_nhwc(d) for d in dys]
    xs = [as_nhwc(x) for x in xs]
    _same_shape(dys, "conv_wgrad_g")
    _same_shape(xs, "conv_wgrad_g")
    n, c, h, wd = xs[0].shape
    k = dys[0].shape[1]
    L = _lib.load()
    prec = _CUR.cprec
    pg = _plan_groups(g)
    wsl, wsb = _ws_g(L.dg_conv_workspace_bytes_p(2, n, h, wd, c, k, stride, pad, prec, pg), g, xs[0].device)
    with _prof("conv_wgrad" if k > 1 else "head1", 2.0 * g * n * dys[0].shape[2] * dys[0].shape[3] * k * c * 16):
        _lib.check(L.dg_conv_wgrad_g(g, int(share), _tab(dys), _tab(xs), _tab(outs), n, h, wd, c, k, stride, pad, prec, pg, int(accumulate),
                                     _tab(wsl), wsb, _stream()), "dg_conv_wgrad_g")


def c3_fwd_g(xs, ws_, act=ACT_NONE, slope=0.2):
    g = len(xs)
    _check_dev(*xs, *ws_)
    xs = [x.contiguous() for x in xs]
    ws_ = [w.contiguous() for w in ws_]
    _same_shape(xs, "c3_fwd_g")
    n, _, h, wd = xs[0].shape
    k = ws_[0].shape[0]
    ys = [empty_nhwc(n, k, h // 2, wd // 2, xs[0].device) for _ in range(g)]
    with _prof("c3_fwd", 2.0 * g * n * (h // 2) * (wd // 2) * k * 48), _hbm("edge_c3_fwd", g * (4.0 * xs[0].numel() + 4.0 * ys[0].numel())):
        _lib.check(_lib.load().dg_conv4x4s2_c3_fwd_g(g, _tab(xs), _tab(ws_), _tab(ys), n, h, wd, k, act, slope, _CUR.cprec, _stream()),
                   "dg_conv4x4s2_c3_fwd_g")
    return ys


def c3_dgrad_g(dys, ws_, act=ACT_NONE):
    g = len(dys)
    _check_dev(*dys, *ws_)
    dys = [as_nhwc(d) for d in dys]
    ws_ = [w.contiguous() for w in ws_]
    _same_shape(dys, "c3_dgrad_g")
    n, k, ho, wo = dys[0].shape
    dxs = [torch.empty((n, 3, 2 * ho, 2 * wo), device=dys[0].device, dtype=torch.float32) for _ in range(g)]
    with _hbm("edge_c3_dgrad", g * (4.0 * dys[0].numel() + 4.0 * dxs[0].numel())):
        _lib.check(_lib.load().dg_conv4x4s2_c3_dgrad_g(g, _tab(dys), _tab(ws_), _tab(dxs), n, 2 * ho, 2 * wo, k, act, _CUR.cprec, _stream()),
                   "dg_conv4x4s2_c3_dgrad_g")
    return dxs


def c3_wgrad_g(dys, xs, outs, accumulate, act_outs=None, act=ACT_NONE, slope=0.0, share=1):
    g = len(dys)
    _check_dev(*dys, *xs)
    dys = [as_nhwc(d) for d in dys]
    xs = [x.contiguous() for x in xs]
    _same_shape(dys, "c3_wgrad_g")
    n, k, ho, wo = dys[0].shape
    h, wd = xs[0].shape[2], xs[0].shape[3]
    fuse = act_outs is not None and act != ACT_NONE
    aos = [as_nhwc(a) for a in act_outs] if fuse else None
    L = _lib.load()
    wsl, wsb = _ws_g(L.dg_c3_wgrad_workspace_bytes(n, h, wd, k), g, dys[0].device)
    with _hbm("edge_c3_wgrad", g * 4.0 * ((2 if fuse else 1) * dys[0].numel() + xs[0].numel())):
        _lib.check(L.dg_conv4x4s2_c3_wgrad_g(g, int(share), _tab(dys), _tab(aos), act if fuse else ACT_NONE, float(slope), _tab(xs), _tab(outs),
                                             n, h, wd, k, _CUR.cprec, int(accumulate), _tab(wsl), wsb, _stream()), "dg_conv4x4s2_c3_wgrad_g")


def bn_train_stats_g(ys, running_means, running_vars, nbts, eps, momentum, share=1):
    g = len(ys)
    _check_dev(*ys)
    ys = [as_nhwc(y) for y in ys]
    _same_shape(ys, "bn_train_stats_g")
    n, c, h, w = ys[0].shape
    m = n * h * w
    saved = [torch.empty((2, c), device=ys[0].device, dtype=torch.float32) for _ in range(g)]
    L = _lib.load()
    wsl, wsb = _ws_g(L.dg_bn_workspace_bytes(m, c), g, ys[0].device)
    with _hbm("bn_stats", 4.0 * g * m * c):
        _lib.check(L.dg_bn_train_stats_g(g, int(share), _tab(ys), m, c, eps, momentum, _tab(running_means), _tab(running_vars), _tab(nbts),
                                         _tab(saved), _tab(wsl), wsb, _stream()), "dg_bn_train_stats_g")
    return saved


def bn_act_fwd_g(ys, saveds, gammas, betas, act, slope=0.2):
    g = len(ys)
    _check_dev(*ys)
    ys = [as_nhwc(y) for y in ys]
    n, c, h, w = ys[0].shape
    zs = [empty_nhwc(n, c, h, w, ys[0].device) for _ in range(g)]
    with _hbm("bn_apply", 8.0 * g * n * h * w * c):
        _lib.check(_lib.load().dg_bn_act_fwd_g(g, _tab(ys), _tab(zs), n * h * w, c, _tab(saveds), _tab(gammas), _tab(betas), act, slope, _stream()),
                   "dg_bn_act_fwd_g")
    return zs


def bn_act_bwd_g(dzs, ys, saveds, gammas, betas, act, slope, dgammas, dbetas, accumulate, share=1):
    """dgammas / dbetas: accumulating views of the flat gradient buffer, or None (parameters frozen)."""
    g = len(dzs)
    _check_dev(*dzs, *ys)
    dzs = [as_nhwc(d) for d in dzs]
    ys = [as_nhwc(y) for y in ys]
    _same_shape(dzs, "bn_act_bwd_g")
    n, c, h, w = ys[0].shape
    m = n * h * w
    dys = [empty_nhwc(n, c, h, w, ys[0].device) for _ in range(g)]
    L = _lib.load()
    wsl, wsb = _ws_g(L.dg_bn_workspace_bytes(m, c), g, ys[0].device)
    with _hbm("bn_backward", 20.0 * g * m * c):
        _lib.check(L.dg_bn_act_bwd_g(g, int(share), _tab(dzs), _tab(ys), _tab(dys), m, c, _tab(saveds), _tab(gammas), _tab(betas), act, slope,
                                     _tab(dgammas), _tab(dbetas), int(accumulate), _tab(wsl), wsb, _stream()), "dg_bn_act_bwd_g")
    return dys


def act_fwd_g(xs, act, slope=0.2):
    _check_dev(*xs)
    xs = [_dense(x) for x in xs]
    ys = [_dense_like(x) for x in xs]
    _lib.check(_lib.load().dg_act_fwd_g(len(xs), _tab(xs), _tab(ys), xs[0].numel(), act, slope, _stream()), "dg_act_fwd_g")
    return ys


def act_bwd_g(dys, outs, act, slope=0.2):
    _check_dev(*dys, *outs)
    outs = [_dense(o) for o in outs]
    dys = list(dys)
    for i, (d, o) in enumerate(zip(dys, outs)):
        if d.stride() != o.stride():
            dl = _dense_like(o)
            dl.copy_(d)
            dys[i] = dl
    dxs = [_dense_like(o) for o in outs]
    _lib.check(_lib.load().dg_act_bwd_g(len(outs), _tab(dys), _tab(outs), _tab(dxs), outs[0].numel(), act, slope, _stream()), "dg_act_bwd_g")
    return dxs


def mse_fwd_g(xs, ts, outs):
    g = len(xs)
    _check_dev(*xs, *ts)
    pairs = [same_layout_pair(x, t) for x, t in zip(xs, ts)]
    xs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    L = _lib.load()
    wsl, wsb = _ws_g(L.dg_loss_workspace_bytes(), g, xs[0].device)
    _lib.check(L.dg_mse_fwd_g(g, _tab(xs), _tab(ts), xs[0].numel(), _tab(outs), _tab(wsl), wsb, _stream()), "dg_mse_fwd_g")
    return xs, ts


def mse_bwd_g(xs, ts, gouts):
    dxs = [_dense_like(x) for x in xs]
    _lib.check(_lib.load().dg_mse_bwd_g(len(xs), _tab(xs), _tab(ts), xs[0].numel(), _tab(gouts), _tab(dxs), _stream()), "dg_mse_bwd_g")
    return dxs


def _labels(labels):
    import ctypes as C
    return (C.c_float * len(labels))(*[float(l) for l in labels])


def bce_fwd_g(ps, labels, outs):
    _check_dev(*ps)
    ps = [p.contiguous() for p in ps]
    _lib.check(_lib.load().dg_bce_fwd_g(len(ps), _tab(ps), ps[0].numel(), _labels(labels), _tab(outs), _stream()), "dg_bce_fwd_g")
    return ps


def bce_bwd_g(ps, labels, gouts):
    dps = [torch.empty_like(p) for p in ps]
    _lib.check(_lib.load().dg_bce_bwd_g(len(ps), _tab(ps), ps[0].numel(), _labels(labels), _tab(gouts), _tab(dps), _stream()), "dg_bce_bwd_g")
    return dps


def fm_fwd_g(reals, fakes, outs):
    g = len(reals)
    _check_dev(*reals, *fakes)
    pairs = [same_layout_pair(r, f) for r, f in zip(reals, fakes)]
    reals, fakes = [p[0] for p in pairs], [p[1] for p in pairs]
    n = reals[0].shape[0]
    j = reals[0].numel() // n
    diffs = [torch.empty(j, device=reals[0].device, dtype=torch.float32) for _ in range(g)]
    L = _lib.load()
    wsl, wsb = _ws_g(L.dg_fm_workspace_bytes(n, j), g, reals[0].device)
    _lib.check(L.dg_fm_fwd_g(g, _tab(reals), _tab(fakes), n, j, _tab(diffs), _tab(outs), _tab(wsl), wsb, _stream()), "dg_fm_fwd_g")
    return diffs, reals, fakes


def fm_bwd_g(diffs, like_reals, like_fakes, gouts, need_real, need_fake):
    n = like_fakes[0].shape[0]
    j = diffs[0].numel()
    dreals = [_dense_like(t) for t in like_reals] if need_real else None
    dfakes = [_dense_like(t) for t in like_fakes] if need_fake else None
    _lib.check(_lib.load().dg_fm_bwd_g(len(diffs), _tab(diffs), n, j, _tab(gouts), _tab(dreals), _tab(dfakes), _stream()), "dg_fm_bwd_g")
    return dreals, dfakes


# ---- module attributes ops.SHADOW / ops.ACT16 / ops.X3: views of the CURRENT context (kept for op-level tests and tools, which set
# them around single calls; a trainer carries its own Context and never touches them) -----------------------------------------------
import sys as _sys
import types as _types


class _OpsModule(_types.ModuleType):
    @property
    def SHADOW(self):
        return _CUR.shadow

    @SHADOW.setter
    def SHADOW(self, v):
        _CUR.shadow = bool(v)

    @property
    def ACT16(self):
        return _CUR.act16

    @ACT16.setter
    def ACT16(self, v):
        _CUR.act16 = bool(v)

    @property
    def _PLANE_TAB(self):           # (the current context's tables, under their pre-round-4 names: op tests look into them)
        return _CUR.plane_tab

    @property
    def _SHADOW_TAB(self):
        return _CUR.shadow_tab

    @property
    def X3(self):
        return _CUR.x3

    @X3.setter
    def X3(self, v):
        _CUR.x3 = bool(v)


_sys.modules[__name__].__class__ = _OpsModule
