// Implicit-GEMM convolution family for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
//
// Replaces (reference file:line): nn.Conv2d(k4,s2,p1) forward / input-grad / weight-grad
// (model.py:11-31,83-103 via autograd), nn.ConvTranspose2d(k4,s2,p1) (model.py:118-140; forward ==
// conv dgrad with the same weight tensor), and the k4 s1 p0 "head" layers (model.py:35,107,114).
//
// One kernel template, four addressing modes.  GEMM view  C[M][Ng] = sum_k A[M][k] * B[k][Ng]:
//   FWD          M = N*Ho*Wo pixels     Ng = K out-ch      k = (r,s,c)      A = im2col(x)  B = w[K][(r,s,c)]
//   DGRAD_S2     M = N*Ho*Wo per output parity (4 classes, 2x2 taps each)
//                                       Ng = C in-ch       k = (tap,kout)   A = dy gather   B = w[k][r][s][:]
//   DGRAD_PLAIN  M = N                  Ng = 16*C          k = kout         A = dy[N][K]    B = w[K][16C]
//   WGRAD        M = K out-ch           Ng = 16*C (r,s,c)  k = pixel        A = dy^T        B = im2col(x)
//   FWD_C3       M = N*Ho*Wo pixels     Ng = K out-ch      k = (c,r,s)=48   A = im2col(x NCHW, 3 ch)  B = w[K][48]
//                (3-channel image side: conv1 forward + fused LeakyReLU, last-convT input-grad)
// Activations are NHWC, weights KRSC, so every operand tile is a set of rows that are CONTIGUOUS in
// HBM: 16-byte global loads -> 16-byte ds_write_b128, no transposes anywhere.
//
// Tiling: 256 threads = 4 waves (WM x WN), each wave owns a 64x64 output tile = 2x2 MFMA 32x32
// accumulators (64 VGPRs).  K-tile KT (32 or 16), LDS double buffered, two register sets, ONE barrier per
// K-tile placed 8 MFMAs before the end of the tile (see the pipeline comment in the kernel): tile t+2 is in
// flight from HBM/L2 while tile t+1 waits in registers and tile t is multiplied out of LDS.
// Addressing (BUF): buffer descriptors + 32-bit offsets, masked vectors read out of range (zeros); the 64-bit
// pointer form (!BUF, masked vectors read a zero block) remains for tensors >= 2 GiB.
// PREC 1: operands rounded to bf16 when they are STAGED into LDS (bf16 tiles, K-tile 64), v_mfma_f32_32x32x16_bf16,
// fp32 accumulate; k-major operand images are read with the hardware transpose ds_read_b64_tr_b16.
// Epilogue: accumulators transposed through LDS, float4 stores (16 lanes per 256-B row segment).
// LDS operand images:
//   "k-contiguous" [row][KT+4]  : lane reads one b128 = its A/B values for FOUR consecutive MFMAs
//                                 (row stride 36 or 20 floats -> conflict-free b128 reads)
//   "k-major"      [kk][rows+4] : lane reads b32 per MFMA, 32 consecutive lanes -> conflict-free.
// The k index inside an 8-wide group is permuted identically for A and B (lane-half h, step j ->
// kk = 4h + j), which a GEMM is invariant to.
//
// Split-K: when the tile grid has fewer than one workgroup per CU (and for every weight gradient), the K loop
// is split over blockIdx; partial fp32 slabs go to the caller's workspace and a second kernel sums them in a
// fixed order (bitwise reproducible; no float atomics).
#include "dg_common.h"
#include <type_traits>

#include "igemm_args.h"

#define NEG_BIG (-(1 << 28))

// Masked (padding / out-of-range) operand vectors are fetched from this zero block instead of being
// zeroed after the load: a select on the loaded value would force an s_waitcnt right behind every
// global load and serialise HBM latency into the MFMA stream.
__device__ float dg_zero16[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

// BUF: operands are addressed through buffer descriptors with 32-bit byte offsets; a masked (padding / ragged)
// vector gets an out-of-range offset and the hardware returns zeros.  Per tile load this is 1-3 VALU
// instructions (add the wave-uniform tap offset, or-in the precomputed per-row validity bit) instead of the
// ~12 of the 64-bit pointer path (mad_i64, three 64-bit adds, compares, pointer select), which matters because
// everything between two MFMAs has to fit the ~48 issue cycles a 64-cycle MFMA leaves (stamps: the K loop ran
// at 87 % of the sustained matrix rate with 12-instruction clumps per load).  !BUF = any size, pointer path.
// PREC 1: the GEMM operands are rounded to bf16 (RNE) on their way from LDS to the matrix cores and multiplied on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation; tensors, LDS tiles, BatchNorm, master weights and Adam stay
// fp32 (BASELINE configs[4]: "bf16 MFMA + fp32 BatchNorm accum").  Products of two bf16 values are exact in
// fp32, so the result equals an fp32 convolution of the rounded operands up to summation order.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// A16 / B16 (PREC 1 only): that operand already is bf16 in HBM (a shadow copy its producer wrote: the Adam kernel for
// weights, bn_act_fwd / bn_bwd_apply / c3_fwd for activations and gradients) -- half the operand bytes, 8 elements per
// 16-byte load, no conversion on the way into LDS.  Numerically identical to rounding the fp32 tensor here (same RNE).
// option "dbg_zero" bit 3 (value 8): the f32x3 forward walks 16-channel chunks with the taps inside (the order before round 3) -- A/B switch
__device__ __forceinline__ bool dg_get_subwalk(const IgemmArgs& p) { return (p.dbg_zero & 8) == 0; }
template <int MODE, int WM, int WN, int KT, bool BUF, int PREC, bool A16 = false, bool B16 = false>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs p) {
    static_assert(!(A16 || B16) || (BUF && (PREC == 1 || PREC == 2)), "bf16 sources exist for the bf16 tile kernels only");
    // PLN (PREC 2 with A16 and B16): BOTH operands arrive as bf16 PLANE TRIPLES (a_plane / b_plane bytes apart), written once by their
    // producers -- the loader fetches three 16-byte vectors of 8 bf16 per item and stores them to the LDS planes as they are: no
    // fp32 -> 3 x bf16 split in this kernel (5.5 VALU instructions per operand element, a quarter of the register-staged f32x3
    // kernel's instruction stream on the 64 -> 128 channel forward, whose matrix pipes were 63 % busy).  Forward form only.
    constexpr bool PLN = PREC == 2 && A16 && B16;
    static_assert(!(PREC == 2 && (A16 != B16)), "f32x3: both operands split in the kernel, or both as plane triples");
    static_assert(!PLN || MODE == MODE_FWD, "the plane-reading register-staged tiles serve the forward form");
    constexpr int AEG = A16 ? 8 : 4, BEG = B16 ? 8 : 4;     // elements per 16-byte load granule
    constexpr int AEB = A16 ? 2 : 4, BEB = B16 ? 2 : 4;     // bytes per element in HBM
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr bool A_KM = (MODE == MODE_WGRAD);
    constexpr bool B_KM = (MODE != MODE_FWD && MODE != MODE_FWD_C3);
    constexpr int LDA = A_KM ? (BM + 4) : (KT + 4);
    constexpr int LDB = B_KM ? (BN + 4) : (KT + 4);
    constexpr int A_ROWS = A_KM ? KT : BM, A_COLS = A_KM ? BM : KT;
    constexpr int B_ROWS = B_KM ? KT : BN, B_COLS = B_KM ? BN : KT;
    constexpr int A_FLOATS = A_ROWS * LDA, B_FLOATS = B_ROWS * LDB;
    constexpr int STAGE = A_FLOATS + B_FLOATS;
    constexpr int A_CQ = A_COLS / AEG, B_CQ = B_COLS / BEG;
    constexpr int NVA = A_ROWS * A_CQ / 256, NVB = B_ROWS * B_CQ / 256;
    constexpr int A_RSTEP = 256 / A_CQ, B_RSTEP = 256 / B_CQ;
    static_assert(NVA >= 1 && NVB >= 1, "tile too small for 256 threads");
    static_assert(256 % A_CQ == 0 && 256 % B_CQ == 0, "row mapping");

    constexpr int EPI_FLOATS = 4 * 32 * 68;          // epilogue transpose regions, one [32][68] per wave
    // PREC 1 (bf16 LDS images, element = 2 bytes): k-contiguous [row][KT+8] (row stride 144 / 80 B: the 16 lanes of a
    // ds_read_b128 group land on 16 different 16-B slots), k-major [kk][cols+32] (row stride = 64 mod 256 B: the four
    // rows of a ds_read_b64_tr_b16 block land on four different 64-B bank segments) -- both conflict-free.
    // PREC 2 ("f32x3"): every fp32 operand value is split into THREE bf16 planes hi + mid + lo (24 significand bits in
    // all), K-tile 16, and each 32x32x16 product block is six bf16 MFMAs (see body_h).
    constexpr int NP = PREC == 2 ? 3 : 1;
    // PREC 2 k-contiguous images: rows of exactly 16 bf16 = two 16-B chunks, no padding; the chunk index is XORed with
    // bit 3 of the row, so the rows a ds_read_b128 lane group touches (r and r + 8k) land on different 16-B slots.
    constexpr bool SWZ = PREC == 2;
    constexpr int LDAH = A_KM ? (BM + 32) : (SWZ ? KT : KT + 8);
    constexpr int LDBH = B_KM ? (BN + 32) : (SWZ ? KT : KT + 8);
    constexpr int APL_BYTES = A_ROWS * LDAH * 2, BPL_BYTES = B_ROWS * LDBH * 2;     // one plane
    constexpr int AH_BYTES = NP * APL_BYTES, BH_BYTES = NP * BPL_BYTES;
    constexpr int STAGEH_FLOATS = (AH_BYTES + BH_BYTES) / 4;
    constexpr int MAIN_FLOATS = PREC >= 1 ? 2 * STAGEH_FLOATS : 2 * STAGE;
    __shared__ __attribute__((aligned(16))) float smem[MAIN_FLOATS > EPI_FLOATS ? MAIN_FLOATS : EPI_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // grouped launch (dg_conv_*_g): blockIdx.z = problem index; the problems share geometry and plan, only the tensors differ
    const int grp = blockIdx.z;
    const float* const pA = dg_group_ptr(p.A, p.gdA, grp);
    const float* const pB = dg_group_ptr(p.B, p.gdB, grp);
    float* const pC = dg_group_ptr(p.C, p.gdC, grp);
    float* const pPart = dg_group_ptr(p.part, p.gdPart, grp);
    float* const pStat = dg_group_ptr(p.stat, p.gdStat, grp);
    long long* const stp = (p.stamps != nullptr && tid == 0) ? p.stamps + (long)blockIdx.x * 8 : nullptr;
    if (stp) {
        stp[0] = wall_clock64();
        stp[1] = clock64();
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stp[6] = ((long long)xcc << 32) | hwid;
    }
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    int bid = blockIdx.x;
    int tn, tm, parity = 0, split;
    if (p.xcd_group == 3) {
        // ONE column tile (forward with <= 128 output channels): vertically adjacent row tiles share half of their input rows (a
        // 4x4 / stride-2 window reaches one input row into the tile above and below).  Dealt round robin, neighbours land on
        // different XCDs and every input row crosses the fabric twice (counters: 1.67x the algorithmic bytes after the group walk);
        // here XCD x takes the row tiles [x tilesM / 8, (x + 1) tilesM / 8) in order, like the window kernels.
        const int per = p.tilesM >> 3, x = bid & 7, j = bid >> 3;
        tn = 0;
        tm = x * per + j % per;
        split = j / per;
    } else if (p.xcd_group == 2) {
        // Weight-dominated layers (deep stages: the B operand is tens to hundreds of MB, A a few MB): the row tiles
        // that stream the SAME weight columns / K-slice are the ones that must share an L2 -- the G = tilesM row tiles
        // of one (column tile, parity, split) get blockIdx values 8 apart (one XCD, one dispatch window).
        const int G = p.tilesM;
        tm = (bid >> 3) % G;
        int rest = (bid / (8 * G)) * 8 + (bid & 7);
        if (MODE == MODE_DGRAD_S2) {
            parity = rest & 3;
            rest >>= 2;
        }
        tn = rest % p.tilesN;
        split = rest / p.tilesN;
    } else if (p.xcd_group) {
        // Workgroups that read the same operand-A rows -- the N-tile columns of a row tile and, for the transposed
        // conv, its four output-parity classes -- should share an L2.  Workgroups are dealt to the 8 XCDs round
        // robin by blockIdx, so such a group of G workgroups gets blockIdx values 8 apart (same XCD, dispatched
        // within one window of 8*G): inner = (bid / 8) % G, (row tile, split) index = (bid / (8G)) * 8 + bid % 8.
        const int G = p.tilesN * (MODE == MODE_DGRAD_S2 ? 4 : 1);
        int inner = (bid >> 3) % G;
        int rest = (bid / (8 * G)) * 8 + (bid & 7);
        if (MODE == MODE_DGRAD_S2) {
            parity = inner & 3;
            inner >>= 2;
        }
        tn = inner;
        tm = rest % p.tilesM;
        split = rest / p.tilesM;
    } else {
        tn = bid % p.tilesN;
        bid /= p.tilesN;
        tm = bid % p.tilesM;
        bid /= p.tilesM;
        split = bid;
        if (MODE == MODE_DGRAD_S2) {
            parity = bid & 3;
            split = bid >> 2;
        }
    }
    const int ph = parity >> 1, pw = parity & 1;
    const int m0 = tm * BM, n0 = tn * BN;
    const int it_begin = split * p.itPerSplit;
    const int it_end = min(p.nIt, it_begin + p.itPerSplit);

    const float* __restrict__ Ag = pA;
    const float* __restrict__ Bg = pB;
    const int H = p.H, W = p.W, Cc = p.Cc, K = p.K, Ho = p.Ho, Wo = p.Wo;
    const int lgWo = p.lgWo, lgHW = p.lgWo + p.lgHo;

    const int acq = tid % A_CQ, arow0 = tid / A_CQ;
    const int bcq = tid % B_CQ, brow0 = tid / B_CQ;

    // ---- per-thread row bookkeeping (fixed over the K loop except in WGRAD) --------------------
    int a_pix[NVA], a_y[NVA], a_x[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int row = arow0 + i * A_RSTEP;
        a_pix[i] = 0;
        a_y[i] = NEG_BIG;
        a_x[i] = 0;
        if (MODE == MODE_FWD) {
            const int m = m0 + row;
            if (m < p.M) {
                const int ox = m & (Wo - 1), oy = (m >> lgWo) & (Ho - 1), n = m >> lgHW;
                a_y[i] = oy * p.stride - p.pad;
                a_x[i] = ox * p.stride - p.pad;
                a_pix[i] = (n * H + a_y[i]) * W + a_x[i];
            }
        } else if (MODE == MODE_FWD_C3) {
            // this thread's float4 is filter row r = acq (4 taps s = 0..3) of channel `it`
            const int m = m0 + row;
            if (m < p.M) {
                const int ox = m & (Wo - 1), oy = (m >> lgWo) & (Ho - 1), n = m >> lgHW;
                a_y[i] = oy * 2 - 1 + acq;
                a_x[i] = ox * 2 - 1;
                a_pix[i] = (n * 3 * H + a_y[i]) * W + a_x[i];  // channel 0 plane, NCHW
            }
        } else if (MODE == MODE_DGRAD_S2) {
            const int m = m0 + row;
            if (m < p.M) {
                a_x[i] = m & (Wo - 1);
                a_y[i] = (m >> lgWo) & (Ho - 1);
                a_pix[i] = m;  // (n*Ho + a)*Wo + b == m
            }
        } else if (MODE == MODE_DGRAD_PLAIN) {
            const int m = m0 + row;
            if (m < p.M) {
                a_y[i] = 0;
                a_pix[i] = m;
            }
        }
    }
    // WGRAD: B columns are (tap, c); fixed per thread
    int wg_r = 0, wg_s = 0, wg_c = 0;
    bool wg_colok = true;
    if (MODE == MODE_WGRAD) {
        const int j = n0 + bcq * BEG;
        wg_colok = j < p.Ng;
        const int tap = wg_colok ? j / Cc : 0;
        wg_c = j - tap * Cc;
        wg_r = tap >> 2;
        wg_s = tap & 3;
    }
    // WGRAD (BUF) addressing constants, see load_B
    const bool wg_s2 = p.stride == 2;
    const int wg_lpm = wg_s2 ? 2 : 4, wg_pxm = wg_s2 ? -1 : 0;
    const int wg_cst = wg_s2 ? (wg_r - 1) * W + (wg_s - 1) : wg_r * 4 + wg_s;
    const int wg_ybad = !wg_s2 ? -1 : (wg_r == 0 ? 0 : (wg_r == 3 ? Ho - 1 : -1));
    const int wg_xbad = !wg_s2 ? -1 : (wg_s == 0 ? 0 : (wg_s == 3 ? Wo - 1 : -1));
    const int wg_colbad = wg_colok ? 0 : -1;

    // ---- BUF: per-row byte offsets and per-row tap-validity bits ----------------------------------------
    constexpr int OOR = (int)0x80000000;     // any offset with this bit set is beyond a < 2 GiB tensor
    // p.dbg_zero (timing experiments only, tools/bench_ops.py --dbg_zero): bit 0 / 1 give the A / B descriptor zero records, so
    // every load through it is dropped by the range check while the instruction stream stays (guide, section 7)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)pA, 0, BUF ? ((p.dbg_zero & 1) ? 0 : (int)p.abytes) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)pB, 0, BUF ? ((p.dbg_zero & 2) ? 0 : (int)p.bbytes) : 0, 0x00020000);
    // PLN: one descriptor per plane (an offset that runs off the end of a plane must not land in the next one)
    const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)pA + (PLN ? p.a_plane : 0)), 0, PLN ? ((p.dbg_zero & 1) ? 0 : (int)p.abytes) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)pA + (PLN ? 2 * p.a_plane : 0)), 0, PLN ? ((p.dbg_zero & 1) ? 0 : (int)p.abytes) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)pB + (PLN ? p.b_plane : 0)), 0, PLN ? ((p.dbg_zero & 2) ? 0 : (int)p.bbytes) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)pB + (PLN ? 2 * p.b_plane : 0)), 0, PLN ? ((p.dbg_zero & 2) ? 0 : (int)p.bbytes) : 0, 0x00020000);
    int a_ob[NVA], a_inv[NVA], b_ob[NVB];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        a_ob[i] = 0;
        a_inv[i] = 0;
        if (BUF && MODE == MODE_FWD) {
            a_ob[i] = (a_pix[i] * Cc + acq * AEG) * AEB;
            int colok = 0, okmask = 0;             // 4 column bits, replicated into every valid filter row
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) colok |= ((unsigned)(a_x[i] + sx) < (unsigned)W) ? (1 << sx) : 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) okmask |= ((unsigned)(a_y[i] + r) < (unsigned)H) ? (colok << (4 * r)) : 0;
            a_inv[i] = ~okmask & 0xFFFF;
        } else if (BUF && MODE == MODE_DGRAD_S2) {
            a_ob[i] = (a_pix[i] * K + acq * AEG) * AEB;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ty = t >> 1, tx = t & 1;
                const int dyo = ph == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 0 : 1);
                const int dxo = pw == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 0 : 1);
                const bool ok = (unsigned)(a_y[i] + dyo) < (unsigned)Ho && (unsigned)(a_x[i] + dxo) < (unsigned)Wo;
                a_inv[i] |= ok ? 0 : (1 << t);
            }
        } else if (BUF && MODE == MODE_WGRAD) {
            const int kcol = m0 + acq * AEG;
            a_ob[i] = kcol < K ? ((arow0 + i * A_RSTEP) * K + kcol) * AEB : OOR;   // rows >= R run off the end: zeros
        }
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
        b_ob[i] = 0;
        if (BUF && MODE == MODE_FWD) {
            const int k = n0 + brow0 + i * B_RSTEP;
            b_ob[i] = k < K ? (k * 16 * Cc + bcq * BEG) * BEB : OOR;
        } else if (BUF && MODE == MODE_DGRAD_S2) {
            const int col = n0 + bcq * BEG;
            b_ob[i] = col < Cc ? ((brow0 + i * B_RSTEP) * 16 * Cc + col) * BEB : OOR;
        }
    }
    auto ld4b = [&](const __amdgpu_buffer_rsrc_t& r, int byte_off) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
    };

    // ---- K-iteration state ------------------------------------------------------------------------
    // FWD: (tap, chunk) over (16, Cc/KT);  DGRAD_S2: (t, chunk) over (4, K/KT)
    int tap = 0, chunk = 0;
    // FWD walks the reduction channel-chunk major, taps inner, and the taps in the order r, s in (0, 2, 1, 3): the
    // taps that read the same input pixels (s and s+2 shift by one output column, r and r+2 by one output row) are
    // then 1 and 4 K-tiles apart instead of 2*kchunks and 8*kchunks, so the re-reads hit the XCD's L2 more often
    // (PMC FETCH_SIZE, profiles/).  `tap` counts 0..15 in that order; (fwd_r, fwd_s) is the filter position.
    // f32x3 forward with whole 64-channel groups (PREC 2: a 16-channel K-tile is a 64-byte HALF of a pixel's 128-byte line):
    // the walk is 64-channel group major, taps inside, and the group's FOUR K-tiles innermost, so the two halves of a line are
    // requested by back-to-back tiles (the order of igemm_dma_x3.hip: there the same change was worth 8-14 %); `chunk` stays the
    // 16-channel chunk index the offsets use, = cgrp * SUBF + sub
    const int SUBF = (PREC == 2 && MODE == MODE_FWD && (Cc & 63) == 0 && dg_get_subwalk(p)) ? 4 : 1;
    int sub = 0, cgrp = 0;
    if (MODE == MODE_FWD) {
        sub = it_begin % SUBF;
        tap = (it_begin / SUBF) & 15;
        cgrp = it_begin / (SUBF * 16);
        chunk = cgrp * SUBF + sub;
    } else if (MODE == MODE_DGRAD_S2) {     // same idea: the 2x2 taps of a parity class are adjacent K-tiles
        chunk = it_begin >> 2;
        tap = it_begin & 3;
    }
    auto fwd_r = [&]() { const int a = tap >> 2; return ((a & 1) << 1) | (a >> 1); };
    auto fwd_s = [&]() { const int b = tap & 3; return ((b & 1) << 1) | (b >> 1); };

    f32x4 ra[2][NVA], rb[2][NVB];   // two register sets: tiles t+1 (waiting to be written to LDS) and t+2 (in flight)
    f32x4 ra1[2][PLN ? NVA : 1], ra2[2][PLN ? NVA : 1], rb1[2][PLN ? NVB : 1], rb2[2][PLN ? NVB : 1];   // PLN: the mid / lo planes

    // Branch-free tile loads: an invalid (padding / out-of-range) vector loads from the tensor base and
    // is zeroed by a select, so the whole K-loop body is ONE basic block and the address arithmetic,
    // global loads, LDS reads and LDS writes can be interleaved between MFMAs (each 64-cycle
    // v_mfma_f32_32x32x2_f32 leaves ~48 issue cycles for other instruction types).
    const float* const zp = dg_zero16;
    auto ld4 = [&](const float* base, long off, bool ok) -> f32x4 {
        const float* ptr = ok ? base + off : zp;
        return *(const f32x4*)ptr;
    };
    auto load_A = [&](int set, int i, int it) {
        if (BUF && MODE == MODE_FWD) {
            const int r = fwd_r(), sx = fwd_s();
            const int soff = ((r * W + sx) * Cc + chunk * KT) * AEB;                // wave-uniform
            const int voff = (a_ob[i] + soff) | -((a_inv[i] >> (r * 4 + sx)) & 1);
            ra[set][i] = ld4b(rA, voff);
            if constexpr (PLN) {
                ra1[set][i] = ld4b(rA1, voff);
                ra2[set][i] = ld4b(rA2, voff);
            }
        } else if (BUF && MODE == MODE_DGRAD_S2) {
            const int ty = tap >> 1, tx = tap & 1;
            const int dyo = ph == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 0 : 1);
            const int dxo = pw == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 0 : 1);
            const int soff = ((dyo * Wo + dxo) * K + chunk * KT) * AEB;             // wave-uniform
            ra[set][i] = ld4b(rA, (a_ob[i] + soff) | -((a_inv[i] >> tap) & 1));
        } else if (BUF && MODE == MODE_DGRAD_PLAIN) {
            const int kcol = it * KT + acq * AEG;
            ra[set][i] = ld4b(rA, (a_y[i] >= 0 && kcol < K) ? (a_pix[i] * K + kcol) * AEB : OOR);
        } else if (BUF && MODE == MODE_WGRAD) {
            ra[set][i] = ld4b(rA, a_ob[i] + it * (KT * AEB) * K);
        } else if (MODE == MODE_FWD) {
            const int r = fwd_r(), s = fwd_s(), c0 = chunk * KT;
            const int iy = a_y[i] + r, ix = a_x[i] + s;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            ra[set][i] = ld4(Ag, (long)(a_pix[i] + r * W + s) * Cc + c0 + acq * 4, ok);
        } else if (MODE == MODE_FWD_C3) {
            static_assert(MODE != MODE_FWD_C3 || KT == 16, "FWD_C3 iterates one input channel (16 taps) per K-tile");
            const bool rowok = (unsigned)a_y[i] < (unsigned)H;
            const long base = (long)a_pix[i] + (long)it * H * W;
            f32x4 v;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool ok = rowok && (unsigned)(a_x[i] + s) < (unsigned)W;
                const float* ptr = ok ? Ag + base + s : zp;
                v[s] = *ptr;
            }
            ra[set][i] = v;
        } else if (MODE == MODE_DGRAD_S2) {
            const int ty = tap >> 1, tx = tap & 1, k0 = chunk * KT;
            const int dyo = ph == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 0 : 1);
            const int dxo = pw == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 0 : 1);
            const int iy = a_y[i] + dyo, ix = a_x[i] + dxo;
            const bool ok = (unsigned)iy < (unsigned)Ho && (unsigned)ix < (unsigned)Wo;
            ra[set][i] = ld4(Ag, (long)(a_pix[i] + dyo * Wo + dxo) * K + k0 + acq * 4, ok);
        } else if (MODE == MODE_DGRAD_PLAIN) {
            const int kcol = it * KT + acq * 4;
            ra[set][i] = ld4(Ag, (long)a_pix[i] * K + kcol, a_y[i] >= 0 && kcol < K);
        } else {  // WGRAD: rows are reduction pixels
            const int mrow = it * KT + arow0 + i * A_RSTEP;
            const int kcol = m0 + acq * 4;
            ra[set][i] = ld4(Ag, (long)mrow * K + kcol, mrow < p.R && kcol < K);
        }
    };
    auto load_B = [&](int set, int i, int it) {
        if (BUF && MODE == MODE_FWD) {
            const int voff = b_ob[i] + (((fwd_r() * 4 + fwd_s()) * Cc) + chunk * KT) * BEB;
            rb[set][i] = ld4b(rB, voff);
            if constexpr (PLN) {
                rb1[set][i] = ld4b(rB1, voff);
                rb2[set][i] = ld4b(rB2, voff);
            }
        } else if (BUF && MODE == MODE_DGRAD_S2) {
            const int ty = tap >> 1, tx = tap & 1;
            const int r = ph == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 2 : 0);
            const int sx = pw == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 2 : 0);
            rb[set][i] = ld4b(rB, b_ob[i] + ((chunk * KT * 16 + r * 4 + sx) * Cc) * BEB);
        } else if (BUF && MODE == MODE_DGRAD_PLAIN) {
            const int col = n0 + bcq * BEG;
            const int k = it * KT + brow0 + i * B_RSTEP;
            rb[set][i] = ld4b(rB, (k < K && col < p.Ng) ? (k * p.Ng + col) * BEB : OOR);
        } else if (BUF && MODE == MODE_WGRAD) {
            // reduction row = output pixel mrow = (n, oy, ox) packed (Ho, Wo powers of two).  k4 s2 p1 has
            // H = 2Ho, W = 2Wo, so the input pixel of tap (r, s) is 4*mrow - 2*ox + (r-1)*W + (s-1) and only
            // (r=0, oy=0), (r=3, oy=Ho-1) and the two column analogues fall into the padding; the 4x4 head
            // (s1 p0) reads pixel 16*mrow + 4r + s.  All flags are per-thread constants (wg_*): no branches.
            const int mrow = it * KT + brow0 + i * B_RSTEP;
            const int oxv = mrow & (Wo - 1), oyv = (mrow >> lgWo) & (Ho - 1);
            const int bad = (oyv == wg_ybad) | (oxv == wg_xbad) | (mrow >= p.R);
            const int pix = (mrow << wg_lpm) - ((oxv << 1) & wg_pxm) + wg_cst;
            rb[set][i] = ld4b(rB, ((pix * Cc + wg_c) * BEB) | wg_colbad | -bad);
        } else if (MODE == MODE_FWD) {
            const int k = n0 + brow0 + i * B_RSTEP;
            rb[set][i] = ld4(Bg, (long)k * 16 * Cc + (long)((fwd_r() * 4 + fwd_s()) * Cc + chunk * KT) + bcq * 4, k < K);
        } else if (MODE == MODE_FWD_C3) {
            const int k = n0 + brow0 + i * B_RSTEP;
            rb[set][i] = ld4(Bg, (long)k * 48 + it * 16 + bcq * 4, k < K);
        } else if (MODE == MODE_DGRAD_S2) {
            const int ty = tap >> 1, tx = tap & 1, k0 = chunk * KT;
            const int r = ph == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 2 : 0);
            const int s = pw == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 2 : 0);
            const int col = n0 + bcq * 4;
            const int k = k0 + brow0 + i * B_RSTEP;
            rb[set][i] = ld4(Bg, (long)(k * 16 + r * 4 + s) * Cc + col, col < Cc);
        } else if (MODE == MODE_DGRAD_PLAIN) {
            const int col = n0 + bcq * 4;
            const int k = it * KT + brow0 + i * B_RSTEP;
            rb[set][i] = ld4(Bg, (long)k * p.Ng + col, k < K && col < p.Ng);
        } else {
            const int mrow = it * KT + brow0 + i * B_RSTEP;
            const int ox = mrow & (Wo - 1), oy = (mrow >> lgWo) & (Ho - 1), n = mrow >> lgHW;
            const int iy = oy * p.stride - p.pad + wg_r, ix = ox * p.stride - p.pad + wg_s;
            const bool ok = wg_colok && mrow < p.R && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            rb[set][i] = ld4(Bg, (long)((n * H + iy) * W + ix) * Cc + wg_c, ok);
        }
    };
    // (tap, chunk) advance without control flow; `go` = 0 freezes the state on the last iteration
    auto advance = [&](int go) {
        if (MODE == MODE_FWD) {
            sub += go;
            const int w1 = (sub == SUBF) ? 1 : 0;
            sub = w1 ? 0 : sub;
            tap += w1;
            const int wrap = (tap == 16) ? 1 : 0;
            tap = wrap ? 0 : tap;
            cgrp += wrap;
            chunk = cgrp * SUBF + sub;
        } else if (MODE == MODE_DGRAD_S2) {
            tap += go;
            const int wrap = (tap == 4) ? 1 : 0;
            tap = wrap ? 0 : tap;
            chunk += wrap;
        }
    };
    auto store_A = [&](float* stage, int set, int i) { *(f32x4*)(stage + (arow0 + i * A_RSTEP) * LDA + acq * 4) = ra[set][i]; };
    auto store_B = [&](float* stage, int set, int i) { *(f32x4*)(stage + A_FLOATS + (brow0 + i * B_RSTEP) * LDB + bcq * 4) = rb[set][i]; };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LDS fragment fetch for one 8-wide k group: a[i][0..3], b[i][0..3] = the lane's operands of 4 MFMAs
    auto fetch = [&](const float* As, const float* Bs, int kb, float (&a)[2][4], float (&b)[2][4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + l31;
            if (!A_KM) {
                const f32x4 v = *(const f32x4*)(As + row * LDA + kb * 8 + 4 * lh);
                a[i][0] = v[0]; a[i][1] = v[1]; a[i][2] = v[2]; a[i][3] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) a[i][j] = As[(kb * 8 + 4 * lh + j) * LDA + row];
            }
            const int col = wn * 64 + i * 32 + l31;
            if (!B_KM) {
                const f32x4 v = *(const f32x4*)(Bs + col * LDB + kb * 8 + 4 * lh);
                b[i][0] = v[0]; b[i][1] = v[1]; b[i][2] = v[2]; b[i][3] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[i][j] = Bs[(kb * 8 + 4 * lh + j) * LDB + col];
            }
        }
    };

    constexpr int NG = KT / 8;            // k groups per tile (16 MFMAs each)
    constexpr int SC = NG * 4;            // sub-chunks of 4 MFMAs
    constexpr int NLD = NVA + NVB;        // global-load / LDS-store work items per tile
    constexpr int EB = 2;                 // the tile barrier sits EB sub-chunks (8 MFMAs) before the end of the tile
    static_assert(NLD <= SC - EB, "more load items than sub-chunks");
    static_assert(NG % 2 == 0, "fragment set 0 must be free during the last k group");
    constexpr int ST0 = SC - EB - NLD;    // first sub-chunk that carries an LDS store
    constexpr int SB = SC - EB;           // sub-chunk that opens with the barrier

    // Pipeline: tile t lives in LDS buffer (t & 1); tile t+1 sits in register set ((t+1) & 1) and is written
    // to the other LDS buffer in sub-chunks ST0..SB-1 of iteration t; tile t+2 is loaded into register set
    // (t & 1) in the first half of iteration t.  Global-load latency tolerance = 1.4 iterations.
    // The ONE barrier per tile comes 8 MFMAs before the end of the tile, and the first fragment group of tile
    // t+1 is fetched right behind it: barrier skew and the LDS read latency are covered by the current
    // tile's last MFMAs instead of opening every tile with an empty matrix pipe (stamps: the loop ran at
    // 86-87 % of the sustained MFMA rate with the barrier at the tile boundary).
    // Tiles past the end are re-loaded from the last position and never used (no branches in the body).
    const int it_last = it_end - 1;
    float fa[2][2][4], fb[2][2][4];   // double-buffered LDS fragments; set 0 is carried across tiles
    if constexpr (PREC == 0) {
    if (it_begin < it_end) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) load_A(0, i, it_begin);
#pragma unroll
        for (int i = 0; i < NVB; ++i) load_B(0, i, it_begin);
        advance(it_begin + 1 < it_end ? 1 : 0);
#pragma unroll
        for (int i = 0; i < NVA; ++i) load_A(1, i, min(it_begin + 1, it_last));
#pragma unroll
        for (int i = 0; i < NVB; ++i) load_B(1, i, min(it_begin + 1, it_last));
        advance(it_begin + 2 < it_end ? 1 : 0);
#pragma unroll
        for (int i = 0; i < NVA; ++i) store_A(smem, 0, i);
#pragma unroll
        for (int i = 0; i < NVB; ++i) store_B(smem, 0, i);
    }
    __syncthreads();
    if (stp) stp[2] = clock64();
    fetch(smem, smem + A_FLOATS, 0, fa[0], fb[0]);
    }

    // one K-tile; P = parity of (it - it_begin) = LDS buffer of the current tile = register set to refill
    auto body = [&](auto P, int it) {
        constexpr int p_ = decltype(P)::value;
        const float* As = smem + p_ * STAGE;
        const float* Bs = As + A_FLOATS;
        float* nxt = smem + (p_ ^ 1) * STAGE;
        const int itn = min(it + 2, it_last);
#pragma unroll
        for (int sc = 0; sc < SC; ++sc) {
            const int g = sc >> 2, j = sc & 3;
            __builtin_amdgcn_sched_barrier(0);
            if (sc >= ST0 && sc < SB) {   // tile it+1: registers (set p^1) -> LDS
                const int q = sc - ST0;
                if (q < NVA) store_A(nxt, p_ ^ 1, q);
                else store_B(nxt, p_ ^ 1, q - NVA);
            }
            if (sc == SB) {
                // every wave's stores of tile it+1 are done and every wave's reads of tile it were consumed
                // (the last k group's fragments were waited for before sub-chunk SC-4)
                __syncthreads();
                fetch(nxt, nxt + A_FLOATS, 0, fa[0], fb[0]);
            }
            if (sc < NVA) load_A(p_, sc, itn);
            else if (sc < NLD) load_B(p_, sc - NVA, itn);
            if (j == 1 && g + 1 < NG) fetch(As, Bs, g + 1, fa[(g + 1) & 1], fb[(g + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i][j], fb[g & 1][jn][j], acc[i][jn], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        advance(it + 3 < it_end ? 1 : 0);
    };
    // ---- PREC 1: bf16 LDS tiles.  The fp32 operand vectors are rounded to bf16 (v_cvt_pk_bf16_f32, RNE) when they
    // are staged into LDS -- once per element instead of once per use -- so LDS carries half the bytes, the K-tile is
    // 64 and a fragment (8 k values of one row) is ONE ds_read_b128 (k-contiguous images) or two ds_read_b64_tr_b16
    // (k-major images [kk][cols]: the hardware transposes a 4 x 16 block per 16 lanes, so the NHWC / KRSC operands that
    // arrive reduction-major need no transposing stores).  Pipeline: ONE register set; iteration t writes tile t+1
    // (loaded during iteration t-1) into the other LDS buffer, re-issues the loads of tile t+2, multiplies tile t,
    // barrier.  Two workgroups share a CU, so one's conversions / LDS writes run under the other's MFMAs.
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) bf16x4* lds_bf4_ptr;
    char* const smb = (char*)smem;
    // fp32 -> bf16 planes.  PREC 1: one RNE rounding.  PREC 2: hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid);
    // both subtractions are exact in fp32, so hi + mid + lo carries 24 significand bits of v.
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    // two floats -> one dword of two RNE-rounded bf16 (ONE v_cvt_pk_bf16_f32); the fp32 value of each half is a shift / a mask
    auto pk2 = [](float a, float b) -> unsigned { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2)); };
    auto lo_f = [](unsigned p) -> float { return __builtin_bit_cast(float, p << 16); };
    auto hi_f = [](unsigned p) -> float { return __builtin_bit_cast(float, p & 0xffff0000u); };
    auto split_store = [&](char* dst, int plane_bytes, const f32x4& v) {
        const unsigned h0 = pk2(v[0], v[1]), h1 = pk2(v[2], v[3]);
        *(u32x2*)dst = (u32x2){h0, h1};
        if constexpr (PREC == 2) {      // 5.5 VALU per element: 1.5 packed converts, 2 unpacks, 2 exact subtractions
            const float r0 = v[0] - lo_f(h0), r1 = v[1] - hi_f(h0), r2 = v[2] - lo_f(h1), r3 = v[3] - hi_f(h1);
            const unsigned m0 = pk2(r0, r1), m1 = pk2(r2, r3);
            *(u32x2*)(dst + plane_bytes) = (u32x2){m0, m1};
            const unsigned l0 = pk2(r0 - lo_f(m0), r1 - hi_f(m0)), l1 = pk2(r2 - lo_f(m1), r3 - hi_f(m1));
            *(u32x2*)(dst + 2 * plane_bytes) = (u32x2){l0, l1};
        }
    };
    // NSR register sets: tile j waits in set (j - it_begin) % NSR between its global load and its LDS store, so loads are
    // in flight for NSR K-tiles (a bf16 / f32x3 K-tile is only 512-768 MFMA cycles: one tile does not cover an HBM miss)
    constexpr int NSR = PREC == 2 ? 2 : 1;
    auto kc_off = [&](int row, int ld, int k) -> int {     // byte offset of element (row, k) of a k-contiguous image
        if (SWZ) return row * (KT * 2) + ((((k >> 3) ^ (row >> 3)) & 1) << 4) + (k & 7) * 2;
        return (row * ld + k) * 2;
    };
    auto sth_A = [&](char* stage, int set, int i) {
        const int row = arow0 + i * A_RSTEP;
        char* dst = stage + (A_KM ? (row * LDAH + acq * AEG) * 2 : kc_off(row, LDAH, acq * AEG));
        if constexpr (PLN) {
            *(f32x4*)dst = ra[set][i];
            *(f32x4*)(dst + APL_BYTES) = ra1[set][i];
            *(f32x4*)(dst + 2 * APL_BYTES) = ra2[set][i];
        } else if constexpr (A16) *(f32x4*)dst = ra[set][i];            // already bf16: 8 elements, one ds_write_b128
        else split_store(dst, APL_BYTES, ra[set][i]);
    };
    auto sth_B = [&](char* stage, int set, int i) {
        const int row = brow0 + i * B_RSTEP;
        char* dst = stage + AH_BYTES + (B_KM ? (row * LDBH + bcq * BEG) * 2 : kc_off(row, LDBH, bcq * BEG));
        if constexpr (PLN) {
            *(f32x4*)dst = rb[set][i];
            *(f32x4*)(dst + BPL_BYTES) = rb1[set][i];
            *(f32x4*)(dst + 2 * BPL_BYTES) = rb2[set][i];
        } else if constexpr (B16) *(f32x4*)dst = rb[set][i];
        else split_store(dst, BPL_BYTES, rb[set][i]);
    };
    // transposed fragment read: lane l = 16g + 4q + p supplies row (k0 + q), columns c0 + 16*(g&1) + 4p .. +3 and
    // receives column c0 + (l & 31), rows k0 .. k0+3 (k0 already includes the lane half's 8*(l>>5))
    const int tr_q = (lane >> 2) & 3, tr_c = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    auto frag_km = [&](const char* img, int ld, int k0, int c0) -> bf16x8 {
        const char* p0 = img + ((k0 + tr_q) * ld + c0 + tr_c) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)p0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(p0 + 4 * ld * 2));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // One K-tile, hand-interleaved (the compiler, left alone, emits the whole conversion / LDS-write block of tile it+1
    // first and the MFMAs back to back behind it, so a wave's matrix phase and its VALU phase never overlap; its
    // scheduler also ignored sched_group_barrier hints here).  The staging work of tile it+1 is cut into steps and
    // one or two steps follow every MFMA, fenced with sched_barrier like the fp32 body:
    //   PREC 1: per load item   {convert + LDS store, re-issue the item's global load for tile it+3}
    //   PREC 2: per load item   {hi planes + store, residuals, mid planes + store, residuals, lo planes + store, re-load}
    // MFMA q of the tile: PREC 1 k16 step q/4, accumulators (q/2)&1, q&1; PREC 2 plane pair q/4 of the six.
    constexpr int NMF = (KT / 16) * 4 * (PREC == 2 ? 6 : 1);          // MFMAs per K-tile per wave
    constexpr int SPI = PLN ? 4 : (PREC == 2 ? 6 : 2);                 // staging steps per load item (PLN: three plane stores, re-load)
    constexpr int NST = NLD * SPI;
    constexpr int SPM = (NST + NMF - 1) / NMF;                         // staging steps behind each MFMA
    // PREC 2 fragments live across tiles: set (tile parity) is multiplied while the next tile's set is fetched right
    // behind the tile barrier, which sits 8 MFMAs before the end of the tile (as in the fp32 body): barrier skew and
    // the LDS read latency are covered by the current tile's last MFMAs.
    bf16x8 xa[2][NP][2], xb[2][NP][2];
    auto frags_x = [&](const char* stage, int set) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32, col = wn * 64 + i * 32;
                const char* Ap = stage + pl * APL_BYTES;
                const char* Bp = stage + AH_BYTES + pl * BPL_BYTES;
                if (!A_KM) xa[set][pl][i] = *(const bf16x8*)(Ap + kc_off(row + l31, LDAH, 8 * lh));
                else xa[set][pl][i] = frag_km(Ap, LDAH, 8 * lh, row);
                if (!B_KM) xb[set][pl][i] = *(const bf16x8*)(Bp + kc_off(col + l31, LDBH, 8 * lh));
                else xb[set][pl][i] = frag_km(Bp, LDBH, 8 * lh, col);
            }
    };
    constexpr int QB = NMF > 8 ? NMF - 8 : 1;                                         // the tile barrier comes before MFMA QB
    constexpr int SPQ = (NST + QB - 1) / QB;                            // staging steps behind each of the first QB MFMAs
    auto body_h = [&](auto P, int it) {
        constexpr int p_ = decltype(P)::value;
        char* nxt = smb + (p_ ^ 1) * (AH_BYTES + BH_BYTES);
        const int itn = min(it + 1 + NSR, it_last);  // tile it+3 -> the register set just emptied (re-loads the last tile at the end)
        unsigned th0 = 0, th1 = 0, tm0 = 0, tm1 = 0;
        float tr0 = 0.f, tr1 = 0.f, tr2 = 0.f, tr3 = 0.f;
        auto stage_step = [&](int st) {
            const int item = st / SPI, ph = st % SPI;
            const bool isA = item < NVA;
            const int ii = isA ? item : item - NVA;
            const f32x4 v = isA ? ra[p_ ^ 1][ii] : rb[p_ ^ 1][ii];
            char* dst;
            int plane;
            if (isA) {
                const int row = arow0 + ii * A_RSTEP;
                dst = nxt + (A_KM ? (row * LDAH + acq * AEG) * 2 : kc_off(row, LDAH, acq * AEG));
                plane = APL_BYTES;
            } else {
                const int row = brow0 + ii * B_RSTEP;
                dst = nxt + AH_BYTES + (B_KM ? (row * LDBH + bcq * BEG) * 2 : kc_off(row, LDBH, bcq * BEG));
                plane = BPL_BYTES;
            }
            if constexpr (PLN) {
                if (ph == 0) *(f32x4*)dst = v;
                else if (ph == 1) *(f32x4*)(dst + plane) = isA ? ra1[p_ ^ 1][ii] : rb1[p_ ^ 1][ii];
                else if (ph == 2) *(f32x4*)(dst + 2 * plane) = isA ? ra2[p_ ^ 1][ii] : rb2[p_ ^ 1][ii];
                else if (isA) load_A(p_ ^ 1, ii, itn);
                else load_B(p_ ^ 1, ii, itn);
                return;
            }
            if (ph == 0) {
                th0 = pk2(v[0], v[1]); th1 = pk2(v[2], v[3]);
                *(u32x2*)dst = (u32x2){th0, th1};
            } else if (ph == SPI - 1) {               // the item's registers are free: re-issue its load for tile it+3
                if (isA) load_A(p_ ^ 1, ii, itn); else load_B(p_ ^ 1, ii, itn);
            } else if (ph == 1) {
                tr0 = v[0] - lo_f(th0); tr1 = v[1] - hi_f(th0); tr2 = v[2] - lo_f(th1); tr3 = v[3] - hi_f(th1);
            } else if (ph == 2) {
                tm0 = pk2(tr0, tr1); tm1 = pk2(tr2, tr3);
                *(u32x2*)(dst + plane) = (u32x2){tm0, tm1};
            } else if (ph == 3) {
                tr0 -= lo_f(tm0); tr1 -= hi_f(tm0); tr2 -= lo_f(tm1); tr3 -= hi_f(tm1);
            } else {
                *(u32x2*)(dst + 2 * plane) = (u32x2){pk2(tr0, tr1), pk2(tr2, tr3)};
            }
        };
#pragma unroll
        for (int q = 0; q < NMF; ++q) {
            __builtin_amdgcn_sched_barrier(0);
            if (q == QB) {
                // every wave's stores of tile it+1 are done (all staging steps sit behind MFMAs < QB) and every wave's
                // fragment reads of tile it were issued before its first MFMA
                __syncthreads();
                frags_x(nxt, p_ ^ 1);
            }
            const int i = (q >> 1) & 1, jn = q & 1;
            // a*b = sum over the 3x3 plane pairs; the three pairs dropped (mid*lo, lo*mid, lo*lo) are together
            // <= 2^-23 of |a*b|.  bf16 x bf16 products are exact in fp32; smallest terms are added first.
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[p_][PA[q >> 2]][i], xb[p_][PB[q >> 2]][jn], acc[i][jn], 0, 0, 0);
            if (q < QB) {
#pragma unroll
                for (int u = 0; u < SPQ; ++u)
                    if (q * SPQ + u < NST) stage_step(q * SPQ + u);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        advance(it + 2 + NSR < it_end ? 1 : 0);
    };
    // PREC 1, hand-interleaved like the PREC 2 body (the compiler otherwise emits all 16 conversions + LDS stores first, each
    // behind its own vmcnt wait, and the MFMAs after them).  ONE register set: behind MFMA q the load item(s) q of tile it+1
    // are converted and stored to the other LDS buffer and immediately re-loaded for tile it+2; fragments are double
    // buffered over the four k16 steps of the tile; the tile barrier is at the end.
    constexpr int IPM = (NLD + NMF - 1) / NMF;                          // load items behind each MFMA
    auto body_i = [&](auto P, int it) {
        constexpr int p_ = decltype(P)::value;
        const char* As = smb + p_ * (AH_BYTES + BH_BYTES);
        const char* Bs = As + AH_BYTES;
        char* nxt = smb + (p_ ^ 1) * (AH_BYTES + BH_BYTES);
        const int itn = min(it + 2, it_last);
        bf16x8 fa_[2][2], fb_[2][2];
        auto frags = [&](int s16, int buf) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32, col = wn * 64 + i * 32;
                if (!A_KM) fa_[buf][i] = *(const bf16x8*)(As + kc_off(row + l31, LDAH, s16 * 16 + 8 * lh));
                else fa_[buf][i] = frag_km(As, LDAH, s16 * 16 + 8 * lh, row);
                if (!B_KM) fb_[buf][i] = *(const bf16x8*)(Bs + kc_off(col + l31, LDBH, s16 * 16 + 8 * lh));
                else fb_[buf][i] = frag_km(Bs, LDBH, s16 * 16 + 8 * lh, col);
            }
        };
        frags(0, 0);
#pragma unroll
        for (int q = 0; q < NMF; ++q) {
            __builtin_amdgcn_sched_barrier(0);
            const int s16 = q >> 2, fbuf = s16 & 1;
            if ((q & 3) == 1 && s16 + 1 < KT / 16) frags(s16 + 1, fbuf ^ 1);
            acc[(q >> 1) & 1][q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[fbuf][(q >> 1) & 1], fb_[fbuf][q & 1], acc[(q >> 1) & 1][q & 1], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < IPM; ++u) {
                const int item = q * IPM + u;
                if (item < NVA) { sth_A(nxt, 0, item); load_A(0, item, itn); }
                else if (item < NLD) { sth_B(nxt, 0, item - NVA); load_B(0, item - NVA, itn); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        advance(it + 3 < it_end ? 1 : 0);
        __syncthreads();
    };
    if constexpr (PREC >= 1) {
        static_assert(!SWZ || KT == 16, "the XOR swizzle is written for 16-element rows");
        static_assert(KT % 16 == 0 && (LDAH * 2) % 16 == 0 && (LDBH * 2) % 16 == 0 && AH_BYTES % 16 == 0, "bf16 LDS image alignment");
        if (it_begin < it_end) {
#pragma unroll
            for (int i = 0; i < NVA; ++i) load_A(0, i, it_begin);
#pragma unroll
            for (int i = 0; i < NVB; ++i) load_B(0, i, it_begin);
            advance(it_begin + 1 < it_end ? 1 : 0);
            if constexpr (NSR == 2) {
#pragma unroll
                for (int i = 0; i < NVA; ++i) load_A(1, i, min(it_begin + 1, it_last));
#pragma unroll
                for (int i = 0; i < NVB; ++i) load_B(1, i, min(it_begin + 1, it_last));
                advance(it_begin + 2 < it_end ? 1 : 0);
            }
#pragma unroll
            for (int i = 0; i < NVA; ++i) sth_A(smb, 0, i);
#pragma unroll
            for (int i = 0; i < NVB; ++i) sth_B(smb, 0, i);
#pragma unroll
            for (int i = 0; i < NVA; ++i) load_A(0, i, min(it_begin + NSR, it_last));
#pragma unroll
            for (int i = 0; i < NVB; ++i) load_B(0, i, min(it_begin + NSR, it_last));
            advance(it_begin + NSR + 1 < it_end ? 1 : 0);
        }
        __syncthreads();
        if (stp) stp[2] = clock64();
        if constexpr (PREC == 2) frags_x(smb, 0);
        for (int it = it_begin; it < it_end; it += 2) {
            if constexpr (PREC == 2) {
                body_h(std::integral_constant<int, 0>{}, it);
                if (it + 1 < it_end) body_h(std::integral_constant<int, 1>{}, it + 1);
            } else {
                body_i(std::integral_constant<int, 0>{}, it);
                if (it + 1 < it_end) body_i(std::integral_constant<int, 1>{}, it + 1);
            }
        }
        if constexpr (PREC == 2) __syncthreads();      // the last tile's fragment prefetch read LDS behind the tile barrier
    } else {
        for (int it = it_begin; it < it_end; it += 2) {
            body(std::integral_constant<int, 0>{}, it);
            if (it + 1 < it_end) body(std::integral_constant<int, 1>{}, it + 1);
        }
    }

    if (stp) stp[3] = clock64();
    // ---- fused BatchNorm statistics: one partial row per (tile, wave row), shifted by the wave tile's
    //      first row so that sum/sumsq never cancel catastrophically; merged by bn_partials_finalize.
    if ((MODE == MODE_FWD || MODE == MODE_DGRAD_S2) && pStat != nullptr && pPart == nullptr) {
        const int row0 = m0 + wm * 64;
        const int nrows = min(64, p.M - row0);
        const int prow = ((MODE == MODE_DGRAD_S2 ? parity : 0) * p.tilesM + tm) * WM + wm;
        float* srow = pStat + (long)prow * p.stat_rs;
        if (wn == 0 && tn == 0 && lane == 0) srow[0] = (float)max(nrows, 0);
        if (nrows > 0) {
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const float sh = __shfl(acc[0][jn][0], l31, 64);
                float ssum = 0.f, ssq = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int lr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (lr < nrows) {
                            const float d = acc[i][jn][r] - sh;
                            ssum += d;
                            ssq += d * d;
                        }
                    }
                ssum += __shfl_xor(ssum, 32, 64);
                ssq += __shfl_xor(ssq, 32, 64);
                const int n = n0 + wn * 64 + jn * 32 + l31;
                if (lh == 0 && n < p.Ng) {
                    srow[4 + n] = sh;
                    srow[4 + p.Ng + n] = ssum;
                    srow[4 + 2 * p.Ng + n] = ssq;
                }
            }
        }
    }

    // ---- epilogue: acc[i][jn][r] = row (r&3)+8*(r>>2)+4*lh, col jn*32+l31 of the wave's 32x64 half tile.
    // Transposed through the (now idle) LDS, one private [32][68] region per wave, so that every lane stores
    // float4 and 16 lanes cover one 256-B output row segment: 16 store instructions per wave instead of 64
    // dword stores (the epilogue is store-ISSUE bound; stamps: 4.6-22 us -> see DESIGN.md 3.1).
    const bool to_part = pPart != nullptr;
    float* const eps = smem + wave * (32 * 68);
    const int erow = lane >> 4, ec4 = (lane & 15) * 4;
    const int ncol = n0 + wn * 64 + ec4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            eps[lr * 68 + l31] = acc[i][0][r];
            eps[lr * 68 + 32 + l31] = acc[i][1][r];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = t * 4 + erow;
            f32x4 v = *(const f32x4*)(eps + row * 68 + ec4);
            const int m = m0 + wm * 64 + i * 32 + row;
            if (m >= p.M || ncol >= p.Ng) continue;
            float* dst;
            long eoff;              // element offset of the 4 outputs in the output tensor (C or the split-K slabs)
            if (to_part) {
                const long srow = (MODE == MODE_DGRAD_S2) ? ((long)split * 4 + parity) * p.M + m
                                                          : (long)split * p.M + m;
                dst = pPart;
                eoff = srow * p.Ng + ncol;
            } else if (MODE == MODE_DGRAD_S2) {
                const int b = m & (Wo - 1), a = (m >> lgWo) & (Ho - 1), n = m >> lgHW;
                dst = pC;
                eoff = (long)((n * H + 2 * a + ph) * W + 2 * b + pw) * Cc + ncol;
            } else {
                dst = pC;
                eoff = (long)m * p.Ng + ncol;
            }
            if (!to_part && p.accumulate) v += *(const f32x4*)(dst + eoff);
            if (!to_part) {
                if (MODE != MODE_FWD_C3 && MODE != MODE_WGRAD && p.bias != nullptr)
                    v += *(const f32x4*)(p.bias + (MODE == MODE_DGRAD_PLAIN ? ncol % Cc : ncol));
                if (MODE != MODE_WGRAD && p.act != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = dg_apply_act(v[e], p.act, p.slope);
                }
            }
            dg_store_out4(dst, eoff, v, (!to_part && MODE != MODE_WGRAD && MODE != MODE_FWD_C3) ? p.out16 : 0);
        }
    }
    if (stp) {
        stp[4] = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stp[5] = wall_clock64();
        stp[7] = clock64();
    }
}

// Sum split-K slabs in a fixed order and scatter to the real output.
// Grouped launches: blockIdx.z = OUTPUT problem z; with p.share = s > 1 (weight gradients of passes that share the weights) it adds the
// slab sums of problems z*s .. z*s + s - 1 to problem z*s's output one after the other -- the value `s` accumulating launches leave.
template <int MODE>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmArgs p, long total4) {
    const int ng4 = p.Ng >> 2;
    const long slab = (long)(MODE == MODE_DGRAD_S2 ? 4 : 1) * p.M * p.Ng;
    const int share = p.share > 1 ? p.share : 1;
    const int g0 = blockIdx.z * share;
    float* const pC = dg_group_ptr(p.C, p.gdC, g0);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const long row = idx / ng4;
        const int c4 = (int)(idx - row * ng4);
        long eoff;
        if (MODE == MODE_DGRAD_S2) {
            const int parity = (int)(row / p.M);
            const int m = (int)(row - (long)parity * p.M);
            const int b = m & (p.Wo - 1), a = (m >> p.lgWo) & (p.Ho - 1), n = m >> (p.lgWo + p.lgHo);
            eoff = (long)((n * p.H + 2 * a + (parity >> 1)) * p.W + 2 * b + (parity & 1)) * p.Cc + c4 * 4;
        } else {
            eoff = row * p.Ng + c4 * 4;
        }
        f32x4 out = {0.f, 0.f, 0.f, 0.f};
        bool have = false;
        if (p.accumulate) { out = *(const f32x4*)(pC + eoff); have = true; }
        for (int j = 0; j < share; ++j) {
            const float* src = dg_group_ptr(p.part, p.gdPart, g0 + j) + row * p.Ng + c4 * 4;
            f32x4 s = *(const f32x4*)src;
            int k = 1;
            for (; k + 7 < p.splits; k += 8) {      // 8 slab loads in flight, summed in slab order (fixed -> deterministic)
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(src + (k + u) * slab);
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; k < p.splits; ++k) s += *(const f32x4*)(src + k * slab);
            if (have) s += out;
            out = s;
            have = true;
        }
        f32x4 s = out;
        if (p.bias != nullptr) s += *(const f32x4*)(p.bias + (c4 * 4) % p.bias_mod);
        if (p.act != 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = dg_apply_act(s[e], p.act, p.slope);
        }
        dg_store_out4(pC, eoff, s, p.out16);
    }
}

// Small outputs (heads, deep weight gradients): the serial form above would leave a few dozen workgroups walking
// up to 64 slabs each.  Here a block is 16 float4 outputs x 16 slab lanes; lane j sums slabs j, j+16, ... and the
// 16 lane sums are added in lane order by one thread (fixed order -> deterministic).  blockIdx.z / p.share as above.
__global__ __launch_bounds__(256) void splitk_reduce_small_kernel(const IgemmArgs p, long total4) {
    __shared__ f32x4 red[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const long idx = (long)blockIdx.x * 16 + tx;
    const int ng4 = p.Ng >> 2;
    const long slab = (long)p.M * p.Ng;
    const int share = p.share > 1 ? p.share : 1;
    const int g0 = blockIdx.z * share;
    float* const pC = dg_group_ptr(p.C, p.gdC, g0);
    long row = 0;
    int c4 = 0;
    if (idx < total4) {
        row = idx / ng4;
        c4 = (int)(idx - row * ng4);
    }
    const long eoff = row * p.Ng + c4 * 4;
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    bool have = false;
    if (ty == 0 && idx < total4 && p.accumulate) { out = *(const f32x4*)(pC + eoff); have = true; }
    for (int j = 0; j < share; ++j) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (idx < total4) {
            const float* src = dg_group_ptr(p.part, p.gdPart, g0 + j) + row * p.Ng + c4 * 4;
            for (int k = ty; k < p.splits; k += 16) s += *(const f32x4*)(src + k * slab);
        }
        if (j > 0) __syncthreads();
        red[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && idx < total4) {
#pragma unroll
            for (int u = 1; u < 16; ++u) s += red[u][tx];
            if (have) s += out;
            out = s;
            have = true;
        }
    }
    if (ty == 0 && idx < total4) {
        f32x4 s = out;
        if (p.bias != nullptr) s += *(const f32x4*)(p.bias + (c4 * 4) % p.bias_mod);
        if (p.act != 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = dg_apply_act(s[e], p.act, p.slope);
        }
        dg_store_out4(pC, eoff, s, p.out16);
    }
}

// Split-K reduction that also emits the BatchNorm partial rows: block = 32 float4 columns x 8 row lanes,
// grid = (column chunks of 128, row chunks); one partial row per row chunk.
template <int MODE>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const IgemmArgs p, int rchunks) {
    __shared__ f32x4 red[2][8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int grp = blockIdx.z;                         // grouped launch: problem index
    float* const pC = dg_group_ptr(p.C, p.gdC, grp);
    const float* const pPart = dg_group_ptr(p.part, p.gdPart, grp);
    float* const pStat = dg_group_ptr(p.stat, p.gdStat, grp);
    const int c = (blockIdx.x * 32 + tx) * 4;
    const long R = (long)(MODE == MODE_DGRAD_S2 ? 4 : 1) * p.M;
    const long slab = R * p.Ng;
    const long rows_per = (R + rchunks - 1) / rchunks;
    const long r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    const bool cok = c < p.Ng;
    if (cok && r0 < r1) {
        const float* src = pPart + r0 * p.Ng + c;
        sh = *(const f32x4*)src;
        for (int k = 1; k < p.splits; ++k) sh += *(const f32x4*)(src + k * slab);
        for (long row = r0 + ty; row < r1; row += 8) {
            const float* sp = pPart + row * p.Ng + c;
            f32x4 v = *(const f32x4*)sp;
            for (int k = 1; k < p.splits; ++k) v += *(const f32x4*)(sp + k * slab);
            long eoff;
            if (MODE == MODE_DGRAD_S2) {
                const int parity = (int)(row / p.M);
                const int m = (int)(row - (long)parity * p.M);
                const int b = m & (p.Wo - 1), a = (m >> p.lgWo) & (p.Ho - 1), n = m >> (p.lgWo + p.lgHo);
                eoff = (long)((n * p.H + 2 * a + (parity >> 1)) * p.W + 2 * b + (parity & 1)) * p.Cc + c;
            } else {
                eoff = row * p.Ng + c;
            }
            dg_store_out4(pC, eoff, v, p.out16);      // statistics from the fp32 sums, whatever the output is rounded to
            const f32x4 d = v - sh;
            s += d;
            q += d * d;
        }
    }
    red[0][ty][tx] = s;
    red[1][ty][tx] = q;
    __syncthreads();
    if (ty == 0) {
        float* srow = pStat + (long)blockIdx.y * p.stat_rs;
        if (blockIdx.x == 0 && tx == 0) srow[0] = (float)max(0L, r1 - r0);
        if (cok && r0 < r1) {
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                s += red[0][j][tx];
                q += red[1][j][tx];
            }
            *(f32x4*)(srow + 4 + c) = sh;
            *(f32x4*)(srow + 4 + p.Ng + c) = s;
            *(f32x4*)(srow + 4 + 2 * p.Ng + c) = q;
        }
    }
}

// ---- K == 1 head (Discriminator conv8, model.py:35): plain reductions ---------------------------------
// XT = float or __bf16: the element type of the activation tensors x / dx (weights, dy and dw stay fp32)
__device__ __forceinline__ f32x4 dg_ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 dg_ld4(const __bf16* p) { return __builtin_convertvector(*(const dg_bf16x4*)p, f32x4); }
__device__ __forceinline__ void dg_st4(float* p, const f32x4& v) { *(f32x4*)p = v; }
__device__ __forceinline__ void dg_st4(__bf16* p, const f32x4& v) { *(dg_bf16x4*)p = __builtin_convertvector(v, dg_bf16x4); }
// grouped launches: blockIdx.y = problem (DgPtrs: one tensor per problem)
template <typename XT>
__global__ __launch_bounds__(256) void head1_fwd_kernel(const DgPtrs xs, const DgPtrs ws, const DgPtrs ys, int J) {
    __shared__ float red[4];
    const XT* __restrict__ x = dg_pick<const XT>(xs, blockIdx.y);
    const float* __restrict__ w = dg_pick<const float>(ws, blockIdx.y);
    float* __restrict__ y = dg_pick<float>(ys, blockIdx.y);
    const XT* xr = x + (long)blockIdx.x * J;
    float s = 0.f;
    for (int j = threadIdx.x * 4; j < J; j += 1024) {
        const f32x4 a = dg_ld4(xr + j), b = *(const f32x4*)(w + j);
        s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    }
    s = dg_block_sum256(s, red);
    if (threadIdx.x == 0) y[blockIdx.x] = s;
}
template <typename XT>
__global__ __launch_bounds__(256) void head1_dgrad_kernel(const DgPtrs dys, const DgPtrs ws, const DgPtrs dxs, int N, int J) {
    const float* __restrict__ dy = dg_pick<const float>(dys, blockIdx.y);
    const float* __restrict__ w = dg_pick<const float>(ws, blockIdx.y);
    XT* __restrict__ dx = dg_pick<XT>(dxs, blockIdx.y);
    const long total4 = (long)N * (J >> 2);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int n = (int)(idx / (J >> 2));
        const int j4 = (int)(idx - (long)n * (J >> 2));
        const float g = dy[n];
        const f32x4 b = *(const f32x4*)(w + j4 * 4);
        dg_st4(dx + (long)n * J + j4 * 4, b * g);
    }
}
// 16 float4 lanes along J x 16 batch lanes; each batch lane walks n = lane, lane+16, ... (4 loads in flight),
// fixed-order tree over the batch lanes through LDS (deterministic).
template <typename XT>
__global__ __launch_bounds__(256) void head1_wgrad_kernel(const DgPtrs dys, const DgPtrs xs, const DgPtrs dws, int N, int J, int accumulate) {
    __shared__ f32x4 red[256];
    const float* __restrict__ dy = dg_pick<const float>(dys, blockIdx.y);
    const XT* __restrict__ x = dg_pick<const XT>(xs, blockIdx.y);
    float* __restrict__ dw = dg_pick<float>(dws, blockIdx.y);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int j4 = blockIdx.x * 16 + tx;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (j4 * 4 < J) {
        const XT* px = x + j4 * 4;
        int n = ty;
        for (; n + 48 < N; n += 64) {
            f32x4 v[4];
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = dg_ld4(px + (long)(n + 16 * u) * J);
                d[u] = dy[n + 16 * u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u] * d[u];
        }
        for (; n < N; n += 16) s += dg_ld4(px + (long)n * J) * dy[n];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 8; h > 0; h >>= 1) {
        if (ty < h) red[threadIdx.x] += red[threadIdx.x + h * 16];
        __syncthreads();
    }
    if (ty == 0 && j4 * 4 < J) {
        f32x4 r = red[tx];
        if (accumulate) r += *(const f32x4*)(dw + j4 * 4);
        *(f32x4*)(dw + j4 * 4) = r;
    }
}

// ---- host side ----------------------------------------------------------------------------------------
struct ConvGeom {
    int N, H, W, C, K, stride, pad, Ho, Wo;
};

static int check_geom(const char* who, int N, int H, int W, int C, int K, int stride, int pad, ConvGeom* g) {
    DG_CHECK_ARG(N >= 1 && C >= 4 && K >= 1, "%s: bad N/C/K (%d,%d,%d)", who, N, C, K);
    DG_CHECK_ARG(C % 4 == 0, "%s: C=%d must be a multiple of 4", who, C);
    DG_CHECK_ARG(K == 1 || K % 4 == 0, "%s: K=%d must be 1 or a multiple of 4", who, K);
    g->N = N; g->H = H; g->W = W; g->C = C; g->K = K; g->stride = stride; g->pad = pad;
    if (stride == 2 && pad == 1) {
        DG_CHECK_ARG(dg_is_pow2(H) && dg_is_pow2(W) && H >= 2 && W >= 2, "%s: H,W must be powers of two >= 2 (%d,%d)", who, H, W);
        g->Ho = H / 2; g->Wo = W / 2;
    } else if (stride == 1 && pad == 0) {
        DG_CHECK_ARG(H == 4 && W == 4, "%s: stride-1 pad-0 head needs a 4x4 input (%d,%d)", who, H, W);
        g->Ho = 1; g->Wo = 1;
    } else {
        return dg_fail(DG_ERR_INVALID, "%s: unsupported (stride,pad)=(%d,%d)", who, stride, pad);
    }
    DG_CHECK_ARG((long)N * H * W < (1L << 31) && (long)N * H * W * (long)C / 4 < (1L << 31), "%s: tensor too large", who);
    return DG_OK;
}

struct Plan {
    int mode, wm, wn, kt;
    IgemmArgs a;
    size_t ws_bytes;
    int stat_rows;   // partial rows the fused BatchNorm statistics would produce (0: not supported)
    int dma;         // 1: igemm_dma.hip (both operands bf16 in HBM, LDS-DMA staging, 256 x 256 tile, 512 threads)
                     // 2: igemm_dma_x3.hip (both operands as three bf16 planes in HBM; same tile, K-tile 16)
                     // 3: igemm_dma_x3_dgw.hip (plane operands, input-grad with few output channels: `ncls` parity classes per
                     //    workgroup, gradient window in LDS)
                     // 4: igemm_dma_dgw.hip (bf16 operands, the same input-grad scheme with a 32-deep K-step)
    int ncls;
};

int dg_igemm_dma_launch(int mode, const IgemmArgs& a, int zmul, hipStream_t st);      // igemm_dma.hip
int dg_igemm_dma_x3_launch(int mode, int wm, int wn, const IgemmArgs& a, int zmul, hipStream_t st);   // igemm_dma_x3.hip
int dg_igemm_x3_dgw_launch(int ncls, const IgemmArgs& a, hipStream_t st);                              // igemm_dma_x3_dgw.hip
int dg_igemm_x3_fww_launch(const IgemmArgs& a, hipStream_t st);                                        // igemm_dma_x3_fww.hip
int dg_igemm_bf16_dgw_launch(int ncls, const IgemmArgs& a, hipStream_t st);                            // igemm_dma_dgw.hip

static int reduce_stats_rchunks(long R, int Ng) {
    const int cch = (Ng + 127) / 128;
    long rc = 1024 / cch;
    if (rc > 256) rc = 256;
    const long maxrc = (R + 7) / 8;
    if (rc > maxrc) rc = maxrc;
    if (rc < 1) rc = 1;
    return (int)rc;
}

static int choose_splits(int base_wgs, int nIt, int dflt_target = 0) {
    int forced = dg_get_option(DG_OPT_SPLITK);
    int target = dg_get_option(DG_OPT_TARGET_WGS);
    if (target <= 0) target = dflt_target > 0 ? dflt_target : 512;
    // Split K only when the tile grid cannot even give every CU one workgroup: at >= 256 workgroups the
    // split's extra slab traffic + reduction kernel cost more than the second co-resident workgroup buys
    // (measured on the 64 px layers: 0.157 ms unsplit vs 0.166 ms split-by-2 for M=16384, N=256, K=2048).
    int min_full = dg_get_option(DG_OPT_SPLIT_BELOW);
    if (min_full <= 0) min_full = 256;
    int s = 1;
    if (forced > 0) s = forced;
    else if (base_wgs < min_full) s = (target + base_wgs - 1) / base_wgs;
    const int min_it = 4;  // keep at least a few K-tiles per split
    if (s > nIt / min_it) s = nIt / min_it;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return s;
}

// op: 0 fwd, 1 dgrad, 2 wgrad;  a16 / b16: that operand is a bf16 tensor (only honoured on the bf16 tile path)
static void make_plan(int op, const ConvGeom& g, Plan* pl, int a16 = 0, int b16 = 0, int allow_fww = 1) {
    IgemmArgs& a = pl->a;
    a = IgemmArgs();
    a.N = g.N; a.H = g.H; a.W = g.W; a.Cc = g.C; a.K = g.K;
    a.Ho = g.Ho; a.Wo = g.Wo; a.lgHo = dg_ilog2(g.Ho); a.lgWo = dg_ilog2(g.Wo);
    a.stride = g.stride; a.pad = g.pad;
    const int npix = g.N * g.Ho * g.Wo;
    int kt_opt = dg_get_option(DG_OPT_KT);
    // K-tile of the exact-fp32 kernels: 32, except the weight gradient, which runs 1.7 % faster on 16 at 512 px / batch 32
    // (41 KB of LDS per workgroup; same-box A/B per layer: 1.017 -> 1.003, 1.015 -> 0.989 ms; the forward loses 3 % on 16);
    // option "kt" 16 | 32 forces one value everywhere
    pl->wm = 2; pl->wn = 2; pl->kt = (kt_opt == 16 || (kt_opt == 0 && op == 2)) ? 16 : 32;
    // bf16 MFMA operands (option "bf16"): bf16 LDS tiles, K-tile 64 (32 with the 256x64 tile, whose LDS would
    // otherwise allow one workgroup per CU only); needs the buffer-descriptor kernels and whole K-tiles per tap
    const int popt = dg_cur_prec();      // this call's arithmetic: 0 exact fp32 MFMA, 1 bf16 operands, 2 fp32 as three bf16 planes
    bool want_bf16 = popt == 1;
    const bool want_x3 = popt == 2;
    if (want_bf16 && op == 0 && g.C % 64 != 0) want_bf16 = false;
    if (want_bf16 && op == 1 && g.stride == 2 && g.K % 64 != 0) want_bf16 = false;
    if (want_bf16) pl->kt = 64;
    if (want_x3) pl->kt = 16;
    int zmul = 1;
    if (op == 0) {
        pl->mode = MODE_FWD;
        a.M = npix; a.Ng = g.K; a.R = 16 * g.C;
        a.nIt = 16 * g.C / pl->kt;
    } else if (op == 1 && g.stride == 2) {
        pl->mode = MODE_DGRAD_S2;
        a.M = npix; a.Ng = g.C; a.R = 4 * g.K;
        a.nIt = 4 * g.K / pl->kt;
        zmul = 4;
        // 256x64 tile for <= 64 output columns, KT 16 so that two workgroups fit a CU (KT 32 = 91 KB of LDS, one
        // workgroup per CU, was measured slower: 189 vs 165 us on 128->64 @32x32, nothing hides prologue/epilogue)
        if (g.C <= 64) { pl->wm = 4; pl->wn = 1; pl->kt = want_bf16 ? 32 : 16; a.nIt = 4 * g.K / pl->kt; }
    } else if (op == 1) {
        pl->mode = MODE_DGRAD_PLAIN;
        a.M = g.N; a.Ng = 16 * g.C; a.R = g.K;
        a.nIt = (g.K + pl->kt - 1) / pl->kt;
    } else {
        pl->mode = MODE_WGRAD;
        a.M = g.K; a.Ng = 16 * g.C; a.R = npix;
        a.nIt = (npix + pl->kt - 1) / pl->kt;
    }
    {   // operand sizes for the buffer-descriptor kernels; 0 = use the 64-bit pointer kernels
        const long xb = (long)g.N * g.H * g.W * g.C * 4, yb = (long)npix * g.K * 4, wb = (long)g.K * 16 * g.C * 4;
        // a16 / b16 = 3: three bf16 planes; the descriptors cover ONE plane (the planes' distance is set by the caller)
        const long ab = (op == 0 ? xb : yb) / (a16 ? 2 : 1), bb = (op == 2 ? xb : wb) / (b16 ? 2 : 1);
        const bool fits = ab < (1L << 31) && bb < (1L << 31) && dg_get_option(DG_OPT_POINTER_PATH) == 0;
        a.abytes = fits ? (unsigned)ab : 0u;
        a.bbytes = fits ? (unsigned)bb : 0u;
        a.a16 = a16; a.b16 = b16;
        a.prec = !fits ? 0 : (want_bf16 ? 1 : (want_x3 ? 2 : 0));
        a.dbg_zero = dg_get_option(DG_OPT_DBG_ZERO);
        if ((want_bf16 || want_x3) && !fits) {            // >= 2 GiB operands: fp32 pointer kernels with their own K-tile
            pl->kt = (pl->mode == MODE_DGRAD_S2 && pl->wm == 4) ? 16 : 32;
            if (pl->mode == MODE_FWD) a.nIt = 16 * g.C / pl->kt;
            else if (pl->mode == MODE_DGRAD_S2) a.nIt = 4 * g.K / pl->kt;
            else if (pl->mode == MODE_DGRAD_PLAIN) a.nIt = (g.K + pl->kt - 1) / pl->kt;
            else a.nIt = (npix + pl->kt - 1) / pl->kt;
        }
    }
    // LDS-DMA kernel (igemm_dma.hip): both operands are bf16 tensors, the 256-wide tile is at least 3/4 used in both
    // directions, whole 64-deep K-tiles per tap.  One 512-thread workgroup per CU: the split grid aims for 256 workgroups.
    pl->dma = 0;
    if (a.prec == 1 && a16 && b16 && dg_get_option(DG_OPT_NO_DMA) == 0 && pl->kt == 64 && a.Ng >= 192 && a.M >= 192 &&
        g.C % 8 == 0 && g.K % 8 == 0 &&
        ((pl->mode == MODE_FWD && g.C % 64 == 0) || (pl->mode == MODE_DGRAD_S2 && g.K % 64 == 0) || pl->mode == MODE_WGRAD))
        pl->dma = 1;
    // plane kernel (igemm_dma_x3.hip): the same tile and grid rules with 16-deep K-tiles
    // bf16 operands, input-grad with <= 128 output channels: the window kernel (igemm_dma_dgw.hip; see the plane form below)
    if (a.prec == 1 && a16 == 1 && b16 == 1 && dg_get_option(DG_OPT_NO_DMA) == 0 && pl->mode == MODE_DGRAD_S2 && g.C <= 128 &&
        g.C % 8 == 0 && g.K % 32 == 0 && g.Wo >= 32 && g.Wo <= 128 && g.Ho * g.Wo >= 256 && dg_get_option(DG_OPT_DMA_MFMA) != 1) {
        pl->dma = 4;
        pl->ncls = g.C <= 64 ? 4 : 2;
        a.nIt = g.K / 32;                           // the split unit is a 32-channel chunk (4 tap steps)
    }
    // wm x wn waves of 128 x 64: the 256 x 256 tile (8 waves, one workgroup per CU); 128 x 256 for a weight gradient of 96..191
    // rows (4 waves, two workgroups per CU: 184 -> 218 TFLOP/s on 64 -> 128 channels).  A 256 x 128 tile for 96..191 COLUMNS was
    // built and measured slower than the register-staged tiles (109 vs 146 forward, 102 vs 188 TFLOP/s input-grad at 512 px):
    // with half the columns the k-contiguous activation operand is fetched twice per FLOP in 32-byte pieces, HBM-bound.
    if (a.prec == 2 && a16 == 3 && b16 == 3 && dg_get_option(DG_OPT_NO_DMA) == 0 && pl->kt == 16 && g.C % 8 == 0 && g.K % 8 == 0 &&
        ((pl->mode == MODE_FWD && g.C % 16 == 0) || (pl->mode == MODE_DGRAD_S2 && g.K % 16 == 0) || pl->mode == MODE_WGRAD)) {
        if (a.Ng >= 192 && a.M >= 192) { pl->dma = 2; pl->wm = 2; pl->wn = 4; }
        else if (pl->mode == MODE_WGRAD && a.M >= 96 && a.Ng >= 192) { pl->dma = 2; pl->wm = 1; pl->wn = 4; }
#ifdef DG_TIMING_KNOBS
        // timing experiment (dbg_zero bit 5): the 256 x 128 tile for a forward with 96..191 output channels
        else if ((dg_get_option(DG_OPT_DBG_ZERO) & 32) && pl->mode == MODE_FWD && g.stride == 2 && a.Ng >= 96 && a.M >= 192 && g.C % 64 == 0) { pl->dma = 2; pl->wm = 2; pl->wn = 2; }
#endif
        // input-grad with <= 128 output channels: all parity classes of a 256-pixel tile in one workgroup, the gradient window in
        // LDS (igemm_dma_x3_dgw.hip); whole image rows per tile, a 32-pixel block inside one row
        else if (pl->mode == MODE_DGRAD_S2 && g.C <= 128 && g.Wo >= 32 && g.Wo <= 128 && g.Ho * g.Wo >= 256 &&
                 dg_get_option(DG_OPT_DMA_MFMA) != 1) {
            pl->dma = 3;
            pl->ncls = g.C <= 64 ? 4 : 2;
            a.nIt = g.K / 16;                       // the split unit is a 16-channel chunk (4 tap steps)
        }
        // forward with <= 128 output channels: the input window of one (chunk, parity class) in LDS, re-used by the class's four taps
        // (igemm_dma_x3_fww.hip); whole output rows per 256-pixel tile, transposed weight planes; no split-K
#ifdef DG_EXPERIMENTS      // the window forward kernel: not faster than the register-staged tiles, experiments library only
        else if (allow_fww && pl->mode == MODE_FWD && g.stride == 2 && g.pad == 1 && g.K <= 128 && g.K % 8 == 0 && g.C % 16 == 0 && g.Wo >= 32 &&
                 g.Wo <= 128 && (g.Ho * g.Wo) % 256 == 0 && dg_get_option(DG_OPT_DMA_MFMA) != 1) {
            pl->dma = 5;
            pl->ncls = 4;
        }
#endif
    }
    const int BM = pl->dma == 2 ? 128 * pl->wm : (pl->dma ? 256 : 64 * pl->wm);
    const int BN = pl->dma == 2 ? 64 * pl->wn : (pl->dma >= 3 ? a.Ng : (pl->dma ? 256 : 64 * pl->wn));
    a.tilesM = (a.M + BM - 1) / BM;
    a.tilesN = (a.Ng + BN - 1) / BN;
    // workgroups of the launch before any K split; a grouped launch planned as a whole (plan_groups of the *_g entry points) counts all
    // its problems: the chip is filled by the group, each problem needs fewer splits (fewer slabs written and re-read)
    const int base = a.tilesM * a.tilesN * (pl->dma >= 3 ? 4 / pl->ncls : zmul) * (pl->dma ? 1 : dg_cur_plan_groups());
    a.splits = pl->dma == 5 ? 1 : choose_splits(base, a.nIt, pl->dma == 2 && pl->wm * pl->wn == 4 ? 512 : (pl->dma ? 256 : 0));
    a.itPerSplit = (a.nIt + a.splits - 1) / a.splits;
    a.splits = (a.nIt + a.itPerSplit - 1) / a.itPerSplit;  // no empty split
    // measured (PMC, MB per launch beyond L2, 64 px layers): forward 83 -> 76, weight-grad 222 -> 90, input-grad with
    // the 256x64 tile 135 -> 104, input-grad with 128x128 tiles 64 -> 76 (worse: left in plain order)
    const int xopt = dg_get_option(DG_OPT_RESERVED);       // "no_xcd_group": 1 = plain order everywhere, 3 = only mode 3 below off
    a.xcd_group = ((a.tilesM * a.splits) % 8 == 0 && (xopt == 0 || xopt == 3) &&
                   !(pl->mode == MODE_DGRAD_S2 && pl->wm != 4)) ? 1 : 0;   // option "no_xcd_group" switches it off
    {   // weight-dominated layers: share the B operand instead (mode 2, see the kernel)
        const long a_bytes = (long)a.M * (pl->mode == MODE_FWD ? 16L * g.C : (pl->mode == MODE_DGRAD_S2 ? 4L * g.K : g.K)) * 4;
        const long b_bytes = (long)g.K * 16 * g.C * 4 / (pl->mode == MODE_DGRAD_S2 ? 4 : 1);
        if ((pl->mode == MODE_FWD || pl->mode == MODE_DGRAD_S2) && (xopt == 0 || xopt == 3) && b_bytes > a_bytes &&
            a.tilesM > 1 && a.tilesM <= 64 && (a.tilesN * zmul * a.splits) % 8 == 0)
            a.xcd_group = 2;
    }
    if (pl->mode == MODE_FWD && pl->dma == 0 && a.tilesN == 1 && (a.tilesM & 7) == 0 && g.stride == 2 && a.xcd_group == 1 && xopt != 3)
        a.xcd_group = 3;              // (option "no_xcd_group" 3: keep the round-robin order -- same-box A/B)
    pl->ws_bytes = a.splits > 1 ? (size_t)a.splits * zmul * a.M * a.Ng * sizeof(float) : 0;
    pl->stat_rows = 0;
    if ((pl->mode == MODE_FWD && g.stride == 2) || pl->mode == MODE_DGRAD_S2)
        pl->stat_rows = a.splits > 1 ? reduce_stats_rchunks((long)zmul * a.M, a.Ng)
                                     : (pl->dma == 3 ? 4 * a.tilesM * 2 : zmul * a.tilesM * pl->wm);   // window kernel: (class, tile, wave row)
    if (pl->dma == 4 && a.splits <= 1) pl->stat_rows = 4 * a.tilesM * 2;      // bf16 window input-grad kernel: (class, tile, wave row) like dma 3
    if (pl->dma == 5) pl->stat_rows = 0;                                      // the window forward kernel emits none
}

template <int MODE, int WM, int WN, int KT>
static void launch_igemm(const IgemmArgs& a, int zmul, hipStream_t st) {
    const dim3 grid(a.tilesM * a.tilesN * zmul * a.splits, 1, a.groups > 1 ? a.groups : 1);     // z: problem of a grouped launch
    if (MODE != MODE_FWD_C3 && a.abytes != 0 && a.bbytes != 0)
        hipLaunchKernelGGL((igemm_kernel<MODE, WM, WN, KT, MODE != MODE_FWD_C3, 0>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((igemm_kernel<MODE, WM, WN, KT, false, 0>), grid, dim3(256), 0, st, a);
}
// bf16 operand tiles (PREC 1) / fp32 as three bf16 planes (PREC 2): buffer-descriptor kernels only
template <int MODE, int WM, int WN, int KT, int PREC>
static void launch_igemm_bf16(const IgemmArgs& a, int zmul, hipStream_t st) {
    const dim3 grid(a.tilesM * a.tilesN * zmul * a.splits, 1, a.groups > 1 ? a.groups : 1);     // z: problem of a grouped launch
    if constexpr (PREC == 1) {
        // bf16 shadow operands: weights (B of forward / input-grad) alone, or both operands
        if (a.a16 && a.b16) { hipLaunchKernelGGL((igemm_kernel<MODE, WM, WN, KT, true, 1, true, true>), grid, dim3(256), 0, st, a); return; }
        if (!a.a16 && a.b16) { hipLaunchKernelGGL((igemm_kernel<MODE, WM, WN, KT, true, 1, false, true>), grid, dim3(256), 0, st, a); return; }
        if (a.a16 && !a.b16) { hipLaunchKernelGGL((igemm_kernel<MODE, WM, WN, KT, true, 1, true, false>), grid, dim3(256), 0, st, a); return; }
    }
#ifdef DG_EXPERIMENTS      // the register-staged plane reader: measured 45 % slower, experiments library only
    if constexpr (PREC == 2 && MODE == MODE_FWD && WM == 2 && WN == 2) {
        // both operands as plane triples (a16 = b16 = 3): the loader copies the planes, no split in the kernel
        if (a.a16 == 3 && a.b16 == 3) { hipLaunchKernelGGL((igemm_kernel<MODE, WM, WN, KT, true, 2, true, true>), grid, dim3(256), 0, st, a); return; }
    }
#endif
    hipLaunchKernelGGL((igemm_kernel<MODE, WM, WN, KT, true, PREC>), grid, dim3(256), 0, st, a);
}

static long long* g_stamp_buf = nullptr;
static size_t g_stamp_cap = 0;
// profiling hook: stamps of the NEXT igemm launches go to buf (8 int64 per workgroup); (nullptr, 0) = off
extern "C" int dg_debug_igemm_stamps(void* buf, size_t bytes) {
    g_stamp_buf = (long long*)buf;
    g_stamp_cap = bytes;
    return DG_OK;
}

static int run_plan(const char* who, Plan& pl, void* ws, size_t ws_bytes, hipStream_t st) {
    IgemmArgs& a = pl.a;
    {
        const size_t grid = (size_t)a.tilesM * a.tilesN * (pl.mode == MODE_DGRAD_S2 ? 4 : 1) * a.splits;
        a.stamps = (g_stamp_buf != nullptr && g_stamp_cap >= grid * 64) ? g_stamp_buf : nullptr;
    }
    if (a.splits > 1) {
        if (ws == nullptr || ws_bytes < pl.ws_bytes)
            return dg_fail(DG_ERR_WORKSPACE, "%s: workspace %zu < required %zu", who, ws_bytes, pl.ws_bytes);
        a.part = (float*)ws;
    }
    const int zmul = pl.mode == MODE_DGRAD_S2 ? 4 : 1;
    const int ngrp = a.groups > 1 ? a.groups : 1;
    const int nout = ngrp / (a.share > 1 ? a.share : 1);      // outputs of the split-K reduction (share: problems summed into one tensor)
    if (pl.dma && ngrp > 1) return dg_fail(DG_ERR_INVALID, "%s: the LDS-DMA kernels take no grouped launch", who);
    if (pl.dma) {
        const int ok =
#ifdef DG_EXPERIMENTS
                       pl.dma == 5 ? dg_igemm_x3_fww_launch(a, st) :
#endif
                       pl.dma == 4 ? dg_igemm_bf16_dgw_launch(pl.ncls, a, st)
                     : pl.dma == 3 ? dg_igemm_x3_dgw_launch(pl.ncls, a, st)
                     : pl.dma == 2 ? dg_igemm_dma_x3_launch(pl.mode, pl.wm, pl.wn, a, zmul, st) : dg_igemm_dma_launch(pl.mode, a, zmul, st);
        if (!ok) return dg_fail(DG_ERR_INVALID, "%s: no LDS-DMA kernel for mode %d", who, pl.mode);
        DG_CHECK_LAUNCH(who);
    } else {
    const int key = a.prec == 1 ? 1000 + pl.mode * 100 + pl.wm * 10 + (pl.kt == 64 ? 1 : 0)
                  : a.prec == 2 ? 2000 + pl.mode * 100 + pl.wm * 10
                                : pl.mode * 100 + pl.wm * 10 + (pl.kt == 32 ? 1 : 0);
    switch (key) {
        case 1000 + MODE_FWD * 100 + 21: launch_igemm_bf16<MODE_FWD, 2, 2, 64, 1>(a, zmul, st); break;
        case 1000 + MODE_DGRAD_S2 * 100 + 21: launch_igemm_bf16<MODE_DGRAD_S2, 2, 2, 64, 1>(a, zmul, st); break;
        case 1000 + MODE_DGRAD_S2 * 100 + 40: launch_igemm_bf16<MODE_DGRAD_S2, 4, 1, 32, 1>(a, zmul, st); break;
        case 1000 + MODE_DGRAD_PLAIN * 100 + 21: launch_igemm_bf16<MODE_DGRAD_PLAIN, 2, 2, 64, 1>(a, zmul, st); break;
        case 1000 + MODE_WGRAD * 100 + 21: launch_igemm_bf16<MODE_WGRAD, 2, 2, 64, 1>(a, zmul, st); break;
        case 2000 + MODE_FWD * 100 + 20: launch_igemm_bf16<MODE_FWD, 2, 2, 16, 2>(a, zmul, st); break;
        case 2000 + MODE_DGRAD_S2 * 100 + 20: launch_igemm_bf16<MODE_DGRAD_S2, 2, 2, 16, 2>(a, zmul, st); break;
        case 2000 + MODE_DGRAD_S2 * 100 + 40: launch_igemm_bf16<MODE_DGRAD_S2, 4, 1, 16, 2>(a, zmul, st); break;
        case 2000 + MODE_DGRAD_PLAIN * 100 + 20: launch_igemm_bf16<MODE_DGRAD_PLAIN, 2, 2, 16, 2>(a, zmul, st); break;
        case 2000 + MODE_WGRAD * 100 + 20: launch_igemm_bf16<MODE_WGRAD, 2, 2, 16, 2>(a, zmul, st); break;
        case MODE_FWD * 100 + 21: launch_igemm<MODE_FWD, 2, 2, 32>(a, zmul, st); break;
        case MODE_FWD * 100 + 20: launch_igemm<MODE_FWD, 2, 2, 16>(a, zmul, st); break;
        case MODE_DGRAD_S2 * 100 + 21: launch_igemm<MODE_DGRAD_S2, 2, 2, 32>(a, zmul, st); break;
        case MODE_DGRAD_S2 * 100 + 20: launch_igemm<MODE_DGRAD_S2, 2, 2, 16>(a, zmul, st); break;
        case MODE_DGRAD_S2 * 100 + 40: launch_igemm<MODE_DGRAD_S2, 4, 1, 16>(a, zmul, st); break;
        case MODE_DGRAD_PLAIN * 100 + 21: launch_igemm<MODE_DGRAD_PLAIN, 2, 2, 32>(a, zmul, st); break;
        case MODE_DGRAD_PLAIN * 100 + 20: launch_igemm<MODE_DGRAD_PLAIN, 2, 2, 16>(a, zmul, st); break;
        case MODE_WGRAD * 100 + 21: launch_igemm<MODE_WGRAD, 2, 2, 32>(a, zmul, st); break;
        case MODE_WGRAD * 100 + 20: launch_igemm<MODE_WGRAD, 2, 2, 16>(a, zmul, st); break;
        default: return dg_fail(DG_ERR_INVALID, "%s: no kernel for mode %d wm %d kt %d", who, pl.mode, pl.wm, pl.kt);
    }
    DG_CHECK_LAUNCH(who);
    }
    if (a.splits > 1 && a.stat != nullptr) {
        const int rc = pl.stat_rows;
        dim3 grid((a.Ng + 127) / 128, rc, ngrp);
        if (pl.mode == MODE_DGRAD_S2) hipLaunchKernelGGL(splitk_reduce_stats_kernel<MODE_DGRAD_S2>, grid, dim3(256), 0, st, a, rc);
        else hipLaunchKernelGGL(splitk_reduce_stats_kernel<MODE_FWD>, grid, dim3(256), 0, st, a, rc);
        DG_CHECK_LAUNCH("splitk_reduce_stats");
    } else if (a.splits > 1) {
        const long total4 = (long)zmul * a.M * a.Ng / 4;
        int grid = (int)((total4 + 255) / 256);
        if (grid > 4096) grid = 4096;
        if (pl.mode != MODE_DGRAD_S2 && a.splits >= 8 && total4 <= 65536) {
            hipLaunchKernelGGL(splitk_reduce_small_kernel, dim3((unsigned)((total4 + 15) / 16), 1, nout), dim3(256), 0, st, a, total4);
            DG_CHECK_LAUNCH("splitk_reduce_small");
            return DG_OK;
        }
        switch (pl.mode) {
            case MODE_DGRAD_S2: hipLaunchKernelGGL(splitk_reduce_kernel<MODE_DGRAD_S2>, dim3(grid, 1, nout), dim3(256), 0, st, a, total4); break;
            default: hipLaunchKernelGGL(splitk_reduce_kernel<MODE_FWD>, dim3(grid, 1, nout), dim3(256), 0, st, a, total4); break;
        }
        DG_CHECK_LAUNCH("splitk_reduce");
    }
    return DG_OK;
}

extern "C" size_t dg_conv_workspace_bytes(int op, int N, int H, int W, int C, int K, int stride, int pad) {
    ConvGeom g;
    if (check_geom("dg_conv_workspace_bytes", N, H, W, C, K, stride, pad, &g) != DG_OK) return 0;
    if (K == 1) return 0;
    Plan pl;
    make_plan(op, g, &pl);
    size_t ws = pl.ws_bytes;
    if (dg_cur_prec() == 1) {      // the bf16-operand forms (dg_conv_*_mixed) may plan a different tile / split
        for (int v = 1; v < 4; ++v) {
            make_plan(op, g, &pl, v & 1, v >> 1);
            if (pl.ws_bytes > ws) ws = pl.ws_bytes;
        }
    }
    if (dg_cur_prec() == 2) {      // the plane-operand forms (dg_conv_*_x3)
        make_plan(op, g, &pl, 3, 3);
        if (pl.ws_bytes > ws) ws = pl.ws_bytes;
    }
    return ws;
}

// ---- the three fp32-tensor ops, grouped form -------------------------------------------------------------------------------
// `groups` problems of identical geometry go out as ONE launch per kernel of the plan (blockIdx.z = problem); a_in / b_in / out / ws /
// stat hold one pointer per problem.  share (op 2 only): `share` consecutive problems accumulate into the same dw (out[z*share + j]
// all equal): one main launch over all problems, the split-K reduction adds their slab sums in problem order.
static int head1_g(int op, int groups, int share, const void* const* a_in, int a16, const void* const* b_in, int b16, void* const* out, int out16,
                   int N, int C, int accumulate, hipStream_t st, const char* who) {
    const int J = 16 * C;
    if (op == 0) {
        DG_CHECK_ARG(!b16 && !out16, "%s: K == 1 head: weights and the [N] output are fp32", who);
        const dim3 grid(N, groups);
        if (a16) hipLaunchKernelGGL(head1_fwd_kernel<__bf16>, grid, dim3(256), 0, st, dg_ptrs(a_in, groups), dg_ptrs(b_in, groups), dg_ptrs(out, groups), J);
        else hipLaunchKernelGGL(head1_fwd_kernel<float>, grid, dim3(256), 0, st, dg_ptrs(a_in, groups), dg_ptrs(b_in, groups), dg_ptrs(out, groups), J);
    } else if (op == 1) {
        DG_CHECK_ARG(!a16 && !b16, "%s: K == 1 head: dy and the weights are fp32", who);
        const long total4 = (long)N * 4 * C;
        int gx = (int)((total4 + 255) / 256);
        if (gx > 2048) gx = 2048;
        const dim3 grid(gx, groups);
        if (out16) hipLaunchKernelGGL(head1_dgrad_kernel<__bf16>, grid, dim3(256), 0, st, dg_ptrs(a_in, groups), dg_ptrs(b_in, groups), dg_ptrs(out, groups), N, J);
        else hipLaunchKernelGGL(head1_dgrad_kernel<float>, grid, dim3(256), 0, st, dg_ptrs(a_in, groups), dg_ptrs(b_in, groups), dg_ptrs(out, groups), N, J);
    } else {
        DG_CHECK_ARG(!a16, "%s: K == 1 head: dy is fp32", who);
        // share > 1: member j of every output's problems per launch, in order (each launch accumulates into what the previous one left)
        const int sh = share > 1 ? share : 1, nout = groups / sh;
        for (int j = 0; j < sh; ++j) {
            const void *aa[DG_MAX_GROUPS], *bb[DG_MAX_GROUPS], *oo[DG_MAX_GROUPS];
            for (int z = 0; z < nout; ++z) { aa[z] = a_in[z * sh + j]; bb[z] = b_in[z * sh + j]; oo[z] = out[z * sh + j]; }
            const dim3 grid((J / 4 + 15) / 16, nout);
            const int acc = (accumulate || j > 0) ? 1 : 0;
            if (b16) hipLaunchKernelGGL(head1_wgrad_kernel<__bf16>, grid, dim3(256), 0, st, dg_ptrs(aa, nout), dg_ptrs(bb, nout), dg_ptrs(oo, nout), N, J, acc);
            else hipLaunchKernelGGL(head1_wgrad_kernel<float>, grid, dim3(256), 0, st, dg_ptrs(aa, nout), dg_ptrs(bb, nout), dg_ptrs(oo, nout), N, J, acc);
        }
    }
    DG_CHECK_LAUNCH(who);
    return DG_OK;
}

// fill the group fields of a plan's arguments from per-problem pointer arrays (problem 0's pointers are already set)
static void set_group_deltas(IgemmArgs& a, int groups, const void* const* A, const void* const* B, void* const* C) {
    a.groups = groups;
    for (int g = 1; g < groups; ++g) {
        a.gdA[g - 1] = (const char*)A[g] - (const char*)A[0];
        a.gdB[g - 1] = (const char*)B[g] - (const char*)B[0];
        a.gdC[g - 1] = (const char*)C[g] - (const char*)C[0];
    }
}

static int conv_g(int op, int groups, int share, const float* const* a_in, const float* const* b_in, float* const* out, int N, int H, int W, int C,
                  int K, int stride, int pad, int prec, int plan_groups, int accumulate, float* const* stat, size_t stat_floats, void* const* ws,
                  size_t ws_bytes, hipStream_t st, const char* who) {
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS, "%s: groups=%d (1..%d)", who, groups, DG_MAX_GROUPS);
    DG_CHECK_ARG(prec >= DG_PREC_DEFAULT && prec <= DG_PREC_F32X3, "%s: prec=%d (DG_PREC_F32 | DG_PREC_BF16 | DG_PREC_F32X3)", who, prec);
    DG_CHECK_ARG(share <= 1 || (op == 2 && groups % share == 0), "%s: share=%d needs the weight gradient and groups %% share == 0", who, share);
    ConvGeom g;
    int rc = check_geom(who, N, H, W, C, K, stride, pad, &g);
    if (rc) return rc;
    DG_CHECK_ARG(a_in && b_in && out, "%s: null pointer table", who);
    for (int i = 0; i < groups; ++i) DG_CHECK_ARG(a_in[i] && b_in[i] && out[i], "%s: null pointer (problem %d)", who, i);
    if (share > 1)
        for (int i = 0; i < groups; ++i) DG_CHECK_ARG(out[i] == out[i / share * share], "%s: problems of one share set must name the same dw", who);
    if (plan_groups < 0) plan_groups = -plan_groups;          // internal: a member launch of a share set keeps its parent's plan
    else DG_CHECK_ARG(plan_groups == 1 || plan_groups == groups, "%s: plan_groups=%d (1 or groups=%d)", who, plan_groups, groups);
    DgPrecScope scope(prec);
    DgPlanScope pscope(plan_groups);
    if (K == 1) {
        DG_CHECK_ARG(stride == 1, "%s: K==1 only for the 4x4 head", who);
        return head1_g(op, groups, share, (const void* const*)a_in, 0, (const void* const*)b_in, 0, (void* const*)out, 0, N, C, accumulate, st, who);
    }
    if (op == 0) DG_CHECK_ARG(C % 32 == 0, "%s: C=%d must be a multiple of 32", who, C);
    if (op == 1 && stride == 2) DG_CHECK_ARG(K % 32 == 0, "%s: K=%d must be a multiple of 32", who, K);
    Plan pl;
    make_plan(op, g, &pl);
    IgemmArgs& a = pl.a;
    if (share > 1 && a.splits <= 1) {
        // no reduction kernel to merge the shared outputs in: member j of every share set per launch, in order
        const int nout = groups / share;
        for (int j = 0; j < share; ++j) {
            const float *aa[DG_MAX_GROUPS], *bb[DG_MAX_GROUPS];
            float* oo[DG_MAX_GROUPS];
            for (int z = 0; z < nout; ++z) { aa[z] = a_in[z * share + j]; bb[z] = b_in[z * share + j]; oo[z] = out[z * share + j]; }
            void* wj[DG_MAX_GROUPS];
            for (int z = 0; z < nout; ++z) wj[z] = ws ? ws[z * share + j] : nullptr;
            // (plan_groups handed on unchanged: the same unsplit plan for every member launch)
            rc = conv_g(op, nout, 1, aa, bb, oo, N, H, W, C, K, stride, pad, prec, -plan_groups, (accumulate || j > 0) ? 1 : 0, nullptr, 0, wj, ws_bytes, st, who);
            if (rc) return rc;
        }
        return DG_OK;
    }
    a.A = a_in[0]; a.B = b_in[0]; a.C = out[0]; a.accumulate = accumulate;
    set_group_deltas(a, groups, (const void* const*)a_in, (const void* const*)b_in, (void* const*)out);
    a.share = share > 1 ? share : 1;
    if (stat != nullptr) {
        const int ncols = op == 0 ? K : C;
        DG_CHECK_ARG(op != 2 && pl.stat_rows > 0 && stat_floats >= (size_t)pl.stat_rows * (3 * ncols + 4),
                     "%s: statistics buffer too small or no fused statistics for this plan (ask dg_conv_bnstats_rows_p)", who);
        for (int i = 0; i < groups; ++i) DG_CHECK_ARG(stat[i], "%s: null statistics pointer (problem %d)", who, i);
        a.stat = stat[0];
        a.stat_rs = 3 * ncols + 4;
        for (int i = 1; i < groups; ++i) a.gdStat[i - 1] = (const char*)stat[i] - (const char*)stat[0];
    }
    void* ws0 = nullptr;
    if (a.splits > 1) {
        DG_CHECK_ARG(ws != nullptr, "%s: this plan splits K and needs a workspace per problem", who);
        for (int i = 0; i < groups; ++i)
            if (ws[i] == nullptr || ws_bytes < pl.ws_bytes) return dg_fail(DG_ERR_WORKSPACE, "%s: workspace %zu < required %zu (problem %d)", who, ws_bytes, pl.ws_bytes, i);
        ws0 = ws[0];
        for (int i = 1; i < groups; ++i) a.gdPart[i - 1] = (const char*)ws[i] - (const char*)ws[0];
    }
    return run_plan(who, pl, ws0, ws_bytes, st);
}

extern "C" int dg_conv_fwd_g(int groups, const float* const* x, const float* const* w, float* const* y, int N, int H, int W, int C, int K, int stride,
                             int pad, int prec, int plan_groups, float* const* stat, size_t stat_floats, void* const* ws, size_t ws_bytes,
                             dg_stream_t stream) {
    return conv_g(0, groups, 1, x, w, y, N, H, W, C, K, stride, pad, prec, plan_groups, 0, stat, stat_floats, ws, ws_bytes, (hipStream_t)stream, "dg_conv_fwd_g");
}
extern "C" int dg_conv_dgrad_g(int groups, const float* const* dy, const float* const* w, float* const* dx, int N, int H, int W, int C, int K, int stride,
                               int pad, int prec, int plan_groups, float* const* stat, size_t stat_floats, void* const* ws, size_t ws_bytes,
                               dg_stream_t stream) {
    return conv_g(1, groups, 1, dy, w, dx, N, H, W, C, K, stride, pad, prec, plan_groups, 0, stat, stat_floats, ws, ws_bytes, (hipStream_t)stream, "dg_conv_dgrad_g");
}
extern "C" int dg_conv_wgrad_g(int groups, int share, const float* const* dy, const float* const* x, float* const* dw, int N, int H, int W, int C, int K,
                               int stride, int pad, int prec, int plan_groups, int accumulate, void* const* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_g(2, groups, share, dy, x, dw, N, H, W, C, K, stride, pad, prec, plan_groups, accumulate, nullptr, 0, ws, ws_bytes, (hipStream_t)stream, "dg_conv_wgrad_g");
}
// planning queries with the arithmetic as an argument (the forms without it read the process default)
extern "C" size_t dg_conv_workspace_bytes_p(int op, int N, int H, int W, int C, int K, int stride, int pad, int prec, int plan_groups) {
    DgPrecScope scope(prec);
    DgPlanScope pscope(plan_groups);
    return dg_conv_workspace_bytes(op, N, H, W, C, K, stride, pad);
}
extern "C" int dg_conv_bnstats_rows_p(int op, int N, int H, int W, int C, int K, int stride, int pad, int prec) {
    DgPrecScope scope(prec);
    return dg_conv_bnstats_rows(op, N, H, W, C, K, stride, pad);
}
extern "C" int dg_conv_plan_splits_p(int op, int N, int H, int W, int C, int K, int stride, int pad, int prec, int plan_groups) {
    DgPrecScope scope(prec);
    DgPlanScope pscope(plan_groups);
    return dg_conv_plan_splits(op, N, H, W, C, K, stride, pad);
}

// the one-problem forms without a precision argument: the process default arithmetic (dg_set_option("bf16"))
extern "C" int dg_conv_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int K,
                           int stride, int pad, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_g(0, 1, 1, &x, &w, &y, N, H, W, C, K, stride, pad, dg_cur_prec(), 1, 0, nullptr, 0, &ws, ws_bytes, (hipStream_t)stream, "dg_conv_fwd");
}
extern "C" int dg_conv_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K,
                             int stride, int pad, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_g(1, 1, 1, &dy, &w, &dx, N, H, W, C, K, stride, pad, dg_cur_prec(), 1, 0, nullptr, 0, &ws, ws_bytes, (hipStream_t)stream, "dg_conv_dgrad");
}
extern "C" int dg_conv_wgrad(const float* dy, const float* x, float* dw, int N, int H, int W, int C, int K,
                             int stride, int pad, int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_g(2, 1, 1, &dy, &x, &dw, N, H, W, C, K, stride, pad, dg_cur_prec(), 1, accumulate, nullptr, 0, &ws, ws_bytes, (hipStream_t)stream, "dg_conv_wgrad");
}

// ---- bf16 shadow operands (option "bf16" = 1 only) -------------------------------------------------------------------
// The same three ops with either operand given as a bf16 tensor of the same logical layout.  Producers write the shadow
// next to the fp32 tensor (dg_adam_step_flat_bf16 for weights, dg_bn_act_fwd_bf16 / dg_bn_act_bwd_bf16 / the c3 forward for
// activations and gradients); the result is bit-identical to passing the fp32 tensors (same RNE rounding, same order).
static int conv_mixed(int op, const void* a_in, int a16, const void* b_in, int b16, void* out, int out16, int N, int H, int W, int C, int K,
                      int stride, int pad, int accumulate, void* ws, size_t ws_bytes, hipStream_t st, float* stat = nullptr,
                      size_t stat_floats = 0) {
    const char* who = op == 0 ? "dg_conv_fwd_mixed" : (op == 1 ? "dg_conv_dgrad_mixed" : "dg_conv_wgrad_mixed");
    ConvGeom g;
    int rc = check_geom(who, N, H, W, C, K, stride, pad, &g);
    if (rc) return rc;
    DG_CHECK_ARG(a_in && b_in && out, "%s: null pointer", who);
    if (K == 1) {
        // the discriminator's 4x4 head (plain reductions): only the ACTIVATION side may be bf16 (x of forward / weight-grad, dx of
        // input-grad); the [N] vector and the weights are fp32
        DG_CHECK_ARG(stride == 1, "%s: K==1 only for the 4x4 head", who);
        return head1_g(op, 1, 1, &a_in, a16, &b_in, b16, &out, out16, N, C, accumulate, st, who);
    }
    // bf16 operands / outputs imply the bf16 matrix path for this call (no process-wide switch involved); an all-fp32 call keeps the
    // caller's arithmetic
    DgPrecScope scope((a16 || b16 || out16) ? DG_PREC_BF16 : dg_cur_prec());
    DG_CHECK_ARG(!(out16 && op == 2), "%s: the weight gradient is always fp32", who);
    if (op == 0) DG_CHECK_ARG(C % 32 == 0, "%s: C=%d must be a multiple of 32", who, C);
    if (op == 1 && stride == 2) DG_CHECK_ARG(K % 32 == 0, "%s: K=%d must be a multiple of 32", who, K);
    // 8-element granules must not straddle a row end: the A operand of the plain GEMM forms has rows of K elements
    if (a16 && op != 0 && K % 8 != 0) return dg_fail(DG_ERR_INVALID, "%s: a bf16 gradient operand needs K %% 8 == 0 (K=%d)", who, K);
    Plan pl;
    make_plan(op, g, &pl, a16, b16);
    if (pl.a.prec != 1 && (a16 || b16 || out16)) return dg_fail(DG_ERR_INVALID, "%s: this shape has no bf16 tile kernel", who);
    pl.a.A = (const float*)a_in; pl.a.B = (const float*)b_in; pl.a.C = (float*)out; pl.a.accumulate = accumulate;
    pl.a.out16 = out16;
    if (stat) {     // fused BatchNorm partial statistics of the output (from the fp32 accumulators): rows from dg_conv_mixed_bnstats_rows
        const int ncols = op == 0 ? K : C;
        DG_CHECK_ARG(op != 2 && pl.stat_rows > 0 && stat_floats >= (size_t)pl.stat_rows * (3 * ncols + 4),
                     "%s: statistics buffer too small or no fused statistics for this plan (ask dg_conv_mixed_bnstats_rows)", who);
        pl.a.stat = stat;
        pl.a.stat_rs = 3 * ncols + 4;
    }
    return run_plan(who, pl, ws, ws_bytes, st);
}
extern "C" int dg_conv_mixed_bnstats_rows(int op, int N, int H, int W, int C, int K, int stride, int pad, int a_bf16, int b_bf16) {
    ConvGeom g;
    if ((op != 0 && op != 1) || check_geom("dg_conv_mixed_bnstats_rows", N, H, W, C, K, stride, pad, &g) != DG_OK || K == 1 || stride != 2) return 0;
    DgPrecScope scope(DG_PREC_BF16);
    Plan pl;
    make_plan(op, g, &pl, a_bf16, b_bf16);
    return pl.stat_rows;
}
extern "C" int dg_conv_fwd_mixed(const void* x, int x_bf16, const void* w, int w_bf16, void* y, int y_bf16, int N, int H, int W, int C, int K,
                                 int stride, int pad, float* stat, size_t stat_floats, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_mixed(0, x, x_bf16, w, w_bf16, y, y_bf16, N, H, W, C, K, stride, pad, 0, ws, ws_bytes, (hipStream_t)stream, stat, stat_floats);
}
extern "C" int dg_conv_dgrad_mixed(const void* dy, int dy_bf16, const void* w, int w_bf16, void* dx, int dx_bf16, int N, int H, int W, int C, int K,
                                   int stride, int pad, float* stat, size_t stat_floats, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_mixed(1, dy, dy_bf16, w, w_bf16, dx, dx_bf16, N, H, W, C, K, stride, pad, 0, ws, ws_bytes, (hipStream_t)stream, stat, stat_floats);
}
extern "C" int dg_conv_wgrad_mixed(const void* dy, int dy_bf16, const void* x, int x_bf16, float* dw, int N, int H, int W, int C, int K,
                                   int stride, int pad, int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_mixed(2, dy, dy_bf16, x, x_bf16, dw, 0, N, H, W, C, K, stride, pad, accumulate, ws, ws_bytes, (hipStream_t)stream);
}
// can this (op, shape) take bf16 operands at all?  (host planning aid: 0 = no, 1 = the register-staged bf16 tile kernel,
// 2 = with BOTH operands bf16 the LDS-DMA kernel of igemm_dma.hip runs)
extern "C" int dg_conv_bf16_operands_ok(int op, int N, int H, int W, int C, int K, int stride, int pad) {
    ConvGeom g;
    if (check_geom("dg_conv_bf16_operands_ok", N, H, W, C, K, stride, pad, &g) != DG_OK || K == 1) return 0;
    DgPrecScope scope(DG_PREC_BF16);
    Plan pl;
    make_plan(op, g, &pl, 1, 1);
    return pl.a.prec == 1 ? (pl.dma ? 2 : 1) : 0;
}

// ---- fp32 operands as three bf16 planes (option "bf16" = 2 only): igemm_dma_x3.hip -----------------------------------------
// a3 / b3 point at plane 0 (hi) of an operand; planes 1 (mid) and 2 (lo) follow a_plane / b_plane BYTES further on (>= the
// tensor's 2 * numel; a weight inside a flat parameter group has the group's plane distance).  Written by dg_f32_to_bf16x3 or
// by the fused producers (dg_adam_step_flat_x3, dg_bn_act_fwd_x3, dg_bn_act_bwd_x3); outputs are fp32.
// forward, stride 2, no LDS-DMA plane kernel for the shape (fewer than 192 output channels or rows): the register-staged 128 x 128
// tiles can read plane triples (16-byte granules of 8 bf16: C % 8 == 0; buffer-descriptor kernels only)
static bool x3_register_staged_planes_ok(int op, const ConvGeom& g, const Plan& pl) {
#ifndef DG_EXPERIMENTS
    return false;           // (the plane-reading register-staged tiles exist in the experiments library only)
#endif
    return op == 0 && g.stride == 2 && pl.a.prec == 2 && pl.mode == MODE_FWD && pl.wm == 2 && pl.wn == 2 && pl.kt == 16 && g.C % 16 == 0 &&
           pl.a.abytes != 0 && pl.a.bbytes != 0;
}
static int conv_x3(int op, const void* a3, long a_plane, const void* b3, long b_plane, int b_transposed, float* out, int N, int H, int W, int C, int K,
                   int stride, int pad, int accumulate, void* ws, size_t ws_bytes, hipStream_t st, int a_layout = 0, float* stat = nullptr,
                   size_t stat_floats = 0) {
    const char* who = op == 0 ? "dg_conv_fwd_x3" : (op == 1 ? "dg_conv_dgrad_x3" : "dg_conv_wgrad_x3");
    ConvGeom g;
    int rc = check_geom(who, N, H, W, C, K, stride, pad, &g);
    if (rc) return rc;
    DG_CHECK_ARG(a3 && b3 && out, "%s: null pointer", who);
    DgPrecScope scope(DG_PREC_F32X3);       // plane operands ARE the f32x3 arithmetic: no process-wide switch involved
    DG_CHECK_ARG(K > 1, "%s: the K == 1 head has no plane form", who);
    Plan pl;
    // forward with plain (not transposed) weight planes where the window forward kernel would apply: the register-staged tiles read
    // the planes instead (igemm_kernel<.., PREC 2, A16, B16>)
    make_plan(op, g, &pl, 3, 3, (op == 0 && !b_transposed) ? 0 : 1);
    const bool rs_planes = pl.dma == 0 && x3_register_staged_planes_ok(op, g, pl);
    if (pl.dma != 2 && pl.dma != 3 && pl.dma != 5 && !rs_planes)
        return dg_fail(DG_ERR_INVALID, "%s: this shape has no plane kernel (ask dg_conv_x3_planes_ok)", who);
    DG_CHECK_ARG(!(rs_planes && b_transposed), "%s: the register-staged plane reader takes the PLAIN weight planes", who);
    DG_CHECK_ARG(a_plane >= (long)pl.a.abytes && b_plane >= (long)pl.a.bbytes && a_plane % 16 == 0 && b_plane % 16 == 0,
                 "%s: plane distances %ld / %ld (operands are %u / %u bytes per plane)", who, a_plane, b_plane, pl.a.abytes, pl.a.bbytes);
    pl.a.A = (const float*)a3; pl.a.B = (const float*)b3; pl.a.C = out; pl.a.accumulate = accumulate;
    pl.a.a_plane = a_plane; pl.a.b_plane = b_plane;
    DG_CHECK_ARG(!b_transposed || op == 0, "%s: only the forward form takes transposed weight planes", who);
    pl.a.b_transposed = b_transposed ? 1 : 0;
    // quad-chunk gradient planes [pixels / 4][K / 16][4][16]: read by the window input-grad kernel and by the plane weight-grad kernel
    DG_CHECK_ARG(a_layout == 0 || a_layout == 1, "%s: plane layout %d", who, a_layout);
    DG_CHECK_ARG(a_layout == 0 || (K % 64 == 0 && ((op == 1 && pl.dma == 3) || (op == 2 && pl.dma == 2))),
                 "%s: quad-chunk planes are only read by the window input-grad and the plane weight-grad kernels (K %% 64 == 0)", who);
    pl.a.a_cm = a_layout;
    if (stat) {     // fused BatchNorm partial statistics of the output (forward / input-grad): rows from dg_conv_x3_bnstats_rows
        const int ncols = op == 0 ? K : C;
        DG_CHECK_ARG(op != 2 && pl.stat_rows > 0 && stat_floats >= (size_t)pl.stat_rows * (3 * ncols + 4),
                     "%s: statistics buffer too small or no fused statistics for this plan (ask dg_conv_x3_bnstats_rows)", who);
        pl.a.stat = stat;
        pl.a.stat_rs = 3 * ncols + 4;
    }
    return run_plan(who, pl, ws, ws_bytes, st);
}
// w_transposed: w3 is the transposed copy wT[(r, s, c)][k] of the weight planes (dg_x3_transpose_planes) -- the faster form
extern "C" int dg_conv_fwd_x3(const void* x3, long x_plane, const void* w3, long w_plane, int w_transposed, float* y, int N, int H, int W,
                              int C, int K, int stride, int pad, float* stat, size_t stat_floats, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_x3(0, x3, x_plane, w3, w_plane, w_transposed, y, N, H, W, C, K, stride, pad, 0, ws, ws_bytes, (hipStream_t)stream, 0, stat, stat_floats);
}
extern "C" int dg_conv_dgrad_x3(const void* dy3, long dy_plane, int dy_layout, const void* w3, long w_plane, float* dx, int N, int H, int W, int C, int K,
                                int stride, int pad, float* stat, size_t stat_floats, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_x3(1, dy3, dy_plane, w3, w_plane, 0, dx, N, H, W, C, K, stride, pad, 0, ws, ws_bytes, (hipStream_t)stream, dy_layout, stat, stat_floats);
}
// partial-statistics rows the plane kernel of (op, shape) emits (0: none -- no plane kernel, or one without the epilogue)
extern "C" int dg_conv_x3_bnstats_rows(int op, int N, int H, int W, int C, int K, int stride, int pad) {
    ConvGeom g;
    if (op == 2 || check_geom("dg_conv_x3_bnstats_rows", N, H, W, C, K, stride, pad, &g) != DG_OK || K == 1) return 0;
    DgPrecScope scope(DG_PREC_F32X3);
    Plan pl;
    make_plan(op, g, &pl, 3, 3);
    return (pl.dma == 2 || pl.dma == 3) ? pl.stat_rows : 0;
}
extern "C" int dg_conv_wgrad_x3(const void* dy3, long dy_plane, int dy_layout, const void* x3, long x_plane, float* dw, int N, int H, int W, int C, int K,
                                int stride, int pad, int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_x3(2, dy3, dy_plane, x3, x_plane, 0, dw, N, H, W, C, K, stride, pad, accumulate, ws, ws_bytes, (hipStream_t)stream, dy_layout);
}
// does this (op, shape) have a plane kernel under option bf16 = 2?  (host planning aid)  0: no; 1: yes; 3: yes -- the window
// FORWARD kernel with the transposed weight planes (w_transposed = 1 of dg_conv_fwd_x3), the register-staged plane reader with the
// plain ones; 4: yes -- the register-staged tiles read the (plain) planes; 2: yes, and it is the window input-grad kernel, which
// wants its gradient operand in the quad-chunk layout (plane_layout 1 of dg_bn_act_*_x3)
extern "C" int dg_conv_x3_planes_ok(int op, int N, int H, int W, int C, int K, int stride, int pad) {
    ConvGeom g;
    if (check_geom("dg_conv_x3_planes_ok", N, H, W, C, K, stride, pad, &g) != DG_OK || K == 1) return 0;
    DgPrecScope scope(DG_PREC_F32X3);
    Plan pl;
    make_plan(op, g, &pl, 3, 3);
    if (pl.dma == 0 && x3_register_staged_planes_ok(op, g, pl)) return 4;
    return pl.dma == 5 ? 3 : (pl.dma == 3 ? (K % 64 == 0 ? 2 : 1) : (pl.dma == 2 ? 1 : 0));
}

// ---- inference path: conv with BatchNorm folded in (scale in the weights, shift as a bias) + activation ----------
// Replaces [Conv2d | ConvTranspose2d] -> BatchNorm2d(eval) -> LeakyReLU/ReLU of inference.py:149,172-187 by ONE kernel:
// the caller multiplies the weights by gamma*invstd per output channel and passes bias = beta - mean*gamma*invstd.
static int conv_bias_act(int op, const float* a_in, const float* w, const float* bias, float* out, int N, int H, int W, int C, int K,
                         int stride, int pad, int act, float slope, void* ws, size_t ws_bytes, hipStream_t st) {
    const char* who = op == 0 ? "dg_conv_fwd_bias_act" : "dg_conv_dgrad_bias_act";
    ConvGeom g;
    int rc = check_geom(who, N, H, W, C, K, stride, pad, &g);
    if (rc) return rc;
    DG_CHECK_ARG(a_in && w && out, "%s: null pointer", who);
    DG_CHECK_ARG(K > 1, "%s: K == 1 head has no folded form", who);
    DG_CHECK_ARG(act == DG_ACT_NONE || act == DG_ACT_LEAKY || act == DG_ACT_RELU, "%s: bad act %d", who, act);
    if (op == 0) DG_CHECK_ARG(C % 32 == 0, "%s: C=%d must be a multiple of 32", who, C);
    if (op == 1 && stride == 2) DG_CHECK_ARG(K % 32 == 0, "%s: K=%d must be a multiple of 32", who, K);
    Plan pl;
    make_plan(op, g, &pl);
    pl.a.A = a_in; pl.a.B = w; pl.a.C = out;
    pl.a.bias = bias; pl.a.act = act; pl.a.slope = slope;
    pl.a.bias_mod = (pl.mode == MODE_DGRAD_PLAIN) ? C : pl.a.Ng;
    return run_plan(who, pl, ws, ws_bytes, st);
}
extern "C" int dg_conv_fwd_bias_act(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int K,
                                    int stride, int pad, int act, float slope, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_bias_act(0, x, w, bias, y, N, H, W, C, K, stride, pad, act, slope, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int dg_conv_dgrad_bias_act(const float* dy, const float* w, const float* bias, float* dx, int N, int H, int W, int C, int K,
                                      int stride, int pad, int act, float slope, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_bias_act(1, dy, w, bias, dx, N, H, W, C, K, stride, pad, act, slope, ws, ws_bytes, (hipStream_t)stream);
}

// ---- conv + fused BatchNorm partial statistics -------------------------------------------------------
extern "C" int dg_conv_plan_splits(int op, int N, int H, int W, int C, int K, int stride, int pad) {
    ConvGeom g;
    if (check_geom("dg_conv_plan_splits", N, H, W, C, K, stride, pad, &g) != DG_OK || K == 1) return 0;
    Plan pl;
    make_plan(op, g, &pl);
    return pl.a.splits;
}
extern "C" int dg_conv_bnstats_rows(int op, int N, int H, int W, int C, int K, int stride, int pad) {
    ConvGeom g;
    if (check_geom("dg_conv_bnstats_rows", N, H, W, C, K, stride, pad, &g) != DG_OK) return 0;
    if (K == 1 || stride != 2 || (op != 0 && op != 1)) return 0;
    Plan pl;
    make_plan(op, g, &pl);
    return pl.stat_rows;
}
static int conv_with_stats(int op, const float* a_in, const float* w, float* out, int N, int H, int W, int C, int K,
                           float* stat, size_t stat_floats, void* ws, size_t ws_bytes, hipStream_t st) {
    const char* who = op == 0 ? "dg_conv_fwd_bnstats" : "dg_conv_dgrad_bnstats";
    ConvGeom g;
    int rc = check_geom(who, N, H, W, C, K, 2, 1, &g);
    if (rc) return rc;
    DG_CHECK_ARG(a_in && w && out && stat, "%s: null pointer", who);
    DG_CHECK_ARG(K % 32 == 0 && C % 32 == 0, "%s: C and K must be multiples of 32", who);
    Plan pl;
    make_plan(op, g, &pl);
    const int ncols = op == 0 ? K : C;
    DG_CHECK_ARG(pl.stat_rows > 0 && stat_floats >= (size_t)pl.stat_rows * (3 * ncols + 4), "%s: statistics buffer too small", who);
    pl.a.A = a_in; pl.a.B = w; pl.a.C = out;
    pl.a.stat = stat; pl.a.stat_rs = 3 * ncols + 4;
    return run_plan(who, pl, ws, ws_bytes, st);
}
extern "C" int dg_conv_fwd_bnstats(const float* x, const float* w, float* y, int N, int H, int W, int C, int K,
                                   float* stat, size_t stat_floats, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_with_stats(0, x, w, y, N, H, W, C, K, stat, stat_floats, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int dg_conv_dgrad_bnstats(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K,
                                     float* stat, size_t stat_floats, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return conv_with_stats(1, dy, w, dx, N, H, W, C, K, stat, stat_floats, ws, ws_bytes, (hipStream_t)stream);
}

// ---- named wrappers --------------------------------------------------------------
extern "C" int dg_conv4x4s2_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int K,
                                void* ws, size_t wsb, dg_stream_t s) {
    return dg_conv_fwd(x, w, y, N, H, W, C, K, 2, 1, ws, wsb, s);
}
extern "C" int dg_conv4x4s2_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K,
                                  void* ws, size_t wsb, dg_stream_t s) {
    return dg_conv_dgrad(dy, w, dx, N, H, W, C, K, 2, 1, ws, wsb, s);
}
extern "C" int dg_conv4x4s2_wgrad(const float* dy, const float* x, float* dw, int N, int H, int W, int C, int K,
                                  int acc, void* ws, size_t wsb, dg_stream_t s) {
    return dg_conv_wgrad(dy, x, dw, N, H, W, C, K, 2, 1, acc, ws, wsb, s);
}
extern "C" int dg_conv4x4_valid_fwd(const float* x, const float* w, float* y, int N, int C, int K,
                                    void* ws, size_t wsb, dg_stream_t s) {
    return dg_conv_fwd(x, w, y, N, 4, 4, C, K, 1, 0, ws, wsb, s);
}
extern "C" int dg_conv4x4_valid_dgrad(const float* dy, const float* w, float* dx, int N, int C, int K,
                                      void* ws, size_t wsb, dg_stream_t s) {
    return dg_conv_dgrad(dy, w, dx, N, 4, 4, C, K, 1, 0, ws, wsb, s);
}
extern "C" int dg_conv4x4_valid_wgrad(const float* dy, const float* x, float* dw, int N, int C, int K,
                                      int acc, void* ws, size_t wsb, dg_stream_t s) {
    return dg_conv_wgrad(dy, x, dw, N, 4, 4, C, K, 1, 0, acc, ws, wsb, s);
}
// ConvTranspose2d(Cin,Cout,4,2,1) == dgrad of Conv2d(C=Cout -> K=Cin) on the 2Hin x 2Win grid
extern "C" int dg_convT4x4s2_fwd(const float* x, const float* w, float* y, int N, int Hin, int Win, int Cin, int Cout,
                                 void* ws, size_t wsb, dg_stream_t s) {
    return dg_conv_dgrad(x, w, y, N, 2 * Hin, 2 * Win, Cout, Cin, 2, 1, ws, wsb, s);
}
extern "C" int dg_convT4x4s2_dgrad(const float* dy, const float* w, float* dx, int N, int Hin, int Win, int Cin, int Cout,
                                   void* ws, size_t wsb, dg_stream_t s) {
    return dg_conv_fwd(dy, w, dx, N, 2 * Hin, 2 * Win, Cout, Cin, 2, 1, ws, wsb, s);
}
extern "C" int dg_convT4x4s2_wgrad(const float* dy, const float* x, float* dw, int N, int Hin, int Win, int Cin, int Cout,
                                   int acc, void* ws, size_t wsb, dg_stream_t s) {
    // dw[cin][r][s][cout] = sum x[.., cin] * dy[.., cout] : roles (dy := x, x := dy)
    return dg_conv_wgrad(x, dy, dw, N, 2 * Hin, 2 * Win, Cout, Cin, 2, 1, acc, ws, wsb, s);
}
extern "C" int dg_convT4x4_1to4_fwd(const float* x, const float* w, float* y, int N, int Cin, int Cout,
                                    void* ws, size_t wsb, dg_stream_t s) {
    return dg_conv_dgrad(x, w, y, N, 4, 4, Cout, Cin, 1, 0, ws, wsb, s);
}
extern "C" int dg_convT4x4_1to4_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cout,
                                      void* ws, size_t wsb, dg_stream_t s) {
    return dg_conv_fwd(dy, w, dx, N, 4, 4, Cout, Cin, 1, 0, ws, wsb, s);
}
extern "C" int dg_convT4x4_1to4_wgrad(const float* dy, const float* x, float* dw, int N, int Cin, int Cout,
                                      int acc, void* ws, size_t wsb, dg_stream_t s) {
    return dg_conv_wgrad(x, dy, dw, N, 4, 4, Cout, Cin, 1, 0, acc, ws, wsb, s);
}

extern "C" int dg_c3_fwd_mfma_launch(const float* x_nchw, const float* w, void* y_nhwc, int y_bf16, int N, int H, int W, int act,
                                     float slope, hipStream_t st);   // edge.hip
extern "C" int dg_c3_fwd_mfma_launch_g(int groups, const float* const* x_tab, const float* const* w_tab, void* const* y_tab, int y_bf16, int N, int H, int W,
                                       int act, float slope, hipStream_t st);   // edge.hip
static int c3_fwd_run(const float* x_nchw, const float* w, float* y_nhwc, int y_bf16, int N, int H, int W, int K,
                      int act, float slope, dg_stream_t stream);
// ---- 3-channel image side, forward direction (conv1 forward / last-convT input-grad) ----------------
// grouped form: the K == 64 streaming kernels only (the network's first conv / last transposed conv), arithmetic as an argument
extern "C" int dg_conv4x4s2_c3_fwd_g(int groups, const float* const* x_nchw, const float* const* w, float* const* y_nhwc, int N, int H, int W, int K,
                                     int act, float slope, int prec, dg_stream_t stream) {
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS && x_nchw && w && y_nhwc, "dg_conv4x4s2_c3_fwd_g: bad group / null table");
    for (int i = 0; i < groups; ++i) DG_CHECK_ARG(x_nchw[i] && w[i] && y_nhwc[i], "dg_conv4x4s2_c3_fwd_g: null pointer");
    DG_CHECK_ARG(prec >= DG_PREC_DEFAULT && prec <= DG_PREC_F32X3, "dg_conv4x4s2_c3_fwd_g: prec=%d", prec);
    DG_CHECK_ARG(N >= 1 && K == 64, "dg_conv4x4s2_c3_fwd_g: the grouped form is the K == 64 streaming kernel (K=%d)", K);
    DG_CHECK_ARG(dg_is_pow2(H) && dg_is_pow2(W) && H >= 2 && W >= 2, "dg_conv4x4s2_c3_fwd_g: H,W must be powers of two");
    DG_CHECK_ARG((long)N * 3 * H * W * 4 < (1L << 30) && (long)N * (H / 2) * (W / 2) < (1L << 30) && dg_get_option(DG_OPT_KT) != 16,
                 "dg_conv4x4s2_c3_fwd_g: tensors must be below 1 GiB");
    DG_CHECK_ARG(act == DG_ACT_NONE || act == DG_ACT_LEAKY || act == DG_ACT_RELU, "dg_conv4x4s2_c3_fwd_g: bad act %d", act);
    DgPrecScope scope(prec);
    int rc = dg_c3_fwd_mfma_launch_g(groups, x_nchw, w, (void* const*)y_nhwc, 0, N, H, W, act, slope, (hipStream_t)stream);
    if (rc != DG_OK) return rc;
    DG_CHECK_LAUNCH("dg_conv4x4s2_c3_fwd_g");
    return DG_OK;
}
extern "C" int dg_conv4x4s2_c3_fwd(const float* x_nchw, const float* w, float* y_nhwc, int N, int H, int W, int K,
                                   int act, float slope, dg_stream_t stream) {
    return c3_fwd_run(x_nchw, w, y_nhwc, 0, N, H, W, K, act, slope, stream);
}
extern "C" int dg_conv4x4s2_c3_fwd_t(const float* x_nchw, const float* w, void* y_nhwc, int y_bf16, int N, int H, int W, int K,
                                     int act, float slope, dg_stream_t stream) {
    return c3_fwd_run(x_nchw, w, (float*)y_nhwc, y_bf16, N, H, W, K, act, slope, stream);
}
extern "C" int dg_conv4x4s2_c3_fwd_p(const float* x_nchw, const float* w, void* y_nhwc, int y_bf16, int N, int H, int W, int K,
                                     int act, float slope, int prec, dg_stream_t stream) {
    DG_CHECK_ARG(prec >= DG_PREC_DEFAULT && prec <= DG_PREC_F32X3, "dg_conv4x4s2_c3_fwd_p: prec=%d", prec);
    DgPrecScope scope(prec);
    return c3_fwd_run(x_nchw, w, (float*)y_nhwc, y_bf16, N, H, W, K, act, slope, stream);
}
static int c3_fwd_run(const float* x_nchw, const float* w, float* y_nhwc, int y_bf16, int N, int H, int W, int K,
                      int act, float slope, dg_stream_t stream) {
    DG_CHECK_ARG(x_nchw && w && y_nhwc, "dg_conv4x4s2_c3_fwd: null pointer");
    DG_CHECK_ARG(N >= 1 && K >= 4 && K % 4 == 0, "dg_conv4x4s2_c3_fwd: bad N/K (%d,%d)", N, K);
    DG_CHECK_ARG(dg_is_pow2(H) && dg_is_pow2(W) && H >= 2 && W >= 2, "dg_conv4x4s2_c3_fwd: H,W must be powers of two");
    DG_CHECK_ARG((long)N * 3 * H * W < (1L << 31), "dg_conv4x4s2_c3_fwd: tensor too large");
    DG_CHECK_ARG(act == DG_ACT_NONE || act == DG_ACT_LEAKY || act == DG_ACT_RELU, "dg_conv4x4s2_c3_fwd: bad act %d", act);
    hipStream_t st = (hipStream_t)stream;
    if (K == 64 && dg_get_option(DG_OPT_KT) != 16 && (long)N * 3 * H * W * 4 < (1L << 30) && (long)N * (H / 2) * (W / 2) < (1L << 30)) {       // streaming per-wave kernel (edge.hip); kt=16 forces the tiled path
        int rc = dg_c3_fwd_mfma_launch(x_nchw, w, y_nhwc, y_bf16, N, H, W, act, slope, st);
        if (rc != DG_OK) return rc;
        DG_CHECK_LAUNCH("dg_conv4x4s2_c3_fwd");
        return DG_OK;
    }
    DG_CHECK_ARG(!y_bf16, "dg_conv4x4s2_c3_fwd_t: a bf16 output needs K == 64 and tensors < 1 GiB (the streaming kernel)");
    IgemmArgs a = IgemmArgs();
    a.A = x_nchw; a.B = w; a.C = y_nhwc;
    a.N = N; a.H = H; a.W = W; a.Cc = 3; a.K = K;
    a.Ho = H / 2; a.Wo = W / 2; a.lgHo = dg_ilog2(a.Ho); a.lgWo = dg_ilog2(a.Wo);
    a.stride = 2; a.pad = 1;
    a.M = N * a.Ho * a.Wo; a.Ng = K; a.R = 48;
    a.nIt = 3; a.itPerSplit = 3; a.splits = 1;
    a.act = act; a.slope = slope;
    if (K <= 64) {
        a.tilesM = (a.M + 255) / 256; a.tilesN = 1;
        launch_igemm<MODE_FWD_C3, 4, 1, 16>(a, 1, st);
    } else {
        a.tilesM = (a.M + 127) / 128; a.tilesN = (K + 127) / 128;
        launch_igemm<MODE_FWD_C3, 2, 2, 16>(a, 1, st);
    }
    DG_CHECK_LAUNCH("dg_conv4x4s2_c3_fwd");
    return DG_OK;
}
