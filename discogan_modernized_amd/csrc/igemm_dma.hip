// Implicit-GEMM convolution on the bf16 MFMA with BOTH operands staged by LDS-DMA (gfx950).
//
// Replaces (reference file:line) the same ops as igemm.hip -- nn.Conv2d(k4,s2,p1) forward / input-grad / weight-grad
// (model.py:11-31,83-103 via autograd) and nn.ConvTranspose2d(k4,s2,p1) (model.py:118-140) -- for the bf16 matrix path
// (BASELINE configs[4]) when both operands exist as bf16 tensors in HBM (the feature maps themselves with bf16 activation
// storage, or the shadows their producers write: Adam for weights, the BatchNorm kernels / first conv for activations).
//
// Why a second kernel: the register-staged bf16 tiles of igemm.hip (128x128 block, 64x64 per wave, global -> VGPR ->
// ds_write) move 1 KB of LDS reads + 0.5 KB of LDS writes per MFMA and 64 B/clk/CU through the vector memory path.  Here
//   * a workgroup is 8 waves (2 x 4) on a 256 x 256 output tile, 128 x 64 per wave (128 accumulator registers):
//     0.75 KB of LDS reads and 0.25 KB of global -> LDS traffic per 32-cycle MFMA slot;
//   * operand tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (one 1-KiB piece per wave-instruction, no VGPRs, no
//     ds_write, no conversions), issued from inline asm (see `dma` below: through the builtin hipcc serialised every tile on
//     the DMA); the LDS image is lane-linear per piece, so the bank-conflict-free layouts are XOR swizzles applied to the
//     per-lane SOURCE address and again in the fragment reads:
//       k-contiguous images [row][64 k] (128-B rows): 16-B granule g of row r sits in slot g ^ ((r >> 1) & 7)
//           -> the 16 lanes of a ds_read_b128 group hit 16 different 16-B slots of the 256-B bank row (both MFMA shapes);
//       reduction-major images [64 k][cols] (rows of 512 B): granule gc of row k sits in slot gc ^ kmswz(k)
//           -> the rows of a ds_read_b64_tr_b16 half hit different 64-B / 32-B bank segments;
//   * padding / ragged rows are out-of-range buffer offsets (the DMA then writes zeros: tools/probes/lds_dma_oob.hip),
//     exactly as in igemm.hip;
//   * two LDS stages (2 x 64 KB), one workgroup per CU, ONE barrier per K-tile placed 256 cycles of MFMA work before the
//     end of the tile (igemm.hip's pipeline): behind it the first fragments of tile t+1 are fetched and the DMA of tile t+2
//     starts into the stage tile t just vacated, so a tile's DMA has a whole K-tile to land; the only vmcnt wait is the
//     one in front of that barrier;
//   * two loop bodies: v_mfma_f32_16x16x32_bf16 (default: the chip holds 1.81-1.85 GHz under it) and
//     v_mfma_f32_32x32x16_bf16 (1.62-1.73 GHz; option "dma_mfma" 32) -- the loop is power-limited, DESIGN.md section 3.1.
// Epilogue (fp32 or bf16 output), split-K slabs and the blockIdx -> tile orders are those of igemm.hip.
#include "igemm_args.h"
#include <type_traits>
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4* lds_bf4_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

#define DG_NEG_BIG (-(1 << 28))
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) -- `#pragma unroll` gives up on the 64-MFMA
// body of the 16x16x32 K-tile (pragma-unroll-threshold), and a runtime index would put the accumulator blocks in scratch
template <int... Q, typename F>
__device__ __forceinline__ void dg_static_for(std::integer_sequence<int, Q...>, F&& f) {
    (f(std::integral_constant<int, Q>{}), ...);
}
// Fetch schedule of the 16x16x32 body: A blocks fetched M16_FD blocks ahead, the tile barrier M16_BB row groups before the end
// of the tile, the next step's B blocks M16_B0 MFMAs into the step.  Six variants of (4..6, 16..24 MFMAs, early / late) measured
// within +-1 % of each other (same-box A/B, round 2); these are the kept values.
constexpr int M16_FD = 4, M16_BB = 4, M16_B0 = 4;

// M16: the MFMA shape.  false: v_mfma_f32_32x32x16_bf16 (4 x 2 accumulator blocks of 32x32 per wave); true:
// v_mfma_f32_16x16x32_bf16 (8 x 4 blocks of 16x16, the same 128 accumulator registers, the same LDS traffic).  The loop is
// power-limited (the chip holds 1.6-1.9 GHz under it) and the 16x16x32 shape delivers more FLOP/s at equal cycles
// (MI355X_MICROARCH.md, DVFS give-back item 7).
template <int MODE, int WM, int WN, bool M16>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN + 3) / 4) void igemm_dma_kernel(const IgemmArgs p) {
    static_assert(MODE == MODE_FWD || MODE == MODE_DGRAD_S2 || MODE == MODE_WGRAD, "modes with an LDS-DMA form");
    constexpr int NW = WM * WN;
    constexpr int FM = 4, FN = 2;                       // 32x32 accumulator blocks per wave: 128 x 64
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN, KT = 64;
    constexpr bool A_KM = MODE == MODE_WGRAD;           // operand image is reduction-major ([k][rows])
    constexpr bool B_KM = MODE != MODE_FWD;
    constexpr int A_BYTES = BM * KT * 2, B_BYTES = BN * KT * 2;
    constexpr int NPA = A_BYTES / 1024 / NW, NPB = B_BYTES / 1024 / NW;     // 1-KiB DMA pieces per wave and tile
    static_assert(NPA * NW * 1024 == A_BYTES && NPB * NW * 1024 == B_BYTES, "pieces must divide over the waves");
    constexpr int NPC = NPA + NPB;
    // pieces issued behind the tile barrier / at the head of the next tile.  The weight gradient streams both operands from
    // HBM with little reuse and wants every piece as early as possible (same-box A/B: all behind the barrier +3.6 % on the
    // weight gradient, -1.7 % on the forward, whose weight tiles come from L2)
    constexpr int NTAIL = MODE == MODE_WGRAD ? NPC : NPC / 2, NHEAD = NPC - NTAIL;
    constexpr int LDS_BYTES = 2 * (A_BYTES + B_BYTES);   // [A stage 0][A stage 1][B stage 0][B stage 1]
    constexpr int EPI_BYTES = NW * 32 * 68 * 4;
    static_assert(EPI_BYTES <= LDS_BYTES, "epilogue transpose regions live in the operand stages");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long* const stp = (p.stamps != nullptr && tid == 0) ? p.stamps + (long)blockIdx.x * 8 : nullptr;
    if (stp) {
        stp[0] = wall_clock64();
        stp[1] = clock64();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stp[6] = ((long long)xcc << 32) | hwid;
    }
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- blockIdx -> (tile, parity, split): the three orders of igemm.hip -------------------------------------
    int bid = blockIdx.x;
    int tn, tm, parity = 0, split;
    if (p.xcd_group == 2) {
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
    const int it_last = it_end - 1;

    const int H = p.H, W = p.W, Cc = p.Cc, K = p.K, Ho = p.Ho, Wo = p.Wo;
    const int lgWo = p.lgWo, lgHW = p.lgWo + p.lgHo;

    constexpr int OOR = (int)0x80000000;     // any offset with this bit set is beyond a < 2 GiB tensor: the DMA writes zeros
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (p.dbg_zero & 1) ? 0 : (int)p.abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (p.dbg_zero & 2) ? 0 : (int)p.bbytes, 0x00020000);

    // granule swizzle of a reduction-major image row k: the four rows of a transposed-read block go to four different 64-B bank
    // segments; with the 16x16x32 fragments a 32-lane half reads two 4-row blocks 8 rows apart in the same 16 columns, which
    // the extra bit puts into the two 32-B halves of the segment
    auto kmswz = [](int k) -> int { return ((k & 3) << 2) ^ (M16 ? ((k >> 3) & 1) << 1 : 0); };
    // ---- per-lane source descriptors of this wave's DMA pieces (fixed over the K loop) --------------------------
    // k-contiguous image: piece pq covers rows 8 pq .. 8 pq + 7; lane L lands in (row 8 pq + L / 8, slot L % 8) and fetches
    // granule slot ^ swizzle(row).  Reduction-major image of NC columns: GR = NC / 8 granules per row, piece pq covers
    // k rows RP pq .. RP pq + RP - 1 (RP = 64 / GR); lane L lands in (k row RP pq + L / GR, slot L % GR).
    int a_ob[NPA], a_inv[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int pq = wave * NPA + i;
        a_ob[i] = 0;
        a_inv[i] = 0;
        if (!A_KM) {
            const int row = pq * 8 + (lane >> 3);
            const int g = (lane & 7) ^ ((row >> 1) & 7);
            const int m = m0 + row;
            if (MODE == MODE_FWD) {
                a_inv[i] = 0xFFFF;
                if (m < p.M) {
                    const int ox = m & (Wo - 1), oy = (m >> lgWo) & (Ho - 1), n = m >> lgHW;
                    const int ay = oy * p.stride - p.pad, ax = ox * p.stride - p.pad;
                    a_ob[i] = (((n * H + ay) * W + ax) * Cc + g * 8) * 2;
                    int colok = 0, okmask = 0;
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) colok |= ((unsigned)(ax + sx) < (unsigned)W) ? (1 << sx) : 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) okmask |= ((unsigned)(ay + r) < (unsigned)H) ? (colok << (4 * r)) : 0;
                    a_inv[i] = ~okmask & 0xFFFF;
                }
            } else {   // DGRAD_S2
                a_inv[i] = 0xF;
                if (m < p.M) {
                    const int bx = m & (Wo - 1), ay = (m >> lgWo) & (Ho - 1);
                    a_ob[i] = (m * K + g * 8) * 2;
                    a_inv[i] = 0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int ty = t >> 1, tx = t & 1;
                        const int dyo = ph == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 0 : 1);
                        const int dxo = pw == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 0 : 1);
                        const bool ok = (unsigned)(ay + dyo) < (unsigned)Ho && (unsigned)(bx + dxo) < (unsigned)Wo;
                        a_inv[i] |= ok ? 0 : (1 << t);
                    }
                }
            }
        } else {       // WGRAD: rows = reduction pixels, columns = out channels m0 .. m0 + BM - 1 of dy[pixel][K]
            constexpr int GR = BM / 8, RP = 64 / GR;
            const int krow = pq * RP + lane / GR;
            const int gc = (lane % GR) ^ kmswz(krow);
            const int col = m0 + gc * 8;
            a_ob[i] = col < K ? (krow * K + col) * 2 : OOR;     // pixel rows >= R run off the end of the tensor: zeros
        }
    }
    int b_ob[NPB];
    // WGRAD: the columns of B are (tap, c) of im2col(x); everything about the column is fixed per lane and piece
    int wg_c[NPB], wg_cst[NPB], wg_ybad[NPB], wg_xbad[NPB], wg_colbad[NPB], wg_krow[NPB];
    const bool wg_s2 = p.stride == 2;
    const int wg_lpm = wg_s2 ? 2 : 4, wg_pxm = wg_s2 ? -1 : 0;
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int pq = wave * NPB + i;
        b_ob[i] = 0;
        wg_c[i] = wg_cst[i] = wg_colbad[i] = wg_krow[i] = 0;
        wg_ybad[i] = wg_xbad[i] = -1;
        if (MODE == MODE_FWD) {
            const int row = pq * 8 + (lane >> 3);
            const int g = (lane & 7) ^ ((row >> 1) & 7);
            const int k = n0 + row;
            b_ob[i] = k < K ? (k * 16 * Cc + g * 8) * 2 : OOR;
        } else {
            constexpr int GR = BN / 8, RP = 64 / GR;
            const int krow = pq * RP + lane / GR;
            const int gc = (lane % GR) ^ kmswz(krow);
            const int col = n0 + gc * 8;
            if (MODE == MODE_DGRAD_S2) {
                b_ob[i] = col < Cc ? (krow * 16 * Cc + col) * 2 : OOR;
            } else {
                const bool colok = col < p.Ng;
                const int tap = colok ? col / Cc : 0;
                const int r = tap >> 2, s = tap & 3;
                wg_c[i] = col - tap * Cc;
                wg_cst[i] = wg_s2 ? (r - 1) * W + (s - 1) : r * 4 + s;
                wg_ybad[i] = !wg_s2 ? -1 : (r == 0 ? 0 : (r == 3 ? Ho - 1 : -1));
                wg_xbad[i] = !wg_s2 ? -1 : (s == 0 ? 0 : (s == 3 ? Wo - 1 : -1));
                wg_colbad[i] = colok ? 0 : -1;
                wg_krow[i] = krow;
            }
        }
    }

    // ---- DMA-side K-iteration state (wave-uniform): the tile the NEXT piece belongs to ----------------------------
    // FWD walks the reduction channel-chunk major with the 16 taps inner in the order r, s in (0, 2, 1, 3) (igemm.hip);
    // DGRAD_S2 chunk major with the 2x2 taps of the parity class inner; WGRAD walks pixel tiles.
    int dt = it_begin, tap = 0, chunk = 0;
    if (MODE == MODE_FWD) {
        chunk = it_begin >> 4;
        tap = it_begin & 15;
    } else if (MODE == MODE_DGRAD_S2) {
        chunk = it_begin >> 2;
        tap = it_begin & 3;
    }
    auto advance = [&]() {                       // next tile, clamped to the last one (re-loaded, never used)
        const int go = dt + 1 < it_end ? 1 : 0;
        dt += go;
        if (MODE == MODE_FWD) {
            tap += go;
            const int wrap = (tap == 16) ? 1 : 0;
            tap = wrap ? 0 : tap;
            chunk += wrap;
        } else if (MODE == MODE_DGRAD_S2) {
            tap += go;
            const int wrap = (tap == 4) ? 1 : 0;
            tap = wrap ? 0 : tap;
            chunk += wrap;
        }
    };
    auto fwd_r = [&]() { const int a = tap >> 2; return ((a & 1) << 1) | (a >> 1); };
    auto fwd_s = [&]() { const int b = tap & 3; return ((b & 1) << 1) | (b >> 1); };

    // The LDS-DMA is issued from inline asm so that hipcc does NOT track it: with the builtin the compiler put an
    // `s_waitcnt vmcnt(0)` in front of the first ds_read_b64_tr_b16 of every K-tile (it cannot tell the transposed reads from the
    // stage being filled), i.e. every tile waited for the whole DMA of the next one (weight-grad K loop at 63 % of the MFMA rate).
    // Ordering is ours: the counted wait + barrier in front of the first read of a stage.  M0 (the LDS destination) is saved
    // and restored around the instruction.
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_ptr)smem;
    auto dma = [&](const __amdgpu_buffer_rsrc_t& r, int lds_off, int voff) {
        unsigned keep;
        const unsigned dst = lds_base + (unsigned)lds_off;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(r), "s"(dst)
                     : "memory");
    };
    // piece k (0 .. NPC-1: A pieces first) of tile `dt` into LDS stage `stage`
    auto issue = [&](int stage, int k) {
        if (k < NPA) {
            const int i = k;
            const int lds_off = stage * A_BYTES + (wave * NPA + i) * 1024;
            int voff;
            if (MODE == MODE_FWD) {
                const int r = fwd_r(), sx = fwd_s();
                const int soff = ((r * W + sx) * Cc + chunk * KT) * 2;                 // wave-uniform
                voff = (a_ob[i] + soff) | -((a_inv[i] >> (r * 4 + sx)) & 1);
            } else if (MODE == MODE_DGRAD_S2) {
                const int ty = tap >> 1, tx = tap & 1;
                const int dyo = ph == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 0 : 1);
                const int dxo = pw == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 0 : 1);
                const int soff = ((dyo * Wo + dxo) * K + chunk * KT) * 2;
                voff = (a_ob[i] + soff) | -((a_inv[i] >> tap) & 1);
            } else {
                voff = a_ob[i] + dt * (KT * 2) * K;
            }
            dma(rA, lds_off, voff);
        } else {
            const int i = k - NPA;
            const int lds_off = 2 * A_BYTES + stage * B_BYTES + (wave * NPB + i) * 1024;
            int voff;
            if (MODE == MODE_FWD) {
                voff = b_ob[i] + (((fwd_r() * 4 + fwd_s()) * Cc) + chunk * KT) * 2;
            } else if (MODE == MODE_DGRAD_S2) {
                const int ty = tap >> 1, tx = tap & 1;
                const int r = ph == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 2 : 0);
                const int sx = pw == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 2 : 0);
                voff = b_ob[i] + ((chunk * KT * 16 + r * 4 + sx) * Cc) * 2;
            } else {
                // reduction row = output pixel mrow = (n, oy, ox) packed; see igemm.hip load_B (WGRAD)
                const int mrow = dt * KT + wg_krow[i];
                const int oxv = mrow & (Wo - 1), oyv = (mrow >> lgWo) & (Ho - 1);
                const int bad = (oyv == wg_ybad[i]) | (oxv == wg_xbad[i]) | (mrow >= p.R);
                const int pix = (mrow << wg_lpm) - ((oxv << 1) & wg_pxm) + wg_cst[i];
                voff = ((pix * Cc + wg_c[i]) * 2) | wg_colbad[i] | -bad;
            }
            dma(rB, lds_off, voff);
        }
    };

    // ---- fragment reads -------------------------------------------------------------------------------------------
    const int tr_q = (lane >> 2) & 3, tr_c = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    auto frag_kc = [&](const char* img, int row, int g) -> bf16x8 {
        return *(const bf16x8*)(img + row * 128 + ((g ^ ((row >> 1) & 7)) << 4));
    };
    // transposed read (igemm.hip frag_km) from the swizzled reduction-major image; rowb = bytes per k row
    auto frag_km = [&](const char* img, int rowb, int k0, int c0) -> bf16x8 {
        const int kr = k0 + tr_q, col = c0 + tr_c;
        const char* p0 = img + kr * rowb + ((((col >> 3) ^ kmswz(kr))) << 4) + (col & 7) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)p0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(p0 + 4 * rowb));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // 16x16x32 operand: lane l holds row / column (l & 15), k = 8 (l >> 4) + j of the 32-deep step.  k-contiguous image: one
    // ds_read_b128 (granule 4 s2 + (l >> 4)); reduction-major image: the 16-lane group l >> 4 transposes the 4-row blocks at
    // k rows 32 s2 + 8 (l >> 4) and + 4, columns c0 .. c0 + 15
    const int l15 = lane & 15, l4 = lane >> 4;
    auto frag16_kc = [&](const char* img, int row0, int s2) -> bf16x8 {
        const int row = row0 + l15;
        return *(const bf16x8*)(img + row * 128 + (((4 * s2 + l4) ^ ((row >> 1) & 7)) << 4));
    };
    auto frag16_km = [&](const char* img, int rowb, int s2, int c0) -> bf16x8 {
        const int kr = 32 * s2 + 8 * l4 + tr_q, col = c0 + (lane & 3) * 4;
        const char* p0 = img + kr * rowb + ((((col >> 3) ^ kmswz(kr))) << 4) + (col & 7) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)p0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(p0 + 4 * rowb));     // kr + 4: same swizzle
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 fa[2][FM], fb[2][FN];
    // fragment f of k16 step s of the tile in `stage`, in the order the MFMAs want them: A0, B0, B1, A1, A2, A3
    auto fetch1 = [&](int stage, int s, int set, int f) {
        const char* As = smem + stage * A_BYTES;
        const char* Bs = smem + 2 * A_BYTES + stage * B_BYTES;
        const bool isB = f >= 1 && f <= FN;
        if (!isB) {
            const int i = f == 0 ? 0 : f - FN;
            const int row = wm * (32 * FM) + i * 32;
            if (!A_KM) fa[set][i] = frag_kc(As, row + l31, 2 * s + lh);
            else fa[set][i] = frag_km(As, BM * 2, s * 16 + 8 * lh, row);
        } else {
            const int j = f - 1;
            const int col = wn * (32 * FN) + j * 32;
            if (!B_KM) fb[set][j] = frag_kc(Bs, col + l31, 2 * s + lh);
            else fb[set][j] = frag_km(Bs, BN * 2, s * 16 + 8 * lh, col);
        }
    };
    auto fetch = [&](int stage, int s, int set) {
#pragma unroll
        for (int f = 0; f < FM + FN; ++f) fetch1(stage, s, set, f);
    };
    // M16 fragments: the A blocks of a k32 step sit in 8 slots (block i in slot i, fetched 4 blocks ahead), the B blocks of a
    // step in one of two sets (fetched during blocks 4, 5 of the step before)
    constexpr int AM = 2 * FM, BNB = 2 * FN;          // 16-row / 16-column blocks per wave: 8 x 4
    bf16x8 ga[AM], gb[2][BNB];
    auto fetchA16 = [&](int stage, int s2, int i) {
        const char* As = smem + stage * A_BYTES;
        const int row0 = wm * (32 * FM) + i * 16;
        if (!A_KM) ga[i] = frag16_kc(As, row0, s2);
        else ga[i] = frag16_km(As, BM * 2, s2, row0);
    };
    auto fetchB16 = [&](int stage, int s2, int set, int j) {
        const char* Bs = smem + 2 * A_BYTES + stage * B_BYTES;
        const int col0 = wn * (32 * FN) + j * 16;
        if (!B_KM) gb[set][j] = frag16_kc(Bs, col0, s2);
        else gb[set][j] = frag16_km(Bs, BN * 2, s2, col0);
    };

    f32x16 acc[M16 ? 1 : FM][M16 ? 1 : FN];
    f32x4 acc16[M16 ? AM : 1][M16 ? BNB : 1];
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < AM; ++i)
#pragma unroll
            for (int j = 0; j < BNB; ++j) acc16[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    // ---- prologue: tile 0 entirely, the first half of tile 1 --------------------------------------------------------
    if (it_begin < it_end) {
#pragma unroll
        for (int k = 0; k < NPC; ++k) issue(0, k);
        advance();
#pragma unroll
        for (int k = 0; k < NTAIL; ++k) issue(1, k);
        if constexpr (NHEAD == 0) advance();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NTAIL) : "memory");      // tile 0 has landed (this wave's pieces)
    }
    __builtin_amdgcn_s_barrier();
    if (stp) stp[2] = clock64();
    if constexpr (M16) {
#pragma unroll
        for (int j = 0; j < BNB; ++j) fetchB16(0, 0, 0, j);
#pragma unroll
        for (int i = 0; i < M16_FD; ++i) fetchA16(0, 0, i);
    } else {
        fetch(0, 0, 0);
    }

    // ---- one K-tile (32 MFMAs per wave): ST = LDS stage of the current tile ------------------------------------------
    constexpr int NMF = 4 * FM * FN;                       // 4 k16 steps x 8 accumulator blocks
    constexpr int PER = FM * FN;
    constexpr int QB = NMF - PER;                          // the tile barrier comes before MFMA QB (the last k16 step)
    auto body = [&](auto ST_) {
        constexpr int ST = decltype(ST_)::value;
#pragma unroll
        for (int q = 0; q < NMF; ++q) {
            const int s = q / PER, w = q % PER;
            __builtin_amdgcn_sched_barrier(0);
            if (q == QB) {
                // this wave's DMA pieces of tile t+1 have landed and its fragment reads of tile t are complete; behind
                // the barrier that holds for every wave: tile t+1 may be read, tile t's stage may be overwritten
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            // the next k16 step's fragments (behind the barrier: the next tile's first step), one fragment per MFMA gap
            // (one fragment per MFMA gap instead of a whole step's worth here made the weight gradient 25 % slower: round 2 A/B)
            if (q == QB) fetch(ST ^ 1, 0, 0);
            if (w == 1 && s + 1 < 4) fetch(ST, s + 1, (s + 1) & 1);
            acc[w / FN][w % FN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][w / FN], fb[s & 1][w % FN], acc[w / FN][w % FN], 0, 0, 0);
            // DMA pieces: the second half of tile t+1 behind the first PER MFMAs, the first half of tile t+2 behind the
            // last PER (after the barrier: into the stage tile t is leaving)
            if constexpr (NHEAD > 0) {
                if (q < PER) {
#pragma unroll
                    for (int k = 0; k < NHEAD; ++k)
                        if (k * PER / NHEAD == q) issue(ST ^ 1, NTAIL + k);
                    if (q == (NHEAD - 1) * PER / NHEAD) advance();
                }
            }
            if (q >= QB) {
#pragma unroll
                for (int k = 0; k < NTAIL; ++k)
                    if (QB + k * PER / NTAIL == q) issue(ST, k);
                if constexpr (NHEAD == 0) {
                    if (q == NMF - 1) advance();
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- M16: one K-tile = 2 k32 steps x (8 x 4) blocks = 64 MFMAs of 16 cycles.  Block (s2, i) runs its 4 MFMAs on A slot i
    // and B set s2; behind its first MFMA the A block 4 ahead is fetched (the next step's, the next TILE's behind the barrier),
    // blocks 4 and 5 also fetch the next step's B blocks.  The tile barrier sits in front of block (1, 4) = 16 MFMAs (256
    // cycles) before the end of the tile: every read of the current stage was issued by block (1, 3).
    // Schedule parameters (same-box A/B, tools/bench_ops.py): FD = how many blocks ahead an A block is fetched (8 slots: <= 7),
    // BB = the block of step 1 the tile barrier sits in front of (every read of the current stage must have been issued: the
    // last one, A(1, 7), is fetched in block (1, 7 - FD) < BB), B0 = the block of step 0 that starts fetching step 1's B blocks.
    auto body16 = [&](auto ST_) {
        constexpr int ST = decltype(ST_)::value;
        constexpr int FD = M16_FD, BB = M16_BB, B0 = M16_B0;
        static_assert(FD >= 1 && FD <= 7 && 7 - FD < BB && BB <= 6 && B0 <= 6, "M16 schedule");
        constexpr int NMF16 = 2 * AM * BNB, QB16 = AM * BNB + BB * BNB;
        dg_static_for(std::make_integer_sequence<int, NMF16>{}, [&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            constexpr int s2 = q / (AM * BNB), i = (q / BNB) % AM, j = q % BNB;
            __builtin_amdgcn_sched_barrier(0);
            if (q == QB16) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (j == 0) {                                     // A block FD ahead
                const int ni = (i + FD) % AM, ns2 = s2 + (i + FD) / AM;          // ns2 == 2: step 0 of the next tile
                fetchA16(ns2 < 2 ? ST : ST ^ 1, ns2 & 1, ni);
            }
            constexpr int BF = s2 == 0 ? B0 : BB;             // the next step's B blocks into the other set (next tile: behind the barrier)
            if (j == 1 && (i == BF || i == BF + 1)) {
                const int ns2 = s2 + 1;
                fetchB16(ns2 < 2 ? ST : ST ^ 1, ns2 & 1, ns2 & 1, 2 * (i - BF));
                fetchB16(ns2 < 2 ? ST : ST ^ 1, ns2 & 1, ns2 & 1, 2 * (i - BF) + 1);
            }
            acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[i], gb[s2][j], acc16[i][j], 0, 0, 0);
            // DMA pieces: as in the 32x32 body, scaled to 64 MFMAs per tile
            if constexpr (NHEAD > 0) {
                if (q < 2 * PER) {
#pragma unroll
                    for (int k = 0; k < NHEAD; ++k)
                        if (k * 2 * PER / NHEAD == q) issue(ST ^ 1, NTAIL + k);
                    if (q == (NHEAD - 1) * 2 * PER / NHEAD) advance();
                }
            }
            if (q >= QB16) {
#pragma unroll
                for (int k = 0; k < NTAIL; ++k)
                    if (QB16 + k * (NMF16 - QB16) / NTAIL == q) issue(ST, k);
                if constexpr (NHEAD == 0) {
                    if (q == NMF16 - 1) advance();
                }
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int it = it_begin; it < it_end; it += 2) {
        if constexpr (M16) {
            body16(std::integral_constant<int, 0>{});
            if (it + 1 < it_end) body16(std::integral_constant<int, 1>{});
        } else {
            body(std::integral_constant<int, 0>{});
            if (it + 1 < it_end) body(std::integral_constant<int, 1>{});
        }
    }
    // the clamped re-loads of the last tile and the fragment prefetch behind the last barrier still touch LDS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (stp) stp[3] = clock64();

    // ---- fused BatchNorm statistics (igemm.hip's scheme): one partial row per (tile, wave row) = 32 FM output rows, shifted by the
    //      wave tile's first row; merged by bn_partials_finalize.  ~3 VALU instructions per accumulator value: invisible next to a
    //      K loop of hundreds of microseconds, and it removes a whole read pass over the conv output (dg_bn_train_stats).
    if ((MODE == MODE_FWD || MODE == MODE_DGRAD_S2) && p.stat != nullptr && p.part == nullptr) {
        const int row0 = m0 + wm * (32 * FM);
        const int nrows = min(32 * FM, p.M - row0);
        const int prow = ((MODE == MODE_DGRAD_S2 ? parity : 0) * p.tilesM + tm) * WM + wm;
        float* srow = p.stat + (long)prow * p.stat_rs;
        if (wn == 0 && tn == 0 && lane == 0) srow[0] = (float)max(nrows, 0);
        if constexpr (M16) {          // acc16[bi][bj][e] = row 16 bi + 4 l4 + e, column 16 bj + l15 of the wave tile
            if (nrows > 0) {
#pragma unroll
                for (int bj = 0; bj < BNB; ++bj) {
                    const float sh = __shfl(acc16[0][bj][0], l15, 64);
                    float ssum = 0.f, ssq = 0.f;
#pragma unroll
                    for (int bi = 0; bi < AM; ++bi)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int lr = bi * 16 + 4 * l4 + e;
                            if (lr < nrows) {
                                const float d = acc16[bi][bj][e] - sh;
                                ssum += d;
                                ssq += d * d;
                            }
                        }
                    ssum += __shfl_xor(ssum, 16, 64);
                    ssq += __shfl_xor(ssq, 16, 64);
                    ssum += __shfl_xor(ssum, 32, 64);
                    ssq += __shfl_xor(ssq, 32, 64);
                    const int n = n0 + wn * (32 * FN) + bj * 16 + l15;
                    if (l4 == 0 && n < p.Ng) {
                        srow[4 + n] = sh;
                        srow[4 + p.Ng + n] = ssum;
                        srow[4 + 2 * p.Ng + n] = ssq;
                    }
                }
            }
        } else
        if (nrows > 0) {
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) {
                const float sh = __shfl(acc[0][jn][0], l31, 64);
                float ssum = 0.f, ssq = 0.f;
#pragma unroll
                for (int i = 0; i < FM; ++i)
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
                const int n = n0 + wn * (32 * FN) + jn * 32 + l31;
                if (lh == 0 && n < p.Ng) {
                    srow[4 + n] = sh;
                    srow[4 + p.Ng + n] = ssum;
                    srow[4 + 2 * p.Ng + n] = ssq;
                }
            }
        }
    }

    // ---- epilogue (igemm.hip): acc[i][jn][r] = row (r&3)+8*(r>>2)+4*lh, col jn*32+l31 of the wave's 32x64 block i,
    // transposed through a private [32][68] LDS region per wave, float4 stores with 16 lanes per 256-B row segment
    const bool to_part = p.part != nullptr;
    float* const eps = (float*)smem + wave * (32 * 68);
    const int erow = lane >> 4, ec4 = (lane & 15) * 4;
    const int ncol = n0 + wn * (32 * FN) + ec4;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        if constexpr (M16) {      // 16x16 blocks: row 4 (lane >> 4) + e, column lane & 15; rows 32 i .. 32 i + 31 = block rows 2i, 2i + 1
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < BNB; ++bj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) eps[(bi * 16 + 4 * l4 + e) * 68 + bj * 16 + l15] = acc16[2 * i + bi][bj][e];
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            eps[lr * 68 + l31] = acc[i][0][r];
            eps[lr * 68 + 32 + l31] = acc[i][1][r];
        }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = t * 4 + erow;
            f32x4 v = *(const f32x4*)(eps + row * 68 + ec4);
            const int m = m0 + wm * (32 * FM) + i * 32 + row;
            if (m >= p.M || ncol >= p.Ng) continue;
            float* dst;
            long eoff;
            if (to_part) {
                const long srow = (MODE == MODE_DGRAD_S2) ? ((long)split * 4 + parity) * p.M + m : (long)split * p.M + m;
                dst = p.part;
                eoff = srow * p.Ng + ncol;
            } else if (MODE == MODE_DGRAD_S2) {
                const int b = m & (Wo - 1), a = (m >> lgWo) & (Ho - 1), n = m >> lgHW;
                dst = p.C;
                eoff = (long)((n * H + 2 * a + ph) * W + 2 * b + pw) * Cc + ncol;
            } else {
                dst = p.C;
                eoff = (long)m * p.Ng + ncol;
            }
            if (!to_part && p.accumulate) v += *(const f32x4*)(dst + eoff);
            dg_store_out4(dst, eoff, v, (!to_part && MODE != MODE_WGRAD) ? p.out16 : 0);
        }
    }
    if (stp) {
        stp[4] = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stp[5] = wall_clock64();
        stp[7] = clock64();
    }
}

// host: launch the LDS-DMA kernel for a plan made by igemm.hip (mode, args); returns 0 when there is no instantiation
int dg_igemm_dma_launch(int mode, const IgemmArgs& a, int zmul, hipStream_t st) {
    const int grid = a.tilesM * a.tilesN * zmul * a.splits;
    if (dg_get_option(DG_OPT_DMA_MFMA) != 32) {     // default: 16x16x32; option "dma_mfma" 32 selects the 32x32x16 body
        switch (mode) {
            case MODE_FWD: hipLaunchKernelGGL((igemm_dma_kernel<MODE_FWD, 2, 4, true>), dim3(grid), dim3(512), 0, st, a); return 1;
            case MODE_DGRAD_S2: hipLaunchKernelGGL((igemm_dma_kernel<MODE_DGRAD_S2, 2, 4, true>), dim3(grid), dim3(512), 0, st, a); return 1;
            case MODE_WGRAD: hipLaunchKernelGGL((igemm_dma_kernel<MODE_WGRAD, 2, 4, true>), dim3(grid), dim3(512), 0, st, a); return 1;
            default: return 0;
        }
    }
    switch (mode) {
        case MODE_FWD: hipLaunchKernelGGL((igemm_dma_kernel<MODE_FWD, 2, 4, false>), dim3(grid), dim3(512), 0, st, a); return 1;
        case MODE_DGRAD_S2: hipLaunchKernelGGL((igemm_dma_kernel<MODE_DGRAD_S2, 2, 4, false>), dim3(grid), dim3(512), 0, st, a); return 1;
        case MODE_WGRAD: hipLaunchKernelGGL((igemm_dma_kernel<MODE_WGRAD, 2, 4, false>), dim3(grid), dim3(512), 0, st, a); return 1;
        default: return 0;
    }
}
