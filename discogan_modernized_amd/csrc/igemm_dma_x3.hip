// fp32-accurate implicit-GEMM convolution on the bf16 MFMA, operands = THREE bf16 PLANES per tensor staged by LDS-DMA (gfx950).
//
// Replaces (reference file:line) the same ops as igemm.hip / igemm_dma.hip -- nn.Conv2d(k4,s2,p1) forward / input-grad /
// weight-grad (model.py:11-31,83-103 via autograd) and nn.ConvTranspose2d(k4,s2,p1) (model.py:118-140) -- for the
// `mfma_dtype="f32x3"` matrix path (option "bf16" = 2) when BOTH operands exist in HBM as plane triples
//     hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid)          (plane-major: [3][numel] bf16; dg_f32_to_bf16x3)
// -- 24 significand bits of v, both subtractions exact in fp32.  A product block is the same SIX bf16 MFMAs as the
// register-staged PREC 2 tiles of igemm.hip (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi into one fp32 accumulator,
// smallest first); where the reduction is also walked in the same order (weight gradient; channel counts that are not a
// multiple of 64) the two kernels agree bit for bit on an unsplit GEMM.
//
// Why a second kernel: the register-staged form splits every fp32 value on the VALU while it is staged (5.5 instructions per
// element) and its two waves per SIMD spend the SIMD's whole issue budget on that -- K loop at 73 % of the matrix rate.  Here
// the split is done ONCE by whoever produces the tensor, and the conv kernel is igemm_dma.hip's pipeline with 3 planes:
//   * 256 x 256 output tile, 8 waves (2 x 4) of 128 x 64; K-tile = 16 reduction elements = 48 MFMAs (1536 matrix cycles) per
//     wave; a fragment (3 planes) is reused by 6 MFMAs from registers: 18 ds_read_b128 (36 ds_read_b64_tr_b16 for
//     reduction-major images) and 6 DMA instructions per wave and tile -- 0.19 KB of LDS reads and 16 B/clk/CU of
//     global -> LDS traffic per MFMA slot (bf16 kernel: 0.75 KB, 32 B/clk/CU);
//   * LDS stage = [A p0][A p1][A p2][B p0][B p1][B p2], 8 KB each (256 rows x 16 k), two stages (96 KB, one workgroup
//     per CU); every wave DMAs one 1-KiB piece per plane and operand;
//       k-contiguous image [row][16 k] (32-B rows): 16-B granule g of row r sits in slot g ^ ((r >> 3) & 1);
//       reduction-major image [16 k][256 cols] (512-B rows): granule gc of row k sits in slot gc ^ ((k & 3) << 2);
//   * fragments live in ONE register set (A: 48, B: 24 + 8 VGPRs) and are replaced as the MFMA sequence releases them: the
//     A blocks of row i behind row i, B lo / mid during the last row, B hi (first operand of the next tile) in a second set;
//   * ONE barrier per K-tile in front of MFMA 8: behind it tile t+1 may be read and the DMA of tile t+2 starts into the
//     stage tile t has left (every fragment of tile t is in registers since the end of tile t-1).
// Epilogue, split-K slabs and the blockIdx -> tile orders are those of igemm.hip.
#include "igemm_args.h"
#include <type_traits>
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4* lds_bf4_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int... Q, typename F>
__device__ __forceinline__ void dg_x3_static_for(std::integer_sequence<int, Q...>, F&& f) {
    (f(std::integral_constant<int, Q>{}), ...);
}
#define DG_X3T_MAX 48
// 16x16x32 body: (block row, pair type) of the 24 MFMA groups of a K-tile, in issue order
constexpr int G16_I[24] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 6, 7, 6, 6, 7, 7};
constexpr int G16_T[24] = {0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 0, 1, 2, 1, 2};

// BT (FWD only): the weight planes are the TRANSPOSED copy wT[(r, s, c)][k] (dg_x3_transpose_planes): the B tile of a K-tile is
// 16 reduction rows x 256 contiguous out channels (512-byte pieces) instead of 256 rows x 32 bytes 16 C elements apart.
// WM x WN waves of 128 x 64: 2 x 4 = the 256 x 256 tile (8 waves, 96 KB of LDS, one workgroup per CU); 1 x 4 = 128 x 256 for a
// weight gradient of 96..191 rows (4 waves, 72 KB, TWO workgroups per CU).  2 x 2 = 256 x 128 also compiles; igemm.hip does
// not plan it (measured slower than the register-staged tiles: see make_plan).
// NS: LDS stages (2 or 3).  With 3 the DMA of a tile has TWO tile periods to land (the 256 x 256 tile: 144 KB of LDS, 246-254
// VGPRs); same-box A/B at 512 px / batch 32: 223-225 / 220-221 / 250 TFLOP/s forward / input-grad / weight-grad with either
// -- the loop waits for the (power-limited) matrix cores, not for the DMA -- so two stages are what is instantiated.
// M16: the MFMA shape of the loop body.  false: six v_mfma_f32_32x32x16_bf16 per 32 x 32 block and K-tile (one per plane product).
// true: THREE v_mfma_f32_16x16x32_bf16 per 16 x 16 block, each 32-deep step pairing two plane products along k --
//   [a_lo | a_hi] . [b_hi | b_lo],  [a_mid | a_mid] . [b_mid | b_hi],  [a_hi | a_hi] . [b_mid | b_hi]
// (a lane's 8 k values come from ONE plane: k groups 0, 1 of the lane quarter l >> 4 read the first plane of a pair, groups 2, 3 the
// second, both at the same 16 k of the tile, so a fragment is still one ds_read_b128 / two transposing reads with a lane-dependent
// plane base).  Same six products, same FLOPs, 1.8x the LDS reads; under it the chip holds 1.9 instead of 1.7 GHz
// (tools/probes/x3_body_shapes.hip: +4 % in the bare loop).  A tile's fragments no longer fit the registers (128): the A fragments
// go through a ring six deep, the tile barrier moves to the last fifth of the tile and the DMA gets a third LDS stage to land in.
template <int MODE, bool BT, int WM, int WN, int NS, bool M16 = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void igemm_dma_x3_kernel(const IgemmArgs p) {
    static_assert(MODE == MODE_FWD || MODE == MODE_DGRAD_S2 || MODE == MODE_WGRAD, "modes with an LDS-DMA form");
    static_assert(!BT || MODE == MODE_FWD, "the transposed weight copy serves the forward form");
    constexpr int NW = WM * WN, FM = 4, FN = 2;         // 4 x 2 accumulator blocks of 32x32 per wave
    constexpr int BM = 32 * FM * WM, BN = 32 * FN * WN, KT = 16;
    static_assert((BM == 256 || BM == 128) && (BN == 256 || BN == 128), "tile");
    constexpr bool A_KM = MODE == MODE_WGRAD;           // operand image is reduction-major ([k][cols])
    constexpr bool B_KM = MODE != MODE_FWD || BT;
    constexpr int PLA = BM * KT * 2, PLB = BN * KT * 2; // one plane of an operand tile: 8 KB per 256 rows
    constexpr int OPA = 3 * PLA, STAGE = OPA + 3 * PLB; // stage = [A p0][A p1][A p2][B p0][B p1][B p2]
    constexpr int NPA = PLA / 1024 / NW, NPB = PLB / 1024 / NW;     // 1-KiB DMA pieces per wave, plane and tile
    static_assert(NPA * NW * 1024 == PLA && NPB * NW * 1024 == PLB, "pieces must divide over the waves");
    constexpr int NPC = 3 * (NPA + NPB);
    constexpr int LDS_BYTES = NS * STAGE;
    static_assert((NS == 2 || NS == 3) && LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(!M16 || (NS == 3 && WM == 2 && WN == 4), "the 16x16x32 body: 256 x 256 tile, three stages");
    constexpr int EPI_BYTES = NW * 32 * 68 * 4;
    static_assert(EPI_BYTES <= LDS_BYTES, "epilogue transpose regions live in the operand stages");
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

    const int H = p.H, W = p.W, Cc = p.Cc, K = p.K, Ho = p.Ho, Wo = p.Wo;
    const int lgWo = p.lgWo, lgHW = p.lgWo + p.lgHo;

    constexpr int OOR = (int)0x80000000;     // any offset with this bit set is beyond a < 2 GiB plane: the DMA writes zeros
    // one buffer descriptor per plane: an offset that runs off the end of a plane must not land in the next one
    __amdgpu_buffer_rsrc_t rA[3], rB[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        rA[pl] = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + pl * p.a_plane), 0, (p.dbg_zero & 1) ? 0 : (int)p.abytes, 0x00020000);
        rB[pl] = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + pl * p.b_plane), 0, (p.dbg_zero & 2) ? 0 : (int)p.bbytes, 0x00020000);
    }

    // granule swizzle of a reduction-major image by k row.  32x32 body: a transposing read covers k rows 8 lh + 0..3 x 64 B; the
    // 16x16 body's 32-lane group covers k rows 0..3 and 8..11 x 32 B, so bit 3 of the row must move the granule too
    auto kmswz = [](int k) -> int { return M16 ? (((k & 3) << 2) | (((k >> 3) & 1) << 1)) : ((k & 3) << 2); };
    // ---- per-lane source descriptors of this wave's DMA pieces (the same for the three planes; fixed over the K loop) ----
    // k-contiguous image: piece pq covers rows 32 pq .. 32 pq + 31; lane L lands in (row 32 pq + L / 2, slot L % 2) and fetches
    // granule slot ^ ((row >> 3) & 1) (16x16 body: granule slot -- its 16-row reads are conflict-free on the plain image).  Reduction-major image of NC columns: GR = NC / 8 granules per row, piece pq covers k rows
    // RP pq .. RP pq + RP - 1 (RP = 64 / GR); lane L lands in (k row RP pq + L / GR, slot L % GR) and fetches granule
    // slot ^ kmswz(k row).  This wave owns pieces wave * NP + i of every plane.
    int a_ob[NPA], a_inv[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int pq = wave * NPA + i;
        a_ob[i] = 0;
        a_inv[i] = 0;
        if (!A_KM) {
            const int row = pq * 32 + (lane >> 1);
            const int g = (lane & 1) ^ (M16 ? 0 : ((row >> 3) & 1));
            const int m = m0 + row;
            if (MODE == MODE_FWD) {
                a_inv[i] = 0xFFFF;
                if (m < p.M) {
                    const int ox = m & (Wo - 1), oy = (m >> lgWo) & (Ho - 1), n = m >> lgHW;
                    const int ay = oy * p.stride - p.pad, ax = ox * p.stride - p.pad;
                    a_ob[i] = (((n * H + ay) * W + ax) * Cc + g * 8) * 2;
#ifdef DG_TIMING_KNOBS
                    // timing experiment (dbg_zero bit 4, WRONG results): the A rows of a K-tile are one contiguous run of 32-byte
                    // pieces -- what a parity-split chunk-major plane layout would give the forward form
                    if (p.dbg_zero & 16) a_ob[i] = (m * 16 + g * 8) * 2;
#endif
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
            // pixel-major dy[pixel][K]: pixel rows >= R run off the end of the plane (zeros).  Quad-chunk planes (a_cm)
            // [R / 4][K / 16][4][16]: the same 16-pixel x K block of bytes, permuted inside -- the tile stride is unchanged
            a_ob[i] = col < K ? (p.a_cm ? (((krow >> 2) * (K >> 4) + (col >> 4)) * 64 + (krow & 3) * 16 + (col & 15)) * 2 : (krow * K + col) * 2) : OOR;
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
        if (MODE == MODE_FWD && !BT) {
            const int row = pq * 32 + (lane >> 1);
            const int g = (lane & 1) ^ (M16 ? 0 : ((row >> 3) & 1));
            const int k = n0 + row;
            b_ob[i] = k < K ? (k * 16 * Cc + g * 8) * 2 : OOR;
        } else {
            constexpr int GR = BN / 8, RP = 64 / GR;
            const int krow = pq * RP + lane / GR;
            const int gc = (lane % GR) ^ kmswz(krow);
            const int col = n0 + gc * 8;
            if (MODE == MODE_FWD) {
                b_ob[i] = col < K ? (krow * K + col) * 2 : OOR;
            } else if (MODE == MODE_DGRAD_S2) {
                b_ob[i] = col < Cc ? (krow * 16 * Cc + col) * 2 : OOR;
            } else {
                const bool colok = col < p.Ng;
                const int tap = colok ? col / Cc : 0;
                const int r = tap >> 2, sx = tap & 3;
                wg_c[i] = col - tap * Cc;
                wg_cst[i] = wg_s2 ? (r - 1) * W + (sx - 1) : r * 4 + sx;
                wg_ybad[i] = !wg_s2 ? -1 : (r == 0 ? 0 : (r == 3 ? Ho - 1 : -1));
                wg_xbad[i] = !wg_s2 ? -1 : (sx == 0 ? 0 : (sx == 3 ? Wo - 1 : -1));
                wg_colbad[i] = colok ? 0 : -1;
                wg_krow[i] = krow;
            }
        }
    }

    // ---- DMA-side K-iteration state (wave-uniform): the tile the NEXT piece belongs to ---------------------------------
    // FWD / DGRAD_S2 walk the reduction in 64-channel chunks, the taps of a chunk (r, s in the order 0, 2, 1, 3 / the 2x2 taps
    // of the parity class) inside, and the chunk's FOUR 16-channel K-tiles innermost: a k-contiguous row gives a K-tile only
    // 32 bytes, so the four tiles that share a 128-byte line run back to back (the line is fetched once), while the taps of
    // a chunk still re-use the same input pixels (same-box A/B against 16-channel chunks with the taps inside: forward
    // 216 -> 233 TFLOP/s at 128 -> 256 channels, 88 -> 100 on the 256 x 128 tile).  Channel counts that are not a multiple of 64 walk 16-channel chunks (the
    // order of igemm.hip's K-tile 16).  WGRAD walks pixel tiles.
    constexpr int NTAP = MODE == MODE_FWD ? 16 : 4;
    const int SUB = (MODE == MODE_WGRAD) ? 1 : (((MODE == MODE_FWD ? Cc : K) & 63) == 0 ? 4 : 1);
    int dt = it_begin, tap = 0, chunk = 0, sub = 0;          // channel offset of the tile = (chunk * SUB + sub) * 16
    if (MODE != MODE_WGRAD) {
        sub = it_begin % SUB;
        tap = (it_begin / SUB) % NTAP;
        chunk = it_begin / (SUB * NTAP);
    }
    auto advance = [&]() {                       // next tile, clamped to the last one (re-loaded, never used)
        const int go = dt + 1 < it_end ? 1 : 0;
        dt += go;
        if (MODE != MODE_WGRAD) {
            sub += go;
            const int w1 = (sub == SUB) ? 1 : 0;
            sub = w1 ? 0 : sub;
            tap += w1;
            const int w2 = (tap == NTAP) ? 1 : 0;
            tap = w2 ? 0 : tap;
            chunk += w2;
        }
    };
    auto coff = [&]() { return (chunk * SUB + sub) * KT; };
    auto fwd_r = [&]() { const int a = tap >> 2; return ((a & 1) << 1) | (a >> 1); };
    auto fwd_s = [&]() { const int b = tap & 3; return ((b & 1) << 1) | (b >> 1); };

    // LDS-DMA from inline asm (igemm_dma.hip: the compiler must not order the fragment reads behind it); M0 saved / restored
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_ptr)smem;
    auto dma = [&](const __amdgpu_buffer_rsrc_t& r, int lds_off, int voff) {
        unsigned keep;
        const unsigned dst = lds_base + (unsigned)lds_off;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(r), "s"(dst)
                     : "memory");
    };
    // the per-lane source offset of this wave's i-th A / B piece of tile `dt` (the same in every plane)
    auto a_voff = [&](int i) -> int {
        if (MODE == MODE_FWD) {
            const int r = fwd_r(), sx = fwd_s();
            int soff = ((r * W + sx) * Cc + coff()) * 2;                 // wave-uniform
#ifdef DG_TIMING_KNOBS
            if (p.dbg_zero & 16) soff = ((dt & 7) * p.M) * 32;         // (8 distinct 32 M-byte regions of the plane, revisited)
            // bit 6 (with bit 4): the regions a parity-split chunk-major layout [chunk16][class][pixel][16] would touch: one per
            // (16-channel chunk, input-parity class of the tap); the taps of a class revisit it (shifted by a pixel / a row)
            if (p.dbg_zero & 64) soff = ((sub * 4 + ((((r + 1) & 1) << 1) | ((sx + 1) & 1))) * p.M) * 32 + (((r >> 1) * Wo + (sx >> 1)) * 32) % 4096;
#endif
            return (a_ob[i] + soff) | -((a_inv[i] >> (r * 4 + sx)) & 1);
        } else if (MODE == MODE_DGRAD_S2) {
            const int ty = tap >> 1, tx = tap & 1;
            const int dyo = ph == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 0 : 1);
            const int dxo = pw == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 0 : 1);
            const int soff = ((dyo * Wo + dxo) * K + coff()) * 2;
            return (a_ob[i] + soff) | -((a_inv[i] >> tap) & 1);
        }
        return a_ob[i] + dt * (KT * 2) * K;
    };
    auto b_voff = [&](int i) -> int {
        if (MODE == MODE_FWD) {
            const int red = (fwd_r() * 4 + fwd_s()) * Cc + coff();        // first reduction element of the tile
            return BT ? b_ob[i] + red * K * 2 : b_ob[i] + red * 2;
        } else if (MODE == MODE_DGRAD_S2) {
            const int ty = tap >> 1, tx = tap & 1;
            const int r = ph == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 2 : 0);
            const int sx = pw == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 2 : 0);
            return b_ob[i] + ((coff() * 16 + r * 4 + sx) * Cc) * 2;
        }
        // reduction row = output pixel mrow = (n, oy, ox) packed; see igemm.hip load_B (WGRAD)
        const int mrow = dt * KT + wg_krow[i];
        const int oxv = mrow & (Wo - 1), oyv = (mrow >> lgWo) & (Ho - 1);
        const int bad = (oyv == wg_ybad[i]) | (oxv == wg_xbad[i]) | (mrow >= p.R);
        const int pix = (mrow << wg_lpm) - ((oxv << 1) & wg_pxm) + wg_cst[i];
        return ((pix * Cc + wg_c[i]) * 2) | wg_colbad[i] | -bad;
    };
    // piece k of tile `dt` into LDS stage `stage`: k = 3 i + plane for the A pieces (i < NPA), then the B pieces the same way
    int vcur = 0;
    auto issue = [&](int stage, int k) {
        const int pl = k % 3;
        if (k < 3 * NPA) {
            const int i = k / 3;
            if (pl == 0) vcur = a_voff(i);
            dma(rA[pl], stage * STAGE + pl * PLA + (wave * NPA + i) * 1024, vcur);
        } else {
            const int i = (k - 3 * NPA) / 3;
            if (pl == 0) vcur = b_voff(i);
            dma(rB[pl], stage * STAGE + OPA + pl * PLB + (wave * NPB + i) * 1024, vcur);
        }
    };

    // ---- fragment reads: plane pl of 32-row block (k-contiguous) / 32-column block (reduction-major) -----------------
    const int tr_q = (lane >> 2) & 3, tr_c = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    auto frag_kc = [&](const char* img, int row) -> bf16x8 {
        return *(const bf16x8*)(img + row * 32 + ((lh ^ ((row >> 3) & 1)) << 4));
    };
    auto frag_km = [&](const char* img, int rowb, int c0) -> bf16x8 {
        const int kr = 8 * lh + tr_q, col = c0 + tr_c;
        const char* p0 = img + kr * rowb + ((((col >> 3) ^ kmswz(kr))) << 4) + (col & 7) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)p0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(p0 + 4 * rowb));     // kr + 4: same swizzle
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 fa[3][FM];            // [plane][row block]
    bf16x8 fb[2][FN], fbh[2][FN];   // mid / lo planes [plane - 1][column block]; hi plane [tile parity][column block]
    auto fetchA = [&](int stage, int pl, int i) {
        const char* img = smem + stage * STAGE + pl * PLA;
        const int row = wm * (32 * FM) + i * 32;
        fa[pl][i] = A_KM ? frag_km(img, BM * 2, row) : frag_kc(img, row + l31);
    };
    auto fetchB = [&](int stage, int pl, int j, int hset) {       // hset: which B hi set (the parity of the tile) when pl == 0
        const char* img = smem + stage * STAGE + OPA + pl * PLB;
        const int col = wn * (32 * FN) + j * 32;
        const bf16x8 v = B_KM ? frag_km(img, BN * 2, col) : frag_kc(img, col + l31);
        if (pl == 0) fbh[hset][j] = v;
        else fb[pl - 1][j] = v;
    };

    // ---- 16x16x32 body: fragment = 16 rows / columns x (16 k of plane X | 16 k of plane Y), lane (l15, l4): plane by l4 >> 1 ----
    constexpr int AM = 2 * FM, BNB = 2 * FN;            // 8 x 4 blocks of 16 x 16 per wave
    constexpr int NG16 = 3 * AM, RING = 8, FD16 = 7;    // 24 groups of 4 MFMAs per K-tile; A fragment ring, fetched FD16 groups ahead
    const int l15 = lane & 15, l4 = lane >> 4, kh16 = l4 >> 1, kg16 = l4 & 1;
    // byte offset inside a stage of the plane this lane reads for pair type ty.  A: [lo | hi], [mid | mid], [hi | hi];  B: [hi | lo], [mid | hi]
    const int a16_pl[3] = {(kh16 ? 0 : 2) * PLA, PLA, 0};
    const int b16_pl[2] = {OPA + (kh16 ? 2 : 0) * PLB, OPA + (kh16 ? 0 : 1) * PLB};
    // k-contiguous image (plain, no swizzle): row (blk0 + l15), granule kg16 -- block and stage are immediate offsets.
    // Reduction-major image: the granule swizzle XORs address bits 5..7, where the block index lives too, so the address of block b
    // is (address of block 0) ^ (b << 5); that XOR is issued per fetch from inline asm -- left to the compiler, every (block, type,
    // far stage) address is hoisted out of the K loop into its own VGPR (60 of them: the weight-gradient form spilled).
    auto km16_base = [&](int plane_off, int rowb, int wave_col0) -> int {
        const int kr = 8 * kg16 + tr_q, col = wave_col0 + (lane & 3) * 4;
        return plane_off + kr * rowb + ((((col >> 3) ^ kmswz(kr))) << 4) + (col & 7) * 2;
    };
    int akm16[3], bkm16[2];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) akm16[ty] = km16_base(a16_pl[ty], BM * 2, wm * (32 * FM));
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) bkm16[ty] = km16_base(b16_pl[ty], BN * 2, wn * (32 * FN));
    auto frag16_km = [&](int stage, int base, int rowb, int blk) -> bf16x8 {
        int off;
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(off) : "s"(blk << 5), "v"(base));
        const char* p0 = smem + stage * STAGE + off;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)p0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(p0 + 4 * rowb));     // kr + 4: same swizzle
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto frag16_kc = [&](const char* img, int blk0) -> bf16x8 {
        return *(const bf16x8*)(img + (blk0 + l15) * 32 + (kg16 << 4));
    };
    bf16x8 ga[M16 ? RING : 1], gb0[M16 ? BNB : 1], gb1[2][M16 ? BNB : 1];
    auto fetchA16 = [&](int stage, int ty, int i, int slot) {
        if constexpr (A_KM) ga[slot] = frag16_km(stage, akm16[ty], BM * 2, i);
        else ga[slot] = frag16_kc(smem + stage * STAGE + a16_pl[ty], wm * (32 * FM) + i * 16);
    };
    auto fetchB16 = [&](int stage, int ty, int j, int set) {
        bf16x8 v;
        if constexpr (B_KM) v = frag16_km(stage, bkm16[ty], BN * 2, j);
        else v = frag16_kc(smem + stage * STAGE + b16_pl[ty], wn * (32 * FN) + j * 16);
        if (ty == 0) gb0[j] = v;
        else gb1[set][j] = v;
    };
    // the order of a tile's groups: (block row, pair type).  Rows 6 and 7 run their [hi | lo] products first so that the single
    // register set of B [hi | lo] is free 16 MFMAs before the end of the tile.  Small terms first inside every block, as in the 32x32 body.
    // (G16_I / G16_T at the top of the file)

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

    // ---- prologue: tiles 0 .. NS-1 --------------------------------------------------------------------------------------
    if (it_begin < it_end) {
        dg_x3_static_for(std::make_integer_sequence<int, NS>{}, [&](auto S_) {
            constexpr int sg = decltype(S_)::value;
#pragma unroll
            for (int k = 0; k < NPC; ++k) issue(sg, k);
            advance();
        });
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC * (NS - 1)) : "memory");      // tile 0 has landed (this wave's pieces)
    }
    __builtin_amdgcn_s_barrier();
    if (stp) stp[2] = clock64();
    if constexpr (M16) {
#pragma unroll
        for (int j = 0; j < BNB; ++j) {
            fetchB16(0, 0, j, 0);
            fetchB16(0, 1, j, 0);
        }
        dg_x3_static_for(std::make_integer_sequence<int, FD16>{}, [&](auto G_) {
            constexpr int g = decltype(G_)::value;
            fetchA16(0, G16_T[g], G16_I[g], g);
        });
    } else {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int j = 0; j < FN; ++j) fetchB(0, pl, j, 0);
#pragma unroll
            for (int i = 0; i < FM; ++i) fetchA(0, pl, i);
        }
    }

    // ---- one K-tile: 48 MFMAs per wave, row of blocks i = q / 12, plane pair (q % 12) / 2, column block q % 2 -------------
    // ST = LDS stage of the current tile t, PAR = t & 1 (which B hi register set holds tile t).  Fragment replacement (tile
    // t+1, stage NX = (ST + 1) % NS), always one MFMA behind the last reader: A row i-1 in front of MFMA 12 i + 1; B lo in
    // front of MFMA 41, B mid in front of 47, A row 3 at the end; B hi (other register set) right behind the barrier.
    constexpr int QB = 8;
    static_assert(QB >= 1 && QB <= 12, "the tile barrier precedes the first read of tile t+1");
    static_assert(QB + 2 * NPC <= 48, "the DMA of tile t+NS is issued inside tile t");
    auto body = [&](auto ST_, auto PAR_) {
        constexpr int ST = decltype(ST_)::value, PAR = decltype(PAR_)::value, NX = (ST + 1) % NS;
        dg_x3_static_for(std::make_integer_sequence<int, 48>{}, [&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            constexpr int i = q / 12, pr = (q % 12) / 2, j = q % 2;
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (q == QB) {
                // this wave's DMA pieces of tile t+1 have landed (tiles t+2 .. t+NS-1 may still be in flight) and its
                // fragment reads of tile t are complete; behind the barrier that holds for every wave: tile t+1 may be read,
                // tile t's stage may be overwritten
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPC * (NS - 2)) : "memory");
                __builtin_amdgcn_s_barrier();
                fetchB(NX, 0, 0, PAR ^ 1);
                fetchB(NX, 0, 1, PAR ^ 1);
            }
            if constexpr (q % 12 == 1 && i > 0 && q > QB) {
                fetchA(NX, 0, i - 1);
                fetchA(NX, 1, i - 1);
                fetchA(NX, 2, i - 1);
            }
            if constexpr (q == 41) { fetchB(NX, 2, 0, 0); fetchB(NX, 2, 1, 0); }
            if constexpr (q == 47) { fetchB(NX, 1, 0, 0); fetchB(NX, 1, 1, 0); }
            if constexpr (PB[pr] == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pr]][i], fbh[PAR][j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pr]][i], fb[PB[pr] > 0 ? PB[pr] - 1 : 0][j], acc[i][j], 0, 0, 0);
            // DMA of tile t+NS into the stage tile t has left: one piece behind every second MFMA after the barrier
            if constexpr (q >= QB && q < QB + 2 * NPC && (q - QB) % 2 == 0) issue(ST, (q - QB) / 2);
            if constexpr (q == QB + 2 * NPC - 1) advance();
        });
        __builtin_amdgcn_sched_barrier(0);
        fetchA(NX, 0, 3);
        fetchA(NX, 1, 3);
        fetchA(NX, 2, 3);
    };
    // ---- one K-tile on the 16x16x32 MFMA: 24 groups (block row, pair type) of 4 MFMAs (column blocks) = 96 MFMAs of 16 cycles.
    // Group g reads A slot g % RING and B [hi | lo] (type 0) or B [mid | hi] of the tile's parity set (types 1, 2).  In front of
    // its first MFMA the A fragment of group g + FD16 is fetched into the slot group g - 1 has left (from group 19 on: the NEXT
    // tile's, behind the tile barrier in front of group 19 -- every read of the current stage was issued by group 16).  Behind the
    // barrier: the next tile's B [mid | hi] into the other set, B [hi | lo] once group 19 (its last reader) has issued, and the DMA
    // of tile t + NS into the stage tile t is leaving; that DMA has two tile periods to land (vmcnt: this wave's pieces of tile
    // t + 1 are in, tile t + 2's may still fly).
    constexpr int GB16 = 19, QB16 = GB16 * BNB;
    static_assert(NG16 - FD16 <= GB16 - 2 && GB16 + FD16 - NG16 < BNB && NG16 % RING == 0 && RING > FD16, "A ring schedule");
    static_assert(QB16 + 2 * NPC <= NG16 * BNB, "the DMA of tile t+NS is issued inside tile t");
    auto body16 = [&](auto ST_, auto PAR_) {
        constexpr int ST = decltype(ST_)::value, PAR = decltype(PAR_)::value, NX = (ST + 1) % NS;
        dg_x3_static_for(std::make_integer_sequence<int, NG16 * BNB>{}, [&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            constexpr int g = q / BNB, j = q % BNB, i = G16_I[g], ty = G16_T[g];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (q == QB16) {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPC * (NS - 2)) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            // A ring: group g fetches group g + FD16.  The tile's last fragment is fetched by group 23 - FD16, THREE groups before
            // the barrier (its lgkmcnt(0) then finds the reads long complete); groups 17, 18 would read the next stage in front of
            // the barrier: their fetches are caught up by group 19, one per MFMA
            if constexpr (j == 0 && g + FD16 < NG16) fetchA16(ST, G16_T[g + FD16], G16_I[g + FD16], (g + FD16) % RING);
            if constexpr (g == GB16 && NG16 + j <= GB16 + FD16) fetchA16(NX, G16_T[j], G16_I[j], (NG16 + j) % RING);
            if constexpr (j == 0 && g > GB16) fetchA16(NX, G16_T[g + FD16 - NG16], G16_I[g + FD16 - NG16], (g + FD16) % RING);
            if constexpr (g >= 19 && g < 19 + BNB && j == 1) fetchB16(NX, 1, g - 19, PAR ^ 1);
            if constexpr (g >= 20 && g < 20 + BNB && j == 2) fetchB16(NX, 0, g - 20, 0);
            if constexpr (ty == 0) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[g % RING], gb0[j], acc16[i][j], 0, 0, 0);
            else acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[g % RING], gb1[PAR][j], acc16[i][j], 0, 0, 0);
            if constexpr (q >= QB16 && q < QB16 + 2 * NPC && (q - QB16) % 2 == 0) issue(ST, (q - QB16) / 2);
            if constexpr (q == QB16 + 2 * NPC - 1) advance();
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    // the bodies cycle through (stage, parity): 2 stages -> 2 bodies, 3 stages -> 6
    constexpr int NB = NS == 2 ? 2 : 6;
    for (int it = it_begin; it < it_end; it += NB) {
        dg_x3_static_for(std::make_integer_sequence<int, NB>{}, [&](auto B_) {
            constexpr int bi = decltype(B_)::value;
            if (it + bi < it_end) {
                if constexpr (M16) body16(std::integral_constant<int, bi % NS>{}, std::integral_constant<int, bi & 1>{});
                else body(std::integral_constant<int, bi % NS>{}, std::integral_constant<int, bi & 1>{});
            }
        });
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
        if constexpr (M16) {      // 16x16 blocks: rows 32 i .. 32 i + 31 of the wave tile = block rows 2 i, 2 i + 1
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
            *(f32x4*)(dst + eoff) = v;
        }
    }
    if (stp) {
        stp[4] = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stp[5] = wall_clock64();
        stp[7] = clock64();
    }
}

// host: launch the plane kernel for a plan made by igemm.hip (mode, args, wm x wn waves); returns 0 when there is no instantiation
template <int WM, int WN, int NS, bool M16>
static int x3_launch_tile(int mode, const IgemmArgs& a, int grid, hipStream_t st) {
    const dim3 blk(64 * WM * WN);
    switch (mode) {
        case MODE_FWD:
            if (a.b_transposed) hipLaunchKernelGGL((igemm_dma_x3_kernel<MODE_FWD, true, WM, WN, NS, M16>), dim3(grid), blk, 0, st, a);
            else hipLaunchKernelGGL((igemm_dma_x3_kernel<MODE_FWD, false, WM, WN, NS, M16>), dim3(grid), blk, 0, st, a);
            return 1;
        case MODE_DGRAD_S2: hipLaunchKernelGGL((igemm_dma_x3_kernel<MODE_DGRAD_S2, false, WM, WN, NS, M16>), dim3(grid), blk, 0, st, a); return 1;
        case MODE_WGRAD: hipLaunchKernelGGL((igemm_dma_x3_kernel<MODE_WGRAD, false, WM, WN, NS, M16>), dim3(grid), blk, 0, st, a); return 1;
        default: return 0;
    }
}
int dg_igemm_dma_x3_launch(int mode, int wm, int wn, const IgemmArgs& a, int zmul, hipStream_t st) {
    const int grid = a.tilesM * a.tilesN * zmul * a.splits;
    // option "x3_mfma": 16 = the 256 x 256 tile on the 16x16x32 MFMA with planes paired along k (three LDS stages); 32 / 0 = the
    // 32x32x16 body (two stages), the default.  Same-box A/B at 512 px / batch 32, 16 against 32: +2.2..4.2 % forward / input-grad /
    // weight-grad on 256 -> 512 and 512 -> 1024 channels, +0..1.8 % on 1024 -> 2048 (256 K-tiles per workgroup each), -1..-3 % on
    // 128 -> 256 and -12 % on the forward of 2048 -> 2048 @ 8 (128 K-tiles per workgroup) in isolated launches -- and NOTHING in the
    // whole step (284.2 / 283.4 images/s with 32, 282.6 with 16, 283.1 / 281.9 with 16 on the long K loops only): DESIGN.md 3.1
    // (built in the experiments library only: make EXPERIMENTS=1 -- the product library has ONE body per tile)
#ifdef DG_EXPERIMENTS
    const int body = dg_get_option(DG_OPT_X3_MFMA) == 16 ? 16 : 32;
    if (wm == 2 && wn == 4 && body == 16) return x3_launch_tile<2, 4, 3, true>(mode, a, grid, st);
#endif
    if (wm == 2 && wn == 4) return x3_launch_tile<2, 4, 2, false>(mode, a, grid, st);
#ifdef DG_TIMING_KNOBS
    if (wm == 2 && wn == 2 && mode == MODE_FWD) return x3_launch_tile<2, 2, 2, false>(mode, a, grid, st);     // timing experiment only
#endif
    if (wm == 1 && wn == 4 && mode == MODE_WGRAD) {
        hipLaunchKernelGGL((igemm_dma_x3_kernel<MODE_WGRAD, false, 1, 4, 2>), dim3(grid), dim3(256), 0, st, a);
        return 1;
    }
    return 0;
}

// ---- transposed weight planes for the forward form -------------------------------------------------------------------------
// For every conv weight of a flat parameter group: planes [3][K][J] (J = 16 C, the KRSC image) -> [3][J][K] at the same flat
// offset of a second plane buffer.  One launch for the whole group: 64 x 64 tiles through LDS, 128-byte rows both ways.
struct X3TransposeTable {
    int n;
    int tile0[DG_X3T_MAX + 1];      // first tile of weight i (prefix sums); tile0[n] = grid
    int K[DG_X3T_MAX], J[DG_X3T_MAX];
    long off[DG_X3T_MAX];           // element offset of the weight inside a plane
};
__global__ __launch_bounds__(256) void x3_transpose_kernel(const __bf16* __restrict__ src, __bf16* __restrict__ dst, long plane, X3TransposeTable t) {
    // the tile as 64 k rows x 32 words (a word = two neighbouring j of one k), pitch 33: the loads write rows, the stores read
    // columns -- 8 words W[8 q .. 8 q + 7][jp] give the 16-byte runs of output rows j = 2 jp (low halves) and 2 jp + 1 (high halves)
    __shared__ unsigned tile[64][33];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    int wi = 0;
    while (wi + 1 < t.n && (int)blockIdx.x >= t.tile0[wi + 1]) ++wi;
    const int K = t.K[wi], J = t.J[wi];
    const int tj = (J + 63) / 64;
    const int b = blockIdx.x - t.tile0[wi];
    const int k0 = (b / tj) * 64, j0 = (b % tj) * 64;
    const bool vec = (K % 8 == 0) && (J % 8 == 0) && (t.off[wi] % 8 == 0);
    const int lk = threadIdx.x >> 2, lv = threadIdx.x & 3;          // load: row k, 16-byte vectors lv and lv + 4 of the row
    const int sq = threadIdx.x & 7, sjp = threadIdx.x >> 3;          // store: k run 8 sq .. 8 sq + 7 of output rows 2 sjp, 2 sjp + 1
    for (int pl = 0; pl < 3; ++pl) {
        const unsigned short* s = (const unsigned short*)src + pl * plane + t.off[wi];
        unsigned short* d = (unsigned short*)dst + pl * plane + t.off[wi];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + lk, j = j0 + (lv + 4 * h) * 8;
            u32x4 v = (u32x4){0u, 0u, 0u, 0u};
            if (k < K) {
                if (vec && j < J) v = *(const u32x4*)(s + (long)k * J + j);
                else if (!vec) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (j + e < J) v[e >> 1] |= (unsigned)s[(long)k * J + j + e] << (16 * (e & 1));
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[lk][(lv + 4 * h) * 4 + e] = v[e];
        }
        __syncthreads();
        unsigned w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = tile[8 * sq + e][sjp];
        u32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = (w[2 * e] & 0xffffu) | (w[2 * e + 1] << 16);
            hi[e] = (w[2 * e] >> 16) | (w[2 * e + 1] & 0xffff0000u);
        }
        const int j = j0 + 2 * sjp, k = k0 + 8 * sq;
        if (vec) {
            if (k < K && j < J) {
                *(u32x4*)(d + (long)j * K + k) = lo;
                *(u32x4*)(d + (long)(j + 1) * K + k) = hi;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (k + e < K && j < J) d[(long)j * K + k + e] = (unsigned short)(lo[e >> 1] >> (16 * (e & 1)));
                if (k + e < K && j + 1 < J) d[(long)(j + 1) * K + k + e] = (unsigned short)(hi[e >> 1] >> (16 * (e & 1)));
            }
        }
        __syncthreads();
    }
}
// w_off / w_K / w_J: host arrays of n conv weights ([K][J] images at element offset w_off inside each plane); src and dst are
// [3][plane_elems] bf16 buffers
extern "C" int dg_x3_transpose_planes(const void* src_planes, void* dst_planes, size_t plane_elems, const int64_t* w_off,
                                      const int* w_K, const int* w_J, int n, dg_stream_t stream) {
    DG_CHECK_ARG(src_planes && dst_planes && w_off && w_K && w_J, "dg_x3_transpose_planes: null pointer");
    DG_CHECK_ARG(n >= 0, "dg_x3_transpose_planes: n=%d", n);
    for (int i0 = 0; i0 < n; i0 += DG_X3T_MAX) {
        X3TransposeTable t;
        t.n = n - i0 < DG_X3T_MAX ? n - i0 : DG_X3T_MAX;
        int tiles = 0;
        for (int i = 0; i < t.n; ++i) {
            const int K = w_K[i0 + i], J = w_J[i0 + i];
            DG_CHECK_ARG(K >= 1 && J >= 1 && w_off[i0 + i] >= 0 && (size_t)(w_off[i0 + i] + (long)K * J) <= plane_elems,
                         "dg_x3_transpose_planes: weight %d (K=%d, J=%d, offset %ld) outside the plane", i0 + i, K, J, (long)w_off[i0 + i]);
            t.tile0[i] = tiles;
            t.K[i] = K; t.J[i] = J; t.off[i] = w_off[i0 + i];
            tiles += ((K + 63) / 64) * ((J + 63) / 64);
        }
        t.tile0[t.n] = tiles;
        if (tiles == 0) continue;
        hipLaunchKernelGGL(x3_transpose_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src_planes,
                           (__bf16*)dst_planes, (long)plane_elems, t);
        DG_CHECK_LAUNCH("x3_transpose");
    }
    return DG_OK;
}
