// Input-gradient of Conv2d(k4,s2,p1) / forward of ConvTranspose2d(k4,s2,p1) for layers with FEW output channels (C <= 128), f32x3
// plane operands: ALL parity classes of a pixel tile in one workgroup, the gradient tile fetched ONCE per 16-channel chunk into
// an LDS WINDOW and re-used by every tap ("dgw" = input-grad with window).
//
// Replaces (reference file:line) nn.Conv2d(k4,s2,p1) input-grad (model.py:11-31,83-103 via autograd) and
// nn.ConvTranspose2d(k4,s2,p1) forward (model.py:118-140) where igemm_dma_x3.hip's DGRAD_S2 form does not apply.
//
// Why: out pixel (2a+ph, 2b+pw) sums dy[a+dyo][b+dxo] . w[k][r][s][c] over the 2x2 taps of its parity class (ph, pw); the
// four classes together use the NINE shifts (dyo, dxo) in {-1,0,1}^2 of the same gradient pixels, sixteen (shift, class) pairs.
// The per-class kernels fetch a shifted 256-pixel tile for every one of them: each gradient element crosses the fabric sixteen
// times, and with C <= 128 output columns there is too little MFMA work per fetched byte (the register-staged f32x3 tiles run
// these layers at 150-190 TFLOP/s, a 256 x 128 plane tile was HBM-bound at 100).  Here
//   * a workgroup owns 256 consecutive gradient pixels (R = 256 / Wo rows of one image) and NCLS parity classes: 4 classes x 64
//     columns (C <= 64) or 2 classes x 128 (C <= 128; ph comes from blockIdx) -- the 256 x 256 accumulator layout of
//     igemm_dma_x3.hip, a wave column (pair) per class;
//   * per 16-channel chunk the (R + 2) x (Wo + 2) pixel window around the tile goes to LDS once per plane (<= 17 KB; halo pixels
//     outside the image are out-of-range DMA offsets = zeros); a tap's A fragment is a ds_read_b128 at row
//     `window row of the pixel + dyo (Wo + 2) + dxo` -- the swizzle `slot = half ^ ((row >> 3) & 1)` is conflict-free for any row
//     offset; A traffic per MFMA drops 7.7x (C <= 64) / 3.9x;
//   * the K loop walks (chunk, tap t = 0..3): step (c, t) multiplies, for every class, the class's t-th tap; its weight tile is
//     [16 k][NCLS x CW columns] with the (r, s) of (class, t) baked into the per-lane DMA offsets;
//   * LDS: 2 window stages (one per chunk parity) + 2 weight stages (one per step parity) <= 150 KB; the window of chunk c+1
//     is fetched one plane per step during steps 0..2 of chunk c; barrier / fragment-replacement scheme of igemm_dma_x3.hip.
// Same six MFMAs per product block; the reduction order per output element is (chunk, tap) like the per-class kernels.
#include "igemm_args.h"
#include <type_traits>
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4* lds_bf4_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int... Q, typename F>
__device__ __forceinline__ void dgw_static_for(std::integer_sequence<int, Q...>, F&& f) {
    (f(std::integral_constant<int, Q>{}), ...);
}

// NCLS: parity classes per workgroup (4: 64 columns each, 2: 128 columns each, ph = blockIdx parity)
// PERSIST: the grid is one workgroup per CU and every workgroup walks the tiles bid = blockIdx.x, + gridDim.x, ...; the DMA of the
// NEXT tile's first window and weight tiles is issued BEFORE the finished tile's statistics / epilogue, which run out of the other
// window stage: with one workgroup per CU (150 KB of LDS) nothing else hides a tile's prologue (stamps, round 2: 9.5 us of an 84 us
// workgroup before the first MFMA, 8.3 us of epilogue).
template <int NCLS, bool PERSIST>
__global__ __launch_bounds__(512, 2) void igemm_x3_dgw_kernel(const IgemmArgs p, const int nblocks) {
    constexpr int WN = 4, FM = 4, FN = 2, KT = 16;      // 2 x 4 waves of 128 x 64
    constexpr int CW = 256 / NCLS;                      // columns per class
    constexpr int WPMAX = 17;                           // window pieces (32 rows each) per plane: (R + 2)(Wo + 2) <= 544 rows
    constexpr int WPB = WPMAX * 1024;                   // bytes per window plane
    constexpr int AST = 3 * WPB;                        // window stage
    constexpr int PLB = 256 * KT * 2, BST = 3 * PLB;    // weight tile plane / stage
    constexpr int B_OFF = 0, A_OFF = 2 * BST;           // weight stages first: their fragment reads then fit ds_read's 16-bit offsets
    constexpr int LDS_BYTES = 2 * AST + 2 * BST;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    constexpr int EPI_OFF = A_OFF + AST;                // epilogue transposes: 8 waves x [16][68] floats in window stage 1 (free after the K loop;
    static_assert(8 * 16 * 68 * 4 <= AST, "epilogue transpose regions live in window stage 1");     // the next tile's prologue fills stage 0 and the weight stages)
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

    // ---- block index -> (pixel tile, ph for NCLS == 2, split) -------------------------------------------------------------
    // Vertically adjacent pixel tiles share two of their (R + 2) window rows and the ZM workgroups of a tile share the whole
    // window: workgroups are dealt to the 8 XCDs round robin, so XCD x takes the tiles [x tilesM / 8, (x + 1) tilesM / 8) in
    // order (the ph workgroups of a tile back to back) and the shared rows are L2 hits instead of a second fabric fetch.
    // (PERSIST: gridDim.x is a multiple of 8 whenever tilesM is, so a workgroup's tiles stay on its XCD's share.)
    constexpr int ZM = 4 / NCLS;
    const int H = p.H, W = p.W, Cc = p.Cc, K = p.K, Ho = p.Ho, Wo = p.Wo;
    const int lgWo = p.lgWo, lgHW = p.lgWo + p.lgHo;
    const int WW = Wo + 2;
    const int WR = (256 / Wo + 2) * WW;                       // window rows
    const int wpieces = (WR + 31) >> 5;
    int tm, zph, split, m0, cb, ce, n_img, a0;
    auto decode = [&](int bid) {
        if ((p.tilesM & 7) == 0) {
            const int per = p.tilesM >> 3, x = bid & 7;
            int j = bid >> 3;
            zph = j % ZM;
            j /= ZM;
            tm = x * per + j % per;
            split = j / per;
        } else {
            tm = bid % p.tilesM;
            bid /= p.tilesM;
            zph = bid % ZM;
            split = bid / ZM;
        }
        m0 = tm * 256;
        cb = split * p.itPerSplit;                             // chunk range of this split
        ce = min(p.nIt, cb + p.itPerSplit);
        // tile = R rows x Wo pixels of image n, rows a0 .. a0 + R - 1; window = rows a0 - 1 .. a0 + R, columns -1 .. Wo
        n_img = m0 >> lgHW;
        a0 = (m0 >> lgWo) & (Ho - 1);
    };
    int bid_cur = blockIdx.x;
    decode(bid_cur);

    constexpr int OOR = (int)0x80000000;
    __amdgpu_buffer_rsrc_t rA[3], rB[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        rA[pl] = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + pl * p.a_plane), 0, (p.dbg_zero & 1) ? 0 : (int)p.abytes, 0x00020000);
        rB[pl] = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + pl * p.b_plane), 0, (p.dbg_zero & 2) ? 0 : (int)p.bbytes, 0x00020000);
    }
    auto kmswz = [](int k) -> int { return (k & 3) << 2; };

    // ---- window DMA descriptors: this wave's pieces w, w + 8, w + 16 of every plane (the third only where it exists) ----
    // piece pc covers window rows 32 pc .. 32 pc + 31; lane L lands in (row 32 pc + L / 2, slot L % 2), fetches granule
    // slot ^ ((row >> 3) & 1) of gradient pixel (a0 - 1 + row / WW, row % WW - 1); outside the image or the window: zeros
    int w_ob[3];
    int b_base, b_ph, b_pw;
    // class of this wave's columns and, per tap t, the window-row shift dyo * WW + dxo of that class
    const int cls = NCLS == 4 ? wn : (wn >> 1);
    int ph, pw, shift[4];
    auto tile_setup = [&]() {          // everything that depends on (tm, zph): called after decode()
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int pc = wave + 8 * j;
            const int row = pc * 32 + (lane >> 1);
            const int g = (lane & 1) ^ ((row >> 3) & 1);
            const int wr = row / WW, wc = row - wr * WW;
            const int a = a0 - 1 + wr, b = wc - 1;
            const bool ok = row < WR && (unsigned)a < (unsigned)Ho && (unsigned)b < (unsigned)Wo;
            const int pix = (n_img * Ho + a) * Wo + b;
            // quad-chunk planes [pixels / 4][K / 16][4][16]: the chunk of 4 consecutive pixels is one 128-byte line -- a window row
            // of a chunk uses every byte of the lines it touches
            w_ob[j] = ok ? (p.a_cm ? ((pix >> 2) * (4 * K) + (pix & 3) * 16 + g * 8) * 2 : (pix * K + g * 8) * 2) : OOR;
        }
        // ---- weight DMA descriptors: tile [16 k][256 columns], column = class-local-index * CW + c; one piece per plane ----
        // piece = k rows 2 w, 2 w + 1; lane L lands in (k row 2 w + L / 32, slot L % 32), fetches granule slot ^ kmswz(k row);
        // the tap (r, s) of (class, t) is part of the per-lane offset: one offset per t
        {
            const int krow = wave * 2 + (lane >> 5);
            const int gc = (lane & 31) ^ kmswz(krow);
            const int col = gc * 8;
            const int bcls = col / CW, cc = col - bcls * CW;
            b_ph = NCLS == 4 ? (bcls >> 1) : zph;
            b_pw = NCLS == 4 ? (bcls & 1) : bcls;
            b_base = cc < Cc ? (krow * 16 * Cc + cc) * 2 : OOR;
        }
        ph = NCLS == 4 ? (cls >> 1) : zph;
        pw = NCLS == 4 ? (cls & 1) : cls;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ty = t >> 1, tx = t & 1;
            const int dyo = ph == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 0 : 1);
            const int dxo = pw == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 0 : 1);
            shift[t] = dyo * WW + dxo;
        }
    };
    tile_setup();

    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_ptr)smem;
    auto dma = [&](const __amdgpu_buffer_rsrc_t& r, int lds_off, int voff) {
        unsigned keep;
        const unsigned dst = lds_base + (unsigned)lds_off;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(r), "s"(dst)
                     : "memory");
    };
    // plane pl of the window of chunk `c` into window stage `ast`
    auto issue_window = [&](int ast, int pl, int c) {
        const int coff = p.a_cm ? c * (4 * KT * 2) : c * KT * 2;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (wave + 8 * j < wpieces)                                   // wave-uniform
                dma(rA[pl], A_OFF + ast * AST + pl * WPB + (wave + 8 * j) * 1024, w_ob[j] + coff);
        }
    };
    // weight tile of step (chunk c, tap t) into weight stage `bst` (three planes)
    auto issue_weights = [&](int bst, int c, int t) {
        const int ty = t >> 1, tx = t & 1;                                  // wave-uniform
        const int r = b_ph == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 2 : 0);    // per lane (the lane's class)
        const int sx = b_pw == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 2 : 0);
        const int voff = b_base + ((r * 4 + sx) * Cc + c * (KT * 16) * Cc) * 2;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dma(rB[pl], B_OFF + bst * BST + pl * PLB + wave * 1024, voff);
    };

    // ---- fragment reads ----------------------------------------------------------------------------------------------
    // window row of lane 0's pixel in the wave's 32-pixel block i (wave-uniform; a block never straddles an image row: Wo >= 32)
    int srow[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int ml = wm * (32 * FM) + i * 32;
        srow[i] = ((ml >> lgWo) + 1) * WW + (ml & (Wo - 1)) + 1;
    }
    const int tr_q = (lane >> 2) & 3, tr_c = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    auto frag_kc = [&](const char* img, int row) -> bf16x8 {
        return *(const bf16x8*)(img + row * 32 + ((lh ^ ((row >> 3) & 1)) << 4));
    };
    auto frag_km = [&](const char* img, int c0) -> bf16x8 {
        const int kr = 8 * lh + tr_q, col = c0 + tr_c;
        const char* p0 = img + kr * 512 + ((((col >> 3) ^ kmswz(kr))) << 4) + (col & 7) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)p0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(p0 + 4 * 512));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 fa[3][FM];
    bf16x8 fb[2][FN], fbh[2][FN];
    // (the window row is recomputed at every fetch: hoisted out of the K loop, the 16 (block, tap) addresses x 2 stages cost more
    // registers than the kernel has -- the opaque asm keeps the compiler from doing that)
    auto fetchA = [&](int ast, int pl, int i, int t) {
        int lv = l31;
        asm volatile("" : "+v"(lv));
        fa[pl][i] = frag_kc(smem + A_OFF + ast * AST + pl * WPB, lv + (srow[i] + shift[t]));
    };
    auto fetchB = [&](int bst, int pl, int j, int hset) {
        const bf16x8 v = frag_km(smem + B_OFF + bst * BST + pl * PLB, wn * (32 * FN) + j * 32);
        if (pl == 0) fbh[hset][j] = v;
        else fb[pl - 1][j] = v;
    };

    f32x16 acc[FM][FN];

    // ---- DMA-side state: the step (wc, wt) whose weight tile is issued next; clamped to the last step -------------------
    int wc = cb, wt = 0;
    auto advance = [&]() {
        const int last = (wc == ce - 1 && wt == 3) ? 1 : 0;
        wt += 1 - last;
        const int wrap = wt == 4 ? 1 : 0;
        wt = wrap ? 0 : wt;
        wc += wrap;
    };
    // ---- prologue DMA of the tile decode() / tile_setup() describe: window of the first chunk, weight tiles of steps 0 and 1 ------
    auto issue_prologue = [&]() {
        wc = cb;
        wt = 0;
        if (cb < ce) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) issue_window(0, pl, cb);
            issue_weights(0, wc, wt);
            advance();
            issue_weights(1, wc, wt);
            advance();
        }
    };
    issue_prologue();

    // ---- one step (chunk c in window stage AS, tap T; weight stage T & 1): 48 MFMAs per wave, schedule of igemm_dma_x3.hip ----
    // next step: tap T + 1 of the same window, or tap 0 of the next chunk's window (stage AS ^ 1, complete since the
    // barrier of this step: its planes were issued in steps 0, 1, 2)
    auto body = [&](auto AS_, auto T_, int c) {
        constexpr int AS = decltype(AS_)::value, T = decltype(T_)::value;
        constexpr int BS = T & 1, NT = (T + 1) & 3, NAS = T == 3 ? AS ^ 1 : AS, NBS = BS ^ 1;
        constexpr int QB = 8;
        dgw_static_for(std::make_integer_sequence<int, 48>{}, [&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            constexpr int i = q / 12, pr = (q % 12) / 2, j = q % 2;
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (q == QB) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                fetchB(NBS, 0, 0, BS ^ 1);
                fetchB(NBS, 0, 1, BS ^ 1);
            }
            if constexpr (q % 12 == 1 && i > 0 && q > QB) {
                fetchA(NAS, 0, i - 1, NT);
                fetchA(NAS, 1, i - 1, NT);
                fetchA(NAS, 2, i - 1, NT);
            }
            if constexpr (q == 41) { fetchB(NBS, 2, 0, 0); fetchB(NBS, 2, 1, 0); }
            if constexpr (q == 47) { fetchB(NBS, 1, 0, 0); fetchB(NBS, 1, 1, 0); }
            if constexpr (PB[pr] == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pr]][i], fbh[BS][j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pr]][i], fb[PB[pr] > 0 ? PB[pr] - 1 : 0][j], acc[i][j], 0, 0, 0);
            // behind the barrier: the weight tile of step + 2 into the stage this step is leaving, and (steps 0..2) plane T of
            // the next chunk's window into the other window stage
            if constexpr (q == QB + 1) {
                issue_weights(BS, wc, wt);
                advance();
            }
            if constexpr (q == QB + 9 && T < 3) issue_window(AS ^ 1, T, min(c + 1, ce - 1));
        });
        __builtin_amdgcn_sched_barrier(0);
        fetchA(NAS, 0, 3, NT);
        fetchA(NAS, 1, 3, NT);
        fetchA(NAS, 2, 3, NT);
    };
    for (;;) {
        // ---- the tile's first window and weight tiles have been issued (before the loop / before the previous tile's epilogue) ----
        // (vmcnt(0): the previous tile's epilogue stores are in the same counter and return out of order with the loads)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (stp) stp[2] = clock64();
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int j = 0; j < FN; ++j) fetchB(0, pl, j, 0);
#pragma unroll
            for (int i = 0; i < FM; ++i) fetchA(0, pl, i, 0);
        }
        for (int c = cb; c < ce; c += 2) {
            dgw_static_for(std::make_integer_sequence<int, 8>{}, [&](auto B_) {
                constexpr int bi = decltype(B_)::value;
                if (c + bi / 4 < ce) body(std::integral_constant<int, bi / 4>{}, std::integral_constant<int, bi % 4>{}, c + bi / 4);
            });
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (stp) stp[3] = clock64();

        // what the statistics and the epilogue of the FINISHED tile need, before decode() / tile_setup() move on to the next one
        const int e_m0 = m0, e_tm = tm, e_split = split, e_ph = ph, e_pw = pw;
        const int nbid = bid_cur + (int)gridDim.x;
        const bool more = PERSIST && nbid < nblocks;                   // workgroup-uniform
        if (more) {
            // every read of the operand stages is complete (barrier above): the next tile's first window goes to window stage 0, its
            // first two weight tiles to the weight stages; the epilogue below works in window stage 1
            bid_cur = nbid;
            decode(nbid);
            tile_setup();
            issue_prologue();
        }

        // ---- fused BatchNorm statistics (igemm.hip's scheme): one partial row per (parity class, tile, wave row) -----------------
        if (p.stat != nullptr && p.part == nullptr) {
            const int row0 = e_m0 + wm * (32 * FM);
            const int nrows = min(32 * FM, p.M - row0);
            const int prow = ((e_ph * 2 + e_pw) * p.tilesM + e_tm) * 2 + wm;
            float* srow = p.stat + (long)prow * p.stat_rs;
            const int cbase = NCLS == 4 ? 0 : (wn & 1) * 64;            // first column of this wave inside its class
            if (cbase == 0 && lane == 0) srow[0] = (float)max(nrows, 0);
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
                    const int n = cbase + jn * 32 + l31;
                    if (lh == 0 && n < Cc) {
                        srow[4 + n] = sh;
                        srow[4 + Cc + n] = ssum;
                        srow[4 + 2 * Cc + n] = ssq;
                    }
                }
            }
        }

        // ---- epilogue: the wave's 128 pixels x 64 columns of class (ph, pw); rows -> out pixel (2a + ph, 2b + pw) -------------
        // transposed through a private [16][68] LDS region per wave, 16 rows (half an accumulator block) at a time
        const bool to_part = p.part != nullptr;
        float* const eps = (float*)(smem + EPI_OFF) + wave * (16 * 68);
        const int erow = lane >> 4, ec4 = (lane & 15) * 4;
        const int ccol = (NCLS == 4 ? 0 : (wn & 1) * 64) + ec4;            // column inside the class
        const int parity = e_ph * 2 + e_pw;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = hf * 8 + r8;
                    const int lr = (r8 & 3) + 8 * (r8 >> 2) + 4 * lh;      // row inside the half: acc row (r&3) + 8 (r>>2) + 4 lh, minus 16 hf
                    eps[lr * 68 + l31] = acc[i][0][r];
                    eps[lr * 68 + 32 + l31] = acc[i][1][r];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = t * 4 + erow;
                    f32x4 v = *(const f32x4*)(eps + row * 68 + ec4);
                    const int m = e_m0 + wm * (32 * FM) + i * 32 + hf * 16 + row;
                    if (m >= p.M || ccol >= Cc) continue;
                    float* dst;
                    long eoff;
                    if (to_part) {
                        dst = p.part;
                        eoff = (((long)e_split * 4 + parity) * p.M + m) * Cc + ccol;
                    } else {
                        const int b = m & (Wo - 1), a = (m >> lgWo) & (Ho - 1), n = m >> lgHW;
                        dst = p.C;
                        eoff = (long)((n * H + 2 * a + e_ph) * W + 2 * b + e_pw) * Cc + ccol;
                    }
                    if (!to_part && p.accumulate) v += *(const f32x4*)(dst + eoff);
                    *(f32x4*)(dst + eoff) = v;
                }
            }
        }
        if (stp) {
            stp[4] = clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stp[5] = wall_clock64();
            stp[7] = clock64();
        }
        if (!more) break;
    }
}

// host: launch for a plan made by igemm.hip (ncls = 4 / 2); blocks = pixel tiles x (4 / ncls) x splits, one workgroup each.  Option
// "dgw_persist" 1 with more blocks than CUs (each workgroup owns a CU: 150 KB of LDS): one persistent workgroup per CU walks them --
// bit-identical, and measured NOT faster (same-box A/B at 512 px / batch 32: input-grad 0.694 / 0.686 -> 0.714 / 0.718 ms at 64
// channels, 0.599 / 0.597 -> 0.602 / 0.601 at 128; whole step 292.8 / 292.1 -> 292.2 / 291.4 images/s): in steady state the tile
// prologue is not the 9.5 us the stamps of a cold launch showed, and the wait for the epilogue's stores at the top of the tile
// loop takes back what the early DMA gains.  DESIGN.md 3.1.
int dg_igemm_x3_dgw_launch(int ncls, const IgemmArgs& a, hipStream_t st) {
    const int nblocks = a.tilesM * (4 / ncls) * a.splits;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) ncu = -1;
        else ncu = prop.multiProcessorCount;
    }
    // (built in the experiments library only: make EXPERIMENTS=1)
#ifdef DG_EXPERIMENTS
    const bool persist = dg_get_option(DG_OPT_DGW_PERSIST) == 1 && ncu > 0 && nblocks > ncu && a.stamps == nullptr;
#else
    const bool persist = false;
#endif
#ifdef DG_EXPERIMENTS
    if (persist) {
        const int grid = ncu & ~7 ? (ncu & ~7) : ncu;       // a multiple of 8: a workgroup's tiles stay on its XCD's share
        if (ncls == 4) hipLaunchKernelGGL((igemm_x3_dgw_kernel<4, true>), dim3(grid), dim3(512), 0, st, a, nblocks);
        else if (ncls == 2) hipLaunchKernelGGL((igemm_x3_dgw_kernel<2, true>), dim3(grid), dim3(512), 0, st, a, nblocks);
        else return 0;
        return 1;
    }
#endif
    (void)persist;
    if (ncls == 4) hipLaunchKernelGGL((igemm_x3_dgw_kernel<4, false>), dim3(nblocks), dim3(512), 0, st, a, nblocks);
    else if (ncls == 2) hipLaunchKernelGGL((igemm_x3_dgw_kernel<2, false>), dim3(nblocks), dim3(512), 0, st, a, nblocks);
    else return 0;
    return 1;
}
