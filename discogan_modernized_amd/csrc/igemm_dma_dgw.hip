// Input-gradient of Conv2d(k4,s2,p1) / forward of ConvTranspose2d(k4,s2,p1) for layers with FEW output channels (C <= 128) on the
// bf16 matrix path (BASELINE configs[4]), bf16 operands in HBM: ALL parity classes of a pixel tile in one workgroup, the gradient
// tile fetched ONCE per 32-channel chunk into an LDS WINDOW and re-used by every tap -- igemm_dma_x3_dgw.hip's scheme with one
// bf16 plane and a 32-deep K-step.
//
// Replaces (reference file:line) nn.Conv2d(k4,s2,p1) input-grad (model.py:11-31,83-103 via autograd) and
// nn.ConvTranspose2d(k4,s2,p1) forward (model.py:118-140) where igemm_dma.hip's 256 x 256 tile does not apply (fewer than 192 columns).
//
//   * workgroup = 256 consecutive gradient pixels (whole image rows) x NCLS parity classes (4 x 64 columns for C <= 64, 2 x 128 for
//     C <= 128, ph from blockIdx); 8 waves of 128 x 64, a wave column (pair) per class;
//   * window [(R + 2)(Wo + 2) pixels][32 k] bf16 = 64-byte rows, <= 33 KB, two stages (chunk parity); 16-byte granule g of row r in
//     slot g ^ ((r >> 2) & 3): a ds_read_b128 group of 16 consecutive rows is conflict-free at ANY row offset, so a tap's A
//     fragment is a read at `window row + dyo (Wo + 2) + dxo`; halo pixels outside the image: out-of-range DMA offsets = zeros;
//   * weight tile of step (chunk c, tap t): [32 k][NCLS x CW columns] = 16 KB, the (r, s) of (class, t) in the per-lane DMA
//     offsets; THREE stages (a step is only 16 MFMAs per wave: a weight tile gets two steps to land);
//   * every wave issues the SAME number of DMA instructions per step (WPS = 2 weight pieces; NWIN = 3 / 2 / 0 / 0 window pieces in
//     taps 0..3, absent pieces are zero-fills of unused LDS), so the counted wait in front of the step barrier is exact for every
//     wave; the counts are compile-time constants, the waits are computed from them and static_asserts tie the issue sites to them;
//   * step = 2 k16 steps x 8 accumulator blocks (v_mfma_f32_32x32x16_bf16); fragments double-buffered over the k16 steps, the
//     barrier in front of the second one, the next step's first fragments behind it (igemm_dma.hip's 32x32 body).
// Epilogue: fp32 or bf16 output (RNE), split-K slabs in igemm.hip's layout.
#include "igemm_args.h"
#include <type_traits>
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4* lds_bf4_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int... Q, typename F>
__device__ __forceinline__ void dgwb_static_for(std::integer_sequence<int, Q...>, F&& f) {
    (f(std::integral_constant<int, Q>{}), ...);
}

template <int NCLS>
__global__ __launch_bounds__(512, 2) void igemm_bf16_dgw_kernel(const IgemmArgs p) {
    constexpr int WN = 4, FM = 4, FN = 2, KT = 32;      // 2 x 4 waves of 128 x 64
    constexpr int CW = 256 / NCLS;                      // columns per class
    constexpr int NWP = 5;                              // window pieces (16 rows of 64 B each) per wave: 40 pieces >= 33
    // DMA schedule, the basis of the counted `vmcnt` waits.  EVERY wave issues, unconditionally (no lane- or wave-dependent
    // branch around a `dma()`: absent window rows / weight columns are out-of-range offsets = zero-fills), exactly
    //   WPS weight pieces in every step, and NWIN[T] window pieces of the next chunk in tap T of a chunk.
    // The waits below are derived from these constants; the asserts tie the issue sites to them.
    constexpr int WPS = 2;                              // weight pieces per wave and step (16 pieces x 1 KiB = one 16 KB stage)
    constexpr int NWIN[4] = {3, 2, 0, 0};               // window pieces per wave issued in taps 0..3
    static_assert(NWIN[0] + NWIN[1] + NWIN[2] + NWIN[3] == NWP, "the four taps of a chunk must issue the whole next window");
    static_assert(8 * WPS * 1024 == 256 * KT * 2, "8 waves x WPS pieces must cover one weight stage");
    static_assert(NWIN[2] == 0 && NWIN[3] == 0, "the window of chunk c + 1 must be complete two barriers before its first read (tap 3's barrier)");
    constexpr int AST = 8 * NWP * 1024;                 // window stage: 40 KB
    constexpr int BST = 256 * KT * 2;                   // weight stage: 16 KB
    constexpr int B_OFF = 0, A_OFF = 3 * BST;           // weight stages first
    constexpr int LDS_BYTES = 3 * BST + 2 * AST;        // 128 KB
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(8 * 32 * 68 * 4 <= LDS_BYTES, "epilogue transpose regions live in the operand stages");
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

    // ---- blockIdx -> (pixel tile, ph for NCLS == 2, split) ----------------------------------------------------------
    constexpr int ZM = 4 / NCLS;
    int bid = blockIdx.x;
    const int tm = bid % p.tilesM;
    bid /= p.tilesM;
    const int zph = bid % ZM;
    const int split = bid / ZM;
    const int m0 = tm * 256;
    const int cb = split * p.itPerSplit;                       // chunk range of this split
    const int ce = min(p.nIt, cb + p.itPerSplit);

    const int H = p.H, W = p.W, Cc = p.Cc, K = p.K, Ho = p.Ho, Wo = p.Wo;
    const int lgWo = p.lgWo, lgHW = p.lgWo + p.lgHo;
    // tile = R rows x Wo pixels of image n, rows a0 .. a0 + R - 1; window = rows a0 - 1 .. a0 + R, columns -1 .. Wo
    const int WW = Wo + 2;
    const int n_img = m0 >> lgHW, a0 = (m0 >> lgWo) & (Ho - 1);
    const int WR = (256 / Wo + 2) * WW;                       // window rows

    constexpr int OOR = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (p.dbg_zero & 1) ? 0 : (int)p.abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (p.dbg_zero & 2) ? 0 : (int)p.bbytes, 0x00020000);
    auto kmswz = [](int k) -> int { return (k & 3) << 2; };

    // ---- window DMA descriptors: this wave's pieces w + 8 j, j < 5 (rows beyond the window: zero-fill) --------------------------
    // piece pc covers window rows 16 pc .. 16 pc + 15; lane L lands in (row 16 pc + L / 4, slot L % 4), fetches granule
    // slot ^ ((row >> 2) & 3) of gradient pixel (a0 - 1 + row / WW, row % WW - 1)
    int w_ob[NWP];
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
        const int pc = wave + 8 * j;
        const int row = pc * 16 + (lane >> 2);
        const int g = (lane & 3) ^ ((row >> 2) & 3);
        const int wr = row / WW, wc = row - wr * WW;
        const int a = a0 - 1 + wr, b = wc - 1;
        const bool ok = row < WR && (unsigned)a < (unsigned)Ho && (unsigned)b < (unsigned)Wo;
        w_ob[j] = ok ? ((((n_img * Ho + a) * Wo + b) * K) + g * 8) * 2 : OOR;
    }
    // ---- weight DMA descriptors: tile [32 k][256 columns], column = class-local-index * CW + c; pieces 2 w, 2 w + 1 ------------
    // piece pq = k rows 2 pq, 2 pq + 1; lane L lands in (k row 2 pq + L / 32, slot L % 32), fetches granule slot ^ kmswz(k row);
    // the tap (r, s) of (class, t) is added per step from the lane's class bits
    int b_base[WPS], b_cls[WPS];          // b_cls: the lane's class bits (ph << 1 | pw) -- per piece: the swizzle moves a lane between classes
#pragma unroll
    for (int i = 0; i < WPS; ++i) {
        const int krow = (wave * 2 + i) * 2 + (lane >> 5);
        const int gc = (lane & 31) ^ kmswz(krow);
        const int col = gc * 8;
        const int bcls = col / CW, cc = col - bcls * CW;
        b_cls[i] = NCLS == 4 ? bcls : (zph * 2 + bcls);
        b_base[i] = cc < Cc ? (krow * 16 * Cc + cc) * 2 : OOR;
    }

    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_ptr)smem;
    auto dma = [&](const __amdgpu_buffer_rsrc_t& r, int lds_off, int voff) {
        unsigned keep;
        const unsigned dst = lds_base + (unsigned)lds_off;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(r), "s"(dst)
                     : "memory");
    };
    // window piece j of chunk `c` into window stage `ast`
    auto issue_window = [&](int ast, int j, int c) {
        dma(rA, A_OFF + ast * AST + (wave + 8 * j) * 1024, w_ob[j] + c * KT * 2);
    };
    // weight tile of step (chunk c, tap t) into weight stage `bst` (runtime: 0..2)
    auto issue_weights = [&](int bst, int c, int t) {
        const int ty = t >> 1, tx = t & 1;                                  // wave-uniform
#pragma unroll
        for (int i = 0; i < WPS; ++i) {
            const int lph = b_cls[i] >> 1, lpw = b_cls[i] & 1;              // per lane (the lane's class)
            const int r = lph == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 2 : 0);
            const int sx = lpw == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 2 : 0);
            dma(rB, B_OFF + bst * BST + (wave * 2 + i) * 1024, b_base[i] + ((r * 4 + sx) * Cc + c * (KT * 16) * Cc) * 2);
        }
    };

    // ---- fragment reads ----------------------------------------------------------------------------------------------
    const int cls = NCLS == 4 ? wn : (wn >> 1);
    const int ph = NCLS == 4 ? (cls >> 1) : zph, pw = NCLS == 4 ? (cls & 1) : cls;
    int shift[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ty = t >> 1, tx = t & 1;
        const int dyo = ph == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 0 : 1);
        const int dxo = pw == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 0 : 1);
        shift[t] = dyo * WW + dxo;
    }
    int srow[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int ml = wm * (32 * FM) + i * 32;
        srow[i] = ((ml >> lgWo) + 1) * WW + (ml & (Wo - 1)) + 1;
    }
    const int tr_q = (lane >> 2) & 3, tr_c = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    auto frag_km = [&](const char* img, int k0, int c0) -> bf16x8 {
        const int kr = k0 + tr_q, col = c0 + tr_c;
        const char* p0 = img + kr * 512 + ((((col >> 3) ^ kmswz(kr))) << 4) + (col & 7) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)p0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(p0 + 4 * 512));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 fa[2][FM], fb[2][FN];          // [k16 step][block]
    // (the window row is recomputed at every fetch: see igemm_dma_x3_dgw.hip)
    auto fetchA = [&](int ast, int s16, int i, int t) {
        int lv = l31;
        asm volatile("" : "+v"(lv));
        const int row = lv + (srow[i] + shift[t]);
        fa[s16][i] = *(const bf16x8*)(smem + A_OFF + ast * AST + row * 64 + (((2 * s16 + lh) ^ ((row >> 2) & 3)) << 4));
    };
    auto fetchB = [&](int bst, int s16, int j) {
        fb[s16][j] = frag_km(smem + B_OFF + bst * BST, s16 * 16 + 8 * lh, wn * (32 * FN) + j * 32);
    };
    auto fetch = [&](int ast, int bst, int s16, int t) {      // order of use: A0, B0, B1, A1, A2, A3
        fetchA(ast, s16, 0, t);
        fetchB(bst, s16, 0);
        fetchB(bst, s16, 1);
        fetchA(ast, s16, 1, t);
        fetchA(ast, s16, 2, t);
        fetchA(ast, s16, 3, t);
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- DMA-side state: the step (wc, wt) whose weight tile is issued next (clamped to the last step) and its stage -------------
    int wc = cb, wt = 0;
    auto advance = [&]() {
        const int last = (wc == ce - 1 && wt == 3) ? 1 : 0;
        wt += 1 - last;
        const int wrap = wt == 4 ? 1 : 0;
        wt = wrap ? 0 : wt;
        wc += wrap;
    };
    int bcur = 0;                         // weight stage of the current step (runtime, wave-uniform): step index % 3

    // ---- prologue: window of the first chunk, weight tiles of steps 0, 1, 2 ----------------------------------------------
    if (cb < ce) {
#pragma unroll
        for (int j = 0; j < NWP; ++j) issue_window(0, j, cb);
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            issue_weights(st, wc, wt);
            advance();
        }
        // issued so far per wave: NWP window pieces, then 3 x WPS weight pieces; all but the last two weight tiles have landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WPS) : "memory");      // window and the first weight tile
    }
    __builtin_amdgcn_s_barrier();
    if (stp) stp[2] = clock64();
    fetch(0, 0, 0, 0);

    // ---- one step (chunk c in window stage AS, tap T, weight stage bcur): 16 MFMAs per wave -------------------------------------
    // counted wait in front of the barrier: everything issued BEFORE the previous step's barrier has landed = the next step's
    // weight tile (issued two steps ago) and, in front of tap 3, the whole next window (issued in steps 0 and 1); what the
    // previous step issued (2 weight pieces + its window pieces) may still be in flight
    auto body = [&](auto AS_, auto T_, int c) {
        constexpr int AS = decltype(AS_)::value, T = decltype(T_)::value;
        constexpr int NT = (T + 1) & 3, NAS = T == 3 ? AS ^ 1 : AS;
        // in flight and allowed to stay so across this step's barrier: what the PREVIOUS step issued (its WPS weight pieces and
        // its NWIN window pieces); everything older -- this step's successor's weight tile, the whole next window in front of
        // tap 3 -- has landed when the counter is down to that number, in EVERY wave, because every wave issued the same count
        constexpr int VM = WPS + NWIN[(T + 3) & 3];
        const int bnext = bcur == 2 ? 0 : bcur + 1;
        dgwb_static_for(std::make_integer_sequence<int, 16>{}, [&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            constexpr int s16 = q / 8, w = q % 8, i = w / 2, j = w % 2;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (q == 8) {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM) : "memory");
                __builtin_amdgcn_s_barrier();
                fetch(NAS, bnext, 0, NT);
            }
            if constexpr (q == 1) fetch(AS, bcur, 1, T);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s16][i], fb[s16][j], acc[i][j], 0, 0, 0);
            if constexpr (q == 9) {                                  // the tile of step + 3 into the stage this step has left
                issue_weights(bcur, wc, wt);
                advance();
            }
            static_assert(10 + NWIN[0] <= 16 && 10 + NWIN[1] <= 16, "window pieces ride behind MFMAs 10.. of the step");
            if constexpr (T == 0 && q >= 10 && q < 10 + NWIN[0]) issue_window(AS ^ 1, q - 10, min(c + 1, ce - 1));
            if constexpr (T == 1 && q >= 10 && q < 10 + NWIN[1]) issue_window(AS ^ 1, q - 10 + NWIN[0], min(c + 1, ce - 1));
        });
        __builtin_amdgcn_sched_barrier(0);
        bcur = bnext;
    };
    for (int c = cb; c < ce; c += 2) {
        dgwb_static_for(std::make_integer_sequence<int, 8>{}, [&](auto B_) {
            constexpr int bi = decltype(B_)::value;
            if (c + bi / 4 < ce) body(std::integral_constant<int, bi / 4>{}, std::integral_constant<int, bi % 4>{}, c + bi / 4);
        });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (stp) stp[3] = clock64();

    // ---- fused BatchNorm statistics (igemm.hip's scheme; from the fp32 accumulators, before any rounding of the output to bf16): one partial row per (parity class, tile, wave row) ---------------------
    if (p.stat != nullptr && p.part == nullptr) {
        const int row0 = m0 + wm * (32 * FM);
        const int nrows = min(32 * FM, p.M - row0);
        const int prow = ((ph * 2 + pw) * p.tilesM + tm) * 2 + wm;
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

    // ---- epilogue: the wave's 128 pixels x 64 columns of class (ph, pw); rows -> out pixel (2a + ph, 2b + pw) -----------------
    const bool to_part = p.part != nullptr;
    float* const eps = (float*)smem + wave * (32 * 68);
    const int erow = lane >> 4, ec4 = (lane & 15) * 4;
    const int ccol = (NCLS == 4 ? 0 : (wn & 1) * 64) + ec4;            // column inside the class
    const int parity = ph * 2 + pw;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
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
            const int m = m0 + wm * (32 * FM) + i * 32 + row;
            if (m >= p.M || ccol >= Cc) continue;
            float* dst;
            long eoff;
            if (to_part) {
                dst = p.part;
                eoff = (((long)split * 4 + parity) * p.M + m) * Cc + ccol;
            } else {
                const int b = m & (Wo - 1), a = (m >> lgWo) & (Ho - 1), n = m >> lgHW;
                dst = p.C;
                eoff = (long)((n * H + 2 * a + ph) * W + 2 * b + pw) * Cc + ccol;
            }
            if (!to_part && p.accumulate) v += *(const f32x4*)(dst + eoff);
            dg_store_out4(dst, eoff, v, to_part ? 0 : p.out16);
        }
    }
    if (stp) {
        stp[4] = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stp[5] = wall_clock64();
        stp[7] = clock64();
    }
}

// host: launch for a plan made by igemm.hip (ncls = 4 / 2); grid = pixel tiles x (4 / ncls) x splits
int dg_igemm_bf16_dgw_launch(int ncls, const IgemmArgs& a, hipStream_t st) {
    const int grid = a.tilesM * (4 / ncls) * a.splits;
    if (ncls == 4) hipLaunchKernelGGL((igemm_bf16_dgw_kernel<4>), dim3(grid), dim3(512), 0, st, a);
    else if (ncls == 2) hipLaunchKernelGGL((igemm_bf16_dgw_kernel<2>), dim3(grid), dim3(512), 0, st, a);
    else return 0;
    return 1;
}
