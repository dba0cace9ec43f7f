// Forward of Conv2d(k4,s2,p1) / input-grad of ConvTranspose2d(k4,s2,p1) for layers with FEW output channels (K <= 128), f32x3 plane
// operands: the input fetched ONCE per (16-channel chunk, input-parity class) into an LDS WINDOW and re-used by the class's four
// taps ("fww" = forward with window) -- the mirror image of igemm_dma_x3_dgw.hip.
//
// Replaces (reference file:line) nn.Conv2d(k4,s2,p1) forward (model.py:11,83: 64 -> 128 channels, 256 -> 128 px) and the input-grad
// of nn.ConvTranspose2d(k4,s2,p1) (model.py:138: 128 -> 64, via autograd) where igemm_dma_x3.hip's 256 x 256 tile does not apply
// (fewer than 192 GEMM columns): until round 3 these ran on the register-staged f32x3 tiles at 142-148 TFLOP/s (every input element
// fetched by four (tap, pixel) pairs of a workgroup, the split into planes redone in every launch).
//
// Why a window works here too: y[oy][ox] = sum_{r,s} x[2 oy - 1 + r][2 ox - 1 + s] . w[k][r][s][c].  Split the input by pixel parity,
// X_q[a][b] = x[2a + qy][2b + qx]: filter row r reads X_{qy}[oy + da] with (qy, da) = (1,-1), (0,0), (1,0), (0,+1) for r = 0..3 (columns
// alike), so every parity class q contributes 2 x 2 taps at shifts da, db in {0, +-1} -- a stride-1 2x2 convolution per class, and
// the forward conv is their SUM over the four classes (the input-grad of the dgw kernel is the same identity read the other way:
// there the classes are output columns, here they are reduction steps).
//   * a workgroup owns 256 consecutive OUTPUT pixels (R = 256 / Wo rows of one image) x all K <= 128 columns; 8 waves of 128 x 32;
//   * K loop over super-chunks sc = (16-channel chunk c, class q), 4 taps each: per super-chunk the (R + 2) x (Wo + 2) window of
//     X_q goes to LDS once per plane -- the dgw kernel's window, with the per-lane DMA source stepping TWO input pixels per window
//     pixel (the de-interleave costs nothing: an LDS-DMA lane fetches any 16-byte granule); halo pixels outside the image are the
//     conv's zero padding = out-of-range DMA offsets;
//   * a tap's A fragment is the window row of the pixel + da (Wo + 2) + db (conflict-free swizzle of the dgw kernel);
//   * weight tiles come from the TRANSPOSED weight planes wT[(r, s, c)][k] (dg_x3_transpose_planes): [16 c][K] rows of 2 K bytes;
//     a weight stage holds the two taps of a tap PAIR (same filter row), so every wave issues one 1-KiB piece per plane and pair;
//   * LDS: 2 window stages + 2 weight stages = 150 KB, one workgroup per CU; barrier / fragment-replacement scheme of the dgw kernel
//     with 24 MFMAs per step and wave (4 blocks x 6 plane products).
// Same six MFMAs per product block; the reduction order per output element is (chunk, class, tap).
#include "igemm_args.h"
#include <type_traits>
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4* lds_bf4_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int... Q, typename F>
__device__ __forceinline__ void fww_static_for(std::integer_sequence<int, Q...>, F&& f) {
    (f(std::integral_constant<int, Q>{}), ...);
}

__global__ __launch_bounds__(512, 2) void igemm_x3_fww_kernel(const IgemmArgs p) {
    constexpr int WN = 4, FM = 4, KT = 16;              // 2 x 4 waves of 128 x 32
    constexpr int WPMAX = 17;                           // window pieces (32 rows each) per plane: (R + 2)(Wo + 2) <= 544 rows
    constexpr int WPB = WPMAX * 1024;                   // bytes per window plane
    constexpr int AST = 3 * WPB;                        // window stage
    constexpr int TAPB = KT * 256;                      // one tap's weight tile of a plane: [16 c][128 k] bf16 = 4 KB
    constexpr int PLB = 2 * TAPB, BST = 3 * PLB;        // a weight stage holds a tap PAIR per plane
    constexpr int B_OFF = 0, A_OFF = 2 * BST;
    constexpr int LDS_BYTES = 2 * AST + 2 * BST;
    constexpr int DUMP_OFF = LDS_BYTES;                 // 1 KiB that absent window pieces are zero-filled into (uniform DMA counts)
    static_assert(LDS_BYTES + 1024 <= 160 * 1024, "LDS");
    static_assert(8 * 32 * 36 * 4 <= LDS_BYTES, "epilogue transpose regions live in the operand stages");
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES + 1024];
    // DMA schedule, the basis of the counted `vmcnt` waits: EVERY wave issues, unconditionally, NWIN window pieces behind the barrier
    // of tap 0 (absent pieces: all lanes out of range, zero-filled into the dump KiB) and NWT weight pieces behind the barriers of
    // taps 1 and 3.  In front of the barrier of tap T a wave may leave in flight what was issued after the loads the step needs:
    //   T = 0: the pair-1 weights issued in the previous tap 3                                   -> NWT
    //   T = 1: the window issued in tap 0 (needs: pair-1 weights of THIS super-chunk, issued before it) -> NWIN
    //   T = 2: window + the pair-0 weights of the next super-chunk issued in tap 1              -> NWIN + NWT
    //   T = 3: nothing (the next window and its pair-0 weights are read behind this barrier)        -> 0
    constexpr int NWIN = 9, NWT = 3;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- blockIdx -> pixel tile: XCD x takes the tiles [x tilesM / 8, (x + 1) tilesM / 8) in order (adjacent tiles share window rows)
    int tm = blockIdx.x;
    if ((p.tilesM & 7) == 0) tm = (tm & 7) * (p.tilesM >> 3) + (tm >> 3);
    const int m0 = tm * 256;

    const int H = p.H, W = p.W, Cc = p.Cc, K = p.K, Ho = p.Ho, Wo = p.Wo;
    const int lgWo = p.lgWo, lgHW = p.lgWo + p.lgHo;
    const int WW = Wo + 2;
    const int n_img = m0 >> lgHW, a0 = (m0 >> lgWo) & (Ho - 1);
    const int WR = (256 / Wo + 2) * WW;                       // window rows
    const int wpieces = (WR + 31) >> 5;
    const int nCh = Cc >> 4;
    const int nSC = nCh * 4;                                  // super-chunks: (channel chunk, parity class)
    // order of the super-chunks: class-major (the chunks of one parity class back to back: the 128-byte lines of a pixel -- 4
    // chunks of 32 bytes -- are requested in consecutive steps) or chunk-major (dbg_zero bit 2: timing experiment)
    const bool cmaj = (p.dbg_zero & 4) != 0;
    auto sc_c = [&](int sc) -> int { return cmaj ? (sc >> 2) : (sc % nCh); };
    auto sc_q = [&](int sc) -> int { return cmaj ? (sc & 3) : (sc / nCh); };

    constexpr int OOR = (int)0x80000000;
    __amdgpu_buffer_rsrc_t rA[3], rB[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        rA[pl] = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + pl * p.a_plane), 0, (p.dbg_zero & 1) ? 0 : (int)p.abytes, 0x00020000);
        rB[pl] = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + pl * p.b_plane), 0, (p.dbg_zero & 2) ? 0 : (int)p.bbytes, 0x00020000);
    }
    auto kmswz = [](int k) -> int { return (k & 3) << 2; };

    // ---- window DMA descriptors (class (0, 0), chunk 0): this wave's pieces w, w + 8, w + 16 of every plane --------------------------
    // piece pc covers window rows 32 pc .. 32 pc + 31; lane L lands in (row 32 pc + L / 2, slot L % 2), fetches granule
    // slot ^ ((row >> 3) & 1) of input pixel (2 (a0 - 1 + row / WW), 2 (row % WW - 1)); outside the sub-image: zeros (conv padding)
    int w_ob[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pc = wave + 8 * j;
        const int row = pc * 32 + (lane >> 1);
        const int g = (lane & 1) ^ ((row >> 3) & 1);
        const int wr = row / WW, wc = row - wr * WW;
        const int a = a0 - 1 + wr, b = wc - 1;
        const bool ok = row < WR && (unsigned)a < (unsigned)Ho && (unsigned)b < (unsigned)Wo;
        w_ob[j] = ok ? ((((n_img * H + 2 * a) * W + 2 * b) * Cc) + g * 8) * 2 : OOR;
#ifdef DG_TIMING_KNOBS
        // timing experiment (WRONG results), dbg_zero bit 128: what would the kernel cost if a window were one CONTIGUOUS run of its plane
        // (a parity-split chunk-major layout)?  Lane L of piece pc fetches granule 64 pc + L of a run that starts at a per-tile offset.
        if (p.dbg_zero & 128) w_ob[j] = row < WR ? (int)(((long)tm * 16 * WR * 32) % (long)(p.abytes - 17 * 1024 * 17)) + (pc * 64 + lane) * 16 : OOR;
#endif
    }
    // ---- weight DMA descriptor: stage image per plane = [tap of the pair][16 c][128 k]; this wave's piece = 4 rows of one tap -------
    // lane L lands in (row 4 (w & 3) + L / 16, slot L % 16) of tap w >> 2, fetches granule slot ^ kmswz(row) of wT row (r, s, c0 + row)
    int b_base;
    {
        const int krow = 4 * (wave & 3) + (lane >> 4);
        const int gc = (lane & 15) ^ kmswz(krow);
        const int col = gc * 8;
        b_base = col < K ? (krow * K + col) * 2 : OOR;
    }
    const int b_tx = wave >> 2;                                // the tap of the pair this wave's piece belongs to

    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_ptr)smem;
    auto dma = [&](const __amdgpu_buffer_rsrc_t& r, int lds_off, int voff) {
        unsigned keep;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)lds_off);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(r), "s"(dst)
                     : "memory");
    };
    // the whole window (three planes) of super-chunk sc into window stage `ast`
    auto issue_window = [&](int ast, int sc) {
        const int c = sc_c(sc), q = sc_q(sc), qy = q >> 1, qx = q & 1;
        int scoff = ((qy * W + qx) * Cc + c * KT) * 2;
#ifdef DG_TIMING_KNOBS
        if (p.dbg_zero & 128) scoff = sc * WR * 32;
#endif
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bool real = wave + 8 * j < wpieces;                     // wave-uniform
                dma(rA[pl], real ? A_OFF + ast * AST + pl * WPB + (wave + 8 * j) * 1024 : DUMP_OFF, real ? w_ob[j] + scoff : OOR);
            }
    };
    static_assert(NWIN == 3 * 3 && NWT == 3, "issue_window / issue_weights issue 9 / 3 pieces per wave");
    // weight tiles of tap pair `ty` of super-chunk sc into weight stage `bst` (three planes): filter row r from (qy, ty), this wave's
    // filter column s from (qx, tx = b_tx)
    auto issue_weights = [&](int bst, int sc, int ty) {
        const int c = sc_c(sc), q = sc_q(sc), qy = q >> 1, qx = q & 1;
        const int r = qy == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 2 : 0);
        const int sx = qx == 0 ? (b_tx == 0 ? 1 : 3) : (b_tx == 0 ? 2 : 0);
        const int voff = b_base + (((r * 4 + sx) * Cc + c * KT) * K) * 2;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dma(rB[pl], B_OFF + bst * BST + pl * PLB + wave * 1024, voff);
    };
    // window-row shift of tap t = (ty, tx) of class q
    auto shift_of = [&](int sc, int t) -> int {
        const int q = sc_q(sc), qy = q >> 1, qx = q & 1, ty = t >> 1, tx = t & 1;
        const int da = ty == 0 ? 0 : (qy ? -1 : 1);
        const int db = tx == 0 ? 0 : (qx ? -1 : 1);
        return da * WW + db;
    };

    // ---- fragment reads ----------------------------------------------------------------------------------------------
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
    auto frag_km = [&](const char* img, int c0) -> bf16x8 {      // [16 rows][256 bytes] tap image
        const int kr = 8 * lh + tr_q, col = c0 + tr_c;
        const char* p0 = img + kr * 256 + ((((col >> 3) ^ kmswz(kr))) << 4) + (col & 7) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)p0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(p0 + 4 * 256));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 fa[3][FM];
    bf16x8 fb[2], fbh[2];                 // mid / lo planes of the current tap; hi plane double-buffered over the steps
    auto fetchA = [&](int ast, int pl, int i, int sh) {
        int lv = l31;
        asm volatile("" : "+v"(lv));
        fa[pl][i] = frag_kc(smem + A_OFF + ast * AST + pl * WPB, lv + (srow[i] + sh));
    };
    auto fetchB = [&](int bst, int tx, int pl, int hset) {
        const bf16x8 v = frag_km(smem + B_OFF + bst * BST + pl * PLB + tx * TAPB, wn * 32);
        if (pl == 0) fbh[hset] = v;
        else fb[pl - 1] = v;
    };

    f32x16 acc[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- prologue: window of super-chunk 0, both tap pairs of its weights ---------------------------------------------------
    issue_window(0, 0);
    issue_weights(0, 0, 0);
    issue_weights(1, 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const int sh0 = shift_of(0, 0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            fetchB(0, 0, pl, 0);
#pragma unroll
            for (int i = 0; i < FM; ++i) fetchA(0, pl, i, sh0);
        }
    }

    // ---- one step (super-chunk sc in window stage AS, tap T): 24 MFMAs per wave --------------------------------------------------
    // weight stage of tap T = T >> 1 (the pair), tap image T & 1.  Next step: tap T + 1 of the same window, or tap 0 of super-chunk
    // sc + 1 (window stage AS ^ 1, issued behind the barrier of step T == 0 of sc; weights: pair 0 issued in step T == 1, pair 1 in
    // step T == 3 -- each two steps before its first fragment read).
    auto body = [&](auto AS_, auto T_, int sc) {
        constexpr int AS = decltype(AS_)::value, T = decltype(T_)::value;
        constexpr int HS = T & 1;                               // hi-plane register set of this step
        constexpr int NT = (T + 1) & 3, NAS = T == 3 ? AS ^ 1 : AS;
        constexpr int NBST = NT >> 1, NTX = NT & 1;
        constexpr int QB = 4;
        const int nsc = T == 3 ? min(sc + 1, nSC - 1) : sc;     // super-chunk of the next step (clamped: re-read, never used)
        const int nsh = shift_of(nsc, NT);
        fww_static_for(std::make_integer_sequence<int, 24>{}, [&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            constexpr int i = q / 6, pr = q % 6;
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (q == QB) {
                constexpr int VM = T == 0 ? NWT : (T == 1 ? NWIN : (T == 2 ? NWIN + NWT : 0));
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM) : "memory");
                __builtin_amdgcn_s_barrier();
                fetchB(NBST, NTX, 0, HS ^ 1);
            }
            if constexpr (pr == 1 && i > 0 && q > QB) {
                fetchA(NAS, 0, i - 1, nsh);
                fetchA(NAS, 1, i - 1, nsh);
                fetchA(NAS, 2, i - 1, nsh);
            }
            if constexpr (q == 20) fetchB(NBST, NTX, 2, 0);      // lo: last used by MFMA 19
            if constexpr (q == 23) fetchB(NBST, NTX, 1, 0);      // mid: last used by MFMA 22
            if constexpr (PB[pr] == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pr]][i], fbh[HS], acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pr]][i], fb[PB[pr] - 1], acc[i], 0, 0, 0);
            // behind the barrier: T == 0: the next super-chunk's window into the other window stage; T == 1 / T == 3: the next
            // super-chunk's tap pair 0 / 1 into the weight stage whose last fragment reads ended with the previous step
            if constexpr (q == QB + 1) {
                if constexpr (T == 0) issue_window(AS ^ 1, min(sc + 1, nSC - 1));
                if constexpr (T == 1) issue_weights(0, min(sc + 1, nSC - 1), 0);
                if constexpr (T == 3) issue_weights(1, min(sc + 1, nSC - 1), 1);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        fetchA(NAS, 0, 3, nsh);
        fetchA(NAS, 1, 3, nsh);
        fetchA(NAS, 2, 3, nsh);
    };
    for (int sc = 0; sc < nSC; sc += 2) {
        fww_static_for(std::make_integer_sequence<int, 8>{}, [&](auto B_) {
            constexpr int bi = decltype(B_)::value;
            if (sc + bi / 4 < nSC) body(std::integral_constant<int, bi / 4>{}, std::integral_constant<int, bi % 4>{}, sc + bi / 4);
        });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: the wave's 128 pixels x 32 columns through a private [32][36] LDS region, 16-byte stores -----------------------
    float* const eps = (float*)smem + wave * (32 * 36);
    const int erow = lane >> 3, ec4 = (lane & 7) * 4;
    const int col = wn * 32 + ec4;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            eps[lr * 36 + l31] = acc[i][r];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = t * 8 + erow;
            f32x4 v = *(const f32x4*)(eps + row * 36 + ec4);
            const int m = m0 + wm * (32 * FM) + i * 32 + row;
            if (m >= p.M || col >= K) continue;
            const long eoff = (long)m * K + col;
            if (p.accumulate) v += *(const f32x4*)(p.C + eoff);
            *(f32x4*)(p.C + eoff) = v;
        }
    }
}

// host: launch for a plan made by igemm.hip; grid = pixel tiles (no split-K: the grid is 8 tiles per CU at the benchmark shape)
int dg_igemm_x3_fww_launch(const IgemmArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(igemm_x3_fww_kernel, dim3(a.tilesM), dim3(512), 0, st, a);
    return 1;
}
