// 3-channel edge layers, image side NCHW (the reference hands over / receives NCHW images:
// image_translation.py:332-333, model.py:8,80,142): the 3 -> K forward for K == 64 (the tiled MODE_FWD_C3 path
// in igemm.hip is the fallback for other K), the K -> 3 direction and the [K][3][4][4] weight gradient.
// All are HBM-bound (AI ~ 20 FLOP/B); the hot kernels feed the matrix cores so the VALU stays free.
//
//   c3_fwd (K == 64): a wave owns groups of 32 consecutive output pixels; MFMA operands gathered straight from
//       the NCHW image with range-checked buffer loads (no LDS, no barriers after the weights are staged).
//   c3_dgrad (K == 64), scatter form: P[pixel][48 = (c,r,s)] = dy[pixel][64] . W[64][48] is a DENSE GEMM on
//       v_mfma_f32_16x16x4_f32 (M = 16 pixels, N = 3 x 16 columns, the [64][48] weights in 48 registers per lane, the
//       dy rows loaded straight into MFMA registers); out[c][2a-1+r][2b-1+s] += P[a][b][c][r][s] is a 4-term
//       overlap-add through LDS inside a tile of 6 x 14 pixels computed with a 1-pixel halo (8 x 16).  The
//       gather form it replaces (9 neighbours x 64 k against a 1/3-dense [576][16] weight image) needed 144 MFMA
//       column-steps per 16 pixels, this one 48 x (8*16)/(6*14) = 73.
//   c3_wgrad: dw[64][48] = dy^T [64 x pixels] . im2col(x) [pixels x 48] on v_mfma_f32_32x32x2_f32; every wave
//       streams its own pixel range with operands loaded straight into MFMA registers (no LDS, no
//       barriers); per-wave partial slabs are summed by a fixed-order reduction kernel (deterministic).
#include "dg_common.h"
#include <type_traits>

__device__ float dg_zero_edge[64];   // zero-initialised; masked lanes read from here instead of branching

// Taps per output parity (conv k4 s2 p1): p=0 -> (r=1, d=0), (r=3, d=-1);  p=1 -> (r=2, d=0), (r=0, d=+1).
__device__ __forceinline__ int tap_of(int parity, int d) {
    // returns the filter index r used by output parity `parity` for neighbour offset d, or -1
    if (parity == 0) return d == 0 ? 1 : (d == -1 ? 3 : -1);
    return d == 0 ? 2 : (d == 1 ? 0 : -1);
}

// ---- generic VALU fallback (any K % 4 == 0) ----------------------------------------------------------
__global__ __launch_bounds__(256) void c3_dgrad_valu_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int N, int H, int W, int K,
                                                            int lgHo, int lgWo, int act) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long nquad = (long)N * Ho * Wo;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquad) return;
    const int b = (int)(q & (Wo - 1)), a = (int)((q >> lgWo) & (Ho - 1)), n = (int)(q >> (lgWo + lgHo));
    float acc[3][2][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c][0][0] = acc[c][0][1] = acc[c][1][0] = acc[c][1][1] = 0.f;
    const float* base = dy + ((long)(n * Ho + a) * Wo + b) * K;
    for (int k0 = 0; k0 < K; k0 += 4) {
        f32x4 g[3][3];
#pragma unroll
        for (int da = -1; da <= 1; ++da)
#pragma unroll
            for (int db = -1; db <= 1; ++db) {
                const bool ok = (unsigned)(a + da) < (unsigned)Ho && (unsigned)(b + db) < (unsigned)Wo;
                g[da + 1][db + 1] = ok ? *(const f32x4*)(base + (long)(da * Wo + db) * K + k0)
                                       : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float* wk = w + (long)(k0 + kk) * 48;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                    for (int ty = 0; ty < 2; ++ty) {
                        const int r = ph == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 2 : 0);
                        const int da = ph == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 0 : 1);
#pragma unroll
                        for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                            for (int tx = 0; tx < 2; ++tx) {
                                const int s = pw == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 2 : 0);
                                const int db = pw == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 0 : 1);
                                acc[c][ph][pw] += g[da + 1][db + 1][kk] * wk[c * 16 + r * 4 + s];
                            }
                    }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            float v0 = acc[c][ph][0], v1 = acc[c][ph][1];
            if (act == DG_ACT_SIGMOID) {
                v0 = 1.f / (1.f + expf(-v0));
                v1 = 1.f / (1.f + expf(-v1));
            }
            *(float2*)(dx + ((long)(n * 3 + c) * H + 2 * a + ph) * W + 2 * b) = make_float2(v0, v1);
        }
}

// ---- MFMA 16x16x4 version, K == 64 ----------------------------------------------------------------------
#define CD_K 64
#define CD_TR 4                 // quad rows per tile = waves per workgroup
#define CD_TC 16                // quad columns per tile = MFMA M
#define CD_PR (CD_TR + 2)       // staged dy pixel rows
#define CD_PC (CD_TC + 2)       // staged dy pixel columns
#define CD_LDP 66               // floats per staged pixel (64 + 2: conflict-free b32 A reads, 8-byte aligned rows)
#define CD_NK (9 * CD_K)        // GEMM K = 576

// weight image Wq[(nb, k)][col = c*4 + ph*2 + pw], built once per call into the workspace
__global__ __launch_bounds__(256) void c3_wq_build_kernel(const float* __restrict__ w, float* __restrict__ wq) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= CD_NK * 16) return;
    const int col = e & 15, row = e >> 4;
    const int k = row & (CD_K - 1), nb = row >> 6;
    const int da = nb / 3 - 1, db = nb % 3 - 1;
    float v = 0.f;
    if (col < 12) {
        const int c = col >> 2, ph = (col >> 1) & 1, pw = col & 1;
        const int r = tap_of(ph, da), s = tap_of(pw, db);
        if (r >= 0 && s >= 0) v = w[(k * 3 + c) * 16 + r * 4 + s];
    }
    wq[e] = v;
}

#define CD_NV ((CD_PR * CD_PC * (CD_K / 4) + 255) / 256)   // float4 per thread per staged tile (7)
__global__ __launch_bounds__(256, 2) void c3_dgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ wq,
                                                               float* __restrict__ dx, int N, int H, int W, int act,
                                                               int tiles_r, int tiles_c, int ntiles) {
    __shared__ __attribute__((aligned(16))) float WqS[CD_NK * 16];                 // [576][16]
    __shared__ __attribute__((aligned(16))) float dyS[CD_PR * CD_PC * CD_LDP];     // [6*18][66]
    __shared__ float outS[CD_TR][16 * 17];                                          // per wave [col][quad]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ho = H >> 1, Wo = W >> 1;
    for (int e = tid; e < CD_NK * 4; e += 256) *(f32x4*)(WqS + e * 4) = *(const f32x4*)(wq + e * 4);

    f32x4 pre[CD_NV];
    auto prefetch = [&](int t) {
        const int tc = t % tiles_c, tr = (t / tiles_c) % tiles_r, n = t / (tiles_c * tiles_r);
        const int a0 = tr * CD_TR, b0 = tc * CD_TC;
#pragma unroll
        for (int i = 0; i < CD_NV; ++i) {
            const int idx = tid + i * 256;
            const int q4 = idx & (CD_K / 4 - 1), pix = idx >> 4;
            const int pr = pix / CD_PC, pc = pix - pr * CD_PC;
            const int a = a0 - 1 + pr, b = b0 - 1 + pc;
            const bool ok = idx < CD_PR * CD_PC * (CD_K / 4) && t < ntiles &&
                            (unsigned)a < (unsigned)Ho && (unsigned)b < (unsigned)Wo;
            pre[i] = ok ? *(const f32x4*)(dy + ((long)(n * Ho + a) * Wo + b) * CD_K + q4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    const int bq = lane & 15, kk = lane >> 4;
    prefetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tc = t % tiles_c, tr = (t / tiles_c) % tiles_r, n = t / (tiles_c * tiles_r);
        const int a0 = tr * CD_TR, b0 = tc * CD_TC;
        __syncthreads();  // previous tile's readers of dyS/outS are done (also orders the WqS copy)
#pragma unroll
        for (int i = 0; i < CD_NV; ++i) {
            const int idx = tid + i * 256;
            if (idx < CD_PR * CD_PC * (CD_K / 4)) {
                float* d = dyS + (idx >> 4) * CD_LDP + (idx & (CD_K / 4 - 1)) * 4;
                *(float2*)d = make_float2(pre[i][0], pre[i][1]);
                *(float2*)(d + 2) = make_float2(pre[i][2], pre[i][3]);
            }
        }
        __syncthreads();
        prefetch(t + gridDim.x);   // next tile's HBM reads fly under this tile's MFMAs
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < 9; ++nb) {
            const int da = nb / 3 - 1, db = nb % 3 - 1;
            const float* ap = dyS + ((wave + 1 + da) * CD_PC + bq + 1 + db) * CD_LDP + kk;
            const float* bp = WqS + (nb * CD_K + kk) * 16 + bq;
#pragma unroll
            for (int ks = 0; ks < CD_K / 4; ks += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[ks * 4], bp[ks * 64], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[ks * 4 + 4], bp[ks * 64 + 64], acc1, 0, 0, 0);
            }
        }
        // C/D layout 16x16: col = lane & 15, row (quad) = (lane >> 4) * 4 + reg
#pragma unroll
        for (int r = 0; r < 4; ++r) outS[wave][bq * 17 + kk * 4 + r] = acc0[r] + acc1[r];
        __syncthreads();
        const int a = a0 + wave;
        const int ph = lane >> 5, x = lane & 31;          // x = 2*quad + pw within the 32-wide output row
        const int qd = x >> 1, pw = x & 1;
        if (a < Ho && b0 + qd < Wo) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = outS[wave][(c * 4 + ph * 2 + pw) * 17 + qd];
                if (act == DG_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
                dx[((long)(n * 3 + c) * H + 2 * a + ph) * W + 2 * b0 + x] = v;
            }
        }
    }
}

// ---- scatter form, K == 64 -------------------------------------------------------------------------------
#define CS_TR 6                 // pixel rows of dy whose outputs a tile completes
#define CS_TC 14                // pixel columns
#define CS_PR (CS_TR + 2)       // computed rows (halo 1)  = 8 = 2 per wave
#define CS_PC 16                // computed columns (halo 1) = one MFMA M group
#define CS_LDP 52               // floats per pixel of the P image (48 + 4: the four k-quarters of a wave store to disjoint banks)
// IN16: dy is bf16 -- a lane's 16 k values of a pixel are 32 contiguous bytes (two 16-byte loads), expanded to fp32 by a
// shift / a mask per value right in front of the MFMA that takes it
// MF16 (with IN16, option "bf16" = 1): the dense GEMM runs on v_mfma_f32_16x16x32_bf16 -- the bf16 dy values ARE the A operand
// (lane (pixel p, kq) holds k = 32 st + 8 kq + j: two 16-byte loads, no unpacking), the weights are rounded to bf16 once per
// lane; 12 MFMAs of 16 cycles per tile and wave instead of 96 of 32.
// X3 (fp32 dy, option "bf16" = 2, the f32x3 path): the same 16x16x32 GEMM with fp32-ACCURATE products -- a lane's 8 fp32 dy values
// of a k32 step and the weights are split into their three bf16 planes in registers (24 significand bits), six MFMAs per block
// (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi): 72 MFMAs of 16 cycles per tile and wave instead of 96 of 32.
// FACT (round 4; dg_conv4x4s2_c3_dgrad_act_p): dy is taken through the backward of the layer's fused LeakyReLU on the way in -- dy * (out > 0 ? 1 :
// slope) with `out` the layer's saved output, loaded with dy's own offsets one tile ahead and applied to the prefetched register set at
// the end of the current tile (one more set of 32 registers, not two) -- instead of a stand-alone act_bwd pass that reads both tensors
// and writes a third (12 B per element at fp32) in front of this kernel.  Same expression as act_bwd_kernel / act_bwd16_kernel (bf16: fp32
// product rounded to bf16, RNE): the input-gradient is bitwise the unfused one.
template <bool IN16, bool MF16 = false, bool X3 = false, bool FACT = false>
__global__ __launch_bounds__(256, 2) void c3_dgrad_scatter_kernel(const DgPtrs dys, const DgPtrs ws_, const DgPtrs dxs, int N, int H, int W, int act,
                                                                  int tiles_r, int tiles_c, int ntiles, unsigned dybytes, const DgPtrs aos, float slope) {
    // grouped launch (dg_conv4x4s2_c3_dgrad_g): blockIdx.y = problem
    const float* __restrict__ dy = dg_pick<const float>(dys, blockIdx.y);
    const float* __restrict__ w = dg_pick<const float>(ws_, blockIdx.y);
    float* __restrict__ dx = dg_pick<float>(dxs, blockIdx.y);
    static_assert(!MF16 || IN16, "the bf16 MFMA form takes a bf16 dy");
    static_assert(!X3 || (!IN16 && !MF16), "the f32x3 form takes an fp32 dy");
    typedef __bf16 bf16x8_d __attribute__((ext_vector_type(8)));
    typedef float f32x8_d __attribute__((ext_vector_type(8)));
    __shared__ __attribute__((aligned(16))) float Ps[CS_PR * CS_PC * CS_LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, kq = lane >> 4;
    const int Ho = H >> 1, Wo = W >> 1;
    // B operand: W[k][col], col = c*16 + r*4 + s (the weight tensor's own order).  MFMA step s multiplies the four k values
    // 16*kq + s (kq = lane >> 4): the k order inside the GEMM is free as long as A uses the same one, and this one lets
    // every lane fetch its 16 A values of a pixel as 64 contiguous bytes.
    float breg[3][(MF16 || X3) ? 1 : 16];
    bf16x8_d breg16[3][2];
    bf16x8_d breg3[X3 ? 3 : 1][3][2];               // X3: [plane][block][k32 step]
    auto split8 = [](const f32x8_d& v, bf16x8_d* pl) {
        pl[0] = __builtin_convertvector(v, bf16x8_d);
        const f32x8_d r1 = v - __builtin_convertvector(pl[0], f32x8_d);
        pl[1] = __builtin_convertvector(r1, bf16x8_d);
        pl[2] = __builtin_convertvector(r1 - __builtin_convertvector(pl[1], f32x8_d), bf16x8_d);
    };
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        if constexpr (X3) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f32x8_d t;
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = w[(32 * st + 8 * kq + j) * 48 + blk * 16 + p];
                bf16x8_d pl[3];
                split8(t, pl);
#pragma unroll
                for (int q = 0; q < 3; ++q) breg3[q][blk][st] = pl[q];
            }
        } else if constexpr (MF16) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f32x8_d t;
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = w[(32 * st + 8 * kq + j) * 48 + blk * 16 + p];
                breg16[blk][st] = __builtin_convertvector(t, bf16x8_d);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) breg[blk][s] = w[(16 * kq + s) * 48 + blk * 16 + p];
        }
    }
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)dybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rao = __builtin_amdgcn_make_buffer_rsrc((void*)(FACT ? dg_pick<const float>(aos, blockIdx.y) : dy), 0, (int)dybytes, 0x00020000);
    constexpr int OOR = (int)0x80000000;
    f32x4 areg[2][2][4];            // [set][row group][4 x float4 = 16 k values]
    f32x4 oreg[FACT ? 2 : 1][4];    // FACT: the saved activation output at the offsets of the set being prefetched
    // FACT: areg[set] <- areg[set] * (oreg > 0 ? 1 : slope), element by element (bf16: both halves of every dword, product rounded RNE)
    auto apply_act = [&](int set) {
        if constexpr (FACT) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < (IN16 ? 2 : 4); ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (IN16) {
                            const unsigned dv = __float_as_uint(areg[set][g][j][e]), ov = __float_as_uint(oreg[g][j][e]);
                            float d0 = __builtin_bit_cast(float, dv << 16), d1 = __builtin_bit_cast(float, dv & 0xffff0000u);
                            d0 = __builtin_bit_cast(float, ov << 16) > 0.f ? d0 : d0 * slope;
                            d1 = __builtin_bit_cast(float, ov & 0xffff0000u) > 0.f ? d1 : d1 * slope;
                            typedef __bf16 bf16x2_a __attribute__((ext_vector_type(2)));
                            typedef float f32x2_a __attribute__((ext_vector_type(2)));
                            areg[set][g][j][e] = __uint_as_float(__builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_a){d0, d1}, bf16x2_a)));
                        } else {
                            const float d = areg[set][g][j][e];
                            areg[set][g][j][e] = oreg[g][j][e] > 0.f ? d : d * slope;
                        }
                    }
        }
    };
    auto fetch = [&](int t, int set) {
        const int tc = t % tiles_c, tr = (t / tiles_c) % tiles_r, n = t / (tiles_c * tiles_r);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int a = tr * CS_TR - 1 + wave + 4 * g, b = tc * CS_TC - 1 + p;
            const bool ok = t < ntiles && (unsigned)a < (unsigned)Ho && (unsigned)b < (unsigned)Wo;
            const int off = ok ? (((n * Ho + a) * Wo + b) * CD_K + ((MF16 || X3) ? 8 : 16) * kq) * (IN16 ? 2 : 4) : OOR;   // out of range reads 0 = zero padding
#pragma unroll
            for (int j = 0; j < (IN16 ? 2 : 4); ++j) {    // X3: float4 j holds k = 32 (j >> 1) + 8 kq + 4 (j & 1) ..+3
                const int o = off + (X3 ? 128 * (j >> 1) + 16 * (j & 1) : (MF16 ? 64 : 16) * j);
                areg[set][g][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, o, 0, 0));
                if constexpr (FACT) oreg[g][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rao, o, 0, 0));
            }
        }
    };
    fetch(blockIdx.x, 0);
    apply_act(0);
    auto tile = [&](auto SB, int t) {
        constexpr int S = decltype(SB)::value;
        const int tc = t % tiles_c, tr = (t / tiles_c) % tiles_r, n = t / (tiles_c * tiles_r);
        const int a0 = tr * CS_TR, b0 = tc * CS_TC;
        fetch(t + gridDim.x, S ^ 1);             // next tile's rows fly under this tile's MFMAs
        f32x4 acc[2][3];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int blk = 0; blk < 3; ++blk) acc[g][blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (X3) {
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // (dy plane, weight plane), smallest product first
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const f32x4 v0 = areg[S][g][2 * st], v1 = areg[S][g][2 * st + 1];
                    bf16x8_d apl[3];
                    split8((f32x8_d){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]}, apl);
#pragma unroll
                    for (int blk = 0; blk < 3; ++blk)
#pragma unroll
                        for (int q = 0; q < 6; ++q)
                            acc[g][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(apl[PA[q]], breg3[PB[q]][blk][st], acc[g][blk], 0, 0, 0);
                }
        } else if constexpr (MF16) {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int blk = 0; blk < 3; ++blk)
                        acc[g][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_d, areg[S][g][st]), breg16[blk][st],
                                                                              acc[g][blk], 0, 0, 0);
        } else
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int blk = 0; blk < 3; ++blk)
                {
                    float av;
                    if constexpr (IN16) {     // value s of the lane's 16 = bf16 half (s & 1) of dword s / 2
                        const float fv = areg[S][g][s >> 3][(s >> 1) & 3];     // (element first: a bit_cast applied directly to the vector-element expression read element 0)
                        const unsigned wv = __float_as_uint(fv);
                        av = __builtin_bit_cast(float, (s & 1) ? (wv & 0xffff0000u) : (wv << 16));
                    } else {
                        av = areg[S][g][s >> 2][s & 3];
                    }
                    acc[g][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, breg[blk][s], acc[g][blk], 0, 0, 0);
                }
        __syncthreads();                          // the previous tile's readers of Ps are done
        // C/D layout 16x16: column = lane & 15, pixel = 4*(lane >> 4) + reg
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int blk = 0; blk < 3; ++blk)
#pragma unroll
                for (int j = 0; j < 4; ++j) Ps[((wave + 4 * g) * CS_PC + 4 * kq + j) * CS_LDP + blk * 16 + p] = acc[g][blk][j];
        __syncthreads();
        // overlap-add: output row y = 2a + ph takes (a, r = 1 + ph) and (a - 1 + 2 ph, r = 3 - 3 ph); columns likewise.
        // Tile-local pixel (la, lb) = (a - a0 + 1, b - b0 + 1); out-of-image pixels were loaded as zeros.
        for (int e = tid; e < 3 * 2 * CS_TR * 2 * CS_TC; e += 256) {
            const int xx = e % (2 * CS_TC), yy = (e / (2 * CS_TC)) % (2 * CS_TR), c = e / (4 * CS_TR * CS_TC);
            const int y = 2 * a0 + yy, x = 2 * b0 + xx;
            const int ph = yy & 1, pw = xx & 1, la = (yy >> 1) + 1, lb = (xx >> 1) + 1;
            const int ra2 = la - 1 + 2 * ph, rr1 = 1 + ph, rr2 = 3 - 3 * ph;
            const int cb2 = lb - 1 + 2 * pw, ss1 = 1 + pw, ss2 = 3 - 3 * pw;
            const float* pc = Ps + c * 16;
            float v = pc[(la * CS_PC + lb) * CS_LDP + rr1 * 4 + ss1] + pc[(la * CS_PC + cb2) * CS_LDP + rr1 * 4 + ss2] +
                      pc[(ra2 * CS_PC + lb) * CS_LDP + rr2 * 4 + ss1] + pc[(ra2 * CS_PC + cb2) * CS_LDP + rr2 * 4 + ss2];
            if (act == DG_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
            if (y < H && x < W) dx[((long)(n * 3 + c) * H + y) * W + x] = v;
        }
        apply_act(S ^ 1);                         // FACT: the prefetched set, its `out` values have landed with it
    };
    // (Round 4: dealing each XCD a run of CONSECUTIVE tiles, so that the halo rows / columns neighbouring tiles share -- 885 MB per launch
    // by the counters against 637 MB of operands + output -- meet in one L2, changed nothing: 0.252 -> 0.258 ms f32x3, 0.284 -> 0.265 fp32,
    // 0.120 -> 0.126 bf16 at 512 px / batch 32; the plain order stays.)
    for (int t = blockIdx.x; t < ntiles; t += 2 * gridDim.x) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + (int)gridDim.x < ntiles) tile(std::integral_constant<int, 1>{}, t + gridDim.x);
    }
}

extern "C" size_t dg_c3_dgrad_workspace_bytes(int K) { return K == CD_K ? (size_t)CD_NK * 16 * sizeof(float) : 0; }
// groups > 1 (dg_conv4x4s2_c3_dgrad_g): the scatter form only (K == 64, fp32 dy)
static int c3_dgrad_run(int groups, const float* const* dy_nhwc, int dy_bf16, const float* const* w, float* const* dx_nchw, int N, int H, int W, int K,
                        int act, void* ws, size_t ws_bytes, dg_stream_t stream, const void* act_out = nullptr, float slope = 0.f);
extern "C" int dg_conv4x4s2_c3_dgrad(const float* dy_nhwc, const float* w, float* dx_nchw, int N, int H, int W, int K,
                                     int act, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return c3_dgrad_run(1, &dy_nhwc, 0, &w, &dx_nchw, N, H, W, K, act, ws, ws_bytes, stream);
}
extern "C" int dg_conv4x4s2_c3_dgrad_t(const void* dy_nhwc, int dy_bf16, const float* w, float* dx_nchw, int N, int H, int W, int K,
                                       int act, void* ws, size_t ws_bytes, dg_stream_t stream) {
    const float* dyp = (const float*)dy_nhwc;
    return c3_dgrad_run(1, &dyp, dy_bf16, &w, &dx_nchw, N, H, W, K, act, ws, ws_bytes, stream);
}
extern "C" int dg_conv4x4s2_c3_dgrad_p(const void* dy_nhwc, int dy_bf16, const float* w, float* dx_nchw, int N, int H, int W, int K,
                                       int act, int prec, void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(prec >= DG_PREC_DEFAULT && prec <= DG_PREC_F32X3, "dg_conv4x4s2_c3_dgrad_p: prec=%d", prec);
    DgPrecScope scope(prec);
    const float* dyp = (const float*)dy_nhwc;
    return c3_dgrad_run(1, &dyp, dy_bf16, &w, &dx_nchw, N, H, W, K, act, ws, ws_bytes, stream);
}
// 1 when the fused form below exists for this K under the options in force (the scatter kernel: K == 64, option "kt" != 16)
extern "C" int dg_c3_dgrad_act_ok(int K) { return K == CD_K && dg_get_option(DG_OPT_KT) != 16 ? 1 : 0; }
// conv1's input-gradient with the backward of the layer's fused LeakyReLU applied to dy in the load path: dx = dgrad(dy * (act_out > 0 ? 1 : slope))
extern "C" int dg_conv4x4s2_c3_dgrad_act_p(const void* dy_nhwc, int dy_bf16, const void* act_out, int in_act, float slope, const float* w, float* dx_nchw,
                                           int N, int H, int W, int K, int act, int prec, void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(prec >= DG_PREC_DEFAULT && prec <= DG_PREC_F32X3, "dg_conv4x4s2_c3_dgrad_act_p: prec=%d", prec);
    DG_CHECK_ARG(act_out && in_act == DG_ACT_LEAKY, "dg_conv4x4s2_c3_dgrad_act_p: needs the saved output of a fused LeakyReLU (in_act=%d)", in_act);
    DG_CHECK_ARG(dg_c3_dgrad_act_ok(K), "dg_conv4x4s2_c3_dgrad_act_p: the fused form is the scatter kernel (K == 64, option kt != 16)");
    DgPrecScope scope(prec);
    const float* dyp = (const float*)dy_nhwc;
    return c3_dgrad_run(1, &dyp, dy_bf16, &w, &dx_nchw, N, H, W, K, act, ws, ws_bytes, stream, act_out, slope);
}
extern "C" int dg_conv4x4s2_c3_dgrad_g(int groups, const float* const* dy_nhwc, const float* const* w, float* const* dx_nchw, int N, int H, int W, int K,
                                       int act, int prec, dg_stream_t stream) {
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS && dy_nhwc && w && dx_nchw, "dg_conv4x4s2_c3_dgrad_g: bad group / null table");
    DG_CHECK_ARG(prec >= DG_PREC_DEFAULT && prec <= DG_PREC_F32X3, "dg_conv4x4s2_c3_dgrad_g: prec=%d", prec);
    DG_CHECK_ARG(K == CD_K && dg_get_option(DG_OPT_KT) != 16, "dg_conv4x4s2_c3_dgrad_g: the grouped form is the scatter kernel (K == 64)");
    DgPrecScope scope(prec);
    return c3_dgrad_run(groups, dy_nhwc, 0, w, dx_nchw, N, H, W, K, act, nullptr, 0, stream);
}
static int c3_dgrad_run(int groups, const float* const* dy_tab, int dy_bf16, const float* const* w_tab, float* const* dx_tab, int N, int H, int W, int K,
                        int act, void* ws, size_t ws_bytes, dg_stream_t stream, const void* act_out, float slope) {
    DG_CHECK_ARG(act_out == nullptr || groups == 1, "dg_conv4x4s2_c3_dgrad: the fused activation backward takes one problem");
    for (int i = 0; i < groups; ++i) DG_CHECK_ARG(dy_tab[i] && w_tab[i] && dx_tab[i], "dg_conv4x4s2_c3_dgrad: null pointer");
    const float* dy_nhwc = dy_tab[0];
    const float* w = w_tab[0];
    float* dx_nchw = dx_tab[0];
    DG_CHECK_ARG(!dy_bf16 || (K == CD_K && dg_get_option(DG_OPT_KT) != 16), "dg_conv4x4s2_c3_dgrad_t: a bf16 dy needs K == 64 (scatter form)");
    DG_CHECK_ARG(N >= 1 && K >= 4 && K % 4 == 0, "dg_conv4x4s2_c3_dgrad: bad N/K (%d,%d)", N, K);
    DG_CHECK_ARG(dg_is_pow2(H) && dg_is_pow2(W) && H >= 2 && W >= 2, "dg_conv4x4s2_c3_dgrad: H,W must be powers of two");
    DG_CHECK_ARG(act == DG_ACT_NONE || act == DG_ACT_SIGMOID, "dg_conv4x4s2_c3_dgrad: bad act %d", act);
    DG_CHECK_ARG((long)N * 3 * H * W < (1L << 31), "dg_conv4x4s2_c3_dgrad: tensor too large");
    hipStream_t st = (hipStream_t)stream;
    const int Ho = H / 2, Wo = W / 2;
    if (K == CD_K && (long)N * Ho * Wo * CD_K * 4 < (1L << 31)) {
        if (dg_get_option(DG_OPT_KT) != 16) {       // scatter form (dense GEMM + overlap-add); "kt" 16 keeps the gather form testable
            const int tiles_r = (Ho + CS_TR - 1) / CS_TR, tiles_c = (Wo + CS_TC - 1) / CS_TC;
            const long ntiles = (long)N * tiles_r * tiles_c;
            DG_CHECK_ARG(ntiles < (1L << 30), "dg_conv4x4s2_c3_dgrad: too many tiles");
            const dim3 grid((unsigned)(ntiles < 512 ? ntiles : 512), groups);
            const DgPtrs pd = dg_ptrs((const void* const*)dy_tab, groups), pw = dg_ptrs((const void* const*)w_tab, groups),
                         px = dg_ptrs((const void* const*)dx_tab, groups);
            const DgPtrs pa = dg_ptrs1(act_out ? act_out : (const void*)dy_tab[0]);
            const unsigned dyb = (unsigned)((long)N * Ho * Wo * CD_K * (dy_bf16 ? 2 : 4));
#define C3_SCATTER(...) hipLaunchKernelGGL((c3_dgrad_scatter_kernel<__VA_ARGS__>), grid, dim3(256), 0, st, pd, pw, px, N, H, W, act, tiles_r, tiles_c, (int)ntiles, dyb, pa, slope)
            if (dy_bf16 && dg_cur_prec() == 1) {      // bf16 matrix path: bf16 MFMA
                if (act_out) C3_SCATTER(true, true, false, true); else C3_SCATTER(true, true);
            } else if (dy_bf16) {
                if (act_out) C3_SCATTER(true, false, false, true); else C3_SCATTER(true, false);
            } else if (dg_cur_prec() == 2) {          // f32x3 path: fp32-accurate products on the bf16 MFMA
                if (act_out) C3_SCATTER(false, false, true, true); else C3_SCATTER(false, false, true);
            } else {
                if (act_out) C3_SCATTER(false, false, false, true); else C3_SCATTER(false, false);
            }
#undef C3_SCATTER
            DG_CHECK_LAUNCH("c3_dgrad_scatter");
            return DG_OK;
        }
        DG_CHECK_ARG(groups == 1 && !act_out, "dg_conv4x4s2_c3_dgrad: the gather form takes one problem and no fused activation backward");
        const int tiles_r = (Ho + CD_TR - 1) / CD_TR, tiles_c = (Wo + CD_TC - 1) / CD_TC;
        const long ntiles = (long)N * tiles_r * tiles_c;
        DG_CHECK_ARG(ntiles < (1L << 31), "dg_conv4x4s2_c3_dgrad: too many tiles");
        if (ws == nullptr || ws_bytes < dg_c3_dgrad_workspace_bytes(K))
            return dg_fail(DG_ERR_WORKSPACE, "dg_conv4x4s2_c3_dgrad: workspace %zu < %zu", ws_bytes, dg_c3_dgrad_workspace_bytes(K));
        hipLaunchKernelGGL(c3_wq_build_kernel, dim3((CD_NK * 16 + 255) / 256), dim3(256), 0, st, w, (float*)ws);
        DG_CHECK_LAUNCH("c3_wq_build");
        const int grid = (int)(ntiles < 512 ? ntiles : 512);
        hipLaunchKernelGGL(c3_dgrad_mfma_kernel, dim3(grid), dim3(256), 0, st, dy_nhwc, (const float*)ws, dx_nchw, N, H, W, act,
                           tiles_r, tiles_c, (int)ntiles);
        DG_CHECK_LAUNCH("c3_dgrad_mfma");
        return DG_OK;
    }
    DG_CHECK_ARG(groups == 1 && !act_out, "dg_conv4x4s2_c3_dgrad: the VALU form takes one problem and no fused activation backward");
    const long nquad = (long)N * Ho * Wo;
    hipLaunchKernelGGL(c3_dgrad_valu_kernel, dim3((unsigned)((nquad + 255) / 256)), dim3(256), 0, st, dy_nhwc, w,
                       dx_nchw, N, H, W, K, dg_ilog2(Ho), dg_ilog2(Wo), act);
    DG_CHECK_LAUNCH("c3_dgrad_valu");
    return DG_OK;
}

// ---- forward, K == 64: per-wave streaming on v_mfma_f32_32x32x2_f32 ---------------------------------------
// y[pix][k] = act(sum_{c,r,q} x[n,c,2oy-1+r,2ox-1+q] * w[k][c][r][q]).  A group is 32 consecutive output pixels
// (one output row at 64 px); a wave owns whole groups.  MFMA A operand (pixel = lane%32, k-half = lane/32) is
// gathered straight from the NCHW image: GEMM-k step s = (c, r, j) and lane half h read column q = 2j+h, so the
// 64 lanes of one load cover 64 consecutive input floats (fully coalesced, every byte used).  The B operand
// (the [48][64] weights) lives in 48 registers per lane for the whole kernel; the next group's 24 image values
// are prefetched under the current group's 48 MFMAs.  LDS only stages the weights once; no barriers after that.  Stores: 32 lanes write the 32
// consecutive channels of one pixel (128 B).
#define CF_K 64
#define CF_S 24            // GEMM-k steps of 2
// OUT16: y is bf16.  The two accumulator blocks then hold the EVEN and the ODD output channels (block nb, column p ->
// channel 2p + nb), so a lane owns two adjacent channels of a pixel and stores them as one dword: one 128-B store per
// pixel row instead of two, half the bytes.
template <int ACT, bool OUT16>
__global__ __launch_bounds__(256, 2) void c3_fwd_mfma_kernel(const DgPtrs xs, const DgPtrs ws_, const DgPtrs ys, int N, int H, int W, int lgHo, int lgWo,
                                                             long npix, int ngroups, float slope, int xbytes) {
    // grouped launch (dg_conv4x4s2_c3_fwd_g): blockIdx.y = problem
    const float* __restrict__ x = dg_pick<const float>(xs, blockIdx.y);
    const float* __restrict__ w = dg_pick<const float>(ws_, blockIdx.y);
    float* __restrict__ y = dg_pick<float>(ys, blockIdx.y);
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int Ho = H >> 1, Wo = W >> 1;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const float* const zp = dg_zero_edge;
    // weights: coalesced copy to LDS ([64][49], odd stride: conflict-free column reads), then 48 registers per lane
    __shared__ float wS[CF_K * 49];
    for (int e = threadIdx.x; e < CF_K * 48; e += 256) wS[(e / 48) * 49 + e % 48] = w[e];
    __syncthreads();
    float wb[2][CF_S];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int s = 0; s < CF_S; ++s) {
            const int c = s >> 3, r = (s >> 1) & 3, q = 2 * (s & 1) + h;
            wb[nb][s] = wS[(OUT16 ? 2 * p + nb : nb * 32 + p) * 49 + c * 16 + r * 4 + q];
        }
    float a[2][CF_S];
    // image gathers: raw buffer loads, 32-bit byte offsets; an invalid row / column / pixel pushes the offset
    // past the descriptor's range and the hardware returns 0 (no per-load select, 8 adds per 24 loads)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, xbytes, 0x00020000);
    const int HW4 = H * W * 4;
    constexpr int BIG = 0x40000000;
    auto gather = [&](int g, float* dst) {
        const int pix = g * 32 + p;
        const bool okp = g < ngroups && pix < (int)npix;
        const int ox = pix & (Wo - 1), oy = (pix >> lgWo) & (Ho - 1), n = pix >> (lgWo + lgHo);
        const int base = n * 3 * HW4;
        int colb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ix = 2 * ox - 1 + 2 * j + h;
            colb[j] = (unsigned)ix < (unsigned)W ? ix * 4 : BIG;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = 2 * oy - 1 + r;
            const int rowb = (okp && (unsigned)iy < (unsigned)H) ? base + iy * W * 4 : BIG;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int voff = rowb + colb[j];
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    dst[c * 8 + r * 2 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff, c * HW4, 0));
            }
        }
    };
    int g = wave;
    if (g < ngroups) gather(g, a[0]);
    auto body = [&](auto PB, int gcur) {
        constexpr int P = decltype(PB)::value;
        gather(gcur + nwaves, a[P ^ 1]);          // prefetch (masked to zero-block reads past the end)
        f32x16 acc0 = {0.f}, acc1 = {0.f};
#pragma unroll
        for (int s = 0; s < CF_S; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[P][s], wb[0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[P][s], wb[1][s], acc1, 0, 0, 0);
        }
        const long pix0 = (long)gcur * 32;
        if constexpr (OUT16) {
            typedef __bf16 bf16x2_e __attribute__((ext_vector_type(2)));
            typedef float f32x2_e __attribute__((ext_vector_type(2)));
            unsigned* o16 = (unsigned*)((__bf16*)y + (pix0 + h * 4) * CF_K + 2 * p);
            const bool whole = pix0 + 32 <= npix;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = (v >> 2) * 8 + (v & 3);
                const f32x2_e pr = {dg_apply_act(acc0[v], ACT, slope), dg_apply_act(acc1[v], ACT, slope)};
                if (whole || pix0 + h * 4 + i < npix)
                    o16[i * (CF_K / 2)] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2_e));
            }
            return;
        }
        float* o = y + (pix0 + h * 4) * CF_K + p;
        if (pix0 + 32 <= npix) {                  // whole group in range (wave-uniform): straight-line stores
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = (v >> 2) * 8 + (v & 3);
                o[i * CF_K] = dg_apply_act(acc0[v], ACT, slope);
                o[i * CF_K + 32] = dg_apply_act(acc1[v], ACT, slope);
            }
        } else {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = (v >> 2) * 8 + (v & 3);
                if (pix0 + h * 4 + i < npix) {
                    o[i * CF_K] = dg_apply_act(acc0[v], ACT, slope);
                    o[i * CF_K + 32] = dg_apply_act(acc1[v], ACT, slope);
                }
            }
        }
    };
    for (; g < ngroups; g += 2 * nwaves) {
        body(std::integral_constant<int, 0>{}, g);
        if (g + nwaves < ngroups) body(std::integral_constant<int, 1>{}, g + nwaves);
    }
}

// ---- forward, K == 64, on the bf16 matrix path (option "bf16" = 1): v_mfma_f32_32x32x16_bf16 -------------------------
// Same per-wave streaming structure, but the image and the weights are rounded to bf16 (RNE) on the way into the MFMA --
// like every other convolution of the bf16 path -- and the 48-deep reduction is THREE k16 steps (one input channel each)
// instead of 24 fp32 steps: 6 MFMAs of 32 cycles per 32 pixels instead of 48 of 64 (the fp32 kernel is matrix-bound at
// ~100 us for 32 x 512 x 512; this one is bound by its loads and stores).  Lane (pixel p, half h) owns k = 16 c + 8 h + j,
// j = 0..7 = filter rows 2h, 2h+1 x the 4 filter columns: two rows of 4 CONSECUTIVE input floats (an aligned 8-byte pair
// plus the two outer dwords); padding rows / columns and out-of-range pixels read out of range (zeros).
typedef __bf16 bf16x8_e __attribute__((ext_vector_type(8)));
typedef float f32x8_e __attribute__((ext_vector_type(8)));
typedef float f32x2_g __attribute__((ext_vector_type(2)));
// X3 (option "bf16" = 2, the f32x3 path): image and weights are NOT rounded -- each fp32 value is split into its three bf16 planes
// hi / mid / lo in registers (dg_split3: 24 significand bits) and every product block is the six MFMAs of igemm.hip's PREC 2
// (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi, smallest first): 36 MFMAs of 32 cycles per 32 pixels against the fp32 kernel's
// 48 of 64 -- fp32-accurate products at 2.7x the matrix rate, so the kernel is bound by its loads and stores like the bf16 one.
// PL (with X3): the epilogue also writes the three bf16 PLANES of y (pixel-major [pixel][64], `pstride` elements apart) -- the next
// layer's weight-gradient reads them, and a stand-alone split pass over the 537 MB tensor (read 4 + write 6 bytes per element) is
// replaced by 6 more bytes per element here.  The two accumulator blocks then hold the even / odd channels (like OUT16), so a lane
// stores an 8-byte fp32 pair and one packed bf16 pair per plane.
template <int ACT, bool OUT16, bool X3 = false, bool PL = false>
__global__ __launch_bounds__(256, 2) void c3_fwd_bf16mfma_kernel(const DgPtrs xs, const DgPtrs ws_, const DgPtrs ys, int N, int H, int W, int lgHo, int lgWo,
                                                                 long npix, int ngroups, float slope, int xbytes,
                                                                 __bf16* __restrict__ planes = nullptr, long pstride = 0) {
    // grouped launch (dg_conv4x4s2_c3_fwd_g): blockIdx.y = problem (the plane output exists in the one-problem form only)
    const float* __restrict__ x = dg_pick<const float>(xs, blockIdx.y);
    const float* __restrict__ w = dg_pick<const float>(ws_, blockIdx.y);
    float* __restrict__ y = dg_pick<float>(ys, blockIdx.y);
    static_assert(!PL || (X3 && !OUT16), "planes are written by the f32x3 form with an fp32 output");
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int Ho = H >> 1, Wo = W >> 1;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    // B operand: block nb, column p = channel (OUT16 ? 2p + nb : 32 nb + p); k = 16 c + 8 h + j are 8 consecutive weights
    constexpr int NPL = X3 ? 3 : 1;
    bf16x8_e wb[NPL][2][3];
    auto split8 = [](const f32x8_e& v, bf16x8_e* pl) {     // X3: pl[0..2] = hi, mid, lo (both residuals are exact in fp32); else pl[0] = RNE(v)
        pl[0] = __builtin_convertvector(v, bf16x8_e);
        if constexpr (X3) {
            const f32x8_e r1 = v - __builtin_convertvector(pl[0], f32x8_e);
            pl[1] = __builtin_convertvector(r1, bf16x8_e);
            pl[2] = __builtin_convertvector(r1 - __builtin_convertvector(pl[1], f32x8_e), bf16x8_e);
        }
    };
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* wp = w + ((OUT16 || PL) ? 2 * p + nb : nb * 32 + p) * 48 + c * 16 + 8 * h;
            const f32x4 lo = *(const f32x4*)wp, hi = *(const f32x4*)(wp + 4);
            bf16x8_e pl[NPL];
            split8((f32x8_e){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, pl);
#pragma unroll
            for (int q = 0; q < NPL; ++q) wb[q][nb][c] = pl[q];
        }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, xbytes, 0x00020000);
    const int HW4 = H * W * 4;
    constexpr int BIG = 0x40000000;
    // [set][channel][filter row 2h + rr]: the 4 consecutive input floats ix = 2ox-1 .. 2ox+2 as {left, pair, right}: the pair
    // (2ox, 2ox+1) is an aligned 8-byte load, the two outer columns are single dwords whose offset goes out of range when the
    // column is padding (zeros, no masking in registers)
    float al[2][3][2], ar[2][3][2];
    f32x2_g ap[2][3][2];
    auto gather = [&](int g, int set) {
        const int pix = g * 32 + p;
        const bool okp = g < ngroups && pix < (int)npix;
        const int ox = pix & (Wo - 1), oy = (pix >> lgWo) & (Ho - 1), n = pix >> (lgWo + lgHo);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int iy = 2 * oy - 1 + 2 * h + rr;
            const bool okr = okp && (unsigned)iy < (unsigned)H;
            const int pb = okr ? n * 3 * HW4 + iy * W * 4 + ox * 8 : BIG;
            const int lb = (okr && ox != 0) ? pb - 4 : BIG;
            const int rb = (okr && ox != Wo - 1) ? pb + 8 : BIG;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                al[set][c][rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, lb, c * HW4, 0));
                ap[set][c][rr] = __builtin_bit_cast(f32x2_g, __builtin_amdgcn_raw_buffer_load_b64(rx, pb, c * HW4, 0));
                ar[set][c][rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, rb, c * HW4, 0));
            }
        }
    };
    int g = wave;
    if (g < ngroups) gather(g, 0);
    auto body = [&](auto PB, int gcur) {
        constexpr int P = decltype(PB)::value;
        gather(gcur + nwaves, P ^ 1);          // prefetch (out of range past the end)
        f32x16 acc0 = {0.f}, acc1 = {0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            bf16x8_e av[NPL];
            split8((f32x8_e){al[P][c][0], ap[P][c][0][0], ap[P][c][0][1], ar[P][c][0],
                             al[P][c][1], ap[P][c][1][0], ap[P][c][1][1], ar[P][c][1]}, av);
            if constexpr (X3) {
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // (a plane, b plane), smallest product first
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[q]], wb[PB[q]][0][c], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[q]], wb[PB[q]][1][c], acc1, 0, 0, 0);
                }
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], wb[0][0][c], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], wb[0][1][c], acc1, 0, 0, 0);
            }
        }
        const long pix0 = (long)gcur * 32;
        const bool whole = pix0 + 32 <= npix;
        if constexpr (OUT16) {
            typedef __bf16 bf16x2_e __attribute__((ext_vector_type(2)));
            typedef float f32x2_e __attribute__((ext_vector_type(2)));
            unsigned* o16 = (unsigned*)((__bf16*)y + (pix0 + h * 4) * CF_K + 2 * p);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = (v >> 2) * 8 + (v & 3);
                const f32x2_e pr = {dg_apply_act(acc0[v], ACT, slope), dg_apply_act(acc1[v], ACT, slope)};
                if (whole || pix0 + h * 4 + i < npix)
                    o16[i * (CF_K / 2)] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2_e));
            }
        } else if constexpr (PL) {
            typedef __bf16 bf16x2_p __attribute__((ext_vector_type(2)));
            typedef float f32x2_p __attribute__((ext_vector_type(2)));
            float* o = y + (pix0 + h * 4) * CF_K + 2 * p;
            __bf16* pb = planes + (pix0 + h * 4) * CF_K + 2 * p;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = (v >> 2) * 8 + (v & 3);
                if (whole || pix0 + h * 4 + i < npix) {
                    const f32x2_p pr = {dg_apply_act(acc0[v], ACT, slope), dg_apply_act(acc1[v], ACT, slope)};
                    *(f32x2_p*)(o + i * CF_K) = pr;
                    const bf16x2_p hi2 = __builtin_convertvector(pr, bf16x2_p);
                    const f32x2_p r1 = pr - __builtin_convertvector(hi2, f32x2_p);
                    const bf16x2_p mid2 = __builtin_convertvector(r1, bf16x2_p);
                    const bf16x2_p lo2 = __builtin_convertvector(r1 - __builtin_convertvector(mid2, f32x2_p), bf16x2_p);
                    *(bf16x2_p*)(pb + i * CF_K) = hi2;
                    *(bf16x2_p*)(pb + pstride + i * CF_K) = mid2;
                    *(bf16x2_p*)(pb + 2 * pstride + i * CF_K) = lo2;
                }
            }
        } else {
            float* o = y + (pix0 + h * 4) * CF_K + p;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = (v >> 2) * 8 + (v & 3);
                if (whole || pix0 + h * 4 + i < npix) {
                    o[i * CF_K] = dg_apply_act(acc0[v], ACT, slope);
                    o[i * CF_K + 32] = dg_apply_act(acc1[v], ACT, slope);
                }
            }
        }
    };
    for (; g < ngroups; g += 2 * nwaves) {
        body(std::integral_constant<int, 0>{}, g);
        if (g + nwaves < ngroups) body(std::integral_constant<int, 1>{}, g + nwaves);
    }
}

// f32x3 path: conv1 forward (+ fused activation) that ALSO writes the plane triple of its output (see PL above).
extern "C" int dg_conv4x4s2_c3_fwd_x3(const float* x_nchw, const float* w, float* y_nhwc, void* y_planes, size_t plane_elems, int N, int H,
                                      int W, int K, int act, float slope, dg_stream_t stream) {
    DG_CHECK_ARG(x_nchw && w && y_nhwc && y_planes, "dg_conv4x4s2_c3_fwd_x3: null pointer");
    DG_CHECK_ARG(K == CF_K, "dg_conv4x4s2_c3_fwd_x3: K must be %d", CF_K);
    DG_CHECK_ARG(N >= 1 && dg_is_pow2(H) && dg_is_pow2(W) && H >= 2 && W >= 2, "dg_conv4x4s2_c3_fwd_x3: H, W must be powers of two");
    DG_CHECK_ARG(act == DG_ACT_NONE || act == DG_ACT_LEAKY || act == DG_ACT_RELU, "dg_conv4x4s2_c3_fwd_x3: bad act %d", act);
    const int Ho = H / 2, Wo = W / 2;
    const long npix = (long)N * Ho * Wo;
    DG_CHECK_ARG((long)N * 3 * H * W * 4 < (1L << 30) && npix < (1L << 30), "dg_conv4x4s2_c3_fwd_x3: tensors must be below 1 GiB");
    DG_CHECK_ARG(plane_elems >= (size_t)npix * CF_K && plane_elems % 8 == 0, "dg_conv4x4s2_c3_fwd_x3: plane distance %zu for %ld elements", plane_elems, npix * CF_K);
    const long ngroups = (npix + 31) / 32;
    long wgs = (ngroups + 31) / 32;
    if (wgs > 4096) wgs = 4096;
    if (wgs < 1) wgs = 1;
    hipStream_t st = (hipStream_t)stream;
#define CFPL_LAUNCH(ACT)                                                                                                          \
    hipLaunchKernelGGL((c3_fwd_bf16mfma_kernel<ACT, false, true, true>), dim3((unsigned)wgs), dim3(256), 0, st, dg_ptrs1(x_nchw), dg_ptrs1(w), dg_ptrs1(y_nhwc), N, H, W, \
                       dg_ilog2(Ho), dg_ilog2(Wo), npix, (int)ngroups, slope, (int)((long)N * 3 * H * W * 4), (__bf16*)y_planes, (long)plane_elems)
    if (act == DG_ACT_LEAKY) { CFPL_LAUNCH(DG_ACT_LEAKY); }
    else if (act == DG_ACT_RELU) { CFPL_LAUNCH(DG_ACT_RELU); }
    else { CFPL_LAUNCH(DG_ACT_NONE); }
#undef CFPL_LAUNCH
    DG_CHECK_LAUNCH("dg_conv4x4s2_c3_fwd_x3");
    return DG_OK;
}

// the K == 64 streaming kernels, `groups` problems per launch (blockIdx.y); arithmetic = dg_cur_prec()
extern "C" int dg_c3_fwd_mfma_launch_g(int groups, const float* const* x_tab, const float* const* w_tab, void* const* y_tab, int y_bf16, int N, int H, int W,
                                       int act, float slope, hipStream_t st) {
    const DgPtrs x_nchw = dg_ptrs((const void* const*)x_tab, groups), w = dg_ptrs((const void* const*)w_tab, groups),
                 y_nhwc = dg_ptrs((const void* const*)y_tab, groups);
    const int Ho = H / 2, Wo = W / 2;
    const long npix = (long)N * Ho * Wo;
    const long ngroups = (npix + 31) / 32;
    if (ngroups >= (1L << 30)) return dg_fail(DG_ERR_INVALID, "dg_conv4x4s2_c3_fwd: too many pixels");
    // 8 groups per wave and one workgroup per CU (96 KB of LDS reserved so the dispatcher spreads the grid):
    // measured 26 us vs 34 us at 4 groups / 2 workgroups per CU (256 x 3 x 64 x 64 -> 64 ch); the fixed
    // per-workgroup prologue (weights, first gather) is what the longer waves amortise.  Small launches (64 px / batch 64: 2048 groups
    // per problem = 64 such workgroups) leave most CUs idle: fewer groups per wave until the launch has a workgroup per CU
    long per = 32;
    while (per > 4 && ((ngroups + per - 1) / per) * groups < 256) per >>= 1;
    long wgs = (ngroups + per - 1) / per;
    if (wgs > 4096) wgs = 4096;
    if (wgs < 1) wgs = 1;
#define CF_LAUNCH(ACT, O16)                                                                                          \
    { static const hipError_t once = hipFuncSetAttribute((const void*)c3_fwd_mfma_kernel<ACT, O16>,                    \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);      \
      (void)once; }                                                                                                     \
    hipLaunchKernelGGL((c3_fwd_mfma_kernel<ACT, O16>), dim3((unsigned)wgs, groups), dim3(256), 96 * 1024, st, x_nchw, w, y_nhwc, N, H, W, \
                       dg_ilog2(Ho), dg_ilog2(Wo), npix, (int)ngroups, slope, (int)((long)N * 3 * H * W * 4))
    if (dg_cur_prec() == 2 && dg_get_option(DG_OPT_KT) != 16 && !y_bf16) {   // f32x3 path: fp32-accurate products on the bf16 MFMA
#define CFX3_LAUNCH(ACT)                                                                                                    \
        hipLaunchKernelGGL((c3_fwd_bf16mfma_kernel<ACT, false, true>), dim3((unsigned)wgs, groups), dim3(256), 0, st, x_nchw, w, y_nhwc, N, H, W, \
                           dg_ilog2(Ho), dg_ilog2(Wo), npix, (int)ngroups, slope, (int)((long)N * 3 * H * W * 4))
        if (act == DG_ACT_LEAKY) { CFX3_LAUNCH(DG_ACT_LEAKY); }
        else if (act == DG_ACT_RELU) { CFX3_LAUNCH(DG_ACT_RELU); }
        else { CFX3_LAUNCH(DG_ACT_NONE); }
#undef CFX3_LAUNCH
        return DG_OK;
    }
    if (dg_cur_prec() == 1 && dg_get_option(DG_OPT_KT) != 16) {     // bf16 matrix path ("kt" 16 keeps the fp32-MFMA kernel testable there)
#define CF16_LAUNCH(ACT, O16)                                                                                              \
        hipLaunchKernelGGL((c3_fwd_bf16mfma_kernel<ACT, O16>), dim3((unsigned)wgs, groups), dim3(256), 0, st, x_nchw, w, y_nhwc, N, H, W,    \
                           dg_ilog2(Ho), dg_ilog2(Wo), npix, (int)ngroups, slope, (int)((long)N * 3 * H * W * 4))
        if (y_bf16) {
            if (act == DG_ACT_LEAKY) { CF16_LAUNCH(DG_ACT_LEAKY, true); }
            else if (act == DG_ACT_RELU) { CF16_LAUNCH(DG_ACT_RELU, true); }
            else { CF16_LAUNCH(DG_ACT_NONE, true); }
        } else {
            if (act == DG_ACT_LEAKY) { CF16_LAUNCH(DG_ACT_LEAKY, false); }
            else if (act == DG_ACT_RELU) { CF16_LAUNCH(DG_ACT_RELU, false); }
            else { CF16_LAUNCH(DG_ACT_NONE, false); }
        }
#undef CF16_LAUNCH
        return DG_OK;
    }
    if (y_bf16) {
        if (act == DG_ACT_LEAKY) { CF_LAUNCH(DG_ACT_LEAKY, true); }
        else if (act == DG_ACT_RELU) { CF_LAUNCH(DG_ACT_RELU, true); }
        else { CF_LAUNCH(DG_ACT_NONE, true); }
    } else {
        if (act == DG_ACT_LEAKY) { CF_LAUNCH(DG_ACT_LEAKY, false); }
        else if (act == DG_ACT_RELU) { CF_LAUNCH(DG_ACT_RELU, false); }
        else { CF_LAUNCH(DG_ACT_NONE, false); }
    }
#undef CF_LAUNCH
    return DG_OK;
}
extern "C" int dg_c3_fwd_mfma_launch(const float* x_nchw, const float* w, void* y_nhwc_v, int y_bf16, int N, int H, int W, int act,
                                     float slope, hipStream_t st) {
    return dg_c3_fwd_mfma_launch_g(1, &x_nchw, &w, &y_nhwc_v, y_bf16, N, H, W, act, slope, st);
}

// ---- weight gradient -------------------------------------------------------------------------------------
// dw[k][c][r][s] (+)= sum_{pixels} dy[pix][k] * x[n,c,2oy-1+r,2ox-1+s]
// Each wave streams its own pixels: the MFMA A operand dy^T[k][pixel] and the B operand
// im2col(x)[pixel][j] are loaded straight into registers (A: 32 lanes read 32 consecutive k of one pixel
// = 128 B; B: each lane owns a fixed (c,r,s) and gathers from the NCHW image), 8 k-steps (16 pixels) per
// batch, two batches in flight.  No LDS, no barriers; masked lanes read a zero block.
#define CW_B 8       // k-steps (of 2 pixels) per batch
// BUF: both tensors < 1 GiB -> raw buffer loads with 32-bit offsets, masked lanes read out of range (= 0).
// FACT: the dy operand is taken through the backward of the layer's fused LeakyReLU/ReLU on the fly
//       (dy * act'(act_out), act_out = the saved forward output), instead of a separate act_bwd pass.
// IN16 (BUF only): dy and act_out are bf16.  A lane then loads ONE dword = channels (2 l31, 2 l31 + 1) of its pixel, so MFMA
//       block i holds the channels of parity i: block i, row rho <-> channel 2 rho + i.
template <bool BUF, bool FACT, bool IN16 = false>
__global__ __launch_bounds__(256, 2) void c3_wgrad_mfma_kernel(const DgPtrs dys, const DgPtrs xs, const DgPtrs parts, int N, int H, int W, int K,
                                                               int lgHo, int lgWo, long npix, int pix_per_wave,
                                                               const DgPtrs act_outs, float slope) {
    // grouped launch (dg_conv4x4s2_c3_wgrad_g): blockIdx.z = problem
    const float* __restrict__ dy = dg_pick<const float>(dys, blockIdx.z);
    const float* __restrict__ x = dg_pick<const float>(xs, blockIdx.z);
    float* __restrict__ part = dg_pick<float>(parts, blockIdx.z);
    const float* __restrict__ act_out = dg_pick<const float>(act_outs, blockIdx.z);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int Ho = H >> 1, Wo = W >> 1;
    const int kg = blockIdx.y;
    const float* const zp = dg_zero_edge;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // this lane's two im2col columns j = l31, 32 + l31 (valid < 48): fixed (c, r, s)
    int jr[2], js[2];
    long joff[2];
    bool jok[2];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const int j = jn * 32 + l31;
        jok[jn] = j < 48;
        const int c = (j >> 4) % 3;
        jr[jn] = (j >> 2) & 3;
        js[jn] = j & 3;
        joff[jn] = ((long)c * H + jr[jn]) * W + js[jn];
    }
    const long wave_id = (long)blockIdx.x * 4 + wave;
    const long p_begin = wave_id * pix_per_wave;
    const long p_end = min(npix, p_begin + pix_per_wave);
    const float* dyk = dy + kg * 64 + l31;

    float fa[2][CW_B][2], fb[2][CW_B][2];
    constexpr int BIG = 0x40000000;
    static_assert(!IN16 || BUF, "bf16 operands: buffer path only");
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, BUF ? (int)(npix * K * (IN16 ? 2 : 4)) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rao = __builtin_amdgcn_make_buffer_rsrc((void*)(FACT ? act_out : dy), 0, BUF ? (int)(npix * K * (IN16 ? 2 : 4)) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, BUF ? N * 3 * H * W * 4 : 0, 0x00020000);
    auto bload = [](const __amdgpu_buffer_rsrc_t& r, int off) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    };
    auto load_batch = [&](int set, long p0) {
#pragma unroll
        for (int st = 0; st < CW_B; ++st) {
            if constexpr (BUF) {
                const int pp = (int)p0 + 2 * st + lh;
                const bool pok = pp < (int)p_end;
                const int ox = pp & (Wo - 1), oy = (pp >> lgWo) & (Ho - 1), n = pp >> (lgWo + lgHo);
                const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
                const int xb = (n * 3 * H + iy0) * W + ix0;
                float a0, a1;
                if constexpr (IN16) {
                    const int aoff = pok ? (pp * K + kg * 64 + 2 * l31) * 2 : BIG;
                    const unsigned wv = __builtin_bit_cast(unsigned, bload(rdy, aoff));
                    a0 = __builtin_bit_cast(float, wv << 16);
                    a1 = __builtin_bit_cast(float, wv & 0xffff0000u);
                    if constexpr (FACT) {
                        const unsigned ov = __builtin_bit_cast(unsigned, bload(rao, aoff));
                        a0 = __builtin_bit_cast(float, ov << 16) > 0.f ? a0 : a0 * slope;
                        a1 = __builtin_bit_cast(float, ov & 0xffff0000u) > 0.f ? a1 : a1 * slope;
                    }
                } else {
                const int aoff = pok ? (pp * K + kg * 64 + l31) * 4 : BIG;
                a0 = bload(rdy, aoff);
                a1 = bload(rdy, aoff + 128);
                if constexpr (FACT) {
                    const float o0 = bload(rao, aoff), o1 = bload(rao, aoff + 128);
                    a0 = o0 > 0.f ? a0 : a0 * slope;
                    a1 = o1 > 0.f ? a1 : a1 * slope;
                }
                }
                fa[set][st][0] = a0;
                fa[set][st][1] = a1;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    const bool ok = pok && jok[jn] && (unsigned)(iy0 + jr[jn]) < (unsigned)H && (unsigned)(ix0 + js[jn]) < (unsigned)W;
                    fb[set][st][jn] = bload(rx, ok ? (xb + (int)joff[jn]) * 4 : BIG);
                }
            } else {
                const long pp = p0 + 2 * st + lh;
                const bool pok = pp < p_end;
                const int ox = (int)(pp & (Wo - 1)), oy = (int)((pp >> lgWo) & (Ho - 1)), n = (int)(pp >> (lgWo + lgHo));
                const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
                const long xb = ((long)n * 3 * H + iy0) * W + ix0;
                const float* ap = pok ? dyk + pp * K : zp;
                float a0 = ap[0], a1 = ap[32];
                if constexpr (FACT) {
                    const float* op = pok ? act_out + kg * 64 + l31 + pp * K : zp;
                    const float o0 = op[0], o1 = op[32];
                    a0 = o0 > 0.f ? a0 : a0 * slope;
                    a1 = o1 > 0.f ? a1 : a1 * slope;
                }
                fa[set][st][0] = a0;
                fa[set][st][1] = a1;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    const bool ok = pok && jok[jn] && (unsigned)(iy0 + jr[jn]) < (unsigned)H && (unsigned)(ix0 + js[jn]) < (unsigned)W;
                    const float* bp = ok ? x + xb + joff[jn] : zp;
                    fb[set][st][jn] = *bp;
                }
            }
        }
    };
    auto mma_batch = [&](int set) {
#pragma unroll
        for (int st = 0; st < CW_B; ++st)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][st][i], fb[set][st][jn], acc[i][jn], 0, 0, 0);
    };
    load_batch(0, p_begin);
    for (long p0 = p_begin; p0 < p_end; p0 += 4 * CW_B) {
        load_batch(1, p0 + 2 * CW_B);
        mma_batch(0);
        load_batch(0, p0 + 4 * CW_B);
        mma_batch(1);
    }
    // the 4 waves' accumulators are summed through LDS in a fixed order -> one slab per workgroup:
    // part[block][K][48]
    __shared__ float slabS[4][64 * 48];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int k = IN16 ? 2 * rho + i : i * 32 + rho;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int j = jn * 32 + l31;
                if (j < 48) slabS[wave][k * 48 + j] = acc[i][jn][r];
            }
        }
    __syncthreads();
    float* slab = part + ((long)blockIdx.x * K + kg * 64) * 48;
    for (int e = tid; e < 64 * 48; e += 256) slab[e] = ((slabS[0][e] + slabS[1][e]) + slabS[2][e]) + slabS[3][e];
}

// ---- weight gradient on the bf16 matrix path (option "bf16" = 1, bf16 dy): v_mfma_f32_32x32x16_bf16 ---------------------
// Same per-wave streaming (no LDS, no barriers), but one MFMA k-step is SIXTEEN pixels: lane (row l31, half lh) supplies 8
// consecutive pixels 8 lh .. 8 lh + 7 of the batch.  A (dy^T): one dword per pixel = channels (2 l31, 2 l31 + 1), the two
// parities are split into the two row blocks with v_perm_b32 (block i, row rho <-> channel 2 rho + i, as in the IN16 kernel
// above); B (im2col of the fp32 image): 8 strided floats per column, rounded to bf16.  4 MFMAs of 32 cycles per 16 pixels
// instead of 32 of 64 -- the kernel is bound by its gathers, not by the matrix pipe.
typedef __bf16 bf16x8_w __attribute__((ext_vector_type(8)));
typedef float f32x8_w __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_w __attribute__((ext_vector_type(4)));
template <bool FACT>
__global__ __launch_bounds__(256, 2) void c3_wgrad_bf16mfma_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   float* __restrict__ part, int N, int H, int W, int K,
                                                                   int lgHo, int lgWo, long npix, int pix_per_wave,
                                                                   const float* __restrict__ act_out, float slope) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int Ho = H >> 1, Wo = W >> 1;
    const int kg = blockIdx.y;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int jr[2], js[2], joff[2];
    bool jok[2];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const int j = jn * 32 + l31;
        jok[jn] = j < 48;
        const int c = (j >> 4) % 3;
        jr[jn] = (j >> 2) & 3;
        js[jn] = j & 3;
        joff[jn] = (c * H + jr[jn]) * W + js[jn];
    }
    const long wave_id = (long)blockIdx.x * 4 + wave;
    const long p_begin = wave_id * pix_per_wave;
    const long p_end = min(npix, p_begin + pix_per_wave);
    constexpr int BIG = 0x40000000;
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)(npix * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rao = __builtin_amdgcn_make_buffer_rsrc((void*)(FACT ? act_out : dy), 0, (int)(npix * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, N * 3 * H * W * 4, 0x00020000);
    auto bload = [](const __amdgpu_buffer_rsrc_t& r, int off) -> unsigned {
        return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    };
    unsigned wa[2][8], wo[2][FACT ? 8 : 1];
    float fb[2][2][8];
    auto load_batch = [&](int set, long p0) {       // 16 pixels p0 .. p0 + 15; this lane's are p0 + 8 lh + t
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int pp = (int)p0 + 8 * lh + t;
            const bool pok = pp < (int)p_end;
            const int ox = pp & (Wo - 1), oy = (pp >> lgWo) & (Ho - 1), n = pp >> (lgWo + lgHo);
            const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
            const int xb = (n * 3 * H + iy0) * W + ix0;
            const int aoff = pok ? (pp * K + kg * 64 + 2 * l31) * 2 : BIG;
            wa[set][t] = bload(rdy, aoff);
            if constexpr (FACT) wo[set][t] = bload(rao, aoff);
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const bool ok = pok && jok[jn] && (unsigned)(iy0 + jr[jn]) < (unsigned)H && (unsigned)(ix0 + js[jn]) < (unsigned)W;
                fb[set][jn][t] = __builtin_bit_cast(float, bload(rx, ok ? (xb + joff[jn]) * 4 : BIG));
            }
        }
    };
    auto mma_batch = [&](int set) {
        unsigned w8[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            unsigned wv = wa[set][t];
            if constexpr (FACT) {     // dy * act'(act_out) per channel of the pair; the product is rounded to bf16 again (RNE)
                const unsigned ov = wo[set][t];
                float a0 = __builtin_bit_cast(float, wv << 16), a1 = __builtin_bit_cast(float, wv & 0xffff0000u);
                a0 = __builtin_bit_cast(float, ov << 16) > 0.f ? a0 : a0 * slope;
                a1 = __builtin_bit_cast(float, ov & 0xffff0000u) > 0.f ? a1 : a1 * slope;
                typedef __bf16 bf16x2_w __attribute__((ext_vector_type(2)));
                typedef float f32x2_w __attribute__((ext_vector_type(2)));
                wv = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_w){a0, a1}, bf16x2_w));
            }
            w8[t] = wv;
        }
        bf16x8_w fa[2], fbv[2];
        u32x4_w ev, od;
#pragma unroll
        for (int u = 0; u < 4; ++u) {     // pixels 2u, 2u+1: low halves = even channel, high halves = odd channel
            ev[u] = __builtin_amdgcn_perm(w8[2 * u + 1], w8[2 * u], 0x05040100u);
            od[u] = __builtin_amdgcn_perm(w8[2 * u + 1], w8[2 * u], 0x07060302u);
        }
        fa[0] = __builtin_bit_cast(bf16x8_w, ev);
        fa[1] = __builtin_bit_cast(bf16x8_w, od);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            f32x8_w t8;
#pragma unroll
            for (int t = 0; t < 8; ++t) t8[t] = fb[set][jn][t];
            fbv[jn] = __builtin_convertvector(t8, bf16x8_w);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fbv[jn], acc[i][jn], 0, 0, 0);
    };
    load_batch(0, p_begin);
    for (long p0 = p_begin; p0 < p_end; p0 += 32) {
        load_batch(1, p0 + 16);
        mma_batch(0);
        load_batch(0, p0 + 32);
        mma_batch(1);
    }
    __shared__ float slabS[4][64 * 48];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int k = 2 * rho + i;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int j = jn * 32 + l31;
                if (j < 48) slabS[wave][k * 48 + j] = acc[i][jn][r];
            }
        }
    __syncthreads();
    float* slab = part + ((long)blockIdx.x * K + kg * 64) * 48;
    for (int e = tid; e < 64 * 48; e += 256) slab[e] = ((slabS[0][e] + slabS[1][e]) + slabS[2][e]) + slabS[3][e];
}

// ---- weight gradient on the bf16 matrix path, image rows staged through LDS (W >= 32) ------------------------------------
// The per-lane im2col gathers of the kernel above touch 24 cache lines per load instruction and bound it at ~1.4 TB/s.  Here a
// workgroup walks output rows (n, oy): the 4 x 3 input rows of the row are loaded COALESCED (float4), rounded to bf16 and written
// to LDS de-interleaved by column parity, in four arrays per (channel, filter row) -- one per filter column s -- such that the 8
// consecutive output pixels a lane needs for ITS (c, r, s) are 8 consecutive bf16 = ONE aligned ds_read_b128:
//   s = 0: x[2 ox - 1] -> O0[ox] (O0[0] = 0: left padding)     s = 1: x[2 ox]     -> E[ox]
//   s = 2: x[2 ox + 1] -> O[ox]                                 s = 3: x[2 ox + 2] -> E1[ox] (E1[Wo-1] = 0: right padding)
// Rows above / below the image are staged as zeros.  Two LDS stages: the next output row is staged while the current one is
// multiplied; the A operand (dy, one dword = two channels of a pixel) is loaded as in the kernel above.
// X3 (fp32 dy / act_out, option "bf16" = 2: the f32x3 path): the image rows stay FP32 in LDS (101 KB, one workgroup per CU), dy is
// loaded as fp32 pairs; both operands are split into their three bf16 planes in registers in front of the MFMAs and every block is
// the six plane products of igemm.hip's PREC 2 -- fp32-accurate, 24 MFMAs of 32 cycles per 16 pixels and wave where the fp32-MFMA
// kernel above needs 96 of 64, with coalesced image loads instead of its 24-lines-per-instruction gathers.
#define CWL_WOMAX 256
template <typename F>
__device__ __forceinline__ void c3_static_for4(F&& f) {
    f(std::integral_constant<int, 0>{});
    f(std::integral_constant<int, 1>{});
    f(std::integral_constant<int, 2>{});
    f(std::integral_constant<int, 3>{});
}
template <bool FACT, bool X3 = false>
__global__ __launch_bounds__(256, X3 ? 1 : 2) void c3_wgrad_lds_kernel(const DgPtrs dys, const DgPtrs xs, const DgPtrs parts, int N, int H, int W, int K,
                                                              long npix, int rows_per_wg, const DgPtrs act_outs, float slope) {
    // grouped launch (dg_conv4x4s2_c3_wgrad_g): blockIdx.z = problem
    const float* __restrict__ dy = dg_pick<const float>(dys, blockIdx.z);
    const float* __restrict__ x = dg_pick<const float>(xs, blockIdx.z);
    float* __restrict__ part = dg_pick<float>(parts, blockIdx.z);
    const float* __restrict__ act_out = dg_pick<const float>(act_outs, blockIdx.z);
    constexpr int ROWE = CWL_WOMAX + 8;                       // elements per LDS array row (zero tail)
    typedef typename std::conditional<X3, float, __bf16>::type ET;           // element type of the staged image rows
    __shared__ __attribute__((aligned(16))) ET img[2][4][12][ROWE];          // [stage][s][c * 4 + r][ox]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int Ho = H >> 1, Wo = W >> 1;
    const int kg = blockIdx.y;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // this lane's two im2col columns j = l31, 32 + l31 (valid < 48): (c, r, s) -> LDS array s, row c * 4 + r
    int jrow[2], jarr[2];
    bool jok[2];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const int j = jn * 32 + l31;
        jok[jn] = j < 48;
        const int jj = jok[jn] ? j : 0;
        jarr[jn] = jj & 3;
        jrow[jn] = (jj >> 4) * 4 + ((jj >> 2) & 3);
    }
    const int nrows = N * Ho;
    const int u0 = blockIdx.x * rows_per_wg, u1 = min(nrows, u0 + rows_per_wg);
    constexpr int BIG = 0x40000000;
    constexpr int EB = X3 ? 4 : 2;                            // bytes per dy / act_out element
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)(npix * K * EB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rao = __builtin_amdgcn_make_buffer_rsrc((void*)(FACT ? act_out : dy), 0, (int)(npix * K * EB), 0x00020000);
    auto bload = [](const __amdgpu_buffer_rsrc_t& r, int off) -> unsigned {
        return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    };
    typedef float f32x2_w __attribute__((ext_vector_type(2)));
    auto bload2 = [](const __amdgpu_buffer_rsrc_t& r, int off) -> f32x2_w {
        return __builtin_bit_cast(f32x2_w, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    };
    // ---- staging of output row u: 12 input rows x W / 4 float4 chunks over 256 threads (<= 6 per thread).  The global loads
    // are issued BEFORE the current row's MFMAs (into registers), converted and written to the other LDS stage after them
    const int wq = W >> 2, chunks = 12 * wq;
    f32x4 sv[6];
    auto stage_load = [&](int u) {
        const int n = u / Ho, oy = u - n * Ho;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int e = tid + 256 * k;
            const int rowi = e / wq, m = e - rowi * wq;
            const int c = rowi >> 2, r = rowi & 3;
            const int iy = 2 * oy - 1 + r;
            sv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (e < chunks && (unsigned)iy < (unsigned)H) sv[k] = *(const f32x4*)(x + ((long)(n * 3 + c) * H + iy) * W + 4 * m);
        }
    };
    auto stage_store = [&](int st) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int e = tid + 256 * k;
            if (e >= chunks) continue;
            const int rowi = e / wq, m = e - rowi * wq;          // rowi = c * 4 + r, chunk m = columns 4m .. 4m+3
            typedef ET etx4_s __attribute__((ext_vector_type(4)));
            typedef ET etx2_s __attribute__((ext_vector_type(2)));
            const etx4_s b = __builtin_convertvector(sv[k], etx4_s);              // x[4m], x[4m+1], x[4m+2], x[4m+3]
            ET* E = &img[st][1][rowi][0];
            ET* O = &img[st][2][rowi][0];
            ET* O0 = &img[st][0][rowi][0];
            ET* E1 = &img[st][3][rowi][0];
            *(etx2_s*)(E + 2 * m) = (etx2_s){b[0], b[2]};
            *(etx2_s*)(O + 2 * m) = (etx2_s){b[1], b[3]};
            O0[2 * m + 1] = b[1];                                             // x[2 ox - 1] at ox = 2m + 1
            if (2 * m + 2 < Wo) O0[2 * m + 2] = b[3];
            E1[2 * m] = b[2];                                                 // x[2 ox + 2] at ox = 2m
            if (m > 0) E1[2 * m - 1] = b[0];
            if (m == 0) O0[0] = (ET)0.f;
            if (2 * m + 2 == Wo) E1[Wo - 1] = (ET)0.f;
        }
    };
    // ---- A operand of batch b of row u: 8 pixels per lane, one dword (two channels) each
    // X3: FOUR register sets -- one workgroup of four waves per CU (101 KB of LDS) is one wave per SIMD, and with one item of 8 x 512 bytes
    // per wave in flight the kernel moved 2.6 TB/s (4 MB under way on the whole chip); three items ahead: see the deep loop below
    constexpr int NSET = X3 ? 4 : 2;
    unsigned wa[2][X3 ? 1 : 8], wo[2][(FACT && !X3) ? 8 : 1];
    f32x2_w fa2[NSET][X3 ? 8 : 1], fo2[NSET][(FACT && X3) ? 8 : 1];      // X3: fp32 pairs (channels 2 l31, 2 l31 + 1) of the lane's 8 pixels
    const int nbatch = Wo >> 4;
    auto load_a = [&](int set, int u, int b) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int pp = u * Wo + 16 * b + 8 * lh + t;
            const int aoff = (u < u1 && b < nbatch) ? (pp * K + kg * 64 + 2 * l31) * EB : BIG;
            if constexpr (X3) {
                fa2[set][t] = bload2(rdy, aoff);
                if constexpr (FACT) fo2[set][t] = bload2(rao, aoff);
            } else {
                wa[set][t] = bload(rdy, aoff);
                if constexpr (FACT) wo[set][t] = bload(rao, aoff);
            }
        }
    };
    typedef __bf16 bf16x8_l __attribute__((ext_vector_type(8)));
    typedef unsigned u32x4_l __attribute__((ext_vector_type(4)));
    typedef float f32x8_l __attribute__((ext_vector_type(8)));
    auto split8 = [](const f32x8_l& v, bf16x8_l* pl) {
        pl[0] = __builtin_convertvector(v, bf16x8_l);
        const f32x8_l r1 = v - __builtin_convertvector(pl[0], f32x8_l);
        pl[1] = __builtin_convertvector(r1, bf16x8_l);
        pl[2] = __builtin_convertvector(r1 - __builtin_convertvector(pl[1], f32x8_l), bf16x8_l);
    };
    auto mma_x3 = [&](int set, int st, int b) {
        if constexpr (X3) {
            f32x8_l a0, a1;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                f32x2_w v = fa2[set][t];
                if constexpr (FACT) {
                    const f32x2_w o = fo2[set][t];
                    v[0] = o[0] > 0.f ? v[0] : v[0] * slope;
                    v[1] = o[1] > 0.f ? v[1] : v[1] * slope;
                }
                a0[t] = v[0];
                a1[t] = v[1];
            }
            bf16x8_l pa0[3], pa1[3];
            split8(a0, pa0);
            split8(a1, pa1);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // (dy plane, image plane), smallest product first
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const float* src = (const float*)&img[st][jarr[jn]][jrow[jn]][16 * b + 8 * lh];
                const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
                f32x8_l bv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (!jok[jn]) bv = (f32x8_l){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                bf16x8_l pb[3];
                split8(bv, pb);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa0[PA[q]], pb[PB[q]], acc[0][jn], 0, 0, 0);
                    acc[1][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa1[PA[q]], pb[PB[q]], acc[1][jn], 0, 0, 0);
                }
            }
        }
    };
    auto mma = [&](int set, int st, int b) {
        if constexpr (X3) {
            mma_x3(set, st, b);
            return;
        } else {
        unsigned w8[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            unsigned wv = wa[set][t];
            if constexpr (FACT) {
                const unsigned ov = wo[set][t];
                float a0 = __builtin_bit_cast(float, wv << 16), a1 = __builtin_bit_cast(float, wv & 0xffff0000u);
                a0 = __builtin_bit_cast(float, ov << 16) > 0.f ? a0 : a0 * slope;
                a1 = __builtin_bit_cast(float, ov & 0xffff0000u) > 0.f ? a1 : a1 * slope;
                typedef __bf16 bf16x2_l __attribute__((ext_vector_type(2)));
                typedef float f32x2_l __attribute__((ext_vector_type(2)));
                wv = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_l){a0, a1}, bf16x2_l));
            }
            w8[t] = wv;
        }
        u32x4_l ev, od;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ev[q] = __builtin_amdgcn_perm(w8[2 * q + 1], w8[2 * q], 0x05040100u);
            od[q] = __builtin_amdgcn_perm(w8[2 * q + 1], w8[2 * q], 0x07060302u);
        }
        const bf16x8_l fa0 = __builtin_bit_cast(bf16x8_l, ev), fa1 = __builtin_bit_cast(bf16x8_l, od);
        bf16x8_l fb[2];
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            fb[jn] = *(const bf16x8_l*)(const void*)&img[st][jarr[jn]][jrow[jn]][16 * b + 8 * lh];
            if (!jok[jn]) fb[jn] = __builtin_bit_cast(bf16x8_l, (u32x4_l){0u, 0u, 0u, 0u});
        }
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb[jn], acc[0][jn], 0, 0, 0);
            acc[1][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb[jn], acc[1][jn], 0, 0, 0);
        }
        }
    };
    // zero the tails of the LDS rows once (columns >= Wo are never written by the staging; batches never reach them, but keep
    // the arrays defined)
    for (int e = tid; e < 2 * 4 * 12 * ROWE; e += 256) (&img[0][0][0][0])[e] = (ET)0.f;
    __syncthreads();
    if (u0 < u1) {
        stage_load(u0);
        stage_store(0);
    }
    load_a(0, u0, wave);
    __syncthreads();
    if constexpr (X3) {
        // deep form (rows of 256 output pixels = 4 items per wave and row, the 512 px layer): item j of a row lives in register set j, the
        // A operand of the item THREE ahead is requested before item j is multiplied.  Same items, same order, same arithmetic.
        if (nbatch == 16) {
            load_a(1, u0, wave + 4);
            load_a(2, u0, wave + 8);
            for (int u = u0; u < u1; ++u) {
                const int st = (u - u0) & 1;
                if (u + 1 < u1) stage_load(u + 1);
                c3_static_for4([&](auto J_) {
                    constexpr int J = decltype(J_)::value;
                    constexpr int JN = (J + 3) & 3;
                    load_a(JN, J + 3 >= 4 ? u + 1 : u, wave + 4 * JN);
                    mma_x3(J, st, wave + 4 * J);
                });
                if (u + 1 < u1) stage_store(st ^ 1);
                __syncthreads();
            }
        }
    }
    // this wave's work items: batches wave, wave + 4, ... of every row; while item k is multiplied out of register set (k & 1)
    // the A operand of item k + 1 (of this or the next row) is loaded into the other set
    bool odd = false;
    auto item = [&](auto S_, int u, int b, int st) {
        constexpr int S = decltype(S_)::value;
        const bool last = b + 4 >= nbatch;
        load_a(S ^ 1, last ? u + 1 : u, last ? wave : b + 4);
        mma(S, st, b);
    };
    for (int u = (X3 && nbatch == 16) ? u1 : u0; u < u1; ++u) {
        const int st = (u - u0) & 1;
        if (u + 1 < u1) stage_load(u + 1);                        // the next row's image rows fly under this row's MFMAs
        for (int b = wave; b < nbatch; b += 4) {
            if (odd) item(std::integral_constant<int, 1>{}, u, b, st);
            else item(std::integral_constant<int, 0>{}, u, b, st);
            odd = !odd;
        }
        if (u + 1 < u1) stage_store(st ^ 1);
        __syncthreads();
    }
    // the 4 waves' accumulators are summed through LDS (the image stages are free now: 4 x 12 KB of their 50 KB)
    static_assert(sizeof(img) >= 4 * 64 * 48 * sizeof(float), "slab reduction lives in the image stages");
    float (*slabS)[64 * 48] = (float (*)[64 * 48]) & img[0][0][0][0];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int k = 2 * rho + i;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int j = jn * 32 + l31;
                if (j < 48) slabS[wave][k * 48 + j] = acc[i][jn][r];
            }
        }
    __syncthreads();
    float* slab = part + ((long)blockIdx.x * K + kg * 64) * 48;
    for (int e = tid; e < 64 * 48; e += 256) slab[e] = ((slabS[0][e] + slabS[1][e]) + slabS[2][e]) + slabS[3][e];
}

// fixed-order reduction over slabs: block = 16 outputs x 16 slab lanes
__global__ __launch_bounds__(256) void c3_wgrad_reduce_kernel(const DgPtrs parts, const DgPtrs dws, int nslabs, int total, int accumulate) {
    const float* __restrict__ part = dg_pick<const float>(parts, blockIdx.y);      // grouped launch: blockIdx.y = problem
    float* __restrict__ dw = dg_pick<float>(dws, blockIdx.y);
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    float s = 0.f;
    if (e < total)
        for (int b = sl; b < nslabs; b += 16) s += part[(long)b * total + e];
    red[sl][el] = s;
    __syncthreads();
    if (sl == 0 && e < total) {
#pragma unroll
        for (int j = 1; j < 16; ++j) s += red[j][el];
        if (accumulate) s += dw[e];
        dw[e] = s;
    }
}

// nb workgroups of 4 waves; every wave owns `ppw` consecutive pixels (multiple of one double batch)
static void c3_wgrad_plan(long npix, int* nb, int* ppw) {
    const int unit = 4 * CW_B;                           // pixels per loop trip
    long per = (npix + 2047) / 2048;                     // aim for ~512 workgroups = 2048 waves
    per = (per + unit - 1) / unit * unit;
    if (per < unit) per = unit;
    *ppw = (int)per;
    const long nwaves = (npix + per - 1) / per;
    *nb = (int)((nwaves + 3) / 4);
}
extern "C" size_t dg_c3_wgrad_workspace_bytes(int N, int H, int W, int K) {
    const long npix = (long)N * (H / 2) * (W / 2);
    int nb, ppw;
    c3_wgrad_plan(npix, &nb, &ppw);
    return (size_t)nb * K * 48 * sizeof(float);
}
// groups problems per launch; share > 1: `share` consecutive problems accumulate into the same dw (one main launch over all problems,
// then one reduction launch per member in problem order, each adding to what the previous one left)
static int c3_wgrad_run(const char* who, int groups, int share, const float* const* dy_tab, const float* const* ao_tab, int act, float slope,
                        const float* const* x_tab, float* const* dw_tab, int N, int H, int W, int K, int accumulate, void* const* ws_tab, size_t ws_bytes,
                        dg_stream_t stream, int io_bf16 = 0);
extern "C" int dg_conv4x4s2_c3_wgrad_t(const void* dy_nhwc, const void* act_out_nhwc, int io_bf16, int act, float slope,
                                       const float* x_nchw, float* dw, int N, int H, int W, int K, int accumulate,
                                       void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(act == DG_ACT_NONE || ((act == DG_ACT_LEAKY || act == DG_ACT_RELU) && act_out_nhwc),
                 "dg_conv4x4s2_c3_wgrad_t: act %d needs the saved activation output (LeakyReLU / ReLU only)", act);
    const float* dyp = (const float*)dy_nhwc;
    const float* aop = act == DG_ACT_NONE ? nullptr : (const float*)act_out_nhwc;
    return c3_wgrad_run("dg_conv4x4s2_c3_wgrad_t", 1, 1, &dyp, &aop, act, act == DG_ACT_RELU ? 0.f : slope, &x_nchw, &dw, N, H, W, K, accumulate, &ws,
                        ws_bytes, stream, io_bf16);
}
extern "C" int dg_conv4x4s2_c3_wgrad(const float* dy_nhwc, const float* x_nchw, float* dw, int N, int H, int W, int K,
                                     int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    const float* aop = nullptr;
    return c3_wgrad_run("dg_conv4x4s2_c3_wgrad", 1, 1, &dy_nhwc, &aop, DG_ACT_NONE, 0.f, &x_nchw, &dw, N, H, W, K, accumulate, &ws, ws_bytes, stream);
}
extern "C" int dg_conv4x4s2_c3_wgrad_act(const float* dy_nhwc, const float* act_out_nhwc, int act, float slope,
                                         const float* x_nchw, float* dw, int N, int H, int W, int K, int accumulate,
                                         void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(act == DG_ACT_NONE || ((act == DG_ACT_LEAKY || act == DG_ACT_RELU) && act_out_nhwc),
                 "dg_conv4x4s2_c3_wgrad_act: act %d needs the saved activation output (LeakyReLU / ReLU only)", act);
    const float* aop = act == DG_ACT_NONE ? nullptr : act_out_nhwc;
    return c3_wgrad_run("dg_conv4x4s2_c3_wgrad_act", 1, 1, &dy_nhwc, &aop, act, act == DG_ACT_RELU ? 0.f : slope, &x_nchw, &dw, N, H, W, K, accumulate, &ws,
                        ws_bytes, stream);
}
extern "C" int dg_conv4x4s2_c3_wgrad_p(const void* dy_nhwc, const void* act_out_nhwc, int io_bf16, int act, float slope,
                                       const float* x_nchw, float* dw, int N, int H, int W, int K, int prec, int accumulate,
                                       void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(prec >= DG_PREC_DEFAULT && prec <= DG_PREC_F32X3, "dg_conv4x4s2_c3_wgrad_p: prec=%d", prec);
    DgPrecScope scope(prec);
    return dg_conv4x4s2_c3_wgrad_t(dy_nhwc, act_out_nhwc, io_bf16, act, slope, x_nchw, dw, N, H, W, K, accumulate, ws, ws_bytes, stream);
}
extern "C" int dg_conv4x4s2_c3_wgrad_g(int groups, int share, const float* const* dy_nhwc, const float* const* act_out_nhwc, int act, float slope,
                                       const float* const* x_nchw, float* const* dw, int N, int H, int W, int K, int prec, int accumulate,
                                       void* const* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS && dy_nhwc && x_nchw && dw && ws, "dg_conv4x4s2_c3_wgrad_g: bad group / null table");
    DG_CHECK_ARG(share >= 1 && groups % share == 0, "dg_conv4x4s2_c3_wgrad_g: share=%d", share);
    DG_CHECK_ARG(prec >= DG_PREC_DEFAULT && prec <= DG_PREC_F32X3, "dg_conv4x4s2_c3_wgrad_g: prec=%d", prec);
    DG_CHECK_ARG(act == DG_ACT_NONE || ((act == DG_ACT_LEAKY || act == DG_ACT_RELU) && act_out_nhwc),
                 "dg_conv4x4s2_c3_wgrad_g: act %d needs the saved activation outputs (LeakyReLU / ReLU only)", act);
    const float* none[DG_MAX_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
    DgPrecScope scope(prec);
    return c3_wgrad_run("dg_conv4x4s2_c3_wgrad_g", groups, share, dy_nhwc, act == DG_ACT_NONE ? none : act_out_nhwc, act, act == DG_ACT_RELU ? 0.f : slope,
                        x_nchw, dw, N, H, W, K, accumulate, ws, ws_bytes, stream);
}
static int c3_wgrad_reduce_g(int groups, int share, void* const* ws_tab, float* const* dw_tab, int nb, int total, int accumulate, hipStream_t st) {
    const int nout = groups / share;
    for (int j = 0; j < share; ++j) {
        const void* pp[DG_MAX_GROUPS];
        const void* dd[DG_MAX_GROUPS];
        for (int z = 0; z < nout; ++z) { pp[z] = ws_tab[z * share + j]; dd[z] = dw_tab[z * share + j]; }
        hipLaunchKernelGGL(c3_wgrad_reduce_kernel, dim3((total + 15) / 16, nout), dim3(256), 0, st, dg_ptrs(pp, nout), dg_ptrs(dd, nout), nb, total,
                           (accumulate || j > 0) ? 1 : 0);
        DG_CHECK_LAUNCH("c3_wgrad_reduce");
    }
    return DG_OK;
}
static int c3_wgrad_run(const char* who, int groups, int share, const float* const* dy_tab, const float* const* ao_tab, int act, float slope,
                        const float* const* x_tab, float* const* dw_tab, int N, int H, int W, int K, int accumulate, void* const* ws_tab, size_t ws_bytes,
                        dg_stream_t stream, int io_bf16) {
    for (int i = 0; i < groups; ++i) DG_CHECK_ARG(dy_tab[i] && x_tab[i] && dw_tab[i], "%s: null pointer", who);
    for (int i = 0; i < groups; ++i) DG_CHECK_ARG(dw_tab[i] == dw_tab[i / share * share], "%s: problems of one share set must name the same dw", who);
    const float* dy_nhwc = dy_tab[0];
    const float* x_nchw = x_tab[0];
    const float* act_out = ao_tab[0];
    void* ws = ws_tab[0];
    for (int i = 0; i < groups; ++i) DG_CHECK_ARG((ao_tab[i] != nullptr) == (act_out != nullptr), "%s: activation outputs for all problems or none", who);
    DG_CHECK_ARG(N >= 1 && K >= 64 && K % 64 == 0, "%s: K=%d must be a multiple of 64", who, K);
    DG_CHECK_ARG(dg_is_pow2(H) && dg_is_pow2(W) && H >= 2 && W >= 2, "%s: H,W must be powers of two", who);
    DG_CHECK_ARG((long)N * 3 * H * W < (1L << 31), "%s: tensor too large", who);
    const long npix = (long)N * (H / 2) * (W / 2);
    int nb, ppw;
    c3_wgrad_plan(npix, &nb, &ppw);
    const size_t need = (size_t)nb * K * 48 * sizeof(float);
    for (int i = 0; i < groups; ++i)
        if (ws_tab[i] == nullptr || ws_bytes < need) return dg_fail(DG_ERR_WORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const bool buf = npix * K * 4 < (1L << 30) && (long)N * 3 * H * W * 4 < (1L << 30) && dg_get_option(DG_OPT_POINTER_PATH) == 0;
    const bool fact = act_out != nullptr;
    const DgPtrs pdy = dg_ptrs((const void* const*)dy_tab, groups), px = dg_ptrs((const void* const*)x_tab, groups),
                 pws = dg_ptrs((const void* const*)ws_tab, groups), pao = dg_ptrs((const void* const*)ao_tab, groups);
    const dim3 grid(nb, K / 64, groups);
#define CW_LAUNCH(B, F)                                                                                                  \
    hipLaunchKernelGGL((c3_wgrad_mfma_kernel<B, F>), grid, dim3(256), 0, st, pdy, px, pws, N, H, W, \
                       K, dg_ilog2(H / 2), dg_ilog2(W / 2), npix, ppw, pao, slope)
    if (io_bf16 && !buf) return dg_fail(DG_ERR_INVALID, "%s: bf16 operands need tensors < 1 GiB", who);
    // f32x3 path: fp32-accurate plane products on the bf16 MFMA, image rows staged through LDS (fp32) -- from 256-pixel rows on.  On
    // shorter rows its one workgroup of four waves per CU is latency-bound and the exact-fp32 MFMA kernel below (at least as accurate)
    // is faster: same-box, ms per call f32 MFMA / this kernel: 64 px batch 64 0.026 / 0.039, batch 256 0.047 / 0.078, 128 px batch 64
    // 0.047 / 0.056, 256 px batch 32 0.076 / 0.075, 512 px batch 32 0.296 / 0.254 (tools/bench_ops.py, round 4)
    if (!io_bf16 && buf && dg_cur_prec() == 2 && W >= 256 && W / 2 <= CWL_WOMAX && dg_get_option(DG_OPT_KT) != 16) {
        const int nrows = N * (H / 2);
        const int rpw = (nrows + nb - 1) / nb;
        if (fact)
            hipLaunchKernelGGL((c3_wgrad_lds_kernel<true, true>), grid, dim3(256), 0, st, pdy, px, pws, N, H, W, K, npix, rpw, pao, slope);
        else
            hipLaunchKernelGGL((c3_wgrad_lds_kernel<false, true>), grid, dim3(256), 0, st, pdy, px, pws, N, H, W, K, npix, rpw, pao, slope);
        DG_CHECK_LAUNCH("c3_wgrad_lds_x3");
        return c3_wgrad_reduce_g(groups, share, ws_tab, dw_tab, nb, K * 48, accumulate, st);
    }
    if (io_bf16 && dg_cur_prec() == 1 && W >= 32 && W / 2 <= CWL_WOMAX && dg_get_option(DG_OPT_KT) != 16) {
        // bf16 matrix path, image rows staged through LDS: the same nb slabs, a contiguous range of output rows per workgroup
        const int nrows = N * (H / 2);
        const int rpw = (nrows + nb - 1) / nb;
        if (fact)
            hipLaunchKernelGGL(c3_wgrad_lds_kernel<true>, grid, dim3(256), 0, st, pdy, px, pws, N, H, W, K, npix, rpw, pao, slope);
        else
            hipLaunchKernelGGL(c3_wgrad_lds_kernel<false>, grid, dim3(256), 0, st, pdy, px, pws, N, H, W, K, npix, rpw, pao, slope);
    } else if (io_bf16 && dg_cur_prec() == 1) {        // bf16 matrix path: bf16 MFMA, per-lane gathers
        DG_CHECK_ARG(groups == 1, "%s: this kernel takes one problem", who);
        if (fact)
            hipLaunchKernelGGL(c3_wgrad_bf16mfma_kernel<true>, dim3(nb, K / 64), dim3(256), 0, st, dy_nhwc, x_nchw, (float*)ws, N, H, W,
                               K, dg_ilog2(H / 2), dg_ilog2(W / 2), npix, ppw, act_out, slope);
        else
            hipLaunchKernelGGL(c3_wgrad_bf16mfma_kernel<false>, dim3(nb, K / 64), dim3(256), 0, st, dy_nhwc, x_nchw, (float*)ws, N, H, W,
                               K, dg_ilog2(H / 2), dg_ilog2(W / 2), npix, ppw, act_out, slope);
    } else if (io_bf16 && fact) {
        hipLaunchKernelGGL((c3_wgrad_mfma_kernel<true, true, true>), grid, dim3(256), 0, st, pdy, px, pws, N, H, W,
                           K, dg_ilog2(H / 2), dg_ilog2(W / 2), npix, ppw, pao, slope);
    } else if (io_bf16) {
        hipLaunchKernelGGL((c3_wgrad_mfma_kernel<true, false, true>), grid, dim3(256), 0, st, pdy, px, pws, N, H, W,
                           K, dg_ilog2(H / 2), dg_ilog2(W / 2), npix, ppw, pao, slope);
    } else if (buf && fact) CW_LAUNCH(true, true);
    else if (buf) CW_LAUNCH(true, false);
    else if (fact) CW_LAUNCH(false, true);
    else CW_LAUNCH(false, false);
#undef CW_LAUNCH
    DG_CHECK_LAUNCH("c3_wgrad_mfma");
    return c3_wgrad_reduce_g(groups, share, ws_tab, dw_tab, nb, K * 48, accumulate, st);
}
