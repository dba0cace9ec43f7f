// 3-channel edge layers, image side NCHW (reference hands over / receives NCHW images:
// image_translation.py:332-333, model.py:8,80,142).  The forward direction (3 -> K) lives in
// igemm.hip (MODE_FWD_C3, MFMA).  Here: the K -> 3 direction and the [K][3][4][4] weight gradient.
// Both are HBM-bound (AI ~ 20 FLOP/B) and run on the VALU.
#include "dg_common.h"

// dx_nchw[n,c,2a+ph,2b+pw] = act( sum_{taps,k} dy[n,a+da,b+db,k] * w[k][c][r][s] )
// One thread = one 2x2 output quad (all 3 channels): weights are wave-uniform -> scalar loads.
// Taps per output parity (conv k4 s2 p1): p=0 -> (r=1, d=0), (r=3, d=-1);  p=1 -> (r=2, d=0), (r=0, d=+1).
__global__ __launch_bounds__(256) void c3_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
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
            float2 o = make_float2(v0, v1);
            *(float2*)(dx + ((long)(n * 3 + c) * H + 2 * a + ph) * W + 2 * b) = o;
        }
}

extern "C" int dg_conv4x4s2_c3_dgrad(const float* dy_nhwc, const float* w, float* dx_nchw, int N, int H, int W, int K,
                                     int act, dg_stream_t stream) {
    DG_CHECK_ARG(dy_nhwc && w && dx_nchw, "dg_conv4x4s2_c3_dgrad: null pointer");
    DG_CHECK_ARG(N >= 1 && K >= 4 && K % 4 == 0, "dg_conv4x4s2_c3_dgrad: bad N/K (%d,%d)", N, K);
    DG_CHECK_ARG(dg_is_pow2(H) && dg_is_pow2(W) && H >= 2 && W >= 2, "dg_conv4x4s2_c3_dgrad: H,W must be powers of two");
    DG_CHECK_ARG(act == DG_ACT_NONE || act == DG_ACT_SIGMOID, "dg_conv4x4s2_c3_dgrad: bad act %d", act);
    const long nquad = (long)N * (H / 2) * (W / 2);
    hipLaunchKernelGGL(c3_dgrad_kernel, dim3((unsigned)((nquad + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dy_nhwc, w, dx_nchw, N, H, W, K, dg_ilog2(H / 2), dg_ilog2(W / 2), act);
    DG_CHECK_LAUNCH("c3_dgrad");
    return DG_OK;
}

// dw[k][c][r][s] (+)= sum_{pixels} dy[pix][k] * x[n,c,2oy-1+r,2ox-1+s]
// Block: 256 threads = 64 k-lanes x 4 parts (12 of the 48 (c,r,s) entries each).  Pixels are staged
// 64 at a time as an im2col patch in LDS (broadcast reads); per-block partial sums go to the workspace
// and a second kernel reduces them in a fixed order (deterministic, no atomics).
#define C3W_PIX 64
__global__ __launch_bounds__(256) void c3_wgrad_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               float* __restrict__ part, int N, int H, int W, int K,
                                                               int lgHo, int lgWo, long npix, int kgroups) {
    __shared__ __attribute__((aligned(16))) float patch[C3W_PIX][48];
    const int Ho = H >> 1, Wo = W >> 1;
    const int tid = threadIdx.x;
    const int kl = tid & 63, prt = tid >> 6;
    const int kg = blockIdx.y;  // 64-wide group of output channels
    const int k = kg * 64 + kl;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (long p0 = (long)blockIdx.x * C3W_PIX; p0 < npix; p0 += (long)gridDim.x * C3W_PIX) {
        __syncthreads();
        // stage the im2col patch: 64 pixels x 48 values, 12 per thread
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int e = tid + i * 256;       // 0..3071
            const int pl = e & 63, qq = e >> 6;  // pixel in stage, (c,r,s) index 0..47
            const long pp = p0 + pl;
            float v = 0.f;
            if (pp < npix) {
                const int ox = (int)(pp & (Wo - 1)), oy = (int)((pp >> lgWo) & (Ho - 1)), n = (int)(pp >> (lgWo + lgHo));
                const int c = qq >> 4, r = (qq >> 2) & 3, s = qq & 3;
                const int iy = 2 * oy - 1 + r, ix = 2 * ox - 1 + s;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[((long)(n * 3 + c) * H + iy) * W + ix];
            }
            patch[pl][qq] = v;
        }
        __syncthreads();
        const int npl = (int)min((long)C3W_PIX, npix - p0);
        if (k < K) {
            for (int pl = 0; pl < npl; ++pl) {
                const float g = dy[(p0 + pl) * K + k];
                const f32x4 a0 = *(const f32x4*)&patch[pl][prt * 12 + 0];
                const f32x4 a1 = *(const f32x4*)&patch[pl][prt * 12 + 4];
                const f32x4 a2 = *(const f32x4*)&patch[pl][prt * 12 + 8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] += g * a0[j];
                    acc[4 + j] += g * a1[j];
                    acc[8 + j] += g * a2[j];
                }
            }
        }
    }
    if (k < K) {
        float* dst = part + ((long)blockIdx.x * K + k) * 48 + prt * 12;
#pragma unroll
        for (int i = 0; i < 12; ++i) dst[i] = acc[i];
    }
}
__global__ __launch_bounds__(256) void c3_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                              int nparts, int total, int accumulate) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float s = 0.f;
    for (int b = 0; b < nparts; ++b) s += part[(long)b * total + e];
    if (accumulate) s += dw[e];
    dw[e] = s;
}

static int c3_wgrad_blocks(long npix) {
    long nb = (npix + C3W_PIX - 1) / C3W_PIX;
    if (nb > 1024) nb = 1024;
    return (int)nb;
}
extern "C" size_t dg_c3_wgrad_workspace_bytes(int N, int H, int W, int K) {
    const long npix = (long)N * (H / 2) * (W / 2);
    return (size_t)c3_wgrad_blocks(npix) * K * 48 * sizeof(float);
}
extern "C" int dg_conv4x4s2_c3_wgrad(const float* dy_nhwc, const float* x_nchw, float* dw, int N, int H, int W, int K,
                                     int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(dy_nhwc && x_nchw && dw, "dg_conv4x4s2_c3_wgrad: null pointer");
    DG_CHECK_ARG(N >= 1 && K >= 1, "dg_conv4x4s2_c3_wgrad: bad N/K (%d,%d)", N, K);
    DG_CHECK_ARG(dg_is_pow2(H) && dg_is_pow2(W) && H >= 2 && W >= 2, "dg_conv4x4s2_c3_wgrad: H,W must be powers of two");
    const long npix = (long)N * (H / 2) * (W / 2);
    const int nb = c3_wgrad_blocks(npix);
    const size_t need = (size_t)nb * K * 48 * sizeof(float);
    if (ws == nullptr || ws_bytes < need) return dg_fail(DG_ERR_WORKSPACE, "dg_conv4x4s2_c3_wgrad: workspace %zu < %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const int kgroups = (K + 63) / 64;
    hipLaunchKernelGGL(c3_wgrad_partial_kernel, dim3(nb, kgroups), dim3(256), 0, st, dy_nhwc, x_nchw, (float*)ws,
                       N, H, W, K, dg_ilog2(H / 2), dg_ilog2(W / 2), npix, kgroups);
    DG_CHECK_LAUNCH("c3_wgrad_partial");
    const int total = K * 48;
    hipLaunchKernelGGL(c3_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)ws, dw, nb, total, accumulate);
    DG_CHECK_LAUNCH("c3_wgrad_reduce");
    return DG_OK;
}
