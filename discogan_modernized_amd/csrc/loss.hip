// Loss kernels: scalars stay in device memory, reductions are two-stage with a fixed order
// (block partials -> one block, fp64) so results are bitwise reproducible.
//   MSE   nn.MSELoss()            image_translation.py:267,349-350
//   BCE   nn.BCELoss()            image_translation.py:268,162-166  (log clamp -100; backward guard 1e-12;
//                                 deliberately NOT fused with the sigmoid)
//   FM    get_fm_loss, one layer  image_translation.py:136-144 (HingeEmbeddingLoss(x, ones) == mean(x))
#include "dg_common.h"

#define LOSS_MAX_BLOCKS 1024

extern "C" size_t dg_loss_workspace_bytes(void) { return LOSS_MAX_BLOCKS * sizeof(double); }

// grouped launches (dg_*_g; round 4): blockIdx.y (blockIdx.x in the one-block kernels) = problem, one tensor per problem (DgPtrs)
__global__ __launch_bounds__(256) void final_sum_kernel(const DgPtrs parts, int nparts, double scale, const DgPtrs outs) {
    const double* __restrict__ part = dg_pick<const double>(parts, blockIdx.x);
    float* __restrict__ out = dg_pick<float>(outs, blockIdx.x);
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = dg_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) * scale);
}

__device__ __forceinline__ void block_partial_store(float v, double* part) {
    __shared__ double red[4];
    double d = dg_wave_sum_d((double)v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void mse_partial_kernel(const DgPtrs xs, const DgPtrs ts, long n, const DgPtrs parts) {
    const float* __restrict__ x = dg_pick<const float>(xs, blockIdx.y);
    const float* __restrict__ t = dg_pick<const float>(ts, blockIdx.y);
    double* __restrict__ part = dg_pick<double>(parts, blockIdx.y);
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 a = *(const f32x4*)(x + i * 4), b = *(const f32x4*)(t + i * 4);
        const f32x4 d = a - b;
        s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long e = n4 * 4; e < n; ++e) s += (x[e] - t[e]) * (x[e] - t[e]);
    block_partial_store(s, part);
}
__global__ __launch_bounds__(256) void mse_bwd_kernel(const DgPtrs xs, const DgPtrs ts, long n, const DgPtrs gouts, const DgPtrs dxs) {
    const float* __restrict__ x = dg_pick<const float>(xs, blockIdx.y);
    const float* __restrict__ t = dg_pick<const float>(ts, blockIdx.y);
    const float* __restrict__ gout = dg_pick<const float>(gouts, blockIdx.y);
    float* __restrict__ dx = dg_pick<float>(dxs, blockIdx.y);
    const float sc = 2.f * gout[0] / (float)n;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 a = *(const f32x4*)(x + i * 4), b = *(const f32x4*)(t + i * 4);
        *(f32x4*)(dx + i * 4) = (a - b) * sc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long e = n4 * 4; e < n; ++e) dx[e] = (x[e] - t[e]) * sc;
}

struct DgLabels {
    float v[DG_MAX_GROUPS];
};
__global__ __launch_bounds__(256) void bce_fwd_kernel(const DgPtrs ps, int n, const DgLabels labels, const DgPtrs losses) {
    const float* __restrict__ p = dg_pick<const float>(ps, blockIdx.x);
    float* __restrict__ loss = dg_pick<float>(losses, blockIdx.x);
    const float label = labels.v[blockIdx.x];
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = p[i];
        const float l1 = fmaxf(logf(v), -100.f), l0 = fmaxf(logf(1.f - v), -100.f);
        s += (double)((label - 1.f) * l0 - label * l1);
    }
    s = dg_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)((red[0] + red[1] + red[2] + red[3]) / n);
}
__global__ __launch_bounds__(256) void bce_bwd_kernel(const DgPtrs ps, int n, const DgLabels labels, const DgPtrs gouts, const DgPtrs dps) {
    const float* __restrict__ p = dg_pick<const float>(ps, blockIdx.y);
    const float* __restrict__ gout = dg_pick<const float>(gouts, blockIdx.y);
    float* __restrict__ dp = dg_pick<float>(dps, blockIdx.y);
    const float label = labels.v[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = p[i];
    dp[i] = gout[0] * (v - label) / fmaxf((1.f - v) * v, 1e-12f) / (float)n;
}

// nn.BCELoss against an arbitrary target tensor (image_translation.py:157-166 builds ones / zeros tensors on the
// host; the trainer passes the constant as a scalar, this form serves callers that keep the reference's tensors).
__global__ __launch_bounds__(256) void bce_target_fwd_kernel(const float* __restrict__ p, const float* __restrict__ t, int n,
                                                             float* __restrict__ loss) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = p[i], y = t[i];
        const float l1 = fmaxf(logf(v), -100.f), l0 = fmaxf(logf(1.f - v), -100.f);
        s += (double)((y - 1.f) * l0 - y * l1);
    }
    s = dg_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)((red[0] + red[1] + red[2] + red[3]) / n);
}
__global__ __launch_bounds__(256) void bce_target_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t, int n,
                                                             const float* __restrict__ gout, float* __restrict__ dp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = p[i];
    dp[i] = gout[0] * (v - t[i]) / fmaxf((1.f - v) * v, 1e-12f) / (float)n;
}

// nn.HingeEmbeddingLoss(margin 1, mean): l_i = x_i (y_i == 1) | max(0, margin - x_i) (y_i == -1)
// (image_translation.py:141-142 calls it with all-ones targets, where it is x.mean()).
__global__ __launch_bounds__(256) void hinge_partial_kernel(const float* __restrict__ x, const float* __restrict__ y, long n,
                                                            float margin, double* __restrict__ part) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float xv = x[i], yv = y[i];
        s += (yv == 1.f ? xv : 0.f) + (yv == -1.f ? fmaxf(margin - xv, 0.f) : 0.f);
    }
    block_partial_store(s, part);
}
__global__ __launch_bounds__(256) void hinge_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, long n,
                                                        float margin, const float* __restrict__ gout, float* __restrict__ dx) {
    const float g = gout[0] / (float)n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float xv = x[i], yv = y[i];
        dx[i] = (yv == 1.f ? g : 0.f) + ((yv == -1.f && margin - xv > 0.f) ? -g : 0.f);
    }
}

// Feature matching, stage 1: batch-chunk partial sums.  grid = (J/4/256 blocks, nchunks); each thread
// owns one float4 column and walks its chunk of the batch (coalesced 4 KB rows per block).
// part layout: [2][nchunks][J]  (0: real, 1: fake)
// T = float or __bf16: element type of the discriminator features (and of their gradients); sums stay fp32
typedef __bf16 fm_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 fm_ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 fm_ld4(const __bf16* p) { return __builtin_convertvector(*(const fm_bf16x4*)p, f32x4); }
__device__ __forceinline__ void fm_st4(float* p, const f32x4& v) { *(f32x4*)p = v; }
__device__ __forceinline__ void fm_st4(__bf16* p, const f32x4& v) { *(fm_bf16x4*)p = __builtin_convertvector(v, fm_bf16x4); }
template <typename T>
__global__ __launch_bounds__(256) void fm_partial_kernel(const DgPtrs reals, const DgPtrs fakes, int N, long J, int nchunks, const DgPtrs parts) {
    const T* __restrict__ real = dg_pick<const T>(reals, blockIdx.z);
    const T* __restrict__ fake = dg_pick<const T>(fakes, blockIdx.z);
    float* __restrict__ part = dg_pick<float>(parts, blockIdx.z);
    const long j4 = (long)blockIdx.x * 256 + threadIdx.x;
    if (j4 * 4 >= J) return;
    const int per = (N + nchunks - 1) / nchunks;
    const int n0 = blockIdx.y * per, n1 = min(N, n0 + per);
    f32x4 sr = {0.f, 0.f, 0.f, 0.f}, sf = {0.f, 0.f, 0.f, 0.f};
    for (int n = n0; n < n1; ++n) {
        sr += fm_ld4(real + (long)n * J + j4 * 4);
        sf += fm_ld4(fake + (long)n * J + j4 * 4);
    }
    *(f32x4*)(part + (long)blockIdx.y * J + j4 * 4) = sr;
    *(f32x4*)(part + ((long)nchunks + blockIdx.y) * J + j4 * 4) = sf;
}
// stage 2: diff[j] = mean_n real - mean_n fake (fixed chunk order), block partials of diff^2
__global__ __launch_bounds__(256) void fm_diff_kernel(const DgPtrs parts, int N, long J, int nchunks, const DgPtrs diffs, const DgPtrs dparts) {
    const float* __restrict__ part = dg_pick<const float>(parts, blockIdx.y);
    float* __restrict__ diff = dg_pick<float>(diffs, blockIdx.y);
    double* __restrict__ dpart = dg_pick<double>(dparts, blockIdx.y);
    float s = 0.f;
    const long j4n = J >> 2;
    const float inv = 1.f / (float)N;
    for (long j4 = (long)blockIdx.x * 256 + threadIdx.x; j4 < j4n; j4 += (long)gridDim.x * 256) {
        f32x4 sr = {0.f, 0.f, 0.f, 0.f}, sf = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunks; ++c) {
            sr += *(const f32x4*)(part + (long)c * J + j4 * 4);
            sf += *(const f32x4*)(part + ((long)nchunks + c) * J + j4 * 4);
        }
        const f32x4 d = sr * inv - sf * inv;
        *(f32x4*)(diff + j4 * 4) = d;
        s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
    block_partial_store(s, dpart);
}
template <typename T>
__global__ __launch_bounds__(256) void fm_bwd_kernel(const DgPtrs diffs, int N, long J, const DgPtrs gouts, const DgPtrs dreals, const DgPtrs dfakes) {
    const float* __restrict__ diff = dg_pick<const float>(diffs, blockIdx.y);
    const float* __restrict__ gout = dg_pick<const float>(gouts, blockIdx.y);
    T* __restrict__ dreal = dg_pick<T>(dreals, blockIdx.y);
    T* __restrict__ dfake = dg_pick<T>(dfakes, blockIdx.y);
    // d loss / d real[n][j] = 2*diff[j] / (N*J);  d/d fake = -that
    const float sc = 2.f * gout[0] / ((float)N * (float)J);
    const long j4n = J >> 2;
    const long total = j4n * N;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long n = idx / j4n, j4 = idx - n * j4n;
        const f32x4 d = *(const f32x4*)(diff + j4 * 4) * sc;
        if (dreal) fm_st4(dreal + n * J + j4 * 4, d);
        if (dfake) fm_st4(dfake + n * J + j4 * 4, -d);
    }
}

// ---- loss mix (image_translation.py:162-166, 367-382) -----------------------------------------------------
// lossvec slots:  0 recon_A  1 recon_B | 2 bce(D_A real,1) 3 bce(D_A fake,0) 4 bce(D_A fake,1) | 5,6,7 same for D_B
//                 8 .. 8+nfm-1  feature-matching layers of D_A | 8+nfm .. 8+2nfm-1  of D_B
// out slots:      0 gen_loss_A 1 gen_loss_B 2 fm_loss_A 3 fm_loss_B 4 dis_loss_A 5 dis_loss_B 6 gen_loss 7 dis_loss
// arch: 0 discogan, 1 recongan, 2 gan.  One thread; fp32 operations in the reference's order.
#define LM_FM0 8
__device__ __forceinline__ void loss_mix_coeffs(int arch, float rate, float* cA_fmgan, float* cB_fmgan, float* cA_rec,
                                                float* cB_rec, float* dA, float* dB) {
    // gen_loss = sA*[(fm_B*0.9 + gen_B*0.1)*(1-rate) + recon_A*rate] + sB*[(fm_A*0.9 + gen_A*0.1)*(1-rate) + recon_B*rate]
    const float omr = 1.f - rate;
    if (arch == 0) { *cA_fmgan = omr; *cB_fmgan = omr; *cA_rec = rate; *cB_rec = rate; *dA = 1.f; *dB = 1.f; }
    else if (arch == 1) { *cA_fmgan = omr; *cB_fmgan = 0.f; *cA_rec = rate; *cB_rec = 0.f; *dA = 0.f; *dB = 1.f; }
    else { *cA_fmgan = 1.f; *cB_fmgan = 0.f; *cA_rec = 0.f; *cB_rec = 0.f; *dA = 0.f; *dB = 1.f; }
}
__global__ void loss_mix_fwd_kernel(const float* __restrict__ lv, float* __restrict__ out, int nfm, float rate, int arch) {
    float fmA = 0.f, fmB = 0.f;
    for (int l = 0; l < nfm; ++l) { fmA += lv[LM_FM0 + l]; fmB += lv[LM_FM0 + nfm + l]; }
    const float disA = (lv[2] + lv[3]) * 0.5f, disB = (lv[5] + lv[6]) * 0.5f;
    const float genA = lv[4], genB = lv[7];
    const float omr = 1.f - rate;
    const float totA = (fmB * 0.9f + genB * 0.1f) * omr + lv[0] * rate;   // gen_loss_A_total (cross-wired, :370)
    const float totB = (fmA * 0.9f + genA * 0.1f) * omr + lv[1] * rate;
    float gen, dis;
    if (arch == 0) { gen = totA + totB; dis = disA + disB; }
    else if (arch == 1) { gen = totA; dis = disB; }
    else { gen = genB * 0.1f + fmB * 0.9f; dis = disB; }
    out[0] = genA; out[1] = genB; out[2] = fmA; out[3] = fmB; out[4] = disA; out[5] = disB; out[6] = gen; out[7] = dis;
}
// gradient seeds for every lossvec slot given d(gen_loss) or d(dis_loss) (which: 6 or 7)
__global__ void loss_mix_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gv, int nfm, float rate, int arch, int which) {
    const float g = gout[0];
    for (int i = 0; i < LM_FM0 + 2 * nfm; ++i) gv[i] = 0.f;
    if (which == 7) {
        const float dA = arch == 0 ? 1.f : 0.f;
        gv[2] = gv[3] = g * dA * 0.5f;
        gv[5] = gv[6] = g * 0.5f;
        return;
    }
    const float omr = 1.f - rate;
    if (arch == 2) {
        gv[7] = g * 0.1f;
        for (int l = 0; l < nfm; ++l) gv[LM_FM0 + nfm + l] = g * 0.9f;
        return;
    }
    // totA path (always): fm_B, gen_B, recon_A
    gv[0] = g * rate;
    gv[7] = g * omr * 0.1f;
    for (int l = 0; l < nfm; ++l) gv[LM_FM0 + nfm + l] = g * omr * 0.9f;
    if (arch == 0) {   // totB path: fm_A, gen_A, recon_B
        gv[1] = g * rate;
        gv[4] = g * omr * 0.1f;
        for (int l = 0; l < nfm; ++l) gv[LM_FM0 + l] = g * omr * 0.9f;
    }
}

static int loss_grid(long work) {
    long g = (work + 255) / 256;
    if (g > LOSS_MAX_BLOCKS) g = LOSS_MAX_BLOCKS;
    if (g < 1) g = 1;
    return (int)g;
}

#define DG_GROUP_TABLES_OK(who, ...)                                                                                     \
    do {                                                                                                                \
        DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS, who ": groups=%d (1..%d)", groups, DG_MAX_GROUPS);            \
        const void* const* tabs__[] = {__VA_ARGS__};                                                                    \
        for (size_t t__ = 0; t__ < sizeof(tabs__) / sizeof(tabs__[0]); ++t__) {                                         \
            DG_CHECK_ARG(tabs__[t__] != nullptr, who ": null pointer table");                                           \
            for (int i__ = 0; i__ < groups; ++i__) DG_CHECK_ARG(tabs__[t__][i__] != nullptr, who ": null pointer (problem %d)", i__); \
        }                                                                                                               \
    } while (0)
#define DG_TAB(x) ((const void* const*)(x))

extern "C" int dg_mse_fwd_g(int groups, const float* const* x, const float* const* t, size_t n, float* const* loss, void* const* ws, size_t ws_bytes,
                            dg_stream_t stream) {
    DG_GROUP_TABLES_OK("dg_mse_fwd", DG_TAB(x), DG_TAB(t), DG_TAB(loss), DG_TAB(ws));
    DG_CHECK_ARG(n > 0, "dg_mse_fwd: bad argument");
    if (ws_bytes < dg_loss_workspace_bytes()) return dg_fail(DG_ERR_WORKSPACE, "dg_mse_fwd: workspace too small");
    const int g = loss_grid((long)(n / 4) + 1);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(g, groups), dim3(256), 0, st, dg_ptrs(DG_TAB(x), groups), dg_ptrs(DG_TAB(t), groups), (long)n,
                       dg_ptrs(DG_TAB(ws), groups));
    DG_CHECK_LAUNCH("mse_partial");
    hipLaunchKernelGGL(final_sum_kernel, dim3(groups), dim3(256), 0, st, dg_ptrs(DG_TAB(ws), groups), g, 1.0 / (double)n, dg_ptrs(DG_TAB(loss), groups));
    DG_CHECK_LAUNCH("mse_final");
    return DG_OK;
}
extern "C" int dg_mse_fwd(const float* x, const float* t, size_t n, float* loss, void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(x && t && loss && n > 0, "dg_mse_fwd: bad argument");
    if (!ws || ws_bytes < dg_loss_workspace_bytes()) return dg_fail(DG_ERR_WORKSPACE, "dg_mse_fwd: workspace too small");
    return dg_mse_fwd_g(1, &x, &t, n, &loss, &ws, ws_bytes, stream);
}
extern "C" int dg_mse_bwd_g(int groups, const float* const* x, const float* const* t, size_t n, const float* const* gout, float* const* dx,
                            dg_stream_t stream) {
    DG_GROUP_TABLES_OK("dg_mse_bwd", DG_TAB(x), DG_TAB(t), DG_TAB(gout), DG_TAB(dx));
    DG_CHECK_ARG(n > 0, "dg_mse_bwd: bad argument");
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(loss_grid((long)(n / 4) + 1) * 2, groups), dim3(256), 0, (hipStream_t)stream, dg_ptrs(DG_TAB(x), groups),
                       dg_ptrs(DG_TAB(t), groups), (long)n, dg_ptrs(DG_TAB(gout), groups), dg_ptrs(DG_TAB(dx), groups));
    DG_CHECK_LAUNCH("mse_bwd");
    return DG_OK;
}
extern "C" int dg_mse_bwd(const float* x, const float* t, size_t n, const float* gout, float* dx, dg_stream_t stream) {
    DG_CHECK_ARG(x && t && gout && dx && n > 0, "dg_mse_bwd: bad argument");
    return dg_mse_bwd_g(1, &x, &t, n, &gout, &dx, stream);
}
static DgLabels bce_labels(int groups, const float* label) {
    DgLabels l;
    for (int i = 0; i < DG_MAX_GROUPS; ++i) l.v[i] = i < groups ? label[i] : 0.f;
    return l;
}
extern "C" int dg_bce_fwd_g(int groups, const float* const* p, int n, const float* label, float* const* loss, dg_stream_t stream) {
    DG_GROUP_TABLES_OK("dg_bce_fwd", DG_TAB(p), DG_TAB(loss));
    DG_CHECK_ARG(label && n > 0, "dg_bce_fwd: bad argument");
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, dg_ptrs(DG_TAB(p), groups), n, bce_labels(groups, label),
                       dg_ptrs(DG_TAB(loss), groups));
    DG_CHECK_LAUNCH("bce_fwd");
    return DG_OK;
}
extern "C" int dg_bce_fwd(const float* p, int n, float label, float* loss, void* ws, size_t ws_bytes, dg_stream_t stream) {
    (void)ws; (void)ws_bytes;
    DG_CHECK_ARG(p && loss && n > 0, "dg_bce_fwd: bad argument");
    return dg_bce_fwd_g(1, &p, n, &label, &loss, stream);
}
extern "C" int dg_bce_bwd_g(int groups, const float* const* p, int n, const float* label, const float* const* gout, float* const* dp,
                            dg_stream_t stream) {
    DG_GROUP_TABLES_OK("dg_bce_bwd", DG_TAB(p), DG_TAB(gout), DG_TAB(dp));
    DG_CHECK_ARG(label && n > 0, "dg_bce_bwd: bad argument");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((n + 255) / 256, groups), dim3(256), 0, (hipStream_t)stream, dg_ptrs(DG_TAB(p), groups), n,
                       bce_labels(groups, label), dg_ptrs(DG_TAB(gout), groups), dg_ptrs(DG_TAB(dp), groups));
    DG_CHECK_LAUNCH("bce_bwd");
    return DG_OK;
}
extern "C" int dg_bce_bwd(const float* p, int n, float label, const float* gout, float* dp, dg_stream_t stream) {
    DG_CHECK_ARG(p && gout && dp && n > 0, "dg_bce_bwd: bad argument");
    return dg_bce_bwd_g(1, &p, n, &label, &gout, &dp, stream);
}
extern "C" int dg_bce_target_fwd(const float* p, const float* target, int n, float* loss, dg_stream_t stream) {
    DG_CHECK_ARG(p && target && loss && n > 0, "dg_bce_target_fwd: bad argument");
    hipLaunchKernelGGL(bce_target_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, target, n, loss);
    DG_CHECK_LAUNCH("bce_target_fwd");
    return DG_OK;
}
extern "C" int dg_bce_target_bwd(const float* p, const float* target, int n, const float* gout, float* dp, dg_stream_t stream) {
    DG_CHECK_ARG(p && target && gout && dp && n > 0, "dg_bce_target_bwd: bad argument");
    hipLaunchKernelGGL(bce_target_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, target, n, gout, dp);
    DG_CHECK_LAUNCH("bce_target_bwd");
    return DG_OK;
}
extern "C" int dg_hinge_fwd(const float* x, const float* y, size_t n, float margin, float* loss, void* ws, size_t ws_bytes,
                            dg_stream_t stream) {
    DG_CHECK_ARG(x && y && loss && n > 0, "dg_hinge_fwd: bad argument");
    if (!ws || ws_bytes < dg_loss_workspace_bytes()) return dg_fail(DG_ERR_WORKSPACE, "dg_hinge_fwd: workspace too small");
    const int g = loss_grid((long)n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(hinge_partial_kernel, dim3(g), dim3(256), 0, st, x, y, (long)n, margin, (double*)ws);
    DG_CHECK_LAUNCH("hinge_partial");
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, dg_ptrs1(ws), g, 1.0 / (double)n, dg_ptrs1(loss));
    DG_CHECK_LAUNCH("hinge_final");
    return DG_OK;
}
extern "C" int dg_hinge_bwd(const float* x, const float* y, size_t n, float margin, const float* gout, float* dx,
                            dg_stream_t stream) {
    DG_CHECK_ARG(x && y && gout && dx && n > 0, "dg_hinge_bwd: bad argument");
    hipLaunchKernelGGL(hinge_bwd_kernel, dim3(loss_grid((long)n)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, margin, gout, dx);
    DG_CHECK_LAUNCH("hinge_bwd");
    return DG_OK;
}
static int fm_chunks(int N, size_t J) {
    const long blocks = (long)((J / 4 + 255) / 256);
    long c = 2048 / blocks;            // aim for ~2048 blocks in stage 1
    if (c > N / 4) c = N / 4;          // at least 4 images per chunk
    if (c > 64) c = 64;
    if (c < 1) c = 1;
    return (int)c;
}
extern "C" size_t dg_fm_workspace_bytes(int N, size_t J) {
    return dg_loss_workspace_bytes() + (size_t)2 * fm_chunks(N, J) * J * sizeof(float);
}
static int fm_fwd_run(int groups, const void* const* real, const void* const* fake, int io_bf16, int N, size_t J, float* const* diff,
                      float* const* loss, void* const* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_GROUP_TABLES_OK("dg_fm_fwd", real, fake, DG_TAB(diff), DG_TAB(loss), DG_TAB(ws));
    DG_CHECK_ARG(N > 0 && J > 0 && J % 4 == 0, "dg_fm_fwd: bad argument");
    if (ws_bytes < dg_fm_workspace_bytes(N, J)) return dg_fail(DG_ERR_WORKSPACE, "dg_fm_fwd: workspace too small");
    const int nch = fm_chunks(N, J);
    const void* parts[DG_MAX_GROUPS];
    for (int i = 0; i < groups; ++i) parts[i] = (const char*)ws[i] + dg_loss_workspace_bytes();
    const DgPtrs pparts = dg_ptrs(parts, groups), pdpart = dg_ptrs(DG_TAB(ws), groups);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid1((unsigned)((J / 4 + 255) / 256), nch, groups);
    if (io_bf16)
        hipLaunchKernelGGL(fm_partial_kernel<__bf16>, grid1, dim3(256), 0, st, dg_ptrs(real, groups), dg_ptrs(fake, groups), N, (long)J, nch, pparts);
    else
        hipLaunchKernelGGL(fm_partial_kernel<float>, grid1, dim3(256), 0, st, dg_ptrs(real, groups), dg_ptrs(fake, groups), N, (long)J, nch, pparts);
    DG_CHECK_LAUNCH("fm_partial");
    const int g = loss_grid((long)(J / 4));
    hipLaunchKernelGGL(fm_diff_kernel, dim3(g, groups), dim3(256), 0, st, pparts, N, (long)J, nch, dg_ptrs(DG_TAB(diff), groups), pdpart);
    DG_CHECK_LAUNCH("fm_diff");
    hipLaunchKernelGGL(final_sum_kernel, dim3(groups), dim3(256), 0, st, pdpart, g, 1.0 / (double)J, dg_ptrs(DG_TAB(loss), groups));
    DG_CHECK_LAUNCH("fm_final");
    return DG_OK;
}
extern "C" int dg_fm_fwd_t(const void* real, const void* fake, int io_bf16, int N, size_t J, float* diff, float* loss, void* ws,
                           size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(real && fake && diff && loss && N > 0 && J > 0 && J % 4 == 0, "dg_fm_fwd: bad argument");
    if (!ws || ws_bytes < dg_fm_workspace_bytes(N, J)) return dg_fail(DG_ERR_WORKSPACE, "dg_fm_fwd: workspace too small");
    return fm_fwd_run(1, &real, &fake, io_bf16, N, J, &diff, &loss, &ws, ws_bytes, stream);
}
extern "C" int dg_fm_fwd(const float* real, const float* fake, int N, size_t J, float* diff, float* loss, void* ws,
                         size_t ws_bytes, dg_stream_t stream) {
    return dg_fm_fwd_t(real, fake, 0, N, J, diff, loss, ws, ws_bytes, stream);
}
extern "C" int dg_fm_fwd_g(int groups, const float* const* real, const float* const* fake, int N, size_t J, float* const* diff, float* const* loss,
                           void* const* ws, size_t ws_bytes, dg_stream_t stream) {
    return fm_fwd_run(groups, DG_TAB(real), DG_TAB(fake), 0, N, J, diff, loss, ws, ws_bytes, stream);
}
static int fm_bwd_run(int groups, const float* const* diff, int N, size_t J, const float* const* gout, void* const* dreal, void* const* dfake,
                      int io_bf16, dg_stream_t stream) {
    DG_GROUP_TABLES_OK("dg_fm_bwd", DG_TAB(diff), DG_TAB(gout));
    DG_CHECK_ARG(N > 0 && J > 0 && J % 4 == 0, "dg_fm_bwd: bad argument");
    const long total = (long)(J / 4) * N;
    long g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    const dim3 grid((unsigned)g, groups);
    const DgPtrs pr = dg_ptrs(DG_TAB(dreal), groups), pf = dg_ptrs(DG_TAB(dfake), groups);
    if (io_bf16)
        hipLaunchKernelGGL(fm_bwd_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, dg_ptrs(DG_TAB(diff), groups), N, (long)J, dg_ptrs(DG_TAB(gout), groups), pr, pf);
    else
        hipLaunchKernelGGL(fm_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, dg_ptrs(DG_TAB(diff), groups), N, (long)J, dg_ptrs(DG_TAB(gout), groups), pr, pf);
    DG_CHECK_LAUNCH("fm_bwd");
    return DG_OK;
}
extern "C" int dg_fm_bwd_t(const float* diff, int N, size_t J, const float* gout, void* dreal, void* dfake, int io_bf16, dg_stream_t stream) {
    DG_CHECK_ARG(diff && gout && N > 0 && J > 0 && J % 4 == 0, "dg_fm_bwd: bad argument");
    return fm_bwd_run(1, &diff, N, J, &gout, &dreal, &dfake, io_bf16, stream);
}
extern "C" int dg_fm_bwd(const float* diff, int N, size_t J, const float* gout, float* dreal, float* dfake, dg_stream_t stream) {
    return dg_fm_bwd_t(diff, N, J, gout, dreal, dfake, 0, stream);
}
/* dreal / dfake tables may be NULL (that side needs no gradient); within a table every problem's entry is set or none is */
extern "C" int dg_fm_bwd_g(int groups, const float* const* diff, int N, size_t J, const float* const* gout, float* const* dreal, float* const* dfake,
                           dg_stream_t stream) {
    return fm_bwd_run(groups, diff, N, J, gout, (void* const*)dreal, (void* const*)dfake, 0, stream);
}

extern "C" int dg_loss_mix_fwd(const float* lossvec, float* out8, int nfm, float rate, int arch, dg_stream_t stream) {
    DG_CHECK_ARG(lossvec && out8 && nfm >= 0 && nfm <= 12 && arch >= 0 && arch <= 2, "dg_loss_mix_fwd: bad argument");
    hipLaunchKernelGGL(loss_mix_fwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, lossvec, out8, nfm, rate, arch);
    DG_CHECK_LAUNCH("loss_mix_fwd");
    return DG_OK;
}
extern "C" int dg_loss_mix_bwd(const float* gout, float* gvec, int nfm, float rate, int arch, int which, dg_stream_t stream) {
    DG_CHECK_ARG(gout && gvec && nfm >= 0 && nfm <= 12 && arch >= 0 && arch <= 2 && (which == 6 || which == 7), "dg_loss_mix_bwd: bad argument");
    hipLaunchKernelGGL(loss_mix_bwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, gout, gvec, nfm, rate, arch, which);
    DG_CHECK_LAUNCH("loss_mix_bwd");
    return DG_OK;
}
