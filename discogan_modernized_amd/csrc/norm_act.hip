// Training-mode BatchNorm2d + activation on NHWC [M][C] tensors, and stand-alone activations.
// Replaces nn.BatchNorm2d + in-place LeakyReLU(0.2)/ReLU (model.py:12-13,...,115-116) and their
// autograd backward; nn.Sigmoid / first-layer LeakyReLU backward (model.py:9,36).
// All kernels are HBM-bound streaming passes: 16-byte loads, lanes along the channel axis.
//
//   stats   : per-channel sum / sum-of-squares, fp32 per-thread partials over short row runs,
//             block partials to the workspace, finalised in fp64 (fixed order -> deterministic)
//   apply   : z = act((y - mean) * (gamma*invstd) + beta)
//   backward: g = dz * act'(u) with u recomputed bit-identically from y;
//             dbeta = sum g, dgamma = sum g*xhat, dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M)
#include "dg_common.h"

#define BN_U 8     // independent row loads in flight per thread (same-box A/B of the whole iteration: U=2 13.46 ms, 4 13.17, 8 13.13)

__device__ __forceinline__ float bn_norm(float y, float mean, float gs, float beta) { return fmaf(y - mean, gs, beta); }

// Reduction-pass geometry: a 256-thread block is TX float4 lanes along channels x TY = 256/TX row lanes, with
// TX = the power of two covering C/4 (capped at 64), so every thread is busy for C = 64 as for C = 512.
// grid = (cchunks, rchunks); each block walks rows r0 + ty + k*TY of its row chunk.
struct BnGrid { int tx, ty, cchunks, rchunks; };
static BnGrid bn_grid(int M, int C) {
    BnGrid g;
    int q = C / 4, tx = 1;
    while (tx < q && tx < 64) tx <<= 1;
    g.tx = tx;
    g.ty = 256 / tx;
    g.cchunks = (q + tx - 1) / tx;
    int rc = M / (g.ty * BN_U);            // at least one unrolled trip per row lane
    int cap = 2048 / g.cchunks;
    if (cap > 512) cap = 512;              // the finalize kernels walk this many partials per channel
    if (rc > cap) rc = cap;
    if (rc < 1) rc = 1;
    g.rchunks = rc;
    return g;
}

// part layout: [2][rchunks][C]  (0: sum of (y - shift), 1: sum of (y - shift)^2; shift = y[row 0])
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ y, float* __restrict__ part,
                                                               int M, int C, int rchunks, int TX) {
    __shared__ f32x4 red[2][256];
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * 4;
    const int rows_per = (M + rchunks - 1) / rchunks;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        // shifted sums: d = y - y[row 0]; var = E[d^2] - E[d]^2 has no catastrophic cancellation
        // because the shift is itself a sample of the channel (|mean - shift| ~ std).
        const f32x4 sh = *(const f32x4*)(y + c);
        const float* p = y + c;
        int r = r0 + ty;
        for (; r + (BN_U - 1) * TY < r1; r += BN_U * TY) {
            f32x4 v[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) v[u] = *(const f32x4*)(p + (long)(r + u * TY) * C);
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                const f32x4 d = v[u] - sh;
                s += d;
                q += d * d;
            }
        }
        for (; r < r1; r += TY) {
            const f32x4 d = *(const f32x4*)(p + (long)r * C) - sh;
            s += d;
            q += d * d;
        }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    for (int h = TY >> 1; h > 0; h >>= 1) {     // fixed-order tree over the row lanes
        if (ty < h) {
            red[0][threadIdx.x] += red[0][threadIdx.x + h * TX];
            red[1][threadIdx.x] += red[1][threadIdx.x + h * TX];
        }
        __syncthreads();
    }
    if (ty == 0 && c < C) {
        *(f32x4*)(part + (long)blockIdx.y * C + c) = red[0][tx];
        *(f32x4*)(part + ((long)rchunks + blockIdx.y) * C + c) = red[1][tx];
    }
}

// Finalize helper: a 256-thread block owns 8 channels; 32 lanes per channel walk the row-chunk
// partials in a fixed order and are combined in fp64 through LDS (deterministic).  Returns the
// channel for the lane that holds the totals, -1 for every other thread.
#define BN_FIN_CH 8
__device__ __forceinline__ int bn_reduce_partials(const float* __restrict__ part, int C, int rchunks, double* s_out, double* q_out) {
    __shared__ double red[2][32][BN_FIN_CH];
    const int cl = threadIdx.x % BN_FIN_CH, pl = threadIdx.x / BN_FIN_CH;
    const int c = blockIdx.x * BN_FIN_CH + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        int r = pl;
        for (; r + 96 < rchunks; r += 128) {       // 8 loads in flight, summed in row order
            float vs[4], vq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vs[u] = part[(long)(r + 32 * u) * C + c];
                vq[u] = part[((long)rchunks + r + 32 * u) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s += (double)vs[u];
                q += (double)vq[u];
            }
        }
        for (; r < rchunks; r += 32) {
            s += (double)part[(long)r * C + c];
            q += (double)part[((long)rchunks + r) * C + c];
        }
    }
    red[0][pl][cl] = s;
    red[1][pl][cl] = q;
    __syncthreads();
    if (pl != 0 || c >= C) return -1;
    for (int j = 1; j < 32; ++j) {
        s += red[0][j][cl];
        q += red[1][j][cl];
    }
    *s_out = s;
    *q_out = q;
    return c;
}

__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ y, const float* __restrict__ part, int M, int C, int rchunks,
                                                                float eps, float momentum, float* __restrict__ running_mean,
                                                                float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                                float* __restrict__ saved) {
    double s, q;
    const int c = bn_reduce_partials(part, C, rchunks, &s, &q);
    if (c < 0) return;
    if (c == 0 && nbt) nbt[0] += 1;
    const double dm = s / M;
    const double mean = (double)y[c] + dm;
    double var = q / M - dm * dm;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    saved[c] = (float)mean;
    saved[C + c] = invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

// Finalize from the conv kernels' partial rows [P][3C+4] = {count,-,-,-, shift[C], sum[C], sumsq[C]}.
// Every partial is re-referenced to ONE global shift G (the first row's shift, itself a sample of the
// channel):  sum(y-G) = s_p + n_p*d,  sum((y-G)^2) = q_p + 2*d*s_p + n_p*d^2  with d = shift_p - G, all in
// fp64 FMAs (no divisions in the loop); block = 4 channels x 64 lanes, lanes combined in a fixed order.
__global__ __launch_bounds__(256) void bn_partials_finalize_kernel(const float* __restrict__ stat, int P, int rs, int M, int C,
                                                                  float eps, float momentum, float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                                  float* __restrict__ saved) {
    __shared__ double red[2][64][4];
    const int cl = threadIdx.x & 3, pl = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + cl;
    double S = 0.0, Q = 0.0;
    double G = 0.0;
    if (c < C) {
        G = (double)stat[4 + c];                    // row 0 always has count > 0
        for (int p = pl; p < P; p += 64) {
            const float* row = stat + (long)p * rs;
            const double np = (double)row[0];
            const double s = (double)row[4 + C + c], q = (double)row[4 + 2 * C + c];
            const double d = (double)row[4 + c] - G;
            if (np > 0.0) {
                S += s + np * d;
                Q += q + d * (2.0 * s + np * d);
            }
        }
    }
    red[0][pl][cl] = S;
    red[1][pl][cl] = Q;
    __syncthreads();
    if (pl != 0 || c >= C) return;
    for (int j = 1; j < 64; ++j) {
        S += red[0][j][cl];
        Q += red[1][j][cl];
    }
    if (c == 0 && nbt) nbt[0] += 1;
    const double dm = S / M;
    const double mean = G + dm;
    double var = Q / M - dm * dm;
    if (var < 0.0) var = 0.0;
    saved[c] = (float)mean;
    saved[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

typedef __bf16 bf16x4_n __attribute__((ext_vector_type(4)));
// Z16: also write a bf16 (RNE) shadow of z for the bf16 matrix path (the next conv reads it instead of z)
template <bool Z16>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ y, float* __restrict__ z, long total4,
                                                         int C, const float* __restrict__ saved,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int act, float slope, __bf16* __restrict__ z16) {
    const int c4n = C >> 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % c4n) * 4;
        const f32x4 v = *(const f32x4*)(y + idx * 4);
        const f32x4 mean = *(const f32x4*)(saved + c), istd = *(const f32x4*)(saved + C + c);
        const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = dg_apply_act(bn_norm(v[j], mean[j], g[j] * istd[j], b[j]), act, slope);
        *(f32x4*)(z + idx * 4) = o;
        if (Z16) *(bf16x4_n*)(z16 + idx * 4) = __builtin_convertvector(o, bf16x4_n);
    }
}

__device__ __forceinline__ float act_grad(float u, int act, float slope) {
    // derivative taken from the sign of the activation input == sign of its output (in-place
    // semantics of the reference: leaky_relu_backward(result), threshold_backward(result))
    if (act == DG_ACT_LEAKY) return u > 0.f ? 1.f : slope;
    if (act == DG_ACT_RELU) return u > 0.f ? 1.f : 0.f;
    return 1.f;
}

// BatchNorm backward reductions and the final expression run in fp64, like PyTorch's CPU kernels
// (accscalar_t = double): dy is a small difference of large terms whenever the incoming gradient is
// nearly constant within a channel (saturated discriminator), and fp32 there costs percents.
typedef double f64x4 __attribute__((ext_vector_type(4)));
// part layout (fp64): [2][rchunks][C]  (0: sum g, 1: sum g*xhat)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                             double* __restrict__ part, int M, int C, int rchunks, int TX,
                                                             const float* __restrict__ saved, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int act, float slope) {
    __shared__ f64x4 red[2][256];
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * 4;
    const int rows_per = (M + rchunks - 1) / rchunks;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    f64x4 s = {0., 0., 0., 0.}, q = {0., 0., 0., 0.};
    if (c < C) {
        const f32x4 mean = *(const f32x4*)(saved + c), istd = *(const f32x4*)(saved + C + c);
        const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
        const f32x4 gs = g * istd;
        auto acc = [&](const f32x4& v, const f32x4& d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = bn_norm(v[j], mean[j], gs[j], b[j]);
                const double gg = (double)(d[j] * act_grad(u, act, slope));
                s[j] += gg;
                q[j] += gg * (((double)v[j] - (double)mean[j]) * (double)istd[j]);
            }
        };
        const float* py = y + c;
        const float* pd = dz + c;
        int r = r0 + ty;
        for (; r + (BN_U - 1) * TY < r1; r += BN_U * TY) {
            f32x4 v[BN_U], d[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                v[u] = *(const f32x4*)(py + (long)(r + u * TY) * C);
                d[u] = *(const f32x4*)(pd + (long)(r + u * TY) * C);
            }
#pragma unroll
            for (int u = 0; u < BN_U; ++u) acc(v[u], d[u]);
        }
        for (; r < r1; r += TY) acc(*(const f32x4*)(py + (long)r * C), *(const f32x4*)(pd + (long)r * C));
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    for (int h = TY >> 1; h > 0; h >>= 1) {
        if (ty < h) {
            red[0][threadIdx.x] += red[0][threadIdx.x + h * TX];
            red[1][threadIdx.x] += red[1][threadIdx.x + h * TX];
        }
        __syncthreads();
    }
    if (ty == 0 && c < C) {
        *(f64x4*)(part + (long)blockIdx.y * C + c) = red[0][tx];
        *(f64x4*)(part + ((long)rchunks + blockIdx.y) * C + c) = red[1][tx];
    }
}

// coef (fp64): [2][C] = dbeta/M, dgamma/M  (kept in the workspace after the partials)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ part, int M, int C, int rchunks,
                                                              double* __restrict__ coef, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
    __shared__ double red[2][32][BN_FIN_CH];
    const int cl = threadIdx.x % BN_FIN_CH, pl = threadIdx.x / BN_FIN_CH;
    const int c = blockIdx.x * BN_FIN_CH + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        int r = pl;
        for (; r + 96 < rchunks; r += 128) {       // 8 loads in flight, summed in row order
            double vs[4], vq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vs[u] = part[(long)(r + 32 * u) * C + c];
                vq[u] = part[((long)rchunks + r + 32 * u) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s += vs[u];
                q += vq[u];
            }
        }
        for (; r < rchunks; r += 32) {
            s += part[(long)r * C + c];
            q += part[((long)rchunks + r) * C + c];
        }
    }
    red[0][pl][cl] = s;
    red[1][pl][cl] = q;
    __syncthreads();
    if (pl != 0 || c >= C) return;
    for (int j = 1; j < 32; ++j) {
        s += red[0][j][cl];
        q += red[1][j][cl];
    }
    coef[c] = s / M;
    coef[C + c] = q / M;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)q;
}

template <bool D16>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                           float* __restrict__ dy, long total4, int C,
                                                           const float* __restrict__ saved, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const double* __restrict__ coef,
                                                           int act, float slope, __bf16* __restrict__ dy16) {
    const int c4n = C >> 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % c4n) * 4;
        const f32x4 v = *(const f32x4*)(y + idx * 4);
        const f32x4 d = *(const f32x4*)(dz + idx * 4);
        const f32x4 mean = *(const f32x4*)(saved + c), istd = *(const f32x4*)(saved + C + c);
        const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
        const f64x4 c1 = *(const f64x4*)(coef + c), c2 = *(const f64x4*)(coef + C + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gs = g[j] * istd[j];
            const float u = bn_norm(v[j], mean[j], gs, b[j]);
            const double gg = (double)(d[j] * act_grad(u, act, slope));
            const double xhat = ((double)v[j] - (double)mean[j]) * (double)istd[j];
            o[j] = (float)((double)g[j] * (double)istd[j] * (gg - c1[j] - xhat * c2[j]));
        }
        *(f32x4*)(dy + idx * 4) = o;
        if (D16) *(bf16x4_n*)(dy16 + idx * 4) = __builtin_convertvector(o, bf16x4_n);
    }
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long total4,
                                                      long n, int act, float slope) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        if (idx * 4 + 3 < n) {
            const f32x4 v = *(const f32x4*)(x + idx * 4);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = act == DG_ACT_SIGMOID ? 1.f / (1.f + expf(-v[j])) : dg_apply_act(v[j], act, slope);
            *(f32x4*)(y + idx * 4) = o;
        } else {
            for (long e = idx * 4; e < n; ++e)
                y[e] = act == DG_ACT_SIGMOID ? 1.f / (1.f + expf(-x[e])) : dg_apply_act(x[e], act, slope);
        }
    }
}

__device__ __forceinline__ float act_bwd_one(float dy, float out, int act, float slope) {
    if (act == DG_ACT_SIGMOID) return dy * (1.f - out) * out;  // sigmoid_backward: grad * (1 - y) * y
    if (act == DG_ACT_LEAKY) return out > 0.f ? dy : dy * slope;
    if (act == DG_ACT_RELU) return out > 0.f ? dy : 0.f;
    return dy;
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out,
                                                      float* __restrict__ dx, long total4, long n, int act, float slope) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        if (idx * 4 + 3 < n) {
            const f32x4 d = *(const f32x4*)(dy + idx * 4), o = *(const f32x4*)(out + idx * 4);
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = act_bwd_one(d[j], o[j], act, slope);
            *(f32x4*)(dx + idx * 4) = r;
        } else {
            for (long e = idx * 4; e < n; ++e) dx[e] = act_bwd_one(dy[e], out[e], act, slope);
        }
    }
}

static int stream_grid(long total4) {
    long g = (total4 + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" size_t dg_bn_workspace_bytes(int M, int C) {
    const BnGrid g = bn_grid(M, C);
    return ((size_t)2 * g.rchunks * C + 2 * (size_t)C) * sizeof(double);   // fp64 partials in the backward
}

extern "C" int dg_bn_train_stats(const float* y, int M, int C, float eps, float momentum, float* running_mean,
                                 float* running_var, int64_t* nbt, float* saved, void* ws, size_t ws_bytes,
                                 dg_stream_t stream) {
    DG_CHECK_ARG(y && saved, "dg_bn_train_stats: null pointer");
    DG_CHECK_ARG(M >= 2, "dg_bn_train_stats: Expected more than 1 value per channel when training (M=%d)", M);
    DG_CHECK_ARG(C >= 4 && C % 4 == 0, "dg_bn_train_stats: C=%d must be a multiple of 4", C);
    if (ws == nullptr || ws_bytes < dg_bn_workspace_bytes(M, C))
        return dg_fail(DG_ERR_WORKSPACE, "dg_bn_train_stats: workspace %zu < %zu", ws_bytes, dg_bn_workspace_bytes(M, C));
    const BnGrid g = bn_grid(M, C);
    const int cc = g.cchunks, rc = g.rchunks;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(cc, rc), dim3(256), 0, st, y, (float*)ws, M, C, rc, g.tx);
    DG_CHECK_LAUNCH("bn_stats_partial");
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + BN_FIN_CH - 1) / BN_FIN_CH), dim3(256), 0, st, y, (const float*)ws, M, C, rc, eps,
                       momentum, running_mean, running_var, nbt, saved);
    DG_CHECK_LAUNCH("bn_stats_finalize");
    return DG_OK;
}

extern "C" int dg_bn_stats_from_partials(const float* stat, int P, int M, int C, float eps, float momentum,
                                         float* running_mean, float* running_var, int64_t* nbt, float* saved,
                                         dg_stream_t stream) {
    DG_CHECK_ARG(stat && saved && P >= 1, "dg_bn_stats_from_partials: bad argument");
    DG_CHECK_ARG(M >= 2, "dg_bn_stats_from_partials: Expected more than 1 value per channel when training (M=%d)", M);
    DG_CHECK_ARG(C >= 4 && C % 4 == 0, "dg_bn_stats_from_partials: C=%d must be a multiple of 4", C);
    hipLaunchKernelGGL(bn_partials_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, stat, P, 3 * C + 4,
                       M, C, eps, momentum, running_mean, running_var, nbt, saved);
    DG_CHECK_LAUNCH("bn_partials_finalize");
    return DG_OK;
}

static int bn_act_fwd_impl(const float* y, float* z, void* z16, int M, int C, const float* saved, const float* gamma,
                           const float* beta, int act, float slope, dg_stream_t stream) {
    DG_CHECK_ARG(y && z && saved && gamma && beta, "dg_bn_act_fwd: null pointer");
    DG_CHECK_ARG(C >= 4 && C % 4 == 0, "dg_bn_act_fwd: C=%d must be a multiple of 4", C);
    DG_CHECK_ARG(act == DG_ACT_NONE || act == DG_ACT_LEAKY || act == DG_ACT_RELU, "dg_bn_act_fwd: bad act %d", act);
    const long total4 = (long)M * C / 4;
    if (z16)
        hipLaunchKernelGGL(bn_act_fwd_kernel<true>, dim3(stream_grid(total4)), dim3(256), 0, (hipStream_t)stream, y, z, total4, C,
                           saved, gamma, beta, act, slope, (__bf16*)z16);
    else
        hipLaunchKernelGGL(bn_act_fwd_kernel<false>, dim3(stream_grid(total4)), dim3(256), 0, (hipStream_t)stream, y, z, total4, C,
                           saved, gamma, beta, act, slope, (__bf16*)nullptr);
    DG_CHECK_LAUNCH("bn_act_fwd");
    return DG_OK;
}
extern "C" int dg_bn_act_fwd(const float* y, float* z, int M, int C, const float* saved, const float* gamma,
                             const float* beta, int act, float slope, dg_stream_t stream) {
    return bn_act_fwd_impl(y, z, nullptr, M, C, saved, gamma, beta, act, slope, stream);
}
extern "C" int dg_bn_act_fwd_bf16(const float* y, float* z, void* z_bf16, int M, int C, const float* saved, const float* gamma,
                                  const float* beta, int act, float slope, dg_stream_t stream) {
    DG_CHECK_ARG(z_bf16, "dg_bn_act_fwd_bf16: null shadow pointer");
    return bn_act_fwd_impl(y, z, z_bf16, M, C, saved, gamma, beta, act, slope, stream);
}

static int bn_act_bwd_impl(const float* dz, const float* y, float* dy, void* dy16, int M, int C, const float* saved,
                           const float* gamma, const float* beta, int act, float slope, float* dgamma, float* dbeta,
                           int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(dz && y && dy && saved && gamma && beta, "dg_bn_act_bwd: null pointer");
    DG_CHECK_ARG(C >= 4 && C % 4 == 0, "dg_bn_act_bwd: C=%d must be a multiple of 4", C);
    DG_CHECK_ARG(act == DG_ACT_NONE || act == DG_ACT_LEAKY || act == DG_ACT_RELU, "dg_bn_act_bwd: bad act %d", act);
    if (ws == nullptr || ws_bytes < dg_bn_workspace_bytes(M, C))
        return dg_fail(DG_ERR_WORKSPACE, "dg_bn_act_bwd: workspace %zu < %zu", ws_bytes, dg_bn_workspace_bytes(M, C));
    const BnGrid g = bn_grid(M, C);
    const int cc = g.cchunks, rc = g.rchunks;
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)ws;
    double* coef = part + (size_t)2 * rc * C;
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(cc, rc), dim3(256), 0, st, dz, y, part, M, C, rc, g.tx, saved, gamma, beta, act, slope);
    DG_CHECK_LAUNCH("bn_bwd_partial");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + BN_FIN_CH - 1) / BN_FIN_CH), dim3(256), 0, st, (const double*)part, M, C, rc, coef,
                       dgamma, dbeta, accumulate);
    DG_CHECK_LAUNCH("bn_bwd_finalize");
    const long total4 = (long)M * C / 4;
    if (dy16)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(stream_grid(total4)), dim3(256), 0, st, dz, y, dy, total4, C, saved, gamma,
                           beta, (const double*)coef, act, slope, (__bf16*)dy16);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(stream_grid(total4)), dim3(256), 0, st, dz, y, dy, total4, C, saved, gamma,
                           beta, (const double*)coef, act, slope, (__bf16*)nullptr);
    DG_CHECK_LAUNCH("bn_bwd_apply");
    return DG_OK;
}
extern "C" int dg_bn_act_bwd(const float* dz, const float* y, float* dy, int M, int C, const float* saved,
                             const float* gamma, const float* beta, int act, float slope, float* dgamma, float* dbeta,
                             int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return bn_act_bwd_impl(dz, y, dy, nullptr, M, C, saved, gamma, beta, act, slope, dgamma, dbeta, accumulate, ws, ws_bytes, stream);
}
extern "C" int dg_bn_act_bwd_bf16(const float* dz, const float* y, float* dy, void* dy_bf16, int M, int C, const float* saved,
                                  const float* gamma, const float* beta, int act, float slope, float* dgamma, float* dbeta,
                                  int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(dy_bf16, "dg_bn_act_bwd_bf16: null shadow pointer");
    return bn_act_bwd_impl(dz, y, dy, dy_bf16, M, C, saved, gamma, beta, act, slope, dgamma, dbeta, accumulate, ws, ws_bytes, stream);
}

extern "C" int dg_act_fwd(const float* x, float* y, size_t n, int act, float slope, dg_stream_t stream) {
    DG_CHECK_ARG(x && y, "dg_act_fwd: null pointer");
    DG_CHECK_ARG(act >= DG_ACT_NONE && act <= DG_ACT_SIGMOID, "dg_act_fwd: bad act %d", act);
    if (n == 0) return DG_OK;
    const long total4 = (long)((n + 3) / 4);
    hipLaunchKernelGGL(act_fwd_kernel, dim3(stream_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, y, total4, (long)n, act, slope);
    DG_CHECK_LAUNCH("act_fwd");
    return DG_OK;
}
extern "C" int dg_act_bwd(const float* dy, const float* out, float* dx, size_t n, int act, float slope, dg_stream_t stream) {
    DG_CHECK_ARG(dy && out && dx, "dg_act_bwd: null pointer");
    DG_CHECK_ARG(act >= DG_ACT_NONE && act <= DG_ACT_SIGMOID, "dg_act_bwd: bad act %d", act);
    if (n == 0) return DG_OK;
    const long total4 = (long)((n + 3) / 4);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_grid(total4)), dim3(256), 0, (hipStream_t)stream, dy, out, dx, total4, (long)n, act, slope);
    DG_CHECK_LAUNCH("act_bwd");
    return DG_OK;
}
