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
#include <string.h>

#define BN_U 8     // independent row loads in flight per thread (same-box A/B of the whole iteration: U=2 13.46 ms, 4 13.17, 8 13.13)

// Activation tensors are fp32 or bf16 (bf16 activation storage of the bf16 matrix path: statistics, normalisation and the
// backward expression stay fp32 / fp64; only what is STORED is rounded, RNE).  Every thread moves 16 bytes per access:
// V = 4 fp32 or 8 bf16 channels.
typedef __bf16 bf16x8_n __attribute__((ext_vector_type(8)));
typedef float f32x8_n __attribute__((ext_vector_type(8)));
template <typename T> struct BnV;
template <> struct BnV<float> { static constexpr int V = 4; };
template <> struct BnV<__bf16> { static constexpr int V = 8; };
__device__ __forceinline__ void bn_ld(const float* p, float (&v)[4]) {
    const f32x4 t = *(const f32x4*)p;
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
__device__ __forceinline__ void bn_ld(const __bf16* p, float (&v)[8]) {
    const f32x8_n t = __builtin_convertvector(*(const bf16x8_n*)p, f32x8_n);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = t[j];
}
// per-channel fp32 parameters (mean, invstd, gamma, beta): V consecutive channels as 16-byte loads
template <int V>
__device__ __forceinline__ void bn_ldp(const float* p, float (&v)[V]) {
#pragma unroll
    for (int h = 0; h < V / 4; ++h) {
        const f32x4 t = *(const f32x4*)(p + 4 * h);
        v[4 * h] = t[0]; v[4 * h + 1] = t[1]; v[4 * h + 2] = t[2]; v[4 * h + 3] = t[3];
    }
}
__device__ __forceinline__ void bn_st(float* p, const float (&v)[4]) { *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ void bn_st(__bf16* p, const float (&v)[8]) {
    f32x8_n t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = v[j];
    *(bf16x8_n*)p = __builtin_convertvector(t, bf16x8_n);
}

template <int V>
__device__ __forceinline__ void bn_stp(float* p, const float (&v)[V]) {
#pragma unroll
    for (int h = 0; h < V / 4; ++h) *(f32x4*)(p + 4 * h) = (f32x4){v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
}

__device__ __forceinline__ float bn_norm(float y, float mean, float gs, float beta) { return fmaf(y - mean, gs, beta); }

// Grouped launches (dg_bn_*_g; round 4): the problems of one launch -- the same BatchNorm layer of the A-side and the B-side network, a
// discriminator layer's real and fake pass (image_translation.py:342-361) -- share M, C and every launch parameter; each has its own
// tensors.  A block index picks the problem (wave-uniform: scalar loads from the kernel arguments); the kernel bodies are the
// one-problem bodies, so a problem's result is bitwise what its own launch computes.
struct BnProb {
    const void* y;        // conv output (statistics, apply, backward)
    const void* dz;       // backward: gradient of the layer's output
    void* out;            // apply: z; backward apply: dy
    void* out16;          // bf16 shadow / plane triple of `out` (one-problem forms only)
    void* ws;             // this problem's workspace: partials [+ backward coefficients]
    float* saved;         // [2][C] mean, invstd
    const float* gamma;
    const float* beta;
    float* rmean;
    float* rvar;
    int64_t* nbt;
    float* dgamma;
    float* dbeta;
};
struct BnGroup {
    BnProb p[DG_MAX_GROUPS];
};

// Reduction-pass geometry: a 256-thread block is TX float4 lanes along channels x TY = 256/TX row lanes, with
// TX = the power of two covering C/4 (capped at 64), so every thread is busy for C = 64 as for C = 512.
// grid = (cchunks, rchunks); each block walks rows r0 + ty + k*TY of its row chunk.
struct BnGrid { int tx, ty, cchunks, rchunks; };
static BnGrid bn_grid(int M, int C, int V = 4) {
    BnGrid g;
    int q = C / V, tx = 1;
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
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const BnGroup G, int M, int C, int rchunks, int TX) {
    constexpr int V = BnV<T>::V;
    const T* __restrict__ y = (const T*)G.p[blockIdx.z].y;
    float* __restrict__ part = (float*)G.p[blockIdx.z].ws;
    __shared__ float red[2][256][V];
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * V;
    const int rows_per = (M + rchunks - 1) / rchunks;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float s[V], q[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = q[j] = 0.f;
    if (c < C) {
        // shifted sums: d = y - y[row 0]; var = E[d^2] - E[d]^2 has no catastrophic cancellation
        // because the shift is itself a sample of the channel (|mean - shift| ~ std).
        float sh[V];
        bn_ld(y + c, sh);
        const T* p = y + c;
        int r = r0 + ty;
        for (; r + (BN_U - 1) * TY < r1; r += BN_U * TY) {
            float v[BN_U][V];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) bn_ld(p + (long)(r + u * TY) * C, v[u]);
#pragma unroll
            for (int u = 0; u < BN_U; ++u)
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float d = v[u][j] - sh[j];
                    s[j] += d;
                    q[j] += d * d;
                }
        }
        for (; r < r1; r += TY) {
            float v[V];
            bn_ld(p + (long)r * C, v);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float d = v[j] - sh[j];
                s[j] += d;
                q[j] += d * d;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        red[0][threadIdx.x][j] = s[j];
        red[1][threadIdx.x][j] = q[j];
    }
    __syncthreads();
    for (int h = TY >> 1; h > 0; h >>= 1) {     // fixed-order tree over the row lanes
        if (ty < h) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                red[0][threadIdx.x][j] += red[0][threadIdx.x + h * TX][j];
                red[1][threadIdx.x][j] += red[1][threadIdx.x + h * TX][j];
            }
        }
        __syncthreads();
    }
    if (ty == 0 && c < C) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            part[(long)blockIdx.y * C + c + j] = red[0][tx][j];
            part[((long)rchunks + blockIdx.y) * C + c + j] = red[1][tx][j];
        }
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

// blockIdx.y = output set; share > 1: `share` consecutive problems are passes through the SAME BatchNorm module (a discriminator's real
// and fake pass): their running-statistics updates are applied one after the other in problem order by the thread that owns the
// channel -- what `share` consecutive launches leave.
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const BnGroup G, int share, int M, int C, int rchunks, float eps, float momentum) {
    for (int j = 0; j < share; ++j) {
        const BnProb& P = G.p[blockIdx.y * share + j];
        const T* __restrict__ y = (const T*)P.y;
        float* __restrict__ running_mean = P.rmean;
        float* __restrict__ running_var = P.rvar;
        int64_t* __restrict__ nbt = P.nbt;
        float* __restrict__ saved = P.saved;
        double s, q;
        const int c = bn_reduce_partials((const float*)P.ws, C, rchunks, &s, &q);
        if (c >= 0) {
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
        if (j + 1 < share) __syncthreads();       // the LDS rows of bn_reduce_partials are reused by the next problem
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
    // block = 4 channels (one 16-byte load per row and array), 256 row lanes; fixed-order tree reduction in fp64.
    // (Round 3: the first form gave every thread ONE channel and 64 row lanes -- 4-byte loads, P / 64 iterations -- and cost 41 us
    // per call on the 1024..4096 partial rows the plane kernels emit, more than the statistics pass it replaces.)
    __shared__ double red[256][8];
    const int t = threadIdx.x;
    const int c4 = blockIdx.x * 4;
    double S[4] = {0.0, 0.0, 0.0, 0.0}, Q[4] = {0.0, 0.0, 0.0, 0.0};
    const f32x4 G4 = *(const f32x4*)(stat + 4 + c4);                    // row 0 always has count > 0: the global shift
    for (int p = t; p < P; p += 256) {
        const float* row = stat + (long)p * rs;
        const double np = (double)row[0];
        if (np > 0.0) {
            const f32x4 sh = *(const f32x4*)(row + 4 + c4), s4 = *(const f32x4*)(row + 4 + C + c4), q4 = *(const f32x4*)(row + 4 + 2 * C + c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double d = (double)sh[j] - (double)G4[j], s = (double)s4[j];
                S[j] += s + np * d;
                Q[j] += (double)q4[j] + d * (2.0 * s + np * d);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[t][j] = S[j];
        red[t][4 + j] = Q[j];
    }
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if (t < half) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[t][j] += red[t + half][j];
        }
        __syncthreads();
    }
    if (t >= 4) return;
    const int c = c4 + t;
    if (c >= C) return;
    if (c == 0 && nbt) nbt[0] += 1;
    const double dm = red[0][t] / M;
    const double mean = (double)G4[t] + dm;
    double var = red[0][4 + t] / M - dm * dm;
    if (var < 0.0) var = 0.0;
    saved[c] = (float)mean;
    saved[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

// Two-level form of the same merge for many partial rows.  Level 1: block (channel quad, row block rb of RB) takes rows
// rb * 256 + t, + RB * 256, ... re-referenced to the global shift like the one-launch form and leaves its eight fp64 sums in
// part[rb][0 = S, 1 = Q][C4].  Level 2: one thread per channel adds the RB rows in order and finishes.
__global__ __launch_bounds__(256) void bn_partials_reduce_kernel(const float* __restrict__ stat, int P, int rs, int C, int RB, double* __restrict__ part) {
    __shared__ double red[256][8];
    const int t = threadIdx.x, rb = blockIdx.y;
    const int c4 = blockIdx.x * 4;
    const int C4 = (C + 3) / 4 * 4;
    double S[4] = {0.0, 0.0, 0.0, 0.0}, Q[4] = {0.0, 0.0, 0.0, 0.0};
    const f32x4 G4 = *(const f32x4*)(stat + 4 + c4);
    for (int p = rb * 256 + t; p < P; p += RB * 256) {
        const float* row = stat + (long)p * rs;
        const double np = (double)row[0];
        if (np > 0.0) {
            const f32x4 sh = *(const f32x4*)(row + 4 + c4), s4 = *(const f32x4*)(row + 4 + C + c4), q4 = *(const f32x4*)(row + 4 + 2 * C + c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double d = (double)sh[j] - (double)G4[j], s = (double)s4[j];
                S[j] += s + np * d;
                Q[j] += (double)q4[j] + d * (2.0 * s + np * d);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[t][j] = S[j];
        red[t][4 + j] = Q[j];
    }
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if (t < half) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[t][j] += red[t + half][j];
        }
        __syncthreads();
    }
    if (t < 8) part[((long)rb * 2 + (t >> 2)) * C4 + c4 + (t & 3)] = red[0][t];
}
__global__ __launch_bounds__(64) void bn_partials_merge_kernel(const double* __restrict__ part, int RB, const float* __restrict__ stat, int M, int C,
                                                               float eps, float momentum, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                               float* __restrict__ saved) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const int C4 = (C + 3) / 4 * 4;
    double S = 0.0, Q = 0.0;
    for (int rb = 0; rb < RB; ++rb) {
        S += part[((long)rb * 2 + 0) * C4 + c];
        Q += part[((long)rb * 2 + 1) * C4 + c];
    }
    if (c == 0 && nbt) nbt[0] += 1;
    const double dm = S / M;
    const double mean = (double)stat[4 + c] + dm;
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
// TI / TO: element types of y and z.  Z16 (fp32 z only) = 1: also write a bf16 (RNE) shadow of z for the bf16 matrix path;
// = 3: the three bf16 planes of z for the f32x3 matrix path (dg_split3; planes `pstride` elements apart).  cm != 0: the planes
// are written in the QUAD-CHUNK layout [M / 4][C / 16][4 pixels][16 channels] instead of pixel-major [M][C] -- the layout the
// window input-grad kernel (igemm_dma_x3_dgw.hip) wants: the 16-channel chunk of 4 consecutive pixels is one 128-byte line, so
// a window row of a chunk uses every byte of the lines it touches, while a 4-pixel block of all channels stays one contiguous
// run (the weight-gradient kernel's 16-pixel tile is as contiguous as before).  The work items are then dealt so that 16
// consecutive lanes hold (4 pixels) x (the 4 channel quads of one chunk): a wave reads 4 x 256 contiguous bytes of the fp32
// tensors and writes whole 128-byte lines of every plane.
struct BnItem { long v; int c; long pi; };      // fp32 vector index (4 channels), first channel, plane element index
__device__ __forceinline__ BnItem bn_item(long i, int cvn, int cm) {
    BnItem it;
    if (!cm) {
        it.v = i;
        it.c = (int)(i % cvn) * 4;
        it.pi = i * 4;
        return it;
    }
    // i = ((grp * (cvn / 16) + h) * 4 + pp) * 16 + q16: pixel 4 grp + pp, channel quad 16 h + q16
    const int q16 = (int)(i & 15), pp = (int)((i >> 4) & 3);
    const long rest = i >> 6;
    const int hs = cvn >> 4;
    const int h = (int)(rest % hs);
    const long grp = rest / hs;
    const int cq = h * 16 + q16;
    it.v = (grp * 4 + pp) * cvn + cq;
    it.c = cq * 4;
    it.pi = ((grp * (cvn >> 2) + (cq >> 2)) * 4 + pp) * 16 + (cq & 3) * 4;
    return it;
}
template <typename TI, typename TO, int Z16>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const BnGroup G, long totalv, int C, int act, float slope, long pstride, int cm) {
    constexpr int V = BnV<TI>::V;
    const BnProb& P = G.p[blockIdx.y];
    const TI* __restrict__ y = (const TI*)P.y;
    TO* __restrict__ z = (TO*)P.out;
    const float* __restrict__ saved = P.saved;
    const float* __restrict__ gamma = P.gamma;
    const float* __restrict__ beta = P.beta;
    __bf16* __restrict__ z16 = (__bf16*)P.out16;
    static_assert(BnV<TI>::V == BnV<TO>::V, "same storage type on both sides");
    const int cvn = C / V;
    for (long lin = (long)blockIdx.x * 256 + threadIdx.x; lin < totalv; lin += (long)gridDim.x * 256) {
        long idx = lin, pi = lin * 4;
        int c = (int)(lin % cvn) * V;
        if constexpr (Z16 == 3) {
            const BnItem it = bn_item(lin, cvn, cm);
            idx = it.v; c = it.c; pi = it.pi;
        }
        float v[V], o[V], mean[V], istd[V], g[V], b[V];
        bn_ld(y + idx * V, v);
        bn_ldp<V>(saved + c, mean);
        bn_ldp<V>(saved + C + c, istd);
        bn_ldp<V>(gamma + c, g);
        bn_ldp<V>(beta + c, b);
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = dg_apply_act(bn_norm(v[j], mean[j], g[j] * istd[j], b[j]), act, slope);
        if (Z16 != 3 || z != nullptr) bn_st(z + idx * V, o);      // plane-only output (f32x3 path): the fp32 copy has no reader
        if constexpr (Z16 == 1) *(bf16x4_n*)(z16 + idx * 4) = __builtin_convertvector((f32x4){o[0], o[1], o[2], o[3]}, bf16x4_n);
        if constexpr (Z16 == 3) {
            dg_bf16x4_t h, md, l;
            dg_split3((f32x4){o[0], o[1], o[2], o[3]}, h, md, l);
            *(dg_bf16x4_t*)(z16 + pi) = h;
            *(dg_bf16x4_t*)(z16 + pstride + pi) = md;
            *(dg_bf16x4_t*)(z16 + 2 * pstride + pi) = l;
        }
    }
}

// fp32 storage in the reduction passes' geometry (round 4): fixed channels per thread -- mean / invstd / gamma / beta loaded ONCE instead
// of four 16-byte loads per item, no 64-bit index arithmetic per item --, BN_U rows in flight.  The per-element arithmetic is the
// item kernel's above, expression for expression: results are bitwise the same.  With the plane outputs of the f32x3 path the item
// kernel ran at 3.5 TB/s of its 10 B per element (five load and three store instructions per item) against 4.9 for the plain split
// kernel; this form: see DESIGN.md 3.3.  Plane element index of (row, c): pixel-major row * C + c; quad-chunk
// [M / 4][C / 16][4 pixels][16 channels].  For the quad-chunk layout the host picks TX = 16, so a wave is 4 pixels x 64 channels: whole
// 128-byte lines of every plane (rows_per is a multiple of 4).
__device__ __forceinline__ long bn_plane_index(int row, int c, int C, int cm) {
    if (!cm) return (long)row * C + c;
    return ((((long)(row >> 2)) * (C >> 4) + (c >> 4)) * 4 + (row & 3)) * 16 + (c & 15);
}
typedef __bf16 bf16x8_p __attribute__((ext_vector_type(8)));
// stores the three planes of V (4 | 8) consecutive channels: one 8- or 16-byte store per plane
template <int V>
__device__ __forceinline__ void bn_put_planes(__bf16* p3, long pstride, long pi, const float (&o)[V]) {
    dg_bf16x4_t h[V / 4], md[V / 4], l[V / 4];
#pragma unroll
    for (int q = 0; q < V / 4; ++q) dg_split3((f32x4){o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]}, h[q], md[q], l[q]);
    if constexpr (V == 4) {
        *(dg_bf16x4_t*)(p3 + pi) = h[0];
        *(dg_bf16x4_t*)(p3 + pstride + pi) = md[0];
        *(dg_bf16x4_t*)(p3 + 2 * pstride + pi) = l[0];
    } else {
        *(bf16x8_p*)(p3 + pi) = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
        *(bf16x8_p*)(p3 + pstride + pi) = __builtin_shufflevector(md[0], md[1], 0, 1, 2, 3, 4, 5, 6, 7);
        *(bf16x8_p*)(p3 + 2 * pstride + pi) = __builtin_shufflevector(l[0], l[1], 0, 1, 2, 3, 4, 5, 6, 7);
    }
}
// V = channels per thread: 4, or 8 with plane outputs (two 16-byte loads, ONE 16-byte store per plane: the plane stores of the V = 4
// form were 8-byte stores, three per item)
template <int Z16, int V>
__global__ __launch_bounds__(256) void bn_act_fwd_rows_kernel(const BnGroup G, int M, int C, int rows_per, int TX, int act, float slope, long pstride, int cm) {
    constexpr int U = V == 4 ? BN_U : BN_U / 2;
    const BnProb& P = G.p[blockIdx.z];
    const float* __restrict__ y = (const float*)P.y;
    float* __restrict__ z = (float*)P.out;
    __bf16* __restrict__ z16 = (__bf16*)P.out16;
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * V;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float mean[V], gs[V], b[V];
    bn_ldp<V>(P.saved + c, mean);
    bn_ldp<V>(P.saved + C + c, b);         // invstd for now
    bn_ldp<V>(P.gamma + c, gs);
#pragma unroll
    for (int j = 0; j < V; ++j) gs[j] = gs[j] * b[j];
    bn_ldp<V>(P.beta + c, b);
    auto put = [&](int row, const float (&v)[V]) {
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = dg_apply_act(bn_norm(v[j], mean[j], gs[j], b[j]), act, slope);
        if (Z16 != 3 || z != nullptr) bn_stp<V>(z + (long)row * C + c, o);      // plane-only output (f32x3 path): the fp32 copy has no reader
        if constexpr (Z16 == 1) *(bf16x4_n*)(z16 + (long)row * C + c) = __builtin_convertvector((f32x4){o[0], o[1], o[2], o[3]}, bf16x4_n);
        if constexpr (Z16 == 3) bn_put_planes<V>(z16, pstride, bn_plane_index(row, c, C, cm), o);
    };
    const float* py = y + c;
    int r = r0 + ty;
    for (; r + (U - 1) * TY < r1; r += U * TY) {
        float v[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) bn_ldp<V>(py + (long)(r + u * TY) * C, v[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) put(r + u * TY, v[u]);
    }
    for (; r < r1; r += TY) {
        float v[V];
        bn_ldp<V>(py + (long)r * C, v);
        put(r, v);
    }
}

// bf16 storage, forward apply with the reduction passes' geometry: fixed channels per thread (parameters loaded once, no
// 64-bit modulo per access), U rows in flight
__global__ __launch_bounds__(256) void bn_act_fwd16_kernel(const BnGroup G, int M, int C, int rchunks, int TX, int act, float slope) {
    constexpr int V = 8, U = 4;
    const BnProb& P = G.p[blockIdx.z];
    const __bf16* __restrict__ y = (const __bf16*)P.y;
    __bf16* __restrict__ z = (__bf16*)P.out;
    const float* __restrict__ saved = P.saved;
    const float* __restrict__ gamma = P.gamma;
    const float* __restrict__ beta = P.beta;
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * V;
    if (c >= C) return;
    const int rows_per = (M + rchunks - 1) / rchunks;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float mean[V], gs[V], b[V];
    bn_ldp<V>(saved + c, mean);
    bn_ldp<V>(saved + C + c, b);
    bn_ldp<V>(gamma + c, gs);
#pragma unroll
    for (int j = 0; j < V; ++j) gs[j] *= b[j];
    bn_ldp<V>(beta + c, b);
    const __bf16* py = y + c;
    __bf16* pz = z + c;
    int r = r0 + ty;
    for (; r + (U - 1) * TY < r1; r += U * TY) {
        float v[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) bn_ld(py + (long)(r + u * TY) * C, v[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < V; ++j) v[u][j] = dg_apply_act(bn_norm(v[u][j], mean[j], gs[j], b[j]), act, slope);
            bn_st(pz + (long)(r + u * TY) * C, v[u]);
        }
    }
    for (; r < r1; r += TY) {
        float v[V];
        bn_ld(py + (long)r * C, v);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = dg_apply_act(bn_norm(v[j], mean[j], gs[j], b[j]), act, slope);
        bn_st(pz + (long)r * C, v);
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
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const BnGroup G, int M, int C, int rchunks, int TX, int act, float slope) {
    constexpr int V = BnV<T>::V;
    const BnProb& P = G.p[blockIdx.z];
    const T* __restrict__ dz = (const T*)P.dz;
    const T* __restrict__ y = (const T*)P.y;
    double* __restrict__ part = (double*)P.ws;
    const float* __restrict__ saved = P.saved;
    const float* __restrict__ gamma = P.gamma;
    const float* __restrict__ beta = P.beta;
    __shared__ double red[2][256][V];
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * V;
    const int rows_per = (M + rchunks - 1) / rchunks;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    double s[V], q[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = q[j] = 0.;
    if (c < C) {
        float mean[V], istd[V], gs[V], b[V];
        bn_ldp<V>(saved + c, mean);
        bn_ldp<V>(saved + C + c, istd);
        bn_ldp<V>(gamma + c, gs);
        bn_ldp<V>(beta + c, b);
#pragma unroll
        for (int j = 0; j < V; ++j) gs[j] *= istd[j];
        auto acc = [&](const float (&v)[V], const float (&d)[V]) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float u = bn_norm(v[j], mean[j], gs[j], b[j]);
                const double gg = (double)(d[j] * act_grad(u, act, slope));
                s[j] += gg;
                q[j] += gg * (((double)v[j] - (double)mean[j]) * (double)istd[j]);
            }
        };
        const T* py = y + c;
        const T* pd = dz + c;
        constexpr int U = V == 4 ? BN_U : BN_U / 2;       // same bytes in flight per thread
        int r = r0 + ty;
        for (; r + (U - 1) * TY < r1; r += U * TY) {
            float v[U][V], d[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                bn_ld(py + (long)(r + u * TY) * C, v[u]);
                bn_ld(pd + (long)(r + u * TY) * C, d[u]);
            }
            if constexpr (V == 8) {
                // bf16 storage: the inputs carry 8 significant bits, so the U terms of a batch are summed in fp32 and only the
                // batch sums go into the fp64 accumulators (the per-element fp64 chain made this pass compute-bound)
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    float ls = 0.f, lq = 0.f;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float um = v[u][j] - mean[j];
                        const float gg = d[u][j] * act_grad(fmaf(um, gs[j], b[j]), act, slope);
                        ls += gg;
                        lq = fmaf(gg, um * istd[j], lq);
                    }
                    s[j] += (double)ls;
                    q[j] += (double)lq;
                }
            } else {
#pragma unroll
            for (int u = 0; u < U; ++u) acc(v[u], d[u]);
            }
        }
        for (; r < r1; r += TY) {
            float v[V], d[V];
            bn_ld(py + (long)r * C, v);
            bn_ld(pd + (long)r * C, d);
            acc(v, d);
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        red[0][threadIdx.x][j] = s[j];
        red[1][threadIdx.x][j] = q[j];
    }
    __syncthreads();
    for (int h = TY >> 1; h > 0; h >>= 1) {
        if (ty < h) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                red[0][threadIdx.x][j] += red[0][threadIdx.x + h * TX][j];
                red[1][threadIdx.x][j] += red[1][threadIdx.x + h * TX][j];
            }
        }
        __syncthreads();
    }
    if (ty == 0 && c < C) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            part[(long)blockIdx.y * C + c + j] = red[0][tx][j];
            part[((long)rchunks + blockIdx.y) * C + c + j] = red[1][tx][j];
        }
    }
}

// coef (fp64): [2][C] = dbeta/M, dgamma/M  (kept in the workspace after the partials)
// blockIdx.y = output set; share > 1: `share` consecutive problems went through the SAME BatchNorm module (a discriminator's real and
// fake pass) and accumulate into the same dgamma / dbeta: added one after the other in problem order, as consecutive launches do.
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const BnGroup G, int share, int M, int C, int rchunks, int accumulate) {
    __shared__ double red[2][32][BN_FIN_CH];
    const int cl = threadIdx.x % BN_FIN_CH, pl = threadIdx.x / BN_FIN_CH;
    const int c = blockIdx.x * BN_FIN_CH + cl;
    for (int j = 0; j < share; ++j) {
        const BnProb& P = G.p[blockIdx.y * share + j];
        const double* __restrict__ part = (const double*)P.ws;
        double* __restrict__ coef = (double*)P.ws + (size_t)2 * rchunks * C;
        float* __restrict__ dgamma = P.dgamma;
        float* __restrict__ dbeta = P.dbeta;
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
        if (j > 0) __syncthreads();
        red[0][pl][cl] = s;
        red[1][pl][cl] = q;
        __syncthreads();
        if (pl == 0 && c < C) {
            for (int k = 1; k < 32; ++k) {
                s += red[0][k][cl];
                q += red[1][k][cl];
            }
            coef[c] = s / M;
            coef[C + c] = q / M;
            const int acc = accumulate || j > 0;
            if (dbeta) dbeta[c] = (acc ? dbeta[c] : 0.f) + (float)s;
            if (dgamma) dgamma[c] = (acc ? dgamma[c] : 0.f) + (float)q;
        }
    }
}

// bf16 storage: dz, y in, dy out, all bf16.  The reduction geometry of the partial pass (fixed channels per thread, TY row
// lanes, U rows in flight), so the per-channel constants are loaded ONCE per thread; fp32 arithmetic in the subtraction-first
// form  dy = gs * ((g - c1) - (y - mean) * istd * c2)  -- the inputs carry 8 significant bits and dy is rounded to 8, which
// is what bounds the result, not the fp32 evaluation (the fp32-storage kernel below keeps fp64: there the fp32 chain would be
// the largest error).  Measured: the fp64 form ran at the SAME element rate as the fp32-storage kernel, i.e. compute-bound.
__global__ __launch_bounds__(256) void bn_bwd_apply16_kernel(const BnGroup G, int M, int C, int rchunks, int TX, int coef_rchunks, int act, float slope) {
    constexpr int V = 8, U = 4;
    const BnProb& P = G.p[blockIdx.z];
    const __bf16* __restrict__ dz = (const __bf16*)P.dz;
    const __bf16* __restrict__ y = (const __bf16*)P.y;
    __bf16* __restrict__ dy = (__bf16*)P.out;
    const float* __restrict__ saved = P.saved;
    const float* __restrict__ gamma = P.gamma;
    const float* __restrict__ beta = P.beta;
    const double* __restrict__ coef = (const double*)P.ws + (size_t)2 * coef_rchunks * C;
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * V;
    if (c >= C) return;
    const int rows_per = (M + rchunks - 1) / rchunks;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float mean[V], gs[V], b[V], c1[V], k2[V];
    bn_ldp<V>(saved + c, mean);
    bn_ldp<V>(saved + C + c, k2);          // invstd for now
    bn_ldp<V>(gamma + c, gs);
    bn_ldp<V>(beta + c, b);
#pragma unroll
    for (int j = 0; j < V; ++j) {
        gs[j] *= k2[j];
        c1[j] = (float)coef[c + j];
        k2[j] = (float)((double)gs[j] * (double)k2[j] * coef[C + c + j]);      // gs * invstd * c2
    }
    auto one = [&](const float (&v)[V], const float (&d)[V], float (&o)[V]) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float um = v[j] - mean[j];
            const float gg = d[j] * act_grad(fmaf(um, gs[j], b[j]), act, slope);
            o[j] = fmaf(-um, k2[j], gs[j] * (gg - c1[j]));
        }
    };
    const __bf16* py = y + c;
    const __bf16* pd = dz + c;
    __bf16* po = dy + c;
    int r = r0 + ty;
    for (; r + (U - 1) * TY < r1; r += U * TY) {
        float v[U][V], d[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bn_ld(py + (long)(r + u * TY) * C, v[u]);
            bn_ld(pd + (long)(r + u * TY) * C, d[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float o[V];
            one(v[u], d[u], o);
            bn_st(po + (long)(r + u * TY) * C, o);
        }
    }
    for (; r < r1; r += TY) {
        float v[V], d[V], o[V];
        bn_ld(py + (long)r * C, v);
        bn_ld(pd + (long)r * C, d);
        one(v, d, o);
        bn_st(po + (long)r * C, o);
    }
}

// T: element type of dz / y / dy.  D16 (fp32 only) = 1: also write a bf16 shadow of dy; = 3: its three bf16 planes
template <typename T, int D16>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnGroup G, long totalv, int C, int coef_rchunks, int act, float slope, long pstride, int cm) {
    constexpr int V = BnV<T>::V;
    const BnProb& P = G.p[blockIdx.y];
    const T* __restrict__ dz = (const T*)P.dz;
    const T* __restrict__ y = (const T*)P.y;
    T* __restrict__ dy = (T*)P.out;
    const float* __restrict__ saved = P.saved;
    const float* __restrict__ gamma = P.gamma;
    const float* __restrict__ beta = P.beta;
    const double* __restrict__ coef = (const double*)P.ws + (size_t)2 * coef_rchunks * C;
    __bf16* __restrict__ dy16 = (__bf16*)P.out16;
    const int cvn = C / V;
    for (long lin = (long)blockIdx.x * 256 + threadIdx.x; lin < totalv; lin += (long)gridDim.x * 256) {
        long idx = lin, pi = lin * 4;
        int c = (int)(lin % cvn) * V;
        if constexpr (D16 == 3) {
            const BnItem it = bn_item(lin, cvn, cm);
            idx = it.v; c = it.c; pi = it.pi;
        }
        float v[V], d[V], o[V], mean[V], istd[V], g[V], b[V];
        bn_ld(y + idx * V, v);
        bn_ld(dz + idx * V, d);
        bn_ldp<V>(saved + c, mean);
        bn_ldp<V>(saved + C + c, istd);
        bn_ldp<V>(gamma + c, g);
        bn_ldp<V>(beta + c, b);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float gs = g[j] * istd[j];
            const float u = bn_norm(v[j], mean[j], gs, b[j]);
            const double gg = (double)(d[j] * act_grad(u, act, slope));
            const double xhat = ((double)v[j] - (double)mean[j]) * (double)istd[j];
            o[j] = (float)((double)g[j] * (double)istd[j] * (gg - coef[c + j] - xhat * coef[C + c + j]));
        }
        if (D16 != 3 || dy != nullptr) bn_st(dy + idx * V, o);   // plane-only output (f32x3 path): the fp32 copy has no reader
        if constexpr (D16 == 1) *(bf16x4_n*)(dy16 + idx * 4) = __builtin_convertvector((f32x4){o[0], o[1], o[2], o[3]}, bf16x4_n);
        if constexpr (D16 == 3) {
            dg_bf16x4_t h, md, l;
            dg_split3((f32x4){o[0], o[1], o[2], o[3]}, h, md, l);
            *(dg_bf16x4_t*)(dy16 + pi) = h;
            *(dg_bf16x4_t*)(dy16 + pstride + pi) = md;
            *(dg_bf16x4_t*)(dy16 + 2 * pstride + pi) = l;
        }
    }
}

// the fp32-storage backward apply in the same geometry (see bn_act_fwd_rows_kernel): per-channel constants -- the four parameters and the
// two fp64 coefficients -- once per thread instead of six 16-byte and eight 8-byte loads per item; the item kernel's fp64 expression.
template <int D16, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_kernel(const BnGroup G, int M, int C, int rows_per, int TX, int coef_rchunks, int act, float slope,
                                                                long pstride, int cm) {
    constexpr int U = V == 4 ? BN_U / 2 : BN_U / 4;
    const BnProb& P = G.p[blockIdx.z];
    const float* __restrict__ dz = (const float*)P.dz;
    const float* __restrict__ y = (const float*)P.y;
    float* __restrict__ dy = (float*)P.out;
    __bf16* __restrict__ dy16 = (__bf16*)P.out16;
    const double* __restrict__ coef = (const double*)P.ws + (size_t)2 * coef_rchunks * C;
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * V;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float mean[V], istd[V], g[V], b[V], gs[V];
    double c1[V], c2[V];
    bn_ldp<V>(P.saved + c, mean);
    bn_ldp<V>(P.saved + C + c, istd);
    bn_ldp<V>(P.gamma + c, g);
    bn_ldp<V>(P.beta + c, b);
#pragma unroll
    for (int j = 0; j < V; ++j) {
        gs[j] = g[j] * istd[j];
        c1[j] = coef[c + j];
        c2[j] = coef[C + c + j];
    }
    auto put = [&](int row, const float (&v)[V], const float (&d)[V]) {
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float u = bn_norm(v[j], mean[j], gs[j], b[j]);
            const double gg = (double)(d[j] * act_grad(u, act, slope));
            const double xhat = ((double)v[j] - (double)mean[j]) * (double)istd[j];
            o[j] = (float)((double)g[j] * (double)istd[j] * (gg - c1[j] - xhat * c2[j]));
        }
        if (D16 != 3 || dy != nullptr) bn_stp<V>(dy + (long)row * C + c, o);   // plane-only output (f32x3 path): the fp32 copy has no reader
        if constexpr (D16 == 1) *(bf16x4_n*)(dy16 + (long)row * C + c) = __builtin_convertvector((f32x4){o[0], o[1], o[2], o[3]}, bf16x4_n);
        if constexpr (D16 == 3) bn_put_planes<V>(dy16, pstride, bn_plane_index(row, c, C, cm), o);
    };
    const float* py = y + c;
    const float* pd = dz + c;
    int r = r0 + ty;
    for (; r + (U - 1) * TY < r1; r += U * TY) {
        float v[U][V], d[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bn_ldp<V>(py + (long)(r + u * TY) * C, v[u]);
            bn_ldp<V>(pd + (long)(r + u * TY) * C, d[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) put(r + u * TY, v[u], d[u]);
    }
    for (; r < r1; r += TY) {
        float v[V], d[V];
        bn_ldp<V>(py + (long)r * C, v);
        bn_ldp<V>(pd + (long)r * C, d);
        put(r, v, d);
    }
}
// geometry of the two kernels above: TX channel lanes (16 for quad-chunk planes: a wave = 4 pixels x 64 channels), row chunks of a multiple of
// 8 rows.  Shapes they do not take (C not a multiple of 64, M not of 8) stay on the item kernels.
struct BnRows { int ok, tx, cchunks, rchunks, rows_per; };
static BnRows bn_rows(int M, int C, int cm, int V) {
    BnRows g = {0, 0, 0, 0, 0};
    if (C % 64 != 0 || M % 8 != 0 || dg_get_option(DG_OPT_BN_ITEMS)) return g;
    g.ok = 1;
    g.tx = 64 / V;                                  // quad-chunk planes: a wave = 4 (8) pixels x 64 channels
    if (!cm) while (g.tx < C / V && g.tx < 64) g.tx <<= 1;
    g.cchunks = C / (V * g.tx);
    const int ty = 256 / g.tx;
    long rc = M / ((long)ty * BN_U);
    long cap = 4096 / g.cchunks;
    if (rc > cap) rc = cap;
    if (rc < 1) rc = 1;
    g.rows_per = (int)(((M + rc - 1) / rc + 7) / 8 * 8);
    g.rchunks = (M + g.rows_per - 1) / g.rows_per;
    return g;
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const DgPtrs xs, const DgPtrs ys, long total4, long n, int act, float slope) {
    const float* __restrict__ x = dg_pick<const float>(xs, blockIdx.y);
    float* __restrict__ y = dg_pick<float>(ys, blockIdx.y);
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
__global__ __launch_bounds__(256) void act_bwd_kernel(const DgPtrs dys, const DgPtrs outs, const DgPtrs dxs, long total4, long n, int act, float slope) {
    const float* __restrict__ dy = dg_pick<const float>(dys, blockIdx.y);
    const float* __restrict__ out = dg_pick<const float>(outs, blockIdx.y);
    float* __restrict__ dx = dg_pick<float>(dxs, blockIdx.y);
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

// bf16 in / bf16 out (n % 8 == 0): the first layer's fused LeakyReLU backward on the bf16-stored gradient
__global__ __launch_bounds__(256) void act_bwd16_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ out,
                                                        __bf16* __restrict__ dx, long total8, int act, float slope) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total8; idx += (long)gridDim.x * 256) {
        float d[8], o[8], r[8];
        bn_ld(dy + idx * 8, d);
        bn_ld(out + idx * 8, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = act_bwd_one(d[j], o[j], act, slope);
        bn_st(dx + idx * 8, r);
    }
}

static int stream_grid(long total4) {
    long g = (total4 + 255) / 256;
    long cap = 2048;
#ifdef DG_EXPERIMENTS
    if (const int v = dg_get_option(DG_OPT_UNDERSTORY); v < 0) cap = -(long)v;      // probe: another cap on the workgroups of a streaming pass
#endif
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" size_t dg_bn_workspace_bytes(int M, int C) {
    const BnGrid g = bn_grid(M, C);                                        // V = 4 needs the larger one (more channel lanes)
    return ((size_t)2 * g.rchunks * C + 2 * (size_t)C) * sizeof(double);   // fp64 partials in the backward
}

// io_bf16: every activation tensor of the call (y / z / dz / dy) is bf16; statistics, parameters and their gradients stay fp32
// groups problems per launch (BnGroup), share: consecutive problems through the same module (see bn_stats_finalize_kernel)
template <typename T>
static int bn_train_stats_impl(int groups, int share, const BnGroup& G, int M, int C, float eps, float momentum, size_t ws_bytes, dg_stream_t stream) {
    constexpr int V = BnV<T>::V;
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS && share >= 1 && groups % share == 0, "dg_bn_train_stats: groups=%d share=%d", groups, share);
    for (int i = 0; i < groups; ++i) DG_CHECK_ARG(G.p[i].y && G.p[i].saved, "dg_bn_train_stats: null pointer (problem %d)", i);
    DG_CHECK_ARG(M >= 2, "dg_bn_train_stats: Expected more than 1 value per channel when training (M=%d)", M);
    DG_CHECK_ARG(C >= V && C % V == 0, "dg_bn_train_stats: C=%d must be a multiple of %d", C, V);
    for (int i = 0; i < groups; ++i)
        if (G.p[i].ws == nullptr || ws_bytes < dg_bn_workspace_bytes(M, C))
            return dg_fail(DG_ERR_WORKSPACE, "dg_bn_train_stats: workspace %zu < %zu", ws_bytes, dg_bn_workspace_bytes(M, C));
    const BnGrid g = bn_grid(M, C, V);
    const int cc = g.cchunks, rc = g.rchunks;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_partial_kernel<T>, dim3(cc, rc, groups), dim3(256), 0, st, G, M, C, rc, g.tx);
    DG_CHECK_LAUNCH("bn_stats_partial");
    hipLaunchKernelGGL(bn_stats_finalize_kernel<T>, dim3((C + BN_FIN_CH - 1) / BN_FIN_CH, groups / share), dim3(256), 0, st, G, share, M, C, rc, eps, momentum);
    DG_CHECK_LAUNCH("bn_stats_finalize");
    return DG_OK;
}
static BnGroup bn_group_zero() {
    BnGroup G;
    memset(&G, 0, sizeof(G));
    return G;
}
extern "C" int dg_bn_train_stats_t(const void* y, int io_bf16, int M, int C, float eps, float momentum, float* running_mean,
                                   float* running_var, int64_t* nbt, float* saved, void* ws, size_t ws_bytes,
                                   dg_stream_t stream) {
    BnGroup G = bn_group_zero();
    G.p[0].y = y; G.p[0].rmean = running_mean; G.p[0].rvar = running_var; G.p[0].nbt = nbt; G.p[0].saved = saved; G.p[0].ws = ws;
    if (io_bf16) return bn_train_stats_impl<__bf16>(1, 1, G, M, C, eps, momentum, ws_bytes, stream);
    return bn_train_stats_impl<float>(1, 1, G, M, C, eps, momentum, ws_bytes, stream);
}
extern "C" int dg_bn_train_stats(const float* y, int M, int C, float eps, float momentum, float* running_mean,
                                 float* running_var, int64_t* nbt, float* saved, void* ws, size_t ws_bytes,
                                 dg_stream_t stream) {
    return dg_bn_train_stats_t(y, 0, M, C, eps, momentum, running_mean, running_var, nbt, saved, ws, ws_bytes, stream);
}
extern "C" int dg_bn_train_stats_g(int groups, int share, const float* const* y, int M, int C, float eps, float momentum, float* const* running_mean,
                                   float* const* running_var, int64_t* const* nbt, float* const* saved, void* const* ws, size_t ws_bytes,
                                   dg_stream_t stream) {
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS && y && saved && ws, "dg_bn_train_stats_g: bad group / null table");
    BnGroup G = bn_group_zero();
    for (int i = 0; i < groups; ++i) {
        G.p[i].y = y[i]; G.p[i].saved = saved[i]; G.p[i].ws = ws[i];
        G.p[i].rmean = running_mean ? running_mean[i] : nullptr;
        G.p[i].rvar = running_var ? running_var[i] : nullptr;
        G.p[i].nbt = nbt ? nbt[i] : nullptr;
    }
    if (share > 1)
        for (int i = 0; i < groups; ++i)
            DG_CHECK_ARG(G.p[i].rmean == G.p[i / share * share].rmean, "dg_bn_train_stats_g: problems of one share set must name the same module buffers");
    return bn_train_stats_impl<float>(groups, share < 1 ? 1 : share, G, M, C, eps, momentum, ws_bytes, stream);
}

static int bn_partials_row_blocks(int P) {
    if (P < 4096) return 1;          // (tools/bench_partials.py: 16384 x 64: 83 -> 28 us, 4096 x 128: 25 -> 15 us; below that one launch wins)
    const int rb = P / 512;
    return rb < 2 ? 2 : (rb > 64 ? 64 : rb);
}
extern "C" size_t dg_bn_partials_workspace_bytes(int P, int C) {
    const int rb = bn_partials_row_blocks(P);
    return rb > 1 ? (size_t)rb * 2 * (size_t)((C + 3) / 4 * 4) * sizeof(double) : 0;
}
extern "C" int dg_bn_stats_from_partials(const float* stat, int P, int M, int C, float eps, float momentum,
                                         float* running_mean, float* running_var, int64_t* nbt, float* saved,
                                         void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(stat && saved && P >= 1, "dg_bn_stats_from_partials: bad argument");
    DG_CHECK_ARG(M >= 2, "dg_bn_stats_from_partials: Expected more than 1 value per channel when training (M=%d)", M);
    DG_CHECK_ARG(C >= 4 && C % 4 == 0, "dg_bn_stats_from_partials: C=%d must be a multiple of 4", C);
    // few channels x many partial rows (the window input-grad at 64 channels: 16384 rows of 784 bytes = 12.8 MB for 16 blocks)
    // is a latency-bound crawl in one launch (20.6 us average, 119 us worst over the 512 px step): from 4096 rows on the rows
    // are first reduced by RB row blocks per channel quad, then merged -- two short launches, every sum still in a fixed order
    const int RB = bn_partials_row_blocks(P);
    if (RB > 1 && ws != nullptr && ws_bytes >= dg_bn_partials_workspace_bytes(P, C)) {
        hipLaunchKernelGGL(bn_partials_reduce_kernel, dim3((C + 3) / 4, RB), dim3(256), 0, (hipStream_t)stream, stat, P, 3 * C + 4, C, RB, (double*)ws);
        DG_CHECK_LAUNCH("bn_partials_reduce");
        hipLaunchKernelGGL(bn_partials_merge_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const double*)ws, RB, stat, M, C, eps,
                           momentum, running_mean, running_var, nbt, saved);
        DG_CHECK_LAUNCH("bn_partials_merge");
        return DG_OK;
    }
    hipLaunchKernelGGL(bn_partials_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, stat, P, 3 * C + 4,
                       M, C, eps, momentum, running_mean, running_var, nbt, saved);
    DG_CHECK_LAUNCH("bn_partials_finalize");
    return DG_OK;
}

template <typename T>
static int bn_act_fwd_impl(int groups, const BnGroup& G, int M, int C, int act, float slope, dg_stream_t stream, long pstride = 0, int plane_cm = 0) {
    constexpr int V = BnV<T>::V;
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS, "dg_bn_act_fwd: groups=%d", groups);
    const void* z16 = G.p[0].out16;
    DG_CHECK_ARG(groups == 1 || z16 == nullptr, "dg_bn_act_fwd: shadow / plane outputs exist in the one-problem forms only");
    for (int i = 0; i < groups; ++i)
        DG_CHECK_ARG(G.p[i].y && (G.p[i].out || (G.p[i].out16 && pstride > 0)) && G.p[i].saved && G.p[i].gamma && G.p[i].beta, "dg_bn_act_fwd: null pointer");
    DG_CHECK_ARG(!plane_cm || (pstride > 0 && C % 64 == 0 && M % 4 == 0), "dg_bn_act_fwd: quad-chunk planes need plane operands, C %% 64 == 0 and M %% 4 == 0 (C=%d, M=%d)", C, M);
    DG_CHECK_ARG(C >= V && C % V == 0, "dg_bn_act_fwd: C=%d must be a multiple of %d", C, V);
    DG_CHECK_ARG(act == DG_ACT_NONE || act == DG_ACT_LEAKY || act == DG_ACT_RELU, "dg_bn_act_fwd: bad act %d", act);
    const long totalv = (long)M * C / V;
    if constexpr (V == 8) {
        const BnGrid g = bn_grid(M, C, V);
        hipLaunchKernelGGL(bn_act_fwd16_kernel, dim3(g.cchunks, g.rchunks, groups), dim3(256), 0, (hipStream_t)stream, G, M, C, g.rchunks, g.tx, act, slope);
        DG_CHECK_LAUNCH("bn_act_fwd16");
        return DG_OK;
    }
    if constexpr (V == 4) {
        const bool planes = z16 && pstride > 0;
        if (const BnRows rg = bn_rows(M, C, plane_cm, planes ? 8 : 4); rg.ok) {
            const dim3 rgrid(rg.cchunks, rg.rchunks, groups);
            if (planes)
                hipLaunchKernelGGL((bn_act_fwd_rows_kernel<3, 8>), rgrid, dim3(256), 0, (hipStream_t)stream, G, M, C, rg.rows_per, rg.tx, act, slope, pstride, plane_cm);
            else if (z16)
                hipLaunchKernelGGL((bn_act_fwd_rows_kernel<1, 4>), rgrid, dim3(256), 0, (hipStream_t)stream, G, M, C, rg.rows_per, rg.tx, act, slope, 0L, 0);
            else
                hipLaunchKernelGGL((bn_act_fwd_rows_kernel<0, 4>), rgrid, dim3(256), 0, (hipStream_t)stream, G, M, C, rg.rows_per, rg.tx, act, slope, 0L, 0);
            DG_CHECK_LAUNCH("bn_act_fwd_rows");
            return DG_OK;
        }
    }
    const dim3 grid(stream_grid(totalv), groups);
    if constexpr (V == 4) {
        if (z16 && pstride > 0) {
            hipLaunchKernelGGL((bn_act_fwd_kernel<T, T, 3>), grid, dim3(256), 0, (hipStream_t)stream, G, totalv, C, act, slope, pstride, plane_cm);
            DG_CHECK_LAUNCH("bn_act_fwd");
            return DG_OK;
        }
        if (z16) {
            hipLaunchKernelGGL((bn_act_fwd_kernel<T, T, 1>), grid, dim3(256), 0, (hipStream_t)stream, G, totalv, C, act, slope, 0L, 0);
            DG_CHECK_LAUNCH("bn_act_fwd");
            return DG_OK;
        }
    }
    hipLaunchKernelGGL((bn_act_fwd_kernel<T, T, 0>), grid, dim3(256), 0, (hipStream_t)stream, G, totalv, C, act, slope, 0L, 0);
    DG_CHECK_LAUNCH("bn_act_fwd");
    return DG_OK;
}
static BnGroup bn_fwd_one(const void* y, void* z, void* z16, const float* saved, const float* gamma, const float* beta) {
    BnGroup G = bn_group_zero();
    G.p[0].y = y; G.p[0].out = z; G.p[0].out16 = z16; G.p[0].saved = (float*)saved; G.p[0].gamma = gamma; G.p[0].beta = beta;
    return G;
}
extern "C" int dg_bn_act_fwd(const float* y, float* z, int M, int C, const float* saved, const float* gamma,
                             const float* beta, int act, float slope, dg_stream_t stream) {
    return bn_act_fwd_impl<float>(1, bn_fwd_one(y, z, nullptr, saved, gamma, beta), M, C, act, slope, stream);
}
extern "C" int dg_bn_act_fwd_g(int groups, const float* const* y, float* const* z, int M, int C, const float* const* saved, const float* const* gamma,
                               const float* const* beta, int act, float slope, dg_stream_t stream) {
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS && y && z && saved && gamma && beta, "dg_bn_act_fwd_g: bad group / null table");
    BnGroup G = bn_group_zero();
    for (int i = 0; i < groups; ++i) {
        G.p[i].y = y[i]; G.p[i].out = z[i]; G.p[i].saved = (float*)saved[i]; G.p[i].gamma = gamma[i]; G.p[i].beta = beta[i];
    }
    return bn_act_fwd_impl<float>(groups, G, M, C, act, slope, stream);
}
extern "C" int dg_bn_act_fwd_bf16(const float* y, float* z, void* z_bf16, int M, int C, const float* saved, const float* gamma,
                                  const float* beta, int act, float slope, dg_stream_t stream) {
    DG_CHECK_ARG(z_bf16, "dg_bn_act_fwd_bf16: null shadow pointer");
    return bn_act_fwd_impl<float>(1, bn_fwd_one(y, z, z_bf16, saved, gamma, beta), M, C, act, slope, stream);
}
// the same pass, also writing the three bf16 planes of z (plane_elems elements apart, >= M * C, % 8 == 0) for the f32x3 matrix path
extern "C" int dg_bn_act_fwd_x3(const float* y, float* z, void* z_planes, size_t plane_elems, int plane_layout, int M, int C, const float* saved,
                                const float* gamma, const float* beta, int act, float slope, dg_stream_t stream) {
    DG_CHECK_ARG(plane_layout == 0 || plane_layout == 1, "dg_bn_act_fwd_x3: plane_layout 0 (pixel-major) or 1 (quad-chunk)");
    DG_CHECK_ARG(z_planes, "dg_bn_act_fwd_x3: null plane pointer");
    DG_CHECK_ARG(plane_elems >= (size_t)M * C && plane_elems % 8 == 0, "dg_bn_act_fwd_x3: plane distance %zu for %ld elements", plane_elems, (long)M * C);
    return bn_act_fwd_impl<float>(1, bn_fwd_one(y, z, z_planes, saved, gamma, beta), M, C, act, slope, stream, (long)plane_elems, plane_layout);
}
extern "C" int dg_bn_act_fwd_t(const void* y, void* z, int io_bf16, int M, int C, const float* saved, const float* gamma,
                               const float* beta, int act, float slope, dg_stream_t stream) {
    if (io_bf16) return bn_act_fwd_impl<__bf16>(1, bn_fwd_one(y, z, nullptr, saved, gamma, beta), M, C, act, slope, stream);
    return bn_act_fwd_impl<float>(1, bn_fwd_one(y, z, nullptr, saved, gamma, beta), M, C, act, slope, stream);
}

template <typename T>
static int bn_act_bwd_impl(int groups, int share, const BnGroup& G, int M, int C, int act, float slope, int accumulate, size_t ws_bytes,
                           dg_stream_t stream, long pstride = 0, int plane_cm = 0) {
    constexpr int V = BnV<T>::V;
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS && share >= 1 && groups % share == 0, "dg_bn_act_bwd: groups=%d share=%d", groups, share);
    const void* dy16 = G.p[0].out16;
    DG_CHECK_ARG(groups == 1 || dy16 == nullptr, "dg_bn_act_bwd: shadow / plane outputs exist in the one-problem forms only");
    for (int i = 0; i < groups; ++i)
        DG_CHECK_ARG(G.p[i].dz && G.p[i].y && (G.p[i].out || (G.p[i].out16 && pstride > 0)) && G.p[i].saved && G.p[i].gamma && G.p[i].beta, "dg_bn_act_bwd: null pointer");
    DG_CHECK_ARG(!plane_cm || (pstride > 0 && C % 64 == 0 && M % 4 == 0), "dg_bn_act_bwd: quad-chunk planes need plane operands, C %% 64 == 0 and M %% 4 == 0 (C=%d, M=%d)", C, M);
    DG_CHECK_ARG(C >= V && C % V == 0, "dg_bn_act_bwd: C=%d must be a multiple of %d", C, V);
    DG_CHECK_ARG(act == DG_ACT_NONE || act == DG_ACT_LEAKY || act == DG_ACT_RELU, "dg_bn_act_bwd: bad act %d", act);
    for (int i = 0; i < groups; ++i)
        if (G.p[i].ws == nullptr || ws_bytes < dg_bn_workspace_bytes(M, C))
            return dg_fail(DG_ERR_WORKSPACE, "dg_bn_act_bwd: workspace %zu < %zu", ws_bytes, dg_bn_workspace_bytes(M, C));
    const BnGrid g = bn_grid(M, C, V);
    const int cc = g.cchunks, rc = g.rchunks;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_partial_kernel<T>, dim3(cc, rc, groups), dim3(256), 0, st, G, M, C, rc, g.tx, act, slope);
    DG_CHECK_LAUNCH("bn_bwd_partial");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + BN_FIN_CH - 1) / BN_FIN_CH, groups / share), dim3(256), 0, st, G, share, M, C, rc, accumulate);
    DG_CHECK_LAUNCH("bn_bwd_finalize");
    const long totalv = (long)M * C / V;
    if constexpr (V == 4) {
        const bool planes = dy16 && pstride > 0;
        if (const BnRows rg = bn_rows(M, C, plane_cm, planes ? 8 : 4); rg.ok) {
            const dim3 rgrid(rg.cchunks, rg.rchunks, groups);
            if (planes)
                hipLaunchKernelGGL((bn_bwd_apply_rows_kernel<3, 8>), rgrid, dim3(256), 0, st, G, M, C, rg.rows_per, rg.tx, rc, act, slope, pstride, plane_cm);
            else if (dy16)
                hipLaunchKernelGGL((bn_bwd_apply_rows_kernel<1, 4>), rgrid, dim3(256), 0, st, G, M, C, rg.rows_per, rg.tx, rc, act, slope, 0L, 0);
            else
                hipLaunchKernelGGL((bn_bwd_apply_rows_kernel<0, 4>), rgrid, dim3(256), 0, st, G, M, C, rg.rows_per, rg.tx, rc, act, slope, 0L, 0);
            DG_CHECK_LAUNCH("bn_bwd_apply_rows");
            return DG_OK;
        }
    }
    const dim3 grid(stream_grid(totalv), groups);
    if constexpr (V == 4) {
        if (dy16 && pstride > 0) {
            hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 3>), grid, dim3(256), 0, st, G, totalv, C, rc, act, slope, pstride, plane_cm);
            DG_CHECK_LAUNCH("bn_bwd_apply");
            return DG_OK;
        }
        if (dy16) {
            hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 1>), grid, dim3(256), 0, st, G, totalv, C, rc, act, slope, 0L, 0);
            DG_CHECK_LAUNCH("bn_bwd_apply");
            return DG_OK;
        }
    } else {
        hipLaunchKernelGGL(bn_bwd_apply16_kernel, dim3(cc, rc, groups), dim3(256), 0, st, G, M, C, rc, g.tx, rc, act, slope);
        DG_CHECK_LAUNCH("bn_bwd_apply16");
        return DG_OK;
    }
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 0>), grid, dim3(256), 0, st, G, totalv, C, rc, act, slope, 0L, 0);
    DG_CHECK_LAUNCH("bn_bwd_apply");
    return DG_OK;
}
static BnGroup bn_bwd_one(const void* dz, const void* y, void* dy, void* dy16, const float* saved, const float* gamma, const float* beta,
                          float* dgamma, float* dbeta, void* ws) {
    BnGroup G = bn_group_zero();
    BnProb& P = G.p[0];
    P.dz = dz; P.y = y; P.out = dy; P.out16 = dy16; P.saved = (float*)saved; P.gamma = gamma; P.beta = beta; P.dgamma = dgamma; P.dbeta = dbeta; P.ws = ws;
    return G;
}
extern "C" int dg_bn_act_bwd(const float* dz, const float* y, float* dy, int M, int C, const float* saved,
                             const float* gamma, const float* beta, int act, float slope, float* dgamma, float* dbeta,
                             int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    return bn_act_bwd_impl<float>(1, 1, bn_bwd_one(dz, y, dy, nullptr, saved, gamma, beta, dgamma, dbeta, ws), M, C, act, slope, accumulate, ws_bytes, stream);
}
extern "C" int dg_bn_act_bwd_g(int groups, int share, const float* const* dz, const float* const* y, float* const* dy, int M, int C,
                               const float* const* saved, const float* const* gamma, const float* const* beta, int act, float slope,
                               float* const* dgamma, float* const* dbeta, int accumulate, void* const* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS && dz && y && dy && saved && gamma && beta && ws, "dg_bn_act_bwd_g: bad group / null table");
    BnGroup G = bn_group_zero();
    for (int i = 0; i < groups; ++i) {
        BnProb& P = G.p[i];
        P.dz = dz[i]; P.y = y[i]; P.out = dy[i]; P.saved = (float*)saved[i]; P.gamma = gamma[i]; P.beta = beta[i]; P.ws = ws[i];
        P.dgamma = dgamma ? dgamma[i] : nullptr;
        P.dbeta = dbeta ? dbeta[i] : nullptr;
    }
    if (share > 1)
        for (int i = 0; i < groups; ++i)
            DG_CHECK_ARG(G.p[i].dgamma == G.p[i / share * share].dgamma && G.p[i].dbeta == G.p[i / share * share].dbeta,
                         "dg_bn_act_bwd_g: problems of one share set must name the same dgamma / dbeta");
    return bn_act_bwd_impl<float>(groups, share < 1 ? 1 : share, G, M, C, act, slope, accumulate, ws_bytes, stream);
}
extern "C" int dg_bn_act_bwd_bf16(const float* dz, const float* y, float* dy, void* dy_bf16, int M, int C, const float* saved,
                                  const float* gamma, const float* beta, int act, float slope, float* dgamma, float* dbeta,
                                  int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(dy_bf16, "dg_bn_act_bwd_bf16: null shadow pointer");
    return bn_act_bwd_impl<float>(1, 1, bn_bwd_one(dz, y, dy, dy_bf16, saved, gamma, beta, dgamma, dbeta, ws), M, C, act, slope, accumulate, ws_bytes, stream);
}
extern "C" int dg_bn_act_bwd_x3(const float* dz, const float* y, float* dy, void* dy_planes, size_t plane_elems, int plane_layout, int M, int C,
                                const float* saved, const float* gamma, const float* beta, int act, float slope, float* dgamma,
                                float* dbeta, int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    DG_CHECK_ARG(plane_layout == 0 || plane_layout == 1, "dg_bn_act_bwd_x3: plane_layout 0 (pixel-major) or 1 (quad-chunk)");
    DG_CHECK_ARG(dy_planes, "dg_bn_act_bwd_x3: null plane pointer");
    DG_CHECK_ARG(plane_elems >= (size_t)M * C && plane_elems % 8 == 0, "dg_bn_act_bwd_x3: plane distance %zu for %ld elements", plane_elems, (long)M * C);
    return bn_act_bwd_impl<float>(1, 1, bn_bwd_one(dz, y, dy, dy_planes, saved, gamma, beta, dgamma, dbeta, ws), M, C, act, slope, accumulate, ws_bytes,
                                  stream, (long)plane_elems, plane_layout);
}
extern "C" int dg_bn_act_bwd_t(const void* dz, const void* y, void* dy, int io_bf16, int M, int C, const float* saved,
                               const float* gamma, const float* beta, int act, float slope, float* dgamma, float* dbeta,
                               int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream) {
    const BnGroup G = bn_bwd_one(dz, y, dy, nullptr, saved, gamma, beta, dgamma, dbeta, ws);
    if (io_bf16) return bn_act_bwd_impl<__bf16>(1, 1, G, M, C, act, slope, accumulate, ws_bytes, stream);
    return bn_act_bwd_impl<float>(1, 1, G, M, C, act, slope, accumulate, ws_bytes, stream);
}

#ifdef DG_EXPERIMENTS
#include <utility>
template <int... Q, typename F>
__device__ __forceinline__ void bn_static_for(std::integer_sequence<int, Q...>, F&& f) {
    (f(std::integral_constant<int, Q>{}), ...);
}
// "Understory" form of a streaming kernel (experiment; tools/probe_corun.py): can an HBM-bound pass run BENEATH a saturating MFMA conv
// kernel instead of taking turns with it?  The conv kernels hold 2 waves x 217-240 registers per SIMD and 96-128 KB of LDS, so what is
// left on a CU is ONE wave of <= 32 registers per SIMD and 32-64 KB of LDS -- far too few registers for the loads in flight that HBM
// latency needs (the plain kernels reach 1 TB/s beside a conv kernel).  Here the loads in flight live in LDS: every wave keeps NP 1-KiB
// pieces under way with `buffer_load_dwordx4 ... lds` (no registers), takes the oldest with ONE ds_read_b128 per lane, and re-issues
// the slot.  A workgroup = 4 waves (one per SIMD), 4 NP KiB of LDS; vmcnt counts the DMA loads and the stores in issue order.
template <int NP>
__global__ __launch_bounds__(256) void act_fwd_understory_kernel(const float* __restrict__ x, float* __restrict__ y, long nbytes, long wgbytes, int act,
                                                                 float slope) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * NP * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long base = (long)blockIdx.x * wgbytes;                       // this workgroup's byte range [base, base + span)
    const long left = nbytes - base;
    const int span = (int)(left < wgbytes ? left : wgbytes);
    const int npc = (span + 1023) >> 10;                                // 1-KiB pieces of the range; wave w takes pieces w, w + 4, ...
    const int cnt = npc > wave ? (npc - wave + 3) >> 2 : 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)x + base), 0, span, 0x00020000);
    const unsigned lds_w = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem + (unsigned)wave * NP * 1024;
    constexpr int OOR = (int)0x80000000;
    auto dma = [&](int slot, int voff) {
        unsigned keep;
        const unsigned dst = lds_w + (unsigned)slot * 1024;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(rx), "s"(dst)
                     : "memory");
    };
    auto off_of = [&](int i) -> int { return i < cnt ? ((wave + 4 * i) << 10) + lane * 16 : OOR; };
#pragma unroll
    for (int k = 0; k < NP; ++k) dma(k, off_of(k));
    for (int i = 0; i < cnt; ++i) {
        // piece i's DMA is followed by (issue order) the rest of the prologue and one store + one DMA per earlier iteration
        if (i >= NP - 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NP - 1)) : "memory");
        } else {
            bn_static_for(std::make_integer_sequence<int, NP - 1>{}, [&](auto q) {
                if (i == q) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP - 1 + (int)q) : "memory");
            });
        }
        const int slot = i % NP;
        f32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_w + (unsigned)slot * 1024 + (unsigned)lane * 16) : "memory");
        const int o = off_of(i);
        dma(slot, off_of(i + NP));
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = act == DG_ACT_SIGMOID ? 1.f / (1.f + expf(-v[j])) : dg_apply_act(v[j], act, slope);
        if (o + 16 <= span) *(f32x4*)((char*)y + base + o) = r;
        else asm volatile("s_nop 0");                                  // (the skipped store only makes the next waits stricter than needed: never looser)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the zero-fill DMAs of the tail still target this workgroup's LDS
}
template <int NP>
static int act_fwd_understory(const float* x, float* y, size_t n, int act, float slope, hipStream_t st, int wgs) {
    const long nbytes = (long)n * 4;
    long wgbytes = 4L * NP * 1024 * 4;                                   // >= 4 rounds of the ring per workgroup
    if (wgs > 0) wgbytes = ((nbytes + wgs - 1) / wgs + 4095) / 4096 * 4096;      // a fixed number of long-lived workgroups (one per CU: 256)
    else while ((nbytes + wgbytes - 1) / wgbytes > 2048) wgbytes *= 2;
    DG_CHECK_ARG(wgbytes < (1L << 31), "act_fwd_understory: range per workgroup");
    const int grid = (int)((nbytes + wgbytes - 1) / wgbytes);
    hipLaunchKernelGGL((act_fwd_understory_kernel<NP>), dim3(grid), dim3(256), 0, st, x, y, nbytes, wgbytes, act, slope);
    DG_CHECK_LAUNCH("act_fwd_understory");
    return DG_OK;
}
#endif

extern "C" int dg_act_fwd_g(int groups, const float* const* x, float* const* y, size_t n, int act, float slope, dg_stream_t stream) {
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS && x && y, "dg_act_fwd: bad group / null table");
    for (int i = 0; i < groups; ++i) DG_CHECK_ARG(x[i] && y[i], "dg_act_fwd: null pointer");
    DG_CHECK_ARG(act >= DG_ACT_NONE && act <= DG_ACT_SIGMOID, "dg_act_fwd: bad act %d", act);
    if (n == 0) return DG_OK;
#ifdef DG_EXPERIMENTS
    if (const int np = dg_get_option(DG_OPT_UNDERSTORY); np && groups == 1 && n % 4 == 0) {
        const int wgs = np / 100, q = np % 100;                          // e.g. 25616: 256 long-lived workgroups, 16 pieces per wave
        if (q == 16) return act_fwd_understory<16>(x[0], y[0], n, act, slope, (hipStream_t)stream, wgs);
        if (q == 8) return act_fwd_understory<8>(x[0], y[0], n, act, slope, (hipStream_t)stream, wgs);
        return act_fwd_understory<4>(x[0], y[0], n, act, slope, (hipStream_t)stream, wgs);
    }
#endif
    const long total4 = (long)((n + 3) / 4);
    hipLaunchKernelGGL(act_fwd_kernel, dim3(stream_grid(total4), groups), dim3(256), 0, (hipStream_t)stream, dg_ptrs((const void* const*)x, groups),
                       dg_ptrs((const void* const*)y, groups), total4, (long)n, act, slope);
    DG_CHECK_LAUNCH("act_fwd");
    return DG_OK;
}
extern "C" int dg_act_fwd(const float* x, float* y, size_t n, int act, float slope, dg_stream_t stream) {
    return dg_act_fwd_g(1, &x, &y, n, act, slope, stream);
}
extern "C" int dg_act_bwd_g(int groups, const float* const* dy, const float* const* out, float* const* dx, size_t n, int act, float slope,
                            dg_stream_t stream) {
    DG_CHECK_ARG(groups >= 1 && groups <= DG_MAX_GROUPS && dy && out && dx, "dg_act_bwd: bad group / null table");
    for (int i = 0; i < groups; ++i) DG_CHECK_ARG(dy[i] && out[i] && dx[i], "dg_act_bwd: null pointer");
    DG_CHECK_ARG(act >= DG_ACT_NONE && act <= DG_ACT_SIGMOID, "dg_act_bwd: bad act %d", act);
    if (n == 0) return DG_OK;
    const long total4 = (long)((n + 3) / 4);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_grid(total4), groups), dim3(256), 0, (hipStream_t)stream, dg_ptrs((const void* const*)dy, groups),
                       dg_ptrs((const void* const*)out, groups), dg_ptrs((const void* const*)dx, groups), total4, (long)n, act, slope);
    DG_CHECK_LAUNCH("act_bwd");
    return DG_OK;
}
extern "C" int dg_act_bwd(const float* dy, const float* out, float* dx, size_t n, int act, float slope, dg_stream_t stream) {
    return dg_act_bwd_g(1, &dy, &out, &dx, n, act, slope, stream);
}
extern "C" int dg_act_bwd_t(const void* dy, const void* out, void* dx, int io_bf16, size_t n, int act, float slope, dg_stream_t stream) {
    if (!io_bf16) return dg_act_bwd((const float*)dy, (const float*)out, (float*)dx, n, act, slope, stream);
    DG_CHECK_ARG(dy && out && dx, "dg_act_bwd_t: null pointer");
    DG_CHECK_ARG(act >= DG_ACT_NONE && act <= DG_ACT_SIGMOID, "dg_act_bwd_t: bad act %d", act);
    DG_CHECK_ARG(n % 8 == 0, "dg_act_bwd_t: bf16 tensors need n %% 8 == 0 (n=%zu)", n);
    if (n == 0) return DG_OK;
    const long total8 = (long)(n / 8);
    hipLaunchKernelGGL(act_bwd16_kernel, dim3(stream_grid(total8)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dy, (const __bf16*)out,
                       (__bf16*)dx, total8, act, slope);
    DG_CHECK_LAUNCH("act_bwd16");
    return DG_OK;
}
