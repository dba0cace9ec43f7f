// optim.Adam over flat fp32 buffers (image_translation.py:275-287: lr 2e-4, betas (0.5, 0.999), eps 1e-8,
// coupled L2 weight_decay 1e-5).  One launch per optimiser step over the whole flat parameter group
// (G_A+G_B or D_A+D_B): 28 B/param of HBM traffic, 16-byte accesses.
// The step counter and the bias-correction scalars live in DEVICE memory and are advanced by a
// one-thread kernel, so an optimiser step is capturable in a hipGraph and replays correctly.
// Operation order follows torch/optim/adam.py::_single_tensor_adam:
//   g' = g + wd*p ; m = m + (g'-m)*(1-b1) ; v = v*b2 + (1-b2)*g'*g' ;
//   denom = sqrt(v)/sqrt(1-b2^t) + eps ; p = p - (lr/(1-b1^t)) * m/denom
#include "dg_common.h"

__global__ void adam_advance_kernel(double* state, double lr, double b1, double b2) {
    const double t = state[0] + 1.0;
    state[0] = t;
    state[1] = lr / (1.0 - pow(b1, t));
    state[2] = sqrt(1.0 - pow(b2, t));
}

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
// One parameter's update, every operation rounded on its own (no fused multiply-adds chosen by the compiler): the kernel's code paths
// (4 or 8 parameters per trip, the scalar tail, the three output forms) must give the same bits, and the reference's CPU ops round
// op by op as well.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float gscale, float wd, float omb1, float b2, float omb2,
                                         float bc2_sqrt, float eps, float step_size) {
#pragma clang fp contract(off)
    const float gr = g * gscale + wd * p;
    m = m + (gr - m) * omb1;
    v = v * b2 + omb2 * gr * gr;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}
// P16 = 1: also write a bf16 (RNE) shadow of the updated parameters for the bf16 matrix path (+2 B/param on 28);
// P16 = 3: the three bf16 planes of the f32x3 matrix path (dg_split3; planes `pstride` elements apart, +6 B/param)
template <int P16>
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n,
                                                        const double* __restrict__ state, float b1, float b2, float eps,
                                                        float wd, float gscale, __bf16* __restrict__ p16, long pstride) {
    const float step_size = (float)state[1];
    const float bc2_sqrt = (float)state[2];
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    const long n4 = n >> 2;
    long i0 = (long)blockIdx.x * 256 + threadIdx.x;
    if constexpr (P16 == 3) {
        // plane outputs: 8 parameters per thread and trip, so that every plane store is 16 bytes (the 8-byte plane stores of the 4-parameter
        // trip made this form 15 % slower per byte than the plain one: 4.73 against 5.55 TB/s alone at 460 M parameters); same arithmetic
        typedef __bf16 bf16x8_a __attribute__((ext_vector_type(8)));
        const long n8 = (((size_t)p16 | (size_t)(pstride * 2)) & 15) == 0 ? n >> 3 : 0;      // a range that starts 8 bytes into a 16-byte plane granule keeps the 4-parameter trip
        for (long i = i0; i < n8; i += (long)gridDim.x * 256) {
            f32x4 pp[2], gg[2], mm[2], vv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                pp[h] = *(const f32x4*)(p + i * 8 + 4 * h);
                gg[h] = *(const f32x4*)(g + i * 8 + 4 * h);
                mm[h] = *(const f32x4*)(m + i * 8 + 4 * h);
                vv[h] = *(const f32x4*)(v + i * 8 + 4 * h);
            }
            dg_bf16x4_t hh[2], md[2], ll[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = pp[h][j], b = mm[h][j], c = vv[h][j];
                    adam_one(a, gg[h][j], b, c, gscale, wd, omb1, b2, omb2, bc2_sqrt, eps, step_size);
                    pp[h][j] = a; mm[h][j] = b; vv[h][j] = c;
                }
                *(f32x4*)(p + i * 8 + 4 * h) = pp[h];
                *(f32x4*)(m + i * 8 + 4 * h) = mm[h];
                *(f32x4*)(v + i * 8 + 4 * h) = vv[h];
                dg_split3(pp[h], hh[h], md[h], ll[h]);
            }
            *(bf16x8_a*)(p16 + i * 8) = __builtin_shufflevector(hh[0], hh[1], 0, 1, 2, 3, 4, 5, 6, 7);
            *(bf16x8_a*)(p16 + pstride + i * 8) = __builtin_shufflevector(md[0], md[1], 0, 1, 2, 3, 4, 5, 6, 7);
            *(bf16x8_a*)(p16 + 2 * pstride + i * 8) = __builtin_shufflevector(ll[0], ll[1], 0, 1, 2, 3, 4, 5, 6, 7);
        }
        i0 += n8 * 2;                 // the 4-parameter loop below takes what is left (n8 > 0: at most one trip of one thread)
    }
    for (long i = i0; i < n4; i += (long)gridDim.x * 256) {
        f32x4 pp = *(const f32x4*)(p + i * 4);
        const f32x4 gg = *(const f32x4*)(g + i * 4);
        f32x4 mm = *(const f32x4*)(m + i * 4);
        f32x4 vv = *(const f32x4*)(v + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = pp[j], b = mm[j], c = vv[j];
            adam_one(a, gg[j], b, c, gscale, wd, omb1, b2, omb2, bc2_sqrt, eps, step_size);
            pp[j] = a; mm[j] = b; vv[j] = c;
        }
        *(f32x4*)(p + i * 4) = pp;
        *(f32x4*)(m + i * 4) = mm;
        *(f32x4*)(v + i * 4) = vv;
        if (P16 == 1) *(bf16x4_t*)(p16 + i * 4) = __builtin_convertvector(pp, bf16x4_t);
        if (P16 == 3) {
            dg_bf16x4_t h, md, l;
            dg_split3(pp, h, md, l);
            *(dg_bf16x4_t*)(p16 + i * 4) = h;
            *(dg_bf16x4_t*)(p16 + pstride + i * 4) = md;
            *(dg_bf16x4_t*)(p16 + 2 * pstride + i * 4) = l;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (long e = n4 * 4; e < n; ++e) {
            float a = p[e], b = m[e], c = v[e];
            adam_one(a, g[e], b, c, gscale, wd, omb1, b2, omb2, bc2_sqrt, eps, step_size);
            p[e] = a; m[e] = b; v[e] = c;
            if (P16 == 1) p16[e] = (__bf16)p[e];
            if (P16 == 3) {
                const __bf16 h = (__bf16)p[e];
                const float r = p[e] - (float)h;
                const __bf16 md = (__bf16)r;
                p16[e] = h;
                p16[pstride + e] = md;
                p16[2 * pstride + e] = (__bf16)(r - (float)md);
            }
        }
    }
}

extern "C" int dg_adam_advance(double* state, double lr, double beta1, double beta2, dg_stream_t stream) {
    DG_CHECK_ARG(state, "dg_adam_advance: null state");
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, lr, beta1, beta2);
    DG_CHECK_LAUNCH("adam_advance");
    return DG_OK;
}
extern "C" int dg_adam_step_flat(float* p, const float* g, float* m, float* v, size_t n, const double* state, float beta1,
                                 float beta2, float eps, float weight_decay, float grad_scale, dg_stream_t stream) {
    DG_CHECK_ARG(p && g && m && v && state, "dg_adam_step_flat: null pointer");
    if (n == 0) return DG_OK;
    long grid = ((long)(n / 4) + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(adam_step_kernel<0>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, state,
                       beta1, beta2, eps, weight_decay, grad_scale, (__bf16*)nullptr, 0L);
    DG_CHECK_LAUNCH("adam_step");
    return DG_OK;
}
extern "C" int dg_adam_step_flat_bf16(float* p, const float* g, float* m, float* v, size_t n, const double* state, float beta1,
                                      float beta2, float eps, float weight_decay, float grad_scale, void* p16, dg_stream_t stream) {
    DG_CHECK_ARG(p && g && m && v && state && p16, "dg_adam_step_flat_bf16: null pointer");
    if (n == 0) return DG_OK;
    long grid = ((long)(n / 4) + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(adam_step_kernel<1>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, state,
                       beta1, beta2, eps, weight_decay, grad_scale, (__bf16*)p16, 0L);
    DG_CHECK_LAUNCH("adam_step_bf16");
    return DG_OK;
}
// the same step, also writing the parameters' three bf16 planes (hi / mid / lo, plane_elems elements apart; plane_elems >= n, % 8 == 0)
extern "C" int dg_adam_step_flat_x3(float* p, const float* g, float* m, float* v, size_t n, const double* state, float beta1,
                                    float beta2, float eps, float weight_decay, float grad_scale, void* p3, size_t plane_elems,
                                    dg_stream_t stream) {
    DG_CHECK_ARG(p && g && m && v && state && p3, "dg_adam_step_flat_x3: null pointer");
    DG_CHECK_ARG(plane_elems >= n && plane_elems % 8 == 0, "dg_adam_step_flat_x3: plane distance %zu for %zu parameters", plane_elems, n);
    if (n == 0) return DG_OK;
    long grid = ((long)(n / 4) + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(adam_step_kernel<3>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, state,
                       beta1, beta2, eps, weight_decay, grad_scale, (__bf16*)p3, (long)plane_elems);
    DG_CHECK_LAUNCH("adam_step_x3");
    return DG_OK;
}
// fp32 -> bf16 (RNE) copy: initial weight shadow / shadows of tensors that have no fused producer
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, long n) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        *(bf16x4_t*)(y + i * 4) = __builtin_convertvector(*(const f32x4*)(x + i * 4), bf16x4_t);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long e = n4 * 4; e < n; ++e) y[e] = (__bf16)x[e];
}
extern "C" int dg_f32_to_bf16(const float* x, void* y, size_t n, dg_stream_t stream) {
    DG_CHECK_ARG(x && y, "dg_f32_to_bf16: null pointer");
    if (n == 0) return DG_OK;
    long grid = ((long)(n / 4) + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)y, (long)n);
    DG_CHECK_LAUNCH("f32_to_bf16");
    return DG_OK;
}
// fp32 -> three bf16 planes (dg_split3): plane operands of tensors that have no fused producer
__global__ __launch_bounds__(256) void f32_to_bf16x3_kernel(const float* __restrict__ x, __bf16* __restrict__ y, long n, long pstride) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        dg_bf16x4_t h, md, l;
        dg_split3(*(const f32x4*)(x + i * 4), h, md, l);
        *(dg_bf16x4_t*)(y + i * 4) = h;
        *(dg_bf16x4_t*)(y + pstride + i * 4) = md;
        *(dg_bf16x4_t*)(y + 2 * pstride + i * 4) = l;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long e = n4 * 4; e < n; ++e) {
            const __bf16 h = (__bf16)x[e];
            const float r = x[e] - (float)h;
            const __bf16 md = (__bf16)r;
            y[e] = h;
            y[pstride + e] = md;
            y[2 * pstride + e] = (__bf16)(r - (float)md);
        }
}
extern "C" int dg_f32_to_bf16x3(const float* x, void* y3, size_t n, size_t plane_elems, dg_stream_t stream) {
    DG_CHECK_ARG(x && y3, "dg_f32_to_bf16x3: null pointer");
    DG_CHECK_ARG(plane_elems >= n && plane_elems % 8 == 0, "dg_f32_to_bf16x3: plane distance %zu for %zu elements", plane_elems, n);
    if (n == 0) return DG_OK;
    long grid = ((long)(n / 4) + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(f32_to_bf16x3_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)y3, (long)n, (long)plane_elems);
    DG_CHECK_LAUNCH("f32_to_bf16x3");
    return DG_OK;
}
