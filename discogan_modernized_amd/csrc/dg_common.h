// Shared helpers for the gfx950 DiscoGAN kernels (internal; the public ABI is include/discogan_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/discogan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 dg_bf16x4_t __attribute__((ext_vector_type(4)));

// ---- error plumbing (thread-local message, never abort) ---------------------------------------
extern thread_local char dg_err_buf[512];
int dg_fail(int code, const char* fmt, ...);

#define DG_CHECK_ARG(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return dg_fail(DG_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define DG_CHECK_LAUNCH(name)                                                                    \
    do {                                                                                         \
        hipError_t e__ = hipGetLastError();                                                      \
        if (e__ != hipSuccess) return dg_fail(DG_ERR_HIP, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline int dg_ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static inline bool dg_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline size_t dg_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int dg_get_option(int idx);
// Arithmetic of the conv products for the CURRENT call (DG_PREC_*): the `prec` argument of the *_g / *_p entry points -- held in a
// thread-local for the duration of that call (DgPrecScope) so that planning helpers and the edge kernels' launchers see it -- else the
// process default dg_set_option("bf16", n), which only the entry points WITHOUT the argument fall back to.
int dg_cur_prec();
struct DgPrecScope {
    int old;
    explicit DgPrecScope(int prec);
    ~DgPrecScope();
};
// Problems the split-K plan of the CURRENT call is sized for (plan_groups of the *_g conv entry points): 1 = every problem planned as if
// launched alone (bitwise the one-problem result), g = the whole grouped launch fills the chip, so fewer K-splits per problem.
int dg_cur_plan_groups();
struct DgPlanScope {
    int old;
    explicit DgPlanScope(int plan_groups);
    ~DgPlanScope();
};
enum { DG_OPT_SPLITK = 0, DG_OPT_KT = 1, DG_OPT_TARGET_WGS = 2, DG_OPT_RESERVED = 3, DG_OPT_SPLIT_BELOW = 4, DG_OPT_POINTER_PATH = 5, DG_OPT_BF16 = 6, DG_OPT_DBG_ZERO = 7, DG_OPT_NO_DMA = 8, DG_OPT_DMA_MFMA = 9, DG_OPT_X3_MFMA = 10, DG_OPT_DGW_PERSIST = 11, DG_OPT_UNDERSTORY = 12, DG_OPT_BN_ITEMS = 13, DG_OPT_COUNT = 14 };

// ---- grouped launches (round 4): one tensor per problem, picked by a block index (wave-uniform: scalar loads) ------------------
struct DgPtrs {
    const void* p[DG_MAX_GROUPS];
};
static inline DgPtrs dg_ptrs(const void* const* arr, int n) {
    DgPtrs r;
    for (int i = 0; i < DG_MAX_GROUPS; ++i) r.p[i] = (arr != nullptr && i < n) ? arr[i] : nullptr;
    return r;
}
static inline DgPtrs dg_ptrs1(const void* p0) {
    DgPtrs r;
    for (int i = 0; i < DG_MAX_GROUPS; ++i) r.p[i] = nullptr;
    r.p[0] = p0;
    return r;
}
template <typename T>
__device__ __forceinline__ T* dg_pick(const DgPtrs& t, int g) { return (T*)t.p[g]; }

// ---- device helpers ----------------------------------------------------------------------------
__device__ __forceinline__ float dg_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double dg_wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in thread 0
__device__ __forceinline__ float dg_block_sum256(float v, float* red /*>=4 floats LDS*/) {
    v = dg_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

__device__ __forceinline__ float dg_apply_act(float u, int act, float slope) {
    if (act == DG_ACT_LEAKY) return u > 0.f ? u : u * slope;
    if (act == DG_ACT_RELU) return u > 0.f ? u : 0.f;
    if (act == DG_ACT_SIGMOID) return 1.f / (1.f + __expf(-u));
    return u;
}

// fp32 -> three bf16 planes (the "f32x3" operand form of igemm.hip PREC 2 / igemm_dma_x3.hip): hi = bf16(v) (RNE),
// mid = bf16(v - hi), lo = bf16(v - hi - mid); both subtractions are exact in fp32, hi + mid + lo carries 24 significand bits
__device__ __forceinline__ void dg_split3(const f32x4& v, dg_bf16x4_t& hi, dg_bf16x4_t& mid, dg_bf16x4_t& lo) {
    hi = __builtin_convertvector(v, dg_bf16x4_t);
    const f32x4 r = v - __builtin_convertvector(hi, f32x4);
    mid = __builtin_convertvector(r, dg_bf16x4_t);
    lo = __builtin_convertvector(r - __builtin_convertvector(mid, f32x4), dg_bf16x4_t);
}
