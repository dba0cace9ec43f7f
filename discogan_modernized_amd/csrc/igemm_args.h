// Arguments shared by the implicit-GEMM conv kernels (igemm.hip: register-staged fp32 / bf16 / f32x3 tiles;
// igemm_dma.hip: bf16 operands staged by LDS-DMA; igemm_dma_x3.hip: fp32 operands as three bf16 planes staged by LDS-DMA).  Internal; the public ABI is include/discogan_hip.h.
#pragma once
#include "dg_common.h"

enum { MODE_FWD = 0, MODE_DGRAD_S2 = 1, MODE_DGRAD_PLAIN = 2, MODE_WGRAD = 3, MODE_FWD_C3 = 4 };

struct IgemmArgs {
    const float* A;
    const float* B;
    float* C;
    float* part;  // split-K slabs (nullptr when splits == 1)
    int N, H, W, Cc, K;  // conv geometry: x[N,H,W,Cc], K out channels
    int Ho, Wo, lgHo, lgWo;
    int stride, pad;
    int M, Ng, R;  // GEMM rows, cols; R = reduction length in elements (WGRAD: pixels)
    int nIt, itPerSplit, splits;
    int tilesM, tilesN;
    int accumulate;
    int act;      // fused activation on the output (FWD_C3 training path; every mode on the inference path)
    float slope;
    int a16, b16;        // PREC 1: operand A / B is a bf16 tensor in HBM (same logical layout, 2 bytes per element);
                         // 3 (PREC 2, igemm_dma_x3.hip): three bf16 planes hi / mid / lo, plane-major, a_plane / b_plane bytes apart
    long a_plane, b_plane;
    int a_cm;            // igemm_dma_x3.hip WGRAD / igemm_dma_x3_dgw.hip: operand A's planes are in the QUAD-CHUNK layout
                         // [pixels / 4][K / 16][4][16] (written so by dg_bn_act_fwd_x3 / dg_bn_act_bwd_x3, plane_layout 1) instead of [pixels][K]
    int b_transposed;    // igemm_dma_x3.hip FWD: the weight planes are the transposed copy [(r, s, c)][k]
    int out16;           // the output tensor C is bf16 (RNE of the fp32 accumulators; FWD / DGRAD modes, never the weight gradient)
    int dbg_zero;        // timing experiments: drop the A (bit 0) / B (bit 1) operand loads
    int bias_mod;        // channels the bias cycles over in the column index (Ng, or Cc for DGRAD_PLAIN's (r,s,c) columns)
    const float* bias;   // inference path (BatchNorm folded into the conv): per-output-channel bias added before act; nullptr = none
    // fused BatchNorm statistics (FWD / DGRAD_S2): per-tile partial rows [P][3*Ng + 4] =
    // {count, -, -, -, shift[Ng], sum(y - shift)[Ng], sum((y - shift)^2)[Ng]}; nullptr = off
    float* stat;
    int stat_rs;
    // profiling hook (tools/igemm_stamps.py): per workgroup {wall0, cyc0, cyc after prologue, cyc after K loop,
    // cyc after the epilogue stores are issued, wall1, XCC/CU id}; nullptr = off
    long long* stamps;
    unsigned abytes, bbytes;   // operand sizes for the buffer descriptors (BUF kernels: both < 2 GiB)
    int prec;                  // 1: bf16 MFMA operands (option "bf16"; BUF kernels only)
    int xcd_group;             // workgroups sharing operand-A rows are placed on one XCD (needs tilesM * splits % 8 == 0)
    // Grouped launch (round 4; dg_conv_*_g): `groups` problems of IDENTICAL geometry and plan go out as one launch, blockIdx.z =
    // problem index g.  A / B / C / part / stat above are problem 0's tensors; gd*[g - 1] is the BYTE distance of problem g's tensor
    // from problem 0's (any two allocations have one).  The reference issues such problems as independent passes
    // (image_translation.py:342-361: G_B(A) | G_A(B), G_A(AB) | G_B(BA), D_A(A) | D_A(BA) | D_B(B) | D_B(AB)).
    // share (weight gradient, split-K reduction kernels only): `share` consecutive problems ACCUMULATE into the same output tensor
    // (a discriminator's real and fake pass, image_translation.py:353-361): the reduction kernel of output problem z adds the slab
    // sums of problems z * share .. z * share + share - 1 one after the other, in that order -- bitwise what `share` launches do.
    int groups, share;
    long gdA[DG_MAX_GROUPS - 1], gdB[DG_MAX_GROUPS - 1], gdC[DG_MAX_GROUPS - 1], gdPart[DG_MAX_GROUPS - 1], gdStat[DG_MAX_GROUPS - 1];
};

// problem g's pointer (g = blockIdx.z; wave-uniform: scalar loads and adds)
template <typename T>
__device__ __forceinline__ T* dg_group_ptr(T* base, const long* gd, int g) {
    return g == 0 ? base : (T*)((const char*)base + gd[g - 1]);
}


typedef __bf16 dg_bf16x4 __attribute__((ext_vector_type(4)));
// store 4 consecutive outputs: fp32 (16 B) or RNE-rounded bf16 (8 B); `elem` = element offset from the tensor base
__device__ __forceinline__ void dg_store_out4(float* base, long elem, const f32x4& v, int out16) {
    if (out16) *(dg_bf16x4*)((__bf16*)base + elem) = __builtin_convertvector(v, dg_bf16x4);
    else *(f32x4*)(base + elem) = v;
}
