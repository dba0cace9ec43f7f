// Probe: which MFMA body should the f32x3 plane kernel run?  LDS fragment reads + MFMAs of one 128 x 64 wave tile, eight waves per
// workgroup, one workgroup per CU (150 KB of LDS), operands = the three bf16 planes of random fp32 values, no global traffic in the loop.
//   body 0: the kernel's current body -- 4 x 2 blocks of v_mfma_f32_32x32x16_bf16, six plane products per block and 16-deep k step
//           (12 + 6 ds_read_b128 per 48 MFMAs of 32 cycles);
//   body 1: DESIGN.md section 7 item 5 -- 8 x 4 blocks of v_mfma_f32_16x16x32_bf16, planes paired along k:
//           [a_lo | a_hi] . [b_hi | b_lo], [a_mid | a_mid] . [b_mid | b_hi], [a_hi | a_hi] . [b_mid | b_hi]
//           (24 + 8 ds_read_b128 per 96 MFMAs of 16 cycles: 1.8x the LDS reads per FLOP).
// Prints TFLOP/s of bf16 MFMA work, the fp32-equivalent rate (/ 6) and the shader clock held.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/x3_body_shapes tools/probes/x3_body_shapes.hip && /tmp/x3_body_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int LDS_BYTES = 150 * 1024;
constexpr int STAGE = 48 * 1024;          // one k step's operand image: 3 planes x (256 rows A + 256 rows B) x 16 k x 2 B = 48 KB

__device__ __forceinline__ bf16x8 lds_read(const char* base, int off) { return *(const bf16x8*)(base + off); }

template <int BODY>
__global__ __launch_bounds__(512, 1) void loop(const unsigned* in, float* out, int iters, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves of 128 x 64
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += 512) ((unsigned*)smem)[i] = in[i % (3 * STAGE / 4)];
    __syncthreads();
    long long t0 = 0, w0 = 0;
    if (threadIdx.x == 0) { w0 = wall_clock64(); t0 = clock64(); }
    // operand image of a stage: plane p of A at p * 8 KB (256 rows x 32 B), plane p of B at 24 KB + p * 8 KB; a fragment read is
    // 64 lanes x 16 B = one contiguous KB (conflict-free), the row block selects the KB
    float s = 0.f;
    if constexpr (BODY == 0) {
        f32x16 acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int aoff = wm * 4096 + lane * 16, boff = 24576 + wn * 2048 + lane * 16;
        for (int it = 0; it < iters; ++it) {
            const char* st = smem + (it % 3) * STAGE;
            bf16x8 fb[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[p][j] = lds_read(st, boff + p * 8192 + j * 1024);
            bf16x8 fa[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[0][p] = lds_read(st, aoff + p * 8192);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i + 1 < 4) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) fa[(i + 1) & 1][p] = lds_read(st, aoff + p * 8192 + (i + 1) * 1024);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // planes 0 / 1 / 2 = hi / mid / lo: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi (small terms first)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 1][2], fb[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 1][0], fb[2][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 1][1], fb[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 1][1], fb[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 1][0], fb[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 1][0], fb[0][j], acc[i][j], 0, 0, 0);
                }
            }
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    } else {
        f32x4 acc[8][4];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // lane = (row l15, k group l4 of 8 values): k groups 0, 1 read the first plane of a pair, groups 2, 3 the second, both at the
        // SAME 16 k values -> a lane-dependent plane base, still one ds_read_b128
        const int l15 = lane & 15, l4 = lane >> 4, kh = l4 >> 1, kg = l4 & 1;
        const int rowb = l15 * 32 + kg * 16;                       // byte offset of the lane's 8 k values inside a 16-row block
        // A pairs: [lo | hi], [mid | mid], [hi | hi]; B pairs: [hi | lo], [mid | hi]
        const int apl[3] = {kh ? 0 : 2, 1, 0}, bpl[2] = {kh ? 2 : 0, kh ? 0 : 1};
        int aoff[3], boff[2];
        for (int q = 0; q < 3; ++q) aoff[q] = wm * 4096 + apl[q] * 8192 + rowb;
        for (int q = 0; q < 2; ++q) boff[q] = 24576 + wn * 2048 + bpl[q] * 8192 + rowb;
        for (int it = 0; it < iters; ++it) {
            const char* st = smem + (it % 3) * STAGE;
            bf16x8 fb[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[q][j] = lds_read(st, boff[q] + j * 512);
            bf16x8 fa[2][3];
#pragma unroll
            for (int q = 0; q < 3; ++q) fa[0][q] = lds_read(st, aoff[q]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i + 1 < 8) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) fa[(i + 1) & 1][q] = lds_read(st, aoff[q] + (i + 1) * 512);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i & 1][0], fb[0][j], acc[i][j], 0, 0, 0);   // lo.hi + hi.lo
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i & 1][1], fb[1][j], acc[i][j], 0, 0, 0);   // mid.mid + mid.hi
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i & 1][2], fb[1][j], acc[i][j], 0, 0, 0);   // hi.mid + hi.hi
                }
            }
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = clock64() - t0; clk[blockIdx.x * 2 + 1] = wall_clock64() - w0; }
}

static unsigned short bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int grid = 256 * 4;
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    unsigned* in; float* out; long long* clk;
    hipMalloc(&in, 3 * STAGE); hipMalloc(&out, grid * 512 * 4); hipMalloc(&clk, grid * 16);
    // three stages of plane triples: stage layout [A hi | A mid | A lo | B hi | B mid | B lo], 8 KB each, random fp32 values split
    std::vector<unsigned short> h(3 * STAGE / 2);
    for (int st = 0; st < 3; ++st)
        for (int ab = 0; ab < 2; ++ab)
            for (int e = 0; e < 4096; ++e) {
                const float v = ((float)rand() / RAND_MAX * 2.f - 1.f) * (1.f + (rand() & 7));
                const unsigned short hi = bf16_rne(v);
                const float r1 = v - bf16_f(hi);
                const unsigned short mid = bf16_rne(r1);
                const unsigned short lo = bf16_rne(r1 - bf16_f(mid));
                const int base = st * (STAGE / 2) + ab * 3 * 4096;
                h[base + e] = hi; h[base + 4096 + e] = mid; h[base + 8192 + e] = lo;
            }
    hipMemcpy(in, h.data(), 3 * STAGE, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)loop<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)loop<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    for (int round = 0; round < 2; ++round)
        for (int body = 0; body < 2; ++body) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (body == 0) hipLaunchKernelGGL(loop<0>, dim3(grid), dim3(512), LDS_BYTES, 0, in, out, iters, clk);
                else hipLaunchKernelGGL(loop<1>, dim3(grid), dim3(512), LDS_BYTES, 0, in, out, iters, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                // per wave and k step: 128 x 64 x 16 x 2 FLOP x 6 plane products
                const double flop = (double)grid * 8 * iters * (128.0 * 64 * 16 * 2 * 6);
                std::vector<long long> c(grid * 2); hipMemcpy(c.data(), clk, grid * 16, hipMemcpyDeviceToHost);
                double ghz = 0; for (int i = 0; i < grid; ++i) ghz += (double)c[2 * i] / ((double)c[2 * i + 1] * 10.0); ghz /= grid;
                const double tf = flop / ms / 1e9;
                // matrix-rate fraction at the clock held: 1024 FLOP per cycle and SIMD
                printf("%s rep %d: %8.2f ms  %7.1f TFLOP/s bf16 MFMA = %6.1f fp32-equivalent  clock %.2f GHz  %.1f %% of the matrix rate at that clock\n",
                       body == 0 ? "32x32x16 six products " : "16x16x32 paired planes", rep, ms, tf, tf / 6, ghz,
                       100.0 * tf * 1e12 / (256.0 * 4 * 1024 * ghz * 1e9));
            }
        }
    return 0;
}
