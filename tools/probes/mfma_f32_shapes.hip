// Probe: sustained fp32 MFMA rate by instruction shape under load (random operands, registers only, one or two waves per SIMD).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32_shapes tools/probes/mfma_f32_shapes.hip && /tmp/mfma_f32_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>   // 0: 32x32x2, 1: 16x16x4
__global__ __launch_bounds__(256) void loop(const float* in, float* out, int iters, long long* clk) {
    const int lane = threadIdx.x & 63;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) % 4096]; b[i] = in[(threadIdx.x * 8 + i + 1777) % 4096]; }
    long long t0 = 0, w0 = 0;
    if (threadIdx.x == 0) { w0 = wall_clock64(); t0 = clock64(); }
    if (SHAPE == 0) {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + 1) & 7], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 1) & 7], b[k], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 1) & 7], b[(k + 1) & 7], acc[3], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 acc[16];
        for (int j = 0; j < 16; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + (j >> 2)) & 7], b[(k + j) & 7], acc[j], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = clock64() - t0; clk[blockIdx.x * 2 + 1] = wall_clock64() - w0; }
    (void)lane;
}

int main() {
    float *in, *out; long long* clk;
    const int grid = 512;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, grid * 256 * 4); hipMalloc(&clk, grid * 16);
    std::vector<float> h(4096); for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int shape = 0; shape < 2; ++shape) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(loop<0>, dim3(grid), dim3(256), 0, 0, in, out, iters, clk);
            else hipLaunchKernelGGL(loop<1>, dim3(grid), dim3(256), 0, 0, in, out, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // FLOP: shape 0: 32 MFMAs x 4096 per iter per wave; shape 1: 64 MFMAs x 2048
            const double flop = (double)grid * 4 * iters * 32.0 * 4096.0;
            std::vector<long long> c(grid * 2); hipMemcpy(c.data(), clk, grid * 16, hipMemcpyDeviceToHost);
            double ghz = 0; for (int i = 0; i < grid; ++i) ghz += (double)c[2 * i] / ((double)c[2 * i + 1] * 10.0); ghz /= grid;
            printf("%s rep %d: %.2f ms  %.1f TFLOP/s  shader clock %.2f GHz\n", shape == 0 ? "32x32x2 f32" : "16x16x4 f32", rep, ms, flop / ms / 1e9, ghz);
        }
    }
    return 0;
}
