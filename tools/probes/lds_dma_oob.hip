#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// probe: what does an out-of-range buffer_load ... lds write into LDS?
__global__ void probe(const float* src, unsigned bytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
    int off = threadIdx.x * 16;
    if (threadIdx.x & 1) off |= 0x80000000;   // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    float *src, *out; 
    hipMalloc(&src, 4096); hipMalloc(&out, 1024);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, 4096u, out);
    float o[256]; hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) printf("%g ", o[i]); printf("\n");
    return 0;
}
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma_oob tools/probes/lds_dma_oob.hip && /tmp/lds_dma_oob
// expected if an out-of-range LDS-DMA lane writes zeros: "0 1 2 3 0 0 0 0 8 9 10 11 0 0 0 0 ..."; "-7" = the lane's LDS bytes were left untouched
