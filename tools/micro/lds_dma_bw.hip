// L2 -> CU load throughput per path, per CU: (a) LDS-DMA (global_load_lds, 16 B/lane), (b) global_load_dwordx4 into
// VGPRs, (c) like (b) followed by ds_write_b128 into LDS.  Every workgroup re-reads its own 64-KiB window (L2-resident
// after the first pass), 8 waves per CU, 8 loads in flight per wave between waits.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/lds_dma_bw.hip -o /tmp/lds_dma_bw && /tmp/lds_dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void bw_kernel(const char* __restrict__ src, unsigned* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * 65536 + wv * 8192 + lane * 16;     // 8 KiB per wave = 8 x 1 KiB pieces
    char* lbase = smem + wv * 8192;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) __builtin_amdgcn_global_load_lds((const void*)(base + j * 1024), (lds_ptr_t)(lbase + j * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const volatile u32x4*)(base + j * 1024);
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) *(u32x4*)(lbase + j * 1024 + lane * 16) = v[j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
    }
    if (MODE == 0) acc[0] = *(unsigned*)(lbase + lane * 4);
    sink[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    char* src; unsigned* sink;
    hipMalloc(&src, 256 * 65536); hipMemset(src, 1, 256 * 65536);
    hipMalloc(&sink, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    auto run = [&](auto kern, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536, 0, src, sink, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536, 0, src, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 256.0 * 65536 * iters;
        const double bps = bytes / (ms * 1e-3);
        printf("%-34s %7.2f TB/s aggregate = %5.1f B/clk/CU at 2.4 GHz\n", name, bps / 1e12, bps / 256 / 2.4e9);
    };
    run(bw_kernel<0>, "global_load_lds b128 (LDS-DMA)");
    run(bw_kernel<1>, "global_load_dwordx4 -> VGPR");
    run(bw_kernel<2>, "global_load_dwordx4 + ds_write_b128");
    return 0;
}
