// Probe only (tools/gemm_prefetch_probe.py): read a buffer with a FEW workgroups so that its lines sit in the Infinity Cache when the kernel that needs them
// starts.  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/touch.hip -o ltx-2-mlx_amd/lib/ab/touch.so
#include <hip/hip_runtime.h>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void touch_kernel(const u32x4_t* __restrict__ p, long n16, unsigned* __restrict__ sink) {
    u32x4_t acc = {0u, 0u, 0u, 0u};
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {        // 8 independent 16-byte loads per lane in flight
        u32x4_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(p + i + j * stride);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
    for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(p + i);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) *sink = acc.x;       // (never in practice: keeps the loads alive)
}
extern "C" int touch_launch(const void* p, long bytes, int blocks, void* sink, void* stream) {
    hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)p, bytes / 16, (unsigned*)sink);
    return (int)hipGetLastError();
}
