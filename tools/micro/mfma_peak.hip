// Sustained dense bf16 MFMA rate of the whole chip with NO memory traffic: every wave loops over independent
// v_mfma_f32_32x32x16_bf16 on register operands.  Calibrates "peak" under the power/clock the GPU actually
// sustains (the 2.5 PFLOP/s figure assumes 2.4 GHz on all 1024 matrix pipes).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void mfma_loop(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(1.0f - e * 0.01f); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // waves per SIMD x independent accumulators per wave (a single wave issuing alone is the ping-pong GEMM's case)
    auto run = [&](auto kern, int nacc, int threads, int blocks_per_cu) {
        const int grid = 256 * blocks_per_cu, iters = 100000;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, 1000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 32 * 32 * 16 * (double)nacc * iters * (grid * (threads / 64.0));
        printf("waves/SIMD=%.0f  independent accumulators=%d : %7.1f TFLOP/s\n", grid * (threads / 64.0) / 1024.0, nacc, flops / ms / 1e9);
    };
    run(mfma_loop<2>, 2, 256, 1);
    run(mfma_loop<4>, 4, 256, 1);
    run(mfma_loop<8>, 8, 256, 1);
    run(mfma_loop<2>, 2, 512, 1);
    run(mfma_loop<4>, 4, 512, 1);
    run(mfma_loop<8>, 8, 512, 1);
    run(mfma_loop<4>, 4, 512, 2);
    return 0;
}
