// Do MFMA and VALU streams of the two waves of a SIMD overlap?  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) issue MFMAs from
// registers, waves 4-7 (their SIMD partners) issue a softmax-like VALU stream (v_fma_f32 + v_exp_f32 + v_add_f32).  Times: MFMA waves alone,
// VALU waves alone, both together -- for v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16, accumulators in VGPRs.  Also the SAME wave
// issuing both streams interleaved (one wave per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: MFMA stream, bit 1: VALU stream.  SAME = both streams in every wave (256 threads / block), else split by wave half (512 threads)
template <int SHAPE, int MODE, bool SAME>
__global__ __launch_bounds__(SAME ? 256 : 512) void k(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    const bool mfma_wave = SAME || threadIdx.x < 256, valu_wave = SAME || threadIdx.x >= 256;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(size_t)(2 * i) * 65536 + (tid & 65535)];
        b[i] = src[(size_t)(2 * i + 1) * 65536 + (tid & 65535)];
    }
    float sum = 0.f;
    f32x16 acc32[4];
    f32x4 acc16[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (float)(tid & 255) + (float)i;
    const float c = 0.999f, d = -0.0001f;
    // role-specialised loops (one wave-uniform branch up front: a per-MFMA `if` costs a scalar branch each and hides the effect under test)
    const int role = __builtin_amdgcn_readfirstlane(SAME ? 3 : (threadIdx.x < 256 ? 1 : 2)) & MODE;
    auto mfma_step = [&](int j) __attribute__((always_inline)) {
        if (SHAPE == 32) {
            acc32[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3], b[(j >> 2) & 3], acc32[j & 3], 0, 0, 0);
        } else {
            acc16[(2 * j) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j >> 2) & 3], acc16[(2 * j) & 15], 0, 0, 0);
            acc16[(2 * j + 1) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(j + 1) & 3], b[(j >> 2) & 3], acc16[(2 * j + 1) & 15], 0, 0, 0);
        }
    };
    auto valu_step = [&](int j) __attribute__((always_inline)) {
        float e = __builtin_fmaf(x[j], c, d);
        float p = __builtin_amdgcn_exp2f(e);
        asm volatile("" : "+v"(p));
        x[j] = __builtin_fmaf(p, 0.25f, x[j] * 0.5f);
    };
    (void)mfma_wave; (void)valu_wave;
    if (role == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                mfma_step(j);
                valu_step(j);
            }
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) mfma_step(j);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) valu_step(j);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc32[i][r];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += acc16[i][r];
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += x[i];
    out[tid] = sum;
}

template <int SHAPE, int MODE, bool SAME>
static float run(const bf16x8* src, float* out, int iters, int blocks = 256) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = SAME ? 256 : 512;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<SHAPE, MODE, SAME>), dim3(blocks), dim3(threads), 0, 0, src, out, iters);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<SHAPE, MODE, SAME>), dim3(blocks), dim3(threads), 0, 0, src, out, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t n = (size_t)8 * 65536;
    std::vector<unsigned short> h(n * 8);
    unsigned long long s = 1234567;
    for (auto& v : h) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; v = (unsigned short)(0x3c00 + ((s >> 40) & 0x3ff)) ^ (unsigned short)(((s >> 20) & 1) << 15); }
    bf16x8* src; float* out;
    hipMalloc(&src, n * 16); hipMalloc(&out, (size_t)256 * 512 * 4);
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    const int iters = 20000;
    const double mf = 2.0 * 32 * 32 * 16 * 16.0 * iters * 256 * 4;     // flops of the MFMA waves (4 per CU)
#define ROW(SHAPE, SAME, label)                                                                                               \
    {                                                                                                                         \
        const float tm = run<SHAPE, 1, SAME>(src, out, iters), tv = run<SHAPE, 2, SAME>(src, out, iters), tb = run<SHAPE, 3, SAME>(src, out, iters); \
        printf("%-46s MFMA %7.2f ms (%6.0f TF/s)  VALU %7.2f ms  both %7.2f ms  = %.2f x max, %.2f x sum\\n", label, tm, mf / tm / 1e9, tv, tb, tb / (tm > tv ? tm : tv), tb / (tm + tv)); \
    }
    ROW(32, false, "32x32x16: MFMA wave + VALU partner per SIMD")
    ROW(16, false, "16x16x32: MFMA wave + VALU partner per SIMD")
    ROW(32, true, "32x32x16: both streams in ONE wave per SIMD")
    ROW(16, true, "16x16x32: both streams in ONE wave per SIMD")
    {   // two such waves per SIMD (512 blocks of 256 threads): twice the work
        const float tm = run<32, 1, true>(src, out, iters, 512), tv = run<32, 2, true>(src, out, iters, 512), tb = run<32, 3, true>(src, out, iters, 512);
        printf("%-46s MFMA %7.2f ms (%6.0f TF/s)  VALU %7.2f ms  both %7.2f ms  = %.2f x max, %.2f x sum\n", "32x32x16: both streams per wave, TWO waves/SIMD", tm, 2 * mf / tm / 1e9, tv, tb, tb / (tm > tv ? tm : tv), tb / (tm + tv));
        const float tm2 = run<16, 1, true>(src, out, iters, 512), tv2 = run<16, 2, true>(src, out, iters, 512), tb2 = run<16, 3, true>(src, out, iters, 512);
        printf("%-46s MFMA %7.2f ms (%6.0f TF/s)  VALU %7.2f ms  both %7.2f ms  = %.2f x max, %.2f x sum\n", "16x16x32: both streams per wave, TWO waves/SIMD", tm2, 2 * mf / tm2 / 1e9, tv2, tb2, tb2 / (tm2 > tv2 ? tm2 : tv2), tb2 / (tm2 + tv2));
    }
    return 0;
}
