// Timing harness for the gemm_sp experiment (tools/micro/gemm_sp.hip), with its ablation knobs:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ltx-2-mlx_amd/csrc -DSP_ABL=n tools/micro/gemm_sp_probe.hip -o sp && ./sp M N K waves
//   SP_ABL: 0 full kernel, 1 no staging, 2 staging from an L2-hot tile, 3 no barrier, 4 MFMA stream only,
//           5 plain global_load_dwordx4 to VGPRs instead of LDS-DMA, 6 buffer_load..lds instead of global_load_lds
#include "gemm_sp.hip"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
void ltx2_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    const int waves = argc > 4 ? atoi(argv[4]) : 4;
    bf16 *a, *w, *o;
    hipMalloc(&a, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&o, (size_t)M * N * 2);
    // pseudo-random bit patterns (finite bf16 values): data toggling matters for the clock the chip sustains
    {
        size_t na = (size_t)M * K, nw = (size_t)N * K;
        unsigned short* h = (unsigned short*)malloc((na > nw ? na : nw) * 2);
        unsigned s = 12345;
        for (size_t i = 0; i < na; ++i) { s = s * 1664525u + 1013904223u; h[i] = (unsigned short)(((s >> 16) & 0x807f) | (((s >> 9) & 7) + 0x3c) << 7); }
        hipMemcpy(a, h, na * 2, hipMemcpyHostToDevice);
        for (size_t i = 0; i < nw; ++i) { s = s * 1664525u + 1013904223u; h[i] = (unsigned short)(((s >> 16) & 0x807f) | (((s >> 9) & 7) + 0x38) << 7); }
        hipMemcpy(w, h, nw * 2, hipMemcpyHostToDevice);
        free(h);
    }
    GemmParams p{};
    p.A = a; p.W = w; p.out = o; p.lda = K; p.ldo = N; p.M = M; p.N = N; p.K = K;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) if (gemm_sp_launch(p, EPI_BF16, waves, 0)) return 1;
    hipEventRecord(e0, 0);
    const int it = 20;
    for (int i = 0; i < it; ++i) gemm_sp_launch(p, EPI_BF16, waves, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
#ifndef SP_ABL
#define SP_ABL 0
#endif
    printf("ABL=%d waves=%d M=%d N=%d K=%d: %.1f us  %.1f TF/s\n", SP_ABL, waves, M, N, K, ms / it * 1e3, 2.0 * M * N * K / (ms / it * 1e-3) / 1e12);
    return 0;
}
