// Standalone timing harness for attn_fwd_kernel (hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DAT_DK=n -DAT_DV=n] tools/micro/attn_probe.hip -o ap; ./ap Nkv [Nq]).
// Constant data: the chip sustains a higher clock on it than on random data (about +15% here), so compare variants, not absolutes.
#include "../../ltx-2-mlx_amd/csrc/attention.hip"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
void ltx2_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 3456, H = 32, HD = 128, D = H * HD;
    const int Nq = argc > 2 ? atoi(argv[2]) : N;
    const int Npad = (N + 63) / 64 * 64;
    bf16 *q, *k, *vt, *o;
    hipMalloc(&q, (size_t)N * D * 2); hipMalloc(&k, (size_t)N * D * 2); hipMalloc(&o, (size_t)N * D * 2);
    hipMalloc(&vt, (size_t)H * HD * Npad * 2);
    hipMemset(q, 0x3c, (size_t)N * D * 2); hipMemset(k, 0x3c, (size_t)N * D * 2); hipMemset(vt, 0x3c, (size_t)H * HD * Npad * 2);
    AttnParams p{q, k, vt, o, D, D, D, (long)HD * Npad, Nq, N, Npad, H, HD, 0.088388f * 1.442695f};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) if (attn_launch(p, 0)) return 1;
    hipEventRecord(a, 0);
    const int it = 20;
    for (int i = 0; i < it; ++i) attn_launch(p, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int occ = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, attn_fwd_kernel<128>, 256, Geo<128>::LDS_BYTES);
    printf("DK=%d DV=%d Nkv=%d Nq=%d occ=%d: %.1f us  %.1f TF/s\n", AT_DK, AT_DV, N, Nq, occ, ms / it * 1e3, 4.0 * Nq * N * D / (ms / it * 1e-3) / 1e12);
    return 0;
}
