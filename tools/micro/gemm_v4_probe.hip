// A/B harness for the hand-scheduled 4-wave GEMM (ltx-2-mlx_amd/csrc/gemm_v4.hip) against the 8-wave ping-pong kernel
// (gemm_pp.hip) on the DiT shapes, with the K-loop ablations the generator can emit.  Not part of libltx2hip.so.
//   python3 ltx-2-mlx_amd/csrc/gen_gemm_v4.py ltx-2-mlx_amd/csrc/gemm_v4_loop.inc --probe
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLTX2_V4_PROBE -I ltx-2-mlx_amd/csrc tools/micro/gemm_v4_probe.hip -o build/gemm_v4_probe
//   ./build/gemm_v4_probe            (all shapes)      ./build/gemm_v4_probe M N K
#include "gemm_pp.hip"
#include "gemm_v4.hip"
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

void ltx2_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }

static unsigned short f2bf_host(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f_host(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Rng { unsigned long long s; float uni() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (float)((s >> 40) & 0xffffff) / 16777216.f; }
             float gauss() { float a = 0; for (int i = 0; i < 4; ++i) a += uni(); return (a - 2.f) * 1.7320508f; } };

static unsigned long long* g_dbg = nullptr;
static float time_launch(int (*fn)(const GemmParams&, void*), const GemmParams& p, void* ctx, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) fn(p, ctx);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) fn(p, ctx);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / iters * 1e3f;
}
struct Ctx { int kind, layout, arg, epi; };     // kind 0: ping-pong; 1: ablation (arg = variant); 2: v4 (arg = bm, 0 = auto)
static int run(const GemmParams& p, void* c) {
    const Ctx* x = (const Ctx*)c;
    if (x->kind == 0) return gemm_pp_launch(p, x->epi, false, 0);
    if (x->kind == 1) return gemm_v4_probe_launch(p, x->layout, x->arg, 0);
    return gemm_v4_launch(p, x->epi, 0, x->layout, x->arg);
}

static void one_shape(int M, int N, int K, bool ablations) {
    const size_t na = (size_t)M * K, nw = (size_t)N * K, no = (size_t)M * N;
    std::vector<unsigned short> ha(na), hw(nw);
    Rng r{0x1234567ULL + (unsigned long long)M * 31 + K};
    const float ws = 1.f / sqrtf((float)K);
    for (auto& v : ha) v = f2bf_host(r.gauss());
    for (auto& v : hw) v = f2bf_host(r.gauss() * ws);
    bf16 *a, *w, *o_ref, *o_new; float *x0, *x1, *gate;
    hipMalloc(&a, na * 2); hipMalloc(&w, nw * 2); hipMalloc(&o_ref, no * 2); hipMalloc(&o_new, no * 2);
    hipMalloc(&x0, no * 4); hipMalloc(&x1, no * 4); hipMalloc(&gate, (size_t)N * 4);
    hipMemcpy(a, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
    { std::vector<float> g(N); for (auto& v : g) v = r.gauss(); hipMemcpy(gate, g.data(), (size_t)N * 4, hipMemcpyHostToDevice); }
    GemmParams p{};
    p.A = a; p.W = w; p.lda = K; p.ldo = N; p.M = M; p.N = N; p.K = K;
    const double flop = 2.0 * M * N * K;
    printf("== M=%d N=%d K=%d  (%.1f GF)\n", M, N, K, flop / 1e9);

    // ---- correctness: v4 (both row-block counts) vs ping-pong, bit for bit, plus sampled fp64 host check ----
    std::vector<unsigned short> href(no), hnew(no);
    p.out = o_ref; hipMemset(o_ref, 0, no * 2);
    Ctx cpp{0, 0, 0, EPI_BF16}; run(p, &cpp); hipDeviceSynchronize();
    hipMemcpy(href.data(), o_ref, no * 2, hipMemcpyDeviceToHost);
    for (int cfg = 0; cfg < 8; ++cfg) {
        const int layout = cfg >> 1, rb = (cfg & 1) ? 256 : 224;
        p.out = o_new; hipMemset(o_new, 0xff, no * 2);
        Ctx c{2, layout, rb, EPI_BF16}; int rc = run(p, &c); hipError_t e = hipDeviceSynchronize();
        hipMemcpy(hnew.data(), o_new, no * 2, hipMemcpyDeviceToHost);
        size_t bad = 0; double maxd = 0; size_t first = (size_t)-1;
        for (size_t i = 0; i < no; ++i) if (href[i] != hnew[i]) { if (!bad) first = i; ++bad; maxd = std::max(maxd, (double)fabsf(bf2f_host(href[i]) - bf2f_host(hnew[i]))); }
        double maxe = 0;
        Rng s{99};
        for (int t = 0; t < 400; ++t) {
            const int m = (int)(s.uni() * M) % M, n = (int)(s.uni() * N) % N;
            double acc = 0;
            for (int k = 0; k < K; ++k) acc += (double)bf2f_host(ha[(size_t)m * K + k]) * (double)bf2f_host(hw[(size_t)n * K + k]);
            maxe = std::max(maxe, fabs(acc - (double)bf2f_host(hnew[(size_t)m * N + n])) / (fabs(acc) + 1.0));
        }
        printf("   v4 layout %d BM %d rc=%d hip=%d: %zu / %zu elements differ from ping-pong (max |d| %.3g, first at row %zu col %zu); sampled rel err vs fp64 %.3g\n",
               layout, rb, rc, (int)e, bad, no, maxd, first == (size_t)-1 ? 0 : first / N, first == (size_t)-1 ? 0 : first % N, maxe);
    }
    // gated fp32 residual epilogue: x += gate_table * (acc + 0)
    {
        std::vector<float> hx(no); for (auto& v : hx) v = r.gauss();
        hipMemcpy(x0, hx.data(), no * 4, hipMemcpyHostToDevice); hipMemcpy(x1, hx.data(), no * 4, hipMemcpyHostToDevice);
        GemmParams q = p; q.gate_table = gate;
        q.out = x0; Ctx c0{0, 0, 0, EPI_RESID_GATE_F32}; run(q, &c0);
        q.out = x1; Ctx c1{2, 1, 0, EPI_RESID_GATE_F32}; run(q, &c1);
        hipDeviceSynchronize();
        std::vector<float> h0(no), h1(no);
        hipMemcpy(h0.data(), x0, no * 4, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), x1, no * 4, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < no; ++i) bad += memcmp(&h0[i], &h1[i], 4) != 0;
        printf("   gated fp32 residual epilogue: %zu / %zu elements differ\n", bad, no);
    }

    // ---- timing: interleaved rounds, best and median of 5 ----
    struct V { const char* name; Ctx c; void* out; std::vector<float> t; unsigned long long cyc[3] = {0, 0, 0}; };
    if (!g_dbg) hipMalloc(&g_dbg, 64);
    std::vector<V> vs;
    vs.push_back({"pp  auto  bf16", {0, 0, 0, EPI_BF16}, o_ref});
    vs.push_back({"v4 L14 auto bf16", {2, 0, 0, EPI_BF16}, o_new});
    vs.push_back({"v4 L22 auto bf16", {2, 1, 0, EPI_BF16}, o_new});
    vs.push_back({"v4 L22 256  bf16", {2, 1, 256, EPI_BF16}, o_new});
    vs.push_back({"v4 M16 auto bf16", {2, 2, 0, EPI_BF16}, o_new});
    vs.push_back({"v4 M16 256  bf16", {2, 2, 256, EPI_BF16}, o_new});
    vs.push_back({"v4 M16x14 auto bf16", {2, 3, 0, EPI_BF16}, o_new});
    vs.push_back({"v4 M16x14 256 bf16", {2, 3, 256, EPI_BF16}, o_new});
    vs.push_back({"pp  auto  gelu", {0, 0, 0, EPI_GELU_BF16}, o_ref});
    vs.push_back({"v4 M16x14 auto gelu", {2, 3, 0, EPI_GELU_BF16}, o_new});
    vs.push_back({"pp  auto  resid", {0, 0, 0, EPI_RESID_GATE_F32}, x0});
    vs.push_back({"v4 L14 auto resid", {2, 0, 0, EPI_RESID_GATE_F32}, x1});
    vs.push_back({"v4 L22 auto resid", {2, 1, 0, EPI_RESID_GATE_F32}, x1});
    vs.push_back({"v4 M16 auto resid", {2, 2, 0, EPI_RESID_GATE_F32}, x1});
    vs.push_back({"v4 M16x14 auto resid", {2, 3, 0, EPI_RESID_GATE_F32}, x1});
    if (ablations) {
        vs.push_back({"v4 M16 256 no-DMA", {1, 2, 1, EPI_BF16}, o_new});
        vs.push_back({"v4 M16 256 no-read", {1, 2, 2, EPI_BF16}, o_new});
        vs.push_back({"v4 M16x14 224 direct epi", {1, 3, 9, EPI_BF16}, o_new});
        vs.push_back({"v4 M16x14 224 rd/2", {1, 3, 10, EPI_BF16}, o_new});
        vs.push_back({"v4 M16x14 224 dma/4", {1, 3, 11, EPI_BF16}, o_new});
    }
    for (int round = 0; round < 5; ++round)
        for (auto& v : vs) {
            GemmParams q = p; q.out = v.out; q.gate_table = gate; q.dbg = g_dbg;
            hipMemset(g_dbg, 0, 24);
            v.t.push_back(time_launch(run, q, &v.c, 20));
            hipMemcpy(v.cyc, g_dbg, 24, hipMemcpyDeviceToHost);
        }
    for (auto& v : vs) {
        std::sort(v.t.begin(), v.t.end());
        printf("   %-26s best %7.1f us (%6.1f TF/s)   median %7.1f us (%6.1f TF/s)   setup %6llu loop %7llu (%6.1f / K-tile) epilogue %6llu block %7llu cyc -> %.2f GHz\n", v.name, v.t[0], flop / v.t[0] / 1e6, v.t[2],
               flop / v.t[2] / 1e6, v.cyc[2], v.cyc[0], (double)v.cyc[0] / (K / 64), v.cyc[1] - v.cyc[0] - v.cyc[2], v.cyc[1], v.cyc[1] / v.t[2] / 1e3);
    }
    hipFree(a); hipFree(w); hipFree(o_ref); hipFree(o_new); hipFree(x0); hipFree(x1); hipFree(gate);
}

int main(int argc, char** argv) {
    if (argc >= 4) { one_shape(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), true); return 0; }
    one_shape(4096, 4096, 4096, true);
    one_shape(3456, 4096, 4096, true);
    one_shape(3456, 4096, 16384, false);
    one_shape(3456, 12288, 4096, false);
    one_shape(3456, 16384, 4096, false);
    one_shape(13824, 4096, 4096, false);
    return 0;
}
