// bf16 MFMA GEMM / implicit-GEMM conv3d for gfx950: parameter block + host launchers.
#pragma once
#include "common.h"

enum GemmEpilogue {
    EPI_BF16 = 0,            // out_bf16 = acc + bias
    EPI_GELU_BF16 = 1,       // out_bf16 = gelu_tanh(acc + bias)
    EPI_SILU_BF16 = 2,       // out_bf16 = silu(acc + bias)
    EPI_F32 = 3,             // out_f32  = acc + bias
    EPI_RESID_GATE_F32 = 4,  // out_f32[m][n] += (gate[m*gate_stride+n] + gate_table[n]) * (acc + bias)
    EPI_ADD_BF16 = 5,        // out_bf16 = acc + bias + res_bf16[m*ldres + n]
    EPI_D2S_BF16 = 6,        // conv only: depth-to-space scatter (+ tiled d2s(x) residual)
    EPI_COUNT = 7
};

struct GemmParams {
    const bf16* A;       // dense: [M][lda];  conv: activations [T][H][W][Cin] (channels-last)
    const bf16* W;       // [N][K]  (K contiguous; conv: K = tap*Cin + c, tap = (kt*3+kh)*3+kw)
    const unsigned char* W8;  // fp8-resident weights (gemm_v4.hip): e4m3fn codes [N][K] instead of W, with
    const float* wscale;      //   one dequantisation scale per output column: w = bf16(f32(code) * wscale[n])
    const unsigned char* A8;  // fp8 COMPUTE (gemm_v4.hip layout 5): e4m3fn activation codes [M][lda] instead of A, with one scale per
    const float* ascale;      //   row; needs W8 / wscale too: out = epilogue(ascale[m] * wscale[n] * sum_k a8[m][k] * w8[n][k] + bias)
    const float* bias;   // [N] or null
    void* out;           // bf16 or f32, [M][ldo]  (D2S: [To][Ho][Wo][Cf])
    const float* gate;   // EPI_RESID_GATE_F32: per-row part  gate[m*gate_stride + n]  (may be null)
    const float* gate_table;  // EPI_RESID_GATE_F32: broadcast part gate_table[n]  (may be null; both null -> 1)
    const bf16* res;     // EPI_ADD_BF16
    long lda, ldo, gate_stride, ldres;
    int M, N, K;
    // conv geometry (implicit GEMM): M = T*H*Wd output positions, stride 1, 3x3x3
    int T, H, Wd, Cin, cin_shift, pad_front;
    int taps_t;          // temporal kernel size: 3 (3x3x3) or 1 (per-frame 3x3); K = 9*taps_t*Cin
    int pad_zero;        // 0: reflect H/W + replicate T (VAE decoder); 1: zero padding in T/H/W (spatial upscaler);
                         // 2: zero padding in H/W + replicate T (VAE encoder)
    // depth-to-space epilogue: column n = s*Cf + c, s = (a*fh + b)*fw + d
    int ft, fh, fw, Cf, cf_shift, drop_first, d2s_residual, c_d2s;
    // gemm_v4.hip, EPI_BF16, dense, layout 3: output columns >= vt_col0 (the V third of a fused QKV projection) are written
    // TRANSPOSED and key-permuted as attention's V^T operand vt[h][vt_hd][vt_npad] (attention.hip, vt_transpose_kernel's
    // layout: rows >= M zero-filled up to vt_npad) instead of into `out`; null = off.  gemm_v4_vt_supported() says when.
    bf16* vt;
    long vt_head_stride;
    int vt_col0, vt_npad, vt_hd;
    // gemm_v4.hip, EPI_BF16, dense, layouts 3 / 5: rowss[m * (N / 64) + n / 64] = sum over the 64-column strip of out[m][n]^2 (of the
    // ROUNDED outputs): the partial sums of a row's squared norm, for a consumer that folds an RMS normalisation of `out` into its own
    // arithmetic (text cross-attention: q_norm as a per-row softmax scale).  null = off.  gemm_rowss_supported() says when.
    float* rowss;
    // ---- round 6: the plain RMS normalisation in front of a projection folded around the GEMMs (gemm_v4.hip layout 3, dense bf16 weights; gemm_fold_supported()) ----
    // rms_norm(x) W^T + b equals r[m] (x W^T) + b with r[m] = rsqrt(mean x[m]^2 + eps): the PRODUCER of x (the gated-residual epilogue that forms the new
    // residual row) also leaves y = bf16(x_new) and the partial sums of squares of x_new (one per 256-column tile); the CONSUMER forms the row factors of its
    // tile's rows from those partials (fetched into LDS in front of its K loop) and multiplies its accumulators by them -- the stand-alone norm pass and the
    // fp32 re-read of x between the two GEMMs are gone.  (A scale (1 + s) rides on the shadow; a shift t would need t W^T per step: measured, not kept -- DESIGN.md.)
    // Producer (EPI_RESID_GATE_F32, row-invariant gate):
    bf16* shadow;              //   shadow[m][n] = bf16(x_new[m][n] * (1 + shadow_scale[n])), row stride ld_shadow; null = off
    const float* shadow_scale; //   [N] or null (plain RMS norm: * 1)
    float* shadow_ss;          //   shadow_ss[(n / 256) * ld_ss + m] = sum of x_new[m][n]^2 over the 256-column tile (fp32, before the scale; the tile's four waves added in wave order)
    long ld_shadow, ld_ss;     //   ld_ss >= the row tiles' extent (ceil(M / tile rows) * tile rows), % 4 == 0
    // Consumer (EPI_BF16 / EPI_GELU_BF16): out = epilogue(r[m] * acc + bias), r[m] = rsqrt(sum_{j < rf_nparts} rf_parts[j * rf_ld + m] / rf_dim + rf_eps)
    const float* rf_parts;     //   a producer's shadow_ss (rf_ld = its ld_ss); null: r = 1
    long rf_ld;
    int rf_nparts, rf_dim;     //   rf_nparts <= GEMM_RF_MAX_PARTS
    float rf_eps;
    int splitk;          // gemm_v4.hip: K split over this many blocks per tile (fp32 slabs + reduce); 0 / 1 = off
    void* dbg;           // ping-pong kernel: optional device buffer for interval timestamps (debug)      // ping-pong kernel: which wave bit selects the staggered group (tuning knob)
};

int gemm_launch(const GemmParams& p, int epilogue, bool conv, hipStream_t stream);
inline bool gemm_v4_prefer_224(const GemmParams& p) {
    const long nt = p.N / 256, cus = 256;
    const long t256 = ((long)(p.M + 255) / 256) * nt, t224 = ((long)(p.M + 223) / 224) * nt;
    const long cost256 = (t256 + cus - 1) / cus * 256, cost224 = (t224 + cus - 1) / cus * 224;
    return p.A8 ? cost224 * 100 <= cost256 * 108 : cost224 < cost256;
}
// Which kernel gemm_launch hands this problem to (host logic only, nothing is launched): the dispatch is a parity surface, so
// tests/test_host_cpu.py enumerates every GEMM the three models issue and pins its route.
enum GemmRoute {
    ROUTE_INVALID = -1,
    ROUTE_SKINNY = 0,      // gemm_skinny.hip: M <= 128 rows (the audio stream), weight-streaming
    ROUTE_V4_224 = 1,      // gemm_v4.hip layout 3, 224-row tiles (the DiT's big GEMMs)
    ROUTE_V4_256 = 2,      // gemm_v4.hip layout 3, 256-row tiles
    ROUTE_V4_W8_224 = 3,   // ... with fp8-resident weights expanded in the loop
    ROUTE_V4_W8_256 = 4,
    ROUTE_V4_F8_224 = 5,   // gemm_v4.hip layout 5: fp8 compute
    ROUTE_V4_F8_256 = 6,
    ROUTE_PP = 7,          // gemm_pp.hip: 256x256 ping-pong (big grids the asm-loop kernel does not take; convs of the upscaler / encoder)
    ROUTE_SMALL = 8,       // gemm.hip 128x128
    ROUTE_NARROW = 9       // gemm.hip 128x64 (N <= 64)
};
int gemm_route(const GemmParams& p, int epilogue, bool conv);
// p.vt set: will gemm_launch route this problem to the kernel that writes V^T from its epilogue?  (false: clear p.vt and run
// vt_transpose_launch after the GEMM; gemm_launch rejects a p.vt it cannot honour.)
bool gemm_vt_fused(const GemmParams& p, int epilogue);
// p.rowss set: will gemm_launch route this problem to a kernel that writes the row partial sums?  (false: the caller must not set it)
bool gemm_rowss_supported(const GemmParams& p, int epilogue);
// p.shadow / p.rf_parts set: will gemm_launch route this problem to the kernel that implements them (the 4-wave layout-3 kernel on dense bf16 weights)?
// gemm_launch rejects what it cannot honour.
bool gemm_fold_supported(const GemmParams& p, int epilogue);
constexpr int GEMM_RF_MAX_PARTS = 24;      // partial sums per row the consumer stages in LDS (a producer of up to 6144 columns)
// fp8 compute (p.A8 / p.ascale / p.W8 / p.wscale set): can the fp8-MFMA kernel (gemm_v4.hip layout 5) take this problem?
bool gemm_v4_f8_supported(const GemmParams& p, int epilogue);

// Skinny path (M <= 16 rows, fp32 activations, bf16 weights): out_f32 = act_out(in_act(a) @ W^T + b)
// act codes: 0 none, 1 silu, 2 gelu_tanh
int gemv_launch(const float* a, long lda, const bf16* W, const float* bias, float* out, long ldo,
                int M, int N, int K, int in_act, int out_act, hipStream_t stream);
