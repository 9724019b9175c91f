// HBM-bound row / elementwise kernels for gfx950: AdaLN-RMSNorm, QK-RMSNorm + SPLIT RoPE,
// timestep sinusoid, x0/Euler update, and the VAE decoder's norm/activation/layout glue.
// All loads are 8/16-byte vectors; reductions use wave64 shuffles.
//
// Reference semantics restated (paths under /root/reference/LTX_2_MLX):
//   norm_mod          model/transformer/transformer.py:16-31 (_compiled_adaln_forward),
//                     attention.py:88-100 (rms_norm), model.py:744-758 (LayerNorm + shift/scale)
//   qknorm_rope       attention.py:231-237 (q_norm/k_norm over the FULL inner dim),
//                     rope.py:92-144 (apply_split_rotary_emb)
//   timestep_sinusoid timestep_embedding.py:10-60 with flip_sin_to_cos=True, shift 0;
//                     video_vae/simple_decoder.py:12-39
//   x0 / euler        model.py:912-918; components/diffusion_steps.py:36-67; pipelines/common.py:169-190
//   vae_*             video_vae/simple_decoder.py:339-342,228-238,492-498,528-553; ops.py:109-125
#include "rowops.h"

namespace {


// two simultaneous block sums over 256 threads
__device__ __forceinline__ void block_sum2_256(float& a, float& b, float* red) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
        red[wv] = a;
        red[4 + wv] = b;
    }
    __syncthreads();
    a = red[0] + red[1] + red[2] + red[3];
    b = red[4] + red[5] + red[6] + red[7];
}

// fp8 compute path: the block's row (already rounded to bf16: the values the bf16 path would hand the GEMM) leaves as per-token
// e4m3fn codes + scale -- ltx2_quantize_rows_fp8's arithmetic on the same bf16 values, so fused and unfused agree bit for bit.
template <int NV>
__device__ __forceinline__ void row_to_fp8(const bf16x4 (&o)[NV], int D, float* red, unsigned char* qrow, float* qscale_row) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = threadIdx.x * 4 + i * 1024;
        if (d < D)
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(bf2f(o[i][e])));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) amax = fmaxf(amax, __shfl_xor(amax, s));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
    const float inv = 1.0f / scale;
    if (threadIdx.x == 0) *qscale_row = scale;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = threadIdx.x * 4 + i * 1024;
        if (d >= D) continue;
        unsigned c = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
        c = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(bf2f(o[i][0]) * inv, bf2f(o[i][1]) * inv, (int)c, false);
        c = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(bf2f(o[i][2]) * inv, bf2f(o[i][3]) * inv, (int)c, true);
#endif
        *(unsigned*)(qrow + d) = c;
    }
}

// One block per row; the row (D <= 1024*NV fp32) is read ONCE and held in registers.
template <int NV>
__global__ __launch_bounds__(256) void norm_mod_kernel(const float* __restrict__ x, long ldx, bf16* __restrict__ out,
                                                       long ldo, int D, float eps, int layer_norm,
                                                       const float* __restrict__ scale_tab,
                                                       const float* __restrict__ shift_tab,
                                                       const float* __restrict__ scale_emb,
                                                       const float* __restrict__ shift_emb, long emb_stride,
                                                       unsigned char* __restrict__ q8, long ldq, float* __restrict__ qscale) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    const float* xr = x + row * ldx;
    f32x4 v[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = threadIdx.x * 4 + i * 1024;
        v[i] = (d < D) ? *(const f32x4*)(xr + d) : f32x4{0.f, 0.f, 0.f, 0.f};
        s1 += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        s2 += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
    block_sum2_256(s1, s2, red);
    const float mean = layer_norm ? s1 / (float)D : 0.f;
    const float ms = s2 / (float)D;
    const float var = layer_norm ? fmaxf(ms - mean * mean, 0.f) : ms;
    const float rstd = rsqrtf(var + eps);
    const float* se = scale_emb ? scale_emb + row * emb_stride : nullptr;
    const float* he = shift_emb ? shift_emb + row * emb_stride : nullptr;
    bf16* orow = out ? out + row * ldo : nullptr;
    bf16x4 ov[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = threadIdx.x * 4 + i * 1024;
        if (d >= D) continue;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (scale_tab) sc += *(const f32x4*)(scale_tab + d);
        if (se) sc += *(const f32x4*)(se + d);
        if (shift_tab) sh += *(const f32x4*)(shift_tab + d);
        if (he) sh += *(const f32x4*)(he + d);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf((v[i][e] - mean) * rstd * (1.f + sc[e]) + sh[e]);
        if (orow) *(bf16x4*)(orow + d) = o;
        ov[i] = o;
    }
    if (q8) row_to_fp8<NV>(ov, D, red, q8 + row * ldq, qscale + row);
}

// Row-invariant modulation (emb_stride == 0: the scalar-sigma case).  A block walks rows blockIdx.x, +gridDim.x, ..
// with the combined (1 + scale) and shift vectors in registers -- read per row, the four 16-KiB tables are 64 KiB of
// L2->L1 traffic beside 24 KiB of data (21.5 -> 18.8 us at 3456 x 4096).  The per-row kernel above stays for per-token
// modulation and for the plain norm, where more blocks in flight matter more (28.5 vs 32.5 us, 14.0 vs 14.5).
template <int NV>
__global__ __launch_bounds__(256) void norm_mod_shared_kernel(const float* __restrict__ x, long ldx, bf16* __restrict__ out,
                                                              long ldo, int rows, int D, float eps, int layer_norm,
                                                              const float* __restrict__ scale_tab,
                                                              const float* __restrict__ shift_tab,
                                                              const float* __restrict__ scale_emb,
                                                              const float* __restrict__ shift_emb,
                                                              unsigned char* __restrict__ q8, long ldq, float* __restrict__ qscale) {
    __shared__ float red[8];
    f32x4 sc1[NV], sh[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = threadIdx.x * 4 + i * 1024;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};     // 1 + (table + embedding): the same rounding whether the sum arrives combined (adaln_combine) or in two parts
        if (d < D) {
            if (scale_tab) sc += *(const f32x4*)(scale_tab + d);
            if (scale_emb) sc += *(const f32x4*)(scale_emb + d);
            if (shift_tab) h += *(const f32x4*)(shift_tab + d);
            if (shift_emb) h += *(const f32x4*)(shift_emb + d);
        }
        sc += 1.f;
        sc1[i] = sc;
        sh[i] = h;
    }
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* xr = x + row * ldx;
        f32x4 v[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int d = threadIdx.x * 4 + i * 1024;
            v[i] = (d < D) ? *(const f32x4*)(xr + d) : f32x4{0.f, 0.f, 0.f, 0.f};
            s1 += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            s2 += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
        }
        block_sum2_256(s1, s2, red);
        const float mean = layer_norm ? s1 / (float)D : 0.f;
        const float ms = s2 / (float)D;
        const float var = layer_norm ? fmaxf(ms - mean * mean, 0.f) : ms;
        const float rstd = rsqrtf(var + eps);
        bf16* orow = out ? out + row * ldo : nullptr;
        bf16x4 ov[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int d = threadIdx.x * 4 + i * 1024;
            if (d >= D) continue;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf((v[i][e] - mean) * rstd * sc1[i][e] + sh[i][e]);
            if (orow) *(bf16x4*)(orow + d) = o;
            ov[i] = o;
        }
        if (q8) row_to_fp8<NV>(ov, D, red, q8 + row * ldq, qscale + row);
    }
}

// Two modulations of ONE normalised row (row-invariant tables): out_g = norm(x) * (1 + scale_g) + shift_g, g = 0, 1.  The AudioVideo block's
// cross-modal attention reads the SAME pre-update RMS-normalised stream twice (transformer.py:556-620: the a2v and the v2a scale / shift rows),
// which two norm_mod launches re-read and re-reduced (round 4: one read of x, one reduction, two stores).
struct NormMod2Tabs {
    const float* scale_tab[2];
    const float* shift_tab[2];
    const float* scale_emb[2];
    const float* shift_emb[2];
};
template <int NV>
__global__ __launch_bounds__(256) void norm_mod_shared2_kernel(const float* __restrict__ x, long ldx, bf16* __restrict__ out0, bf16* __restrict__ out1,
                                                               long ldo, int rows, int D, float eps, NormMod2Tabs t) {
    __shared__ float red[8];
    f32x4 sc1[2][NV], sh[2][NV];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int d = threadIdx.x * 4 + i * 1024;
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};
            if (d < D) {
                if (t.scale_tab[g]) sc += *(const f32x4*)(t.scale_tab[g] + d);
                if (t.scale_emb[g]) sc += *(const f32x4*)(t.scale_emb[g] + d);
                if (t.shift_tab[g]) h += *(const f32x4*)(t.shift_tab[g] + d);
                if (t.shift_emb[g]) h += *(const f32x4*)(t.shift_emb[g] + d);
            }
            sc += 1.f;
            sc1[g][i] = sc;
            sh[g][i] = h;
        }
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* xr = x + row * ldx;
        f32x4 v[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int d = threadIdx.x * 4 + i * 1024;
            v[i] = (d < D) ? *(const f32x4*)(xr + d) : f32x4{0.f, 0.f, 0.f, 0.f};
            s2 += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
        }
        block_sum2_256(s1, s2, red);
        const float rstd = rsqrtf(s2 / (float)D + eps);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            bf16* orow = (g ? out1 : out0) + row * ldo;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int d = threadIdx.x * 4 + i * 1024;
                if (d >= D) continue;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(v[i][e] * rstd * sc1[g][i][e] + sh[g][i][e]);
                *(bf16x4*)(orow + d) = o;
            }
        }
    }
}

struct QKSegs {
    int off[2];
    const float* w[2];
};

// One block per row, D/2 <= 2048 rotation pairs: thread t owns pairs [8t, 8t+8) of EVERY segment
// (q and k share the cos/sin row), everything register-resident, one read + one write per element.
template <int NSEG>
__global__ __launch_bounds__(256) void qknorm_rope_kernel(bf16* __restrict__ buf, long ld, int rows, int D, int head_dim,
                                                          QKSegs segs, float eps, const float* __restrict__ cosp,
                                                          const float* __restrict__ sinp) {
    __shared__ float red[8];
    const int half = head_dim >> 1;
    const int p0 = threadIdx.x * 8;
    const bool act = p0 < D / 2;
    const int ia = act ? (p0 / half) * head_dim + (p0 % half) : 0;
    const int ib = ia + half;
    // the norm weights do not depend on the row: once per block, in registers (a block walks rows blockIdx.x, +gridDim.x, ..)
    float wa[NSEG][8], wb[NSEG][8];
#pragma unroll
    for (int g = 0; g < NSEG; ++g) {
        const float* wt = segs.w[g];
        const f32x4 a0 = *(const f32x4*)(wt + ia), a1 = *(const f32x4*)(wt + ia + 4);
        const f32x4 b0 = *(const f32x4*)(wt + ib), b1 = *(const f32x4*)(wt + ib + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            wa[g][e] = a0[e];
            wa[g][4 + e] = a1[e];
            wb[g][e] = b0[e];
            wb[g][4 + e] = b1[e];
        }
    }
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        bf16x8 va[NSEG], vb[NSEG];
        float ss[2] = {0.f, 0.f};
#pragma unroll
        for (int g = 0; g < NSEG; ++g) {
            bf16* xr = buf + row * ld + segs.off[g];
            if (act) {
                va[g] = *(const bf16x8*)(xr + ia);
                vb[g] = *(const bf16x8*)(xr + ib);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = bf2f(va[g][e]), b = bf2f(vb[g][e]);
                    ss[g] += a * a + b * b;
                }
            }
        }
        block_sum2_256(ss[0], ss[1], red);
        if (!act) continue;
        f32x4 c4[2], s4[2];
        if (cosp) {
            c4[0] = *(const f32x4*)(cosp + row * (D / 2) + p0);
            c4[1] = *(const f32x4*)(cosp + row * (D / 2) + p0 + 4);
            s4[0] = *(const f32x4*)(sinp + row * (D / 2) + p0);
            s4[1] = *(const f32x4*)(sinp + row * (D / 2) + p0 + 4);
        }
#pragma unroll
        for (int g = 0; g < NSEG; ++g) {
            bf16* xr = buf + row * ld + segs.off[g];
            const float rstd = rsqrtf(ss[g] / (float)D + eps);
            bf16x8 oa, ob;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = bf2f(va[g][e]) * rstd * wa[g][e];
                const float b = bf2f(vb[g][e]) * rstd * wb[g][e];
                if (cosp) {
                    const float c = e < 4 ? c4[0][e & 3] : c4[1][e & 3];
                    const float sn = e < 4 ? s4[0][e & 3] : s4[1][e & 3];
                    oa[e] = f2bf(a * c - b * sn);
                    ob[e] = f2bf(b * c + a * sn);
                } else {
                    oa[e] = f2bf(a);
                    ob[e] = f2bf(b);
                }
            }
            *(bf16x8*)(xr + ia) = oa;
            *(bf16x8*)(xr + ib) = ob;
        }
    }
}

__global__ void ctx_mod_kernel(const bf16* __restrict__ ctx, bf16* __restrict__ out, long n4, int D,
                               const float* __restrict__ scale_tab, const float* __restrict__ shift_tab,
                               const float* __restrict__ scale_emb, const float* __restrict__ shift_emb) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)((i * 4) % D);
        const bf16x4 v = *(const bf16x4*)(ctx + i * 4);
        const f32x4 sc = *(const f32x4*)(scale_tab + d) + *(const f32x4*)(scale_emb + d);
        const f32x4 sh = *(const f32x4*)(shift_tab + d) + *(const f32x4*)(shift_emb + d);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(bf2f(v[e]) * (1.f + sc[e]) + sh[e]);
        *(bf16x4*)(out + i * 4) = o;
    }
}

// Tall-skinny gate-logit GEMM (attention.py:241-243): logits[M][H] = X[M][K] @ Wg[H][K]^T + bg, H <= 32.
// One block per 16 rows; the 4 waves split K and reduce through LDS.  v_mfma_f32_16x16x32_bf16:
// A/B lane l holds row/col l&15, k = 8*(l>>4)..+7; D lane l holds rows 4*(l>>4)+r, col l&15.
__global__ __launch_bounds__(256) void gate_logits_kernel(const bf16* __restrict__ X, long ldx, const bf16* __restrict__ Wg,
                                                          const float* __restrict__ bg, float* __restrict__ out, long ldo,
                                                          int M, int K, int H) {
    __shared__ float part[4][16][33];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int row = min((int)blockIdx.x * 16 + r16, M - 1);
    const int kw = K / 4;                                   // this wave's K range
    const bf16* xp = X + (long)row * ldx + wv * kw + kq * 8;
    const bf16* w0 = Wg + (long)min(r16, H - 1) * K + wv * kw + kq * 8;
    const bf16* w1 = Wg + (long)min(16 + r16, H - 1) * K + wv * kw + kq * 8;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // 216 blocks x 4 waves is 3.4 waves per CU: the kernel is bound by the bytes each wave keeps in flight, so a wave requests DEPTH
    // k-steps (3 x 1 KiB each) before it consumes the first (round 4: DEPTH 4 ran the 28 MB of X at 1.4 TB/s, 20 us at 3456 x 4096)
    constexpr int DEPTH = 16;
    int k = 0;
    for (; k + 32 * DEPTH <= kw; k += 32 * DEPTH) {
        bf16x8 a[DEPTH], b0[DEPTH], b1[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            a[j] = *(const bf16x8*)(xp + k + 32 * j);
            b0[j] = *(const bf16x8*)(w0 + k + 32 * j);
            b1[j] = *(const bf16x8*)(w1 + k + 32 * j);
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            acc0 = LTX2_MFMA_16x16x32(a[j], b0[j], acc0, 0, 0, 0);
            acc1 = LTX2_MFMA_16x16x32(a[j], b1[j], acc1, 0, 0, 0);
        }
    }
#pragma unroll 4
    for (; k < kw; k += 32) {
        const bf16x8 a = *(const bf16x8*)(xp + k);
        const bf16x8 b0 = *(const bf16x8*)(w0 + k);
        const bf16x8 b1 = *(const bf16x8*)(w1 + k);
        acc0 = LTX2_MFMA_16x16x32(a, b0, acc0, 0, 0, 0);
        acc1 = LTX2_MFMA_16x16x32(a, b1, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[wv][4 * kq + r][r16] = acc0[r];
        part[wv][4 * kq + r][16 + r16] = acc1[r];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 32; i += 256) {
        const int rr = i >> 5, cc = i & 31;
        const int grow = blockIdx.x * 16 + rr;
        if (grow < M && cc < H) out[(long)grow * ldo + cc] = part[0][rr][cc] + part[1][rr][cc] + part[2][rr][cc] + part[3][rr][cc] + bg[cc];
    }
}

// The same product for many rows (round 5): the kernel above re-reads the whole 32 x K gate weight (256 KiB at K = 4096) from L2 in every 16-row block -- 55 MB of
// weight traffic beside 28 MB of activations at 3456 x 4096, 14-20 us.  Here a block takes 64 rows x ONE K slice (K / KS columns): the slice of the weight
// (32 x K/KS, 32 KiB at K = 4096 with KS = GATE_LOGIT_PARTS = 8; a 4-slice form measured no faster than the round-4 kernel) is staged in LDS once and shared by the block's four waves (16 rows each); X streams from global memory.  The KS partial
// sums are NOT reduced here: parts[ks][M][H] (fp32, no bias) -- the consumer (the attention kernel's epilogue) adds the KS values and the bias in a fixed order.
template <int KS>
__global__ __launch_bounds__(256) void gate_logits_parts_kernel(const bf16* __restrict__ X, long ldx, const bf16* __restrict__ Wg, float* __restrict__ parts,
                                                                int M, int K, int H) {
    extern __shared__ __attribute__((aligned(16))) char gl_smem[];
    const int Kc = K / KS, ks = blockIdx.y, ROWB = (Kc + 8) * 2;       // padded LDS row: 16 rows x 16 B land on 16 different bank groups
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * 64 + wv * 16;
    const bf16* xp = X + (long)min(row0 + r16, M - 1) * ldx + (long)ks * Kc + kq * 8;
    const char* w0 = gl_smem + r16 * ROWB + kq * 16;
    const char* w1 = gl_smem + (16 + r16) * ROWB + kq * 16;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int DEPTH = 16;
#ifndef LTX2_GATE_X_FIRST
#define LTX2_GATE_X_FIRST 1     // (A/B) the first DEPTH activation fragments are requested BEFORE the weight slice is staged: the two memory round trips overlap
#endif
    bf16x8 a[DEPTH];
    if (LTX2_GATE_X_FIRST) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) a[j] = (32 * j < Kc) ? *(const bf16x8*)(xp + 32 * j) : bf16x8{};
    }
    // stage the weight slice: 32 rows x Kc columns as 16-byte pieces (rows >= H repeat row H - 1: their products are never stored)
    const int pieces_per_row = Kc / 8;
    for (int i = tid; i < 32 * pieces_per_row; i += 256) {
        const int h = i / pieces_per_row, c = i - h * pieces_per_row;
        const bf16x8 v = *(const bf16x8*)(Wg + (long)min(h, H - 1) * K + (long)ks * Kc + c * 8);
        *(bf16x8*)(gl_smem + h * ROWB + c * 16) = v;
    }
    __syncthreads();
    for (int k = 0; k < Kc; k += 32 * DEPTH) {
        if (!LTX2_GATE_X_FIRST || k > 0) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) a[j] = (k + 32 * j < Kc) ? *(const bf16x8*)(xp + k + 32 * j) : bf16x8{};
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            if (k + 32 * j < Kc) {
                const bf16x8 b0 = *(const bf16x8*)(w0 + (k + 32 * j) * 2), b1 = *(const bf16x8*)(w1 + (k + 32 * j) * 2);
                acc0 = LTX2_MFMA_16x16x32(a[j], b0, acc0, 0, 0, 0);
                acc1 = LTX2_MFMA_16x16x32(a[j], b1, acc1, 0, 0, 0);
            }
        }
    }
    float* out = parts + (long)ks * M * H;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * kq + r;
        if (row < M) {
            if (r16 < H) out[(long)row * H + r16] = acc0[r];
            if (16 + r16 < H) out[(long)row * H + 16 + r16] = acc1[r];
        }
    }
}

__global__ void head_gate_kernel(bf16* __restrict__ att, long ld, const float* __restrict__ logits, long ldl, long n8,
                                 int per_row8, int hd) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long row = i / per_row8;
        const int col = (int)(i - row * per_row8) * 8;
        const float g = 2.f / (1.f + __expf(-logits[row * ldl + col / hd]));
        bf16x8* p = (bf16x8*)(att + row * ld + col);
        bf16x8 v = *p;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) * g);
        *p = v;
    }
}

// out[t'][h'][w'][co] = y_s2d[co] + mean_{q in group co} x_s2d[q]   (VAE encoder downsample, see ltx2hip.h)
__global__ __launch_bounds__(256) void s2d_downsample_kernel(const bf16* __restrict__ y, const bf16* __restrict__ x,
                                                             bf16* __restrict__ out, int T, int H, int W, int Cc, int Cin,
                                                             int st, int sh, int sw, long n) {
    const int sp = st * sh * sw, Cout = Cc * sp, gs = Cin / Cc;
    const int To = T / st, Ho = H / sh, Wo = W / sw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        long pos = i / Cout;
        const int wo = (int)(pos % Wo);
        pos /= Wo;
        const int ho = (int)(pos % Ho), to = (int)(pos / Ho);
        (void)To;
        auto src_pos = [&](int s) {
            const int a = s / (sh * sw), b = (s / sw) % sh, d = s % sw;
            return ((long)(to * st + a) * H + (ho * sh + b)) * W + (wo * sw + d);
        };
        float v = bf2f(y[src_pos(co % sp) * Cc + co / sp]);
        float m = 0.f;
        for (int q = co * gs; q < (co + 1) * gs; ++q) m += bf2f(x[src_pos(q % sp) * Cin + q / sp]);
        out[i] = f2bf(v + m / (float)gs);
    }
}

// ---- GroupNorm (spatial upscaler): statistics over (C/G channels x all positions) per group ----
// Deterministic two-level reduction (no atomics, fixed summation order => bit-reproducible statistics):
// block b reduces rows [16b, 16b+16) to per-channel sums in LDS (every channel has exactly one owner thread),
// then to per-group partials partial[b][2G]; groupnorm_finish_kernel folds the partials in a fixed tree.
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const bf16* __restrict__ x, float* __restrict__ partial, long P,
                                                              int C, int G, int rows_per_block) {
    __shared__ float ch1[2048], ch2[2048];
    const int cpg = C / G;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < P ? r0 + rows_per_block : P;
    const int vec_per_row = C / 4;
    for (int v = threadIdx.x; v < vec_per_row; v += 256) {
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        for (long r = r0; r < r1; ++r) {
            const bf16x4 q = *(const bf16x4*)(x + r * C + v * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = bf2f(q[e]);
                s1[e] += f;
                s2[e] += f * f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ch1[v * 4 + e] = s1[e];
            ch2[v * 4 + e] = s2[e];
        }
    }
    __syncthreads();
    if (threadIdx.x < G) {
        float a = 0.f, b = 0.f;
        for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) {
            a += ch1[c];
            b += ch2[c];
        }
        partial[(long)blockIdx.x * 2 * G + threadIdx.x] = a;
        partial[(long)blockIdx.x * 2 * G + G + threadIdx.x] = b;
    }
}

// sums[k] = sum_b partial[b][k], k < 2G: one block per k, strided serial sums + fixed-order LDS tree
__global__ __launch_bounds__(256) void groupnorm_finish_kernel(const float* __restrict__ partial, float* __restrict__ sums,
                                                               int nblk, int G2) {
    __shared__ float red[256];
    const int k = blockIdx.x;
    float a = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) a += partial[(long)b * G2 + k];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[k] = red[0];
}

__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const bf16* __restrict__ x, const bf16* __restrict__ res,
                                                              bf16* __restrict__ y, const float* __restrict__ sums,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              long n4, int C, int G, float inv_count, float eps, int act) {
    const int cpg = C / G;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i * 4) % C);
        float mean[4], rstd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (c + e) / cpg;
            mean[e] = sums[g] * inv_count;
            rstd[e] = rsqrtf(fmaxf(sums[G + g] * inv_count - mean[e] * mean[e], 0.f) + eps);
        }
        const bf16x4 q = *(const bf16x4*)(x + i * 4);
        const f32x4 ga = *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
        bf16x4 r4 = {f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
        if (res) r4 = *(const bf16x4*)(res + i * 4);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = (bf2f(q[e]) - mean[e]) * rstd[e] * ga[e] + be[e] + bf2f(r4[e]);
            o[e] = f2bf(act ? silu_f(v) : v);
        }
        *(bf16x4*)(y + i * 4) = o;
    }
}

// x bf16 [P][C] -> out fp32 [C][P] = (x - mean[c]) / std[c]
__global__ __launch_bounds__(256) void latent_normalize_nchw_kernel(const bf16* __restrict__ x, const float* __restrict__ mean,
                                                                    const float* __restrict__ stdv, float* __restrict__ out,
                                                                    int C, long P) {
    __shared__ float tile[64][65];
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int pr = i >> 6, cc = i & 63;
        tile[pr][cc] = (p0 + pr < P && c0 + cc < C) ? bf2f(x[(p0 + pr) * C + c0 + cc]) : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cc = i >> 6, pr = i & 63;
        if (p0 + pr < P && c0 + cc < C) out[(long)(c0 + cc) * P + p0 + pr] = (tile[pr][cc] - mean[c0 + cc]) / stdv[c0 + cc];
    }
}

// SPLIT-RoPE tables on the GPU (rope.py:214-328,365-418): cos/sin fp32 [N][half], slot pad + f*n_dims + d, identity
// padding at the FRONT; freq = grid[f] * (2*mid_d/max_pos[d] - 1), mid = (start + end) / 2.  cosf/sinf are the
// accurate (range-reducing) versions: the argument reaches ~1.6e4 rad.
__global__ void rope_tables_kernel(const float* __restrict__ pos, const float* __restrict__ grid, const float* __restrict__ max_pos,
                                   int N, int n_dims, int n_freq, int half, float* __restrict__ cosb, float* __restrict__ sinb) {
    const long total = (long)N * half;
    const int pad = half - n_dims * n_freq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / half), slot = (int)(i - (long)n * half);
        float c = 1.f, s = 0.f;
        if (slot >= pad) {
            const int j = slot - pad, f = j / n_dims, d = j - f * n_dims;
            const float mid = (pos[((long)d * N + n) * 2] + pos[((long)d * N + n) * 2 + 1]) * 0.5f;
            const float arg = grid[f] * ((mid / max_pos[d]) * 2.f - 1.f);
            c = cosf(arg);
            s = sinf(arg);
        }
        cosb[i] = c;
        sinb[i] = s;
    }
}

__global__ void timestep_sinusoid_kernel(const float* __restrict__ t, long t_stride, float t_scalar, float mult, int T,
                                         int dim, float* __restrict__ of, bf16* __restrict__ ob) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim >> 1;
    if (idx >= T * half) return;
    const int row = idx / half, i = idx - row * half;
    const float freq = __expf(-9.210340371976184f * (float)i / (float)half);   // ln(10000)
    const float arg = (t ? t[row * t_stride] : t_scalar) * mult * freq;
    const float c = cosf(arg), s = sinf(arg);
    if (of) {
        of[(long)row * dim + i] = c;
        of[(long)row * dim + half + i] = s;
    }
    if (ob) {
        ob[(long)row * dim + i] = f2bf(c);
        ob[(long)row * dim + half + i] = f2bf(s);
    }
}

// fp8 e4m3fn (OCP: bias 7, no inf, 0x7f/0xff = NaN) -> fp32, exact (loader/fp8_loader.py:14-51)
// out_bf16[i] = bf16( f32(fp8[i]) * scale )   (weight_converter.py:391-395: fp8 -> f32 * weight_scale -> target dtype)
__global__ void dequant_fp8_kernel(const unsigned char* __restrict__ in, float scale, bf16* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * blockDim.x * 8;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 7 < n) {
            const u32x2 w = *(const u32x2*)(in + i);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(e4m3fn_to_f32((w[e >> 2] >> (8 * (e & 3))) & 0xffu) * scale);
            *(bf16x8*)(out + i) = o;
        } else {
            for (long j = i; j < n; ++j) out[j] = f2bf(e4m3fn_to_f32(in[j]) * scale);
        }
    }
}

// Per-row e4m3fn quantiser of the fp8 compute path: one workgroup per row.
//   amax = max_k |x[k]| ; scale = amax / 448 (1 when the row is all zeros) ; inv = 1 / scale (IEEE) ; code[k] = e4m3fn_rne(x[k] * inv)
// The row stays in registers between the two passes (K <= 16384: 8 x 16 bytes per thread).  v_cvt_pk_fp8_f32 is the OCP conversion
// (round to nearest even, subnormals kept); |x * inv| <= 448 (1 + 2^-23) never reaches the overflow threshold (464).
template <int NIT>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16* __restrict__ x, long ldx, int K, unsigned char* __restrict__ out, long ldo,
                                                             float* __restrict__ scale_out) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const bf16* xr = x + (long)row * ldx;
    bf16x8 v[NIT];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int k = (i * 256 + tid) * 8;
        if (k < K) {
            v[i] = *(const bf16x8*)(xr + k);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(bf2f(v[i][e])));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
    const float inv = 1.0f / scale;
    if (tid == 0) scale_out[row] = scale;
    unsigned char* orow = out + (long)row * ldo;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int k = (i * 256 + tid) * 8;
        if (k < K) {
            u32x2 c = {0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
            c[0] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[i][0]) * inv, bf2f(v[i][1]) * inv, (int)c[0], false);
            c[0] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[i][2]) * inv, bf2f(v[i][3]) * inv, (int)c[0], true);
            c[1] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[i][4]) * inv, bf2f(v[i][5]) * inv, (int)c[1], false);
            c[1] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[i][6]) * inv, bf2f(v[i][7]) * inv, (int)c[1], true);
#endif
            *(u32x2*)(orow + k) = c;
        }
    }
}

// key mask fp32 [S] (non-zero = attend) -> one 64-bit word per 64 keys (bit i of word t = key 64 t + i); keys >= S read as masked
__global__ void keymask_words_kernel(const float* __restrict__ mask, int S, unsigned long long* __restrict__ words, int nwords) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= nwords) return;
    const int k = w * 64 + lane;
    const unsigned long long b = __ballot(k < S && mask[k] != 0.f);
    if (lane == 0) words[w] = b;
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *(const float4*)(in + i);
            bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
            *(bf16x4*)(out + i) = o;
        } else {
            for (long j = i; j < n; ++j) out[j] = f2bf(in[j]);
        }
    }
}

__global__ void x0_from_velocity_kernel(const float* __restrict__ latent, const float* __restrict__ vel,
                                        const float* __restrict__ ts_ptr, long ts_stride, float ts_scalar,
                                        float* __restrict__ x0, int rows, int C) {
    const long n = (long)rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long row = i / C;
        const float ts = ts_ptr ? ts_ptr[row * ts_stride] : ts_scalar;
        x0[i] = latent[i] - ts * vel[i];
    }
}

__global__ void euler_step_kernel(const float* __restrict__ x, const float* __restrict__ x0,
                                  const float* __restrict__ mask, const float* __restrict__ clean, float inv_sigma,
                                  float dt, float* __restrict__ out, int rows, int C) {
    const long n = (long)rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float d = x0[i];
        if (mask) {
            const float m = mask[i / C];
            d = d * m + clean[i] * (1.f - m);
        }
        const float xv = x[i];
        out[i] = xv + (xv - d) * inv_sigma * dt;
    }
}

// ---------------------------------- VAE glue ----------------------------------
__global__ void vae_prepare_latent_kernel(const float* __restrict__ latent, const float* __restrict__ stdv,
                                          const float* __restrict__ meanv, const float* __restrict__ noise,
                                          float ns, bf16* __restrict__ out, int C, long P) {
    // out[p][c] <- latent[c][p]; tile transpose through LDS (32 positions x 32 channels)
    __shared__ float tile[32][33];
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k;
        const long pp = p0 + tx;
        float v = 0.f;
        if (c < C && pp < P) {
            v = latent[(long)c * P + pp] * stdv[c] + meanv[c];
            if (ns > 0.f) v = (noise ? noise[(long)c * P + pp] : 0.f) * ns + (1.f - ns) * v;
        }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const long pp = p0 + k;
        const int c = c0 + tx;
        if (c < C && pp < P) out[pp * C + c] = f2bf(tile[tx][k]);
    }
}

// LP lanes per position, each lane owns C/LP contiguous channels (8 or 16).  A thread keeps its channels' (1+scale) and
// shift values in registers and walks positions with a grid stride: loaded per position, the four 16-byte table reads
// per 16 bytes of data made the kernel VMEM-issue-bound (3.85 TB/s at C = 128).
// PAD: the output is the PADDED channels-last volume [T+2][H+2][W+2][C] the implicit-GEMM conv of gemm_v4.hip reads (replicate
// in T with `pad_front` leading frames, reflect in H / W -- reference simple_decoder.py:105-134): the loop walks padded
// positions and normalises the source position each one mirrors (a few % more rows than the interior).
template <int E, bool PAD>
__global__ __launch_bounds__(256) void pixnorm_mod_silu_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long P,
                                                               int C, int lp_shift, float eps,
                                                               const float* __restrict__ tab,
                                                               const float* __restrict__ te, int shift_row,
                                                               int scale_row, int T, int H, int W, int pad_front) {
    const int LP = 1 << lp_shift;
    const long gthread = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pos_stride = ((long)gridDim.x * blockDim.x) >> lp_shift;
    const int sub = (int)(gthread & (LP - 1));
    const int c0 = sub * E;
    float sc1[E], sh[E];
#pragma unroll
    for (int q4 = 0; q4 < E / 4; ++q4) {
        const int c = c0 + q4 * 4;
        f32x4 a = *(const f32x4*)(tab + shift_row * C + c), b = *(const f32x4*)(tab + scale_row * C + c);
        if (te) {
            a += *(const f32x4*)(te + shift_row * C + c);
            b += *(const f32x4*)(te + scale_row * C + c);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sh[q4 * 4 + e] = a[e];
            sc1[q4 * 4 + e] = 1.f + b[e];
        }
    }
    const float inv_c = 1.f / (float)C;
    const int Hp = H + 2, Wp = W + 2;
    // the LP lanes of a position share `pos`, so a shuffle group is always wholly inside or wholly outside the loop
    for (long pos = gthread >> lp_shift; pos < P; pos += pos_stride) {
        long src = pos;
        if (PAD) {
            const int tp = (int)(pos / ((long)Hp * Wp)), r2 = (int)(pos - (long)tp * Hp * Wp), hp = r2 / Wp, wp = r2 - hp * Wp;
            const int t = min(max(tp - pad_front, 0), T - 1);
            int h = hp - 1, w = wp - 1;
            h = h < 0 ? -h : (h >= H ? 2 * H - 2 - h : h);
            w = w < 0 ? -w : (w >= W ? 2 * W - 2 - w : w);
            src = ((long)t * H + h) * W + w;
        }
        float v[E];
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < E / 8; ++i) {
            const bf16x8 t8 = *(const bf16x8*)(x + src * C + c0 + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i * 8 + e] = bf2f(t8[e]);
                s2 += v[i * 8 + e] * v[i * 8 + e];
            }
        }
        for (int o = LP >> 1; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
        const float rstd = rsqrtf(s2 * inv_c + eps);
#pragma unroll
        for (int i = 0; i < E / 8; ++i) {
            bf16x8 o8;
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = f2bf(silu_f(v[i * 8 + e] * rstd * sc1[i * 8 + e] + sh[i * 8 + e]));
            *(bf16x8*)(y + pos * C + c0 + i * 8) = o8;
        }
    }
}

__global__ void vae_unpatchify_kernel(const bf16* __restrict__ x, float* __restrict__ video, int T, int H, int W) {
    // video[c][t][h*4+rh][w*4+rw] = x[t][h][w][c*16 + rw*4 + rh]
    const int HO = H * 4, WO = W * 4;
    const long n = (long)3 * T * HO * WO;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % WO);
        long r = i / WO;
        const int yo = (int)(r % HO);
        r /= HO;
        const int t = (int)(r % T);
        const int c = (int)(r / T);
        const int w = xo >> 2, rw = xo & 3, h = yo >> 2, rh = yo & 3;
        video[i] = bf2f(x[(((long)t * H + h) * W + w) * 48 + c * 16 + rw * 4 + rh]);
    }
}

__global__ void video_to_uint8_kernel(const float* __restrict__ video, unsigned char* __restrict__ frames, int T, int H,
                                      int W) {
    const long plane = (long)T * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (video[c * plane + i] + 1.f) * 0.5f;
            v = fminf(fmaxf(v, 0.f), 1.f) * 255.f;
            frames[i * 3 + c] = (unsigned char)v;
        }
    }
}

// One temporal chunk of decode_latent -> uint8 frames, cross-faded with the previous chunk on the first `ov` frames
// (simple_decoder.py:760-798): frame j of `cur` lands on output frame t_dst0 + j; for j < ov the value is
// prev[prev_T - ov + j] * (1 - ramp[j]) + cur[j] * ramp[j] with torch.linspace's ramp, rounded exactly like the separate torch
// ops it replaces (no FMA contraction); frames at or beyond T_out are dropped (the final trim).
__global__ void video_chunk_to_uint8_kernel(const float* __restrict__ cur, const float* __restrict__ prev, const float* __restrict__ ramp,
                                            unsigned char* __restrict__ frames, int Tc, int prev_T, int ov, int H, int W,
                                            int t_dst0, int T_out) {
    const long hw = (long)H * W, plane = (long)Tc * hw, pplane = (long)prev_T * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i / hw);
        const long px = i - (long)j * hw;
        const int t = t_dst0 + j;
        if (t >= T_out) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = cur[c * plane + i];
            if (prev && j < ov) {
                const float r = ramp[j];
                const float a = prev[c * pplane + (long)(prev_T - ov + j) * hw + px];
                v = __fadd_rn(__fmul_rn(a, __fsub_rn(1.0f, r)), __fmul_rn(v, r));
            }
            v = (v + 1.f) * 0.5f;
            v = fminf(fmaxf(v, 0.f), 1.f) * 255.f;
            frames[((long)t * hw + px) * 3 + c] = (unsigned char)v;
        }
    }
}

// decode_tiled (tiling.py:380-412): out[c][t0+t][h0+h][w0+w] += tile[c][t][h][w] * m, wsum[...] += m with the separable
// trapezoid mask m = mt[t] * mh[h] * mw[w], one pass over the tile instead of three torch passes over volume slices.
__global__ void tile_blend_accumulate_kernel(const float* __restrict__ tile, int dt, int dh, int dw, int nt, int nh, int nw,
                                             const float* __restrict__ mt, const float* __restrict__ mh, const float* __restrict__ mw,
                                             float* __restrict__ out, float* __restrict__ wsum, int OT, int OH, int OW, int t0, int h0, int w0) {
    const long n = (long)nt * nh * nw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % nw);
        const long r = i / nw;
        const int h = (int)(r % nh), t = (int)(r / nh);
        const float m = __fmul_rn(__fmul_rn(mt[t], mh[h]), mw[w]);
        const long o = ((long)(t0 + t) * OH + (h0 + h)) * OW + (w0 + w);
        wsum[o] = __fadd_rn(wsum[o], m);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const long oc = (long)c * OT * OH * OW + o;
            out[oc] = __fadd_rn(out[oc], __fmul_rn(tile[((long)c * dt + t) * dh * dw + (long)h * dw + w], m));
        }
    }
}

__global__ void tile_blend_finish_kernel(float* __restrict__ out, const float* __restrict__ wsum, long plane) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        const float d = fmaxf(wsum[i], 1e-8f);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c * plane + i] = __fdiv_rn(out[c * plane + i], d);
    }
}

inline int grid_for(long n, int block, int cap = 4096) {
    long g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

int norm_mod_launch(const float* x, long ldx, bf16* out, long ldo, int rows, int D, float eps, int layer_norm,
                    const float* scale_tab, const float* shift_tab, const float* scale_emb, const float* shift_emb,
                    long emb_stride, hipStream_t stream, unsigned char* q8, long ldq, float* qscale) {
    LTX2_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0, "norm_mod: D, ldx, ldo must be multiples of 4");
    LTX2_CHECK_ARG((out || q8) && (!q8 || (qscale && ldq % 4 == 0)), "norm_mod: no output / fp8 output without a scale vector");
    LTX2_CHECK_ARG(D <= 8192 && emb_stride % 4 == 0, "norm_mod: D=%d exceeds 8192 or emb_stride not a multiple of 4", D);
    const bool shared_mod = emb_stride == 0 && (scale_tab || shift_tab || scale_emb || shift_emb) && rows > 1024;
#ifndef LTX2_NORM_BLOCKS
#define LTX2_NORM_BLOCKS 1024       // blocks of the row-invariant form (each loads the combined tables once); A/B builds: -DLTX2_NORM_BLOCKS=512 / 2048
#endif
    if (shared_mod) {
        const int per_block = (rows + LTX2_NORM_BLOCKS - 1) / LTX2_NORM_BLOCKS;
        const int grid = (rows + per_block - 1) / per_block;
        if (D <= 4096)
            hipLaunchKernelGGL((norm_mod_shared_kernel<4>), dim3(grid), dim3(256), 0, stream, x, ldx, out, ldo, rows, D, eps, layer_norm,
                               scale_tab, shift_tab, scale_emb, shift_emb, q8, ldq, qscale);
        else
            hipLaunchKernelGGL((norm_mod_shared_kernel<8>), dim3(grid), dim3(256), 0, stream, x, ldx, out, ldo, rows, D, eps, layer_norm,
                               scale_tab, shift_tab, scale_emb, shift_emb, q8, ldq, qscale);
    } else if (D <= 4096) {
        hipLaunchKernelGGL((norm_mod_kernel<4>), dim3(rows), dim3(256), 0, stream, x, ldx, out, ldo, D, eps, layer_norm,
                           scale_tab, shift_tab, scale_emb, shift_emb, emb_stride, q8, ldq, qscale);
    } else {
        hipLaunchKernelGGL((norm_mod_kernel<8>), dim3(rows), dim3(256), 0, stream, x, ldx, out, ldo, D, eps, layer_norm,
                           scale_tab, shift_tab, scale_emb, shift_emb, emb_stride, q8, ldq, qscale);
    }
    LTX2_CHECK_LAUNCH("norm_mod_kernel");
    return LTX2_OK;
}

// comb[l][i] = tab[l][i] + emb[i]: the AdaLN rows of EVERY layer for this step in one launch (row-invariant modulation: one sigma per step).  The norm
// kernels then read two vectors per row set instead of four and the gated-residual GEMMs take their gate as a table (no per-row gate loads).
static __global__ __launch_bounds__(256) void adaln_combine_kernel(const float* __restrict__ tab, const float* __restrict__ emb, float* __restrict__ out, long n, long total) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 1024) {
        const f32x4 a = *(const f32x4*)(tab + i), e = *(const f32x4*)(emb + i % n);
        *(f32x4*)(out + i) = a + e;
    }
}
int adaln_combine_launch(const float* tab, const float* emb, float* out, int layers, long n, hipStream_t stream) {
    LTX2_CHECK_ARG(tab && emb && out && layers > 0 && n > 0 && n % 4 == 0, "adaln_combine: bad operand");
    const long total = (long)layers * n, want = (total / 4 + 255) / 256;
    hipLaunchKernelGGL(adaln_combine_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, stream, tab, emb, out, n, total);
    LTX2_CHECK_LAUNCH("adaln_combine_kernel");
    return LTX2_OK;
}

int norm_mod2_launch(const float* x, long ldx, bf16* out0, bf16* out1, long ldo, int rows, int D, float eps, const float* const* scale_tab,
                     const float* const* shift_tab, const float* const* scale_emb, const float* const* shift_emb, hipStream_t stream) {
    LTX2_CHECK_ARG(x && out0 && out1 && rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && D <= 8192, "norm_mod2: bad operand (D <= 8192, multiples of 4)");
    NormMod2Tabs t{};
    for (int g = 0; g < 2; ++g) {
        t.scale_tab[g] = scale_tab[g];
        t.shift_tab[g] = shift_tab[g];
        t.scale_emb[g] = scale_emb[g];
        t.shift_emb[g] = shift_emb[g];
    }
    const int per_block = (rows + LTX2_NORM_BLOCKS - 1) / LTX2_NORM_BLOCKS;
    const int grid = (rows + per_block - 1) / per_block;
    if (D <= 4096)
        hipLaunchKernelGGL((norm_mod_shared2_kernel<4>), dim3(grid), dim3(256), 0, stream, x, ldx, out0, out1, ldo, rows, D, eps, t);
    else        // the two norm_mod launches this replaced took D <= 8192
        hipLaunchKernelGGL((norm_mod_shared2_kernel<8>), dim3(grid), dim3(256), 0, stream, x, ldx, out0, out1, ldo, rows, D, eps, t);
    LTX2_CHECK_LAUNCH("norm_mod_shared2_kernel");
    return LTX2_OK;
}

int qknorm_rope_launch(bf16* buf, long ld, int rows, int D, int head_dim, int nseg, const int* seg_off,
                       const float* const* weights, float eps, const float* cos, const float* sin, hipStream_t stream) {
    LTX2_CHECK_ARG(nseg >= 1 && nseg <= 2, "qknorm_rope: nseg must be 1 or 2");
    LTX2_CHECK_ARG(head_dim % 16 == 0 && D % head_dim == 0 && ld % 8 == 0, "qknorm_rope: head_dim %% 16, D %% head_dim, ld %% 8");
    LTX2_CHECK_ARG((cos == nullptr) == (sin == nullptr), "qknorm_rope: cos and sin must both be set or both null");
    QKSegs s{};
    for (int i = 0; i < nseg; ++i) {
        LTX2_CHECK_ARG(seg_off[i] % 8 == 0 && weights[i], "qknorm_rope: bad segment");
        s.off[i] = seg_off[i];
        s.w[i] = weights[i];
    }
    LTX2_CHECK_ARG(D <= 4096, "qknorm_rope: inner dim %d exceeds 4096", D);
#ifndef LTX2_QK_BLOCKS
#define LTX2_QK_BLOCKS 2048         // up to this many blocks; beyond that a block takes several rows (each block loads the norm weights once; 512 .. 4096 measured the same)
#endif
    const int per_block = (rows + LTX2_QK_BLOCKS - 1) / LTX2_QK_BLOCKS;
    const int grid = (rows + per_block - 1) / per_block;
    if (nseg == 2)
        hipLaunchKernelGGL((qknorm_rope_kernel<2>), dim3(grid), dim3(256), 0, stream, buf, ld, rows, D, head_dim, s, eps, cos, sin);
    else
        hipLaunchKernelGGL((qknorm_rope_kernel<1>), dim3(grid), dim3(256), 0, stream, buf, ld, rows, D, head_dim, s, eps, cos, sin);
    LTX2_CHECK_LAUNCH("qknorm_rope_kernel");
    return LTX2_OK;
}

int ctx_mod_launch(const bf16* ctx, bf16* out, int rows, int D, const float* scale_tab, const float* shift_tab,
                   const float* scale_emb, const float* shift_emb, hipStream_t stream) {
    LTX2_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0, "ctx_mod: D must be a multiple of 4");
    LTX2_CHECK_ARG(scale_tab && shift_tab && scale_emb && shift_emb, "ctx_mod: null table");
    const long n4 = (long)rows * D / 4;
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(ctx_mod_kernel, dim3(grid), dim3(256), 0, stream, ctx, out, n4, D, scale_tab, shift_tab, scale_emb, shift_emb);
    LTX2_CHECK_LAUNCH("ctx_mod_kernel");
    return LTX2_OK;
}

int gate_logits_launch(const bf16* X, long ldx, const bf16* Wg, const float* bg, float* out, long ldo, int M, int K, int H,
                       hipStream_t stream) {
    LTX2_CHECK_ARG(M > 0 && H > 0 && H <= 32 && K % 128 == 0 && ldx % 8 == 0, "gate_logits: need H <= 32, K %% 128 == 0, ldx %% 8 == 0");
    hipLaunchKernelGGL(gate_logits_kernel, dim3((M + 15) / 16), dim3(256), 0, stream, X, ldx, Wg, bg, out, ldo, M, K, H);
    LTX2_CHECK_LAUNCH("gate_logits_kernel");
    return LTX2_OK;
}

int gate_logits_parts_launch(const bf16* X, long ldx, const bf16* Wg, float* parts, int M, int K, int H, hipStream_t stream) {
    LTX2_CHECK_ARG(M > 0 && H > 0 && H <= 32 && K % (GATE_LOGIT_PARTS * 32) == 0 && ldx % 8 == 0, "gate_logits_parts: need H <= 32, K %% %d == 0, ldx %% 8 == 0", GATE_LOGIT_PARTS * 32);
    const int lds = 32 * (K / GATE_LOGIT_PARTS + 8) * 2;
    LTX2_CHECK_ARG(lds <= 160 * 1024, "gate_logits_parts: K=%d too wide for the LDS-resident weight slice", K);
    static PerDeviceOnce once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)gate_logits_parts_kernel<GATE_LOGIT_PARTS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(gate_logits_parts_kernel<GATE_LOGIT_PARTS>, dim3((M + 63) / 64, GATE_LOGIT_PARTS), dim3(256), lds, stream, X, ldx, Wg, parts, M, K, H);
    LTX2_CHECK_LAUNCH("gate_logits_parts_kernel");
    return LTX2_OK;
}

int head_gate_launch(bf16* att, long ld, const float* logits, long ldl, int rows, int H, int hd, hipStream_t stream) {
    LTX2_CHECK_ARG(rows > 0 && H > 0 && hd % 8 == 0 && ld % 8 == 0, "head_gate: head_dim and ld must be multiples of 8");
    const int per_row8 = H * hd / 8;
    const long n8 = (long)rows * per_row8;
    const int grid = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
    hipLaunchKernelGGL(head_gate_kernel, dim3(grid), dim3(256), 0, stream, att, ld, logits, ldl, n8, per_row8, hd);
    LTX2_CHECK_LAUNCH("head_gate_kernel");
    return LTX2_OK;
}

int s2d_downsample_launch(const bf16* y, const bf16* x, bf16* out, int T, int H, int W, int Cc, int Cin, int st, int sh, int sw,
                          hipStream_t stream) {
    LTX2_CHECK_ARG(y && x && out && T > 0 && H > 0 && W > 0, "s2d_downsample: bad argument");
    LTX2_CHECK_ARG(st >= 1 && sh >= 1 && sw >= 1 && T % st == 0 && H % sh == 0 && W % sw == 0, "s2d_downsample: dims must divide by the stride");
    LTX2_CHECK_ARG(Cc > 0 && Cin % Cc == 0, "s2d_downsample: Cin=%d must be a multiple of the conv width %d", Cin, Cc);
    const long n = (long)(T / st) * (H / sh) * (W / sw) * Cc * st * sh * sw;
    hipLaunchKernelGGL(s2d_downsample_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, stream, y, x, out, T, H, W, Cc, Cin, st, sh, sw, n);
    LTX2_CHECK_LAUNCH("s2d_downsample_kernel");
    return LTX2_OK;
}

int groupnorm_silu_launch(const bf16* x, const bf16* res, bf16* y, long P, int C, int G, float eps, const float* gamma,
                          const float* beta, float* scratch, int act, hipStream_t stream) {
    LTX2_CHECK_ARG(x && y && gamma && beta && scratch && P > 0, "groupnorm: null operand");
    LTX2_CHECK_ARG(G >= 1 && G <= 64 && C % G == 0 && C % 4 == 0 && C <= 2048, "groupnorm: need groups <= 64, C %% groups == 0, C %% 4 == 0, C <= 2048 (C=%d, groups=%d)", C, G);
    const int rows_per_block = 16;
    const int nblk = (int)((P + rows_per_block - 1) / rows_per_block);
    float* sums = scratch;                 // [2G]
    float* partial = scratch + 2 * G;      // [nblk][2G]
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(nblk), dim3(256), 0, stream, x, partial, P, C, G, rows_per_block);
    LTX2_CHECK_LAUNCH("groupnorm_stats_kernel");
    hipLaunchKernelGGL(groupnorm_finish_kernel, dim3(2 * G), dim3(256), 0, stream, partial, sums, nblk, 2 * G);
    LTX2_CHECK_LAUNCH("groupnorm_finish_kernel");
    const long n4 = P * C / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(grid), dim3(256), 0, stream, x, res, y, sums, gamma, beta, n4, C, G,
                       1.0f / ((float)P * (float)(C / G)), eps, act);
    LTX2_CHECK_LAUNCH("groupnorm_apply_kernel");
    return LTX2_OK;
}

int latent_normalize_nchw_launch(const bf16* x, const float* mean, const float* stdv, float* out, int C, long P, hipStream_t stream) {
    LTX2_CHECK_ARG(x && mean && stdv && out && C > 0 && P > 0, "latent_normalize: bad argument");
    hipLaunchKernelGGL(latent_normalize_nchw_kernel, dim3((int)((P + 63) / 64), (C + 63) / 64), dim3(256), 0, stream, x, mean, stdv, out, C, P);
    LTX2_CHECK_LAUNCH("latent_normalize_nchw_kernel");
    return LTX2_OK;
}

int rope_tables_launch(const float* pos, const float* grid, const float* max_pos, int N, int n_dims, int n_freq, int half,
                       float* cosb, float* sinb, hipStream_t stream) {
    LTX2_CHECK_ARG(pos && grid && max_pos && cosb && sinb && N > 0, "rope_tables: null operand");
    LTX2_CHECK_ARG(n_dims >= 1 && n_freq >= 1 && n_dims * n_freq <= half, "rope_tables: n_dims*n_freq=%d exceeds D/2=%d", n_dims * n_freq, half);
    hipLaunchKernelGGL(rope_tables_kernel, dim3(grid_for((long)N * half, 256, 8192)), dim3(256), 0, stream, pos, grid, max_pos, N, n_dims,
                       n_freq, half, cosb, sinb);
    LTX2_CHECK_LAUNCH("rope_tables_kernel");
    return LTX2_OK;
}

int timestep_sinusoid_launch(const float* t, long t_stride, float t_scalar, float mult, int T, int dim, float* out_f32,
                             bf16* out_bf16, hipStream_t stream) {
    LTX2_CHECK_ARG(T > 0 && dim % 2 == 0, "timestep_sinusoid: bad shape");
    const int n = T * (dim / 2);
    hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, t, t_stride, t_scalar, mult, T,
                       dim, out_f32, out_bf16);
    LTX2_CHECK_LAUNCH("timestep_sinusoid_kernel");
    return LTX2_OK;
}

int dequant_fp8_launch(const unsigned char* in, float scale, bf16* out, long n, hipStream_t stream) {
    LTX2_CHECK_ARG(in && out && n > 0, "dequant_fp8: bad argument");
    LTX2_CHECK_ARG((((uintptr_t)in) & 7) == 0 && (((uintptr_t)out) & 15) == 0, "dequant_fp8: pointers must be 8/16-byte aligned");
    const long blocks = (n / 8 + 255) / 256;
    hipLaunchKernelGGL(dequant_fp8_kernel, dim3((int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks))), dim3(256), 0, stream, in, scale, out, n);
    LTX2_CHECK_LAUNCH("dequant_fp8_kernel");
    return LTX2_OK;
}

int quant_rows_fp8_launch(const bf16* x, long ldx, int rows, int K, unsigned char* out, long ldo, float* scale, hipStream_t stream) {
    LTX2_CHECK_ARG(rows > 0 && K > 0 && K % 8 == 0 && K <= 16 * 2048 && ldx % 8 == 0 && ldo % 8 == 0, "quant_rows_fp8: rows=%d K=%d (K %% 8 == 0, K <= 32768, 16-byte rows)", rows, K);
    const int nit = (K + 2047) / 2048;
    if (nit <= 2)
        hipLaunchKernelGGL(quant_rows_fp8_kernel<2>, dim3(rows), dim3(256), 0, stream, x, ldx, K, out, ldo, scale);
    else if (nit <= 8)
        hipLaunchKernelGGL(quant_rows_fp8_kernel<8>, dim3(rows), dim3(256), 0, stream, x, ldx, K, out, ldo, scale);
    else
        hipLaunchKernelGGL(quant_rows_fp8_kernel<16>, dim3(rows), dim3(256), 0, stream, x, ldx, K, out, ldo, scale);
    LTX2_CHECK_LAUNCH("quant_rows_fp8_kernel");
    return LTX2_OK;
}

int keymask_words_launch(const float* mask, int S, unsigned long long* words, int nwords, hipStream_t stream) {
    LTX2_CHECK_ARG(mask && words && S > 0 && nwords * 64 >= S, "keymask_words: bad argument");
    hipLaunchKernelGGL(keymask_words_kernel, dim3((nwords + 3) / 4), dim3(256), 0, stream, mask, S, words, nwords);
    LTX2_CHECK_LAUNCH("keymask_words_kernel");
    return LTX2_OK;
}

int cast_f32_bf16_launch(const float* in, bf16* out, long n, hipStream_t stream) {
    LTX2_CHECK_ARG(n > 0, "cast: empty");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, stream, in, out, n);
    LTX2_CHECK_LAUNCH("cast_f32_bf16_kernel");
    return LTX2_OK;
}

int x0_from_velocity_launch(const float* latent, const float* vel, const float* ts_ptr, long ts_stride, float ts_scalar,
                            float* x0, int rows, int C, hipStream_t stream) {
    hipLaunchKernelGGL(x0_from_velocity_kernel, dim3(grid_for((long)rows * C, 256)), dim3(256), 0, stream, latent, vel,
                       ts_ptr, ts_stride, ts_scalar, x0, rows, C);
    LTX2_CHECK_LAUNCH("x0_from_velocity_kernel");
    return LTX2_OK;
}

int euler_step_launch(const float* x, const float* x0, const float* mask, const float* clean, float sigma,
                      float sigma_next, float* out, int rows, int C, hipStream_t stream) {
    LTX2_CHECK_ARG(sigma != 0.f, "Sigma can't be 0.0");   // reference core_utils.py:54-55
    LTX2_CHECK_ARG((mask == nullptr) == (clean == nullptr), "euler_step: mask and clean go together");
    hipLaunchKernelGGL(euler_step_kernel, dim3(grid_for((long)rows * C, 256)), dim3(256), 0, stream, x, x0, mask, clean,
                       1.0f / sigma, sigma_next - sigma, out, rows, C);
    LTX2_CHECK_LAUNCH("euler_step_kernel");
    return LTX2_OK;
}

int vae_prepare_latent_launch(const float* latent, const float* std, const float* mean, const float* noise,
                              float noise_scale, bf16* out, int C, long P, hipStream_t stream) {
    dim3 grid((unsigned)((P + 31) / 32), (C + 31) / 32);
    hipLaunchKernelGGL(vae_prepare_latent_kernel, grid, dim3(256), 0, stream, latent, std, mean, noise, noise_scale, out, C, P);
    LTX2_CHECK_LAUNCH("vae_prepare_latent_kernel");
    return LTX2_OK;
}

// y = x copied into the padded channels-last volume [T+2][H+2][W+2][C] of the implicit-GEMM conv (same padding rule as the padded
// pixel norm above, no arithmetic): the input of the depth-to-space upsample convs, which no norm precedes
__global__ __launch_bounds__(256) void pad_volume_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long P, int cshift, int T, int H, int W,
                                                         int pad_front) {
    const int Hp = H + 2, Wp = W + 2, C8 = 1 << cshift;            // 16-byte chunks per position
    const long total = P << cshift;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pos = i >> cshift;
        const int c = (int)(i & (C8 - 1));
        const int tp = (int)(pos / ((long)Hp * Wp)), r2 = (int)(pos - (long)tp * Hp * Wp), hp = r2 / Wp, wp = r2 - hp * Wp;
        const int t = min(max(tp - pad_front, 0), T - 1);
        int h = hp - 1, w = wp - 1;
        h = h < 0 ? -h : (h >= H ? 2 * H - 2 - h : h);
        w = w < 0 ? -w : (w >= W ? 2 * W - 2 - w : w);
        const long src = ((long)t * H + h) * W + w;
        *(u32x4*)(y + (pos << (cshift + 3)) + c * 8) = *(const u32x4*)(x + (src << (cshift + 3)) + c * 8);
    }
}

static int pixnorm_launch_impl(const bf16* x, bf16* y, long P, int C, float eps, const float* tab, const float* te,
                               int shift_row, int scale_row, bool pad, int T, int H, int W, int pad_front, hipStream_t stream) {
    LTX2_CHECK_ARG(C >= 64 && (C & (C - 1)) == 0 && C <= 1024, "pixnorm: C=%d must be a power of two in [64,1024]", C);
    const int E = (C >= 1024) ? 16 : 8;
    const int LP = C / E;
    int lp_shift = 0;
    while ((1 << lp_shift) < LP) ++lp_shift;
    const long threads = P * LP;
    const long want = (threads + 255) / 256;
    const unsigned grid = (unsigned)(want < 8192 ? want : 8192);        // 32 blocks per CU; the kernel walks the rest
#define PIX_LAUNCH(EE, PP) \
    hipLaunchKernelGGL((pixnorm_mod_silu_kernel<EE, PP>), dim3(grid), dim3(256), 0, stream, x, y, P, C, lp_shift, eps, tab, te, shift_row, scale_row, T, H, W, pad_front)
    if (E == 16) {
        if (pad) PIX_LAUNCH(16, true);
        else PIX_LAUNCH(16, false);
    } else {
        if (pad) PIX_LAUNCH(8, true);
        else PIX_LAUNCH(8, false);
    }
#undef PIX_LAUNCH
    LTX2_CHECK_LAUNCH("pixnorm_mod_silu_kernel");
    return LTX2_OK;
}

int pixnorm_mod_silu_launch(const bf16* x, bf16* y, long P, int C, float eps, const float* tab, const float* te,
                            int shift_row, int scale_row, hipStream_t stream) {
    return pixnorm_launch_impl(x, y, P, C, eps, tab, te, shift_row, scale_row, false, 0, 0, 0, 0, stream);
}

// y = the padded volume [T+2][H+2][W+2][C] (see the kernel comment)
int pixnorm_mod_silu_padded_launch(const bf16* x, bf16* y, int T, int H, int W, int C, float eps, const float* tab, const float* te,
                                   int shift_row, int scale_row, int pad_front, hipStream_t stream) {
    LTX2_CHECK_ARG(H >= 2 && W >= 2 && T >= 1 && (pad_front == 1 || pad_front == 2), "pixnorm (padded): reflect padding needs H, W >= 2");
    return pixnorm_launch_impl(x, y, (long)(T + 2) * (H + 2) * (W + 2), C, eps, tab, te, shift_row, scale_row, true, T, H, W, pad_front, stream);
}

int pad_volume_launch(const bf16* x, bf16* y, int T, int H, int W, int C, int pad_front, hipStream_t stream) {
    LTX2_CHECK_ARG(H >= 2 && W >= 2 && T >= 1 && (pad_front == 1 || pad_front == 2) && C >= 8 && (C & (C - 1)) == 0, "pad_volume: reflect padding needs H, W >= 2; C a power of two >= 8");
    int cshift = 0;
    while ((8 << cshift) < C) ++cshift;
    const long P = (long)(T + 2) * (H + 2) * (W + 2);
    hipLaunchKernelGGL(pad_volume_kernel, dim3(grid_for(P << cshift, 256, 16384)), dim3(256), 0, stream, x, y, P, cshift, T, H, W, pad_front);
    LTX2_CHECK_LAUNCH("pad_volume_kernel");
    return LTX2_OK;
}

int vae_unpatchify_launch(const bf16* x, float* video, int T, int H, int W, hipStream_t stream) {
    const long n = (long)3 * T * H * 4 * W * 4;
    hipLaunchKernelGGL(vae_unpatchify_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, stream, x, video, T, H, W);
    LTX2_CHECK_LAUNCH("vae_unpatchify_kernel");
    return LTX2_OK;
}

int video_chunk_to_uint8_launch(const float* cur, const float* prev, const float* ramp, unsigned char* frames, int Tc, int prev_T, int ov,
                                int H, int W, int t_dst0, int T_out, hipStream_t stream) {
    LTX2_CHECK_ARG(Tc > 0 && H > 0 && W > 0 && (!prev || (ov >= 1 && ov <= Tc && ov <= prev_T && ramp)), "video_chunk_to_uint8: bad overlap");
    const long n = (long)Tc * H * W;
    hipLaunchKernelGGL(video_chunk_to_uint8_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, stream, cur, prev, ramp, frames, Tc, prev_T, ov,
                       H, W, t_dst0, T_out);
    LTX2_CHECK_LAUNCH("video_chunk_to_uint8_kernel");
    return LTX2_OK;
}

int tile_blend_accumulate_launch(const float* tile, int dt, int dh, int dw, int nt, int nh, int nw, const float* mt, const float* mh,
                                 const float* mw, float* out, float* wsum, int OT, int OH, int OW, int t0, int h0, int w0, hipStream_t stream) {
    LTX2_CHECK_ARG(nt <= dt && nh <= dh && nw <= dw && t0 + nt <= OT && h0 + nh <= OH && w0 + nw <= OW, "tile_blend_accumulate: tile outside the volume");
    const long n = (long)nt * nh * nw;
    hipLaunchKernelGGL(tile_blend_accumulate_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, stream, tile, dt, dh, dw, nt, nh, nw, mt, mh, mw,
                       out, wsum, OT, OH, OW, t0, h0, w0);
    LTX2_CHECK_LAUNCH("tile_blend_accumulate_kernel");
    return LTX2_OK;
}

int tile_blend_finish_launch(float* out, const float* wsum, long plane, hipStream_t stream) {
    hipLaunchKernelGGL(tile_blend_finish_kernel, dim3(grid_for(plane, 256, 16384)), dim3(256), 0, stream, out, wsum, plane);
    LTX2_CHECK_LAUNCH("tile_blend_finish_kernel");
    return LTX2_OK;
}

int video_to_uint8_launch(const float* video, unsigned char* frames, int T, int H, int W, hipStream_t stream) {
    const long n = (long)T * H * W;
    hipLaunchKernelGGL(video_to_uint8_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, stream, video, frames, T, H, W);
    LTX2_CHECK_LAUNCH("video_to_uint8_kernel");
    return LTX2_OK;
}
