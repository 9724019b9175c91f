// bf16 GEMM  C[M,N] = A[M,K] . W[N,K]^T  (+ fused epilogues) and implicit-GEMM 3x3x3 conv3d
// for gfx950 / CDNA4.
//
// Replaces: MLX nn.Linear matmuls of the DiT block (reference
// LTX_2_MLX/model/transformer/attention.py:225-228,253; feed_forward.py:23,49;
// model.py:49-56,242,757) and the 3 x mx.conv2d temporal-tap loop of Conv3dSimple
// (LTX_2_MLX/model/video_vae/simple_decoder.py:146-175).
//
// Structure (v1): 128x128x64 block tile, 4 wave64 as 2x2, each wave 64x64 = 2x2 tiles of
// v_mfma_f32_32x32x16_bf16.  Both operands are K-contiguous, so A and B fragments are single
// 16-byte LDS reads.  Tiles are staged HBM->LDS with global_load_lds (16 B per lane, no VGPR
// round trip) into a double-buffered 2 x 32 KiB ring; the LDS image is lane-linear, so the bank
// swizzle (16-B chunk index ^= (row>>1)&7 on 128-B rows: conflict-free for ds_read_b128's
// 16-lane groups) is applied to the per-lane SOURCE address and again on the read.
// Block ids are remapped so that an XCD (private L2) owns a contiguous run of tiles, walked in
// groups of 8 row-tiles so concurrently resident tiles share A/W panels.
//
// Conv mode gathers the A operand on the fly from the channels-last activation volume
// [T][H][W][Cin]: row m = output position, K = tap*Cin + c; reflect padding in H/W and
// replicate padding in T are index arithmetic on the per-lane source address.
#include "gemm.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;       // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // double buffer = 64 KiB

__device__ __forceinline__ void glds16(const bf16* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void*)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

template <int EPI, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- block -> tile (XCD-contiguous, grouped row-tiles) ----
    const int Mt = (p.M + BM - 1) / BM, Nt = (p.N + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GROUP = 8;
    const int per_group = GROUP * Nt;
    const int g = id / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(Mt - first_m, GROUP);
    const int rem = id - g * per_group;
    const int m0 = (first_m + rem % gsz) * BM;
    const int n0 = (rem / gsz) * BN;

    // ---- per-lane staging addresses: 4 row-slots each for A and W ----
    const bf16* a_ptr[4];
    const bf16* w_ptr[4];
    int ct[4], chh[4], cww[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rt = (wv * 4 + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rt >> 1) & 7);
        const int m = min(m0 + rt, p.M - 1);
        const int n = min(n0 + rt, p.N - 1);
        w_ptr[j] = p.W + (long)n * p.K + chunk * 8;
        if (CONV) {
            const int hw = p.H * p.Wd;
            ct[j] = m / hw;
            const int r2 = m - ct[j] * hw;
            chh[j] = r2 / p.Wd;
            cww[j] = r2 - chh[j] * p.Wd;
            a_ptr[j] = p.A + chunk * 8;
        } else {
            a_ptr[j] = p.A + (long)m * p.lda + chunk * 8;
        }
    }

    auto stage = [&](int kt, int buf) {
        char* sa = smem + buf * STAGE_BYTES + wv * 4096;
        char* sb = sa + TILE_BYTES;
        const int k0 = kt * BK;
        if (CONV) {
            const int tap = k0 >> p.cin_shift;
            const int c0 = k0 & (p.Cin - 1);
            const int kt_ = tap / 9;
            const int kh_ = (tap - kt_ * 9) / 3;
            const int kw_ = tap - kt_ * 9 - kh_ * 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int tt = ct[j] + kt_ - p.pad_front;
                tt = max(0, min(tt, p.T - 1));
                int hh = chh[j] + kh_ - 1;
                hh = hh < 0 ? -hh : (hh >= p.H ? 2 * p.H - 2 - hh : hh);
                int ww = cww[j] + kw_ - 1;
                ww = ww < 0 ? -ww : (ww >= p.Wd ? 2 * p.Wd - 2 - ww : ww);
                const long pos = ((long)tt * p.H + hh) * p.Wd + ww;
                glds16(a_ptr[j] + pos * p.Cin + c0, sa + j * 1024);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(a_ptr[j] + k0, sa + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(w_ptr[j] + k0, sb + j * 1024);
    };

    // ---- MFMA fragment read offsets ----
    const int wr = wv >> 1, wc = wv & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int xbase = hi ^ ((l31 >> 1) & 7);          // chunk(ks) = (2*ks) ^ xbase
    const int a_row_off = (wr * 64 + l31) * 128;
    const int b_row_off = TILE_BYTES + (wc * 64 + l31) * 128;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* base = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = ((2 * ks) ^ xbase) << 4;
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *(const bf16x8*)(base + a_row_off + i * 32 * 128 + coff);
                bfr[i] = *(const bf16x8*)(base + b_row_off + i * 32 * 128 + coff);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: C layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 64 + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
        int s_idx = 0, c_idx = 0, da = 0, db = 0, dd = 0;
        if (EPI == EPI_D2S_BF16) {
            s_idx = col >> p.cf_shift;
            c_idx = col & (p.Cf - 1);
            da = s_idx / (p.fh * p.fw);
            db = (s_idx / p.fw) % p.fh;
            dd = s_idx % p.fw;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (EPI == EPI_BF16) {
                    ((bf16*)p.out)[(long)row * p.ldo + col] = f2bf(v);
                } else if (EPI == EPI_GELU_BF16) {
                    ((bf16*)p.out)[(long)row * p.ldo + col] = f2bf(gelu_tanh(v));
                } else if (EPI == EPI_SILU_BF16) {
                    ((bf16*)p.out)[(long)row * p.ldo + col] = f2bf(silu_f(v));
                } else if (EPI == EPI_F32) {
                    ((float*)p.out)[(long)row * p.ldo + col] = v;
                } else if (EPI == EPI_RESID_GATE_F32) {
                    float gt = 1.f;
                    if (p.gate || p.gate_table)
                        gt = (p.gate ? p.gate[(long)row * p.gate_stride + col] : 0.f) + (p.gate_table ? p.gate_table[col] : 0.f);
                    float* o = (float*)p.out + (long)row * p.ldo + col;
                    *o = *o + gt * v;
                } else if (EPI == EPI_ADD_BF16) {
                    v += bf2f(p.res[(long)row * p.ldres + col]);
                    ((bf16*)p.out)[(long)row * p.ldo + col] = f2bf(v);
                } else if (EPI == EPI_D2S_BF16) {
                    const int hw = p.H * p.Wd;
                    const int t = row / hw;
                    const int r2 = row - t * hw;
                    const int h = r2 / p.Wd;
                    const int w = r2 - h * p.Wd;
                    const int to = t * p.ft + da - p.drop_first;
                    if (to < 0) continue;
                    if (p.d2s_residual) {
                        const int cin_idx = (c_idx % p.c_d2s) * (p.ft * p.fh * p.fw) + s_idx;
                        v += bf2f(p.A[(long)row * p.Cin + cin_idx]);
                    }
                    const long opos = ((long)to * (p.H * p.fh) + (h * p.fh + db)) * (p.Wd * p.fw) + (w * p.fw + dd);
                    ((bf16*)p.out)[opos * p.Cf + c_idx] = f2bf(v);
                }
            }
        }
    }
}

template <int EPI, bool CONV>
int launch_t(const GemmParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<EPI, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const int Mt = (p.M + BM - 1) / BM, Nt = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_kernel<EPI, CONV>), dim3(Mt * Nt), dim3(256), LDS_BYTES, stream, p);
    LTX2_CHECK_LAUNCH("gemm_kernel");
    return LTX2_OK;
}

// ---------------------------------------------------------------------------------------------
// Skinny path: M <= 16 rows of fp32 activations against bf16 weights (timestep-embedding MLPs:
// reference timestep_embedding.py:89-124,187-202; simple_decoder.py:42-59).  HBM-bound weight
// stream: one wave per output column, 16-B weight loads, wave-shuffle reduction.
// ---------------------------------------------------------------------------------------------
template <int MAXM>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ a, long lda, const bf16* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ out, long ldo,
                                                   int M, int N, int K, int in_act, int out_act) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float accv[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) accv[m] = 0.f;
    const bf16* wrow = W + (long)n * K;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        const bf16x8 wv8 = *(const bf16x8*)(wrow + k);
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            if (m < M) {
                const float4 x0 = *(const float4*)(a + m * lda + k);
                const float4 x1 = *(const float4*)(a + m * lda + k + 4);
                float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xv = xs[e];
                    if (in_act == 1) xv = silu_f(xv);
                    else if (in_act == 2) xv = gelu_tanh(xv);
                    accv[m] += xv * bf2f(wv8[e]);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        if (m < M) {
            float v = wave_sum(accv[m]);
            if (lane == 0) {
                v += bias ? bias[n] : 0.f;
                if (out_act == 1) v = silu_f(v);
                else if (out_act == 2) v = gelu_tanh(v);
                out[m * ldo + n] = v;
            }
        }
    }
}

}  // namespace

int gemm_launch(const GemmParams& p, int epilogue, bool conv, hipStream_t stream) {
    LTX2_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    LTX2_CHECK_ARG(p.K % BK == 0, "gemm: K=%d must be a multiple of %d", p.K, BK);
    LTX2_CHECK_ARG(p.A && p.W && p.out, "gemm: null operand");
    if (conv) {
        LTX2_CHECK_ARG(p.Cin >= 64 && (p.Cin & (p.Cin - 1)) == 0, "conv3d: Cin=%d must be a power of two >= 64", p.Cin);
        LTX2_CHECK_ARG(p.K == 27 * p.Cin, "conv3d: K=%d != 27*Cin", p.K);
        LTX2_CHECK_ARG(p.H >= 2 && p.Wd >= 2 && p.T >= 1, "conv3d: reflect padding needs H,W >= 2");
        LTX2_CHECK_ARG((long)p.T * p.H * p.Wd == p.M, "conv3d: M != T*H*W");
    } else {
        LTX2_CHECK_ARG(p.lda % 8 == 0, "gemm: lda must be a multiple of 8 elements (16-byte rows)");
    }
#define CASE(E)                                                       \
    case E:                                                           \
        return conv ? launch_t<E, true>(p, stream) : launch_t<E, false>(p, stream);
    switch (epilogue) {
        CASE(EPI_BF16)
        CASE(EPI_GELU_BF16)
        CASE(EPI_SILU_BF16)
        CASE(EPI_F32)
        CASE(EPI_RESID_GATE_F32)
        CASE(EPI_ADD_BF16)
        case EPI_D2S_BF16:
            LTX2_CHECK_ARG(conv, "EPI_D2S_BF16 is conv-only");
            return launch_t<EPI_D2S_BF16, true>(p, stream);
        default:
            ltx2_set_error("gemm: unknown epilogue %d", epilogue);
            return LTX2_E_INVALID;
    }
#undef CASE
}

int gemv_launch(const float* a, long lda, const bf16* W, const float* bias, float* out, long ldo, int M, int N, int K,
                int in_act, int out_act, hipStream_t stream) {
    LTX2_CHECK_ARG(M >= 1 && M <= 16, "gemv: M=%d out of [1,16]", M);
    LTX2_CHECK_ARG(K % 8 == 0 && lda % 4 == 0, "gemv: K %% 8 and lda %% 4 required");
    const int grid = (N + 3) / 4;
    if (M <= 1)
        hipLaunchKernelGGL((gemv_kernel<1>), dim3(grid), dim3(256), 0, stream, a, lda, W, bias, out, ldo, M, N, K, in_act, out_act);
    else if (M <= 4)
        hipLaunchKernelGGL((gemv_kernel<4>), dim3(grid), dim3(256), 0, stream, a, lda, W, bias, out, ldo, M, N, K, in_act, out_act);
    else
        hipLaunchKernelGGL((gemv_kernel<16>), dim3(grid), dim3(256), 0, stream, a, lda, W, bias, out, ldo, M, N, K, in_act, out_act);
    LTX2_CHECK_LAUNCH("gemv_kernel");
    return LTX2_OK;
}
