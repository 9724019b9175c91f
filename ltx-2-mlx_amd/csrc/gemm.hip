// bf16 GEMM  C[M,N] = A[M,K] . W[N,K]^T  (+ fused epilogues) and implicit-GEMM 3x3x3 conv3d
// for gfx950 / CDNA4.
//
// Replaces: MLX nn.Linear matmuls of the DiT block (reference
// LTX_2_MLX/model/transformer/attention.py:225-228,253; feed_forward.py:23,49;
// model.py:49-56,242,757) and the 3 x mx.conv2d temporal-tap loop of Conv3dSimple
// (LTX_2_MLX/model/video_vae/simple_decoder.py:146-175).
//
// Structure: BM x BN x 64 block tile, WAVES_M x WAVES_N wave64 grid, each wave a grid of
// v_mfma_f32_32x32x16_bf16 tiles.  Two configurations are instantiated:
//   * 256x256, 8 waves (2x4, wave tile 128x64), 1 block/CU: large problems (128 FLOP per staged
//     byte, keeps the L2->LDS stream well under the per-XCD L2 bandwidth);
//   * 128x128, 4 waves (2x2, wave tile 64x64), 2 blocks/CU: small M / narrow N.
// Both operands are K-contiguous, so A and B fragments are single 16-byte LDS reads.  Tiles are
// staged HBM->LDS with global_load_lds (16 B per lane, no VGPR round trip) into a double-buffered
// ring; the LDS image is lane-linear, so the bank swizzle (16-B chunk index ^= (row>>1)&7 on
// 128-B rows: conflict-free for ds_read_b128's 16-lane groups) is applied to the per-lane SOURCE
// address and again on the read.  Block ids are remapped so that an XCD (private L2) owns a
// contiguous run of tiles, walked in groups of row-tiles so co-resident tiles share A/W panels.
//
// Conv mode gathers the A operand on the fly from the channels-last activation volume
// [T][H][W][Cin]: row m = output position, K = ((kt*3+kh)*3+kw)*Cin + c; the padding rules are applied
// once per row in tap tables and a per-row pointer iterator walks the K-tiles (gemm_epilogue.h).
#include <stdlib.h>
#include <string.h>

#include "gemm_epilogue.h"

namespace {

constexpr int BK = 64;

__device__ __forceinline__ void glds16(const bf16* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void*)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
struct TileCfg {
    static constexpr int NW = WAVES_M * WAVES_N;
    static constexpr int NT = NW * 64;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;     // wave tile
    static constexpr int TM = WM / 32, TN = WN / 32;               // MFMA tiles per wave
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr int A_LOADS = BM / 8 / NW, B_LOADS = BN / 8 / NW;   // 1-KiB glds per wave per tile
    static constexpr int OCC = (NT == 512) ? 2 : 2;                // min waves/SIMD for launch bounds
};

template <class CFG, int EPI, bool CONV>
__global__ __launch_bounds__(CFG::NT, 2) void gemm_kernel(const GemmParams p) {
    constexpr int TBM = CFG::A_BYTES / (BK * 2), TBN = CFG::B_BYTES / (BK * 2);
    constexpr int WAVES_N = TBN / CFG::WN;
    constexpr int TM = CFG::TM, TN = CFG::TN, AL = CFG::A_LOADS, BL = CFG::B_LOADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- block -> tile (XCD-contiguous, grouped row-tiles) ----
    const int Mt = (p.M + TBM - 1) / TBM, Nt = (p.N + TBN - 1) / TBN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GROUP = 8;
    const int per_group = GROUP * Nt;
    const int g = id / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(Mt - first_m, GROUP);
    const int rem = id - g * per_group;
    const int m0 = (first_m + rem % gsz) * TBM;
    const int n0 = (rem / gsz) * TBN;

    // ---- per-lane staging addresses ----
    const bf16* a_ptr[AL];
    const bf16* w_ptr[BL];
    ConvRow crow[CONV ? AL : 1];
    unsigned coff[CONV ? AL : 1];
    ConvIter<(CONV ? AL : 1)> cit;          // conv: source pointers of the NEXT K-tile to stage
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const int rt = (wv * AL + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rt >> 1) & 7);
        const int m = min(m0 + rt, p.M - 1);
        if (CONV) {
            crow[CONV ? j : 0] = conv_row_setup(p, m);
            coff[CONV ? j : 0] = chunk * 16;
            a_ptr[j] = p.A;
        } else {
            a_ptr[j] = p.A + (long)m * p.lda + chunk * 8;
        }
    }
#pragma unroll
    for (int j = 0; j < BL; ++j) {
        const int rt = (wv * BL + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rt >> 1) & 7);
        const int n = min(n0 + rt, p.N - 1);
        w_ptr[j] = p.W + (long)n * p.K + chunk * 8;
    }

    auto stage = [&](int kt, int buf) {
        char* sa = smem + buf * CFG::STAGE_BYTES + wv * (AL * 1024);
        char* sb = smem + buf * CFG::STAGE_BYTES + CFG::A_BYTES + wv * (BL * 1024);
        const int k0 = kt * BK;
        if (CONV) {             // K-tiles are staged strictly in order: issue the iterator's tile, step it
#pragma unroll
            for (int j = 0; j < AL; ++j) glds16(cit.ptr[CONV ? j : 0], sa + j * 1024);
            if (kt + 1 < p.K / BK) cit.next(p, crow, coff);
        } else {
#pragma unroll
            for (int j = 0; j < AL; ++j) glds16(a_ptr[j] + k0, sa + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < BL; ++j) glds16(w_ptr[j] + k0, sb + j * 1024);
    };

    // ---- MFMA fragment read offsets (chunk(ks) = (2*ks) ^ xbase, the staging swizzle applied again) ----
    const int wr = wv / WAVES_N, wc = wv % WAVES_N;
    const int l31 = lane & 31, hi = lane >> 5;
    const int xbase = hi ^ ((l31 >> 1) & 7);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    unsigned a_off[4], b_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned c = ((2 * ks) ^ xbase) << 4;
        a_off[ks] = (wr * CFG::WM + l31) * 128 + c;
        b_off[ks] = CFG::A_BYTES + (wc * CFG::WN + l31) * 128 + c;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Fragment stream, double-buffered across the four k-steps of a K-tile and issued by hand: hipcc guards
    // ds_read_b128 results with `s_waitcnt lgkmcnt(0)` only, which drains every read in flight, so a
    // compiler-scheduled loop pays the LDS latency once per k-step.  Here R(ks+1) goes out before M(ks) and
    // MFMA (i, j) waits with a counted lgkmcnt for exactly the reads it consumes (LGKM returns in order for
    // LDS operations; read order inside a k-step: b0, a0, b1.., a1..).
    constexpr int R = TM + TN;
    u32x4 fa[2][TM], fb[2][TN];
    unsigned sbase = lds0;
    auto read_ks = [&](auto KS) {
        constexpr int ks = decltype(KS)::value, s = ks & 1;
        fb[s][0] = lds_read16<0>(sbase + b_off[ks]);
        fa[s][0] = lds_read16<0>(sbase + a_off[ks]);
        static_for<1, TN>([&](auto J) { fb[s][decltype(J)::value] = lds_read16<decltype(J)::value * 4096>(sbase + b_off[ks]); });
        static_for<1, TM>([&](auto I) { fa[s][decltype(I)::value] = lds_read16<decltype(I)::value * 4096>(sbase + a_off[ks]); });
    };
    constexpr auto pos_b = [](int j) { return j == 0 ? 0 : 1 + j; };
    constexpr auto pos_a = [](int i) { return i == 0 ? 1 : TN + i; };

    const int nk = p.K / BK;
    if (CONV) cit.init(p, crow, coff);
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        sbase = lds0 + (kt & 1) * CFG::STAGE_BYTES;
        read_ks(std::integral_constant<int, 0>{});
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        static_for<0, 4>([&](auto KS) {
            constexpr int ks = decltype(KS)::value, s = ks & 1;
            constexpr int younger = ks < 3 ? R : 0;     // reads of k-step ks+1 issued behind ours
            if constexpr (ks < 3) read_ks(std::integral_constant<int, ks + 1>{});
            static_for<0, TM * TN>([&](auto MI) {
                constexpr int m = decltype(MI)::value, i = m / TN, j = m % TN;
                if constexpr (i == 0) lds_wait<younger + R - 1 - pos_b(j)>(fb[s][j]);
                if constexpr (j == 0) lds_wait<younger + R - 1 - pos_a(i)>(fa[s][i]);
                acc[i][j] = LTX2_MFMA_32x32x16(as_bf16x8(fb[s][j]), as_bf16x8(fa[s][i]), acc[i][j], 0, 0, 0);   // C^T orientation
            });
        });
    }

    // ---- epilogue: lane owns rows (l31 per row slot) x 4-column groups (gemm_epilogue.h) ----
    f32x4 bias4[TN][4];
    f32x4 gate4[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int col = n0 + wc * CFG::WN + j * 32 + 8 * gq + 4 * hi;
            bias4[j][gq] = (p.bias && col < p.N) ? *(const f32x4*)(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            gate4[j][gq] = (EPI == EPI_RESID_GATE_F32 && p.gate_table && col < p.N) ? *(const f32x4*)(p.gate_table + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            asm volatile("" : "+v"(bias4[j][gq]));      // retire the loads once, here
            if (EPI == EPI_RESID_GATE_F32) asm volatile("" : "+v"(gate4[j][gq]));
        }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + wr * CFG::WM + i * 32 + l31;
        if (row >= p.M) continue;
        const EpiRow er = epi_row_setup<EPI>(p, row);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int col = n0 + wc * CFG::WN + j * 32 + 8 * gq + 4 * hi;
                if (col >= p.N) continue;
                const f32x4 v = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
                epi_store4<EPI>(p, er, row, col, v, bias4[j][gq], gate4[j][gq]);
            }
    }
}

using CfgBig = TileCfg<256, 256, 2, 4>;
using CfgSmall = TileCfg<128, 128, 2, 2>;
using CfgNarrow = TileCfg<128, 64, 2, 2>;      // N <= 64 (the decoder's 128 -> 48 conv_out): 1.19 -> 0.77 ms at 49x128x192 (a 256x64 tile measured the same)

template <class CFG, int EPI, bool CONV>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<CFG, EPI, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  CFG::LDS_BYTES);
    }
    constexpr int TBM = CFG::A_BYTES / (BK * 2), TBN = CFG::B_BYTES / (BK * 2);
    const int Mt = (p.M + TBM - 1) / TBM, Nt = (p.N + TBN - 1) / TBN;
    hipLaunchKernelGGL((gemm_kernel<CFG, EPI, CONV>), dim3(Mt * Nt), dim3(CFG::NT), CFG::LDS_BYTES, stream, p);
    LTX2_CHECK_LAUNCH("gemm_kernel");
    return LTX2_OK;
}

// Tile selection: the 256x256 tile needs enough tiles to fill the 256 CUs reasonably and wide N.
inline bool use_big_tile(const GemmParams& p) {
    if (p.N < 256 || p.M < 1024) return false;
    const long tiles_big = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    return tiles_big >= 160;
}

// Grids of 128..159 big tiles (the V2.3 text K/V projection: 1024 x 8192 x 4096 = 160 tiles of 224 rows) on the 4-wave kernel: one
// round of tiles costs K/64 x ~1.27 us whatever the grid, the 128x128 kernel runs the chip at ~0.75 PF/s -- take the 4-wave kernel
// only where that estimate says so (the a2v query projection, 3456 x 2048 x 4096 = 128 tiles, stays on the small tile: 68 vs ~89 us).
inline bool v4_wins_medium_grid(const GemmParams& p) {
    if (p.N < 256 || p.N % 256 != 0 || p.M < 1024) return false;
    const long t224 = (long)((p.M + 223) / 224) * (p.N / 256);
    if (t224 < 128 || t224 > 256) return false;
    const double t_v4 = (p.K / 64) * 1.27e-6 + 8e-6, t_small = 2.0 * p.M * p.N * p.K / 0.75e15 + 10e-6;
    return t_v4 < t_small;
}

// The whole dispatch, as data: gemm_launch follows it, ltx2_gemm_route reports it (round 3: the LTX2_GEMM_TILE / LTX2_V4_LAYOUT /
// LTX2_PP_BM / LTX2_VT_FUSE overrides are gone -- same-box A/B runs load a second build through LTX2HIP_LIB instead).
template <bool CONV>
int route_of(const GemmParams& p, int epi) {
    if (p.A8) return gemm_v4_f8_supported(p, epi) ? (gemm_v4_prefer_224(p) ? ROUTE_V4_F8_224 : ROUTE_V4_F8_256) : ROUTE_INVALID;
    if (p.W8) {
        if (CONV) return ROUTE_INVALID;
        if (gemm_skinny_supported(p, epi)) return ROUTE_SKINNY;
        return gemm_v4_w8_supported(p, epi) ? (gemm_v4_prefer_224(p) ? ROUTE_V4_W8_224 : ROUTE_V4_W8_256) : ROUTE_INVALID;
    }
    if (!CONV && epi != EPI_D2S_BF16 && gemm_skinny_supported(p, epi)) return ROUTE_SKINNY;      // M <= 128: the audio stream
    if (!CONV && (use_big_tile(p) || v4_wins_medium_grid(p)) && gemm_v4_supported(p, epi, CONV)) return gemm_v4_prefer_224(p) ? ROUTE_V4_224 : ROUTE_V4_256;
    if (use_big_tile(p)) return ROUTE_PP;
    if (p.N <= 64 && p.M >= 4096) return ROUTE_NARROW;
    return ROUTE_SMALL;
}

template <int EPI, bool CONV>
int launch_t(const GemmParams& p, hipStream_t stream) {
    switch (route_of<CONV>(p, EPI)) {
        case ROUTE_SKINNY: return gemm_skinny_launch(p, EPI, stream);
        case ROUTE_V4_224: return gemm_v4_launch(p, EPI, stream, 3, 224);
        case ROUTE_V4_256: return gemm_v4_launch(p, EPI, stream, 3, 256);
        case ROUTE_PP: return gemm_pp_launch(p, EPI, CONV, stream);
        case ROUTE_NARROW: return launch_cfg<CfgNarrow, EPI, CONV>(p, stream);
        default: return launch_cfg<CfgSmall, EPI, CONV>(p, stream);
    }
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

bool gemm_vt_fused(const GemmParams& p, int epilogue) {
    if (!p.vt || p.lda % 8 != 0) return false;
    if (p.A8) return gemm_v4_vt_supported(p, epilogue, 5);
    if (gemm_skinny_supported(p, epilogue)) return false;      // M <= 128 goes to the skinny kernel
    if (p.W8) return gemm_v4_vt_supported(p, epilogue, 3);
    return (use_big_tile(p) || v4_wins_medium_grid(p)) && gemm_v4_vt_supported(p, epilogue, 3);
}

bool gemm_rowss_supported(const GemmParams& p, int epilogue) {
    if (epilogue != EPI_BF16 || p.vt || p.N % 64 != 0) return false;
    const int r = route_of<false>(p, epilogue);
    return r == ROUTE_V4_224 || r == ROUTE_V4_256 || r == ROUTE_V4_W8_224 || r == ROUTE_V4_W8_256 || r == ROUTE_V4_F8_224 || r == ROUTE_V4_F8_256;
}

bool gemm_fold_supported(const GemmParams& p, int epilogue) {
    if (p.W8 || p.A8 || !p.W) return false;
    const int r = route_of<false>(p, epilogue);
    if (r != ROUTE_V4_224 && r != ROUTE_V4_256) return false;
    const int bm = r == ROUTE_V4_224 ? 224 : 256;
    if (p.shadow) {         // producer: the gated-residual epilogue with a row-invariant gate
        if (epilogue != EPI_RESID_GATE_F32 || (p.gate && p.gate_stride != 0) || !p.shadow_ss || p.ld_shadow % 4 != 0 || ((uintptr_t)p.shadow & 7)) return false;
        if (p.ld_ss % 4 != 0 || p.ld_ss < (long)((p.M + bm - 1) / bm) * bm || ((uintptr_t)p.shadow_ss & 15)) return false;
    }
    if (p.rf_parts) {
        if (epilogue != EPI_BF16 && epilogue != EPI_GELU_BF16) return false;
        if ((p.rf_nparts < 1 || p.rf_nparts > GEMM_RF_MAX_PARTS || p.rf_dim < 1 || p.rf_ld % 4 != 0 || p.rf_ld < (long)((p.M + bm - 1) / bm) * bm ||
                           ((uintptr_t)p.rf_parts & 15) || (long)p.rf_nparts * p.rf_ld * 4 >= (1L << 31))) return false;
    }
    return true;
}

int gemm_route(const GemmParams& p, int epilogue, bool conv) { return conv ? route_of<true>(p, epilogue) : route_of<false>(p, epilogue); }

int gemm_launch(const GemmParams& p, int epilogue, bool conv, hipStream_t stream) {
    LTX2_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    LTX2_CHECK_ARG(!p.vt || (!conv && gemm_vt_fused(p, epilogue)), "gemm: a fused V^T output needs the 4-wave layout-3 / layout-5 kernel (ask gemm_vt_fused first)");
    LTX2_CHECK_ARG(p.K % BK == 0, "gemm: K=%d must be a multiple of %d", p.K, BK);
    LTX2_CHECK_ARG(!p.rowss || (!conv && gemm_rowss_supported(p, epilogue)), "gemm: row partial sums need the 4-wave kernel's bf16 epilogue (ask gemm_rowss_supported first)");
    LTX2_CHECK_ARG(!(p.shadow || p.rf_parts) || (!conv && gemm_fold_supported(p, epilogue)), "gemm: a folded norm (shadow / rf_parts) needs the 4-wave layout-3 kernel on dense bf16 weights (ask gemm_fold_supported first)");
    if (p.A8) {     // fp8 compute: both operands e4m3fn codes + scales, fp8 MFMA (gemm_v4.hip layout 5)
        LTX2_CHECK_ARG(!conv && p.out, "gemm: fp8 compute is dense-only");
        return gemm_v4_launch(p, epilogue, stream, 0, 0);        // layout 6 (16x16x128 blocks) for 224-row tiles, 5 (32x32x64) for 256-row ones
    }
    LTX2_CHECK_ARG(p.A && (p.W || p.W8) && p.out, "gemm: null operand");
    if (p.W8) {     // fp8-resident weights: the 4-wave asm-loop kernel, or the skinny-M kernel where the bf16 path would take it too
        LTX2_CHECK_ARG(!conv && p.lda % 8 == 0, "gemm: fp8-resident weights are dense-only");
        if (gemm_skinny_supported(p, epilogue)) return gemm_skinny_launch(p, epilogue, stream);
        return gemm_v4_launch(p, epilogue, stream, 3, 0);
    }
    LTX2_CHECK_ARG(p.N % 4 == 0 && p.ldo % 4 == 0 && p.ldres % 4 == 0 && p.gate_stride % 4 == 0,
                   "gemm: N, ldo, ldres and gate_stride must be multiples of 4 (vector epilogue)");
    if (conv) {
        LTX2_CHECK_ARG(p.Cin >= 64 && (p.Cin & (p.Cin - 1)) == 0, "conv3d: Cin=%d must be a power of two >= 64", p.Cin);
        LTX2_CHECK_ARG(p.taps_t == 3 || p.taps_t == 1, "conv3d: temporal kernel size %d (3 or 1)", p.taps_t);
        LTX2_CHECK_ARG(p.K == 9 * p.taps_t * p.Cin, "conv3d: K=%d != 9*kt*Cin", p.K);
        LTX2_CHECK_ARG(p.T >= 1 && (p.pad_zero || (p.H >= 2 && p.Wd >= 2)), "conv3d: reflect padding needs H,W >= 2");
        LTX2_CHECK_ARG((long)p.T * p.H * p.Wd == p.M, "conv3d: M != T*H*W");
        LTX2_CHECK_ARG((long)p.M * p.Cin * 2 < (1L << 31), "conv3d: activation volume must be < 2 GiB (32-bit tap offsets); decode in tiles");
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
