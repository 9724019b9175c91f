// Skinny-M dense GEMM for gfx950: out[M][N] = epilogue(A[M][K] @ W[N][K]^T + bias), M <= 128.
//
// The audio stream of the AudioVideo DiT has 68 tokens (reference LTX_2_MLX/model/transformer/transformer.py:457-648: the
// audio half of every BasicAVTransformerBlock -- to_q/to_k/to_v/to_out of audio_attn1/2, the audio sides of the cross-modal
// attentions, audio_ff).  On the 128x128 tile kernel such a GEMM is ONE row of N/128 = 16..64 workgroups on 256 CUs streaming
// the weight matrix at a tenth of the HBM rate (60-85 us per launch, 11 launches per layer, four of them on the video
// stream's critical path).  Here the weight matrix is the only operand that matters, so the grid is cut along N into
// 16- or 32-column strips (128..512 workgroups) and K is cut over the 4 or 8 waves of a workgroup:
//   * each lane loads its W fragment (16 B = 8 k of one column) straight from global memory -- no LDS, nothing is shared --
//     and the activation fragments of the <= 8 row blocks from L2 (A is M x K bf16, a few hundred KB);
//   * v_mfma_f32_16x16x32_bf16 in the C^T orientation of the tile kernels (A operand = W fragment): a lane ends up with one
//     output row and 4 consecutive columns, so the epilogues of gemm_epilogue.h apply unchanged;
//   * the K-partials of the other waves meet wave 0's in LDS and are added in wave order (deterministic); wave 0 runs the epilogue.
// HBM-bound by construction: bytes = N*K*2 (+ out); the MFMA pipes idle.
#include "gemm_epilogue.h"

namespace {

constexpr int SK_MAXRB = 8;      // row blocks of 16: M <= 128

// NB: 16-column blocks per workgroup strip; NW: waves per workgroup = K slices (8 where the grid alone cannot fill the chip: the
// loop is a chain of dependent global-load round trips, so halving a wave's K range halves the launch time)
// W8: fp8-resident weights (p.W8 codes + p.wscale per output column): a lane's 8 codes are expanded to bf16(f32(code) * scale) --
// the load-time dequantiser's arithmetic -- so the result is bit-identical to this kernel on the dequantised weights.
template <int EPI, int NB, int NW, bool W8>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(const GemmParams p) {
    __shared__ f32x4 red[NW - 1][SK_MAXRB][NB][64];      // partials of waves 1.. (wave 0 keeps its own)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (16 * NB);
    const int nrb = (p.M + 15) >> 4;                      // block-uniform
    const int kw = p.K / NW;                              // K range of this wave: [w * kw, (w + 1) * kw), kw % 64 == 0
    const bf16* wp[NB];                                   // W8: byte pointers into the code matrix, kept in the same array
    float wsc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const long e = (long)(n0 + nb * 16 + c) * p.K + w * kw + 8 * g;
        wp[nb] = W8 ? (const bf16*)(p.W8 + e) : p.W + e;
        wsc[nb] = W8 ? p.wscale[n0 + nb * 16 + c] : 1.f;
    }
    // fragment of k-step offset `ko` (elements) of column block nb
    auto load_w = [&](int nb, int ko) __attribute__((always_inline)) -> bf16x8 {
        if constexpr (W8) {
            const u32x2 cw = __builtin_nontemporal_load((const u32x2*)((const unsigned char*)wp[nb] + ko));
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(e4m3fn_to_f32((cw[e >> 2] >> (8 * (e & 3))) & 0xffu) * wsc[nb]);
            return o;
        } else {
            return __builtin_nontemporal_load((const bf16x8*)(wp[nb] + ko));     // streamed once
        }
    };
    const bf16* ap[SK_MAXRB];
#pragma unroll
    for (int rb = 0; rb < SK_MAXRB; ++rb) ap[rb] = p.A + (long)min(rb * 16 + c, p.M - 1) * p.lda + w * kw + 8 * g;

    f32x4 acc[SK_MAXRB][NB];
#pragma unroll
    for (int rb = 0; rb < SK_MAXRB; ++rb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The weight stream is the only HBM traffic and a wave's loop is a chain of dependent round trips, so the bytes in flight set
    // the rate: 8 k-steps (256 k) of W fragments are requested up front per chunk (8 NB x 1 KB per wave); the activation
    // fragments (L2-resident) follow two k-steps at a time, just ahead of their MFMAs.
    auto two_ksteps = [&](int k, const bf16x8* wf) __attribute__((always_inline)) {
        bf16x8 af[2][SK_MAXRB];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int rb = 0; rb < SK_MAXRB; ++rb)
                if (rb < nrb) af[s][rb] = *(const bf16x8*)(ap[rb] + k + 32 * s);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int rb = 0; rb < SK_MAXRB; ++rb)
                if (rb < nrb) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[rb][nb] = LTX2_MFMA_16x16x32(wf[s * NB + nb], af[s][rb], acc[rb][nb], 0, 0, 0);
                }
    };
    int k = 0;
    for (; k + 256 <= kw; k += 256) {
        bf16x8 wf[8 * NB];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wf[s * NB + nb] = load_w(nb, k + 32 * s);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) two_ksteps(k + 64 * s2, wf + 2 * NB * s2);
    }
    for (; k < kw; k += 64) {
        bf16x8 wf[2 * NB];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wf[s * NB + nb] = load_w(nb, k + 32 * s);
        two_ksteps(k, wf);
    }

    // ---- K-partials: waves 1.. park theirs in LDS; wave 0 adds them in wave order and runs the epilogue ----
    if (w > 0) {
#pragma unroll
        for (int rb = 0; rb < SK_MAXRB; ++rb)
            if (rb < nrb) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) red[w - 1][rb][nb][lane] = acc[rb][nb];
            }
    }
    __syncthreads();
    if (w != 0) return;
    f32x4 bias4[NB], gate4[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = n0 + nb * 16 + 4 * g;
        bias4[nb] = p.bias ? *(const f32x4*)(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        gate4[nb] = (EPI == EPI_RESID_GATE_F32 && p.gate_table) ? *(const f32x4*)(p.gate_table + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int rb = 0; rb < SK_MAXRB; ++rb) {
        if (rb >= nrb) continue;
        const int row = rb * 16 + c;
        const EpiRow er = epi_row_setup<EPI>(p, row);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 v = acc[rb][nb];
#pragma unroll
            for (int q = 0; q < NW - 1; ++q) v += red[q][rb][nb][lane];
            if (row < p.M) epi_store4<EPI>(p, er, row, n0 + nb * 16 + 4 * g, v, bias4[nb], gate4[nb]);
        }
    }
}

template <int EPI, bool W8>
int launch_skinny(const GemmParams& p, hipStream_t stream) {
    // 32-column strips while that still gives every CU a workgroup, else 16-column strips, with K over 8 waves where it divides
    if (p.N % 32 == 0 && p.N / 32 >= 256)
        hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 2, 4, W8>), dim3(p.N / 32), dim3(256), 0, stream, p);
    else if (p.K % 512 == 0)
        hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 1, 8, W8>), dim3(p.N / 16), dim3(512), 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 1, 4, W8>), dim3(p.N / 16), dim3(256), 0, stream, p);
    LTX2_CHECK_LAUNCH("gemm_skinny_kernel");
    return LTX2_OK;
}

}  // namespace

bool gemm_skinny_supported(const GemmParams& p, int epilogue) {
    if (!(p.W || (p.W8 && p.wscale)) || p.M < 1 || p.M > 16 * SK_MAXRB) return false;
    if (p.N % 16 != 0 || p.K % 256 != 0) return false;                   // K/4 per wave in 64-k iterations
    if (p.lda % 8 != 0 || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || ((uintptr_t)p.W8 & 7)) return false;
    if (p.ldo % 4 != 0 || ((uintptr_t)p.out & 15)) return false;
    switch (epilogue) {
        case EPI_BF16: case EPI_GELU_BF16: case EPI_SILU_BF16: case EPI_F32: case EPI_RESID_GATE_F32: case EPI_ADD_BF16: return true;
        default: return false;
    }
}

int gemm_skinny_launch(const GemmParams& p, int epilogue, hipStream_t stream) {
    LTX2_CHECK_ARG(gemm_skinny_supported(p, epilogue), "gemm_skinny: unsupported problem (M=%d N=%d K=%d epilogue=%d)", p.M, p.N, p.K, epilogue);
#define CASE(E) \
    case E:     \
        return p.W8 ? launch_skinny<E, true>(p, stream) : launch_skinny<E, false>(p, stream);
    switch (epilogue) {
        CASE(EPI_BF16)
        CASE(EPI_GELU_BF16)
        CASE(EPI_SILU_BF16)
        CASE(EPI_F32)
        CASE(EPI_RESID_GATE_F32)
        CASE(EPI_ADD_BF16)
    }
#undef CASE
    return LTX2_E_INVALID;
}
