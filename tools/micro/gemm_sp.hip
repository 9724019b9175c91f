// EXPERIMENT (not part of libltx2hip.so): 256x256x64 bf16 MFMA GEMM for gfx950 with a hand-pipelined
// LDS->register fragment stream ("sp"), 4 waves x 128x128 or 8 waves x 128x64.  Correct (passes the gemm
// parity tests when wired in), 0.95x the ping-pong kernel.  What it measured on MI355X (8192^3, bf16):
//   MFMA stream alone 1.56 PF/s; + fragment reads 1.47; + LDS-DMA staging 1.16 -- i.e. the 16 one-KiB
//   VMEM issues per wave per K-tile cost ~46 cycles each, the same for global_load_lds, buffer_load..lds
//   and plain global_load_dwordx4, and with one wave per SIMD nothing covers them.  Staggering the four waves' issues
//   (-DSP_STAGGER: wave w issues in MFMA slots g % 4 == w, one specialised K loop per wave) is 4 % slower still, so the
//   cost is not the waves queueing behind each other at the TA.  See DESIGN.md.
//
// Same operands, LDS image, swizzle, tile order and epilogues as gemm.hip / gemm_pp.hip (dense only).
// What differs is who schedules the inner loop.  hipcc only ever guards ds_read_b128 results with
// `s_waitcnt lgkmcnt(0)`, which drains every fragment read in flight, so a compiler-scheduled K loop
// pays the LDS latency once per read group (gemm.hip) unless a partner wave covers it by construction
// (gemm_pp.hip: two staggered groups, four barriers per K-tile).  Here the fragment reads and their
// COUNTED waits are issued by hand (asm), double-buffered across the four k-steps of a K-tile:
//
//     K-tile t:  | vmcnt(0), s_barrier | R(t,0) | R(t,1) M(t,0) | R(t,2) M(t,1) | R(t,3) M(t,2) | M(t,3) |
//
// R(t,ks) = the TM + TN ds_read_b128 of k-step ks, M(t,ks) = its TM x TN v_mfma_f32_32x32x16_bf16; the
// MFMA (i, j) waits with lgkmcnt(n) for exactly the reads it consumes (LGKM returns in order, so "at most
// n younger operations outstanding" proves them landed).  The LDS-DMA pieces of K-tile t+1 are issued one
// per second MFMA of M(t,0) and M(t,1): an issue costs the wave 50-150 cycles, which hides in the
// 32-cycle matrix-pipe occupancy of the neighbouring MFMAs, and the last piece keeps two k-steps of lead
// before the next barrier.  One barrier per K-tile.
//
// Two wave layouts: 2x2 waves (wave tile 128x128, 256 accumulator registers, one wave per SIMD: 32
// fragment reads per 64 MFMAs) and 2x4 waves (wave tile 128x64, two waves per SIMD).
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "gemm_epilogue.h"   // -I ltx-2-mlx_amd/csrc

#ifndef SP_ABL
#define SP_ABL 0
#endif

namespace {

constexpr int BK = 64;
constexpr int TB = 256;                     // BM = BN
constexpr int A_BYTES = TB * BK * 2;
constexpr int STAGE_BYTES = 2 * A_BYTES;    // 64 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES;  // 128 KiB

__device__ __forceinline__ void glds16(const bf16* g, unsigned lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void*)g, (lds_ptr_t)(uintptr_t)lds_wave_base, 16, 0, 0);
}
// buffer form of the LDS-DMA: SGPR resource + 32-bit per-lane byte offset + SGPR offset
__device__ __forceinline__ void bufl_lds16(const void* base, unsigned lds_wave_base, unsigned voff, unsigned soff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(uintptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
template <int NWN, int EPI>
__global__ __launch_bounds__(NWN * 128, NWN == 2 ? 1 : 2) void gemm_sp_kernel(const GemmParams p) {
    constexpr int NW = 2 * NWN;                 // waves: 2 along M x NWN along N
    constexpr int WN = TB / NWN;                // wave tile 128 x WN
    constexpr int TM = 4, TN = WN / 32;
    constexpr int AL = 32 / NW, BL = 32 / NW;   // 1-KiB LDS-DMA pieces per wave per K-tile (A, B)
    constexpr int R = TM + TN;                  // fragment reads per k-step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- block -> tile (XCD-contiguous, grouped row-tiles) ----
    const int Mt = (p.M + TB - 1) / TB, Nt = (p.N + TB - 1) / TB;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GROUP = 8;
    const int per_group = GROUP * Nt;
    const int g = id / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(Mt - first_m, GROUP);
    const int rem = id - g * per_group;
    const int m0 = (first_m + rem % gsz) * TB;
    const int n0 = (rem / gsz) * TB;

    // ---- per-lane staging sources (row clamp at the ragged edge; chunk swizzle on the source) ----
    const bf16* src[AL + BL];
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const int rt = (wv * AL + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rt >> 1) & 7);
        src[j] = p.A + (long)min(m0 + rt, p.M - 1) * p.lda + chunk * 8;
    }
#pragma unroll
    for (int j = 0; j < BL; ++j) {
        const int rt = (wv * BL + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rt >> 1) & 7);
        src[AL + j] = p.W + (long)min(n0 + rt, p.N - 1) * p.K + chunk * 8;
    }
    // piece q of K-tile kt into stage `buf`: pieces [0, AL) are A rows, [AL, AL+BL) are B rows
    u32x4 sink = {0, 0, 0, 0};
    u32x4 ld[AL + BL];
    unsigned voff[AL + BL];
#pragma unroll
    for (int q = 0; q < AL + BL; ++q) { voff[q] = (unsigned)((const char*)src[q] - (const char*)(q < AL ? p.A : p.W)); ld[q] = sink; }
    auto stage_piece = [&](int kt, int buf, int q) {
        const unsigned dst = lds0 + buf * STAGE_BYTES + (q < AL ? wv * (AL * 1024) + q * 1024 : A_BYTES + wv * (BL * 1024) + (q - AL) * 1024);
        if (SP_ABL == 5) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[q]) : "v"(src[q] + kt * BK) : "memory");
        } else if (SP_ABL == 6) {
            bufl_lds16(q < AL ? (const void*)p.A : (const void*)p.W, dst, voff[q], kt * BK * 2);
        } else {
            glds16(src[q] + kt * BK, dst);
        }
    };

    // ---- fragment read addresses: row part + swizzled chunk per k-step; the stage bit is toggled per K-tile ----
    const int wr = wv / NWN, wc = wv % NWN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int xbase = hi ^ ((l31 >> 1) & 7);          // chunk(ks) = (2*ks) ^ xbase
    unsigned a_addr[4], b_addr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned c = ((2 * ks) ^ xbase) << 4;
        a_addr[ks] = lds0 + (wr * 128 + l31) * 128 + c;
        b_addr[ks] = lds0 + A_BYTES + (wc * WN + l31) * 128 + c;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment double buffer; read order inside a k-step: b0, a0, b1.., a1..  (MFMA (0,0) first)
    u32x4 fa[2][TM], fb[2][TN];
    auto read_ks = [&](auto KS) {
        constexpr int ks = decltype(KS)::value, s = ks & 1;
        fb[s][0] = lds_read16<0>(b_addr[ks]);
        fa[s][0] = lds_read16<0>(a_addr[ks]);
        static_for<1, TN>([&](auto J) { fb[s][decltype(J)::value] = lds_read16<decltype(J)::value * 4096>(b_addr[ks]); });
        static_for<1, TM>([&](auto I) { fa[s][decltype(I)::value] = lds_read16<decltype(I)::value * 4096>(a_addr[ks]); });
    };
    // position of a fragment in that order
    constexpr auto pos_b = [](int j) { return j == 0 ? 0 : 1 + j; };
    constexpr auto pos_a = [](int i) { return i == 0 ? 1 : TN + i; };

    const int nk = p.K / BK;
#pragma unroll
    for (int q = 0; q < AL + BL; ++q) stage_piece(0, 0, q);

    // SP_STAGGER: one specialised copy of the K loop per wave slot (wv & 3), so that wave w issues its LDS-DMA pieces in
    // MFMA slots g with g % 4 == w -- statically, no per-MFMA branch -- and the four waves of a SIMD row never present
    // VMEM instructions to the TA in the same slot.
    auto run_loop = [&](auto WS) __attribute__((always_inline)) {
        constexpr int WSLOT = decltype(WS)::value;
        (void)WSLOT;
    for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (SP_ABL != 3 && SP_ABL != 4) __syncthreads();
            if (SP_ABL == 5) {
#pragma unroll
                for (int q = 0; q < AL + BL; ++q) { asm volatile("" : "+v"(ld[q])); sink ^= ld[q]; }
            }
            const int ktn = min(kt + 1, nk - 1);        // the last tile re-stages itself into the idle buffer (branch-free body)
            const int nbuf = (kt + 1) & 1;
            if (SP_ABL != 4 || kt == 0) read_ks(std::integral_constant<int, 0>{});
            static_for<0, 4>([&](auto KS) {
                constexpr int ks = decltype(KS)::value, s = ks & 1;
                constexpr int younger = ks < 3 ? R : 0;     // reads of k-step ks+1 issued behind ours
                if constexpr (ks < 3) { if (SP_ABL != 4 || kt == 0) read_ks(std::integral_constant<int, ks + 1>{}); }
                static_for<0, TM * TN>([&](auto MI) {
                    constexpr int m = decltype(MI)::value, i = m / TN, j = m % TN;
                    if constexpr (i == 0 && SP_ABL != 4) lds_wait<younger + R - 1 - pos_b(j)>(fb[s][j]);
                    if constexpr (j == 0 && SP_ABL != 4) lds_wait<younger + R - 1 - pos_a(i)>(fa[s][i]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fb[s][j]), as_bf16x8(fa[s][i]), acc[i][j], 0, 0, 0);   // C^T orientation
#ifndef SP_SPREAD
#define SP_SPREAD 2         // one LDS-DMA piece per SP_SPREAD MFMAs, starting with M(t,0)
#endif
#ifdef SP_STAGGER
                    // wave w issues in MFMA slots g with g % 4 == w % 4: the four waves of a CU never present their
                    // VMEM instructions to the TA at the same time
                    {
                        constexpr int g = ks * TM * TN + m, q = g / 4;
                        if constexpr (q < AL + BL && SP_ABL != 1 && SP_ABL != 4) {
                            if constexpr ((g & 3) == WSLOT) stage_piece(SP_ABL == 2 ? 0 : ktn, nbuf, q);
                        }
                    }
#else
                    if constexpr ((ks * TM * TN + m) % SP_SPREAD == SP_SPREAD - 1) {
                        constexpr int q = (ks * TM * TN + m) / SP_SPREAD;
                        if constexpr (q < AL + BL && SP_ABL != 1 && SP_ABL != 4) stage_piece(SP_ABL == 2 ? 0 : ktn, nbuf, q);
                    }
#endif
                });
            });
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                a_addr[ks] ^= STAGE_BYTES;
                b_addr[ks] ^= STAGE_BYTES;
            }
        }
    };
#ifdef SP_STAGGER
    switch (wv & 3) {
        case 0: run_loop(std::integral_constant<int, 0>{}); break;
        case 1: run_loop(std::integral_constant<int, 1>{}); break;
        case 2: run_loop(std::integral_constant<int, 2>{}); break;
        default: run_loop(std::integral_constant<int, 3>{}); break;
    }
#else
    run_loop(std::integral_constant<int, 0>{});
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SP_ABL == 5 && sink[0] == 0x12345678 && sink[1] == 77 && sink[2] == 5 && sink[3] == 9) acc[0][0][0] += 1.f;

    // ---- epilogue: lane owns rows (l31 per row slot) x 4-column groups (gemm_epilogue.h) ----
    f32x4 bias4[TN][4];
    f32x4 gate4[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int col = n0 + wc * WN + j * 32 + 8 * gq + 4 * hi;
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
        const int row = m0 + wr * 128 + i * 32 + l31;
        if (row >= p.M) continue;
        const EpiRow er = epi_row_setup<EPI>(p, row);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int col = n0 + wc * WN + j * 32 + 8 * gq + 4 * hi;
                if (col >= p.N) continue;
                const f32x4 v = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
                epi_store4<EPI>(p, er, row, col, v, bias4[j][gq], gate4[j][gq]);
            }
    }
}

template <int NWN, int EPI>
int launch_sp(const GemmParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_sp_kernel<NWN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const int Mt = (p.M + TB - 1) / TB, Nt = (p.N + TB - 1) / TB;
    hipLaunchKernelGGL((gemm_sp_kernel<NWN, EPI>), dim3(Mt * Nt), dim3(NWN * 128), LDS_BYTES, stream, p);
    LTX2_CHECK_LAUNCH("gemm_sp_kernel");
    return LTX2_OK;
}

}  // namespace

int gemm_sp_launch(const GemmParams& p, int epilogue, int waves, hipStream_t stream) {
#define CASE(E) \
    case E:     \
        return waves == 4 ? launch_sp<2, E>(p, stream) : launch_sp<4, E>(p, stream);
    switch (epilogue) {
        CASE(EPI_BF16)
        CASE(EPI_GELU_BF16)
        CASE(EPI_SILU_BF16)
        CASE(EPI_F32)
        CASE(EPI_RESID_GATE_F32)
        CASE(EPI_ADD_BF16)
        default:
            ltx2_set_error("gemm_sp: unsupported epilogue %d", epilogue);
            return LTX2_E_INVALID;
    }
#undef CASE
}
