// 256x256x64 (and 224x256x64) "ping-pong" bf16 MFMA GEMM / implicit-GEMM conv3d for gfx950.
//
// 8 wave64 (2 row groups x 4 column waves), wave tile 128x64, one workgroup per CU, 128 KiB of
// LDS (2 K-tile buffers).  The two row groups run the same program staggered by ONE barrier, so
// on every SIMD one wave is in a pure-MFMA interval while its partner is in an LDS-read /
// LDS-DMA-issue interval, and the matrix pipe never waits on ds_read latency:
//
//   group 0:      La | Ma | Lb | Mb | La' ...
//   group 1:  --  |  La | Ma | Lb | Mb ...                          ( | = s_barrier )
//
// A K-tile (A 256x64 + B 256x64) is staged as four 16-KiB half-tiles: A0/A1 = the first/second
// 64 rows of each wave's 128-row slab, B0/B1 = the first/second 32 columns of each wave's
// 64-column slab.  Per K-tile each wave runs two L/M pairs (4 barriers; a barrier event costs
// ~110 cycles on this chip, so intervals are sized at 16 MFMAs = 512 matrix-pipe cycles):
//        La: read A0,B0,B1 (16 x ds_read_b128)
//        Ma: 16 x v_mfma_f32_32x32x16_bf16 (rows [0,64) x cols [0,64) of the wave slab, K=64)
//            with the LDS-DMA of A1(t+1) (2 x global_load_lds, 16 B/lane) interleaved
//        Lb: read A1 (8 x ds_read_b128)               + s_waitcnt vmcnt(2)
//        Mb: 16 MFMAs (rows [64,128)) with the LDS-DMA of A0,B0,B1(t+2) (6) interleaved
//                                                     + s_waitcnt vmcnt(6)
// Measured on MI355X: issuing one LDS-DMA instruction costs 60-150 cycles and stretches an L
// interval; inside an M interval it hides behind the 32-cycle pipe occupancy of the MFMAs.
// Loads stay in flight across barriers (raw s_barrier, never vmcnt(0) in the loop):
//   * vmcnt(2) at the end of Lb(t) retires this wave's share of A0,B0,B1(t+1) one barrier before
//     the partner group reads them in its La(t+1);
//   * vmcnt(6) at the end of Mb(t) retires A1(t+1), leaving A0,B0,B1(t+2) outstanding.
// Each L interval ends with lgkmcnt(0) BEFORE its barrier and every slot is refilled at least
// two intervals after its last read by either group (WAR).  Out-of-range K-tiles in the tail are
// clamped to the last tile (same slots, dead data), so the wait counts are uniform.
//
// BM = 224 variant (tile-quantisation fix): the DiT's M = 3456 tokens are 13.5 tiles of 256 rows,
// so with N = 4096 a 256-row grid is 224 tiles on 256 CUs (and 3.5 waves for N = 16384); 224-row
// tiles give 16 x 16 = 256 tiles of 0.875 the work each.  Row group 1 then owns 96 rows (three
// 32-row blocks b0,b1 in A0 and b2 in A1) and runs 12 + 12 MFMAs per K-tile:
//        La : as group 0 (A0 = b0,b1 ; B0,B1)      Ma': b0 x ks0..3 (8) + b1 x ks0,1 (4)
//        Lb': read b2 (4 x ds_read_b128)            Mb': b1 x ks2,3 (4) + b2 x ks0..3 (8)
// with the same barriers, DMA issue slots and wait counts as group 0 (the staged A1 half-tile
// keeps 128 rows; its last 32 are never read).
// LDS swizzle / XCD-aware tile order as in gemm.hip.
#include <stdlib.h>

#include "gemm_epilogue.h"

namespace {

constexpr int BK = 64;
constexpr int TBN = 256;                // BN
constexpr int HALF = 16384;             // one half-tile
constexpr int BUF = 4 * HALF;           // A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * BUF;      // 128 KiB

__device__ __forceinline__ void glds16(const bf16* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void*)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

#define PP_BARRIER()                            \
    do {                                        \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_s_barrier();           \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

template <int EPI, bool CONV, int BM>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmParams p) {
    static_assert(BM == 256 || BM == 224, "row tile is 256 or 224");
    constexpr bool ASYM = BM == 224;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3;

    // ---- block -> tile (XCD-contiguous, grouped row-tiles) ----
    const int Mt = (p.M + BM - 1) / BM, Nt = (p.N + TBN - 1) / TBN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GROUP = 8;
    const int per_group = GROUP * Nt;
    const int g = id / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(Mt - first_m, GROUP);
    const int rem = id - g * per_group;
    const int m0 = (first_m + rem % gsz) * BM;
    const int n0 = (rem / gsz) * TBN;

    // ---- staging addresses: per half-tile kind, 2 LDS-DMA instructions per wave ----
    // LDS row r' = (wv*2 + j)*8 + lane/8 of the 128-row half-tile; chunk swizzle on the source.
    const bf16* a_src[2][2];
    const bf16* b_src[2][2];
    ConvRow crow[CONV ? 2 : 1][CONV ? 2 : 1];
    unsigned coff[CONV ? 2 : 1][CONV ? 2 : 1];
    // conv: one K-walk iterator per half-tile kind (gemm_epilogue.h) -- cit[1] runs one K-tile ahead of the loop
    // (A1(t+1) is issued in Ma), cit[0] two ahead (A0(t+2) in Mb); both are stepped inside L intervals, so the M
    // intervals carry nothing but MFMAs and the LDS-DMA issues themselves.
    ConvIter<(CONV ? 2 : 1)> cit[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rp = (wv * 2 + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rp >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int arow = (rp >> 6) * 128 + h * 64 + (rp & 63);
            const int m = min(m0 + arow, p.M - 1);
            if (CONV) {
                crow[CONV ? h : 0][CONV ? j : 0] = conv_row_setup(p, m);
                coff[CONV ? h : 0][CONV ? j : 0] = chunk * 16;
                a_src[h][j] = p.A;
            } else {
                a_src[h][j] = p.A + (long)m * p.lda + chunk * 8;
            }
            const int bcol = (rp >> 5) * 64 + h * 32 + (rp & 31);
            const int n = min(n0 + bcol, p.N - 1);
            b_src[h][j] = p.W + (long)n * p.K + chunk * 8;
        }
    }
    const int nk = p.K / BK;

    auto issue_a = [&](int h, int tile) {
        char* dst = smem + (tile & 1) * BUF + h * HALF + wv * 2048;
        const int k0 = min(tile, nk - 1) * BK;
        if (CONV) {             // the iterator already stands on K-tile min(tile, nk-1)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(cit[h].ptr[CONV ? j : 0], dst + j * 1024);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(a_src[h][j] + k0, dst + j * 1024);
        }
    };
    auto issue_b = [&](int h, int tile) {
        char* dst = smem + (tile & 1) * BUF + (2 + h) * HALF + wv * 2048;
        const int k0 = min(tile, nk - 1) * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(b_src[h][j] + k0, dst + j * 1024);
    };

    // ---- fragment reads ----
    const int l31 = lane & 31, hi = lane >> 5;
    const int xbase = hi ^ ((l31 >> 1) & 7);
    const int a_off = (wr * 64 + l31) * 128;            // + i*32*128 within an A half
    const int b_off = (wc * 32 + l31) * 128;            // within a B half
    bf16x8 af[2][4], bfr[2][4];
    f32x16 acc[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][b][r] = 0.f;

    auto read_a = [&](const char* buf, int h) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                af[i][ks] = *(const bf16x8*)(buf + h * HALF + a_off + i * 4096 + (((2 * ks) ^ xbase) << 4));
    };
    auto read_a2 = [&](const char* buf) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) af[0][ks] = *(const bf16x8*)(buf + HALF + a_off + (((2 * ks) ^ xbase) << 4));   // b0's fragments are dead by now
    };
    auto read_b = [&](const char* buf, int h) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            bfr[h][ks] = *(const bf16x8*)(buf + (2 + h) * HALF + b_off + (((2 * ks) ^ xbase) << 4));
    };
// C^T orientation: A_op = W fragment, B_op = A fragment (see gemm_epilogue.h).  The empty asm
// statements pin the MFMAs inside their interval: hipcc otherwise sinks these register-only
// instructions across s_barrier (observed: 0/4/7/21 MFMAs in four M intervals of 8).
// The LDS-DMA issues of an M interval are spread between the MFMAs with sched_group_barrier:
// issuing one global_load_lds costs 60-150 cycles, which hides in the shadow of the 32-cycle
// matrix-pipe occupancy of the neighbouring MFMAs instead of stretching an L interval.
#define PP_MFMA(QA, I, QB, KS) \
    acc[QA][I][QB] = LTX2_MFMA_32x32x16(bfr[QB][KS], af[I][KS], acc[QA][I][QB], 0, 0, 0)
#define PP_MFMA2(QB, KS) \
    acc[1][0][QB] = LTX2_MFMA_32x32x16(bfr[QB][KS], af[0][KS], acc[1][0][QB], 0, 0, 0)
#define PP_PIN_IN() asm volatile("" : "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[0][2]), "+v"(bfr[0][3]))
#define PP_PIN_OUT(QA) asm volatile("" : "+v"(acc[QA][0][0]), "+v"(acc[QA][1][0]), "+v"(acc[QA][0][1]), "+v"(acc[QA][1][1]))
#define PP_PIN_OUT_B() asm volatile("" : "+v"(acc[0][1][0]), "+v"(acc[0][1][1]), "+v"(acc[1][0][0]), "+v"(acc[1][0][1]))
#define PP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define SGB_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define SGB_VMEM(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0)

    // ---- prologue: tile 0 complete + A0,B0,B1 of tile 1 in flight ----
    auto step = [&](int h) { cit[h].next(p, crow[CONV ? h : 0], coff[CONV ? h : 0]); };
    if (CONV) {
        cit[0].init(p, crow[0], coff[0]);
        cit[1].init(p, crow[CONV ? 1 : 0], coff[CONV ? 1 : 0]);
    }
    issue_a(0, 0);
    issue_b(0, 0);
    issue_b(1, 0);
    issue_a(1, 0);
    if (CONV && nk > 1) {
        step(0);
        step(1);            // cit[1] -> K-tile 1 for Ma(0)
    }
    issue_a(0, 1);
    if (CONV && nk > 2) step(0);        // cit[0] -> K-tile 2 for Mb(0)
    issue_b(0, 1);
    issue_b(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    PP_BARRIER();
    if (wr == 1) PP_BARRIER();          // stagger the second group by one interval

    // optional interval timestamps (tools/pp_timeline.py): waves 0 and 4 of block 0, K-tiles 8..11
    unsigned long long* dbg = (p.dbg && blockIdx.x == 0 && lane == 0 && (wv == 0 || wv == 4)) ? (unsigned long long*)p.dbg + (wv >> 2) * 256 : nullptr;
    int dbi = 0;
#ifdef PP_TIMELINE      // compile-time only: the branches would split the scheduling regions
#define PP_STAMP() do { if (dbg && t >= 8 && t < 12 && dbi < 256) dbg[dbi++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PP_STAMP() do { (void)dbg; (void)dbi; } while (0)
#endif
    // Two copies of the K loop (one per row-group program) rather than a branch inside one loop:
    // with the branch inside, hipcc keeps a second copy of the 128 accumulator registers across the
    // merge point and spills ~130 VGPRs.
    if (!ASYM || wr == 0) {
        for (int t = 0; t < nk; ++t) {
            const char* cb = smem + (t & 1) * BUF;
            // La: fragments of A0, B0, B1 (both groups)
            PP_STAMP();
            read_a(cb, 0);
            read_b(cb, 0);
            read_b(cb, 1);
            if (CONV && t >= 1 && t + 2 < nk) step(0);      // cit[0] -> K-tile t+2 (issued in Mb)
            PP_STAMP();
            PP_LGKM0();
            PP_STAMP();
            PP_BARRIER();
            PP_STAMP();
            // Ma: 16 MFMAs on rows [0,64) of the wave slab + LDS-DMA of A1(t+1)
            PP_PIN_IN();
            __builtin_amdgcn_s_setprio(1);
            issue_a(1, t + 1);
    #pragma unroll
            for (int qb = 0; qb < 2; ++qb)
    #pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    PP_MFMA(0, 0, qb, ks);
                    PP_MFMA(0, 1, qb, ks);
                }
            SGB_MFMA(4);
            SGB_VMEM(1);
            SGB_MFMA(6);
            SGB_VMEM(1);
            SGB_MFMA(6);
            __builtin_amdgcn_s_setprio(0);
            PP_PIN_OUT(0);
            PP_STAMP();
            PP_BARRIER();
            // Lb: fragments of A1; this wave's share of A0,B0,B1(t+1) must have landed before the barrier
            PP_STAMP();
            read_a(cb, 1);
            if (CONV && t + 2 < nk) step(1);                // cit[1] -> K-tile t+2 (issued in the next Ma)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            PP_STAMP();
            PP_LGKM0();
            PP_STAMP();
            PP_BARRIER();
            PP_STAMP();
            // Mb: 16 MFMAs on rows [64,128) + LDS-DMA of A0,B0,B1(t+2); A1(t+1) must have landed
            PP_PIN_IN();
            __builtin_amdgcn_s_setprio(1);
            issue_a(0, t + 2);
            issue_b(0, t + 2);
            issue_b(1, t + 2);
    #pragma unroll
            for (int qb = 0; qb < 2; ++qb)
    #pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    PP_MFMA(1, 0, qb, ks);
                    PP_MFMA(1, 1, qb, ks);
                }
            SGB_MFMA(2);
            SGB_VMEM(1);
            SGB_MFMA(2);
            SGB_VMEM(1);
            SGB_MFMA(2);
            SGB_VMEM(1);
            SGB_MFMA(2);
            SGB_VMEM(1);
            SGB_MFMA(2);
            SGB_VMEM(1);
            SGB_MFMA(2);
            SGB_VMEM(1);
            SGB_MFMA(4);
            __builtin_amdgcn_s_setprio(0);
            PP_PIN_OUT(1);
            PP_STAMP();
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            PP_STAMP();
            PP_BARRIER();
        }
    } else {
        for (int t = 0; t < nk; ++t) {
            const char* cb = smem + (t & 1) * BUF;
            // La: fragments of A0, B0, B1 (both groups)
            PP_STAMP();
            read_a(cb, 0);
            read_b(cb, 0);
            read_b(cb, 1);
            if (CONV && t >= 1 && t + 2 < nk) step(0);      // cit[0] -> K-tile t+2 (issued in Mb)
            PP_STAMP();
            PP_LGKM0();
            PP_STAMP();
            PP_BARRIER();
            PP_STAMP();
            // Ma': block b0 (K=64) + first half of b1's K + LDS-DMA of A1(t+1)
            PP_PIN_IN();
            __builtin_amdgcn_s_setprio(1);
            issue_a(1, t + 1);
            // issue order keeps every accumulator at least two MFMAs away from its previous use
    #pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                PP_MFMA(0, 0, 0, ks);
                PP_MFMA(0, 0, 1, ks);
                PP_MFMA(0, 1, 0, ks);
                PP_MFMA(0, 1, 1, ks);
            }
    #pragma unroll
            for (int ks = 2; ks < 4; ++ks) {
                PP_MFMA(0, 0, 0, ks);
                PP_MFMA(0, 0, 1, ks);
            }
            SGB_MFMA(4);
            SGB_VMEM(1);
            SGB_MFMA(4);
            SGB_VMEM(1);
            SGB_MFMA(4);
            __builtin_amdgcn_s_setprio(0);
            PP_PIN_OUT(0);
            PP_STAMP();
            PP_BARRIER();
            // Lb': fragments of b2 (first 32 rows of this group's A1)
            PP_STAMP();
            read_a2(cb);
            if (CONV && t + 2 < nk) step(1);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            PP_STAMP();
            PP_LGKM0();
            PP_STAMP();
            PP_BARRIER();
            PP_STAMP();
            // Mb': second half of b1's K + block b2 + LDS-DMA of A0,B0,B1(t+2)
            PP_PIN_IN();
            __builtin_amdgcn_s_setprio(1);
            issue_a(0, t + 2);
            issue_b(0, t + 2);
            issue_b(1, t + 2);
    #pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                PP_MFMA(0, 1, 0, 2 + ks);
                PP_MFMA2(0, ks);
                PP_MFMA(0, 1, 1, 2 + ks);
                PP_MFMA2(1, ks);
            }
    #pragma unroll
            for (int ks = 2; ks < 4; ++ks) {
                PP_MFMA2(0, ks);
                PP_MFMA2(1, ks);
            }
            SGB_MFMA(2);
            SGB_VMEM(1);
            SGB_MFMA(2);
            SGB_VMEM(1);
            SGB_MFMA(2);
            SGB_VMEM(1);
            SGB_MFMA(2);
            SGB_VMEM(1);
            SGB_MFMA(1);
            SGB_VMEM(1);
            SGB_MFMA(1);
            SGB_VMEM(1);
            SGB_MFMA(2);
            __builtin_amdgcn_s_setprio(0);
            PP_PIN_OUT_B();
            PP_STAMP();
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            PP_STAMP();
            PP_BARRIER();
        }
    }
    if (wr == 0) PP_BARRIER();          // rebalance the barrier count
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: lane owns rows (l31 per row slot) x 4-column groups (gemm_epilogue.h) ----
    f32x4 bias4[2][4];
    f32x4 gate4[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int col = n0 + wc * 64 + qb * 32 + 8 * gq + 4 * hi;
            bias4[qb][gq] = (p.bias && col < p.N) ? *(const f32x4*)(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            gate4[qb][gq] = (EPI == EPI_RESID_GATE_F32 && p.gate_table && col < p.N) ? *(const f32x4*)(p.gate_table + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            asm volatile("" : "+v"(bias4[qb][gq]));      // retire the loads once, here
            if (EPI == EPI_RESID_GATE_F32) asm volatile("" : "+v"(gate4[qb][gq]));
        }
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (ASYM && wr == 1 && qa == 1 && i == 1) continue;      // rows [224,256) belong to the next tile
            const int row = m0 + wr * 128 + qa * 64 + i * 32 + l31;
            if (row >= p.M) continue;
            const EpiRow er = epi_row_setup<EPI>(p, row);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int col = n0 + wc * 64 + qb * 32 + 8 * gq + 4 * hi;
                    if (col >= p.N) continue;
                    const f32x4 v = {acc[qa][i][qb][4 * gq], acc[qa][i][qb][4 * gq + 1], acc[qa][i][qb][4 * gq + 2],
                                     acc[qa][i][qb][4 * gq + 3]};
                    epi_store4<EPI>(p, er, row, col, v, bias4[qb][gq], gate4[qb][gq]);
                }
        }
}

template <int EPI, bool CONV, int BM>
int launch_pp(const GemmParams& p, hipStream_t stream) {
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, CONV, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    const int Mt = (p.M + BM - 1) / BM, Nt = (p.N + TBN - 1) / TBN;
    hipLaunchKernelGGL((gemm_pp_kernel<EPI, CONV, BM>), dim3(Mt * Nt), dim3(512), LDS_BYTES, stream, p);
    LTX2_CHECK_LAUNCH("gemm_pp_kernel");
    return LTX2_OK;
}

// 224-row tiles when they need fewer CU-rounds of work than 256-row tiles (dense GEMMs only).
bool prefer_224(const GemmParams& p) {
    const long nt = (p.N + TBN - 1) / TBN, cus = 256;
    const long t256 = ((long)(p.M + 255) / 256) * nt, t224 = ((long)(p.M + 223) / 224) * nt;
    const long cost256 = (t256 + cus - 1) / cus * 256, cost224 = (t224 + cus - 1) / cus * 224;
    return cost224 < cost256;
}

}  // namespace

int gemm_pp_launch(const GemmParams& p_in, int epilogue, bool conv, hipStream_t stream) {
    GemmParams p = p_in;
    const bool b224 = !conv && prefer_224(p);
#define CASE(E)                                                                            \
    case E:                                                                                \
        return conv ? launch_pp<E, true, 256>(p, stream)                                   \
                    : (b224 ? launch_pp<E, false, 224>(p, stream) : launch_pp<E, false, 256>(p, stream));
    switch (epilogue) {
        CASE(EPI_BF16)
        CASE(EPI_GELU_BF16)
        CASE(EPI_SILU_BF16)
        CASE(EPI_F32)
        CASE(EPI_RESID_GATE_F32)
        case EPI_ADD_BF16:
            return conv ? launch_pp<EPI_ADD_BF16, true, 256>(p, stream) : launch_pp<EPI_ADD_BF16, false, 256>(p, stream);
        case EPI_D2S_BF16:
            return launch_pp<EPI_D2S_BF16, true, 256>(p, stream);
        default:
            ltx2_set_error("gemm: unknown epilogue %d", epilogue);
            return LTX2_E_INVALID;
    }
#undef CASE
}
