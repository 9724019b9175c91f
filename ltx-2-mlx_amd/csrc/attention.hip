// Flash-attention forward (non-causal, optional key mask, head_dim 128 or 64) for gfx950 / CDNA4.
//
// Replaces mx.fast.scaled_dot_product_attention as called from the reference's
// _compiled_attention_core_no_mask (LTX_2_MLX/model/transformer/attention.py:12-34):
// out[q, h*128:(h+1)*128] = softmax(Q_h K_h^T / sqrt(128)) V_h, tokens-major (B=1, [T, H*d]).
//
// Design (round 5; rounds 1-4 ran the same structure on 32x32x16 blocks with an online maximum -- DESIGN.md section 4 has the measurements
// that retired it, its stream-K launch form and the 64-rows-per-wave experiment):
//  * block = 4 wave64 = 128 query rows of one head, two blocks per CU; KV tile = 64 keys; K tile [64][HD] and V^T tile [HD][64]
//    double-buffered in LDS (64 KiB), staged with buffer_load ... lds (16 B/lane), bank swizzle applied on the source address and again
//    on the ds_read_b128.
//  * v_mfma_f32_16x16x32 blocks (the socket runs attention at its power cap; this shape moves a quarter of the accumulator traffic per flop
//    of 32x32x16: tools/micro/mfma_power.hip measures 2.06-2.10 against 1.75-1.80 PF/s at the cap), "swapped" products so that every
//    softmax statistic is lane-local.  Per wave and tile:
//      S^T[kb][qb] (16 keys x 16 queries, kb < 4, qb < 2) += K[kb][ks] (A: 16 keys x 32 dims) . Q[qb][ks]^T (B)      4 * HD/32 fragment reads, 2 MFMAs each
//      O^T[db][qb] (16 dims x 16 queries, db < HD/16)     += V^T[db][kk] (A: 16 dims x 32 keys) . P[qb][kk]^T (B)    2 * HD/16 fragment reads, 2 MFMAs each
//    Lane (c = lane & 15, g = lane >> 4) owns query column 16 qb + c of both query blocks; of a 16-key block it holds keys 4 g + r (r < 4),
//    so a row's statistics live in the four lanes {c, c+16, c+32, c+48}: one v_permlane32_swap + one v_permlane16_swap fold BOTH query blocks.
//  * P never leaves registers: the B fragment of key half kk is [S[2kk][qb][0..3], S[2kk+1][qb][0..3]] = keys {32kk + 4g + r, 32kk + 16 + 4g + r}
//    at MFMA k-slots 8g + e, so V^T keeps, inside every 32-key block, key 16h + 4g + r at position 8g + 4h + r (vt_transpose_kernel and
//    gemm_v4's fused V^T epilogue write that order): the V^T fragment is one conflict-free 16-byte LDS read.
//  * softmax with a STALE row maximum (below): Q pre-scaled, the MFMA result is the exponent's argument, no per-tile maximum on the common path.
#include "attention.h"
#include <type_traits>

namespace {

constexpr int KVB = 64;
// HD = 128 (video streams) or 64 (audio streams and audio<->video cross-modal attention):
// K tile [64][HD] has rows of 2*HD bytes, V^T tile [HD][64] rows of 128 bytes.
template <int HD>
struct Geo {
    static constexpr int K_TILE = KVB * HD * 2;
    static constexpr int V_TILE = HD * KVB * 2;
    static constexpr int STAGE = K_TILE + V_TILE;
    static constexpr int LDS_BYTES = 2 * STAGE;
    static constexpr int NJ = HD / 32;          // LDS-DMA instructions per wave per tile (K and V^T each)
};

// LDS fragment reads kept in flight ahead of the MFMA that consumes them (AT_DK for K, AT_DV for V^T).
#ifndef AT_DK
#define AT_DK 6
#endif
#ifndef AT_DV
#define AT_DV 6
#endif
#ifndef AT_XCD
#define AT_XCD 1        // whole heads dealt to the XCDs (same-box A/B: self-attention N = 3456 187.1 -> 182.2 us, with the priorities above 194.9 -> 179-181; 0 = the plain (q-tile, head) grid order)
#endif
#ifndef AT_DMA
#define AT_DMA 0        // where the next tile's 2 NJ LDS-DMA pieces issue: 0 = between the QK^T MFMAs, 1 = all behind them (in the softmax's VALU stretch), 2 = K in, V^T behind
#endif
#ifndef AT_PQK
#define AT_PQK 2        // s_setprio level inside the QK^T cluster / the P.V cluster / between them (softmax tail, barrier).  Round 5, same-box A/B
                        // (tools/attn_time.py, self-attention N = 3456): no priorities 212 us, 1 / 1 / 0 (rounds 3-4) 187, 1 / 0 / 0 182.5, 2 / 1 / 0 179-181, 3 / 3 / 0 186:
                        // the wave inside QK^T (which carries the exponentials of the previous key block) must win over a partner inside P.V
#endif
#ifndef AT_PPV
#define AT_PPV 1
#endif
#ifndef AT_PGAP
#define AT_PGAP 0
#endif
#ifndef AT_PRIO
#define AT_PRIO 1       // s_setprio 1 over the two MFMA clusters of a tile (round 3, same-box A/B: self-attention -0.7 %, text cross-attention -2 %:
                        // the wave inside an MFMA cluster wins the issue slot, its SIMD partner's softmax VALU fills the gaps)
#endif

// v_max3_f32 directly: fmaxf() lowers to llvm.maxnum, which first canonicalises every MFMA output (one extra v_max x,x per score).  The hazard
// recogniser does not see an asm's operands and an MFMA result has no hardware interlock against a VALU read, so the max3 chain starts with
// mfma16_result_guard(): the software wait states the ISA requires between an XDL write and a VALU read.
__device__ __forceinline__ float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// score of a masked key: far below any real score and small enough that (score - max) stays finite
// (the reference adds -3.4e38 to the score, attention.py:38-70: the same softmax, also for a row whose keys are all masked -> uniform)
#define AT_KEY_MASKED (-1.0e30f)

__device__ __forceinline__ void quad_fold_max(float& a, float& b) {
    // in: a = value of query block 0, b = of query block 1 (per lane).  out: a = max over the row's four lanes for block 0, b = for block 1, in EVERY lane.
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));        // a = [a.lo | b.lo], b = [a.hi | b.hi]
    float c = fmaxf(a, b), d = c;                                                                 // lanes 0-31: block 0 (g, g+2 folded), lanes 32-63: block 1
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));        // c = rows [c0 c0 c2 c2], d = rows [c1 c1 c3 c3]
    float e = fmaxf(c, d);                                                                        // rows [m0 m0 m1 m1]
    a = e;
    b = e;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));        // a = m0 everywhere, b = m1 everywhere
}
__device__ __forceinline__ void quad_fold_sum(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float c = a + b, d = c;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
    float e = c + d;
    a = e;
    b = e;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void mfma16_result_guard(f32x4 (&s)[4][2], float& t0, float& t1) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]), "+v"(s[2][0]), "+v"(s[2][1]), "+v"(s[3][0]), "+v"(s[3][1]),
                 "+v"(t0), "+v"(t1));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The softmax: a STALE row maximum.  The online softmax spends ~35 of its ~145 VALU instructions per tile and lane on
// the running maximum (16 v_max3, the cross-lane fold, the rescale decision) and 32 more on the exponent's fma -- and VALU issue, not the MFMA pipe, is
// what bounds this kernel.  Neither is needed on the common path:
//   * Q is scaled by scale * log2(e) (times the row's RMS factor in the QS form) ONCE, when its fragments are loaded, and the score accumulators start
//     at -M (M = the row's reference exponent, in exp2 units) instead of 0: the MFMA result IS the exponent's argument, P = exp2(S) with no fma.
//   * M is the maximum of the row's FIRST tile and stays there: any reference works mathematically (softmax is shift-invariant, O and l are fp32, P keeps
//     its relative precision at any magnitude) as long as nothing overflows.  That is checked, not assumed: a tile whose exponentials sum past AT_P_BIG
//     in any lane (one compare per query block) takes the classic path for that tile -- true tile maximum, O and l rescaled, M moved up, P recomputed
//     from the still-intact scores.  M never exceeds the row's true running maximum, so nothing underflows that the classic form would keep.
//   * with no maximum to wait for, the exponentials of key block kb - 1 issue between this wave's OWN QK^T MFMAs of key block kb (order: key block ->
//     k-step -> query block), one pair per fragment step; only the last block's are exposed.
// The first tile, the ragged last tile and every tile of the key-mask form take the classic path (the key-mask form keeps absolute scores: a row whose
// keys so far are all masked has M = -1e30, which a relative accumulator start would swallow the real scores in).
#ifndef AT_P_BIG
#ifdef LTX2_F16
#define AT_P_BIG 32768.0f           // P is IEEE half in this build (max 65504): every P <= the lane's 16-key tile sum <= 2^15 (round 5: 2^12; round 6 measured
                                    // how often real-shaped scores cross either bound: profiles/r06_attn_data_dependence.md)
#else
#define AT_P_BIG 1073741824.0f      // 2^30: l and O have 2^90 of fp32 headroom left over 3456 keys
#endif
#endif

#ifdef AT_COUNT_FALLBACK
// measurement build only (tools/ab_build.py ... attention.hip=-DAT_COUNT_FALLBACK; never shipped): [0] = wave-tiles that entered the stale-maximum path,
// [1] = those that fell back to the classic path because a lane's exponential sum crossed AT_P_BIG
__device__ unsigned long long ltx2_at_counts[2];
#endif

// NW = waves per workgroup: 4 (128 query rows, two workgroups per CU, each with its own K / V^T stage) or 8 (256 query rows, ONE workgroup per CU: the
// head's tiles are fetched and filled into LDS once per 256 rows instead of once per 128 -- half the LDS-DMA issues per wave, half the fills per unit of work)
template <int HD, bool QS = false, bool KM = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_fwd_kernel(const AttnParams p) {
    using G = Geo<HD>;
    constexpr int K_TILE = G::K_TILE, STAGE = G::STAGE, NJ = G::NJ * 4 / NW, QB = NW * 32;
    static_assert(NW == 4 || (NW == 8 && HD == 128), "waves per workgroup");
    constexpr int NKS = HD / 32, NDB = HD / 16;
    constexpr int NK = 4 * NKS, NV = 2 * NDB;
    constexpr int DK = AT_DK < NK ? AT_DK : NK, DV = AT_DV < NV ? AT_DV : NV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int nt = (p.Nkv + KVB - 1) / KVB, nfull = p.Nkv / KVB;

    unsigned k_vo[NJ], v_vo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int kr = HD == 128 ? (wv * NJ + j) * 4 + (lane >> 4) : (wv * NJ + j) * 8 + (lane >> 3);
        const int kchunk = HD == 128 ? ((lane & 15) ^ (kr & 15)) : ((lane & 7) ^ ((kr >> 1) & 7));
        k_vo[j] = ((unsigned)kr * (unsigned)p.ldk + kchunk * 8) * 2;
        const int vr = (wv * NJ + j) * 8 + (lane >> 3);
        const int vchunk = (lane & 7) ^ ((vr >> 1) & 7);
        v_vo[j] = ((unsigned)vr * (unsigned)p.Npad + vchunk * 8) * 2;
    }
    const unsigned k_tile_bytes = (unsigned)KVB * (unsigned)p.ldk * 2;
    const unsigned k_bytes = ((unsigned)(p.Nkv - 1) * (unsigned)p.ldk + HD) * 2, v_bytes = (unsigned)HD * (unsigned)p.Npad * 2;
    const int k_xor = HD == 128 ? l15 : ((l15 >> 1) & 7);
    const int v_xor = (l15 >> 1) & 7;
    unsigned k_lane[NKS], v_lane[2];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) k_lane[ks] = lds0 + l15 * (2 * HD) + (((4 * ks + g) ^ k_xor) << 4);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) v_lane[kk] = lds0 + K_TILE + l15 * 128 + (((4 * kk + g) ^ v_xor) << 4);

    int head = blockIdx.y, qt = blockIdx.x;
#if AT_XCD
    // workgroup b runs on XCD b % 8 (observed placement): deal whole heads to the XCDs, so a head's K / V^T tiles are fetched by ONE L2 instead of all eight
    if ((gridDim.y & 7) == 0) {
        const int b = blockIdx.y * gridDim.x + blockIdx.x, xcd = b & 7, idx = b >> 3;
        head = xcd + 8 * (idx / (int)gridDim.x);
        qt = idx % (int)gridDim.x;
    }
#endif
    const int tb = nt;
    const int q0 = qt * QB + wv * 32;

    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(p.K + head * HD), 0, (int)k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(p.VT + (long)head * p.vt_head_stride), 0, (int)v_bytes, 0x00020000);
    auto stage_piece = [&](int t, int buf, int i) {
        char* dst = smem + buf * STAGE + wv * (NJ * 1024);
#if defined(__HIP_DEVICE_COMPILE__)
        if (i < NJ)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr_t)(dst + i * 1024), 16, k_vo[i], t * k_tile_bytes, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(dst + K_TILE + (i - NJ) * 1024), 16, v_vo[i - NJ], t * (KVB * 2), 0, 0);
#else
        (void)dst; (void)t; (void)i;
#endif
    };
    // tile 0 is on its way while the Q fragments are loaded and scaled
#pragma unroll
    for (int i = 0; i < 2 * NJ; ++i) stage_piece(0, 0, i);

    // ---- softmax scale of this lane's two query rows (exp2 domain), folded into Q ----
    float c[2] = {p.scale_log2e, p.scale_log2e};
    if constexpr (QS) {
        const int quarter = p.q_ss_ld >> 2;
        float ss[2] = {0.f, 0.f};
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float* sp = p.q_ss + (long)min(q0 + 16 * qb + l15, p.Nq - 1) * p.q_ss_ld + g * quarter;
#pragma unroll 4
            for (int jj = 0; jj < quarter; jj += 4) {
                const f32x4 t = *(const f32x4*)(sp + jj);
                ss[qb] += (t[0] + t[1]) + (t[2] + t[3]);
            }
        }
        quad_fold_sum(ss[0], ss[1]);
        c[0] *= rsqrtf(ss[0] / (float)p.q_norm_dim + p.q_eps);
        c[1] *= rsqrtf(ss[1] / (float)p.q_norm_dim + p.q_eps);
    }
    bf16x8 qf[2][NKS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = min(q0 + 16 * qb + l15, p.Nq - 1);
        const bf16* qp = p.Q + (long)qrow * p.ldq + head * HD + 8 * g;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 raw = *(const bf16x8*)(qp + 32 * ks);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[qb][ks][e] = f2bf(bf2f(raw[e]) * c[qb]);
        }
    }

    f32x4 o[NDB][2];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) o[d][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float M[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};     // M: reference exponent of the row (exp2 units); set by the first tile
    // one KV tile.  FAST: stale-maximum path with the exponentials inside the QK^T cluster; otherwise the classic path (FIRST: runtime flag)
    auto tile = [&](const int t, auto fast, auto masked, auto par, const bool first) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast)::value, MASKED = decltype(masked)::value;
        constexpr int PAR = decltype(par)::value;
        static_assert(!(FAST && (MASKED || KM)), "the stale-maximum path takes whole, unmasked tiles");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tn = min(t + 1, tb - 1);
        constexpr int nbuf = 1 - PAR;
#if AT_PRIO
        __builtin_amdgcn_s_setprio(AT_PQK);
#endif
        // ---- S^T = K . (c Q)^T - M ; fragment n = NKS kb + ks feeds the two query blocks ----
        f32x4 s[4][2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float init = KM ? 0.f : -M[qb];
                s[kb][qb] = f32x4{init, init, init, init};
            }
        bf16x8 pf[2][2];                // [qb][kk]
        float ps[2] = {0.f, 0.f};       // this tile's exponential sums per query block
        // exponential pair Q4 (< 4) of key block KB: query block Q4 / 2, rows 2 (Q4 % 2), +1 -> elements 4 (KB % 2) + row of P[qb][KB / 2]
        auto exp_pair = [&](auto KB, auto Q4) __attribute__((always_inline)) {
            constexpr int kb = decltype(KB)::value, q4 = decltype(Q4)::value, qb = q4 / 2, r = 2 * (q4 % 2), kk = kb / 2, e = 4 * (kb % 2) + r;
            asm volatile("" : "+v"(s[kb][qb]));
            float p0 = __builtin_amdgcn_exp2f(s[kb][qb][r]), p1 = __builtin_amdgcn_exp2f(s[kb][qb][r + 1]);
            asm volatile("" : "+v"(p0), "+v"(p1));
            ps[qb] += p0 + p1;
            pf[qb][kk][e] = f2bf(p0);
            pf[qb][kk][e + 1] = f2bf(p1);
        };
        u32x4 kf[NK];
        auto read_k = [&](auto N) {
            constexpr int n = decltype(N)::value, kb = n / NKS, ks = n % NKS;
            kf[n] = lds_read16<PAR * STAGE + kb * 16 * 2 * HD>(k_lane[ks]);
        };
        static_for<0, DK>(read_k);
        static_for<0, NK>([&](auto N) __attribute__((always_inline)) {
            constexpr int n = decltype(N)::value, kb = n / NKS, ks = n % NKS;
            if constexpr (n + DK < NK) read_k(std::integral_constant<int, n + DK>{});
            lds_wait<(n + DK < NK ? DK : NK - 1 - n)>(kf[n]);
            s[kb][0] = LTX2_MFMA_16x16x32(as_bf16x8(kf[n]), qf[0][ks], s[kb][0], 0, 0, 0);
            if constexpr (FAST && kb > 0) static_for<(4 * ks) / NKS, (4 * ks + 2) / NKS>([&](auto Q4) { exp_pair(std::integral_constant<int, kb - 1>{}, Q4); });
            s[kb][1] = LTX2_MFMA_16x16x32(as_bf16x8(kf[n]), qf[1][ks], s[kb][1], 0, 0, 0);
            if constexpr (FAST && kb > 0) static_for<(4 * ks + 2) / NKS, (4 * (ks + 1)) / NKS>([&](auto Q4) { exp_pair(std::integral_constant<int, kb - 1>{}, Q4); });
#if AT_DMA == 0
            if constexpr ((n & 1) && n / 2 < 2 * NJ) stage_piece(tn, nbuf, n / 2);
#elif AT_DMA == 2
            if constexpr ((n & 1) && n / 2 < NJ) stage_piece(tn, nbuf, n / 2);        // the K pieces here, the V^T pieces in the VALU gap below
#endif
        });
#if AT_DMA == 0
        static_for<NK / 2, 2 * NJ>([&](auto I) { stage_piece(tn, nbuf, decltype(I)::value); });
#elif AT_DMA == 1
        static_for<0, 2 * NJ>([&](auto I) { stage_piece(tn, nbuf, decltype(I)::value); });
#else
        static_for<(NK / 2 < NJ ? NK / 2 : NJ), 2 * NJ>([&](auto I) { stage_piece(tn, nbuf, decltype(I)::value); });
#endif
        // the first V^T fragments go out now: the remaining softmax work hides their latency
        u32x4 vf[NV];
        auto read_v = [&](auto N) {
            constexpr int n = decltype(N)::value, kk = n / NDB, db = n % NDB;
            vf[n] = lds_read16<PAR * STAGE + db * 16 * 128>(v_lane[kk]);
        };
        static_for<0, DV>(read_v);
#if AT_PRIO
        __builtin_amdgcn_s_setprio(AT_PGAP);
#endif
        // O^T += V^T . P^T over the fragments [N0, N1) (fragment n = NDB kk + db feeds the two query blocks)
        auto pv = [&](auto N0, auto N1) __attribute__((always_inline)) {
            constexpr int n0 = decltype(N0)::value, n1 = decltype(N1)::value;
#if AT_PRIO
            __builtin_amdgcn_s_setprio(AT_PPV);
#endif
            static_for<n0, n1>([&](auto N) __attribute__((always_inline)) {
                constexpr int n = decltype(N)::value, kk = n / NDB, db = n % NDB;
                if constexpr (n + DV < NV) read_v(std::integral_constant<int, n + DV>{});
                lds_wait<(n + DV < NV ? DV : NV - 1 - n)>(vf[n]);
                o[db][0] = LTX2_MFMA_16x16x32(as_bf16x8(vf[n]), pf[0][kk], o[db][0], 0, 0, 0);
                o[db][1] = LTX2_MFMA_16x16x32(as_bf16x8(vf[n]), pf[1][kk], o[db][1], 0, 0, 0);
            });
#if AT_PRIO
            __builtin_amdgcn_s_setprio(AT_PGAP);
#endif
        };
        using I0 = std::integral_constant<int, 0>;
        using IN = std::integral_constant<int, NV>;
        bool classic = !FAST;
        if constexpr (FAST) {
            static_for<0, 4>([&](auto Q4) { exp_pair(std::integral_constant<int, 3>{}, Q4); });
            classic = __any(!(ps[0] <= AT_P_BIG && ps[1] <= AT_P_BIG));          // (also catches a NaN sum)
#ifdef AT_COUNT_FALLBACK
            if (lane == 0) {
                atomicAdd(&ltx2_at_counts[0], 1ull);
                if (classic) atomicAdd(&ltx2_at_counts[1], 1ull);
            }
#endif
        }
        if (classic) {
            if constexpr (MASKED) {
                const int kv0 = t * KVB;
                unsigned long long km = ~0ull;
                if constexpr (KM) km = p.kmask[t];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kl = 16 * kb + 4 * g + r;
                        const bool oob = kv0 + kl >= p.Nkv, off = KM && !((km >> kl) & 1ull);
#pragma unroll
                        for (int qb = 0; qb < 2; ++qb) {
                            if (oob) s[kb][qb][r] = -INFINITY;
                            else if (off) s[kb][qb][r] = AT_KEY_MASKED;
                        }
                    }
            }
            float tmax[2] = {-INFINITY, -INFINITY};
            mfma16_result_guard(s, tmax[0], tmax[1]);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    tmax[qb] = max3(tmax[qb], s[0][qb][r], s[1][qb][r]);
                    tmax[qb] = max3(tmax[qb], s[2][qb][r], s[3][qb][r]);
                }
            quad_fold_max(tmax[0], tmax[1]);
            float sub[2];       // what the exponent subtracts from the accumulator
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if constexpr (KM) {         // absolute scores: tmax is the tile's maximum itself
                    const float m_new = first ? tmax[qb] : fmaxf(M[qb], tmax[qb]);
                    if (!first) {
                        const float alpha = __builtin_amdgcn_exp2f(M[qb] - m_new);
                        l_run[qb] *= alpha;
#pragma unroll
                        for (int d = 0; d < NDB; ++d) o[d][qb] *= alpha;
                    }
                    M[qb] = m_new;
                    sub[qb] = m_new;
                } else {                    // scores relative to M: tmax is how far the tile's maximum lies above the reference
                    const float up = first ? tmax[qb] : fmaxf(tmax[qb], 0.f);
                    if (!first) {
                        const float alpha = __builtin_amdgcn_exp2f(-up);
                        l_run[qb] *= alpha;
#pragma unroll
                        for (int d = 0; d < NDB; ++d) o[d][qb] *= alpha;
                    }
                    M[qb] += up;
                    sub[qb] = up;
                }
            }
            ps[0] = 0.f;
            ps[1] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const float p0 = __builtin_amdgcn_exp2f(s[kb][qb][r] - sub[qb]), p1 = __builtin_amdgcn_exp2f(s[kb][qb][r + 1] - sub[qb]);
                        ps[qb] += p0 + p1;
                        pf[qb][kb / 2][4 * (kb % 2) + r] = f2bf(p0);
                        pf[qb][kb / 2][4 * (kb % 2) + r + 1] = f2bf(p1);
                    }
        }
        l_run[0] += ps[0];
        l_run[1] += ps[1];
        pv(I0{}, IN{});
    };

    using T_ = std::true_type;
    using F_ = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    if constexpr (KM) {             // every tile masked, classic
        for (int t = 0; t < tb; ++t) {
            if (t & 1) tile(t, F_{}, T_{}, P1{}, false);
            else tile(t, F_{}, T_{}, P0{}, t == 0);
        }
    } else {
        const int t_fast_end = min(tb, nfull);
        int t = 0;
        if (nfull > 0) {
            tile(0, F_{}, F_{}, P0{}, true);
            t = 1;
            for (; t + 1 < t_fast_end; t += 2) {
                tile(t, T_{}, F_{}, P1{}, false);
                tile(t + 1, T_{}, F_{}, P0{}, false);
            }
            if (t < t_fast_end) {
                tile(t, T_{}, F_{}, P1{}, false);
                ++t;
            }
        }
        if (nfull < tb) {           // the ragged last tile (the first one too when Nkv < 64)
            if (nfull & 1) tile(nfull, F_{}, T_{}, P1{}, false);
            else tile(nfull, F_{}, T_{}, P0{}, nfull == 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    float lt[2] = {l_run[0], l_run[1]};
    quad_fold_sum(lt[0], lt[1]);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = q0 + 16 * qb + l15;
        float inv = 1.0f / lt[qb];
        if (p.gate && qrow < p.Nq) {
            float gl = p.gate[(long)qrow * p.gate_ld + head];
            if (p.gate_parts > 1) {         // partial sums over K slices + bias, in slice order
                for (int pt = 1; pt < p.gate_parts; ++pt) gl += p.gate[((long)pt * p.Nq + qrow) * p.gate_ld + head];
                gl += p.gate_bias[head];
            }
            inv *= 2.f / (1.f + __expf(-gl));
        }
        bf16* op = p.O + (long)min(qrow, p.Nq - 1) * p.ldo + head * HD + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
        for (int i = 0; i < NDB / 2; ++i) {
            bf16x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = f2bf(o[2 * i][qb][e] * inv);
                b[e] = f2bf(o[2 * i + 1][qb][e] * inv);
            }
            u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1" : "+v"(ua[0]), "+v"(ua[1]), "+v"(ub[0]), "+v"(ub[1]));
            const u32x4 w = {ua[0], ua[1], ub[0], ub[1]};
            if (qrow < p.Nq) *(u32x4*)(op + 32 * i) = w;
        }
    }
}

// V [Nkv][ld] (head h at columns h*HD) -> VT[h][HD][Npad] with the key permutation
// pos(kv = 32b + 16h + 4g + r) = 32b + 8g + 4h + r (h < 2, g < 4, r < 4); keys >= Nkv are zero-filled.
template <int HD>
__global__ __launch_bounds__(256) void vt_transpose_kernel(const bf16* __restrict__ V, long ld, bf16* __restrict__ VT,
                                                           int Nkv, int Npad, long head_stride) {
    __shared__ bf16 tile[64][HD + 2];
    const int head = blockIdx.y, kv0 = blockIdx.x * 64, tid = threadIdx.x;
    // load 64 keys x HD dims: thread -> (row = tid/4, HD/4 dims) as 16-B pieces
    {
        constexpr int W = HD / 4;
        const int r = tid >> 2, c0 = (tid & 3) * W;
        const int kv = kv0 + r;
        if (kv < Nkv) {
            const bf16* src = V + (long)kv * ld + head * HD + c0;
#pragma unroll
            for (int i = 0; i < W / 8; ++i) {
                const bf16x8 v = *(const bf16x8*)(src + i * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) tile[r][c0 + i * 8 + e] = v[e];
            }
        } else {
#pragma unroll
            for (int i = 0; i < W; ++i) tile[r][c0 + i] = f2bf(0.f);
        }
    }
    __syncthreads();
    // store: thread -> (d = tid/2, 32 permuted key slots) as 4 x 16-B
    if (tid < 2 * HD) {
        const int d = tid >> 1, p0 = (tid & 1) * 32;
        bf16* dst = VT + (long)head * head_stride + (long)d * Npad + kv0 + p0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int pos = i * 8 + e;                       // position within the 32-block
                const int kvl = 16 * ((pos >> 2) & 1) + 4 * (pos >> 3) + (pos & 3);      // position 8 g + 4 h + r holds key 16 h + 4 g + r
                v[e] = tile[p0 + kvl][d];
            }
            *(bf16x8*)(dst + i * 8) = v;
        }
    }
}

}  // namespace

int attn_launch(const AttnParams& p, hipStream_t stream) {
    LTX2_CHECK_ARG(p.Nq > 0 && p.Nkv > 0 && p.H > 0, "attention: empty problem");
    LTX2_CHECK_ARG(p.head_dim == 0 || p.head_dim == 128 || p.head_dim == 64, "attention: head_dim=%d, only 128 and 64 are implemented", p.head_dim);
    LTX2_CHECK_ARG(p.Npad % 64 == 0 && p.Npad >= p.Nkv, "attention: Npad=%d must be a multiple of 64 >= Nkv", p.Npad);
    LTX2_CHECK_ARG(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldo % 8 == 0, "attention: row strides must keep 16-byte alignment");
    LTX2_CHECK_ARG(p.ldk > 0 && ((long)p.Nkv + 2 * KVB) * p.ldk * 2 < (1L << 31), "attention: Nkv * ldk exceeds the 31-bit K byte offset");
    LTX2_CHECK_ARG((long)p.Npad * (p.head_dim == 64 ? 64 : 128) * 2 < (1L << 31), "attention: Npad * head_dim exceeds the 31-bit V^T byte offset");
    if (p.q_ss)
        LTX2_CHECK_ARG(p.head_dim != 64 && p.q_ss_ld > 0 && p.q_ss_ld % 16 == 0 && p.q_norm_dim > 0, "attention: the per-row scale form needs head_dim 128 and q_ss_ld %% 16 == 0");
#ifndef AT_NW
#define AT_NW 4         // waves per workgroup of the head_dim-128 forms without a key mask: 4 (128 query rows, two workgroups per CU) or 8 (256 rows, one per CU, ONE K / V^T
#endif                  // stage for all of them).  Round 6, same box: the 8-wave form is 2.7 % FASTER alone at N = 13 824 (2469-2477 -> 2393-2417 us) and 2-4 % slower at
                        // N = 3456 (1.75 rounds of 256 rows against 1.69 of 128) -- and inside the two-stage pipeline (tools/bench_two_stage.py, two alternations) picking it
                        // for the N = 13 824 steps made the 8 + 3 steps SLOWER: 1.796 / 1.818 s against 1.777 / 1.778.  Kept as an A/B build only (-DAT_NW=8).
    const int nw = (p.head_dim != 64 && !p.kmask) ? AT_NW : 4;
    dim3 grid((p.Nq + nw * 32 - 1) / (nw * 32), p.H);
#define AT_LAUNCH_NW(HDV, QSV, KMV, NWV)                                                                                                        \
    do {                                                                                                                                        \
        static PerDeviceOnce once_;                                                                                                             \
        if (once_.first())                                                                                                                      \
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<HDV, QSV, KMV, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<HDV>::LDS_BYTES); \
        hipLaunchKernelGGL((attn_fwd_kernel<HDV, QSV, KMV, NWV>), grid, dim3(NWV * 64), Geo<HDV>::LDS_BYTES, stream, p);                         \
    } while (0)
#define AT_LAUNCH(HDV, QSV, KMV) AT_LAUNCH_NW(HDV, QSV, KMV, 4)
    if (p.head_dim == 64) {
        if (p.kmask) AT_LAUNCH(64, false, true);
        else AT_LAUNCH(64, false, false);
    } else if (p.kmask) {
        if (p.q_ss) AT_LAUNCH(128, true, true);
        else AT_LAUNCH(128, false, true);
    } else if (p.q_ss) {
        if (nw == 8) AT_LAUNCH_NW(128, true, false, 8);
        else AT_LAUNCH(128, true, false);
    } else if (nw == 8) AT_LAUNCH_NW(128, false, false, 8);
    else AT_LAUNCH(128, false, false);
#undef AT_LAUNCH
#undef AT_LAUNCH_NW
    LTX2_CHECK_LAUNCH("attn_fwd_kernel");
    return LTX2_OK;
}

#ifdef AT_COUNT_FALLBACK
extern "C" int ltx2_attn_fallback_counts(unsigned long long* out2, int reset) {
    if (hipMemcpyFromSymbol(out2, HIP_SYMBOL(ltx2_at_counts), 16) != hipSuccess) return LTX2_E_HIP;
    if (reset) {
        const unsigned long long z[2] = {0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ltx2_at_counts), z, 16) != hipSuccess) return LTX2_E_HIP;
    }
    return LTX2_OK;
}
#endif

int vt_transpose_launch(const bf16* V, long ld, bf16* VT, int Nkv, int Npad, int H, hipStream_t stream, int head_dim) {
    LTX2_CHECK_ARG(Npad % 64 == 0 && Npad >= Nkv && ld % 8 == 0, "vt_transpose: bad strides");
    LTX2_CHECK_ARG(head_dim == 128 || head_dim == 64, "vt_transpose: head_dim=%d, only 128 and 64 are implemented", head_dim);
    dim3 grid(Npad / 64, H);
    if (head_dim == 64)
        hipLaunchKernelGGL(vt_transpose_kernel<64>, grid, dim3(256), 0, stream, V, ld, VT, Nkv, Npad, 64L * Npad);
    else
        hipLaunchKernelGGL(vt_transpose_kernel<128>, grid, dim3(256), 0, stream, V, ld, VT, Nkv, Npad, 128L * Npad);
    LTX2_CHECK_LAUNCH("vt_transpose_kernel");
    return LTX2_OK;
}
