// Flash-attention forward (non-causal, no mask, head_dim 128 or 64) for gfx950 / CDNA4.
//
// Replaces mx.fast.scaled_dot_product_attention as called from the reference's
// _compiled_attention_core_no_mask (LTX_2_MLX/model/transformer/attention.py:12-34):
// out[q, h*128:(h+1)*128] = softmax(Q_h K_h^T / sqrt(128)) V_h, tokens-major (B=1, [T, H*d]).
//
// Design (v1):
//  * block = 4 wave64 = 128 query rows of one head; KV tile = 64 keys; K tile [64][128] and
//    V^T tile [128][64] double-buffered in LDS (64 KiB), staged with global_load_lds (16 B/lane)
//    with the bank swizzle applied on the source address and again on the ds_read_b128.
//  * "swapped" products so that every softmax statistic is lane-local:
//      S^T[kv][q] = mfma(A = K frag, B = Q frag)   -> lane owns query column q = lane&31
//      O^T[d][q]  = mfma(A = V^T frag, B = P frag) -> same owner; alpha/l need no broadcasts.
//    The two lane halves (lane>>5) own complementary key subsets; row max is one xor-32 shuffle.
//  * P never leaves registers: S^T's accumulator layout gives lane-half `hi`, k-step ks the keys
//    {16ks+4hi+e, 16ks+8+4hi+e}; instead of permuting P across lanes, V^T is stored with exactly
//    that key order inside every 32-key block (done once by vt_transpose_kernel), which makes the
//    V^T fragment a single conflict-free 16-byte LDS read.
//  * online softmax in the exp2 domain (scale * log2(e) folded into the scores), fp32 statistics.
//  * stream-K variant (SK = true, round 2): N = 3456 tokens x 32 heads is 864 (q-tile, head) units on 512 workgroup slots,
//    1.69 rounds -> the second round runs 69 % full.  The SK kernel launches one persistent workgroup per slot.  Phase A:
//    every workgroup computes whole units, one per round, all starting at KV tile 0 together (that lockstep is what lets the
//    workgroups of an XCD share K / V^T tiles through its L2; a first version that cut ALL the work into equal ranges lost
//    it and was 7-10 % slower per tile).  Phase B: the units left over (fewer than there are workgroups) are cut into EQUAL
//    contiguous ranges of (unit, KV-tile) items; a unit cut by range boundaries is computed in pieces that are merged with
//    the usual (m, l, O) rule.  Ranges are walked backwards, so the piece that must WAIT (the tail of a unit) is the last
//    thing a workgroup does and the pieces it waits for come from LOWER-numbered workgroups that published before their
//    own tails: no workgroup ever waits on a later-dispatched one, and the merge order is fixed (piece order), so results
//    stay bit-reproducible.  With H % 8 == 0 the heads are dealt to the XCDs (workgroup b runs on XCD b % 8): a head's
//    K / V^T is then fetched by ONE L2 instead of all eight.
#include "attention.h"
#include <type_traits>

namespace {

constexpr int QB = 128, KVB = 64;
constexpr float RESCALE_THR = 6.0f;     // P <= 2^6: exact in bf16's 8-bit exponent, fp32 accumulation has ample headroom
// HD = 128 (video streams) or 64 (audio streams and audio<->video cross-modal attention):
// K tile [64][HD] has rows of 2*HD bytes, V^T tile [HD][64] rows of 128 bytes.
template <int HD>
struct Geo {
    static constexpr int K_TILE = KVB * HD * 2;
    static constexpr int V_TILE = HD * KVB * 2;
    static constexpr int STAGE = K_TILE + V_TILE;
    static constexpr int LDS_BYTES = 2 * STAGE;
    static constexpr int NKS = HD / 16;         // k-steps of S^T = K Q^T
    static constexpr int ND = HD / 32;          // 32-row blocks of O^T
    static constexpr int NJ = HD / 32;          // LDS-DMA instructions per wave per tile (K and V^T each)
};

// LDS fragment reads kept in flight ahead of the MFMA that consumes them (AT_DK for K, AT_DV for V^T).
#ifndef AT_DK
#define AT_DK 6
#endif
#ifndef AT_DV
#define AT_DV 6
#endif
#ifndef AT_MAX2
#define AT_MAX2 0
#endif
#ifndef AT_DV16X
#define AT_DV16X 4
#endif
#ifndef AT_PVX16
#define AT_PVX16 2      // the 16x16x32 kernel's same-wave exp / MFMA interleave: 0 never, 1 always, 2 where the grid's last round leaves SIMDs with one wave
#endif
#ifndef AT_PRIO
#define AT_PRIO 1       // 1 = s_setprio 1 over the two MFMA clusters of a tile (round 3, same-box A/B: self-attention -0.7 %, text cross-attention -2 %:
                        // the wave inside an MFMA cluster wins the issue slot, its SIMD partner's softmax VALU fills the gaps); 2 = over the softmax (no gain)
#endif
#ifndef AT_SFMA
#define AT_SFMA 1       // scalar-slot fma / add for the exponent argument and the row sum (round 3, same-box A/B: -2.5 % vs the packed forms)
#endif


// v_max3_f32 directly: fmaxf() lowers to llvm.maxnum, which first canonicalises every MFMA output (one extra
// v_max x,x per score).  The hazard recogniser does not see an asm's operands and an MFMA result has no hardware
// interlock against a VALU read, so the max3 chain starts with mfma_result_guard(): the software wait states
// the ISA requires between an XDL write and a VALU read, tied to both accumulators and to the running max so
// neither the MFMAs nor the max3 chain can cross it.
// Exchange between the two 32-lane halves with v_permlane32_swap_b32: lo = hi = x on entry; afterwards lo holds
// x[lane & 31] and hi holds x[32 + (lane & 31)] in every lane.  Issued as asm: with this toolchain
// __builtin_amdgcn_permlane32_swap returns its first result in both elements.
__device__ __forceinline__ void half_pair(float x, float& lo, float& hi) {
    lo = x;
    hi = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
}
__device__ __forceinline__ void mfma_result_guard(f32x16& a, f32x16& b, float& tmax) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(tmax));
}
__device__ __forceinline__ float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// K / V^T fragments are read with lds_read16 / lds_wait (common.h): measured here, dropping the K reads alone
// saved as much time as dropping the 16 QK MFMAs while hipcc scheduled them behind lgkmcnt(0).
// partial (O, m, l) of a unit's head piece: per workgroup slot, per wave: ND*4 x (64 lanes x 16 B) of O, then 64 x {m, l}
template <int HD>
struct SkSlot {
    static constexpr int WAVE_BYTES = (HD / 32) * 4 * 1024 + 512;
    static constexpr int BYTES = 4 * WAVE_BYTES;
};

// score of a masked key: far below any real score and small enough that (score - max) * scale stays finite for any row scale
// (the reference adds -3.4e38 to the score, attention.py:38-70: the same softmax, also for a row whose keys are all masked -> uniform)
#define AT_KEY_MASKED (-1.0e30f)

// PVX: the exponentials of key chunk j + 1 issue between the PV MFMAs of chunk j (same-wave interleave).  It pays where a SIMD holds ONE wave of
// this kernel -- the partially filled last round of a plain grid (self-attention at N = 3456: 864 workgroups on 512 slots) -- and costs ~1 % where
// every SIMD has two (tools/micro/mfma_valu_overlap.hip: only a wave's OWN MFMAs cover its VALU work); the launcher picks per grid.
template <int HD, bool SK, bool QS = false, bool KM = false, bool PVX = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnParams p) {
    static_assert(!(SK && KM), "the key mask runs on the plain grid");
    static_assert(!KM || AT_SFMA, "the key mask's exponent form lives in the scalar-fma softmax");
    using G = Geo<HD>;
    constexpr int K_TILE = G::K_TILE, STAGE = G::STAGE, NKS = G::NKS, ND = G::ND, NJ = G::NJ;
    constexpr int NK = 2 * NKS, NV = 4 * ND;        // K / V^T fragment reads (= MFMAs) per wave per tile
    constexpr int DK = AT_DK < NK ? AT_DK : NK, DVW = PVX && HD == 128 ? AT_DV16X : AT_DV, DV = DVW < NV ? DVW : NV;     // (the interleaved form at head_dim 128 spills with 6 fragments in flight)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = (p.Nkv + KVB - 1) / KVB, nfull = p.Nkv / KVB;
    const int nqt = (p.Nq + QB - 1) / QB;

    // ---- staging: buffer loads straight to LDS; a piece's address is (per-lane byte offset, fixed for the launch) + (tile byte
    //      offset, scalar) against a descriptor of this head's K / V^T rows -- no vector ALU work per issue (global_load_lds wants a
    //      64-bit per-lane pointer: ~4 VALU per piece, 32 per KV tile beside 32 MFMAs).  Rows past Nkv are outside the K
    //      descriptor's range and read as zeros (the ragged last tile masks them anyway). ----
    unsigned k_vo[NJ], v_vo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        // one LDS-DMA instruction covers 1 KiB: 4 K rows of 256 B (HD 128) or 8 rows of 128 B (HD 64)
        const int kr = HD == 128 ? (wv * NJ + j) * 4 + (lane >> 4) : (wv * NJ + j) * 8 + (lane >> 3);   // 0..63
        const int kchunk = HD == 128 ? ((lane & 15) ^ (kr & 15)) : ((lane & 7) ^ ((kr >> 1) & 7));
        k_vo[j] = ((unsigned)kr * (unsigned)p.ldk + kchunk * 8) * 2;
        const int vr = (wv * NJ + j) * 8 + (lane >> 3);                // 0..HD-1
        const int vchunk = (lane & 7) ^ ((vr >> 1) & 7);
        v_vo[j] = ((unsigned)vr * (unsigned)p.Npad + vchunk * 8) * 2;
    }
    const unsigned k_tile_bytes = (unsigned)KVB * (unsigned)p.ldk * 2;
    const unsigned k_bytes = ((unsigned)(p.Nkv - 1) * (unsigned)p.ldk + HD) * 2, v_bytes = (unsigned)HD * (unsigned)p.Npad * 2;

    // per-lane fragment offsets inside a stage (the swizzle of the staging, applied again on the read)
    const int k_xor = HD == 128 ? (l31 & 15) : ((l31 >> 1) & 7);
    const int v_xor = (l31 >> 1) & 7;
    unsigned k_lane[NKS], v_lane[4];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) k_lane[ks] = lds0 + l31 * (2 * HD) + (((2 * ks + hi) ^ k_xor) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) v_lane[i] = lds0 + K_TILE + l31 * 128 + (((2 * i + hi) ^ v_xor) << 4);

    // ---- this workgroup's work (SK): unit u = head * nqt + q-tile.  Phase A: whole units, one per round, every workgroup
    //      of the group starting at KV tile 0 together (they share K / V^T tiles through the L2 exactly like the plain grid).
    //      Phase B: the units_g % wl units left over are cut into equal ranges of (unit, KV-tile) items, walked backwards. ----
    int j = 0, wl = 1, u_base = 0, slot_stride = 1, slot_off = 0;       // worker j of wl in its group; slot(k) = k * stride + off
    int full_rounds = 0, rem_base = 0, tot_b = 0, wlb = 1, it_lo = 0, it_hi = 0, ra = 0;
    if constexpr (SK) {
        const int w = blockIdx.x;
        int units_g;
        if (p.sk_xcd) {                     // heads dealt to the XCDs: group = w % 8 (the XCD workgroup w runs on)
            j = w >> 3;
            wl = gridDim.x >> 3;
            units_g = (p.H >> 3) * nqt;
            u_base = (w & 7) * units_g;
            slot_stride = 8;
            slot_off = w & 7;
        } else {
            j = w;
            wl = gridDim.x;
            units_g = p.H * nqt;
        }
        full_rounds = units_g / wl;
        rem_base = full_rounds * wl;
        tot_b = (units_g - rem_base) * nt;
        wlb = min(wl, max(1, tot_b / 4));   // no range shorter than 4 KV tiles
        if (j < wlb) {
            it_lo = (int)((long)tot_b * j / wlb);
            it_hi = (int)((long)tot_b * (j + 1) / wlb);
        }
    }
    auto range_lo = [&](int k) { return (int)((long)tot_b * k / wlb); };

    bool more = true;
    bool first_seg = true;
    while (more) {
        int head, qt, ta, tb, ub = 0;
        if constexpr (SK) {
            int u;
            if (ra < full_rounds) {
                u = ra * wl + j;
                ta = 0;
                tb = nt;
                ++ra;
            } else if (it_hi > it_lo) {
                ub = (it_hi - 1) / nt;
                ta = max(it_lo, ub * nt) - ub * nt;
                tb = it_hi - ub * nt;
                it_hi = ub * nt + ta;
                u = rem_base + ub;
            } else {
                break;
            }
            more = ra < full_rounds || it_hi > it_lo;
            const int ug = u_base + u;
            head = ug / nqt;
            qt = ug - head * nqt;
            if (!first_seg) __syncthreads();        // every wave is done reading the previous segment's last tile
            first_seg = false;
        } else {
            head = blockIdx.y;
            qt = blockIdx.x;
            ta = 0;
            tb = nt;
            more = false;
        }
        const int q0 = qt * QB + wv * 32;

        // ---- Q fragments (B operand): Q[q0 + l31][16*ks + 8*hi .. +8], kept in registers ----
        bf16x8 qf[NKS];
        {
            const int qrow = min(q0 + l31, p.Nq - 1);
            const bf16* qp = p.Q + (long)qrow * p.ldq + head * HD + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8*)(qp + 16 * ks);
        }
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(p.K + head * HD), 0, (int)k_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(p.VT + (long)head * p.vt_head_stride), 0, (int)v_bytes, 0x00020000);
        // piece i of tile t's staging into buffer `buf`: pieces [0, NJ) are K, [NJ, 2 NJ) are V^T. One LDS-DMA issue
        // costs the wave ~50 cycles (a burst of 8: ~100 each), so the pieces are spread between the QK MFMAs.
        auto stage_piece = [&](int t, int buf, int i) {
            char* dst = smem + buf * STAGE + wv * (NJ * 1024);
#if defined(__HIP_DEVICE_COMPILE__)          // (the host pass of hipcc does not know this builtin and silently drops the kernel's stub)
            if (i < NJ)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr_t)(dst + i * 1024), 16, k_vo[i], t * k_tile_bytes, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(dst + K_TILE + (i - NJ) * 1024), 16, v_vo[i - NJ], t * (KVB * 2), 0, 0);
#else
            (void)dst; (void)t; (void)i;
#endif
        };

        f32x16 o[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        // softmax scale of this lane's query row (exp2 domain): uniform, or with the row's RMS factor folded in (QS)
        float c = p.scale_log2e;
        if constexpr (QS) {
            // the two lanes of a row (hi = 0 / 1) add one half of its partial sums each; the loads of a half are independent
            const int half = p.q_ss_ld >> 1;
            const float* sp = p.q_ss + (long)min(q0 + l31, p.Nq - 1) * p.q_ss_ld + hi * half;
            float ss = 0.f;
#pragma unroll 8
            for (int jj = 0; jj < half; jj += 4) {
                const f32x4 t = *(const f32x4*)(sp + jj);
                ss += (t[0] + t[1]) + (t[2] + t[3]);
            }
            ss += __shfl_xor(ss, 32);
            c *= rsqrtf(ss / (float)p.q_norm_dim + p.q_eps);
        }

#pragma unroll
        for (int i = 0; i < 2 * NJ; ++i) stage_piece(ta, 0, i);

        // One KV tile. MASKED is a compile-time flag: only the ragged last tile carries the key-bound compares (left
        // in the common body, hipcc if-converts them into ~115 predicated VALU ops on EVERY tile -- SQ_INSTS_VALU
        // showed 236 non-MFMA VALU per tile against 32 MFMAs).
        // PAR: which of the two stage buffers holds tile t -- a template flag, so the buffer's byte offset rides in the ds_read
        // immediates instead of one v_add per fragment read (12 per tile)
        auto tile = [&](const int t, auto masked, auto par) __attribute__((always_inline)) {
            constexpr bool MASKED = decltype(masked)::value;
            constexpr int PAR = decltype(par)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int tn = min(t + 1, tb - 1);          // the tail re-stages the last tile into the idle buffer (branch-free body)
            constexpr int nbuf = 1 - PAR;

#if AT_PRIO == 1
            __builtin_amdgcn_s_setprio(1);
#endif
            // ---- S^T = K . Q^T  (two 32-key blocks); fragment i = b * NKS + ks ----
            f32x16 s[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[b][r] = 0.f;
            u32x4 kf[NK];
            // issue order n = 2 ks + b (fragment i = b * NKS + ks): consecutive MFMAs share the Q fragment and alternate between
            // the two independent accumulators -- one operand register changes per MFMA and no back-to-back dependent pair
            // (the socket runs at its power cap: operand traffic is time; the GEMM's K loop gained 0.8 % from the same rule)
            auto read_k = [&](auto N) {
                constexpr int n = decltype(N)::value, i = (n & 1) * NKS + (n >> 1);
                kf[i] = lds_read16<PAR * STAGE + (i / NKS) * 32 * 2 * HD>(k_lane[i % NKS]);
            };
            static_for<0, DK>(read_k);
            static_for<0, NK>([&](auto N) {
                constexpr int n = decltype(N)::value, i = (n & 1) * NKS + (n >> 1);
                if constexpr (n + DK < NK) read_k(std::integral_constant<int, n + DK>{});
                lds_wait<(n + DK < NK ? DK : NK - 1 - n)>(kf[i]);
                s[i / NKS] = LTX2_MFMA_32x32x16(as_bf16x8(kf[i]), qf[i % NKS], s[i / NKS], 0, 0, 0);
                if constexpr ((n & 1) && n / 2 < 2 * NJ) stage_piece(tn, nbuf, n / 2);
            });
            static_for<NK / 2, 2 * NJ>([&](auto I) { stage_piece(tn, nbuf, decltype(I)::value); });

            // ---- mask the ragged tail, running max (raw score domain; the softmax scale * log2(e) is
            //      folded into one fma in front of v_exp_f32: p = 2^(s*c - m*c)) ----
            if constexpr (MASKED) {
                const int kv0 = t * KVB;
                unsigned long long km = ~0ull;
                if constexpr (KM) km = p.kmask[t];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kl = b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (kv0 + kl >= p.Nkv) s[b][r] = -INFINITY;
                        else if (KM && !((km >> kl) & 1ull)) s[b][r] = AT_KEY_MASKED;
                    }
            }
            // first V^T fragments go out before the row max and the exponentials, which hide their LDS latency
            // (and part of the MFMA-result wait below);
            // fragment i = d * 4 + (2 b + k2)
            u32x4 vf[NV];
            // issue order n = ND j + d (fragment i = 4 d + j): the P fragment stays put over ND consecutive MFMAs, the accumulators rotate
            auto read_v = [&](auto N) {
                constexpr int n = decltype(N)::value, i = (n % ND) * 4 + n / ND;
                vf[i] = lds_read16<PAR * STAGE + (i / 4) * 32 * 128>(v_lane[i % 4]);
            };
            static_for<0, DV>(read_v);

            float tmax = -INFINITY;
            mfma_result_guard(s[0], s[1], tmax);
#if AT_PRIO == 1
            __builtin_amdgcn_s_setprio(0);
#elif AT_PRIO == 2
            __builtin_amdgcn_s_setprio(1);
#endif
#if AT_MAX2
            {   // two independent v_max3 chains in two asm blocks (the compiler puts a hazard s_nop behind every single-instruction asm)
                float ta = -INFINITY, tb2 = -INFINITY;
#define MX4(o) "v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %1, %1, %8, %9\n\t" \
               "v_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %1, %1, %12, %13\n\tv_max3_f32 %0, %0, %14, %15\n\tv_max3_f32 %1, %1, %16, %17"
                asm volatile(MX4(0) : "+v"(ta), "+v"(tb2) : "v"(s[0][0]), "v"(s[1][0]), "v"(s[0][1]), "v"(s[1][1]), "v"(s[0][2]), "v"(s[1][2]), "v"(s[0][3]), "v"(s[1][3]),
                             "v"(s[0][4]), "v"(s[1][4]), "v"(s[0][5]), "v"(s[1][5]), "v"(s[0][6]), "v"(s[1][6]), "v"(s[0][7]), "v"(s[1][7]));
                asm volatile(MX4(0) : "+v"(ta), "+v"(tb2) : "v"(s[0][8]), "v"(s[1][8]), "v"(s[0][9]), "v"(s[1][9]), "v"(s[0][10]), "v"(s[1][10]), "v"(s[0][11]), "v"(s[1][11]),
                             "v"(s[0][12]), "v"(s[1][12]), "v"(s[0][13]), "v"(s[1][13]), "v"(s[0][14]), "v"(s[1][14]), "v"(s[0][15]), "v"(s[1][15]));
#undef MX4
                asm volatile("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(ta), "v"(tb2));
            }
#else
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = max3(tmax, s[0][r], s[1][r]);
#endif
            {   // other lane half: VALU swap, no LDS-queue operation between the counted waits
                float t_lo, t_hi;
                half_pair(tmax, t_lo, t_hi);
                tmax = fmaxf(t_lo, t_hi);
            }
            // Deferred rescale: keep the old running max while the tile max exceeds it by at most
            // RESCALE_THR (in exponent units), so P stays <= 2^THR and the O rescale pass is skipped.
            // The decision is wave-uniform; when taken, O, l and the new P all move to the new max
            // before anything is accumulated, so no term is ever at a stale scale.
            if (!__all((tmax - m_run) * c <= RESCALE_THR)) {
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);     // first tile: 2^-inf = 0
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            const float mc = m_run * c;
            // two scores per VALU op where the ISA has a packed form (v_pk_fma_f32, v_pk_add_f32); v_exp_f32 is scalar
            bf16x8 pf[2][2];
          if constexpr (PVX) {
            static_assert(AT_SFMA, "the interleaved form uses the scalar-fma softmax");
            // Key chunk j = 2 b + h (16 keys: P fragment pf[b][h], elements r = 8 h .. 8 h + 7 of score block b) feeds the ND PV MFMAs of issue slots
            // [ND j, ND (j + 1)).  Chunk 0's exponentials come first; chunk j + 1's are spread over chunk j's MFMAs (pinned between their fragment
            // waits), so three quarters of the exponent work issues in the shadow of this wave's own MFMAs.
            const float nmc = -mc;
            float ps0 = 0.f, ps1 = 0.f;
            auto exp_pair = [&](auto J, auto Q) __attribute__((always_inline)) {        // pair Q (0..3) of chunk J
                constexpr int j = decltype(J)::value, b = j / 2, r = 8 * (j % 2) + 2 * decltype(Q)::value;
                asm volatile("" : "+v"(s[b][r]), "+v"(s[b][r + 1]));
                const float e0 = KM ? (s[b][r] - m_run) * c : __builtin_fmaf(s[b][r], c, nmc);
                const float e1 = KM ? (s[b][r + 1] - m_run) * c : __builtin_fmaf(s[b][r + 1], c, nmc);
                float p0 = __builtin_amdgcn_exp2f(e0), p1 = __builtin_amdgcn_exp2f(e1);
                asm volatile("" : "+v"(p0), "+v"(p1));
                ps0 += p0;
                ps1 += p1;
                pf[b][r >> 3][r & 7] = f2bf(p0);
                pf[b][r >> 3][(r & 7) + 1] = f2bf(p1);
            };
            static_for<0, 4>([&](auto Q) { exp_pair(std::integral_constant<int, 0>{}, Q); });
#if AT_PRIO == 1
            __builtin_amdgcn_s_setprio(1);
#endif
            static_for<0, NV>([&](auto N) __attribute__((always_inline)) {
                constexpr int n = decltype(N)::value, i = (n % ND) * 4 + n / ND, j = n / ND, d = n % ND;
                if constexpr (n + DV < NV) read_v(std::integral_constant<int, n + DV>{});
                lds_wait<(n + DV < NV ? DV : NV - 1 - n)>(vf[i]);
                o[i / 4] = LTX2_MFMA_32x32x16(as_bf16x8(vf[i]), pf[(i % 4) / 2][i % 2], o[i / 4], 0, 0, 0);
                if constexpr (j < 3) {      // 4 pairs of chunk j + 1 over the ND MFMAs of chunk j
                    static_for<(4 * d) / ND, (4 * (d + 1)) / ND>([&](auto Q) { exp_pair(std::integral_constant<int, j + 1>{}, Q); });
                }
            });
            l_run += ps0 + ps1;
          } else {
#if AT_SFMA
            // scalar-slot v_fma_f32 / v_add_f32 (packed f32 VALU beside MFMAs costs more than its two scalar halves: MI355X_MICROARCH.md)
            const float nmc = -mc;
            float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    // KM: (s - m) c, exact 0 for a masked key of a row whose every key so far is masked (s = m = AT_KEY_MASKED: the fma
                    // form would leave the rounding error of m c, ~1e22, in the exponent)
                    const float e0 = KM ? (s[b][r] - m_run) * c : __builtin_fmaf(s[b][r], c, nmc);
                    const float e1 = KM ? (s[b][r + 1] - m_run) * c : __builtin_fmaf(s[b][r + 1], c, nmc);
                    const float p0 = __builtin_amdgcn_exp2f(e0), p1 = __builtin_amdgcn_exp2f(e1);
                    ps0 += p0;
                    ps1 += p1;
                    pf[b][r >> 3][r & 7] = f2bf(p0);
                    pf[b][r >> 3][(r & 7) + 1] = f2bf(p1);
                }
            l_run += ps0 + ps1;
#else
            const f32x2 c2 = {c, c}, nmc2 = {-mc, -mc};
            f32x2 psum2 = {0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 sv = {s[b][r], s[b][r + 1]};
                    const f32x2 e = __builtin_elementwise_fma(sv, c2, nmc2);
                    const f32x2 pv = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
                    psum2 += pv;
                    pf[b][r >> 3][r & 7] = f2bf(pv[0]);
                    pf[b][r >> 3][(r & 7) + 1] = f2bf(pv[1]);
                }
            l_run += psum2[0] + psum2[1];
#endif

#if AT_PRIO == 1
            __builtin_amdgcn_s_setprio(1);
#elif AT_PRIO == 2
            __builtin_amdgcn_s_setprio(0);
#endif
            // ---- O^T += V^T . P^T ----
            static_for<0, NV>([&](auto N) {
                constexpr int n = decltype(N)::value, i = (n % ND) * 4 + n / ND;
                if constexpr (n + DV < NV) read_v(std::integral_constant<int, n + DV>{});
                lds_wait<(n + DV < NV ? DV : NV - 1 - n)>(vf[i]);
                o[i / 4] = LTX2_MFMA_32x32x16(as_bf16x8(vf[i]), pf[(i % 4) / 2][i % 2], o[i / 4], 0, 0, 0);
            });
          }       // PVX
#if AT_PRIO == 1
            __builtin_amdgcn_s_setprio(0);
#endif
        };

        const int t_unmasked_end = KM ? ta : min(tb, nfull);       // with a key mask EVERY tile takes the masked body
        int t = ta;
        for (; t + 1 < t_unmasked_end; t += 2) {
            tile(t, std::false_type{}, std::integral_constant<int, 0>{});
            tile(t + 1, std::false_type{}, std::integral_constant<int, 1>{});
        }
        if (t < t_unmasked_end) {
            tile(t, std::false_type{}, std::integral_constant<int, 0>{});
            ++t;
        }
        if constexpr (KM) {
            for (; t < tb; ++t) {
                if ((t - ta) & 1) tile(t, std::true_type{}, std::integral_constant<int, 1>{});
                else tile(t, std::true_type{}, std::integral_constant<int, 0>{});
            }
        } else if (nfull < tb) {
            if ((nfull - ta) & 1) tile(nfull, std::true_type{}, std::integral_constant<int, 1>{});
            else tile(nfull, std::true_type{}, std::integral_constant<int, 0>{});
        }

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        if constexpr (SK) {
            using SL = SkSlot<HD>;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.sk_ws, 0, 0x7fffffff, 0x00020000);
            if (tb < nt) {
                // a piece that does not end its unit (head or middle piece): publish (O, m, l) write-through (sc1: the bytes
                // leave this XCD's L2), drain, then ONE lane raises the flag.  A workgroup publishes at most once per launch.
                const int base = blockIdx.x * SL::BYTES + wv * SL::WAVE_BYTES + lane * 16;
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {o[d][g * 4], o[d][g * 4 + 1], o[d][g * 4 + 2], o[d][g * 4 + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, base + (d * 4 + g) * 1024, 0, 16);
                    }
                const f32x2 ml = {m_run, l_run};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ml), rs,
                                                      blockIdx.x * SL::BYTES + wv * SL::WAVE_BYTES + ND * 4 * 1024 + lane * 8, 0, 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(p.sk_flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
            if (ta > 0) {
                // tail piece: the earlier pieces of this unit were published by the workgroups `first .. j-1` of this group
                // (all lower-numbered, dispatched earlier; each did so BEFORE its own tail piece).  One lane polls relaxed, one
                // agent-scope acquire, barrier, then every wave folds its slabs in piece order (fixed -> bit-reproducible).
                const int X = ub * nt;
                int first = (int)((long)X * wlb / tot_b);
                while (range_lo(first + 1) <= X) ++first;
                while (range_lo(first) > X) --first;
                if (tid == 0) {
                    // bounded: a producer that never publishes (it cannot, see the progress argument at the launcher) must not
                    // hang the GPU -- after ~0.2 s of polling the piece is dropped and the sticky error word (flags[1023]) is set
                    for (int k = first; k < j; ++k) {
                        unsigned spins = 0;
                        while (__hip_atomic_load(p.sk_flags + k * slot_stride + slot_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > (1u << 20)) {
                                __hip_atomic_store(p.sk_flags + 1023, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                for (int k = first; k < j; ++k) {
                    const int sb = (k * slot_stride + slot_off) * SL::BYTES + wv * SL::WAVE_BYTES;
                    const f32x2 ml = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, sb + ND * 4 * 1024 + lane * 8, 0, 16));
                    const float m_new = fmaxf(ml[0], m_run);
                    const float a1 = __builtin_amdgcn_exp2f((ml[0] - m_new) * c), a2 = __builtin_amdgcn_exp2f((m_run - m_new) * c);
                    m_run = m_new;
                    l_run = ml[1] * a1 + l_run * a2;
#pragma unroll
                    for (int d = 0; d < ND; ++d)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, sb + lane * 16 + (d * 4 + g) * 1024, 0, 16));
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[d][g * 4 + e] = v[e] * a1 + o[d][g * 4 + e] * a2;
                        }
                }
                __syncthreads();            // every wave has its slabs: the flags go back to 0 for the next launch
                if (tid == 0)
                    for (int k = first; k < j; ++k)
                        __hip_atomic_store(p.sk_flags + k * slot_stride + slot_off, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }

        // ---- finalize: lane owns query q0+l31, dims d*32 + (r&3) + 8*(r>>2) + 4*hi ----
        float l_lo, l_hi;
        half_pair(l_run, l_lo, l_hi);
        const float l_tot = l_lo + l_hi;
        float inv = 1.0f / l_tot;
        const int qrow = q0 + l31;
        if (p.gate && qrow < p.Nq) inv *= 2.f / (1.f + __expf(-p.gate[(long)qrow * p.gate_ld + head]));        // per-head gate, folded into the normaliser
        if (qrow < p.Nq) {
            bf16* op = p.O + (long)qrow * p.ldo + head * HD + 4 * hi;
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = f2bf(o[d][g * 4 + e] * inv);
                    *(bf16x4*)(op + d * 32 + g * 8) = v;
                }
        }
    }
}

#if AT_FORM16
// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the same kernel on v_mfma_f32_16x16x32 blocks.  Why: the socket runs attention at its power cap, and tools/micro/mfma_power.hip measures the
// 16x16x32 form at 2.06-2.10 PF/s against 1.75-1.80 for 32x32x16 on random operands AT that cap (a quarter of the accumulator traffic per flop) -- the
// reason gemm_v4 uses it.  Same workgroup (4 waves x 32 query rows, 64-key tiles, the same LDS image and staging), per wave and tile:
//   S^T[kb][qb] (16 keys x 16 queries, kb < 4, qb < 2) += K[kb][ks] (A: 16 keys x 32 dims) . Q[qb][ks]^T (B)      4 * NKS fragment reads, 2 MFMAs each
//   O^T[db][qb] (16 dims x 16 queries, db < HD/16)     += V^T[db][kk] (A: 16 dims x 32 keys) . P[qb][kk]^T (B)    2 * NDB fragment reads, 2 MFMAs each
// Lane (c = lane & 15, g = lane >> 4) owns query column 16 qb + c of both query blocks; of a 16-key block it holds keys 4 g + r (r < 4), so a row's
// statistics live in the four lanes {c, c+16, c+32, c+48}: one v_permlane32_swap + one v_permlane16_swap fold BOTH query blocks' maxima at once.
// P never leaves registers: the B fragment of key half kk is [S[2kk][qb][0..3], S[2kk+1][qb][0..3]] = keys {32kk + 4g + r, 32kk + 16 + 4g + r} at MFMA
// k-slots 8g + e, so V^T keeps, inside every 32-key block, key 16h + 4g + r at position 8g + 4h + r (vt_transpose_kernel / gemm_v4's fused V^T epilogue
// under the same macro): the V^T fragment stays one conflict-free 16-byte LDS read.
// PV order: key half kk -> dim block db -> query block qb (the V^T fragment stays put over its two MFMAs; AT_DV fragments in flight).  PVX: the
// exponentials of key half 1 issue between the MFMAs of key half 0 (one pair per MFMA).  [key half -> query block -> dim block, which hides three
// quarters of the exponentials, needs all 8 fragments of a key half live at once: 25 registers spilled]
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

template <int HD, bool QS = false, bool KM = false, bool PVX = false>
__global__ __launch_bounds__(256, 2) void attn16_fwd_kernel(const AttnParams p) {
    using G = Geo<HD>;
    constexpr int K_TILE = G::K_TILE, STAGE = G::STAGE, NJ = G::NJ;
    constexpr int NKS = HD / 32, NDB = HD / 16;
    constexpr int NK = 4 * NKS, NV = 2 * NDB;       // K / V^T fragment reads per wave per tile (two MFMAs each)
    constexpr int DK = AT_DK < NK ? AT_DK : NK, DVW = PVX && HD == 128 ? AT_DV16X : AT_DV, DV = DVW < NV ? DVW : NV;     // (the interleaved form at head_dim 128 spills with 6 fragments in flight)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int nt = (p.Nkv + KVB - 1) / KVB, nfull = p.Nkv / KVB;

    // ---- staging: exactly the 32x32 kernel's (same LDS image, same source-side swizzle) ----
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

    // fragment addresses: K row 16 kb + c, 16-byte chunk 4 ks + g of the row; V^T row 16 db + c, chunk 4 kk + g (swizzled as staged)
    const int k_xor = HD == 128 ? l15 : ((l15 >> 1) & 7);
    const int v_xor = (l15 >> 1) & 7;
    unsigned k_lane[NKS], v_lane[2];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) k_lane[ks] = lds0 + l15 * (2 * HD) + (((4 * ks + g) ^ k_xor) << 4);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) v_lane[kk] = lds0 + K_TILE + l15 * 128 + (((4 * kk + g) ^ v_xor) << 4);

    const int head = blockIdx.y, qt = blockIdx.x, ta = 0, tb = nt;
    const int q0 = qt * QB + wv * 32;

    // ---- Q fragments (B operand): Q[q0 + 16 qb + c][32 ks + 8 g .. +8] ----
    bf16x8 qf[2][NKS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = min(q0 + 16 * qb + l15, p.Nq - 1);
        const bf16* qp = p.Q + (long)qrow * p.ldq + head * HD + 8 * g;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = *(const bf16x8*)(qp + 32 * ks);
    }
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

    f32x4 o[NDB][2];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) o[d][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    float c[2] = {p.scale_log2e, p.scale_log2e};
    if constexpr (QS) {
        // the four lanes of a row add a quarter of its partial sums each
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

#pragma unroll
    for (int i = 0; i < 2 * NJ; ++i) stage_piece(ta, 0, i);

    auto tile = [&](const int t, auto masked, auto par) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked)::value;
        constexpr int PAR = decltype(par)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tn = min(t + 1, tb - 1);
        constexpr int nbuf = 1 - PAR;

#if AT_PRIO == 1
        __builtin_amdgcn_s_setprio(1);
#endif
        // ---- S^T = K . Q^T ; fragment n = 4 ks + kb feeds the two query blocks (A stays put over the pair, 8 MFMAs between dependent ones) ----
        f32x4 s[4][2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) s[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 kf[NK];
        auto read_k = [&](auto N) {
            constexpr int n = decltype(N)::value, ks = n / 4, kb = n % 4;
            kf[n] = lds_read16<PAR * STAGE + kb * 16 * 2 * HD>(k_lane[ks]);
        };
        static_for<0, DK>(read_k);
        static_for<0, NK>([&](auto N) {
            constexpr int n = decltype(N)::value, ks = n / 4, kb = n % 4;
            if constexpr (n + DK < NK) read_k(std::integral_constant<int, n + DK>{});
            lds_wait<(n + DK < NK ? DK : NK - 1 - n)>(kf[n]);
            s[kb][0] = LTX2_MFMA_16x16x32(as_bf16x8(kf[n]), qf[0][ks], s[kb][0], 0, 0, 0);
            s[kb][1] = LTX2_MFMA_16x16x32(as_bf16x8(kf[n]), qf[1][ks], s[kb][1], 0, 0, 0);
            if constexpr ((n & 1) && n / 2 < 2 * NJ) stage_piece(tn, nbuf, n / 2);
        });
        static_for<NK / 2, 2 * NJ>([&](auto I) { stage_piece(tn, nbuf, decltype(I)::value); });

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
        // the first V^T fragments go out before the row maximum and the exponentials (which hide their latency); fragment n = NDB kk + db
        u32x4 vf[NV];
        auto read_v = [&](auto N) {
            constexpr int n = decltype(N)::value, kk = n / NDB, db = n % NDB;
            vf[n] = lds_read16<PAR * STAGE + db * 16 * 128>(v_lane[kk]);
        };
        static_for<0, DV>(read_v);

        float tmax[2] = {-INFINITY, -INFINITY};
        mfma16_result_guard(s, tmax[0], tmax[1]);
#if AT_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tmax[qb] = max3(tmax[qb], s[0][qb][r], s[1][qb][r]);
                tmax[qb] = max3(tmax[qb], s[2][qb][r], s[3][qb][r]);
            }
        quad_fold_max(tmax[0], tmax[1]);
        if (!__all((tmax[0] - m_run[0]) * c[0] <= RESCALE_THR && (tmax[1] - m_run[1]) * c[1] <= RESCALE_THR)) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float m_new = fmaxf(m_run[qb], tmax[qb]);
                const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c[qb]);
                m_run[qb] = m_new;
                l_run[qb] *= alpha;
#pragma unroll
                for (int d = 0; d < NDB; ++d) o[d][qb] *= alpha;
            }
        }
        const float nmc[2] = {-m_run[0] * c[0], -m_run[1] * c[1]};
        bf16x8 pf[2][2];                // [qb][kk]
        float ps[2] = {0.f, 0.f};
        // the 4 exponential pairs Q of P[qb][kk] (pair Q: elements 2 Q, 2 Q + 1 of the fragment = rows 2 (Q & 1), +1 of key block 2 kk + (Q >> 1))
        auto exp_pair = [&](auto KK, auto QB, auto Q) __attribute__((always_inline)) {
            constexpr int kk = decltype(KK)::value, qb = decltype(QB)::value, q = decltype(Q)::value, kb = 2 * kk + q / 2, r = 2 * (q % 2);
            asm volatile("" : "+v"(s[kb][qb]));
            const float e0 = KM ? (s[kb][qb][r] - m_run[qb]) * c[qb] : __builtin_fmaf(s[kb][qb][r], c[qb], nmc[qb]);
            const float e1 = KM ? (s[kb][qb][r + 1] - m_run[qb]) * c[qb] : __builtin_fmaf(s[kb][qb][r + 1], c[qb], nmc[qb]);
            float p0 = __builtin_amdgcn_exp2f(e0), p1 = __builtin_amdgcn_exp2f(e1);
            asm volatile("" : "+v"(p0), "+v"(p1));
            ps[qb] += p0 + p1;
            pf[qb][kk][2 * q] = f2bf(p0);
            pf[qb][kk][2 * q + 1] = f2bf(p1);
        };
        // pair index x < 8 of a key half: query block x / 4, pair x % 4
        auto exp_x = [&](auto KK, auto X) __attribute__((always_inline)) {
            constexpr int x = decltype(X)::value;
            exp_pair(KK, std::integral_constant<int, x / 4>{}, std::integral_constant<int, x % 4>{});
        };
        static_for<0, 8>([&](auto X) { exp_x(std::integral_constant<int, 0>{}, X); });
        if constexpr (!PVX) static_for<0, 8>([&](auto X) { exp_x(std::integral_constant<int, 1>{}, X); });
#if AT_PRIO == 1
        __builtin_amdgcn_s_setprio(1);
#endif
        // ---- O^T += V^T . P^T : fragment n = NDB kk + db feeds the two query blocks; PVX: key half 1's exponentials between key half 0's MFMAs ----
        static_for<0, NV>([&](auto N) __attribute__((always_inline)) {
            constexpr int n = decltype(N)::value, kk = n / NDB, db = n % NDB;
            if constexpr (n + DV < NV) read_v(std::integral_constant<int, n + DV>{});
            lds_wait<(n + DV < NV ? DV : NV - 1 - n)>(vf[n]);
            o[db][0] = LTX2_MFMA_16x16x32(as_bf16x8(vf[n]), pf[0][kk], o[db][0], 0, 0, 0);
            if constexpr (PVX && kk == 0) static_for<(8 * db) / NDB, (8 * db + 4) / NDB>([&](auto X) { exp_x(std::integral_constant<int, 1>{}, X); });
            o[db][1] = LTX2_MFMA_16x16x32(as_bf16x8(vf[n]), pf[1][kk], o[db][1], 0, 0, 0);
            if constexpr (PVX && kk == 0) static_for<(8 * db + 4) / NDB, (8 * (db + 1)) / NDB>([&](auto X) { exp_x(std::integral_constant<int, 1>{}, X); });
        });
        l_run[0] += ps[0];
        l_run[1] += ps[1];
#if AT_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    const int t_unmasked_end = KM ? ta : min(tb, nfull);
    int t = ta;
    for (; t + 1 < t_unmasked_end; t += 2) {
        tile(t, std::false_type{}, std::integral_constant<int, 0>{});
        tile(t + 1, std::false_type{}, std::integral_constant<int, 1>{});
    }
    if (t < t_unmasked_end) {
        tile(t, std::false_type{}, std::integral_constant<int, 0>{});
        ++t;
    }
    if constexpr (KM) {
        for (; t < tb; ++t) {
            if ((t - ta) & 1) tile(t, std::true_type{}, std::integral_constant<int, 1>{});
            else tile(t, std::true_type{}, std::integral_constant<int, 0>{});
        }
    } else if (nfull < tb) {
        if ((nfull - ta) & 1) tile(nfull, std::true_type{}, std::integral_constant<int, 1>{});
        else tile(nfull, std::true_type{}, std::integral_constant<int, 0>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- finalize: lane owns queries q0 + 16 qb + c, dims 16 db + 4 g + r.  Dim blocks 2i / 2i + 1 trade halves between lane rows (g, g ^ 1) with
    //      v_permlane16_swap so every lane stores 16 contiguous bytes: 8 dwordx4 stores per lane instead of 16 dwordx2 (the store tail is issue-bound) ----
    float lt[2] = {l_run[0], l_run[1]};
    quad_fold_sum(lt[0], lt[1]);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = q0 + 16 * qb + l15;
        float inv = 1.0f / lt[qb];
        if (p.gate && qrow < p.Nq) inv *= 2.f / (1.f + __expf(-p.gate[(long)qrow * p.gate_ld + head]));
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
            // rows (g odd) of ua <-> rows (g even) of ub: g even keeps its block-2i half and gets its neighbour's -> dims [16 (2i) + 8 (g >> 1), +8);
            // g odd gets its neighbour's block-(2i+1) half and keeps its own -> dims [16 (2i + 1) + 8 (g >> 1), +8)
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1" : "+v"(ua[0]), "+v"(ua[1]), "+v"(ub[0]), "+v"(ub[1]));
            const u32x4 w = {ua[0], ua[1], ub[0], ub[1]};
            if (qrow < p.Nq) *(u32x4*)(op + 32 * i) = w;
        }
    }
}
#endif  // AT_FORM16

// V [Nkv][ld] (head h at columns h*HD) -> VT[h][HD][Npad] with the key permutation
// pos(kv = 32b + 8g + 4hi + e) = 32b + 16(g>>1) + 8hi + 4(g&1) + e ; keys >= Nkv are zero-filled.
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
#if AT_FORM16
                const int kvl = 16 * ((pos >> 2) & 1) + 4 * (pos >> 3) + (pos & 3);      // position 8 g + 4 h + r holds key 16 h + 4 g + r (attn16_fwd_kernel)
#else
                const int ks = pos >> 4, hh = (pos >> 3) & 1, g0 = (pos >> 2) & 1, ee = pos & 3;
                const int kvl = 8 * (2 * ks + g0) + 4 * hh + ee; // inverse of pos()
#endif
                v[e] = tile[p0 + kvl][d];
            }
            *(bf16x8*)(dst + i * 8) = v;
        }
    }
}

}  // namespace

// Stream-K geometry for a problem: number of persistent workgroups (0 = use the plain grid) and whether the heads are
// dealt to the XCDs.  Only problems that need MORE than one round of the 2-per-CU slots are worth it.
static int sk_workers(const AttnParams& p, bool* xcd) {
    static int slots = 0;
    if (!slots) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        slots = 2 * prop.multiProcessorCount;
    }
    const int nqt = (p.Nq + QB - 1) / QB;
    const long units = (long)nqt * p.H;
    if (slots % 8) return 0;
    // Round 3: the stream-K forms are taken only when asked for (sk_force: the unit tests and tools/attn_sk_time.py).  Same-box, after the PVX form
    // below and s_setprio: self-attention N = 3456 plain 188.4-189.1 us vs stream-K 192.6-194.1 (round 2: 220 vs 208 -- the plain grid's
    // half-empty last round is exactly where the same-wave interleave pays); N = 13824 2608 vs 2684; the split-KV form for few query tiles
    // (68 audio queries x 3456 video keys, head_dim 64) 48.4 vs 38.4-39.1 plain (round 2: 28 vs 38).
    if (!p.sk_force) return 0;
    if (units * 4 <= slots && p.Nkv >= 16 * KVB) {
        // FEW query tiles against a long KV range: phase B alone cuts every unit's KV range over several workgroups (>= 4 tiles each), the
        // unit's last workgroup folds the pieces in order -- the split-KV form of the same hand-off
        const long nt = (p.Nkv + KVB - 1) / KVB;
        const long w = units * nt / 4;
        *xcd = false;
        return (int)(w < slots ? w : slots);
    }
    if (units <= slots) return 0;
    *xcd = p.H % 8 == 0 && (long)(p.H / 8) * nqt >= slots / 8;
    return slots;
}

long attn_sk_workspace_bytes(int head_dim) {
    const long slot = head_dim == 64 ? SkSlot<64>::BYTES : SkSlot<128>::BYTES;
    return 4096 + 1024L * slot;            // the flags (one 4-KiB page, zero before first use) + up to 1024 workgroup slots
}

int attn_launch(const AttnParams& p, hipStream_t stream) {
    LTX2_CHECK_ARG(p.Nq > 0 && p.Nkv > 0 && p.H > 0, "attention: empty problem");
    LTX2_CHECK_ARG(p.head_dim == 0 || p.head_dim == 128 || p.head_dim == 64, "attention: head_dim=%d, only 128 and 64 are implemented", p.head_dim);
    LTX2_CHECK_ARG(p.Npad % 64 == 0 && p.Npad >= p.Nkv, "attention: Npad=%d must be a multiple of 64 >= Nkv", p.Npad);
    LTX2_CHECK_ARG(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldo % 4 == 0, "attention: row strides must keep 16-byte alignment");
    LTX2_CHECK_ARG(p.ldk > 0 && ((long)p.Nkv + 2 * KVB) * p.ldk * 2 < (1L << 31), "attention: Nkv * ldk exceeds the 31-bit K byte offset");
    LTX2_CHECK_ARG((long)p.Npad * (p.head_dim == 64 ? 64 : 128) * 2 < (1L << 31), "attention: Npad * head_dim exceeds the 31-bit V^T byte offset");
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<128>::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<64>::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<128>::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<64>::LDS_BYTES);
    }
    // On the DiT's self-attention the stream-K form is 5 % faster as a kernel (208 vs 220 us in round 2) and worth 0.45 ms on the step -- less
    // than the slot arithmetic promises (0.84 -> 1.0) because the socket sits at its 1400 W cap: a half-empty last round also runs at a
    // higher clock.  Progress argument for the in-launch wait: a workgroup only ever waits on LOWER-numbered workgroups of its
    // group (same XCD when the heads are dealt), each XCD dispatches its workgroups in order, and the lowest-numbered unfinished
    // workgroup never waits on an unfinished one.
    bool xcd = false;
    if (p.q_ss)
        LTX2_CHECK_ARG(p.head_dim != 64 && p.q_ss_ld > 0 && p.q_ss_ld % 8 == 0 && p.q_norm_dim > 0, "attention: the per-row scale form needs head_dim 128 and q_ss_ld %% 8 == 0");
#if AT_FORM16
    {
        LTX2_CHECK_ARG(!p.q_ss || p.q_ss_ld % 16 == 0, "attention: the per-row scale form needs q_ss_ld %% 16 == 0");
        dim3 grid16((p.Nq + QB - 1) / QB, p.H);
        static int slots16 = 0;
        if (!slots16) {
            int dev = 0;
            hipDeviceProp_t prop;
            slots16 = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? 2 * prop.multiProcessorCount : 512;
        }
        const long units16 = (long)grid16.x * grid16.y, rem16 = units16 % slots16;
        const bool lone16 = units16 < slots16 || (rem16 != 0 && rem16 * 10 < (long)slots16 * 9);
        const bool pvx = AT_PVX16 == 1 || (AT_PVX16 == 2 && lone16 && !p.q_ss && !p.kmask);
#define AT16_LAUNCH(HDV, QSV, KMV, PVXV)                                                                                                                  \
        do {                                                                                                                                              \
            static PerDeviceOnce once_;                                                                                                                   \
            if (once_.first())                                                                                                                            \
                (void)hipFuncSetAttribute((const void*)attn16_fwd_kernel<HDV, QSV, KMV, PVXV>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<HDV>::LDS_BYTES); \
            hipLaunchKernelGGL((attn16_fwd_kernel<HDV, QSV, KMV, PVXV>), grid16, dim3(256), Geo<HDV>::LDS_BYTES, stream, p);                               \
        } while (0)
        if (p.head_dim == 64) {
            if (p.kmask) AT16_LAUNCH(64, false, true, false);
            else if (pvx) AT16_LAUNCH(64, false, false, true);
            else AT16_LAUNCH(64, false, false, false);
        } else if (p.kmask) {
            if (p.q_ss) AT16_LAUNCH(128, true, true, false);
            else AT16_LAUNCH(128, false, true, false);
        } else if (p.q_ss) {
            if (AT_PVX16 == 1) AT16_LAUNCH(128, true, false, true);
            else AT16_LAUNCH(128, true, false, false);
        } else if (pvx) AT16_LAUNCH(128, false, false, true);
        else AT16_LAUNCH(128, false, false, false);
#undef AT16_LAUNCH
        LTX2_CHECK_LAUNCH("attn16_fwd_kernel");
        return LTX2_OK;
    }
#endif
    const int workers = (p.sk_ws && !p.kmask) ? sk_workers(p, &xcd) : 0;
    if (workers > 0) {
        LTX2_CHECK_ARG(workers < 1024 && p.sk_ws_bytes >= attn_sk_workspace_bytes(p.head_dim), "attention: stream-K workspace too small");
        AttnParams q = p;
        q.sk_xcd = xcd;
        q.sk_flags = (unsigned*)p.sk_ws;
        q.sk_ws = (char*)p.sk_ws + 4096;
        if (p.q_ss) {
            static PerDeviceOnce sq_once;
            if (sq_once.first())
                (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<128>::LDS_BYTES);
            hipLaunchKernelGGL((attn_fwd_kernel<128, true, true>), dim3(workers), dim3(256), Geo<128>::LDS_BYTES, stream, q);
        } else if (p.head_dim == 64)
            hipLaunchKernelGGL((attn_fwd_kernel<64, true>), dim3(workers), dim3(256), Geo<64>::LDS_BYTES, stream, q);
        else
            hipLaunchKernelGGL((attn_fwd_kernel<128, true>), dim3(workers), dim3(256), Geo<128>::LDS_BYTES, stream, q);
        LTX2_CHECK_LAUNCH("attn_fwd_kernel<SK>");
        return LTX2_OK;
    }
    dim3 grid((p.Nq + QB - 1) / QB, p.H);
    if (p.kmask) {      // masked text cross-attention: plain grid, every tile on the masked body
        static PerDeviceOnce km_once;
        if (km_once.first()) {
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<128>::LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<128>::LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<64, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<64>::LDS_BYTES);
        }
        if (p.head_dim == 64)
            hipLaunchKernelGGL((attn_fwd_kernel<64, false, false, true>), grid, dim3(256), Geo<64>::LDS_BYTES, stream, p);
        else if (p.q_ss)
            hipLaunchKernelGGL((attn_fwd_kernel<128, false, true, true>), grid, dim3(256), Geo<128>::LDS_BYTES, stream, p);
        else
            hipLaunchKernelGGL((attn_fwd_kernel<128, false, false, true>), grid, dim3(256), Geo<128>::LDS_BYTES, stream, p);
        LTX2_CHECK_LAUNCH("attn_fwd_kernel<KM>");
        return LTX2_OK;
    }
    if (p.q_ss) {
        static PerDeviceOnce qs_once;
        if (qs_once.first())
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<128>::LDS_BYTES);
        hipLaunchKernelGGL((attn_fwd_kernel<128, false, true>), grid, dim3(256), Geo<128>::LDS_BYTES, stream, p);
        LTX2_CHECK_LAUNCH("attn_fwd_kernel<QS>");
        return LTX2_OK;
    }
    // a grid whose last (or only) round leaves SIMDs with ONE wave of this kernel takes the PVX form (see the kernel's template comment)
    static int slots = 0;
    if (!slots) {
        int dev = 0;
        hipDeviceProp_t prop;
        slots = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? 2 * prop.multiProcessorCount : 512;
    }
    const long units = (long)grid.x * grid.y, rem = units % slots;
    const bool lone = units < slots || (rem != 0 && rem * 10 < (long)slots * 9);
    if (lone) {
        static PerDeviceOnce px_once;
        if (px_once.first()) {
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<128>::LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<64, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<64>::LDS_BYTES);
        }
        if (p.head_dim == 64)
            hipLaunchKernelGGL((attn_fwd_kernel<64, false, false, false, true>), grid, dim3(256), Geo<64>::LDS_BYTES, stream, p);
        else
            hipLaunchKernelGGL((attn_fwd_kernel<128, false, false, false, true>), grid, dim3(256), Geo<128>::LDS_BYTES, stream, p);
        LTX2_CHECK_LAUNCH("attn_fwd_kernel<PVX>");
        return LTX2_OK;
    }
    if (p.head_dim == 64)
        hipLaunchKernelGGL((attn_fwd_kernel<64, false>), grid, dim3(256), Geo<64>::LDS_BYTES, stream, p);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<128, false>), grid, dim3(256), Geo<128>::LDS_BYTES, stream, p);
    LTX2_CHECK_LAUNCH("attn_fwd_kernel");
    return LTX2_OK;
}

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
