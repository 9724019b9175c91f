// Flash-attention forward v2 for gfx950 (large problems): 8 wave64 x 32 query rows = 256 query
// rows of one head per workgroup (one per CU), 64-key tiles, head_dim 128.
//
// Same arithmetic and register layouts as attention.hip (swapped products S^T = K.Q^T and
// O^T = V^T.P^T, key-permuted V^T, exp2-domain online softmax with deferred rescale); what changes
// is the schedule:
//   * K/V tiles live in a 4-stage LDS ring (4 x 32 KiB) filled by LDS-DMA (global_load_lds) that
//     stays in flight across barriers (counted vmcnt, raw s_barrier): tile t+2 is issued while
//     tile t is multiplied;
//   * each wave alternates a matrix interval  M(t) = { S(t) = K(t).Q^T , O += V(t-1).P(t-1) }
//     (32 MFMAs + their 32 ds_read_b128 + 4 LDS-DMA issues) with a vector interval
//     V(t) = online softmax of S(t) -> P(t)   (the PV product is software-pipelined by one tile);
//   * the two wave groups (waves 0-3 / 4-7, i.e. the two waves of every SIMD) run staggered by
//     one barrier, so on each SIMD the matrix pipe (group in M) and the VALU/transcendental pipe
//     (group in V) are busy at the same time.
// Stage lifetime: K(t) is read in M(t), V(t) in M(t+1); tile t+2 is DMA'd into stage (t+2)&3 =
// (t-2)&3 during M(t), two intervals after its last reader.  vmcnt(4) at the end of M(t) retires
// tile t+1 (this wave's share) one barrier before anyone reads it.
#include <stdlib.h>

#include "attention.h"

namespace {

constexpr int QB = 256, KVB = 64, HD = 128;
constexpr int K_TILE = KVB * HD * 2;           // 16 KiB, rows of 256 B
constexpr int V_TILE = HD * KVB * 2;           // 16 KiB, rows of 128 B
constexpr int STAGE = K_TILE + V_TILE;
constexpr int NSTAGE = 4;
constexpr int LDS_BYTES = NSTAGE * STAGE;      // 128 KiB
constexpr float RESCALE_THR = 6.0f;

__device__ __forceinline__ void glds16(const bf16* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void*)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

#define AP_BARRIER()                            \
    do {                                        \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_s_barrier();           \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

__global__ __launch_bounds__(512, 2) void attn_fwd_pp_kernel(const AttnParams p, int nqt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wv >> 2;
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-contiguous block order: an XCD (private L2) walks the query tiles of a few heads
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int head = id / nqt;
    const int q0 = (id - head * nqt) * QB + wv * 32;

    // ---- Q fragments (B operand): Q[q0 + l31][16*ks + 8*hi .. +8] ----
    bf16x8 qf[8];
    {
        const int qrow = min(q0 + l31, p.Nq - 1);
        const bf16* qp = p.Q + (long)qrow * p.ldq + head * HD + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + 16 * ks);
    }

    // ---- LDS-DMA source addresses: 2 K + 2 V^T instructions per wave per tile ----
    const bf16* k_src[2];
    const bf16* v_src[2];
    int k_rowi[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kr = (wv * 2 + j) * 4 + (lane >> 4);                 // 0..63
        k_rowi[j] = kr;
        k_src[j] = p.K + head * HD + ((lane & 15) ^ (kr & 15)) * 8;
        const int vr = (wv * 2 + j) * 8 + (lane >> 3);                 // 0..127
        v_src[j] = p.VT + (long)head * p.vt_head_stride + (long)vr * p.Npad + ((lane & 7) ^ ((vr >> 1) & 7)) * 8;
    }
    const int nt = (p.Nkv + KVB - 1) / KVB;
    auto issue_tile = [&](int t) {
        char* sk = smem + (t & (NSTAGE - 1)) * STAGE + wv * 2048;
        char* sv = sk + K_TILE;
        const int kv0 = min(t, nt - 1) * KVB;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(k_src[j] + (long)min(kv0 + k_rowi[j], p.Nkv - 1) * p.ldk, sk + j * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(v_src[j] + kv0, sv + j * 1024);
    };

    f32x16 o[4], s[2];
    bf16x8 pf[2][2];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[b][k2][e] = f2bf(0.f);
    float m_run = -INFINITY, l_run = 0.f;
    const int k_xor = l31 & 15;
    const int v_xor = (l31 >> 1) & 7;
    const float c = p.scale_log2e;

    // Per-lane LDS read offsets.  The XOR swizzle makes every (ks) / (b,k2) fragment a distinct
    // lane-dependent address: 8 + 4 persistent VGPRs; the (uniform) stage offset is added per tile
    // through an opaque SGPR so the compiler cannot hoist 4 stages x 32 full addresses into
    // registers (that version spilled); the b / d sub-block strides are immediates.
    int k_off[8], v_off[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) k_off[ks] = l31 * 256 + (((2 * ks + hi) ^ k_xor) << 4);
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) v_off[bk] = K_TILE + l31 * 128 + (((2 * bk + hi) ^ v_xor) << 4);

    // S(t) = K(t).Q^T into s[]  (stage t&3)
    auto mma_s = [&](int t) {
        int so = (t & (NSTAGE - 1)) * STAGE;
        asm volatile("" : "+s"(so));
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[b][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(smem + (k_off[ks] + so) + b * 8192);
                s[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[b], 0, 0, 0);
            }
        }
    };
    // O^T += V^T(t).P(t)^T  (stage t&3)
    auto mma_pv = [&](int t) {
        int so = (t & (NSTAGE - 1)) * STAGE;
        asm volatile("" : "+s"(so));
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const bf16x8 vf = *(const bf16x8*)(smem + (v_off[2 * b + k2] + so) + d * 4096);
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[b][k2], o[d], 0, 0, 0);
                }
        }
    };
    // online softmax of s[] (tile t) -> pf[]
    auto softmax = [&](int t) {
        const int kv0 = t * KVB;
        if (kv0 + KVB > p.Nkv) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (kv >= p.Nkv) s[b][r] = -INFINITY;
                }
        }
        float tmax = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, fmaxf(s[0][r], s[1][r]));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        if (!__all((tmax - m_run) * c <= RESCALE_THR)) {
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
        const float mc = m_run * c;
        float psum = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[b][r], c, -mc));
                psum += pv;
                pf[b][r >> 3][r & 7] = f2bf(pv);
            }
        l_run += psum;
    };
// Scheduling pattern of a matrix interval: keep DEPTH fragment reads in flight ahead of the MFMAs
// (hipcc otherwise emits ds_read -> s_waitcnt lgkmcnt(0) -> mfma per fragment, exposing the LDS
// latency 32 times per tile) and sprinkle the LDS-DMA issues between MFMAs.
#ifndef AP_DEPTH
#define AP_DEPTH 3
#endif
#define SGB_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define SGB_DSR(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define SGB_VMEM(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0)
#ifdef AP_NOPIPE
#define AP_PIPE(NPAIR, NVMEM) do {} while (0)
#else
#define AP_PIPE(NPAIR, NVMEM)                                      \
    do {                                                           \
        SGB_DSR(AP_DEPTH);                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < (NPAIR)-AP_DEPTH; ++i_) { \
            SGB_MFMA(1);                                           \
            SGB_DSR(1);                                            \
            if ((NVMEM) && (i_ & 3) == 3 && (i_ >> 2) < (NVMEM)) SGB_VMEM(1); \
        }                                                          \
        SGB_MFMA(AP_DEPTH);                                        \
    } while (0)
#endif
#define AP_PIN_S() asm volatile("" : "+v"(s[0]), "+v"(s[1]))
#define AP_PIN_O() asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]))
#define AP_PIN_P() asm volatile("" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[1][0]), "+v"(pf[1][1]))

    unsigned long long* dbg = (p.dbg && blockIdx.x == 0 && lane == 0 && (wv == 0 || wv == 4)) ? (unsigned long long*)p.dbg + grp * 256 : nullptr;
    int dbi = 0;
#define AP_STAMP(T) do { if (dbg && (T) >= 8 && (T) < 12 && dbi < 256) dbg[dbi++] = __builtin_amdgcn_s_memtime(); } while (0)
    // ---- prologue: tiles 0..3 in flight, tile 0 landed ----
    issue_tile(0);
    issue_tile(1);
    issue_tile(2);
    issue_tile(3);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    AP_BARRIER();
    if (grp == 1) AP_BARRIER();          // stagger the second group by one interval

    // ---- M(0): S(0) only ----
    __builtin_amdgcn_s_setprio(1);
    mma_s(0);
    AP_PIPE(16, 0);
    __builtin_amdgcn_s_setprio(0);
    AP_PIN_S();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // tile 1 landed (this wave's share)
    AP_BARRIER();
    for (int t = 0; t < nt; ++t) {
        // ---- V(t): softmax of S(t) ----
        AP_STAMP(t);
        softmax(t);
        AP_PIN_P();
        AP_PIN_O();
        AP_STAMP(t);
        AP_BARRIER();
        AP_STAMP(t);
        // ---- M(t+1): S(t+1) (if any) and O += V(t).P(t); LDS-DMA of tile t+3 ----
        __builtin_amdgcn_s_setprio(1);
        if (t + 1 < nt) {
            mma_s(t + 1);
            if (t + 1 >= 2) issue_tile(t + 3);
            mma_pv(t);
            AP_PIPE(32, 4);
        } else {
            mma_pv(t);
            AP_PIPE(16, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        AP_PIN_S();
        AP_PIN_O();
        AP_STAMP(t);
        if (t + 1 >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile t+2 landed
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              // t+1 == 1: tiles 2 landed, 3 outstanding
        AP_STAMP(t);
        AP_BARRIER();
    }
    if (grp == 0) AP_BARRIER();          // rebalance the barrier count
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- finalize: lane owns query q0+l31, dims d*32 + (r&3) + 8*(r>>2) + 4*hi ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + l31;
    if (qrow < p.Nq) {
        bf16* op = p.O + (long)qrow * p.ldo + head * HD + 4 * hi;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f2bf(o[d][g * 4 + e] * inv);
                *(bf16x4*)(op + d * 32 + g * 8) = v;
            }
    }
}

}  // namespace

int attn_pp_launch(const AttnParams& p_in, hipStream_t stream) {
    AttnParams p = p_in;
    {
        const char* d = getenv("LTX2_ATTN_DBG");
        p.dbg = d ? (void*)strtoull(d, nullptr, 0) : nullptr;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const int nqt = (p.Nq + QB - 1) / QB;
    hipLaunchKernelGGL(attn_fwd_pp_kernel, dim3(nqt * p.H), dim3(512), LDS_BYTES, stream, p, nqt);
    LTX2_CHECK_LAUNCH("attn_fwd_pp_kernel");
    return LTX2_OK;
}
