// 4-wave bf16 MFMA GEMM / implicit-GEMM conv3d for gfx950 with a hand-scheduled K loop: one wave per SIMD, accumulators in
// AGPRs.  The whole K loop is ONE asm block generated and order-checked by gen_gemm_v4.py (gemm_v4_loop.inc): hipcc schedules
// nothing in it.  Wave layouts (template LAYOUT), tile BM x BN x 64:
//   (0-2: round 2's 1x4 / 2x2 layouts on 32x32x16 and 2x2 on 16x16x32 -- deleted in round 3, they lost every same-box comparison)
//   3  BN 256, 1x4 waves, 16x16x32: 14|16 x 4 blocks of 16 (BM 224|256; balanced for 224 rows) -- the DiT default
//   4  BN 128, 4x1 waves, 16x16x32: wave w owns rows [BM/4 w, +BM/4) x all 128 columns, 7|8 x 8 blocks (BM 448|512) -- the
//      VAE decoder's 128-channel convs
//   7  (round 4) BN 64, 4x1 waves, 16x16x32: 8 x 4 blocks per wave (BM 512), conv only, N <= 64 with the columns past N masked -- the VAE
//      decoder's conv_out (128 -> 48 channels)
//   5  fp8 COMPUTE (round 3, BASELINE config 3): BOTH operands e4m3fn codes, K-tile = 128 elements (the same 128-byte LDS rows, swizzle
//      and DMA schedule as 64 bf16), 1x4 waves, v_mfma_f32_32x32x64_f8f6f4 (fp8 x fp8 at twice the bf16 rate): 7|8 x 2 blocks of 32;
//      the per-row activation scale and the per-column weight scale multiply the accumulators in the epilogue
// What the measurements on MI355X say: with random operands the chip is POWER-limited -- the 32x32x16
// K loop issues an MFMA every 34 cycles but the chip clocks at 1.45-1.55 GHz, an MFMA-only loop at 2.0 GHz -- and the
// 16x16x32 instruction (half the accumulator-register traffic per flop) sustains 12-20 % more on the same shapes.
// Operands: activations [M][lda] bf16 (dense) or a PADDED channels-last volume [T+2][H+2][W+2][Cin] (conv: every tap of a
// row is the row's base address plus one wave-uniform offset, see the generator), weights [N][K] bf16, K contiguous.
// N % BN == 0, K % 128 == 0, K >= 256; everything else stays on gemm_pp.hip / gemm.hip.  Same LDS image, swizzle, tile order
// and fused epilogues (gemm_epilogue.h) as those kernels; outputs are bit-identical to the ping-pong kernel's (tested).
#include <stdlib.h>

#include "gemm_epilogue.h"
#include "gemm_v4_loop.inc"

namespace {

typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));

#define V4_INS_DENSE : "{v[0:15]}"(voffA), "{v[16:23]}"(voffB), "{v[24:27]}"(addrA), "{v[28:31]}"(addrB), [ra] "s"(ra), [rw] "s"(rw), \
                       [nk] "s"(nk), [la] "s"(la), [lw] "s"(lw), [kb] "s"(kb)
#define V4_INS_CONV V4_INS_DENSE, "{v32}"(tapv), [lcpt] "s"(lcpt), [cptm1] "s"(cptm1)
#define V4_OUT16                                                                                                          \
    "={a[0:15]}"(acc[0]), "={a[16:31]}"(acc[1]), "={a[32:47]}"(acc[2]), "={a[48:63]}"(acc[3]), "={a[64:79]}"(acc[4]),   \
        "={a[80:95]}"(acc[5]), "={a[96:111]}"(acc[6]), "={a[112:127]}"(acc[7]), "={a[128:143]}"(acc[8]),                 \
        "={a[144:159]}"(acc[9]), "={a[160:175]}"(acc[10]), "={a[176:191]}"(acc[11]), "={a[192:207]}"(acc[12]),          \
        "={a[208:223]}"(acc[13]), "={a[224:239]}"(acc[14]), "={a[240:255]}"(acc[15])
#define V4_ASM(LOOP) asm volatile(LOOP : V4_OUT16 V4_INS_DENSE : LTX2_V4_CLOBBERS)
#define V4_ASM_CONV(LOOP) asm volatile(LOOP : V4_OUT16 V4_INS_CONV : LTX2_V4_CLOBBERS)

template <int LAYOUT, int BM, bool W8 = false>
struct V4Geo {
    static constexpr bool L41 = LAYOUT == 4 || LAYOUT == 7;        // 4x1 waves: wave w owns rows [BM/4 w, +BM/4) x all BN columns
    static constexpr int BN = LAYOUT == 4 ? 128 : LAYOUT == 7 ? 64 : 256;
    static_assert(LAYOUT == 3 || LAYOUT == 4 || LAYOUT == 5 || LAYOUT == 6 || LAYOUT == 7, "wave layout");
    static constexpr int MB = LAYOUT == 5 ? 32 : 16;               // MFMA block
    static constexpr bool L14 = !L41;                              // 1x4 waves
    static constexpr int WM = L41 ? BM / 4 : BM;
    static constexpr int WN = LAYOUT == 4 ? 128 : 64;
    static constexpr int RBW = (WM + MB - 1) / MB, CBW = WN / MB;  // blocks per wave (first wave row)
    static constexpr int NKS = MB == 16 ? 2 : 4;
    static constexpr int WROW = W8 ? 64 : 128;                     // bytes of one weight row per K-tile (fp8 codes / bf16)
    static constexpr int NPA = BM / 32, NPW = BN * WROW / 4096;    // 1-KiB LDS-DMA pieces per wave and K-tile
    static constexpr int A_STAGE = L41 ? BM * 128 : 32768, W_BASE = 2 * A_STAGE, W_STAGE = BN * WROW;
    static_assert(!W8 || LAYOUT == 3, "fp8-resident weights run on layout 3");
    static constexpr int LOOP_BYTES = W_BASE + 2 * W_STAGE;
    static constexpr int EPI_BYTES = 4 * WM * WN * 2;      // bf16 outputs leave through LDS (per-wave slabs)
    static constexpr int LDS_BYTES = LOOP_BYTES > EPI_BYTES ? LOOP_BYTES : EPI_BYTES;
    static_assert(LAYOUT == 4 ? (BM == 384 || BM == 448 || BM == 512) : LAYOUT == 7 ? BM == 512 : (BM == 224 || BM == 256), "tile rows");
    static_assert(LAYOUT != 6 || BM == 224, "layout 6 tile rows");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// One output tile (the whole kernel body); `bid` = this workgroup's block number.  Must be inlined exactly once per kernel: the K loop is an
// asm block over physical registers.
template <int EPI, int LAYOUT, int BM, bool CONV, int VAR>
__device__ __forceinline__ void gemm_v4_tile(const GemmParams& p, const int bid) {
    constexpr bool W8 = VAR == 20;          // fp8-resident weights: p.W8 codes [N][K] + p.wscale[N]
    constexpr bool F8 = LAYOUT == 5 || LAYOUT == 6;        // fp8 compute (5: 32x32x64 blocks, 6: 16x16x128 blocks): p.A8 codes [M][lda] + p.ascale[M], p.W8 codes [N][K] + p.wscale[N]
    using G = V4Geo<LAYOUT, BM, W8>;
    constexpr int TBN = G::BN, NPA = G::NPA, NPW = G::NPW, MB = G::MB, WM = G::WM, WN = G::WN, RBW = G::RBW, CBW = G::CBW, NKS = G::NKS;
    static_assert(!CONV || !(LAYOUT == 5 || LAYOUT == 6), "conv runs on the bf16 layouts");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = G::L41 ? w : 0, wc = G::L41 ? 0 : w;
#ifdef LTX2_V4_PROBE
    const unsigned long long t_k0 = __builtin_amdgcn_s_memtime();
#endif

    // ---- block -> tile (XCD-contiguous, grouped row-tiles; as gemm_pp.hip) ----
    // split-K (p.splitk > 1): blockIdx = split * tiles + tile; this block accumulates K-tiles [split * nk, +nk) and writes an
    // fp32 partial tile into slab `split` of p.out ([splitk][M][ldo] fp32); splitk_reduce_kernel adds the slabs
    // round 6 (GemmParams "fold" fields): the dense bf16 layout-3 kernels only
    // (VAR == 30: its own instantiations -- compiled into the plain kernels the extra parameters and LDS cost them 2-4 % with nothing folded: same-box layer
    //  traces, profiles/r06_layer_trace_*.txt)
    constexpr bool FOLD_OK = LAYOUT == 3 && !CONV && VAR == 30;
    constexpr bool FOLD_CONS = FOLD_OK && (EPI == EPI_BF16 || EPI == EPI_GELU_BF16);         // consumer side: row factors
    constexpr bool FOLD_PROD = FOLD_OK && EPI == EPI_RESID_GATE_F32;                          // producer side: bf16 shadow + partial sums of squares
    const int Mt = (p.M + BM - 1) / BM, Nt = LAYOUT == 7 ? 1 : p.N / TBN;        // (layout 7: ONE column tile, N <= 64, columns >= N masked)
    const int ntiles = Mt * Nt;
    const int split = p.splitk > 1 ? bid / ntiles : 0;
    const int id = xcd_remap(bid - split * ntiles, ntiles);
    constexpr int GROUP = 8;
    const int per_group = GROUP * Nt;
    const int g = id / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(Mt - first_m, GROUP);
    const int rem = id - g * per_group;
    const int m0 = (first_m + rem % gsz) * BM;
    const int n0 = (rem / gsz) * TBN;

    // ---- LDS-DMA sources: piece j of this wave = 8 tile rows x 128 B; chunk swizzle on the source, row clamp at the ragged edge ----
    u32x16 voffA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    u32x8 voffB = {0, 0, 0, 0, 0, 0, 0, 0};
    const int Hp = p.H + 2, Wp = p.Wd + 2;
    // layout 3, 224-row tiles, dense bf16 weights: the ragged last row tile runs a K loop over only the row blocks that hold real
    // rows (6 of 14 for 3456 rows, 10 of 14 for 13824) -- fewer DMA pieces per wave, so the wave -> tile-row mapping shrinks with it
    constexpr bool SHORT_OK = LAYOUT == 3 && BM == 224 && !CONV;
    const int valid_rows = p.M - m0;
    const int short_rb = !SHORT_OK ? 0 : valid_rows <= 96 ? 6 : valid_rows <= 160 ? 10 : 0;      // block-uniform
    const int npa_rt = short_rb ? short_rb / 2 : NPA;
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int r = (w * npa_rt + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int m = min(m0 + r, p.M - 1);
        if constexpr (CONV) {       // output position (t, h, w) = padded position of its (0,0,0) tap
            const int hw = p.H * p.Wd;
            const int t = m / hw, r2 = m - t * hw, h = r2 / p.Wd, x = r2 - h * p.Wd;
            voffA[j] = (unsigned)((t * Hp + h) * Wp + x) * (unsigned)(p.Cin * 2) + chunk * 16;
        } else {
            voffA[j] = (unsigned)m * (unsigned)(p.lda * (F8 ? 1 : 2)) + chunk * 16;
        }
    }
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        if constexpr (W8) {     // 16 rows x 64 B of codes per piece; 8-byte chunks swizzled with 2*((row>>2)&3) (pairs stay 16-byte units)
            const int r = (w * NPW + j) * 16 + (lane >> 2);
            const int c8 = (2 * (lane & 3)) ^ (2 * ((r >> 2) & 3));
            voffB[j] = (unsigned)(n0 + r) * (unsigned)p.K + c8 * 8;
        } else {
            const int r = (w * NPW + j) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);
            const int wrow = LAYOUT == 7 ? min(n0 + r, p.N - 1) : n0 + r;      // (layout 7: weight rows past N repeat the last one; their columns are never stored)
            voffB[j] = (unsigned)wrow * (unsigned)(p.K * (F8 ? 1 : 2)) + chunk * 16;
        }
    }
    // ---- fragment read addresses (first block of the wave): row lr, 16-byte chunk (kchunk(ks) + kq) ^ ((row >> 1) & 7) ----
    const int lr = lane & (MB - 1), kq = lane / MB;         // row in block, k-quarter (32x32x16: 0..1, 16x16x32: 0..3)
    const int xbase = kq ^ ((lr >> 1) & 7);
    u32x4 addrA = {0, 0, 0, 0}, addrB = {0, 0, 0, 0};
    if constexpr (F8) {
        // a lane's fragment of a 64-element k-step = 32 bytes = the two chunks (4 ks + 2 kq + j), j = 0, 1, of its row: register [2 ks + j]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned c = (unsigned)((4 * (q >> 1) + 2 * kq + (q & 1)) ^ ((lr >> 1) & 7)) << 4;
            addrA[q] = lds0 + (wr * WM + lr) * 128 + c;
            addrB[q] = lds0 + G::W_BASE + (wc * WN + lr) * 128 + c;
        }
    }
#pragma unroll
    for (int ks = 0; ks < (F8 ? 0 : NKS); ++ks) {
        const unsigned c = (unsigned)(((MB == 32 ? 2 : 4) * ks) ^ xbase) << 4;
        const unsigned a0 = lds0 + (wr * WM + lr) * 128 + c;
        const unsigned b0 = W8 ? lds0 + G::W_BASE + (wc * WN + lr) * 64 + ((unsigned)((4 * ks + kq) ^ (2 * ((lr >> 2) & 3))) << 3)
                               : lds0 + G::W_BASE + (wc * WN + lr) * 128 + c;
        if constexpr (MB == 16) {       // per-stage bases: [ks + 2 * stage]
            addrA[ks] = a0;
            addrA[ks + 2] = a0 + G::A_STAGE;
            addrB[ks] = b0;
            addrB[ks + 2] = b0 + G::W_STAGE;
        } else {
            addrA[ks] = a0;
            addrB[ks] = b0;
        }
    }
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(F8 ? (void*)p.A8 : (void*)p.A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((W8 || F8) ? (void*)p.W8 : (void*)p.W, 0, 0x7fffffff, 0x00020000);
    const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + w * npa_rt * 1024);
    const unsigned lw = __builtin_amdgcn_readfirstlane(lds0 + G::W_BASE + w * NPW * 1024);
    // split-K: split i takes K-tiles [2 floor(i h / s), 2 floor((i + 1) h / s)), h = half the K-tiles -- even counts that differ by at most 2
    const unsigned nkt = p.K / (F8 ? 128 : 64);
    const unsigned kb = __builtin_amdgcn_readfirstlane(p.splitk > 1 ? 2 * ((unsigned)split * (nkt / 2) / (unsigned)p.splitk) : 0u);
    const unsigned nk = __builtin_amdgcn_readfirstlane(p.splitk > 1 ? 2 * (((unsigned)split + 1) * (nkt / 2) / (unsigned)p.splitk) - kb : nkt);
    // conv: lane i holds the byte offset of tap i = (a*3 + b)*3 + c (a absent for per-frame 3x3 convs) in the padded volume
    unsigned tapv = 0;
    unsigned lcpt = 0, cptm1 = 0;
    if constexpr (CONV) {
        const int ntap = 9 * p.taps_t;
        const int a = p.taps_t == 3 ? lane / 9 : 0, bc = p.taps_t == 3 ? lane % 9 : lane;
        tapv = lane < ntap ? (unsigned)((a * Hp + bc / 3) * Wp + bc % 3) * (unsigned)(p.Cin * 2) : 0u;
        lcpt = __builtin_amdgcn_readfirstlane(p.cin_shift - 6);
        cptm1 = __builtin_amdgcn_readfirstlane((p.Cin >> 6) - 1);
    }

    // bias / gate-table column vectors of this lane (4 consecutive columns per column block).  Layout 3 requests the bias BEFORE the K
    // loop: 16 VGPRs above the loop's register range carry it across, and that L2 round trip is off the epilogue's start.
    constexpr int NG = MB == 32 ? 4 : 1;
    constexpr bool HOIST_COL_VECTORS = LAYOUT == 3 && MB == 16 && BM == 224 && !CONV &&        // (the instantiations with the registers to spare: checked with -Rpass-analysis)
                                       (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_RESID_GATE_F32);
    f32x4 bias4[CBW][NG];
    f32x4 gate4[CBW][NG];
    auto load_bias = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int gq = 0; gq < NG; ++gq)
                bias4[cb][gq] = (p.bias && (LAYOUT != 7 || n0 + wc * WN + cb * MB + 8 * gq + 4 * kq < p.N)) ? *(const f32x4*)(p.bias + n0 + wc * WN + cb * MB + 8 * gq + 4 * kq)
                                                                                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto load_gate = [&]() __attribute__((always_inline)) {       // (the gated-residual kernel has no room to carry these across the loop too)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int gq = 0; gq < NG; ++gq)
                gate4[cb][gq] = (EPI == EPI_RESID_GATE_F32 && p.gate_table) ? *(const f32x4*)(p.gate_table + n0 + wc * WN + cb * MB + 8 * gq + 4 * kq)
                                                                             : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // consumer side of a folded norm: the producer's partial sums of squares of this tile's rows ([rf_nparts][BM] floats) arrive by LDS-DMA in the LDS above
    // the stage buffers while the K loop runs (no register carries them: the loop owns the file); behind the loop BM threads turn them into row factors
    constexpr int RF_OFF = G::LOOP_BYTES, RF_FAC = RF_OFF + GEMM_RF_MAX_PARTS * BM * 4;
    if constexpr (FOLD_CONS) {
        if (p.rf_parts) {       // (block-uniform)
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)p.rf_parts, 0, 0x7fffffff, 0x00020000);
            const int total = p.rf_nparts * (BM / 4);          // 16-byte pieces, part-major
            for (int it = 0; it * 256 < total; ++it) {
                const int idx = it * 256 + tid;
                if (idx < total) {
                    const int part = idx / (BM / 4), c4 = idx - part * (BM / 4);
                    const unsigned voff = (unsigned)(((long)part * p.rf_ld + m0 + 4 * c4) * 4);
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr_t)(smem + RF_OFF + it * 4096 + w * 1024), 16, voff, 0, 0, 0);
#else
                    (void)voff;
#endif
                }
            }
        }
    }
    [[maybe_unused]] float rf[FOLD_CONS ? RBW : 1];
    if constexpr (HOIST_COL_VECTORS) load_bias();
    // ---- gated fp32-residual epilogue (EPI_RESID_GATE_F32 through LDS, below): the read-back side's geometry, declared here because the FIRST
    //      part's residual rows are requested BEFORE the K loop where the loop leaves registers (224-row dense bf16 loops end at v179; round 4):
    //      x is this tile's alone, so the rows can be read any time, and the epilogue then starts with its first 32 rows already on chip
    //      instead of behind an HBM round trip that all 256 tiles of a one-round grid take at the same moment ----
    constexpr bool RESID_LDS_OK = EPI == EPI_RESID_GATE_F32 && (LAYOUT == 3 || LAYOUT == 5 || LAYOUT == 6) && VAR != 9;
    constexpr bool RESID_PRELOAD = RESID_LDS_OK && LAYOUT == 3 && BM == 224 && !W8 && !CONV;        // (must match launch_v4's LDS size)
    constexpr int ROWB = WN * 4, PR = 32, RPP = PR / MB, NP = RBW / RPP, NIT = PR / 4;   // bytes per slab row, rows / row blocks per part, parts, readback steps per part
    const bool rowgate = p.gate && p.gate_stride != 0;          // (block-uniform) per-row gates: per-token timesteps (image conditioning)
    const int rr = lane >> 4, cc = lane & 15;                            // readback: 4 rows x 16 chunks per instruction
    float* xg = (float*)p.out + (long)(m0 + rr) * p.ldo + n0 + wc * WN + cc * 4;
    const int nparts = min(NP, (p.M - m0 + PR - 1) / PR);                // (block-uniform) ragged last row tile
    const float* gg = rowgate ? p.gate + (long)(m0 + rr) * p.gate_stride + n0 + wc * WN + cc * 4 : nullptr;
    // (as plain loads into registers the compiler spilled them around the asm block: the LDS-DMA needs no register; the 32 KiB above the
    //  stage buffers are this kernel's alone: 8 KiB per wave = one part of 32 rows x 256 B, lane i's 16 bytes at byte 16 i of every KiB)
    constexpr int X0_OFF = G::LOOP_BYTES;
    if constexpr (RESID_PRELOAD) {
        if (!rowgate) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const long grow = min((long)m0 + i * 4 + rr, (long)p.M - 1);
                const unsigned voff = (unsigned)((grow * p.ldo + n0 + wc * WN + cc * 4) * 4);
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(smem + X0_OFF + w * (PR * ROWB) + i * 1024), 16, voff, 0, 0, 0);
#else
                (void)voff;
#endif
            }
        }
    }
    f32x16 acc[16];
#ifdef LTX2_V4_PROBE
    const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (W8) {
        // per-column dequantisation scale of this lane's weight rows (column cb*16 + lr of the wave's 64), as (s, s) pairs
        u32x8 scl;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const float sv = p.wscale[n0 + wc * WN + cb * MB + lr];
            scl[2 * cb] = scl[2 * cb + 1] = __builtin_bit_cast(unsigned, sv);
        }
        if constexpr (BM == 224) {
            if (short_rb == 6)
                asm volatile(LTX2_V4_L14_M16_RB6_W8 : V4_OUT16 V4_INS_DENSE, LTX2_V4_L14_M16_RB6_W8_SCL(scl) : LTX2_V4_L14_M16_RB6_W8_CLOBBERS);
            else if (short_rb == 10)
                asm volatile(LTX2_V4_L14_M16_RB10_W8 : V4_OUT16 V4_INS_DENSE, LTX2_V4_L14_M16_RB10_W8_SCL(scl) : LTX2_V4_L14_M16_RB10_W8_CLOBBERS);
            else
                asm volatile(LTX2_V4_L14_M16_RB14_W8 : V4_OUT16 V4_INS_DENSE, LTX2_V4_L14_M16_RB14_W8_SCL(scl) : LTX2_V4_L14_M16_RB14_W8_CLOBBERS);
        } else
            asm volatile(LTX2_V4_L14_M16_RB16_W8 : V4_OUT16 V4_INS_DENSE, LTX2_V4_L14_M16_RB16_W8_SCL(scl) : LTX2_V4_L14_M16_RB16_W8_CLOBBERS);
    } else if constexpr (CONV) {
        if constexpr (LAYOUT == 3) {
            if constexpr (BM == 224) V4_ASM_CONV(LTX2_V4_L14_M16_RB14_CONV);
            else V4_ASM_CONV(LTX2_V4_L14_M16_RB16_CONV);
        } else {
            if constexpr (LAYOUT == 7) V4_ASM_CONV(LTX2_V4_L41N_M16_RB8_CONV);
            else if constexpr (BM == 384) V4_ASM_CONV(LTX2_V4_L41_M16_RB6_CONV);
            else if constexpr (BM == 448) V4_ASM_CONV(LTX2_V4_L41_M16_RB7_CONV);
            else V4_ASM_CONV(LTX2_V4_L41_M16_RB8_CONV);
        }
    } else if constexpr (LAYOUT == 3) {
        if constexpr (BM == 224) {
            if (short_rb == 6) V4_ASM(LTX2_V4_L14_M16_RB6);
            else if (short_rb == 10) V4_ASM(LTX2_V4_L14_M16_RB10);
            else V4_ASM(LTX2_V4_L14_M16_RB14);
        } else {
#ifdef LTX2_V4_PROBE
            if constexpr (VAR == 1) V4_ASM(LTX2_V4_L14_M16_RB16_NODMA);
            else if constexpr (VAR == 2) V4_ASM(LTX2_V4_L14_M16_RB16_NOREAD);
            else
#endif
                V4_ASM(LTX2_V4_L14_M16_RB16);
        }
    } else if constexpr (LAYOUT == 4) {
        if constexpr (BM == 448) V4_ASM(LTX2_V4_L41_M16_RB7);
        else V4_ASM(LTX2_V4_L41_M16_RB8);
    } else if constexpr (LAYOUT == 6) {     // fp8 x fp8 on 16x16x128 blocks (224-row tiles only: the fragment registers of 16 row blocks do not fit)
        static_assert(BM == 224, "layout 6 tile rows");
        V4_ASM(LTX2_V4_F8_M16_RB14);
    } else {                                // layout 5: fp8 x fp8
        if constexpr (BM == 224) V4_ASM(LTX2_V4_F8_RB7);
        else V4_ASM(LTX2_V4_F8_RB8);
    }
#ifdef LTX2_V4_PROBE
    const unsigned long long t_loop1 = __builtin_amdgcn_s_memtime();
    if (p.dbg && bid == 0 && tid == 0) {
        ((unsigned long long*)p.dbg)[0] = t_loop1 - t_loop0;
        ((unsigned long long*)p.dbg)[2] = t_loop0 - t_k0;
    }
#endif

    GemmParams pe = p;
    if (EPI == EPI_F32 && p.splitk > 1) pe.out = (float*)p.out + (long)split * p.M * p.ldo;
    // ---- epilogue (gemm_epilogue.h): a lane owns ONE row of every row block and 4-column groups of it ----
    // 32x32 block: row lr, groups gq = 0..3 at columns 8 gq + 4 kq (accumulator registers 4 gq .. 4 gq + 3)
    // 16x16 block: row lr, one group at columns 4 kq (accumulator registers 0..3)
    if constexpr (!HOIST_COL_VECTORS) load_bias();
    // (fold-specific addresses are formed from these copies: built from lr / kq directly hipcc forms them in FRONT of the loop and carries them across it in scratch)
    [[maybe_unused]] int lr_e = lr, tid_e = tid;
    if constexpr (FOLD_OK) asm volatile("" : "+v"(lr_e), "+v"(tid_e));
    if constexpr (FOLD_CONS) {
        if (p.rf_parts && tid_e < BM) {       // row tid of the tile: the partials in part order (deterministic), then the RMS factor
            const float* pl = (const float*)(smem + RF_OFF);
            float sq = 0.f;
            for (int j = 0; j < p.rf_nparts; ++j) sq += pl[j * BM + tid_e];
            *(float*)(smem + RF_FAC + tid_e * 4) = rsqrtf(sq / (float)p.rf_dim + p.rf_eps);
        }
    }
    load_gate();
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            asm volatile("" : "+v"(bias4[cb][gq]));      // retire the loads once, here
            if (EPI == EPI_RESID_GATE_F32) asm volatile("" : "+v"(gate4[cb][gq]));
        }
    // fp8 compute: sum_k a8 w8 is exact in the fp32 accumulator's terms; the row scale of the activations and the column scale of the
    // weights come back here, one multiply per output (out = as[m] * ws[n] * acc)
    float as_row[F8 ? RBW : 1];
    f32x4 ws4[F8 ? CBW : 1][NG];
    if constexpr (F8) {
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) as_row[rb] = p.ascale[min(m0 + wr * WM + rb * MB + lr, p.M - 1)];
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) ws4[cb][gq] = *(const f32x4*)(p.wscale + n0 + wc * WN + cb * MB + 8 * gq + 4 * kq);
    }
    auto acc_group = [&](int rb, int cb, int gq) -> f32x4 {
        // accumulator block index as the generator numbers it: rb * cbw + cb (both wave rows share cbw = CBW)
        const int blk = rb * CBW + cb;
        if constexpr (F8 && MB == 32) {
            const f32x16& a = acc[blk];
            return f32x4{a[4 * gq], a[4 * gq + 1], a[4 * gq + 2], a[4 * gq + 3]} * (ws4[cb][gq] * as_row[rb]);
        } else if constexpr (F8) {
            const f32x16& a = acc[blk >> 2];
            const int o = (blk & 3) * 4;
            return f32x4{a[o], a[o + 1], a[o + 2], a[o + 3]} * (ws4[cb][gq] * as_row[rb]);
        } else if constexpr (MB == 32) {
            const f32x16& a = acc[blk];
            return f32x4{a[4 * gq], a[4 * gq + 1], a[4 * gq + 2], a[4 * gq + 3]};
        } else {
            const f32x16& a = acc[blk >> 2];
            const int o = (blk & 3) * 4;
            return f32x4{a[o], a[o + 1], a[o + 2], a[o + 3]};
        }
    };
    constexpr bool BF16_OUT = EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_SILU_BF16;
    const bool resid_lds = RESID_LDS_OK;
    if constexpr (RESID_LDS_OK) {
      if (resid_lds) {
        // x += gate * (acc + bias) through LDS (the gate: row-invariant for a scalar sigma and folded in before the slab; per ROW for per-token
        // timesteps and then applied at the read-back, its rows prefetched beside the residual's): a lane's accumulator groups are 4 columns
        // of 16 different rows, so touching x straight from them moves 16 x 64 B per instruction -- half of every 128-byte line,
        // twice.  Transposed through this wave's share of the dead stage buffers (half a wave tile at a time: rows x 256 B, 16-byte
        // chunks XOR-swizzled with the row) every global load / store instruction covers 4 whole 256-byte row segments.
        static_assert(RBW % RPP == 0 && ROWB == 256 && 4 * 2 * PR * ROWB <= G::LDS_BYTES, "resid epilogue slab");
        // The wave tile leaves in parts of two row blocks (32 rows).  Each wave owns a double-buffered slab of its own, so after
        // the one barrier that ends the K loop nothing synchronises: slab write of part p, then the residual rows of part p + 2
        // are requested, then part p is read back row-contiguously, added and stored -- two parts' loads are always in flight.
        // Written as straight-line code over compile-time part numbers, once per gate form (round 4: with a runtime `break` in the
        // unrolled loop and the gate form tested inside it hipcc kept the in-flight rows in a scratch array -- global_load ->
        // s_waitcnt vmcnt(0) -> scratch_store per row group).
        auto run = [&](auto RG) __attribute__((always_inline)) {
            constexpr bool ROWGATE = decltype(RG)::value;
            f32x4 g4[CBW][NG];
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                for (int gq = 0; gq < NG; ++gq) {
                    g4[cb][gq] = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (!ROWGATE && (p.gate || p.gate_table)) {
                        g4[cb][gq] = gate4[cb][gq];
                        if (p.gate) g4[cb][gq] += *(const f32x4*)(p.gate + n0 + wc * WN + cb * MB + 8 * gq + 4 * kq);
                    }
                }
            char* wl = smem + w * (2 * PR * ROWB);
            f32x4 gtab = {0.f, 0.f, 0.f, 0.f};              // the read-back lane's 4 columns of the broadcast part
            if (ROWGATE && p.gate_table) gtab = *(const f32x4*)(p.gate_table + n0 + wc * WN + cc * 4);
            // round 6: the new residual rows also leave as the NEXT projection's operand (bf16, times the next norm's 1 + scale) with their partial sums
            // of squares, so the norm pass between this GEMM and the next disappears (GemmParams::shadow; row-invariant gates only)
            const bool shadow = FOLD_PROD && !ROWGATE && p.shadow != nullptr;          // (block-uniform)
            f32x4 sm4 = {1.f, 1.f, 1.f, 1.f};
            bf16* yg = nullptr;
            float* sspart = (float*)(smem + 4 * (2 * PR * ROWB)) + w * BM;       // this wave's strip sums per tile row, behind the four waves' slabs
            if constexpr (FOLD_PROD && !ROWGATE) {
                if (shadow) {
                    if (p.shadow_scale) sm4 += *(const f32x4*)(p.shadow_scale + n0 + wc * WN + cc * 4);
                    yg = p.shadow + (long)(m0 + rr) * p.ld_shadow + n0 + wc * WN + cc * 4;
                }
            }
            f32x4 xv[3][NIT];
            [[maybe_unused]] f32x4 gv[ROWGATE ? 3 : 1][NIT];
            auto load_part = [&](auto PART) __attribute__((always_inline)) {
                constexpr int part = decltype(PART)::value, slot = part % 3;
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    const long grow = min((long)m0 + part * PR + i * 4 + rr, (long)p.M - 1) - (m0 + rr);
                    xv[slot][i] = *(const f32x4*)(xg + grow * p.ldo);
                    if constexpr (ROWGATE) gv[slot][i] = *(const f32x4*)(gg + grow * p.gate_stride);
                }
            };
            if constexpr (RESID_PRELOAD && !ROWGATE) {      // landed long ago (the K loop waits vmcnt(0) every K-tile): 8 conflict-free LDS reads
#pragma unroll
                for (int i = 0; i < NIT; ++i) xv[0][i] = *(const f32x4*)(smem + X0_OFF + w * (PR * ROWB) + i * 1024 + lane * 16);
            } else {
                load_part(std::integral_constant<int, 0>{});
            }
            if (nparts > 1) load_part(std::integral_constant<int, 1>{});
            __syncthreads();                // every wave has finished its fragment reads: the stage buffers are free
            static_for<0, NP>([&](auto PART) __attribute__((always_inline)) {
                constexpr int part = decltype(PART)::value;
                if (part < nparts) {            // (block-uniform)
                    char* sl = wl + (part & 1) * (PR * ROWB);
#pragma unroll
                    for (int r = 0; r < RPP; ++r) {
                        const int rb = part * RPP + r, row = r * MB + lr;
#pragma unroll
                        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                            for (int gq = 0; gq < NG; ++gq) {
                                const f32x4 v = g4[cb][gq] * (acc_group(rb, cb, gq) + bias4[cb][gq]);
                                const int chunk = (cb * MB + 8 * gq + 4 * kq) >> 2;          // 16-byte chunk of this lane's 4 columns
                                *(f32x4*)(sl + row * ROWB + ((chunk ^ (row & 15)) << 4)) = v;
                            }
                    }
                    if constexpr (part + 2 < NP) {
                        if (part + 2 < nparts) load_part(std::integral_constant<int, part + 2>{});
                    }
#pragma unroll
                    for (int i = 0; i < NIT; ++i) {
                        const int row = i * 4 + rr;
                        asm volatile("" : "+v"(xv[part % 3][i]));       // consume in issue order: counted waits, not vmcnt(0)
                        f32x4 d = *(const f32x4*)(sl + row * ROWB + ((cc ^ (row & 15)) << 4));
                        if constexpr (ROWGATE) d *= gtab + gv[part % 3][i];
                        const f32x4 xn = xv[part % 3][i] + d;
                        const bool live = m0 + part * PR + row < p.M;
                        if (live) *(f32x4*)(xg + ((long)part * PR + row - rr) * p.ldo) = xn;
                        if constexpr (FOLD_PROD && !ROWGATE) {
                            if (shadow) {
                                // the row's 64 columns of this wave sit in the 16 lanes of one DPP row: quad, quad, half-mirror, mirror
                                float ss = (xn[0] * xn[0] + xn[1] * xn[1]) + (xn[2] * xn[2] + xn[3] * xn[3]);
                                ss += dpp_f32<0xB1>(ss);
                                ss += dpp_f32<0x4E>(ss);
                                ss += dpp_f32<0x141>(ss);
                                ss += dpp_f32<0x140>(ss);
                                const f32x4 y = xn * sm4;
                                if (live) {
                                    *(bf16x4*)(yg + ((long)part * PR + row - rr) * p.ld_shadow) = pack_bf16x4(y[0], y[1], y[2], y[3]);
                                }
                                if (cc == 0) sspart[part * PR + row] = ss;
                            }
                        }
                    }
                }
            });
            if constexpr (FOLD_PROD && !ROWGATE) {
                if (shadow) {       // the tile's four 64-column strip sums, added in wave order: one partial per (row, 256-column tile), stored tile-major (coalesced)
                    __syncthreads();
                    const float* sp = (const float*)(smem + 4 * (2 * PR * ROWB));
                    if (tid < BM && m0 + tid < p.M) p.shadow_ss[(long)(n0 / TBN) * p.ld_ss + m0 + tid] = (sp[tid] + sp[BM + tid]) + (sp[2 * BM + tid] + sp[3 * BM + tid]);
                }
            }
        };
        if (rowgate) run(std::true_type{});
        else run(std::false_type{});
      }
    }
    if constexpr (EPI == EPI_RESID_GATE_F32) {
      if (!resid_lds) {
        // x += gate * (acc + bias).  The accumulators live in AGPRs and the fragment registers are dead, so the VGPR file
        // is free: the residual tile is read HALF A WAVE TILE AT A TIME with every load in flight at once (32 x 16 B per
        // lane), i.e. the HBM latency is paid twice per tile instead of once per row block (measured: one row block per
        // round trip cost 46-75 k cycles of a 165 k-cycle block).
        constexpr int RBH = (RBW + 1) / 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 xs[RBH][CBW][NG];
#pragma unroll
            for (int r = 0; r < RBH; ++r) {
                const int rb = half * RBH + r;
                if (rb >= RBW || wr * WM + rb * MB >= BM || m0 + wr * WM + rb * MB >= p.M) continue;     // (block-uniform)
                const int row = min(m0 + wr * WM + rb * MB + lr, p.M - 1);        // clamped for the load; the store checks
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                    for (int gq = 0; gq < NG; ++gq)
                        xs[r][cb][gq] = *(const f32x4*)((const float*)p.out + (long)row * p.ldo + (n0 + wc * WN + cb * MB + 8 * gq + 4 * kq));
            }
#pragma unroll
            for (int r = 0; r < RBH; ++r) {
                const int rb = half * RBH + r;
                if (rb >= RBW || wr * WM + rb * MB >= BM || m0 + wr * WM + rb * MB >= p.M) continue;
                const int row = m0 + wr * WM + rb * MB + lr;
                const bool ok = row < p.M;
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                    for (int gq = 0; gq < NG; ++gq) {
                        asm volatile("" : "+v"(xs[r][cb][gq]));       // consume in issue order: counted waits, not vmcnt(0)
                        const int col = n0 + wc * WN + cb * MB + 8 * gq + 4 * kq;
                        const f32x4 v = acc_group(rb, cb, gq) + bias4[cb][gq];
                        f32x4 gt = {1.f, 1.f, 1.f, 1.f};
                        if (p.gate || p.gate_table) {
                            gt = gate4[cb][gq];
                            if (p.gate && ok) gt += *(const f32x4*)(p.gate + (long)row * p.gate_stride + col);
                        }
                        if (ok) *(f32x4*)((float*)p.out + (long)row * p.ldo + col) = xs[r][cb][gq] + gt * v;
                    }
            }
        }
      }
    } else if constexpr (VAR != 9 && (BF16_OUT || EPI == EPI_ADD_BF16)) {
        // bf16 outputs of the row-slab layouts leave through LDS: a lane's accumulator groups are 4 columns of 16 different
        // rows (8-byte stores into 32-byte row segments); transposed through this wave's share of the (now dead) stage
        // buffers every store instruction writes whole rows: 8 rows x 128 B (layout 3) or 4 rows x 256 B (layout 4).
        // LDS image [WM rows][WN bf16], 16-byte chunks XOR-swizzled with the row: conflict-free for the 8-byte writes
        // (16 rows x one column group per lane group) and for the 16-byte row reads.
        constexpr int ROWB = WN * 2, CPR = ROWB / 16, RPI = 64 / CPR;          // bytes per row, chunks per row, rows per instruction
        __syncthreads();                                    // every wave has finished its fragment reads
        if constexpr (FOLD_CONS) {
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) rf[rb] = p.rf_parts ? *(const float*)(smem + RF_FAC + (rb * MB + lr_e) * 4) : 1.f;
        }
        char* wl = smem + w * (WM * ROWB);
        if constexpr (EPI == EPI_ADD_BF16) {
            // out = bf16(acc + bias + res): the residual tile comes in through the same slab with whole-row loads; each lane then
            // replaces ITS 8 bytes (4 columns of one row) by the sum, rounded once -- bit-identical to adding from global memory
            // LDS-DMA: one instruction lands 1 KiB = RPI whole slab rows with no register in between, so the wave's WM / RPI requests are
            // all in flight at once behind ONE wait (as plain loads hipcc emitted load -> s_waitcnt vmcnt(0) -> ds_write per instruction: a
            // memory round trip per 1 KiB, 16 of the 128-channel conv's 70 us per tile).  The DMA writes lane i's 16 bytes at byte 16 i, so
            // the chunk swizzle moves to the source: physical chunk c of slab row r holds the row's chunk c ^ sw(r).
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int it = 0; it < WM / RPI; ++it) {
                const int r = it * RPI + lane / CPR, c = lane % CPR;
                const int sw = CPR == 8 ? ((r >> 1) & 7) : (r & 15);
                const int row = min(m0 + wr * WM + r, p.M - 1);
                const unsigned voff = (unsigned)(((long)row * p.ldres + n0 + wc * WN + (c ^ sw) * 8) * 2);
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_ptr_t)(wl + it * 1024), 16, voff, 0, 0, 0);
#else
                (void)voff;
#endif
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
            if (m0 + wr * WM + rb * MB >= p.M) continue;     // no real row in this block (block-uniform): its slab rows are never read as data
            const int r = rb * MB + lr;
            const int sw = CPR == 8 ? ((r >> 1) & 7) : (r & 15);
            // (EPI_ADD_BF16) the row block's residual slots are read up front: in source order read -> add -> write per slot, hipcc keeps
            // every LDS read behind the previous slot's write (same char* slab) and the block becomes CBW * NG LDS round trips
            [[maybe_unused]] bf16x4 rsv[CBW][NG];
            if constexpr (EPI == EPI_ADD_BF16) {
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                    for (int gq = 0; gq < NG; ++gq) {
                        const int byte = (cb * MB + 8 * gq + 4 * kq) * 2;
                        rsv[cb][gq] = *(const bf16x4*)(wl + r * ROWB + (((byte >> 4) ^ sw) << 4) + (byte & 15));
                    }
            }
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
              for (int gq = 0; gq < NG; ++gq) {
                f32x4 v;
                if constexpr (FOLD_CONS) {
                    v = acc_group(rb, cb, gq) * rf[rb] + bias4[cb][gq];
                } else {
                    v = acc_group(rb, cb, gq) + bias4[cb][gq];
                }
                if (EPI == EPI_GELU_BF16) {
                    const f32x2 g0 = gelu_tanh2(f32x2{v[0], v[1]}), g1 = gelu_tanh2(f32x2{v[2], v[3]});
                    v = f32x4{g0[0], g0[1], g1[0], g1[1]};
                }
                if (EPI == EPI_SILU_BF16) v = f32x4{silu_f(v[0]), silu_f(v[1]), silu_f(v[2]), silu_f(v[3])};
                const int byte = (cb * MB + 8 * gq + 4 * kq) * 2;        // this lane's 4 columns inside the slab row
                const int chunk = (byte >> 4) ^ sw;
                bf16x4* slot = (bf16x4*)(wl + r * ROWB + chunk * 16 + (byte & 15));
                if constexpr (EPI == EPI_ADD_BF16) {
                    const bf16x4 rs = rsv[cb][gq];
                    v += f32x4{bf2f(rs[0]), bf2f(rs[1]), bf2f(rs[2]), bf2f(rs[3])};
                }
                *slot = pack_bf16x4(v[0], v[1], v[2], v[3]);
              }
        }
        if ((LAYOUT == 3 || LAYOUT == 5 || LAYOUT == 6) && EPI == EPI_BF16 && !CONV && p.vt && n0 >= p.vt_col0) {
            // V tile of a fused QKV projection: this wave's 64 columns are 64 dims of ONE head; leave as V^T rows
            // vt[head][d][.] with attention's key order inside every 32-key block: position 8 g + 4 h + r holds
            // key 16 h + 4 g + r.  One instruction stores 16 dims x 64 bytes (4 chunks of 8 positions); every chunk is
            // two runs of 4 consecutive keys gathered from the slab with 2-byte LDS reads.  Keys >= M are zeros (up to Npad).
            const int cw = n0 + wc * WN - p.vt_col0;
            const int head = cw / p.vt_hd, d0 = cw - head * p.vt_hd;
            bf16* vrow = p.vt + (long)head * p.vt_head_stride + (long)(d0 + (lane >> 2)) * p.vt_npad + m0;
#pragma unroll 1
            for (int dg = 0; dg < WN / 16; ++dg) {
                const int c = dg * 16 + (lane >> 2);                     // column of the slab = dim d0 + c
                const int coff = (c & 7) * 2, cchunk = c >> 3;
#pragma unroll
                for (int cg = 0; cg < BM / 32; ++cg) {
                    const int ch = cg * 4 + (lane & 3);                   // 8-position chunk of the tile's key range
                    const int kbase = (ch >> 2) * 32 + (ch & 3) * 4;                              // chunk g of a 32-key block: keys 4 g + r, then 16 + 4 g + r
                    constexpr int RUN2 = 16;
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = kbase + (e >> 2) * RUN2 + (e & 3);
                        const bf16 x = *(const bf16*)(wl + r * ROWB + ((cchunk ^ ((r >> 1) & 7)) << 4) + coff);
                        v[e] = (m0 + r < p.M) ? x : f2bf(0.f);
                    }
                    if (m0 + ch * 8 < p.vt_npad) *(bf16x8*)(vrow + (long)dg * 16 * p.vt_npad + ch * 8) = v;
                }
            }
        } else {
            // slab rows leave in batches of RBATCH instructions: the LDS reads of a batch go out together (one latency per batch, not per row group)
            constexpr int RBATCH = 4;
            static_assert((WM / RPI) % RBATCH == 0, "readback batches");
            u32x4 rbv[RBATCH];
#pragma unroll
            for (int it = 0; it < WM / RPI; ++it) {
                const int r = it * RPI + lane / CPR, c = lane % CPR;
                if (it % RBATCH == 0) {
#pragma unroll
                    for (int b = 0; b < RBATCH; ++b) {
                        const int rb2 = (it + b) * RPI + lane / CPR;
                        const int sw2 = CPR == 8 ? ((rb2 >> 1) & 7) : (rb2 & 15);
                        rbv[b] = *(const u32x4*)(wl + rb2 * ROWB + ((c ^ sw2) << 4));
                    }
                }
                const u32x4 v = rbv[it % RBATCH];
                const int row = m0 + wr * WM + r;
                if (row < p.M && (LAYOUT != 7 || n0 + wc * WN + c * 8 < p.N)) *(u32x4*)((bf16*)p.out + (long)row * p.ldo + n0 + wc * WN + c * 8) = v;
                if constexpr (EPI == EPI_BF16 && !CONV && (LAYOUT == 3 || LAYOUT == 5 || LAYOUT == 6)) {
                    if (p.rowss) {      // (block-uniform) squared norm of this wave's 64-column strip of the row, from the ROUNDED values
                        const bf16x8 h = as_bf16x8(v);
                        float ss = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ss += bf2f(h[e]) * bf2f(h[e]);
                        ss += __shfl_xor(ss, 1);
                        ss += __shfl_xor(ss, 2);
                        ss += __shfl_xor(ss, 4);
                        if (c == 0 && row < p.M) p.rowss[(long)row * (p.N / 64) + (n0 + wc * WN) / 64] = ss;
                    }
                }
            }
        }
    } else if constexpr (EPI == EPI_D2S_BF16 && CONV && LAYOUT == 3) {
        // depth-to-space scatter (+ the tiled d2s(x) residual).  Column n = s Cf + c with Cf >= 64 a power of two: the wave's 64 columns share s,
        // so the sub-position (da, db, dd) is wave-uniform and computed once; a lane's rows advance by MB positions per row block, so (t, h, w)
        // is stepped, not divided; and the residual elements of a row block (2-byte reads 16 bytes apart: column c of the output is channel
        // (c mod Cin/sp) sp + s of the input) are all requested before the first is used.  Written per group as in round 2 (epi_store4: three
        // integer divisions, four loads, a wait and a store per 4 outputs, 56 times per lane) this epilogue cost 17-28 us per tile.
        const int colw = n0 + wc * WN;
        const int s_idx = colw >> p.cf_shift, c_base = colw & (p.Cf - 1);
        const int fhw = p.fh * p.fw;
        const int da = s_idx / fhw, db = (s_idx - da * fhw) / p.fw, dd = s_idx - da * fhw - db * p.fw;
        const int sp = p.ft * fhw, cmask = p.c_d2s - 1;
        const bf16* xs = p.res ? p.res : p.A;
        const int Ho = p.H * p.fh, Wo = p.Wd * p.fw;
        int row = m0 + lr;
        const int hw = p.H * p.Wd;
        int pt = row / hw, ph = (row - pt * hw) / p.Wd, pw = row - pt * hw - ph * p.Wd;
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb, row += MB) {
            if (rb) {
                pw += MB;
                while (pw >= p.Wd) {
                    pw -= p.Wd;
                    if (++ph >= p.H) {
                        ph = 0;
                        ++pt;
                    }
                }
            }
            const int to = pt * p.ft + da - p.drop_first;
            const bool ok = row < p.M && to >= 0;
            unsigned short rs[CBW][4];
            if (p.d2s_residual) {
                const bf16* xr = xs + (long)min(row, p.M - 1) * p.Cin + s_idx;
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) rs[cb][e] = *(const unsigned short*)(xr + ((c_base + cb * MB + 4 * kq + e) & cmask) * sp);
            }
            bf16* orow = (bf16*)p.out + (((long)to * Ho + (ph * p.fh + db)) * Wo + (pw * p.fw + dd)) * p.Cf + c_base + 4 * kq;
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
                f32x4 v = acc_group(rb, cb, 0) + bias4[cb][0];
                if (p.d2s_residual) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bf2f(__builtin_bit_cast(bf16, rs[cb][e]));
                }
                if (ok) *(bf16x4*)(orow + cb * MB) = pack_bf16x4(v[0], v[1], v[2], v[3]);
            }
        }
    } else {
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
            if (wr * WM + rb * MB >= BM) continue;          // second wave row of a 224-row tile: rows [224, 256) belong to the next tile
            const int row = m0 + wr * WM + rb * MB + lr;
            if (row >= p.M) continue;
            const EpiRow er = epi_row_setup<EPI>(p, row);
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                for (int gq = 0; gq < NG; ++gq) {
                    const int col = n0 + wc * WN + cb * MB + 8 * gq + 4 * kq;
                    epi_store4<EPI>(pe, er, row, col, acc_group(rb, cb, gq), bias4[cb][gq], gate4[cb][gq]);
                }
        }
    }
#ifdef LTX2_V4_PROBE
    if (p.dbg && bid == 0 && tid == 0) ((unsigned long long*)p.dbg)[1] = __builtin_amdgcn_s_memtime() - t_k0;
#endif
}

template <int EPI, int LAYOUT, int BM, bool CONV, int VAR>
__global__ __launch_bounds__(256) void gemm_v4_kernel(const GemmParams p) {
    gemm_v4_tile<EPI, LAYOUT, BM, CONV, VAR>(p, (int)blockIdx.x);
}

template <int EPI, int LAYOUT, int BM, bool CONV = false, int VAR = 0>
int launch_v4(const GemmParams& p, hipStream_t stream) {
    using G = V4Geo<LAYOUT, BM, VAR == 20>;
    // the gated-residual kernel of the 224-row dense bf16 loop parks the first 32 residual rows of every wave above the stage buffers
    // (RESID_PRELOAD in the kernel): + 32 KiB = the CU's whole 160 KiB
    // (round 6) the bf16 / GELU kernels of the dense bf16 loop stage a folded norm's partial sums + row factors above the stage buffers: (GEMM_RF_MAX_PARTS + 1) x BM floats
    constexpr int LDS = G::LDS_BYTES + ((EPI == EPI_RESID_GATE_F32 && LAYOUT == 3 && BM == 224 && !CONV && VAR != 20 && VAR != 9) ? 32768 : 0) +
                        ((LAYOUT == 3 && !CONV && VAR == 30 && (EPI == EPI_BF16 || EPI == EPI_GELU_BF16)) ? (GEMM_RF_MAX_PARTS + 1) * BM * 4 : 0);
    static_assert(LDS <= 160 * 1024, "LDS");
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_v4_kernel<EPI, LAYOUT, BM, CONV, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }
    const int Mt = (p.M + BM - 1) / BM, Nt = LAYOUT == 7 ? 1 : p.N / G::BN;
    hipLaunchKernelGGL((gemm_v4_kernel<EPI, LAYOUT, BM, CONV, VAR>), dim3(Mt * Nt * (p.splitk > 1 ? p.splitk : 1)), dim3(256), LDS, stream, p);
    LTX2_CHECK_LAUNCH("gemm_v4_kernel");
    return LTX2_OK;
}

// out_bf16[m][n] = sum_s part[s][m][n] + bias[n] (+ res[m][n]): the slabs are added in slab order, so the result does not depend
// on which block finished first (deterministic split-K)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, long mn, int N, const float* __restrict__ bias,
                                                            const bf16* __restrict__ res, bf16* __restrict__ out) {
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < mn; i += (long)gridDim.x * blockDim.x * 4) {
        f32x4 a = *(const f32x4*)(part + i);
        for (int s = 1; s < splits; ++s) a += *(const f32x4*)(part + (long)s * mn + i);
        const int n = (int)(i % N);
        if (bias) a += *(const f32x4*)(bias + n);
        if (res) {
            const bf16x4 r = *(const bf16x4*)(res + i);
            a += f32x4{bf2f(r[0]), bf2f(r[1]), bf2f(r[2]), bf2f(r[3])};
        }
        *(bf16x4*)(out + i) = pack_bf16x4(a[0], a[1], a[2], a[3]);
    }
}

}  // namespace

bool gemm_v4_supported(const GemmParams& p, int epilogue, bool conv) {
    if (conv || epilogue == EPI_D2S_BF16) return false;
    if (p.N % 256 != 0 || p.K % 128 != 0 || p.K < 256 || p.M < 1024) return false;
    if ((long)p.M * p.lda * 2 >= (1L << 31) || (long)p.N * p.K * 2 >= (1L << 31)) return false;     // 32-bit buffer offsets
    if (p.ldo % 8 != 0 || ((uintptr_t)p.out & 15)) return false;                                       // 16-byte output rows
    if (epilogue == EPI_ADD_BF16 && (!p.res || (long)p.M * p.ldres * 2 >= (1L << 31) || p.ldres % 8 != 0 || ((uintptr_t)p.res & 15))) return false;      // the residual tile arrives by LDS-DMA
    return true;
}

// Fused V^T output (GemmParams::vt): layout 3, dense, EPI_BF16; V columns start on a tile boundary, a wave's 64 columns stay
// inside one head, and the row tiles reach vt_npad (the zero padding of the last 64-key block is written by the last tile).
bool gemm_v4_vt_supported(const GemmParams& p, int epilogue, int layout) {
    if (epilogue != EPI_BF16 || (layout != 3 && layout != 5) || !p.vt) return false;
    if (!(p.A8 ? gemm_v4_f8_supported(p, epilogue) : p.W8 ? gemm_v4_w8_supported(p, epilogue) : gemm_v4_supported(p, epilogue, false))) return false;
    if (p.vt_col0 % 256 != 0 || (p.vt_hd != 64 && p.vt_hd != 128) || (p.N - p.vt_col0) % p.vt_hd != 0) return false;
    if (p.vt_npad % 64 != 0 || p.vt_npad < p.M) return false;
    const int bm = gemm_v4_prefer_224(p) ? 224 : 256;
    return (long)((p.M + bm - 1) / bm) * bm >= p.vt_npad;
}

bool gemm_v4_w8_supported(const GemmParams& p, int epilogue) {
    if (epilogue != EPI_BF16 && epilogue != EPI_GELU_BF16 && epilogue != EPI_F32 && epilogue != EPI_RESID_GATE_F32) return false;
    if (p.N % 256 != 0 || p.K % 128 != 0 || p.K < 256 || p.M < 1) return false;
    if ((long)p.M * p.lda * 2 >= (1L << 31) || (long)p.N * p.K >= (1L << 31)) return false;
    if (p.ldo % 8 != 0 || ((uintptr_t)p.out & 15) || ((uintptr_t)p.W8 & 15)) return false;
    return true;
}

bool gemm_v4_f8_supported(const GemmParams& p, int epilogue) {
    if (epilogue != EPI_BF16 && epilogue != EPI_GELU_BF16 && epilogue != EPI_F32 && epilogue != EPI_RESID_GATE_F32) return false;
    if (!p.A8 || !p.ascale || !p.W8 || !p.wscale) return false;
    if (p.N % 256 != 0 || p.K % 256 != 0 || p.K < 512 || p.M < 1) return false;        // an even number (>= 4) of 128-element K-tiles
    if ((long)p.M * p.lda >= (1L << 31) || (long)p.N * p.K >= (1L << 31)) return false;
    if (p.lda % 16 != 0 || p.ldo % 8 != 0 || ((uintptr_t)p.out & 15) || ((uintptr_t)p.W8 & 15) || ((uintptr_t)p.A8 & 15)) return false;
    return true;
}

// layout: 3 | 4 | 5 (see the file comment); bm: 0 = pick, 224 | 256 (layouts 3, 5), 448 | 512 (layout 4)
int gemm_v4_launch(const GemmParams& p, int epilogue, hipStream_t stream, int layout, int bm) {
    if (p.A8) {     // fp8 compute: layout 5
        LTX2_CHECK_ARG(gemm_v4_f8_supported(p, epilogue), "gemm_v4: fp8 compute needs A8 + ascale + W8 + wscale, N %% 256 == 0, K %% 256 == 0, K >= 512, a dense bf16/gelu/f32/residual epilogue (N=%d K=%d epilogue=%d)", p.N, p.K, epilogue);
        const bool b224 = bm ? bm == 224 : gemm_v4_prefer_224(p);
#ifndef LTX2_F8_FORCE_L5
#define LTX2_F8_FORCE_L5 0          // (A/B builds: -DLTX2_F8_FORCE_L5=1 keeps every fp8 GEMM on layout 5)
#endif
        const bool l5 = layout == 5 || LTX2_F8_FORCE_L5;
        // 224-row tiles: the 16x16x128 form (layout 6; +16 % FLOP per joule on random e4m3 operands, tools/micro/mfma_fp8_power.hip); layout 5 keeps
        // the 256-row tiles (and the 224-row ones when asked for by `layout`)
#define CASEF(E) \
    case E:      \
        return b224 ? (l5 ? launch_v4<E, 5, 224>(p, stream) : launch_v4<E, 6, 224>(p, stream)) : launch_v4<E, 5, 256>(p, stream);
        switch (epilogue) {
            CASEF(EPI_BF16)
            CASEF(EPI_GELU_BF16)
            CASEF(EPI_F32)
            CASEF(EPI_RESID_GATE_F32)
        }
#undef CASEF
        return LTX2_E_INVALID;
    }
    if (p.W8) {     // fp8-resident weights: layout 3 only
        LTX2_CHECK_ARG(p.wscale && gemm_v4_w8_supported(p, epilogue), "gemm_v4: fp8-resident weights need N %% 256 == 0, K %% 128 == 0, K >= 256, a dense bf16/gelu/f32/residual epilogue (N=%d K=%d epilogue=%d)", p.N, p.K, epilogue);
        const bool b224 = bm ? bm == 224 : gemm_v4_prefer_224(p);
#define CASE8(E) \
    case E:      \
        return b224 ? launch_v4<E, 3, 224, false, 20>(p, stream) : launch_v4<E, 3, 256, false, 20>(p, stream);
        switch (epilogue) {
            CASE8(EPI_BF16)
            CASE8(EPI_GELU_BF16)
            CASE8(EPI_F32)
            CASE8(EPI_RESID_GATE_F32)
        }
#undef CASE8
        return LTX2_E_INVALID;
    }
    if (layout == 4) {
        LTX2_CHECK_ARG(p.N % 128 == 0, "gemm_v4 layout 4: N %% 128");
        const bool b448 = bm == 448;
#define CASE4(E) \
    case E:      \
        return b448 ? launch_v4<E, 4, 448>(p, stream) : launch_v4<E, 4, 512>(p, stream);
        switch (epilogue) {
            CASE4(EPI_BF16)
            CASE4(EPI_ADD_BF16)
            default:
                ltx2_set_error("gemm_v4 layout 4: unsupported epilogue %d", epilogue);
                return LTX2_E_INVALID;
        }
#undef CASE4
    }
    const bool b224 = bm ? bm == 224 : gemm_v4_prefer_224(p);
    LTX2_CHECK_ARG(layout == 3, "gemm_v4: wave layout %d (3 = bf16 dense, 4 = 128-column convs, 5 = fp8 compute)", layout);
    if (p.shadow || p.rf_parts) {      // a folded norm's producer / consumer half: the VAR = 30 instantiations
        switch (epilogue) {
            case EPI_BF16: return b224 ? launch_v4<EPI_BF16, 3, 224, false, 30>(p, stream) : launch_v4<EPI_BF16, 3, 256, false, 30>(p, stream);
            case EPI_GELU_BF16: return b224 ? launch_v4<EPI_GELU_BF16, 3, 224, false, 30>(p, stream) : launch_v4<EPI_GELU_BF16, 3, 256, false, 30>(p, stream);
            case EPI_RESID_GATE_F32: return b224 ? launch_v4<EPI_RESID_GATE_F32, 3, 224, false, 30>(p, stream) : launch_v4<EPI_RESID_GATE_F32, 3, 256, false, 30>(p, stream);
            default:
                ltx2_set_error("gemm_v4: a folded norm on epilogue %d", epilogue);
                return LTX2_E_INVALID;
        }
    }
#define CASE(E) \
    case E:     \
        return b224 ? launch_v4<E, 3, 224>(p, stream) : launch_v4<E, 3, 256>(p, stream);
    switch (epilogue) {
        CASE(EPI_BF16)
        CASE(EPI_GELU_BF16)
        CASE(EPI_SILU_BF16)
        CASE(EPI_F32)
        CASE(EPI_RESID_GATE_F32)
        CASE(EPI_ADD_BF16)
        default:
            ltx2_set_error("gemm_v4: unsupported epilogue %d", epilogue);
            return LTX2_E_INVALID;
    }
#undef CASE
}

// Implicit-GEMM 3x3x3 (or per-frame 3x3) conv over a PADDED channels-last activation volume: p.A = [T+2][H+2][Wd+2][Cin] with
// the padding rule already applied by the producer (replicate in T with `pad_front` leading frames, reflect in H / W), p.T /
// p.H / p.Wd = the OUTPUT extent, M = T*H*Wd, K = 9*taps_t*Cin, weights [Cout][taps][Cin].
bool gemm_v4_conv_supported(const GemmParams& p, int epilogue) {
    if (epilogue != EPI_BF16 && epilogue != EPI_ADD_BF16 && epilogue != EPI_D2S_BF16) return false;
    // depth-to-space scatter: layout 3 only; a wave's 64 columns share the sub-position (Cf >= 64, a power of two); the residual comes from p.res
    if (epilogue == EPI_D2S_BF16 && (p.N % 256 != 0 || p.Cf < 64 || (p.Cf & (p.Cf - 1)) || (p.d2s_residual && (!p.res || p.c_d2s < 1 || (p.c_d2s & (p.c_d2s - 1)))))) return false;
#ifdef LTX2_NO_CONV7        // (same-box A/B builds: tools/ab_build.py noconv7 gemm_v4.hip=-DLTX2_NO_CONV7)
    const bool narrow = false;
#else
    const bool narrow = epilogue == EPI_BF16 && p.N <= 64 && p.N % 8 == 0 && p.N >= 8 && p.ldo == p.N;      // layout 7: one masked 64-column tile
#endif
    if (p.Cin < 128 || (p.Cin & (p.Cin - 1)) || (p.N % 128 != 0 && !narrow)) return false;          // an even number of K-tiles: 27 * Cin / 64
    if (p.taps_t != 3 && p.taps_t != 1) return false;
    if (p.M < 512) return false;
    if ((long)(p.T + 2) * (p.H + 2) * (p.Wd + 2) * p.Cin * 2 >= (1L << 31) || (long)p.N * p.K * 2 >= (1L << 31)) return false;
    if (p.ldo % 8 != 0 || ((uintptr_t)p.out & 15)) return false;
    if (epilogue == EPI_ADD_BF16 && (!p.res || (long)p.M * p.ldres * 2 >= (1L << 31) || p.ldres % 8 != 0 || ((uintptr_t)p.res & 15))) return false;      // the residual tile arrives by LDS-DMA
    return true;
}

// splits for a conv with `tiles` output tiles and nk K-tiles (any count 1..16; the kernel deals even K-tile counts that differ by at most 2):
// the cheapest by a small model in K-tile times -- rounds of the 256 CUs x (the longest split + ~6 for the fp32 tile's way out) + the reduce
// pass (s slabs of mn floats at ~4 TB/s against ~1.4 us per K-tile).  Round 2 took the largest POWER OF TWO that kept the grid in one round:
// 48 tiles x 432 K-tiles (the 1024-channel convs of a 7-frame chunk) ran 4 splits on 192 CUs; 5 splits fill 240.
static int conv_splits(long tiles, int nk, long mn, long ws_bytes, double* cost_out = nullptr) {
    int best = 1;
    double best_cost = (double)((tiles + 255) / 256) * (nk + 6);
    for (int s = 2; s <= 16; ++s) {
        const int h = nk / 2, lo = 2 * (h / s), hi = 2 * ((h + s - 1) / s);
        if (lo < 8 || (long)s * mn * 4 > ws_bytes) break;
        const double cost = (double)((tiles * s + 255) / 256) * (hi + 6) + 3.0 + (double)s * (double)mn / 1.44e6;
        if (cost < best_cost * 0.97) {          // a split must buy at least 3 %
            best_cost = cost;
            best = s;
        }
    }
    if (cost_out) *cost_out = best_cost;
    return best;
}

int gemm_v4_conv_launch(const GemmParams& p_in, int epilogue, hipStream_t stream, void* splitk_ws, long ws_bytes) {
    GemmParams p = p_in;
    LTX2_CHECK_ARG(gemm_v4_conv_supported(p, epilogue), "gemm_v4 conv: unsupported problem (Cin=%d N=%d M=%d epilogue=%d)", p.Cin, p.N, p.M, epilogue);
    p.splitk = 1;
    if (p.N <= 64) return launch_v4<EPI_BF16, 7, 512, true>(p, stream);
    if (p.N % 256 != 0) {
        // 128-channel outputs: BM x 128 tiles, BM in {512, 448, 384} by whole rounds of the 256 CUs x rows per round (the last round of a
        // 512-row grid is often nearly empty: 49 x 128 x 192 positions = 2352 tiles = 9.2 rounds; 448 rows: 10.5 -> 11 x 448 < 10 x 512).
        // A/B on one box (round 3): 448 rows -1 % on the decode; 320 / 384 rows (the XCD's 32 concurrent tiles + the weight panel inside
        // the 4 MB L2) the same time as 512 -- the convs are not bound by the fabric re-reads.
        long best_cost = 0;
        int bm = 512;
        for (const int cand : {512, 448, 384}) {
            const long tiles = ((long)p.M + cand - 1) / cand * (p.N / 128), cost = (tiles + 255) / 256 * cand;
            if (!best_cost || cost < best_cost) {
                best_cost = cost;
                bm = cand;
            }
        }
#define CONV4(E) (bm == 512 ? launch_v4<E, 4, 512, true>(p, stream) : bm == 448 ? launch_v4<E, 4, 448, true>(p, stream) : launch_v4<E, 4, 384, true>(p, stream))
        if (epilogue == EPI_BF16) return CONV4(EPI_BF16);
        return CONV4(EPI_ADD_BF16);
#undef CONV4
    }
    const long t256 = ((long)(p.M + 255) / 256) * (p.N / 256), t224 = ((long)(p.M + 223) / 224) * (p.N / 256);
    bool b224 = (t224 + 255) / 256 * 224 < (t256 + 255) / 256 * 256 || t224 < 256;
    const long mn = (long)p.M * p.N;
    if (epilogue == EPI_D2S_BF16) return b224 ? launch_v4<EPI_D2S_BF16, 3, 224, true>(p, stream) : launch_v4<EPI_D2S_BF16, 3, 256, true>(p, stream);
    int splits = 1;
    if (splitk_ws && p.ldo == p.N) {
        // tile height and split count together: 96 tiles of 224 rows take 2 splits (192 CUs), the same problem as 84 tiles of 256 rows takes 3 (252)
        double c224 = 0, c256 = 0;
        const int s224 = conv_splits(t224, p.K / 64, mn, ws_bytes, &c224), s256 = conv_splits(t256, p.K / 64, mn, ws_bytes, &c256);
        if (s224 > 1 || s256 > 1) b224 = c224 * 224 <= c256 * 256;
        splits = b224 ? s224 : s256;
    }
    if (splits > 1) {
        GemmParams q = p;
        q.splitk = splits;
        q.out = splitk_ws;
        q.bias = nullptr;
        q.res = nullptr;
        const int rc = b224 ? launch_v4<EPI_F32, 3, 224, true>(q, stream) : launch_v4<EPI_F32, 3, 256, true>(q, stream);
        if (rc != LTX2_OK) return rc;
        const long want = (mn / 4 + 255) / 256;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, stream, (const float*)splitk_ws, splits, mn,
                           p.N, p.bias, epilogue == EPI_ADD_BF16 ? p.res : nullptr, (bf16*)p.out);
        LTX2_CHECK_LAUNCH("splitk_reduce_kernel");
        return LTX2_OK;
    }
    if (epilogue == EPI_BF16) return b224 ? launch_v4<EPI_BF16, 3, 224, true>(p, stream) : launch_v4<EPI_BF16, 3, 256, true>(p, stream);
    return b224 ? launch_v4<EPI_ADD_BF16, 3, 224, true>(p, stream) : launch_v4<EPI_ADD_BF16, 3, 256, true>(p, stream);
}

#ifdef LTX2_V4_PROBE
// ablations of the 256-row DiT kernel: var 1 = no DMA in the loop, 2 = no fragment reads, 9 = direct (untransposed) epilogue
int gemm_v4_probe_launch(const GemmParams& p, int layout, int var, hipStream_t stream) {
    (void)layout;
    if (var == 9) return launch_v4<EPI_BF16, 3, 224, false, 9>(p, stream);
    return var == 1 ? launch_v4<EPI_BF16, 3, 256, false, 1>(p, stream) : launch_v4<EPI_BF16, 3, 256, false, 2>(p, stream);
}
#endif
