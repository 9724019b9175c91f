// Fused GEMM / conv3d epilogues shared by the tile kernels (see gemm.h for the epilogue list).
//
// The tile kernels issue mfma(A_op = W fragment, B_op = A fragment), so a 32x32 accumulator holds
// C^T: lane l owns output ROW (l & 31) and, per register group g = r>>2, the 4 CONSECUTIVE columns
// 8g + 4(l>>5) + (r&3).  The epilogue therefore works on float4 column groups: 8-byte bf16x4 or
// 16-byte fp32x4 accesses, one row-bound check per row slot, no per-element branches.
#pragma once
#include "gemm.h"

struct EpiRow {          // per output row (per lane, per row slot)
    int t, h, w;         // conv position (D2S only)
};

template <int EPI>
__device__ __forceinline__ EpiRow epi_row_setup(const GemmParams& p, int row) {
    EpiRow r{};
    if (EPI == EPI_D2S_BF16) {
        const int hw = p.H * p.Wd;
        r.t = row / hw;
        const int r2 = row - r.t * hw;
        r.h = r2 / p.Wd;
        r.w = r2 - r.h * p.Wd;
    }
    return r;
}

__device__ __forceinline__ bf16x4 pack_bf16x4(float a, float b, float c, float d) {
    bf16x4 o = {f2bf(a), f2bf(b), f2bf(c), f2bf(d)};
    return o;
}

// v = 4 raw accumulators for columns col..col+3 of `row`; row < M and col+3 < N guaranteed.
template <int EPI>
// gate_tab4 = p.gate_table[col..col+3] (EPI_RESID_GATE_F32; zeros when there is no table): like the bias it depends on
// the column only, so the caller loads it once per column group instead of once per row.
__device__ __forceinline__ void epi_store4(const GemmParams& p, const EpiRow& er, int row, int col, f32x4 v,
                                           f32x4 bias4, f32x4 gate_tab4 = f32x4{0.f, 0.f, 0.f, 0.f}) {
    v += bias4;
    if (EPI == EPI_BF16) {
        *(bf16x4*)((bf16*)p.out + (long)row * p.ldo + col) = pack_bf16x4(v[0], v[1], v[2], v[3]);
    } else if (EPI == EPI_GELU_BF16) {
        *(bf16x4*)((bf16*)p.out + (long)row * p.ldo + col) =
            pack_bf16x4(gelu_tanh(v[0]), gelu_tanh(v[1]), gelu_tanh(v[2]), gelu_tanh(v[3]));
    } else if (EPI == EPI_SILU_BF16) {
        *(bf16x4*)((bf16*)p.out + (long)row * p.ldo + col) = pack_bf16x4(silu_f(v[0]), silu_f(v[1]), silu_f(v[2]), silu_f(v[3]));
    } else if (EPI == EPI_F32) {
        *(f32x4*)((float*)p.out + (long)row * p.ldo + col) = v;
    } else if (EPI == EPI_RESID_GATE_F32) {
        f32x4 gt = {1.f, 1.f, 1.f, 1.f};
        if (p.gate || p.gate_table) {
            gt = gate_tab4;
            if (p.gate) gt += *(const f32x4*)(p.gate + (long)row * p.gate_stride + col);
        }
        f32x4* o = (f32x4*)((float*)p.out + (long)row * p.ldo + col);
        *o = *o + gt * v;
    } else if (EPI == EPI_ADD_BF16) {
        const bf16x4 rs = *(const bf16x4*)(p.res + (long)row * p.ldres + col);
        *(bf16x4*)((bf16*)p.out + (long)row * p.ldo + col) =
            pack_bf16x4(v[0] + bf2f(rs[0]), v[1] + bf2f(rs[1]), v[2] + bf2f(rs[2]), v[3] + bf2f(rs[3]));
    } else if (EPI == EPI_D2S_BF16) {
        // column n' = s*Cf + c (Cf a power of two >= 4, so the 4 columns share s)
        const int s_idx = col >> p.cf_shift;
        const int c_idx = col & (p.Cf - 1);
        const int da = s_idx / (p.fh * p.fw);
        const int db = (s_idx / p.fw) % p.fh;
        const int dd = s_idx % p.fw;
        const int to = er.t * p.ft + da - p.drop_first;
        if (to < 0) return;
        if (p.d2s_residual) {
            const int sp = p.ft * p.fh * p.fw;
            const bf16* xs = p.res ? p.res : p.A;      // the padded-volume kernel reads its taps from a padded copy: p.res = the plain [M][Cin] input
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bf2f(xs[(long)row * p.Cin + ((c_idx + e) % p.c_d2s) * sp + s_idx]);
        }
        const long opos = ((long)to * (p.H * p.fh) + (er.h * p.fh + db)) * (p.Wd * p.fw) + (er.w * p.fw + dd);
        *(bf16x4*)((bf16*)p.out + opos * p.Cf + c_idx) = pack_bf16x4(v[0], v[1], v[2], v[3]);
    }
}

// Zeros: the LDS-DMA source of out-of-range taps under zero padding.  Long enough for a whole channel run
// (Cin <= 1024 bf16 = 2 KiB) plus the lane's chunk offset, because the iterator below walks it like a real row.
__device__ static const unsigned int ltx2_zero_row[512 + 32] = {0};

// ---- conv A-operand gather -------------------------------------------------------------------------------------
// Conv weights are K-ordered  k = ((kt*3 + kh)*3 + kw)*Cin + c  (for per-frame 3x3 convs kt is absent).  Per output
// row the kernels keep small tap tables (byte offsets of the 3 vertical and 3 horizontal neighbours with the
// padding rule applied) and an iterator that rebuilds the source pointer only when the tap changes; advancing
// inside a channel run is one 64-bit add per row.  (A K order with kt inside (kh, kw) rebuilt pointers less often
// but walked three frames per (kh, kw): fabric reads of the 128-channel convs rose 5.7x -- the frame-major order
// keeps all nine spatial taps of a frame together in L2.)  (PMC, 128->128 conv on the
// 128^2 tile kernel before this: 105 VALU + 86 SALU instructions per K-tile beside 16 MFMAs -- tap decode by integer
// division, reflect / clamp arithmetic and 64-bit multiplies per LDS-DMA issue -- i.e. issue-bound at 29 % MFMA busy.)
// Padding: replicate in T, reflect in H/W (reference simple_decoder.py:105-134); pad_zero = 1: zero padding in all
// three dims (upscaler/spatial.py:44-52); pad_zero = 2: zero padding in H/W with the replicated (causal) temporal
// edge of the VAE encoder (simple_encoder.py:56-75).  0xffffffff marks a zero-padded tap.
struct ConvRow {
    unsigned ho[3], wo[3];      // vertical / horizontal neighbour offsets (bytes), padding rule applied
    int t;                      // frame index
};

__device__ __forceinline__ ConvRow conv_row_setup(const GemmParams& p, int m) {
    const int hw = p.H * p.Wd;
    const int t = m / hw, r2 = m - t * hw, h = r2 / p.Wd, w = r2 - h * p.Wd;
    ConvRow r;
    r.t = t;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int hh = h + k - 1, ww = w + k - 1;
        bool ho_ok = true, wo_ok = true;
        if (p.pad_zero) {           // 1: zero padding in T/H/W; 2: zero padding in H/W, replicate in T
            ho_ok = hh >= 0 && hh < p.H;
            wo_ok = ww >= 0 && ww < p.Wd;
        } else {
            hh = hh < 0 ? -hh : (hh >= p.H ? 2 * p.H - 2 - hh : hh);
            ww = ww < 0 ? -ww : (ww >= p.Wd ? 2 * p.Wd - 2 - ww : ww);
        }
        r.ho[k] = ho_ok ? (unsigned)hh * (unsigned)p.Wd * (unsigned)p.Cin * 2u : 0xffffffffu;
        r.wo[k] = wo_ok ? (unsigned)ww * (unsigned)p.Cin * 2u : 0xffffffffu;
    }
    return r;
}

__device__ __forceinline__ unsigned sel3(const unsigned (&a)[3], int k) { return k == 0 ? a[0] : (k == 1 ? a[1] : a[2]); }

// Walks the K-tiles of NR rows in K order (channel run innermost, then kw, kh, kt).  ct / kw / kh / kt are
// wave-uniform; `chunk_off[j]` = this lane's byte offset of its 16-byte column inside a row (chunk*16).  The
// temporal part of the offset is refreshed only when kt changes (every 9*Cin/64 K-tiles), the (kh, kw) part every
// Cin/64 K-tiles, and a step inside a channel run is one 64-bit add per row.  The launcher requires the
// activation volume to be < 2 GiB, so bit 31 of an offset is set only by the sentinel.
template <int NR>
struct ConvIter {
    int ct, kw, kh, kt;
    unsigned toff[NR];          // clamp(t + kt - pad_front) * frame bytes, or 0xffffffff (zero-padded frame)
    const bf16* ptr[NR];        // source of the current K-tile's piece

    __device__ __forceinline__ void set_t(const GemmParams& p, const ConvRow (&rows)[NR]) {
        const unsigned st = (unsigned)(p.H * p.Wd * p.Cin * 2);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int tt = rows[j].t + kt - p.pad_front;
            const bool oob = p.pad_zero == 1 && (tt < 0 || tt >= p.T);
            toff[j] = oob ? 0xffffffffu : (unsigned)max(0, min(tt, p.T - 1)) * st;
        }
    }
    __device__ __forceinline__ void set_ptr(const GemmParams& p, const ConvRow (&rows)[NR], const unsigned (&chunk_off)[NR]) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const unsigned b = sel3(rows[j].ho, kh), c = sel3(rows[j].wo, kw);
            const bool oob = (int)(b | c | toff[j]) < 0;
            const unsigned long off = (unsigned long)toff[j] + b + c;
            ptr[j] = (const bf16*)(oob ? (const char*)ltx2_zero_row + chunk_off[j] : (const char*)p.A + off + chunk_off[j]);
        }
    }
    __device__ __forceinline__ void init(const GemmParams& p, const ConvRow (&rows)[NR], const unsigned (&chunk_off)[NR]) {
        ct = kw = kh = kt = 0;
        set_t(p, rows);
        set_ptr(p, rows, chunk_off);
    }
    // to the next K-tile: 64 channels further, else the next tap
    __device__ __forceinline__ void next(const GemmParams& p, const ConvRow (&rows)[NR], const unsigned (&chunk_off)[NR]) {
        if (++ct < (p.Cin >> 6)) {
#pragma unroll
            for (int j = 0; j < NR; ++j) ptr[j] += 64;
        } else {
            ct = 0;
            if (++kw == 3) {
                kw = 0;
                if (++kh == 3) {
                    kh = 0;
                    ++kt;
                    set_t(p, rows);
                }
            }
            set_ptr(p, rows, chunk_off);
        }
    }
};

// Launcher of the 256x256 ping-pong kernel (gemm_pp.hip)
int gemm_pp_launch(const GemmParams& p, int epilogue, bool conv, hipStream_t stream);
// Skinny-M kernel (gemm_skinny.hip): M <= 128 rows, weight-streaming-bound, grid cut along N and K cut over the waves.
bool gemm_skinny_supported(const GemmParams& p, int epilogue);
int gemm_skinny_launch(const GemmParams& p, int epilogue, hipStream_t stream);

// 4-wave kernel with the generated asm K loop (gemm_v4.hip): dense, N % 256 == 0, K % 128 == 0, K >= 256.
// layout 0: 1x4 waves, 32x32x16 MFMA; 1: 2x2 waves, 32x32x16; 2: 2x2 waves, 16x16x32; 3: 1x4 waves, 16x16x32 (default);
// 4: BN = 128, 4x1 waves, 16x16x32.  bm: 0 = pick, 224 | 256 (448 | 512 for layout 4).
bool gemm_v4_supported(const GemmParams& p, int epilogue, bool conv);
bool gemm_v4_w8_supported(const GemmParams& p, int epilogue);
bool gemm_v4_vt_supported(const GemmParams& p, int epilogue, int layout);   // p.vt set: V^T written by the epilogue     // p.W8 / p.wscale set: fp8-resident weights
int gemm_v4_launch(const GemmParams& p, int epilogue, hipStream_t stream, int layout, int bm);
// implicit-GEMM conv over a PADDED activation volume (p.A = [T+2][H+2][Wd+2][Cin], padding rule applied by the producer;
// p.T / p.H / p.Wd = output extent): EPI_BF16 / EPI_ADD_BF16, Cin >= 128, Cout % 128 == 0
bool gemm_v4_conv_supported(const GemmParams& p, int epilogue);
// splitk_ws: optional fp32 scratch ([splits][M][N]) that lets small-M / long-K convs split K deterministically
int gemm_v4_conv_launch(const GemmParams& p, int epilogue, hipStream_t stream, void* splitk_ws = nullptr, long ws_bytes = 0);
