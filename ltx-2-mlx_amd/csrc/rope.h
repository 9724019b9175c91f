// SPLIT-RoPE table lookup shared by the row kernels (rowops.hip) and the attention prologue (attention.hip).
//
// The reference builds cos / sin [N][D/2] per prompt (LTX_2_MLX/model/transformer/rope.py:214-328): slot = pad + f * n_dims + d holds
// cos / sin of grid[f] * (2 mid_d(token) / max_pos[d] - 1), identity padding at the front.  A slot therefore depends on ONE axis of
// the token's position, and a latent grid has few distinct coordinates per axis (9 / 16 / 24 at 768x512x65): the COMPACT form keeps
// the cos / sin rows of the U distinct coordinates only -- ct[u][slot] = (cos, sin) the full table holds at `slot` for a token
// whose axis-d(slot) coordinate is the u-th distinct one -- plus idx[d][token] = that u.  393 KB instead of 56 MB at the BASELINE
// shape: it lives in every XCD's L2, so applying RoPE costs no HBM traffic for the tables (round 4).  The values ARE the full
// table's values (the same kernel writes both), so either form gives bit-identical results.
#pragma once
#include "common.h"

struct RopeTab {
    const float* cosb;      // full tables [N][half] (used when ct == null)
    const float* sinb;
    const f32x2* ct;        // compact [U][half] (cos, sin), or null
    const int* idx;         // compact [3][N]: row of ct for the token, per position axis
    int half;               // D / 2 slots per token
    int pad;                // identity slots in front: half - 3 * n_freq (compact form: 3 axes)
    int N;                  // tokens (stride of idx)
    // AXIS-MAJOR compact form (row kernels): cta[(d * U + u) * n_freq + f] = (cos, sin) of slot pad + 3 f + d at the u-th distinct coordinate of
    // axis d -- a token's table is three CONTIGUOUS runs of n_freq entries, which a block loads coalesced and re-reads from LDS in slot order
    const f32x2* cta;
    int U, n_freq;
};

// (cos, sin) of 8 consecutive slots [p0, p0 + 8) of token `row`
__device__ __forceinline__ void rope_cs8(const RopeTab& t, int row, int p0, float (&c)[8], float (&s)[8]) {
    if (t.ct) {
        // slot p0 + e belongs to axis (p0 + e - pad) mod 3 (slots < pad hold the identity in every row); rotate the token's three
        // table rows once so that element e takes r[e % 3]
        const int c3 = ((p0 - t.pad) % 3 + 3) % 3;
        const int i0 = t.idx[row], i1 = t.idx[t.N + row], i2 = t.idx[2 * t.N + row];
        const int r0 = c3 == 0 ? i0 : c3 == 1 ? i1 : i2;
        const int r1 = c3 == 0 ? i1 : c3 == 1 ? i2 : i0;
        const int r2 = c3 == 0 ? i2 : c3 == 1 ? i0 : i1;
        const f32x2* b0 = t.ct + (long)r0 * t.half + p0;
        const f32x2* b1 = t.ct + (long)r1 * t.half + p0;
        const f32x2* b2 = t.ct + (long)r2 * t.half + p0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f32x2 v = (e % 3 == 0 ? b0 : e % 3 == 1 ? b1 : b2)[e];
            c[e] = v[0];
            s[e] = v[1];
        }
    } else {
        const f32x4 c0 = *(const f32x4*)(t.cosb + (long)row * t.half + p0), c1 = *(const f32x4*)(t.cosb + (long)row * t.half + p0 + 4);
        const f32x4 s0 = *(const f32x4*)(t.sinb + (long)row * t.half + p0), s1 = *(const f32x4*)(t.sinb + (long)row * t.half + p0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c[e] = c0[e];
            c[4 + e] = c1[e];
            s[e] = s0[e];
            s[4 + e] = s1[e];
        }
    }
}
