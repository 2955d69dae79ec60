// One 16-query tile of the short-sequence spatial attention (S <= 16 NK <= 160 keys), shared by attention.hip's attn_spatial_1p_kernel and gemm.hip's fused
// spatial to_qkv + attention kernel: ONE body, so the two launch forms round identically by construction.
//   Ks  LDS image of the head's keys    [16 NK][128 B], 16-byte chunk c of row r stored at c ^ (r & 7); rows >= S may hold anything finite or not (masked)
//   Vs  LDS image of the head's values^T [64][16 NK + 8] halves; columns S .. 16 NK - 1 must be ZERO (their probabilities are exp2(-inf) = 0, and 0 x NaN is NaN)
//   qf  the lane's query fragment: head features 8 g .. 8 g + 7 and 32 + 8 g .. of query q0 + li   (li = lane & 15, g = lane >> 4)
// Writes softmax(q K^T / 8) V of queries q0 .. q0 + 15 (those < S) into the tile-major fp16 matrix O (logical row length Dm) at rows row0 + li, columns
// col0 .. col0 + 63.  S^T = K Q^T keeps keys on accumulator rows, so P^T feeds the second product straight from the score registers (attention.hip's header).
#pragma once
#include "common.h"
#include "ops.h"

namespace gtav {

template <int NK>
__device__ __forceinline__ void attn_1p_tile(const char* Ks, const char* Vs, const f16x8 (&qf)[2], int S, int q0, f16* __restrict__ O, int row0, int col0, int Dm,
                                             int lane, int sc1) {
    constexpr int S_pad = 16 * NK;
    constexpr int vstride = (S_pad + 8) * 2;      // bytes per Vt row
    const int li = lane & 15, g = lane >> 4;
    // S^T = K Q^T for every key tile: NK independent accumulators
    f32x4 sc[NK];
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
        const int key = kt * 16 + li;
        const char* kr = Ks + key * 128;
        const f16x8 k0 = *(const f16x8*)(kr + (((0 + g) ^ (key & 7)) << 4));
        const f16x8 k1 = *(const f16x8*)(kr + (((4 + g) ^ (key & 7)) << 4));
        sc[kt] = mfma16(k0, qf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        sc[kt] = mfma16(k1, qf[1], sc[kt], 0, 0, 0);
    }
    // padded keys (only the last tiles can hold any) never win the maximum and contribute exp2(-inf) = 0
    float bmax = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (kt * 16 + 15 >= S && kt * 16 + 4 * g + r >= S) sc[kt][r] = -INFINITY;
            bmax = fmaxf(bmax, sc[kt][r]);
        }
    bmax = fmaxf(bmax, __shfl_xor(bmax, 16, 64));
    bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f((sc[kt][r] - bmax) * kAttnQScale);   // raw v_exp_f32: argument <= 0
            sc[kt][r] = pv;
            psum += pv;
        }
    // O^T = Vt P^T, 32 keys per step; P^T straight from the score registers (k-slot j of the B operand <-> key 32 s + 16 (j >> 2) + 4 g + (j & 3))
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < NK / 2; ++s2) {
        f16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf[r] = (f16)sc[2 * s2][r];
            pf[4 + r] = (f16)sc[2 * s2 + 1][r];
        }
        const int kcol = (32 * s2 + 4 * g) * 2;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const char* vr = Vs + (dt * 16 + li) * vstride + kcol;
            union { f16x8 v8; f16x4 v4[2]; } vf;
            vf.v4[0] = *(const f16x4*)(vr);
            vf.v4[1] = *(const f16x4*)(vr + 32);
            o[dt] = mfma16(vf.v8, pf, o[dt], 0, 0, 0);
        }
    }
    float lt = psum + __shfl_xor(psum, 16, 64);
    lt = lt + __shfl_xor(lt, 32, 64);
    const float inv = 1.0f / lt;
    if (q0 + li < S) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f16x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (f16)(o[dt][r] * inv);
            store_f16x4_paired<16>(O + tiled_off(row0 + li, col0 + dt * 16 + 4 * g, Dm), h, lane, sc1);
        }
    }
}

}  // namespace gtav
