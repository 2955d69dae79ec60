// Big-M fp16 MFMA GEMM with fused epilogues:  Y[m][n] = sum_k X[m][k] * W[n][k]  (+ epilogue)
// X: activations [M][ldx] fp16 (K contiguous), W: torch-Linear layout [N][K] fp16 (K contiguous).
#pragma once
#include "common.h"

namespace gtav {

enum GemmEpi : int {
    EPI_F32 = 0,        // out_f32[m][n] = acc + bias
    EPI_F16 = 1,        // out_f16[m][n] = acc + bias
    EPI_GELU_TANH = 2,  // out_f16[m][n] = gelu_tanh(acc + bias)   (DiT Mlp, model/dit.py:161)
    EPI_GELU_ERF = 3,   // out_f16[m][n] = gelu_erf(acc + bias)    (VAE Mlp, model/vae.py:128)
    EPI_RESID = 4,      // resid_f32[m][n] += gate[row(m)][n] * (acc + bias)   (model/dit.py:207-223)
    EPI_QKV = 5,        // bias, RoPE on q/k, scatter to attention layouts (model/attention.py:50-58,109-118)
};

enum QkvMode : int {
    QKV_SPATIAL = 0,   // Q,K -> [nb][head][S][64], V -> Vt [nb][head][64][S]   (nb = m / S)
    QKV_TEMPORAL = 1,  // q -> [m][D]; k,v -> kv cache [b][Tmax][P][2][D]
};

struct GemmParams {
    const f16* X;
    int ldx;
    const f16* W;  // [round_up(N,128)][K]
    int M, N, K;   // K % 64 == 0
    const float* bias;  // [N] or nullptr
    void* out;          // EPI_F32/F16/GELU/RESID target, [M][ldo]
    int ldo;
    // EPI_RESID
    const float* gate;     // nullptr => gate = 1
    int gate_stride;       // floats between consecutive gate rows
    const int* gate_rows;  // optional indirection: row = gate_rows[m / rows_per_gate]
    int rows_per_gate;     // tokens that share one gate vector (P)
    // EPI_QKV
    int qkv_mode;
    f16* q;
    f16* k;
    f16* v;
    int D;      // model width (N == 3*D)
    int S;      // spatial: tokens per attention item; temporal: P tokens per frame
    int Tq;     // temporal: frames carried by this call's tokens
    int t0;     // temporal: window index of the first of those frames
    int Tmax;   // temporal: frames in the kv cache per batch item
    const float* rope_cos;  // [npos][64]
    const float* rope_sin;
};

// Enqueues the GEMM on `stream`. Returns 0 on success.
int launch_gemm(const GemmParams& p, int epi, hipStream_t stream);

}  // namespace gtav
