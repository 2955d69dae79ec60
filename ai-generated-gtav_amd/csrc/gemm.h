// Big-M fp16 MFMA GEMM with fused epilogues:  Y[m][n] = sum_k X[m][k] * W[n][k]  (+ epilogue)
// X (activations, logical [M][K]) and W (torch-Linear [N][K]) are both fp16 in the TILE-MAJOR layout of
// common.h `tiled_off` (rows padded to 128, K to 64).  fp16 outputs that feed another GEMM are written tile-major too.
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
    EPI_PARTIAL = 6,    // split-K: out_f32[ks][m][n] = raw partial sums (no bias); the following LayerNorm kernel
                        // reduces the slabs and applies bias + gate + residual (ops.h: LnPending)
    EPI_F16_TILED = 7,  // out_f16 = acc + bias, TILE-MAJOR like the GELU epilogues (no activation): training keeps the MLP's
                        // pre-activation, and the backward pass's activation gradients are GEMM operands themselves
    // ---- LayerNorm fold (docs/LABNOTES.md 4.7): the LayerNorm + adaLN modulate between a residual GEMM and the GEMM that consumes its
    // output is split over the two epilogues instead of being a launch of its own (model/dit.py:19-27,200-225) ----
    EPI_RESID_FOLD = 8,      // producer (out-proj, fc2; full K): x = resid[m][n] += gate (acc + bias) in place, AND the next GEMM's operand
                             // A[m][n] = fp16(x (1 + scale_next + 1e-6)) tile-major, AND per-row partial sums (sum x, sum x^2) per 64-feature slot
    EPI_QKV_FOLD = 9,        // consumers: X = A; y = (acc - mean c1[frame][n]) rstd + c2[frame][n], then as EPI_QKV / EPI_GELU_TANH / EPI_F32
    EPI_GELU_TANH_FOLD = 10, //   with mean / rstd from the producer's partial sums and c1 = sum_k (1 + scale_k) W[n][k],
    EPI_F32_FOLD = 11,       //   c2 = sum_k shift_k W[n][k] + bias[n] from the per-frame tables (gemm_grouped: one launch per forward)
};
constexpr bool epi_is_fold_consumer(int e) { return e == EPI_QKV_FOLD || e == EPI_GELU_TANH_FOLD || e == EPI_F32_FOLD; }
constexpr int epi_base(int e) { return e == EPI_QKV_FOLD ? EPI_QKV : e == EPI_GELU_TANH_FOLD ? EPI_GELU_TANH : e == EPI_F32_FOLD ? EPI_F32 : e; }

enum QkvMode : int {
    QKV_SPATIAL = 0,   // Q,K -> [nb][head][S][64], V -> Vt [nb][head][64][S]   (nb = m / S)
    QKV_TEMPORAL = 1,  // q -> [m][D]; k,v -> kv cache [b][Tmax][P][2][D]
};

struct GemmParams {
    const f16* X;  // tile-major [round_up(M,128)][K]
    int ldx;       // unused (kept for ABI stability of the struct users): K is the logical row length
    const f16* W;  // tile-major [round_up(N,128)][K]
    int M, N, K;   // K % 64 == 0
    int debug;     // -DGTAV_EXPERIMENTS builds only: bit 0 = skip the LDS fills after the prologue, bit 1 = skip LDS reads + MFMA
    unsigned long long* stamps;   // -DGTAV_EXPERIMENTS builds only: per-block timeline stamps (tools/gemm_stamps.py), else null
    int* err_flag; // device error word of the owning handle (bit ERR_F16_SAT is raised when an fp16 output saturated); may be null
    int out_sc1;   // set by launch_gemm: 16-byte output stores bypass-and-drop in L2 (large outputs)
    int splitk;    // EPI_PARTIAL only: number of K slices (grid = tiles * splitk); (K / 64) % splitk == 0
    const float* bias;  // [N] or nullptr
    void* out2;         // EPI_F16_TILED only, optional: a second tile-major f16 image = GELU-tanh of the values written to `out` (training forward: u and h = GELU(u))
    void* out;          // EPI_F32/RESID/PARTIAL: f32 row-major [M][ldo]; EPI_F16: f16 row-major [M][ldo];
                        // EPI_GELU_*: f16 TILE-MAJOR with logical row length ldo (the next GEMM's K)
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
    const float* rope_cs;   // [npos][32][2]: (cos, sin) of rotation pair k at [pos][k] (interleaved-pair RoPE;
                            // cos/sin of features 2k and 2k+1 are equal, rotary_embedding_torch.py:337)
    const float* rope_cs_q; // optional: the table the q features (n < D) rotate by instead — the VAE passes rope_cs scaled by 1/8 log2 e, so that q leaves
                            // the epilogue in the exponent's unit of its flash attention (one fp32 multiply folded into the rotation, no extra rounding)
    // block -> tile map constants of the loader-wave kernels, filled by their launcher (host): the map's three integer divisions
    // by run-time values cost ~110 scalar instructions (two float-reciprocal sequences) = 0.3-0.4 us in front of the first fill;
    // with the divisors' 32-bit reciprocals (a / d == mulhi(a, ceil(2^32 / d)) for a * d < 2^32) they are three s_mul_hi_u32
    struct TileMap { int tiles_m, tiles_n, gn, group, tiles; unsigned rcp_tiles, rcp_group, rcp_gn, rcp_gnlast; } tm;
    // ---- LayerNorm fold ----
    // Tokens are grouped in frames of f_P (a multiple of 16; a wave's token span must not exceed it); the per-frame vectors (gate and
    // f_scale of the producer, f_c1 / f_c2 of the consumer) of frame fr = m / f_P are row (f_rows ? f_rows[fr] : fr) of their table.
    int f_P;
    const int* f_rows;
    // consumer (EPI_*_FOLD)
    const float* f_stats;   // [M][f_nslot][2]: (sum x, sum x^2) over features 64 s .. 64 s + 63 of row m, written by the producer
    int f_nslot;            // K / 64 (a multiple of 4)
    const float* f_c1;      // row r at f_c1 + r * f_ldc: [N]
    const float* f_c2;
    int f_ldc;
    // producer (EPI_RESID_FOLD): out = resid, gate / gate_stride as EPI_RESID (rows by f_P / f_rows)
    float* f_stats_out;     // [M][N / 64][2]
    const float* f_scale;   // scale vectors of the NEXT LayerNorm: row r at f_scale + r * gate_stride
    f16* f_a;               // tile-major [round_up(M, 128)][N]
    // ---- L2 prefetch of the NEXT GEMM's weight by the compute waves of the loader-wave kernels at small M (common.h PrefetchDesc; docs/LABNOTES.md 4.10) ----
    PrefetchDesc pf;
    // ---- persistent 256-token-tile kernel (gemm_p256_kernel, shape 40; docs/LABNOTES.md 4.11).  sk_ws / sk_flags: the caller's split workspace — fp32 partial
    // tiles (GEMM_SK_MAX_SPLIT tiles of 256 x 256 floats) and one int per split tile, ZERO between launches (the kernel resets what it sets); null = whole
    // tiles only.  sk_dp / sk_r are filled by the launcher: tiles [0, sk_dp) run whole (tile t on block t % grid), each of the sk_r remainder tiles is split
    // in two K halves — block 2 i + 1 runs the K tail FIRST in its sequence and hands its partial sums over, block 2 i runs the K head LAST and owns the epilogue ----
    float* sk_ws;
    int* sk_flags;
    int sk_dp, sk_r;
};
constexpr int GEMM_SK_MAX_SPLIT = 128;   // split tiles per launch (half the CUs)
constexpr size_t gemm_sk_ws_bytes() { return (size_t)GEMM_SK_MAX_SPLIT * 256 * 256 * 4; }
constexpr size_t gemm_sk_flag_bytes() { return (size_t)GEMM_SK_MAX_SPLIT * 4; }

// One group of a grouped launch (launch_gemm_grouped): out[m][n] = sum_k X[m][k] W[n][k] + bias[n], m < M (common), n < N.
struct GemmGroup { const f16* X; const f16* W; float* out; const float* bias; int N; int ldo; };
// blockIdx.y = group; every group has the same M and K; out is f32 row-major with leading dimension ldo.  `groups` is a DEVICE array.
int launch_gemm_grouped(const GemmGroup* groups_dev, int n_groups, int max_N, int M, int K, hipStream_t stream);

// Grouped weight-gradient launch (training): out_g[m][n] (f32 row-major, ldo) += sum_k X_g[m][k] W_g[n][k] for up to GEMM_DW_MAX_GROUPS
// independent GEMMs that share the contraction length K (the tokens), as ONE grid of 256 x 256 tiles.  X_g / W_g tile-major [M_g][K] / [N_g][K];
// M_g, N_g multiples of 256.  gemm_dw_grouped_ok: shapes fit and the grouped grid is between half a round and two rounds of tiles.
constexpr int GEMM_DW_MAX_GROUPS = 4;
struct GemmDwGroup { const f16* X; const f16* W; float* out; int M; int N; int ldo; };
bool gemm_dw_grouped_ok(const GemmDwGroup* g, int n, int K);
// tn = true: X / W are the operands themselves, tile-major [K tokens][M | N features] (no transposed copies; K % 128 == 0): gemm.hip mainloop256_tn
int launch_gemm_dw_grouped(const GemmDwGroup* g, int n, int K, int* err_flag, hipStream_t stream, bool tn = false);

// Enqueues the GEMM on `stream`. Returns 0 on success.
int launch_gemm(const GemmParams& p, int epi, hipStream_t stream);
// Temporal QKV projection + causal temporal attention in one launch (gemm.hip: gemm_qkvt_attn_kernel): X rows in the LayerNorm's
// tperm order, W = head-major to_qkv weight; writes the temporal K/V cache (p.k) and the attention output (p.out, f16 tile-major).
bool gemm_qkvt_attn_ok(int M, int D, int S, int Tq, int t0);
int launch_gemm_qkvt_attn(const GemmParams& p, hipStream_t stream);
// Spatial QKV projection + spatial attention in one launch (gemm.hip: gemm_qkvs_attn_kernel; frames of S = 144 tokens): X rows in (b, frame, position) order,
// W = the to_qkv weight in launch_qkv_head_major's mode-1 order, p.rope_cs = the spatial table; writes the attention output (p.out, f16 tile-major) and nothing else.
bool gemm_qkvs_attn_ok(int M, int D, int S);
int launch_gemm_qkvs_attn(const GemmParams& p, hipStream_t stream);
// Weight-gradient GEMM without operand transposes: out[m][n] (f32 row-major, ldo) += sum_t X[t][m] W[t][n]; X, W tile-major fp16
// [tokens][features] (p.M = X features, p.N = W features, p.K = tokens).  M, N multiples of 128, K of 64 (gemm_tn_ok); gemm_tn_pays: also at
// least 128 output tiles (where it beats two operand transposes + the NT kernel).
bool gemm_tn_ok(int M, int N, int K);
bool gemm_tn_pays(int M, int N, int K);
int launch_gemm_tn(const GemmParams& p, hipStream_t stream);
// True when launch_gemm would run this shape on the persistent ping-pong kernel (large M): residual GEMMs then use the in-place
// EPI_RESID epilogue (hidden under the other wave group's main loop) instead of split-K slabs.
bool gemm_pp_ok(int M, int N, int K, int epi);
// True when launch_gemm runs a residual GEMM of this shape on the persistent loader-wave kernel (shape 31, large M), whose in-place gated residual epilogue
// (EPI_RESID: the residual tile is requested at the head of the tile's K loop) replaces slab + LayerNorm reduction.
bool gemm_resid_inplace_ok(int M, int N, int K, int rows_per_gate);
// Split-K factor used for a residual GEMM of this shape (1 = no split): fills the 256 CUs when M is small.
int gemm_choose_splitk(int M, int N, int K);
// Pipeline depth override for experiments (0 = heuristic, else 2 or 4 LDS stages).
void gemm_set_stages(int ns);
// Block shape override (0 = heuristic; shape numbers in gemm.hip launch_epi).  Every shape computes the same result.
void gemm_set_wm(int wm);
#ifdef GTAV_EXPERIMENTS
void gemm_set_debug(int bits);   // timing experiments, WRONG results
// per-block timeline: 8 x u64 per block {s_memtime at entry, first K-tile landed, main loop done, epilogue done (end);
// s_memrealtime at entry and end (100 MHz); XCC id; 0}; buf must hold 8 * grid u64 (null switches it off)
void gemm_set_stamps(unsigned long long* buf, int max_blocks);
#endif

}  // namespace gtav
