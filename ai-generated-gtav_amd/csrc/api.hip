// C-ABI of gtav_amd (see include/gtav_amd.h): handles own repacked weights + workspace in HBM and
// enqueue the kernel sequence of each reference entry point on the caller's stream.
#include "api_internal.h"

namespace gtav_shared { thread_local hipEvent_t g_launch_ev[2] = {nullptr, nullptr}; }   // common.h GTAV_LAUNCH: the profiler's event pair for the next launch of this thread
namespace gtav {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }
}  // namespace gtav
// the bf16 twin objects (ops_bf16.h) report through the same thread-local string
namespace gtav_bf16 {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(gtav::g_err, sizeof(gtav::g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return gtav::g_err; }
}  // namespace gtav_bf16

// ---- the two sets of launchers (ops_bf16.h): fp16 operands (default) and their bf16 twins ----
namespace {
const gtav_bf16::GemmParams& bfp(const GemmParams& p) { return reinterpret_cast<const gtav_bf16::GemmParams&>(p); }
const gtav_bf16::LnPending* bfl(const LnPending* p) { return reinterpret_cast<const gtav_bf16::LnPending*>(p); }
int f16_attn_spatial(const f16* Q, const f16* K, const f16* Vt, f16* O, int NB, int heads, int S, hipStream_t st, bool qp) { return launch_attn_spatial(Q, K, Vt, O, NB, heads, S, st, qp); }
int bf_gemm(const GemmParams& p, int epi, hipStream_t st) { return gtav_bf16::launch_gemm(bfp(p), epi, st); }
int bf_ln_modulate(float* x, int ldx, f16* out, int ldo, int M, int D, const float* shift, const float* scale, int mod_stride, const int* rows, int rpm,
                   const LnPending* pend, int* ef, hipStream_t st) {
    return gtav_bf16::launch_ln_modulate(x, ldx, (__bf16*)out, ldo, M, D, shift, scale, mod_stride, rows, rpm, bfl(pend), ef, st);
}
int bf_ln_affine(float* x, int ldx, f16* out, int ldo, int M, int D, const float* gamma, const float* beta, const LnPending* pend, int* ef, hipStream_t st) {
    return gtav_bf16::launch_ln_affine(x, ldx, (__bf16*)out, ldo, M, D, gamma, beta, bfl(pend), ef, st);
}
int bf_patchify(const float* img, const int* fi, int NB, int C, int H, int W, int p, f16* out, int ldo, float a, float b, int* ef, hipStream_t st) {
    return gtav_bf16::launch_patchify(img, fi, NB, C, H, W, p, (__bf16*)out, ldo, a, b, ef, st);
}
int f16_convert_pad(const float* src, int lds, int R, int C, f16* dst, int Rp, int Cp, float scale, int tiled, hipStream_t st, int* ef) {
    return launch_convert_pad_f16(src, lds, R, C, dst, Rp, Cp, scale, tiled, st, ef);
}
int bf_convert_pad(const float* src, int lds, int R, int C, f16* dst, int Rp, int Cp, float scale, int tiled, hipStream_t st, int* ef) {
    return gtav_bf16::launch_convert_pad_f16(src, lds, R, C, (__bf16*)dst, Rp, Cp, scale, tiled, st, ef);
}
int bf_unpad(const f16* src, int lds, int R, int C, float* dst, int tiled, hipStream_t st) { return gtav_bf16::launch_unpad_f16_to_f32((const __bf16*)src, lds, R, C, dst, tiled, st); }
int bf_attn_spatial(const f16* Q, const f16* K, const f16* Vt, f16* O, int NB, int heads, int S, hipStream_t st, bool qp) {
    return gtav_bf16::launch_attn_spatial((const __bf16*)Q, (const __bf16*)K, (const __bf16*)Vt, (__bf16*)O, NB, heads, S, st, qp);
}
int bf_attn_temporal(const f16* q, const f16* kv, f16* O, int B, int P, int D, int Tq, int t0, int Tmax, hipStream_t st) {
    return gtav_bf16::launch_attn_temporal((const __bf16*)q, (const __bf16*)kv, (__bf16*)O, B, P, D, Tq, t0, Tmax, st);
}
const OperandOps OPS_F16 = {launch_gemm, launch_ln_modulate, launch_ln_affine, launch_patchify, f16_convert_pad, launch_unpad_f16_to_f32, f16_attn_spatial,
                            launch_attn_temporal, false};
const OperandOps OPS_BF16 = {bf_gemm, bf_ln_modulate, bf_ln_affine, bf_patchify, bf_convert_pad, bf_unpad, bf_attn_spatial, bf_attn_temporal, true};
}  // namespace
const OperandOps& gtav::operand_ops(bool bf16) { return bf16 ? OPS_BF16 : OPS_F16; }

extern "C" {

const char* gtav_last_error(void) { return gtav::last_error(); }
int gtav_abi_version(void) { return 4; }   // 3: training step (gtav_dit_train_*), collectives (gtav_comm_*); 4: LayerNorm fold switch, optimizer state (gtav_dit_{get,set}_opt_state)

// ------------------------------------------------------------------------------------------------
// elementwise entry points
// ------------------------------------------------------------------------------------------------
int gtav_clamp_frames(float* x, int32_t B, int32_t F, int32_t first, int32_t n, float lo, float hi, void* stream) {
    GTAV_REQUIRE(x && B >= 1 && first >= 0 && first <= F && n >= 1, "clamp_frames: bad arguments");
    if (first == F) return 0;
    GTAV_REQUIRE((int64_t)F * n < (int64_t)1 << 31, "clamp_frames: sample of %d x %d floats too large", F, n);
    return launch_clamp_cols(x, B, F * n, first * n, F * n, lo, hi, (hipStream_t)stream);
}
int gtav_ddim_update(const float* x, const float* v, float* out, int32_t rows, int32_t n, const float* alpha_t,
                     const float* alpha_next, int32_t is_final, void* stream) {
    GTAV_REQUIRE(x && v && out && alpha_t && (alpha_next || is_final), "ddim_update: null argument");
    return launch_ddim_update(x, n, v, n, out, n, rows, n, alpha_t, alpha_next, is_final, (hipStream_t)stream);
}
int gtav_add_noise(const float* x, const float* noise, const float* alpha, float* out, int32_t rows, int32_t n, float clamp_abs,
                   void* stream) {
    return launch_add_noise(x, noise, alpha, out, rows, n, clamp_abs, (hipStream_t)stream);
}
int gtav_vtarget(const float* x, const float* noise, const float* alpha, float* vt, int32_t rows, int32_t n, float clamp_abs,
                 void* stream) {
    return launch_vtarget(x, noise, alpha, vt, rows, n, clamp_abs, (hipStream_t)stream);
}
int gtav_axpy_f32(float* y, const float* x, float alpha, int64_t n, void* stream) {
    GTAV_REQUIRE(y && x && n > 0, "axpy_f32: bad argument");
    return launch_axpy_f32(y, x, alpha, (size_t)n, (hipStream_t)stream);
}
int gtav_mse(const float* a, int64_t a_stride, const float* b, int64_t b_stride, int32_t rows, int32_t n, float* out, void* stream) {
    return launch_mse(a, (size_t)a_stride, b, (size_t)b_stride, rows, n, out, (hipStream_t)stream);
}
int gtav_frames_to_u8(const float* img, uint8_t* out, int32_t N, int32_t H, int32_t W, void* stream) {
    return launch_frames_to_u8(img, out, N, H, W, (hipStream_t)stream);
}
int gtav_moments_to_latents(const float* mom, float* lat, int32_t N, int32_t hw, int32_t latent, int32_t mom_ch, float scale,
                            void* stream) {
    return launch_moments_to_latents(mom, lat, N, hw, latent, mom_ch, scale, (hipStream_t)stream);
}
int gtav_strip_to_frames(const uint8_t* strip, int32_t H, int32_t W, int32_t n_frames, float* out, int32_t OH, int32_t OW, void* stream) {
    return launch_resize_aa(strip, 1, out, n_frames, H, W, OH, OW, (hipStream_t)stream);
}
int gtav_resize_frames(const float* src, float* dst, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream) {
    return launch_resize_aa(src, 0, dst, N, H, W, OH, OW, (hipStream_t)stream);
}
int gtav_latents_to_tokens(const float* lat, float* z, int32_t N, int32_t hw, int32_t latent, void* stream) {
    return launch_latents_to_tokens(lat, z, N, hw, latent, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// kernel-level entry points
// ------------------------------------------------------------------------------------------------
// split workspace of the persistent 256-token-tile kernel for the kernel-level entry points (a handle owns its own): allocated on first use per device —
// these test / tool entry points are never called under stream capture
static int op_sk_workspace(GemmParams& g) {
#ifndef GTAV_EXPERIMENTS
    (void)g;
    return 0;      // (the kernel that splits tiles lives in the experiments build)
#else
    static float* ws[64] = {nullptr};
    static int* flags[64] = {nullptr};
    int dev = 0;
    GTAV_CHECK_HIP(hipGetDevice(&dev));
    dev &= 63;
    if (!ws[dev]) {
        GTAV_CHECK_HIP(hipMalloc((void**)&ws[dev], gemm_sk_ws_bytes()));
        GTAV_CHECK_HIP(hipMalloc((void**)&flags[dev], gemm_sk_flag_bytes()));
        GTAV_CHECK_HIP(hipMemset(flags[dev], 0, gemm_sk_flag_bytes()));
    }
    g.sk_ws = ws[dev];
    g.sk_flags = flags[dev];
    return 0;
#endif
}
int gtav_op_gemm_f16(const void* x, int32_t ldx, const void* w, const float* bias, void* out, int32_t ldo, int32_t M, int32_t N,
                     int32_t K, int32_t epilogue, const float* gate, int32_t gate_stride, int32_t rows_per_gate, void* stream) {
    GTAV_REQUIRE((epilogue >= 0 && epilogue <= 4) || epilogue == EPI_PARTIAL || epilogue == EPI_F16_TILED, "op_gemm_f16: epilogue %d", epilogue);
    GemmParams g;
    memset(&g, 0, sizeof(g));
    RET_IF(op_sk_workspace(g));
    if (epilogue == EPI_PARTIAL) g.splitk = gate_stride > 0 ? gate_stride : 1;  // split-K factor travels in gate_stride
    g.X = (const f16*)x; g.ldx = ldx; g.W = (const f16*)w; g.M = M; g.N = N; g.K = K; g.bias = bias; g.out = out; g.ldo = ldo;
    g.gate = gate; g.gate_stride = gate_stride; g.rows_per_gate = rows_per_gate;
    return launch_gemm(g, epilogue, (hipStream_t)stream);
}
int gtav_op_gemm_qkv(const void* x, int32_t ldx, const void* w, const float* bias, int32_t M, int32_t D, int32_t mode, void* q,
                     void* k, void* v, int32_t S, int32_t Tq, int32_t t0, int32_t Tmax, const float* rope_cs, void* stream) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.X = (const f16*)x; g.ldx = ldx; g.W = (const f16*)w; g.M = M; g.N = 3 * D; g.K = D; g.bias = bias; g.D = D; g.S = S;
    g.qkv_mode = mode; g.q = (f16*)q; g.k = (f16*)k; g.v = (f16*)v; g.Tq = Tq; g.t0 = t0; g.Tmax = Tmax;
    g.rope_cs = rope_cs;
    return launch_gemm(g, EPI_QKV, (hipStream_t)stream);
}
#ifdef GTAV_EXPERIMENTS   // csrc/experiments.h
int gtav_op_gemm_fold_producer(const void* x, const void* w, const float* bias, float* resid, int32_t M, int32_t N, int32_t K, const float* gate,
                               const float* next_scale, int32_t mod_stride, int32_t tokens_per_frame, void* a_out, float* stats_out, void* stream) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.X = (const f16*)x; g.ldx = K; g.W = (const f16*)w; g.M = M; g.N = N; g.K = K; g.bias = bias; g.out = resid; g.ldo = N;
    g.gate = gate; g.gate_stride = mod_stride; g.rows_per_gate = tokens_per_frame;
    g.f_P = tokens_per_frame; g.f_scale = next_scale; g.f_stats_out = stats_out; g.f_a = (f16*)a_out;
    return launch_gemm(g, EPI_RESID_FOLD, (hipStream_t)stream);
}
int gtav_op_gemm_fold_consumer(const void* a, const void* w, int32_t M, int32_t N, int32_t K, int32_t epi, const float* stats, const float* c1, const float* c2,
                               int32_t ldc, int32_t tokens_per_frame, void* out, int32_t ldo, void* stream) {
    GTAV_REQUIRE(epi == EPI_F32 || epi == EPI_GELU_TANH, "gemm_fold_consumer: epilogue %d (0 = f32 row-major, 2 = GELU-tanh fp16 tile-major)", epi);
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.X = (const f16*)a; g.ldx = K; g.W = (const f16*)w; g.M = M; g.N = N; g.K = K; g.out = out; g.ldo = ldo;
    g.f_P = tokens_per_frame; g.f_stats = stats; g.f_nslot = K / 64; g.f_c1 = c1; g.f_c2 = c2; g.f_ldc = ldc;
    return launch_gemm(g, epi == EPI_F32 ? EPI_F32_FOLD : EPI_GELU_TANH_FOLD, (hipStream_t)stream);
}
#endif
int gtav_op_skinny_f32(const float* x, int32_t ldx, const float* w, const float* bias, float* y, int32_t ldy, int32_t M,
                       int32_t N, int32_t K, int32_t act_silu, void* stream) {
    RET_IF(skinny_init());
    return launch_skinny_f32(x, ldx, w, bias, y, ldy, M, N, K, act_silu, (hipStream_t)stream);
}
int gtav_op_ln_modulate(const float* x, void* out, int32_t M, int32_t D, const float* shift, const float* scale,
                        int32_t mod_stride, int32_t rows_per_mod, void* stream) {
    return launch_ln_modulate((float*)x, D, (f16*)out, D, M, D, shift, scale, mod_stride, nullptr, rows_per_mod, nullptr, nullptr, (hipStream_t)stream);
}
int gtav_op_ln_affine(const float* x, void* out, int32_t M, int32_t D, const float* gamma, const float* beta, void* stream) {
    return launch_ln_affine((float*)x, D, (f16*)out, D, M, D, gamma, beta, nullptr, nullptr, (hipStream_t)stream);
}
int gtav_op_attn_spatial(const void* q, const void* k, const void* vt, void* o, int32_t NB, int32_t heads, int32_t S, void* stream) {
    return launch_attn_spatial((const f16*)q, (const f16*)k, (const f16*)vt, (f16*)o, NB, heads, S, (hipStream_t)stream);
}
int gtav_op_attn_temporal(const void* q, const void* kv, void* o, int32_t B, int32_t P, int32_t D, int32_t Tq, int32_t t0,
                          int32_t Tmax, void* stream) {
    return launch_attn_temporal((const f16*)q, (const f16*)kv, (f16*)o, B, P, D, Tq, t0, Tmax, (hipStream_t)stream);
}
int gtav_op_qkv_head_major(const void* w, void* w_hm, int32_t D, void* stream) {
    return launch_qkv_head_major((const f16*)w, (f16*)w_hm, D, (hipStream_t)stream);
}
int gtav_op_gemm_qkvt_attn(const void* x_tperm, const void* w_hm, int32_t M, int32_t D, int32_t P, int32_t Tq, int32_t t0,
                           int32_t Tmax, const float* rope_cs, void* kv, void* o, void* stream) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.X = (const f16*)x_tperm; g.ldx = D; g.W = (const f16*)w_hm; g.M = M; g.N = 3 * D; g.K = D; g.D = D; g.S = P;
    g.qkv_mode = QKV_TEMPORAL; g.k = (f16*)kv; g.v = (f16*)kv; g.out = o; g.ldo = D; g.Tq = Tq; g.t0 = t0; g.Tmax = Tmax;
    g.rope_cs = rope_cs;
    return launch_gemm_qkvt_attn(g, (hipStream_t)stream);
}
int gtav_op_qkv_head_major_spatial(const void* w, void* w_hm, int32_t D, void* stream) {
    return launch_qkv_head_major((const f16*)w, (f16*)w_hm, D, (hipStream_t)stream, 1);
}
int gtav_op_gemm_qkvs_attn(const void* x, const void* w_hm, int32_t M, int32_t D, int32_t P, const float* rope_cs, void* o, void* stream) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.X = (const f16*)x; g.ldx = D; g.W = (const f16*)w_hm; g.M = M; g.N = 3 * D; g.K = D; g.D = D; g.S = P;
    g.qkv_mode = QKV_SPATIAL; g.out = o; g.ldo = D; g.rope_cs = rope_cs;
    return launch_gemm_qkvs_attn(g, (hipStream_t)stream);
}
int gtav_op_attn_spatial_bwd(const void* q, const void* k, const void* vt, const void* d_o, int32_t NB, int32_t heads, int32_t S,
                             const float* rope_cs, void* dqkv, void* stream) {
    return launch_attn_spatial_bwd((const f16*)q, (const f16*)k, (const f16*)vt, (const f16*)d_o, NB, heads, S, heads * 64, rope_cs, (f16*)dqkv, nullptr,
                                   (hipStream_t)stream);
}
int gtav_op_gemm_tn(const void* x, const void* w, int32_t M, int32_t N, int32_t K, float* out, int32_t ldo, void* stream) {
    GemmParams q;
    memset(&q, 0, sizeof(q));
    q.X = (const f16*)x; q.ldx = M; q.W = (const f16*)w; q.M = M; q.N = N; q.K = K; q.out = out; q.ldo = ldo;
    return launch_gemm_tn(q, (hipStream_t)stream);
}
int gtav_op_gemm_dw_grouped(int32_t n, const void* const* x, const void* const* w, float* const* out, const int32_t* M, const int32_t* N, const int32_t* ldo,
                            int32_t K, void* stream) {
    GTAV_REQUIRE(n >= 1 && n <= GEMM_DW_MAX_GROUPS && x && w && out && M && N && ldo, "op_gemm_dw_grouped: 1 .. %d groups", GEMM_DW_MAX_GROUPS);
    GemmDwGroup g[GEMM_DW_MAX_GROUPS];
    for (int i = 0; i < n; ++i) g[i] = GemmDwGroup{(const f16*)x[i], (const f16*)w[i], out[i], M[i], N[i], ldo[i]};
    return launch_gemm_dw_grouped(g, n, K, nullptr, (hipStream_t)stream);
}
int gtav_op_gemm_splitk_ln(const void* x, int32_t ldx, const void* w, const float* bias, int32_t M, int32_t N, int32_t K,
                           int32_t splitk, float* parts, float* resid, const float* gate, int32_t gate_stride,
                           int32_t rows_per_gate, void* out_f16, const float* shift, const float* scale, int32_t mod_stride,
                           void* stream) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.X = (const f16*)x; g.ldx = ldx; g.W = (const f16*)w; g.M = M; g.N = N; g.K = K; g.out = parts; g.ldo = N;
    g.splitk = splitk > 0 ? splitk : gemm_choose_splitk(M, N, K);
    RET_IF(launch_gemm(g, EPI_PARTIAL, (hipStream_t)stream));
    LnPending pd;
    memset(&pd, 0, sizeof(pd));
    pd.parts = parts; pd.nsplit = g.splitk; pd.slab_stride = (size_t)M * N; pd.ld = N; pd.bias = bias; pd.gate = gate;
    pd.gate_stride = gate_stride; pd.rows_per_gate = rows_per_gate;
    return launch_ln_modulate(resid, N, (f16*)out_f16, N, M, N, shift, scale, mod_stride, nullptr, rows_per_gate, &pd, nullptr, (hipStream_t)stream);
}
int gtav_op_rope_interleave(const float* cos_t, const float* sin_t, float* cs, int32_t npos, void* stream) {
    return launch_rope_interleave(cos_t, sin_t, cs, npos, (hipStream_t)stream);
}
int gtav_op_gemm_choose_splitk(int32_t M, int32_t N, int32_t K) { return gemm_choose_splitk(M, N, K); }
int gtav_op_gemm_resid_inplace(int32_t M, int32_t N, int32_t K) { return gemm_resid_inplace_ok(M, N, K, 0) ? 1 : 0; }
void gtav_op_gemm_set_stages(int32_t ns) { gemm_set_stages(ns); }
#ifdef GTAV_EXPERIMENTS
void gtav_op_gemm_set_debug(int32_t bits) { gemm_set_debug(bits); }   // libgtav_amd_exp.so only (csrc/experiments.h)
void gtav_op_gemm_set_stamps(void* buf_dev, int32_t max_blocks) { gemm_set_stamps((unsigned long long*)buf_dev, max_blocks); }
#endif
void gtav_op_gemm_set_wm(int32_t wm) { gemm_set_wm(wm); }

// Calibration of the in-situ profiler (gtav_dit_profile / gtav_vae_profile): `reps` launches of a one-wave kernel that spins `spin_us` microseconds on the
// device's own 100 MHz clock, enqueued back to back, each with an event pair attached to its dispatch exactly like a profiled kernel.  Returns the mean event-pair
// reading and the mean duration the kernel measured itself; their difference is what an attached event pair adds to a kernel's time.
int gtav_timer_calibrate(int32_t spin_us, int32_t reps, double* event_us_mean, double* device_us_mean, void* stream) {
    GTAV_REQUIRE(spin_us >= 0 && spin_us <= 1000 && reps >= 1 && reps <= 256 && event_us_mean && device_us_mean, "timer_calibrate: bad argument");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* dev = nullptr;
    GTAV_CHECK_HIP(hipMalloc(&dev, sizeof(unsigned long long) * reps));
    std::vector<hipEvent_t> ev(2 * reps, nullptr);
    int rc = 0;
    for (int i = 0; i < 2 * reps && !rc; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) { set_error("timer_calibrate: hipEventCreate failed"); rc = 1; }
    if (!rc) rc = launch_calib_spin((unsigned long long)spin_us * 100ull, dev, s);   // warm-up (module load), no events
    if (!rc && hipStreamSynchronize(s) != hipSuccess) { set_error("timer_calibrate: synchronize failed"); rc = 1; }
    for (int i = 0; i < reps && !rc; ++i) {
        g_launch_ev[0] = ev[2 * i];
        g_launch_ev[1] = ev[2 * i + 1];
        rc = launch_calib_spin((unsigned long long)spin_us * 100ull, dev + i, s);
        g_launch_ev[0] = nullptr;
    }
    std::vector<unsigned long long> ticks(reps, 0);
    if (!rc && (hipStreamSynchronize(s) != hipSuccess ||
                hipMemcpy(ticks.data(), dev, sizeof(unsigned long long) * reps, hipMemcpyDeviceToHost) != hipSuccess)) { set_error("timer_calibrate: read-back failed"); rc = 1; }
    double e = 0, d = 0;
    for (int i = 0; i < reps && !rc; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) { set_error("timer_calibrate: hipEventElapsedTime failed"); rc = 1; break; }
        e += ms * 1e3;
        d += (double)ticks[i] * 0.01;
    }
    for (hipEvent_t x : ev) if (x) (void)hipEventDestroy(x);
    (void)hipFree(dev);
    if (rc) return rc;
    *event_us_mean = e / reps;
    *device_us_mean = d / reps;
    return 0;
}

int gtav_op_convert_f16(const float* src, int32_t lds, int32_t R, int32_t C, void* dst, int32_t Rp, int32_t Cp, int32_t tiled,
                        void* stream) {
    return launch_convert_pad_f16(src, lds, R, C, (f16*)dst, Rp, Cp, 1.0f, tiled, (hipStream_t)stream);
}

}  // extern "C"
