// C-ABI of gtav_amd (see include/gtav_amd.h): handles own repacked weights + workspace in HBM and
// enqueue the kernel sequence of each reference entry point on the caller's stream.
#include "../../include/gtav_amd.h"
#include "../../include/gtav_amd_testing.h"
#include "ops.h"
#include "ops_bf16.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

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

using namespace gtav;

#define RET_IF(expr)            \
    do {                        \
        int _rc = (expr);       \
        if (_rc) return _rc;    \
    } while (0)

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
int bf_convert_pad(const float* src, int lds, int R, int C, f16* dst, int Rp, int Cp, float scale, int tiled, hipStream_t st) {
    return gtav_bf16::launch_convert_pad_f16(src, lds, R, C, (__bf16*)dst, Rp, Cp, scale, tiled, st);
}
int bf_unpad(const f16* src, int lds, int R, int C, float* dst, int tiled, hipStream_t st) { return gtav_bf16::launch_unpad_f16_to_f32((const __bf16*)src, lds, R, C, dst, tiled, st); }
int bf_attn_spatial(const f16* Q, const f16* K, const f16* Vt, f16* O, int NB, int heads, int S, hipStream_t st, bool qp) {
    return gtav_bf16::launch_attn_spatial((const __bf16*)Q, (const __bf16*)K, (const __bf16*)Vt, (__bf16*)O, NB, heads, S, st, qp);
}
int bf_attn_temporal(const f16* q, const f16* kv, f16* O, int B, int P, int D, int Tq, int t0, int Tmax, hipStream_t st) {
    return gtav_bf16::launch_attn_temporal((const __bf16*)q, (const __bf16*)kv, (__bf16*)O, B, P, D, Tq, t0, Tmax, st);
}
const OperandOps OPS_F16 = {launch_gemm, launch_ln_modulate, launch_ln_affine, launch_patchify, launch_convert_pad_f16, launch_unpad_f16_to_f32, f16_attn_spatial,
                            launch_attn_temporal, false};
const OperandOps OPS_BF16 = {bf_gemm, bf_ln_modulate, bf_ln_affine, bf_patchify, bf_convert_pad, bf_unpad, bf_attn_spatial, bf_attn_temporal, true};
}  // namespace
const OperandOps& gtav::operand_ops(bool bf16) { return bf16 ? OPS_BF16 : OPS_F16; }

namespace {

// ------------------------------------------------------------------------------------------------
struct Arena {  // owns every device allocation of a handle
    std::vector<void*> ptrs;
    size_t total = 0;
    int alloc(void** out, size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        GTAV_CHECK_HIP(hipMalloc(out, bytes));
        GTAV_CHECK_HIP(hipMemset(*out, 0, bytes));
        ptrs.push_back(*out);
        total += bytes;
        return 0;
    }
    template <typename T>
    int alloc_t(T** out, size_t count) { return alloc((void**)out, count * sizeof(T)); }
    ~Arena() {
        for (void* p : ptrs) (void)hipFree(p);
    }
};

enum SlotKind { SLOT_F16_PAD, SLOT_F32 };
struct Slot {
    SlotKind kind;
    int R, C;        // logical (torch) shape flattened to 2-D
    void* dst;       // f16 [Rp][Cp] or f32 base
    int Rp, Cp;      // padded shape (f16) ; for f32: Cp = destination leading dim
    int c0;          // f32: column offset in destination
    bool set = false;
    bool required = true;
    // training (gtav_dit_train_enable): fp32 master copy (f16 slots; f32 slots train in place), gradient (contiguous [R][C], a
    // slice of the gradient arena), AdamW moments, and for f16 GEMM weights the tile-major copy of the TRANSPOSE (dX = dY W)
    float *master = nullptr, *grad = nullptr, *am = nullptr, *av = nullptr;
    f16* wT = nullptr;
    bool trainable = false;
    // operand type of an f16 slot's device image (common.h "operand type"): the group of layers it belongs to (gtav_dit_set_operand_dtype; -1 = the handle
    // as a whole) and whether the image is bf16.  A type change un-sets the slot: the caller sends the fp32 weight again.
    int group = -1;
    bool bf16 = false;
};

struct WeightTable {
    std::map<std::string, Slot> slots;
    void add_f16(const std::string& n, int R, int C, f16* dst, int Rp, int Cp, int group = -1) {
        slots[n] = Slot{SLOT_F16_PAD, R, C, dst, Rp, Cp, 0, false, true};
        slots[n].group = group;
    }
    // operand type of every f16 slot of `group` (-1: all f16 slots): images of the other type are stale -> the slots count as not set
    int set_dtype(int group, bool bf16) {
        int changed = 0;
        for (auto& kv : slots) {
            Slot& sl = kv.second;
            if (sl.kind != SLOT_F16_PAD || (group >= 0 && sl.group != group) || sl.bf16 == bf16) continue;
            sl.bf16 = bf16;
            sl.set = false;
            ++changed;
        }
        return changed;
    }
    void add_f32(const std::string& n, int R, int C, float* dst, int ldd, int c0 = 0, bool required = true) {
        slots[n] = Slot{SLOT_F32, R, C, dst, R, ldd, c0, false, required};
    }
    int set(const char* name, const float* src, int64_t numel, hipStream_t s) {
        auto it = slots.find(name);
        GTAV_REQUIRE(it != slots.end(), "set_weight: unexpected key '%s'", name);
        Slot& sl = it->second;
        GTAV_REQUIRE(numel == (int64_t)sl.R * sl.C, "set_weight: '%s' has %lld elements, expected %d x %d", name,
                     (long long)numel, sl.R, sl.C);
        if (sl.kind == SLOT_F16_PAD) RET_IF(operand_ops(sl.bf16).convert_pad(src, sl.C, sl.R, sl.C, (f16*)sl.dst, sl.Rp, sl.Cp, 1.0f, 1, s));
        else RET_IF(launch_copy_f32(src, sl.C, sl.R, sl.C, (float*)sl.dst, sl.Cp, sl.c0, s));
        if (sl.master && sl.kind == SLOT_F16_PAD) RET_IF(launch_copy_f32(src, sl.C, sl.R, sl.C, sl.master, sl.C, 0, s));
        if (sl.wT) RET_IF(launch_convert_T_f16(src, sl.C, sl.R, sl.C, sl.wT, s));
        sl.set = true;
        return 0;
    }
    int get(const char* name, float* dst, int64_t numel, hipStream_t s) {
        auto it = slots.find(name);
        GTAV_REQUIRE(it != slots.end(), "get_weight: unknown key '%s'", name);
        Slot& sl = it->second;
        GTAV_REQUIRE(numel == (int64_t)sl.R * sl.C, "get_weight: '%s' size mismatch", name);
        if (sl.kind == SLOT_F16_PAD && sl.master) RET_IF(launch_copy_f32_strided(sl.master, sl.C, sl.R, sl.C, dst, sl.C, s));   // training: the fp32 master
        else if (sl.kind == SLOT_F16_PAD) RET_IF(operand_ops(sl.bf16).unpad((const f16*)sl.dst, sl.Cp, sl.R, sl.C, dst, 1, s));
        else RET_IF(launch_copy_f32_strided((const float*)sl.dst + sl.c0, sl.Cp, sl.R, sl.C, dst, sl.C, s));
        return 0;
    }
    int check_complete() {
        for (auto& kv : slots)
            GTAV_REQUIRE(kv.second.set || !kv.second.required, "finalize: missing weight '%s'", kv.first.c_str());
        return 0;
    }
};

// torch.linspace(start, end, steps) in fp32 (symmetric two-sided evaluation of the CPU kernel)
static std::vector<float> linspace_f32(float start, float end, int steps) {
    std::vector<float> v(steps);
    if (steps == 1) {
        v[0] = start;
        return v;
    }
    const float step = (end - start) / (float)(steps - 1);
    const int half = steps / 2;
    for (int i = 0; i < steps; ++i) v[i] = i < half ? start + step * (float)i : end - step * (float)(steps - i - 1);
    return v;
}

struct RopeTable {
    float* cos_dev = nullptr;
    float* sin_dev = nullptr;
    float* cs_dev = nullptr;   // interleaved (cos, sin) table consumed by the QKV epilogue
    float* csq_dev = nullptr;  // VAE only: cs_dev x 1/8 log2 e, the table the q features rotate by when the flash attention kernel follows (GemmParams::rope_cs_q)
    int npos = 0;
    bool set_cos = false, set_sin = false;
};

static int upload(float* dst, const std::vector<float>& v) {
    GTAV_CHECK_HIP(hipMemcpy(dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// axial "pixel" RoPE table (rotary_embedding_torch.py:290-317): per position (r, c) of a gh x gw grid,
// head dims [0, 2F) rotate with the row angle, [2F, 4F) with the column angle (each freq repeated twice),
// remaining dims are identity.
static void build_axial_table(const std::vector<float>& freqs, int gh, int gw, std::vector<float>& c, std::vector<float>& s) {
    const int F = (int)freqs.size();
    c.assign((size_t)gh * gw * 64, 1.0f);
    s.assign((size_t)gh * gw * 64, 0.0f);
    std::vector<float> ph = linspace_f32(-1.f, 1.f, gh), pw = linspace_f32(-1.f, 1.f, gw);
    for (int r = 0; r < gh; ++r)
        for (int q = 0; q < gw; ++q)
            for (int d = 0; d < 4 * F && d < 64; ++d) {
                const float ang = d < 2 * F ? ph[r] * freqs[d / 2] : pw[q] * freqs[(d - 2 * F) / 2];
                c[((size_t)r * gw + q) * 64 + d] = cosf(ang);
                s[((size_t)r * gw + q) * 64 + d] = sinf(ang);
            }
}

}  // namespace

// In-situ kernel timing (opt-in): HIP events on the launch stream around every kernel of a forward,
// accumulated per kernel class.  Used by bench.py for the roofline line; off in normal operation.
enum ProfClass { PC_LN = 0, PC_QKV, PC_ATTN_S, PC_ATTN_T, PC_OUT, PC_FC1, PC_FC2, PC_OTHER, PC_EMPTY, PC_COUNT };
struct Profiler {
    bool on = false, attached = false;
    std::vector<hipEvent_t> ev;   // pairs
    std::vector<int> cls;
    size_t used = 0;
    double ms[PC_COUNT] = {0};
    long long n[PC_COUNT] = {0};
    int begin(int c, hipStream_t s) {
        if (!on) return 0;
        if (used + 2 > ev.size()) {
            for (int i = 0; i < 2; ++i) {
                hipEvent_t e;
                GTAV_CHECK_HIP(hipEventCreate(&e));
                ev.push_back(e);
            }
        }
        cls.resize(ev.size() / 2);
        cls[used / 2] = c;
        if (c != PC_OTHER && c != PC_EMPTY) {
            // single-kernel classes (GEMMs, LayerNorm, attention): the events ride on the kernel's own dispatch packet
            g_launch_ev[0] = ev[used];
            g_launch_ev[1] = ev[used + 1];
            attached = true;
            return 0;
        }
        attached = false;
        GTAV_CHECK_HIP(hipEventRecord(ev[used], s));
        return 0;
    }
    int end(hipStream_t s) {
        if (!on) return 0;
        if (attached && g_launch_ev[0]) {   // nothing was launched: fall back to a plain pair
            g_launch_ev[0] = nullptr;
            GTAV_CHECK_HIP(hipEventRecord(ev[used], s));
            attached = false;
        }
        if (!attached) GTAV_CHECK_HIP(hipEventRecord(ev[used + 1], s));
        used += 2;
        return 0;
    }
    int collect(hipStream_t s) {
        if (!on || used == 0) return 0;
        GTAV_CHECK_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < used; i += 2) {
            float t = 0.f;
            GTAV_CHECK_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
            ms[cls[i / 2]] += t;
            n[cls[i / 2]] += 1;
        }
        used = 0;
        return 0;
    }
    ~Profiler() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
};
#define PROF(h, c, s, expr)            \
    do {                               \
        RET_IF((h)->prof.begin(c, s)); \
        RET_IF(expr);                  \
        RET_IF((h)->prof.end(s));      \
    } while (0)

// device error word -> message (common.h ERR_*)
static int report_err_flag(int flag, const char* who) {
    GTAV_REQUIRE(!(flag & ERR_TIMESTEP), "%s: a timestep outside [0, 999] was passed", who);
    GTAV_REQUIRE(!(flag & ERR_NONFINITE), "%s: a NaN or inf was found in the input tensor", who);
    GTAV_REQUIRE(!(flag & ERR_F16_SAT), "%s: an activation exceeded the fp16 range (|x| > 65504) and was saturated; results since the "
                 "last check are finite but clipped (the reference runs this path in bf16, which has fp32 range)", who);
    return 0;
}

// ================================================================================================
// DiT
// ================================================================================================
struct gtav_dit {
    Profiler prof;
    gtav_dit_config cfg;
    int D, L, heads, P, gh, gw, C, p, H, W, Hm, Hm_pad, A, Apad, MODW, Kpe, Nfin, maxB, maxT, Mmax, max_rows;
    Arena arena;
    WeightTable wt;
    // fp16 GEMM weights
    f16 *w_pe = nullptr, *w_final = nullptr;
    struct Half { f16 *w_qkv, *w_out, *w_fc1, *w_fc2; float *b_out, *b_fc1, *b_fc2; f16* w_qkv_hm; };   // w_qkv_hm: temporal halves only, head-major rows (fused QKV + attention GEMM), made by finalize
    std::vector<Half> halves;  // [L*2]
    float *b_pe = nullptr, *b_final = nullptr;
    // fp32 conditioning path
    float *w_t0, *b_t0, *w_t2cat, *b_t2, *b_ext, *b_t2a, *w_ada, *b_ada;
    float* sincos = nullptr;  // [1000][256]
    bool sincos_set = false;
    RopeTable rope_s, rope_t;
    std::vector<float> freqs_s, freqs_t;
    float *freqs_s_dev = nullptr, *freqs_t_dev = nullptr;
    // workspace
    f16 *xp, *xn, *qs, *ks, *vts, *qt, *ao, *hbuf;
    std::vector<f16*> kvcache;  // [L]
    float *resid, *fo, *vout, *E, *HC, *Sc, *mod, *parts;
    size_t parts_rows = 0;
    // device error words: [0] the handle's (bad timestep, non-finite input, training-side saturation), [4 + g] one per operand group g — the fp16 stores of group
    // g's kernels raise ERR_F16_SAT THERE, so that gtav_dit_autorange can move exactly the saturated layers to bf16 operands
    int* err_flag = nullptr;
    // operand groups: g = 2 l + (0 spatial | 1 temporal) half of block l, 2 L = patch embedding, 2 L + 1 = final layer.  grp_bf16[g]: the group's 2-byte tensors
    // (LayerNorm output, q / k / v, attention output, MLP hidden, K/V cache, its GEMM weights) are bf16 instead of fp16 (common.h "operand type")
    std::vector<unsigned char> grp_bf16;
    int n_groups = 0;
    bool any_bf16 = false;
    const OperandOps& ops(int g) const { return operand_ops(grp_bf16[g] != 0); }
    int* err_of(int g) const { return err_flag + 4 + g; }
    int* frame_idx = nullptr;   // [maxB * maxT]
    StepParams* step_dev = nullptr;
    int* mod_rows_dev = nullptr;   // [maxB * maxT] rows of the per-frame conditioning table used by the current step
    // prepared steps read the modulation from mod_cur [maxB * maxT][MODW]: slot i holds row mod_rows_dev[i] of the table (gathered per step,
    // only the slots whose row changed: mod_last / mod_changed), so the kernels index it by frame slot without the row indirection
    float* mod_cur = nullptr;
    int *mod_last = nullptr, *mod_changed = nullptr;
    int* t_steps_dev = nullptr;    // [1024]
    struct { bool valid = false, fold_tables = false; int B = 0, F = 0, start = 0, cur = 0, n_steps = 0; const float* actions = nullptr; } prepared;
    // which window the per-layer temporal K/V caches currently describe: written by a full-window (mode 0) sampler step,
    // required by a context-cached (mode 1) step, invalidated by anything else that writes the caches (gtav_dit_forward)
    struct { bool valid = false; int B = 0, F = 0, start = 0, cur = 0; const void* x = nullptr; } kvrec;
    // captured hipGraphs of the fused sampler step, keyed by (shape, mode, buffers)
    struct GraphKey {
        int B, F, T, mode;
        const void *x, *actions, *vout;
        bool operator<(const GraphKey& o) const {
            return std::tie(B, F, T, mode, x, actions, vout) < std::tie(o.B, o.F, o.T, o.mode, o.x, o.actions, o.vout);
        }
    };
    std::map<GraphKey, hipGraphExec_t> graphs;   // nullptr value = shape seen once (eager warm-up done), not yet captured
    bool use_graph = true;
    // window steps at batch 1: temporal QKV projection + temporal attention in one launch (gemm.hip gemm_qkvt_attn_kernel; bit-identical
    // to the split path).  OFF by default: measured 1-2 % SLOWER per forward than the two kernels (profiles/round2/
    // forward_ab_B1_fused_temporal.txt).  gtav_dit_set_fused_temporal() is the switch (it allocates the head-major weight copies);
    // handles with training enabled keep the split path (the copies are not refreshed by the optimizer).
    bool fuse_tattn = false;
    bool w_prefetch = true;   // L2 prefetch of the next GEMM's weight at small M (gemm.h pf_next)
    // per consumer class (0 out-proj, 1 fc1, 2 fc2, 3 to_qkv): 0 skip, 1 the whole slice, k >= 2 the first k K tiles (PrefetchDesc::kt_limit).  The default is the
    // setting that was never slower than no prefetch on any GPU of the round-5 survey (-2 ... -5 % per batch-1 step on every one of them); prefetching every
    // weight whole is 7-12 % faster on some GPUs and 2-16 % slower on others: generate.tune_weight_prefetch finds it where it pays.
    int w_prefetch_cls[4] = {1, 4, 4, 1};
    int resid_inplace_min_m = GTAV_ENV_INT("GTAV_RESID_INPLACE_MIN_M", 1 << 30);   // experiments build only
    // ---- LayerNorm fold (docs/LABNOTES.md 4.7; gemm.h EPI_*_FOLD): the LayerNorm + modulate between a residual GEMM and its consumer runs inside the
    // two GEMM epilogues.  Seam A = out-proj -> fc1, seam B = fc2 -> next to_qkv / final projection.  Per-frame c1 / c2 tables for every
    // consumer: ctab [max_rows][CTW] (built next to the modulation table, one grouped GEMM), ctab_cur [maxB * maxT][CTW] = the rows of the
    // current sampler step (gathered with mod_cur).  Groups are ordered fc1 seams, to_qkv seams, final: a launch over the first n covers a prefix.
    struct Fold {
        bool geom_ok = false;         // tokens per frame % 16 == 0 and >= 64, D % 256 == 0
        bool ok = false;              // ... and the buffers exist (fold_alloc: first gtav_dit_set_fold that enables anything)
        int mode = 1;                 // 0 = never, 1 = heuristic (min_m_a / min_m_b), 2 = every seam at every M (tests)
        // Measured (profiles/round3/fold_v*_ab_B{1,8}.txt, one process per A/B): the folded path is CORRECT (tests/test_gpu_fold.py) but not
        // faster on MI355X at any size tried — B = 8 forward 8.11 ms unfolded, 8.36 ms with seam A folded, 9.03 ms with both; B = 1 2.33 /
        // 2.47 ms — so the default thresholds never fold; gtav_dit_set_fold(h, 1, a, b) / (h, 2, ..) select it (docs/LABNOTES.md 4.7 has the why:
        // the LayerNorm's bytes move into GEMM tails that every resident block reaches at the same time).
        int min_m_a = 1 << 30, min_m_b = 1 << 30;
        int CTW = 0, n_groups = 0, n_groups_a = 0, Rp = 0;
        std::vector<int> col_c;       // column of seam s's c1 in a ctab row (c2 follows at + N_s): s = 2 hb (to_qkv), 2 hb + 1 (fc1), 4 L (final)
        float *ctab = nullptr, *ctab_cur = nullptr, *stats = nullptr;
        f16* sx = nullptr;
        GemmGroup* groups_dev = nullptr;
        int *gcol_dev = nullptr, *gscale_dev = nullptr;
    } fold;
    hipStream_t cap_stream = nullptr;            // private stream the step is captured on (the caller's may be the null stream)
    ~gtav_dit() {
        for (auto& kv : graphs)
            if (kv.second) (void)hipGraphExecDestroy(kv.second);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
    }
    float* ac_table = nullptr;  // alphas_cumprod [1000]
    std::vector<float> ac_host;
    bool finalized = false;
    // ---- training: saved activations of the last training forward, backward workspace, optimizer state ----
    struct Train {
        bool on = false, have_fwd = false, have_actions = false;
        int B = 0, T = 0, M = 0, Mp = 0, rows = 0;
        float loss_scale = 65536.0f;
        float grad_div = 1.0f;              // the arena holds the sum over this many ranks (gtav_dit_set_grad_divisor)
        std::vector<Slot*> params;          // trainable slots in a fixed (sorted-by-name) order
        float* grad_arena = nullptr;        // all gradients, contiguous (one all-reduce); caller-owned when passed to train_enable
        size_t grad_count = 0;
        float* ctl = nullptr;               // [8]: sumsq, step coefficient, skipped steps, grad norm, applied steps, bias corrections
        float *ln_part = nullptr;   // per-(frame, 16-row chunk) partial rows of the fused LayerNorm backward (train.hip ln_mod_bwd_fused_kernel)
        float *red_ws = nullptr, *sumsq_part = nullptr;   // partial sums of the fixed-order reductions (bias gradients, gradient norm)
        AdamParam* adam_params = nullptr;   // device tables of the multi-tensor AdamW launch
        AdamItem* adam_items = nullptr;
        int adam_n_items = 0;
        std::vector<float*> res;            // residual states r_0 .. r_4L, fp32 [M][D]
        struct HB { f16 *xnA, *ao, *y1, *xnB, *u, *hh, *y2, *q, *k, *v; };   // per half-block (spatial: q, k = [nb][head][S][64], v = Vt; temporal: q [M][D], k = kv cache)
        std::vector<HB> hb;
        f16 *xnF = nullptr, *xp = nullptr;
        float *z0 = nullptr, *cpre = nullptr;                   // pre-SiLU values of the conditioning path
        float *dres = nullptr, *dtmp = nullptr, *stats = nullptr, *dmod = nullptr, *dSc = nullptr, *ada_part = nullptr, *dc = nullptr, *dh0 = nullptr, *dz0 = nullptr;
        f16 *g_d = nullptr, *g_d2 = nullptr, *g_h = nullptr, *g_u = nullptr, *g_qkv = nullptr, *dao = nullptr, *tA = nullptr, *tB = nullptr, *dfo = nullptr;
        // grouped weight gradients (launch_gemm_dw_grouped): the transposed operand pairs of a half-block's four dW GEMMs (fc2, fc1, out-proj, QKV)
        // stay alive until its ONE grouped launch; null when the widths are not multiples of 256
        f16 *tAg[4] = {nullptr, nullptr, nullptr, nullptr}, *tBg[4] = {nullptr, nullptr, nullptr, nullptr};
    } tr;
};

static int g_dw_grouped = GTAV_ENV_INT("GTAV_DW_GROUPED", 1);   // experiments build: 0 = one launch per weight gradient (A/B runs)
static int g_fuse_gelu_fwd = GTAV_ENV_INT("GTAV_FUSE_GELU_FWD", 1); // experiments build: 0 = h = GELU(u) by the flat elementwise kernel behind fc1 (A/B runs)
static int g_fuse_gelu = GTAV_ENV_INT("GTAV_FUSE_GELU_BWD", 1); // experiments build: 0 = gelu_bwd and the fc1 bias column sums as two launches (A/B runs)
static int g_fuse_ln = GTAV_ENV_INT("GTAV_FUSE_LN_BWD", 1);     // experiments build: 0 = ln_mod_bwd and frame_reduce_ln as two launches (A/B runs)
static int g_fuse_gate = GTAV_ENV_INT("GTAV_FUSE_GATE", 1);     // experiments build: 0 = gate_bwd, frame_reduce_gate and the bias column sums as three launches (A/B runs)
static int g_dw_tn = GTAV_ENV_INT("GTAV_DW_TN", 1);             // experiments build: 0 = transposed operand copies in front of the grouped launch (A/B runs)

// LayerNorm fold (round 3: correct, measured slower at every size — experiments build only; in the product `fold.ok` stays false and every seam keeps its
// LayerNorm launch): tables, statistics and the grouped-GEMM descriptors, allocated by the first gtav_dit_set_fold that can fold anything
#ifdef GTAV_EXPERIMENTS
static int fold_alloc(gtav_dit* h) {
    gtav_dit::Fold& f = h->fold;
    if (f.ok) return 0;
    GTAV_REQUIRE(f.geom_ok, "dit_set_fold: this geometry has no LayerNorm fold (tokens per frame %d must be a multiple of 16 and >= 64, hidden %% 256 == 0)", h->P);
    Arena& a = h->arena;
    const int D = h->D;
    const size_t Mx = round_up(h->Mmax, 128);
    int rc = 0;
#define A_(expr) do { if (!rc) rc = (expr); } while (0)
    {
        const int nhb = h->L * 2, nseam = 2 * nhb + 1;
        // ctab row: [fc1 seams | to_qkv seams | final], each seam c1 [N] then c2 [N]
        f.col_c.assign(nseam, 0);
        int col = 0;
        for (int hb = 0; hb < nhb; ++hb) { f.col_c[2 * hb + 1] = col; col += 2 * h->Hm; }
        for (int hb = 0; hb < nhb; ++hb) { f.col_c[2 * hb] = col; col += 2 * 3 * D; }
        f.col_c[2 * nhb] = col; col += 2 * h->Nfin;
        f.CTW = col;
        f.n_groups = 2 * nseam; f.n_groups_a = 2 * nhb;
        f.Rp = round_up(h->max_rows, 128);
        A_(a.alloc_t(&f.ctab, (size_t)h->max_rows * f.CTW));
        A_(a.alloc_t(&f.ctab_cur, (size_t)h->maxB * h->maxT * f.CTW));
        A_(a.alloc_t(&f.stats, Mx * (size_t)(D / 64) * 2));
        A_(a.alloc_t(&f.sx, (size_t)f.n_groups * f.Rp * D));
        A_(a.alloc_t(&f.groups_dev, f.n_groups)); A_(a.alloc_t(&f.gcol_dev, f.n_groups)); A_(a.alloc_t(&f.gscale_dev, f.n_groups));
        if (!rc) {
            // group 2 q + kind (kind 0: scale -> c1, kind 1: shift -> c2), q = position of the seam in the ctab row order
            std::vector<GemmGroup> groups(f.n_groups);
            std::vector<int> gcol(f.n_groups), gsc(f.n_groups);
            auto add = [&](int q, int seam, const f16* W, int N, const float* bias, int shift_col, int scale_col) {
                for (int kind = 0; kind < 2; ++kind) {
                    GemmGroup& g = groups[2 * q + kind];
                    g.X = f.sx + (size_t)(2 * q + kind) * f.Rp * D; g.W = W; g.N = N; g.ldo = f.CTW;
                    g.out = f.ctab + f.col_c[seam] + (kind ? N : 0); g.bias = kind ? bias : nullptr;
                    gcol[2 * q + kind] = kind ? shift_col : scale_col; gsc[2 * q + kind] = kind ? 0 : 1;
                }
            };
            for (int hb = 0; hb < nhb; ++hb) {   // chunk order of a half-block's modulation: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
                add(hb, 2 * hb + 1, h->halves[hb].w_fc1, h->Hm, h->halves[hb].b_fc1, (hb * 6 + 3) * D, (hb * 6 + 4) * D);
                add(nhb + hb, 2 * hb, h->halves[hb].w_qkv, 3 * D, nullptr, (hb * 6 + 0) * D, (hb * 6 + 1) * D);
            }
            add(2 * nhb, 2 * nhb, h->w_final, h->Nfin, h->b_final, h->L * 12 * D, h->L * 12 * D + D);
            if (hipMemcpy(f.groups_dev, groups.data(), groups.size() * sizeof(GemmGroup), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(f.gcol_dev, gcol.data(), gcol.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(f.gscale_dev, gsc.data(), gsc.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                set_error("dit_create: upload of the LayerNorm-fold group tables failed");
                rc = 1;
            }
            f.ok = !rc;
        }
    }
#undef A_
    return rc;
}
#endif

// LayerNorm fold: which seams run folded at M tokens (seam A = out-proj -> fc1, seam B = fc2 -> next to_qkv / final projection)
static void fold_policy(const gtav_dit* h, int M, bool& fa, bool& fb) {
    const gtav_dit::Fold& f = h->fold;
    fa = fb = false;
    if (!f.ok || f.mode == 0 || h->tr.on || h->fuse_tattn || h->any_bf16) return;
    fa = f.mode == 2 || M >= f.min_m_a;
    fb = f.mode == 2 || M >= f.min_m_b;
}
// c1 / c2 tables of `rows` rows of the modulation table h->mod (same row numbering): fp16 operands, ONE grouped GEMM over every needed seam
static int dit_fold_tables(gtav_dit* h, int rows, bool fa, bool fb, hipStream_t s) {
    if (!fa && !fb) return 0;
#ifndef GTAV_EXPERIMENTS
    (void)h; (void)rows; (void)s;
    GTAV_REQUIRE(false, "the LayerNorm fold exists only in the experiments build");
#else
    gtav_dit::Fold& f = h->fold;
    const int ng = fb ? f.n_groups : f.n_groups_a;   // (seam B alone still builds the fc1 groups in front of it: never selected by the policy)
    RET_IF(launch_ctab_inputs(h->mod, h->MODW, rows, round_up(rows, 128), h->D, f.gcol_dev, f.gscale_dev, ng, f.sx, (size_t)f.Rp * h->D, s));
    return launch_gemm_grouped(f.groups_dev, ng, h->Hm > 3 * h->D ? h->Hm : 3 * h->D, rows, h->D, s);
#endif
}

static int dit_cond(gtav_dit* h, const int64_t* t64, int rows, int Tq, const StepParams* sp, int use_cur, const float* actions,
                    int64_t act_outer, int64_t act_inner, hipStream_t s) {
    GTAV_REQUIRE(rows <= h->max_rows, "conditioning rows %d exceed max_cond_rows %d", rows, h->max_rows);
    const int ldhc = h->D + h->Apad;
    RET_IF(launch_cond_inputs(t64, rows, Tq, sp, use_cur, h->sincos, h->E, actions, act_outer, act_inner, h->A, h->HC, ldhc,
                              h->D, h->Apad, h->err_flag, s));
    RET_IF(launch_skinny_f32(h->E, 256, h->w_t0, h->b_t0, h->HC, ldhc, rows, h->D, 256, 1, s));
    RET_IF(launch_skinny_f32(h->HC, ldhc, h->w_t2cat, actions ? h->b_t2a : h->b_t2, h->Sc, h->D, rows, h->D, ldhc, 1, s));
    RET_IF(launch_skinny_f32(h->Sc, h->D, h->w_ada, h->b_ada, h->mod, h->MODW, rows, h->MODW, h->D, 0, s));
    bool fa, fb;
    fold_policy(h, rows * h->P, fa, fb);     // one forward over these rows' frames: the LayerNorm-fold tables of the seams it will fold
    return dit_fold_tables(h, rows, fa, fb, s);
}

// x_src: frames of C*H*W floats; frame_index (device, optional) selects the NB = B*Tq frames to process.
// ctab: the c1 / c2 tables of the LayerNorm fold with the row numbering of `mod` (h->fold.ctab beside h->mod, h->fold.ctab_cur beside h->mod_cur).
static int dit_forward_core(gtav_dit* h, const float* x_src, const int* frame_index, int B, int Tq, int t0,
                            const float* mod, const int* mod_rows, const float* ctab, float* v_out, hipStream_t s) {
    const int D = h->D, P = h->P, NB = B * Tq, M = NB * P;
    GTAV_REQUIRE(M <= h->Mmax, "forward: %d tokens exceed workspace (%d)", M, h->Mmax);
    bool fold_a, fold_b;
    fold_policy(h, M, fold_a, fold_b);
    const gtav_dit::Fold& fo = h->fold;
    // consumer side of a folded seam: X = xn holds x (1 + scale), statistics in fo.stats, tables of seam `seam`
    auto fold_consumer = [&](GemmParams& q, int seam, int N) {
        q.bias = nullptr;
        q.f_P = P; q.f_rows = mod_rows; q.f_stats = fo.stats; q.f_nslot = D / 64;
        q.f_c1 = ctab + fo.col_c[seam]; q.f_c2 = q.f_c1 + N; q.f_ldc = fo.CTW;
    };
    // producer side: in-place gated residual update + operand and statistics of the LayerNorm that follows (scale vectors at `next_scale`)
    auto fold_producer = [&](int cls, const f16* X, const f16* Wt, int K, const float* bias, const float* gate, const float* next_scale) -> int {
        GemmParams q;
        memset(&q, 0, sizeof(q));
        q.X = X; q.ldx = K; q.W = Wt; q.M = M; q.N = D; q.K = K; q.out = h->resid; q.ldo = D; q.bias = bias; q.err_flag = h->err_flag;
        q.gate = gate; q.gate_stride = h->MODW; q.gate_rows = mod_rows; q.rows_per_gate = P;
        q.f_P = P; q.f_rows = mod_rows; q.f_scale = next_scale; q.f_stats_out = fo.stats; q.f_a = h->xn;
        PROF(h, cls, s, launch_gemm(q, EPI_RESID_FOLD, s));
        return 0;
    };
    const int g_embed = 2 * h->L, g_final = 2 * h->L + 1;      // operand groups (gtav_dit::grp_bf16)
    // (patchify reports a non-finite input and a finite latent beyond the fp16 range into the embedding group's word: gtav_dit_check folds every word together)
    PROF(h, PC_OTHER, s, h->ops(g_embed).patchify(x_src, frame_index, NB, h->C, h->H, h->W, h->p, h->xp, h->Kpe, 1.f, 0.f, h->err_of(g_embed), s));
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.X = h->xp; g.ldx = h->Kpe; g.W = h->w_pe; g.M = M; g.N = D; g.K = h->Kpe; g.bias = h->b_pe; g.out = h->resid; g.ldo = D;
    PROF(h, PC_OTHER, s, h->ops(g_embed).gemm(g, EPI_F32, s));
    // Residual GEMMs (out-proj, fc2) write split-K partial slabs; the LayerNorm that always follows reduces them and
    // applies bias + gate + residual (LnPending), so the GEMM epilogue has no read-modify-write and small-M
    // launches can spread their K loop over all CUs.
    LnPending pend;
    bool have_pend = false;
    // L2 prefetch of the NEXT GEMM's weight by the loader-wave kernels (gemm.h pf_next): at the few hundred tokens of a batch-1 step every launch
    // otherwise starts on weights that come from HBM
    const int pf_max_m = 1536;   // (above: measured slower, the persistent large-M kernels lose more than their successors gain)
    // (not at the 144 tokens of a context-cached step: those launches are short weight streams themselves, and a second stream beside them cost
    // 1.5 % of the step — profiles/round3/sampler_ab_cached_skinny_shapes_and_prefetch.txt)
    // (Prefetching for to_qkv / fc1 from the LayerNorm launch right in front of them instead — 64 extra blocks beside its row blocks — gained nothing
    // for the consumers and made every LayerNorm 1.6 us longer: profiles/round3/*prefetch_from_layernorm_vs_from_gemm.txt.  The issuing GEMM pays
    // 0.6-0.9 us for its prefetch, the consumer gains 1.5-2 us.)
    static const int pf_min_m = GTAV_ENV_INT("GTAV_PF_MIN_M", 256);   // 320 tokens (window step of the 256 x 256-frame preset): -2.1 %; 144 (cached step): +1.5 %; experiments build: A/B
    const bool pf_on = h->w_prefetch && M >= pf_min_m && M <= pf_max_m;
    // what the GEMM launch at position `pos` of half-block `hb` (launch order: 0 to_qkv, 1 out-proj, 2 fc1, 3 fc2) prefetches: the weight of the next GEMM
    // launch of the step.  (One more launch of lead — the weight of the GEMM after the next — was measured in round 5 and gained nothing on either kind
    // of GPU: profiles/round5/prefetch_box_survey.txt.)
    struct PfNext { const f16* W; int N, K, sk, consumer; };
    auto pf_target = [&](int hb, int pos) -> PfNext {
        const int q = pos + 1, hb2 = hb + q / 4, p2 = q % 4;
        if (!pf_on || hb2 >= 2 * h->L) return PfNext{nullptr, 0, 0, 1, 0};
        const gtav_dit::Half& w2 = h->halves[hb2];
        if (p2 == 0) {
            const bool fused2 = (hb2 & 1) && h->fuse_tattn && !h->tr.on && w2.w_qkv_hm && gemm_qkvt_attn_ok(M, D, P, Tq, t0);
            return PfNext{fused2 ? w2.w_qkv_hm : w2.w_qkv, 3 * D, D, 1, 3};
        }
        if (p2 == 1) return PfNext{w2.w_out, D, D, gemm_choose_splitk(M, D, D), 0};
        if (p2 == 2) return PfNext{w2.w_fc1, h->Hm, D, 1, 1};
        return PfNext{w2.w_fc2, D, h->Hm_pad, gemm_choose_splitk(M, D, h->Hm_pad), 2};
    };
    auto set_pf = [&](GemmParams& q, const PfNext& t) {
        const int v = h->w_prefetch_cls[t.consumer];      // 0 skip, 1 the whole slice, k >= 2: the first k K tiles of every row tile
        if (!pf_on || !t.W || !v) return;
        const int nkt = t.K / 64;
        int skn = t.sk;
        if (skn < 1 || nkt % skn || (skn >= 8 ? skn % 8 : 8 % skn)) skn = 1;
        q.pf = PrefetchDesc{t.W, cdiv(t.N, 128), nkt, skn, v >= 2 ? v : 0};
    };
    auto resid_gemm = [&](const OperandOps& ops, int cls, const f16* X, int ldx, const f16* Wt, int K, const float* bias, const float* gate, const PfNext& pfn) -> int {
        GemmParams q;
        memset(&q, 0, sizeof(q));
        q.X = X; q.ldx = ldx; q.W = Wt; q.M = M; q.N = D; q.K = K; q.out = h->parts; q.ldo = D;
        set_pf(q, pfn);
        q.splitk = gemm_choose_splitk(M, D, K);
        if (gemm_pp_ok(M, D, K, EPI_RESID) || (q.splitk == 1 && M >= h->resid_inplace_min_m) || gemm_resid_inplace_ok(M, D, K, P)) {   // (also on training handles: this plain forward keeps no activations)
            // Large M: gated residual update x += gate * (acc + bias) in the epilogue of the persistent ping-pong GEMM — its
            // read-modify-write hides under the other wave group's main loop, and without split-K slabs the next LayerNorm only
            // reads resid.  (With the one-shot kernels the same epilogue was a loss: B = 8 out-proj 0.77 -> 1.26 ms per forward;
            // resid_inplace_min_m keeps that experiment reachable in the experiments build.)
            q.splitk = 0; q.out = h->resid; q.bias = bias; q.gate = gate; q.gate_stride = h->MODW; q.gate_rows = mod_rows;
            q.rows_per_gate = P;
            PROF(h, cls, s, ops.gemm(q, EPI_RESID, s));
            have_pend = false;
            return 0;
        }
        GTAV_REQUIRE((size_t)q.splitk * M <= h->parts_rows, "split-K slabs exceed workspace");
        PROF(h, cls, s, ops.gemm(q, EPI_PARTIAL, s));
        memset(&pend, 0, sizeof(pend));
        pend.parts = h->parts; pend.nsplit = q.splitk; pend.slab_stride = (size_t)M * D; pend.ld = D; pend.bias = bias;
        pend.gate = gate; pend.gate_stride = h->MODW; pend.gate_rows = mod_rows; pend.rows_per_gate = P;
        have_pend = true;
        return 0;
    };
    bool folded_in = false;   // the LayerNorm in front of the next to_qkv / final projection was folded into the fc2 before it (seam B)
    for (int l = 0; l < h->L; ++l) {
        for (int hf = 0; hf < 2; ++hf) {
            const int hb = l * 2 + hf;
            const gtav_dit::Half& w = h->halves[hb];
            const OperandOps& ops = h->ops(hb);     // this half-block's operand type: every 2-byte tensor below lives and dies inside the half-block
            int* const ef = h->err_of(hb);
            const float* mb = mod + (size_t)hb * 6 * D;
            // temporal half of a batch-1 window step: QKV projection and attention in one launch, on LayerNorm rows written in
            // (b, 16 positions, frame) tile order
            const bool fused_t = hf == 1 && h->fuse_tattn && !h->tr.on && !ops.bf16 && w.w_qkv_hm && gemm_qkvt_attn_ok(M, D, P, Tq, t0);
            if (fused_t) {
                if (!have_pend) memset(&pend, 0, sizeof(pend));   // no slabs (the residual GEMM before updated in place): the descriptor carries the row permutation only
                pend.tperm_T = Tq; pend.tperm_P = P;
            }
            if (!folded_in)
                PROF(h, PC_LN, s, ops.ln_modulate(h->resid, D, h->xn, D, M, D, mb, mb + D, h->MODW, mod_rows, P, (have_pend || fused_t) ? &pend : nullptr, ef, s));
            have_pend = false;
            memset(&g, 0, sizeof(g));
            g.X = h->xn; g.ldx = D; g.W = w.w_qkv; g.M = M; g.N = 3 * D; g.K = D; g.D = D; g.S = P; g.err_flag = ef;
            if (folded_in) fold_consumer(g, 2 * hb, 3 * D);
            set_pf(g, pf_target(hb, 0));
            if (fused_t) {
                g.W = w.w_qkv_hm; g.qkv_mode = QKV_TEMPORAL; g.k = h->kvcache[l]; g.v = h->kvcache[l]; g.out = h->ao; g.ldo = D;
                g.Tq = Tq; g.t0 = t0; g.Tmax = h->maxT; g.rope_cs = h->rope_t.cs_dev;
                PROF(h, PC_QKV, s, launch_gemm_qkvt_attn(g, s));
            } else {
                if (hf == 0) {
                    g.qkv_mode = QKV_SPATIAL; g.q = h->qs; g.k = h->ks; g.v = h->vts;
                    g.rope_cs = h->rope_s.cs_dev;
                } else {
                    g.qkv_mode = QKV_TEMPORAL; g.q = h->qt; g.k = h->kvcache[l]; g.v = h->kvcache[l];
                    g.Tq = Tq; g.t0 = t0; g.Tmax = h->maxT;
                    g.rope_cs = h->rope_t.cs_dev;
                }
                PROF(h, PC_QKV, s, ops.gemm(g, folded_in ? EPI_QKV_FOLD : EPI_QKV, s));
                if (hf == 0) PROF(h, PC_ATTN_S, s, ops.attn_spatial(h->qs, h->ks, h->vts, h->ao, NB, h->heads, P, s, false));
                else PROF(h, PC_ATTN_T, s, ops.attn_temporal(h->qt, h->kvcache[l], h->ao, B, P, D, Tq, t0, h->maxT, s));
            }
            folded_in = false;
            memset(&g, 0, sizeof(g));
            g.X = h->xn; g.ldx = D; g.W = w.w_fc1; g.M = M; g.N = h->Hm; g.K = D; g.bias = w.b_fc1; g.out = h->hbuf; g.ldo = h->Hm_pad; g.err_flag = ef;
            set_pf(g, pf_target(hb, 2));
            if (fold_a) {
                // seam A: out-proj updates the residual in place and emits fc1's operand + row statistics; fc1 normalises in its epilogue
                RET_IF(fold_producer(PC_OUT, h->ao, w.w_out, D, w.b_out, mb + 2 * D, mb + 4 * D));
                fold_consumer(g, 2 * hb + 1, h->Hm);
                PROF(h, PC_FC1, s, launch_gemm(g, EPI_GELU_TANH_FOLD, s));
            } else {
                RET_IF(resid_gemm(ops, PC_OUT, h->ao, D, w.w_out, D, w.b_out, mb + 2 * D, pf_target(hb, 1)));
                PROF(h, PC_LN, s, ops.ln_modulate(h->resid, D, h->xn, D, M, D, mb + 3 * D, mb + 4 * D, h->MODW, mod_rows, P, have_pend ? &pend : nullptr, ef, s));
                have_pend = false;
                PROF(h, PC_FC1, s, ops.gemm(g, EPI_GELU_TANH, s));
            }
            if (fold_b) {
                // seam B: the LayerNorm that follows fc2 is the next half-block's first one (scale_msa) or the final layer's
                const float* next_scale = hb + 1 < 2 * h->L ? mod + (size_t)(hb + 1) * 6 * D + D : mod + (size_t)h->L * 12 * D + D;
                RET_IF(fold_producer(PC_FC2, h->hbuf, w.w_fc2, h->Hm_pad, w.b_fc2, mb + 5 * D, next_scale));
                folded_in = true;
            } else {
                RET_IF(resid_gemm(ops, PC_FC2, h->hbuf, h->Hm_pad, w.w_fc2, h->Hm_pad, w.b_fc2, mb + 5 * D, pf_target(hb, 3)));
            }
        }
    }
    const float* mf = mod + (size_t)h->L * 12 * D;
    if (!folded_in)
        PROF(h, PC_LN, s, h->ops(g_final).ln_modulate(h->resid, D, h->xn, D, M, D, mf, mf + D, h->MODW, mod_rows, P, have_pend ? &pend : nullptr, h->err_of(g_final), s));
    memset(&g, 0, sizeof(g));
    g.X = h->xn; g.ldx = D; g.W = h->w_final; g.M = M; g.N = h->Nfin; g.K = D; g.bias = h->b_final; g.out = h->fo; g.ldo = h->Nfin;
    if (folded_in) fold_consumer(g, 4 * h->L, h->Nfin);
    PROF(h, PC_OTHER, s, h->ops(g_final).gemm(g, folded_in ? EPI_F32_FOLD : EPI_F32, s));
    PROF(h, PC_OTHER, s, launch_unpatchify(h->fo, h->Nfin, v_out, NB, h->C, h->H, h->W, h->p, 0, 1.f, 0.f, s));
    PROF(h, PC_EMPTY, s, 0);   // an event pair around nothing: the per-pair overhead to subtract from every class
    return h->prof.collect(s);
}

extern "C" {

const char* gtav_last_error(void) { return gtav::last_error(); }
int gtav_abi_version(void) { return 4; }   // 3: training step (gtav_dit_train_*), collectives (gtav_comm_*); 4: LayerNorm fold switch, optimizer state (gtav_dit_{get,set}_opt_state)

int gtav_dit_create(const gtav_dit_config* c, gtav_dit** out) {
    GTAV_REQUIRE(c && out, "dit_create: null argument");
    GTAV_REQUIRE(c->hidden_size % 256 == 0 && c->hidden_size <= 2048, "hidden_size=%d must be a multiple of 256, <= 2048", c->hidden_size);
    GTAV_REQUIRE(c->num_heads > 0 && c->hidden_size / c->num_heads == 64 && c->hidden_size % c->num_heads == 0,
                 "only head_dim 64 is implemented (hidden %d, heads %d)", c->hidden_size, c->num_heads);
    GTAV_REQUIRE(c->input_h % c->patch_size == 0 && c->input_w % c->patch_size == 0, "input %dx%d not divisible by patch %d",
                 c->input_h, c->input_w, c->patch_size);
    GTAV_REQUIRE(c->max_frames >= 1 && c->max_frames <= 8, "max_frames=%d must be in [1, 8]", c->max_frames);
    GTAV_REQUIRE(c->max_batch >= 1 && c->depth >= 1, "bad max_batch/depth");
    RET_IF(skinny_init());
    gtav_dit* h = new gtav_dit();
    h->cfg = *c;
    h->D = c->hidden_size; h->L = c->depth; h->heads = c->num_heads; h->C = c->in_channels; h->p = c->patch_size;
    h->H = c->input_h; h->W = c->input_w; h->gh = h->H / h->p; h->gw = h->W / h->p; h->P = h->gh * h->gw;
    const int D = h->D;
    if ((h->P % 8) != 0) {
        set_error("tokens per frame P=%d must be a multiple of 8", h->P);
        delete h;
        return 2;
    }
    h->Hm = (int)(D * c->mlp_ratio); h->Hm_pad = round_up(h->Hm, 128);
    h->A = c->external_cond_dim > 0 ? c->external_cond_dim : 0; h->Apad = round_up(h->A > 0 ? h->A : 1, 32);
    h->MODW = h->L * 12 * D + 2 * D;
    h->Kpe = round_up(h->C * h->p * h->p, 64);
    h->Nfin = h->p * h->p * h->C;
    h->maxB = c->max_batch; h->maxT = c->max_frames; h->Mmax = h->maxB * h->maxT * h->P;
    h->max_rows = c->max_cond_rows > h->maxB * h->maxT ? c->max_cond_rows : h->maxB * h->maxT;
    Arena& a = h->arena;
    WeightTable& wt = h->wt;
    int rc = 0;
#define A_(expr) do { if (!rc) rc = (expr); } while (0)
    A_(a.alloc_t(&h->w_pe, (size_t)round_up(D, 128) * h->Kpe));
    h->n_groups = 2 * h->L + 2;
    h->grp_bf16.assign(h->n_groups, 0);
    wt.add_f16("x_embedder.proj.weight", D, h->C * h->p * h->p, h->w_pe, round_up(D, 128), h->Kpe, 2 * h->L);
    A_(a.alloc_t(&h->b_pe, D)); wt.add_f32("x_embedder.proj.bias", 1, D, h->b_pe, D);
    A_(a.alloc_t(&h->w_t0, (size_t)D * 256)); wt.add_f32("t_embedder.mlp.0.weight", D, 256, h->w_t0, 256);
    A_(a.alloc_t(&h->b_t0, D)); wt.add_f32("t_embedder.mlp.0.bias", 1, D, h->b_t0, D);
    const int ldhc = D + h->Apad;
    A_(a.alloc_t(&h->w_t2cat, (size_t)D * ldhc)); wt.add_f32("t_embedder.mlp.2.weight", D, D, h->w_t2cat, ldhc, 0);
    A_(a.alloc_t(&h->b_t2, D)); wt.add_f32("t_embedder.mlp.2.bias", 1, D, h->b_t2, D);
    A_(a.alloc_t(&h->b_ext, D)); A_(a.alloc_t(&h->b_t2a, D));
    if (h->A > 0) {
        wt.add_f32("external_cond.weight", D, h->A, h->w_t2cat, ldhc, D);
        wt.add_f32("external_cond.bias", 1, D, h->b_ext, D);
    }
    A_(a.alloc_t(&h->w_ada, (size_t)h->MODW * D)); A_(a.alloc_t(&h->b_ada, h->MODW));
    h->halves.resize(h->L * 2);
    for (int l = 0; l < h->L && !rc; ++l)
        for (int hf = 0; hf < 2; ++hf) {
            gtav_dit::Half& w = h->halves[l * 2 + hf];
            char pre[64];
            snprintf(pre, sizeof(pre), "blocks.%d.%c_", l, hf == 0 ? 's' : 't');
            std::string P_(pre);
            const int grp = l * 2 + hf;
            A_(a.alloc_t(&w.w_qkv, (size_t)3 * D * D)); wt.add_f16(P_ + "attn.to_qkv.weight", 3 * D, D, w.w_qkv, 3 * D, D, grp);
            w.w_qkv_hm = nullptr;   // allocated by gtav_dit_set_fused_temporal(h, 1)
            A_(a.alloc_t(&w.w_out, (size_t)D * D)); wt.add_f16(P_ + "attn.to_out.weight", D, D, w.w_out, D, D, grp);
            A_(a.alloc_t(&w.b_out, D)); wt.add_f32(P_ + "attn.to_out.bias", 1, D, w.b_out, D);
            A_(a.alloc_t(&w.w_fc1, (size_t)h->Hm_pad * D)); wt.add_f16(P_ + "mlp.fc1.weight", h->Hm, D, w.w_fc1, h->Hm_pad, D, grp);
            A_(a.alloc_t(&w.b_fc1, h->Hm_pad)); wt.add_f32(P_ + "mlp.fc1.bias", 1, h->Hm, w.b_fc1, h->Hm);
            A_(a.alloc_t(&w.w_fc2, (size_t)D * h->Hm_pad)); wt.add_f16(P_ + "mlp.fc2.weight", D, h->Hm, w.w_fc2, D, h->Hm_pad, grp);
            A_(a.alloc_t(&w.b_fc2, D)); wt.add_f32(P_ + "mlp.fc2.bias", 1, D, w.b_fc2, D);
            const size_t row0 = (size_t)(l * 2 + hf) * 6 * D;
            wt.add_f32(P_ + "adaLN_modulation.1.weight", 6 * D, D, h->w_ada + row0 * D, D);
            wt.add_f32(P_ + "adaLN_modulation.1.bias", 1, 6 * D, h->b_ada + row0, 6 * D);
        }
    A_(a.alloc_t(&h->w_final, (size_t)round_up(h->Nfin, 128) * D));
    wt.add_f16("final_layer.linear.weight", h->Nfin, D, h->w_final, round_up(h->Nfin, 128), D, 2 * h->L + 1);
    A_(a.alloc_t(&h->b_final, round_up(h->Nfin, 128))); wt.add_f32("final_layer.linear.bias", 1, h->Nfin, h->b_final, h->Nfin);
    {
        const size_t row0 = (size_t)h->L * 12 * D;
        wt.add_f32("final_layer.adaLN_modulation.1.weight", 2 * D, D, h->w_ada + row0 * D, D);
        wt.add_f32("final_layer.adaLN_modulation.1.bias", 1, 2 * D, h->b_ada + row0, 2 * D);
    }
    // tables (optional overrides; computed in finalize when absent)
    A_(a.alloc_t(&h->sincos, (size_t)1000 * 256)); wt.add_f32("tables.timestep_sincos", 1000, 256, h->sincos, 256, 0, false);
    h->rope_s.npos = h->P; h->rope_t.npos = h->maxT;
    A_(a.alloc_t(&h->rope_s.cos_dev, (size_t)h->P * 64)); wt.add_f32("tables.rope_spatial_cos", h->P, 64, h->rope_s.cos_dev, 64, 0, false);
    A_(a.alloc_t(&h->rope_s.sin_dev, (size_t)h->P * 64)); wt.add_f32("tables.rope_spatial_sin", h->P, 64, h->rope_s.sin_dev, 64, 0, false);
    A_(a.alloc_t(&h->rope_t.cos_dev, (size_t)h->maxT * 64)); wt.add_f32("tables.rope_temporal_cos", h->maxT, 64, h->rope_t.cos_dev, 64, 0, false);
    A_(a.alloc_t(&h->rope_t.sin_dev, (size_t)h->maxT * 64)); wt.add_f32("tables.rope_temporal_sin", h->maxT, 64, h->rope_t.sin_dev, 64, 0, false);
    A_(a.alloc_t(&h->rope_s.cs_dev, (size_t)h->P * 64)); A_(a.alloc_t(&h->rope_t.cs_dev, (size_t)h->maxT * 64));
    A_(a.alloc_t(&h->freqs_s_dev, 16)); wt.add_f32("spatial_rotary_emb.freqs", 1, 16, h->freqs_s_dev, 16, 0, false);
    A_(a.alloc_t(&h->freqs_t_dev, 32)); wt.add_f32("temporal_rotary_emb.freqs", 1, 32, h->freqs_t_dev, 32, 0, false);
    // workspace
    const size_t Mx = round_up(h->Mmax, 128);   // tile-major A-operands: rows padded to the 128-row tile
    A_(a.alloc_t(&h->xp, Mx * h->Kpe)); A_(a.alloc_t(&h->xn, Mx * D)); A_(a.alloc_t(&h->qs, Mx * D)); A_(a.alloc_t(&h->ks, Mx * D));
    A_(a.alloc_t(&h->vts, Mx * D)); A_(a.alloc_t(&h->qt, Mx * D)); A_(a.alloc_t(&h->ao, Mx * D)); A_(a.alloc_t(&h->hbuf, Mx * h->Hm_pad));
    h->kvcache.resize(h->L);
    for (int l = 0; l < h->L; ++l) A_(a.alloc_t(&h->kvcache[l], Mx * 2 * D));
    A_(a.alloc_t(&h->resid, Mx * D)); A_(a.alloc_t(&h->fo, Mx * h->Nfin));
    A_(a.alloc_t(&h->vout, Mx / h->P * h->C * h->H * h->W));
    // split-K slabs: splitk * M * D floats; gemm_choose_splitk keeps tiles * splitk < 384, i.e. < 384 * 128 * 128 = 6.3 M floats
    h->parts_rows = (2 * Mx * D > (size_t)(8u << 20) ? 2 * Mx * D : (size_t)(8u << 20)) / D;   // two slabs at the largest M
    A_(a.alloc_t(&h->parts, h->parts_rows * D));
    const size_t R = h->max_rows;
    A_(a.alloc_t(&h->E, R * 256)); A_(a.alloc_t(&h->HC, R * ldhc)); A_(a.alloc_t(&h->Sc, R * D)); A_(a.alloc_t(&h->mod, R * h->MODW));
    A_(a.alloc_t(&h->err_flag, 4 + h->n_groups)); A_(a.alloc_t(&h->frame_idx, (size_t)h->maxB * h->maxT)); A_(a.alloc_t(&h->ac_table, 1000));
    A_(a.alloc_t(&h->step_dev, 4)); A_(a.alloc_t(&h->mod_rows_dev, (size_t)h->maxB * h->maxT));
    A_(a.alloc_t(&h->mod_cur, (size_t)h->maxB * h->maxT * h->MODW)); A_(a.alloc_t(&h->mod_last, (size_t)h->maxB * h->maxT)); A_(a.alloc_t(&h->mod_changed, (size_t)h->maxB * h->maxT)); A_(a.alloc_t(&h->t_steps_dev, 1024));
    h->use_graph = GTAV_ENV_INT("GTAV_GRAPH", 1) != 0;   // the shipped library reads no environment: gtav_dit_set_graph() is the switch
    h->fold.geom_ok = h->P % 16 == 0 && h->P >= 64 && D % 256 == 0 && h->Hm % 128 == 0 && h->Nfin % 4 == 0;   // buffers: gtav_dit_set_fold (fold_alloc)
#undef A_
    if (rc) {
        delete h;
        return rc;
    }
    *out = h;
    return 0;
}

void gtav_dit_destroy(gtav_dit* h) { delete h; }

int gtav_dit_set_weight(gtav_dit* h, const char* name, const float* src, int64_t numel, void* stream) {
    GTAV_REQUIRE(h && name && src, "dit_set_weight: null argument");
    std::string n(name);
    // any alias of the two shared rotary freqs parameters
    if (n.size() > 16 && n.compare(n.size() - 16, 16, "rotary_emb.freqs") == 0) {
        const bool spatial = n.rfind("spatial_", 0) == 0 || n.find(".s_attn.") != std::string::npos;
        n = spatial ? "spatial_rotary_emb.freqs" : "temporal_rotary_emb.freqs";
    }
    h->finalized = false;
    return h->wt.set(n.c_str(), src, numel, (hipStream_t)stream);
}

int gtav_dit_get_weight(gtav_dit* h, const char* name, float* dst, int64_t numel, void* stream) {
    GTAV_REQUIRE(h && name && dst, "dit_get_weight: null argument");
    return h->wt.get(name, dst, numel, (hipStream_t)stream);
}

int gtav_dit_finalize(gtav_dit* h, void* stream) {
    GTAV_REQUIRE(h, "dit_finalize: null handle");
    hipStream_t s = (hipStream_t)stream;
    RET_IF(h->wt.check_complete());
    GTAV_CHECK_HIP(hipStreamSynchronize(s));
    const int D = h->D;
    // b_t2a = b_t2 + b_ext (bias of c when actions are given, model/dit.py:363-364)
    RET_IF(launch_add_f32(h->b_t2, h->b_ext, h->b_t2a, D, s));
    // rotary frequencies: loaded values win, otherwise the constructor formulas (dit.py:259-262)
    std::vector<float> fs(16), ft(32);
    if (h->wt.slots["spatial_rotary_emb.freqs"].set) GTAV_CHECK_HIP(hipMemcpy(fs.data(), h->freqs_s_dev, 64, hipMemcpyDeviceToHost));
    else { std::vector<float> l = linspace_f32(1.0f, 128.0f, 16); for (int i = 0; i < 16; ++i) fs[i] = l[i] * (float)M_PI; }
    if (h->wt.slots["temporal_rotary_emb.freqs"].set) GTAV_CHECK_HIP(hipMemcpy(ft.data(), h->freqs_t_dev, 128, hipMemcpyDeviceToHost));
    else for (int i = 0; i < 32; ++i) ft[i] = 1.0f / powf(10000.0f, (float)(2 * i) / 64.0f);
    if (!(h->wt.slots["tables.rope_spatial_cos"].set && h->wt.slots["tables.rope_spatial_sin"].set)) {
        std::vector<float> c, sn;
        build_axial_table(fs, h->gh, h->gw, c, sn);
        RET_IF(upload(h->rope_s.cos_dev, c)); RET_IF(upload(h->rope_s.sin_dev, sn));
    }
    if (!(h->wt.slots["tables.rope_temporal_cos"].set && h->wt.slots["tables.rope_temporal_sin"].set)) {
        std::vector<float> c((size_t)h->maxT * 64), sn((size_t)h->maxT * 64);
        for (int t = 0; t < h->maxT; ++t)
            for (int d = 0; d < 64; ++d) {
                const float ang = (float)t * ft[d / 2];
                c[t * 64 + d] = cosf(ang); sn[t * 64 + d] = sinf(ang);
            }
        RET_IF(upload(h->rope_t.cos_dev, c)); RET_IF(upload(h->rope_t.sin_dev, sn));
    }
    if (!h->wt.slots["tables.timestep_sincos"].set) {
        std::vector<float> tab((size_t)1000 * 256);
        for (int k = 0; k < 128; ++k) {
            const float f = expf(-logf(10000.0f) * (float)k / 128.0f);
            for (int t = 0; t < 1000; ++t) {
                const float arg = (float)t * f;
                tab[(size_t)t * 256 + k] = cosf(arg);
                tab[(size_t)t * 256 + 128 + k] = sinf(arg);
            }
        }
        RET_IF(upload(h->sincos, tab));
    }
    for (auto& w : h->halves)
        if (w.w_qkv_hm) RET_IF(launch_qkv_head_major(w.w_qkv, w.w_qkv_hm, D, s));
    RET_IF(launch_rope_interleave(h->rope_s.cos_dev, h->rope_s.sin_dev, h->rope_s.cs_dev, h->P, s));
    RET_IF(launch_rope_interleave(h->rope_t.cos_dev, h->rope_t.sin_dev, h->rope_t.cs_dev, h->maxT, s));
    GTAV_CHECK_HIP(hipStreamSynchronize(s));
    h->finalized = true;
    return 0;
}

int gtav_dit_forward(gtav_dit* h, const float* x, const int64_t* t, const float* actions, float* out, int32_t B, int32_t T,
                     void* stream) {
    GTAV_REQUIRE(h && x && t && out, "dit_forward: null argument");
    GTAV_REQUIRE(h->finalized, "dit_forward: call gtav_dit_finalize first");
    GTAV_REQUIRE(B >= 1 && B <= h->maxB && T >= 1 && T <= h->maxT, "dit_forward: B=%d T=%d outside capacity (%d, %d)", B, T, h->maxB, h->maxT);
    GTAV_REQUIRE(!actions || h->A > 0, "dit_forward: model has no external_cond");
    hipStream_t s = (hipStream_t)stream;
    // a plain forward overwrites the first B*T rows of the conditioning buffers (mod, E, HC, Sc) and the temporal K/V caches
    // at t0 = 0: a table prepared by gtav_dit_prepare_frame and the context cached by a window step are gone after it
    h->prepared.valid = false;
    h->kvrec.valid = false;
    RET_IF(dit_cond(h, t, B * T, 1, nullptr, 0, actions, h->A, 0, s));
    return dit_forward_core(h, x, nullptr, B, T, 0, h->mod, nullptr, h->fold.ctab, out, s);
}

int gtav_dit_set_schedule(gtav_dit* h, const float* ac, int32_t n) {
    GTAV_REQUIRE(h && ac && n == 1000, "dit_set_schedule: expected 1000 alphas_cumprod values");
    h->ac_host.assign(ac, ac + n);
    GTAV_CHECK_HIP(hipMemcpy(h->ac_table, ac, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// the kernel sequence of one fused sampler step; every step-varying scalar is read from h->step_dev
static int denoise_step_body(gtav_dit* h, float* x, int B, int F, int T, const float* actions, int mode, float* v_out,
                             bool prepared, hipStream_t s) {
    const size_t fsz = (size_t)h->C * h->H * h->W;
    const int Tq = mode == 1 ? 1 : T, t0 = mode == 1 ? T - 1 : 0;
    if (!prepared) RET_IF(dit_cond(h, nullptr, B * Tq, Tq, h->step_dev, mode == 1, actions, (int64_t)F * h->A, h->A, s));
    RET_IF(dit_forward_core(h, x, h->frame_idx, B, Tq, t0, prepared ? h->mod_cur : h->mod, nullptr, prepared ? h->fold.ctab_cur : h->fold.ctab, h->vout, s));
    // DDIM update of frame `cur` (train_dit.py:110-125, generate.py:220)
    const float* vlast = h->vout + (size_t)(Tq - 1) * fsz;
    RET_IF(launch_ddim_update_step(x, F, vlast, (size_t)Tq * fsz, B, (int)fsz, h->step_dev, s));
    if (v_out) RET_IF(launch_copy_rows_f32(vlast, (size_t)Tq * fsz, v_out, fsz, B, fsz, s));
    return 0;
}

int gtav_dit_prepare_frame(gtav_dit* h, int32_t B, int32_t F, int32_t start, int32_t cur, int32_t t_ctx,
                           const int32_t* t_steps_host, int32_t n_steps, const float* actions, void* stream) {
    GTAV_REQUIRE(h && t_steps_host, "prepare_frame: null argument");
    GTAV_REQUIRE(h->finalized, "prepare_frame: finalize the model first");
    const int T = cur - start + 1;
    GTAV_REQUIRE(start >= 0 && cur < F && T >= 1 && T <= h->maxT && B >= 1 && B <= h->maxB && n_steps >= 1 && n_steps <= 1024,
                 "prepare_frame: bad window [%d, %d] / steps %d", start, cur, n_steps);
    GTAV_REQUIRE(!actions || h->A > 0, "prepare_frame: model has no external_cond");
    const int rows = B * (T - 1) + n_steps * B;
    GTAV_REQUIRE(rows <= h->max_rows, "prepare_frame: %d conditioning rows exceed max_cond_rows %d", rows, h->max_rows);
    hipStream_t s = (hipStream_t)stream;
    // the host array may be freed by the caller after this call returns: synchronous copy (once per generated frame)
    GTAV_CHECK_HIP(hipStreamSynchronize(s));
    GTAV_CHECK_HIP(hipMemcpy(h->t_steps_dev, t_steps_host, n_steps * sizeof(int), hipMemcpyHostToDevice));
    const int ldhc = h->D + h->Apad;
    RET_IF(launch_cond_inputs_frame(rows, B, T, F, start, cur, t_ctx, h->t_steps_dev, h->sincos, h->E, actions, h->A, h->HC, ldhc,
                                    h->D, h->Apad, h->err_flag, s));
    RET_IF(launch_skinny_f32(h->E, 256, h->w_t0, h->b_t0, h->HC, ldhc, rows, h->D, 256, 1, s));
    RET_IF(launch_skinny_f32(h->HC, ldhc, h->w_t2cat, actions ? h->b_t2a : h->b_t2, h->Sc, h->D, rows, h->D, ldhc, 1, s));
    RET_IF(launch_skinny_f32(h->Sc, h->D, h->w_ada, h->b_ada, h->mod, h->MODW, rows, h->MODW, h->D, 0, s));
    {   // LayerNorm-fold tables of every row, for the seams a full-window step (B T P tokens) or a context-cached step (B P tokens) folds
        bool fa, fb, fa1, fb1;
        fold_policy(h, B * T * h->P, fa, fb);
        fold_policy(h, B * h->P, fa1, fb1);
        RET_IF(dit_fold_tables(h, rows, fa || fa1, fb || fb1, s));
        h->prepared.fold_tables = fa || fa1 || fb || fb1;
    }
    GTAV_CHECK_HIP(hipMemsetAsync(h->mod_last, 0xFF, (size_t)h->maxB * h->maxT * sizeof(int), s));   // the table changed: every slot of mod_cur is stale
    h->prepared.valid = true; h->prepared.B = B; h->prepared.F = F; h->prepared.start = start; h->prepared.cur = cur;
    h->prepared.n_steps = n_steps; h->prepared.actions = actions;
    return 0;
}

int gtav_dit_denoise_step(gtav_dit* h, float* x, int32_t B, int32_t F, int32_t start, int32_t cur, int32_t t_ctx,
                          int32_t t_cur, int32_t t_next, int32_t is_final, const float* actions, int32_t mode,
                          int32_t cond_step, float* v_out, void* stream) {
    GTAV_REQUIRE(h && x, "denoise_step: null argument");
    GTAV_REQUIRE(h->finalized && !h->ac_host.empty(), "denoise_step: finalize the model and set the schedule first");
    const int T = cur - start + 1;
    GTAV_REQUIRE(start >= 0 && cur < F && T >= 1 && T <= h->maxT && B >= 1 && B <= h->maxB, "denoise_step: bad window [%d, %d] of %d frames", start, cur, F);
    GTAV_REQUIRE(t_cur >= 0 && t_cur < 1000 && t_next >= 0 && t_next < 1000 && t_ctx >= 0 && t_ctx < 1000, "denoise_step: timestep out of range");
    GTAV_REQUIRE(!actions || h->A > 0, "denoise_step: model has no external_cond");
    GTAV_REQUIRE(mode == 0 || mode == 1, "denoise_step: mode %d", mode);
    const bool prepared = cond_step >= 0;
    if (prepared)
        GTAV_REQUIRE(h->prepared.valid && h->prepared.B == B && h->prepared.F == F && h->prepared.start == start &&
                         h->prepared.cur == cur && cond_step < h->prepared.n_steps && h->prepared.actions == actions,
                     "denoise_step: cond_step=%d but gtav_dit_prepare_frame was not called for this window", cond_step);
    else
        h->prepared.valid = false;  // the inline path overwrites the conditioning table
    if (mode == 1) {
        GTAV_REQUIRE(h->kvrec.valid && h->kvrec.B == B && h->kvrec.F == F && h->kvrec.start == start && h->kvrec.cur == cur &&
                         h->kvrec.x == (const void*)x,
                     "denoise_step: context-cached step (mode 1) on window [%d, %d] without a preceding full-window step (mode 0) "
                     "on the same batch / window / latent buffer: the temporal K/V caches would be stale", start, cur);
    } else {
        h->kvrec.valid = true; h->kvrec.B = B; h->kvrec.F = F; h->kvrec.start = start; h->kvrec.cur = cur; h->kvrec.x = x;
    }
    hipStream_t s = (hipStream_t)stream;
    StepParams sp;
    sp.first = start; sp.cur = cur; sp.t_ctx = t_ctx; sp.t_cur = t_cur; sp.is_final = is_final != 0;
    sp.alpha_t = h->ac_host[t_cur]; sp.alpha_next = h->ac_host[t_next]; sp.cond_step = cond_step;
    RET_IF(launch_step_setup(h->step_dev, sp, h->frame_idx, h->mod_rows_dev, prepared ? h->mod_last : nullptr, h->mod_changed, B, mode == 1 ? 1 : T, T, F,
                             mode == 1, s));
    if (prepared) RET_IF(launch_gather_rows(h->mod, h->mod_rows_dev, h->mod_changed, h->mod_cur, B * (mode == 1 ? 1 : T), h->MODW,
                                            h->prepared.fold_tables ? h->fold.ctab : nullptr, h->fold.ctab_cur, h->fold.CTW, s));
    if (!h->use_graph || h->prof.on) return denoise_step_body(h, x, B, F, T, actions, mode, v_out, prepared, s);

    // hipGraph path: the first step of a new (shape, buffers) key runs eagerly (warm-up: lazy module load, function
    // attributes), the second one is captured, later ones replay the captured graph (~240 kernel nodes, one launch).
    gtav_dit::GraphKey key{B, F, T, mode * 2 + (prepared ? 1 : 0), x, actions, v_out};
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        if (h->graphs.size() > 64) {
            for (auto& kv : h->graphs)
                if (kv.second) (void)hipGraphExecDestroy(kv.second);
            h->graphs.clear();
        }
        h->graphs[key] = nullptr;
        return denoise_step_body(h, x, B, F, T, actions, mode, v_out, prepared, s);
    }
    if (!it->second) {
        // capture on a private non-blocking stream (stream capture is not permitted on the legacy null stream, which is
        // what torch hands out by default); nothing executes during capture, the graph is launched on the caller's stream
        if (!h->cap_stream && hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) {
            h->use_graph = false;
            return denoise_step_body(h, x, B, F, T, actions, mode, v_out, prepared, s);
        }
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            h->use_graph = false;
            return denoise_step_body(h, x, B, F, T, actions, mode, v_out, prepared, s);
        }
        const int rc = denoise_step_body(h, x, B, F, T, actions, mode, v_out, prepared, h->cap_stream);
        const hipError_t ce = hipStreamEndCapture(h->cap_stream, &graph);
        if (rc || ce != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            h->use_graph = false;  // capture is not available here: fall back to eager launches for good
            if (rc) return rc;
            return denoise_step_body(h, x, B, F, T, actions, mode, v_out, prepared, s);
        }
        hipGraphExec_t exec = nullptr;
        const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess || !exec) {
            h->use_graph = false;
            return denoise_step_body(h, x, B, F, T, actions, mode, v_out, prepared, s);
        }
        it->second = exec;
    }
    GTAV_CHECK_HIP(hipGraphLaunch(it->second, s));
    return 0;
}

int gtav_dit_set_graph(gtav_dit* h, int32_t enable) {
    GTAV_REQUIRE(h, "dit_set_graph: null handle");
    h->use_graph = enable != 0;
    return 0;
}

int gtav_dit_set_weight_prefetch(gtav_dit* h, int32_t enable) {
    GTAV_REQUIRE(h, "dit_set_weight_prefetch: null handle");
    GTAV_REQUIRE(enable == 0 || enable == 1 || (enable >> 16) == 1,
                 "dit_set_weight_prefetch: mode %d (0 off, 1 on, 0x10000 | per-class nibbles: bits 0-3 out-proj's weight, 4-7 fc1's, 8-11 fc2's, 12-15 to_qkv's; "
                 "nibble 0 = not prefetched, 1 = the whole slice, k >= 2 = the first k K tiles of every row tile)", enable);
    int cls[4];
    for (int c = 0; c < 4; ++c) cls[c] = enable == 0 ? 0 : enable == 1 ? 1 : (enable >> (4 * c)) & 15;
    const bool on = cls[0] || cls[1] || cls[2] || cls[3];
    if (h->w_prefetch != on || memcmp(cls, h->w_prefetch_cls, sizeof(cls))) {   // captured sampler steps carry the other kernel parameters
        for (auto& kv : h->graphs)
            if (kv.second) (void)hipGraphExecDestroy(kv.second);
        h->graphs.clear();
    }
    h->w_prefetch = on;
    memcpy(h->w_prefetch_cls, cls, sizeof(cls));
    return 0;
}

#ifdef GTAV_EXPERIMENTS   // csrc/experiments.h
int gtav_dit_set_fold(gtav_dit* h, int32_t mode, int32_t min_tokens_a, int32_t min_tokens_b) {
    GTAV_REQUIRE(h && mode >= 0 && mode <= 2, "dit_set_fold: mode %d", mode);
    if (min_tokens_a >= 0) h->fold.min_m_a = min_tokens_a;
    if (min_tokens_b >= 0) h->fold.min_m_b = min_tokens_b;
    if (mode == 2 || (mode == 1 && (h->fold.min_m_a < (1 << 30) || h->fold.min_m_b < (1 << 30)))) RET_IF(fold_alloc(h));
    for (auto& kv : h->graphs)       // captured sampler steps contain the other kernel sequence
        if (kv.second) (void)hipGraphExecDestroy(kv.second);
    h->graphs.clear();
    h->prepared.valid = false;       // the per-frame tables were built for the old policy
    h->fold.mode = mode;
    return 0;
}
#endif

int gtav_dit_set_fused_temporal(gtav_dit* h, int32_t enable) {
    GTAV_REQUIRE(h, "dit_set_fused_temporal: null handle");
    if (h->fuse_tattn != (enable != 0)) {   // captured sampler steps contain the other kernel sequence
        for (auto& kv : h->graphs)
            if (kv.second) (void)hipGraphExecDestroy(kv.second);
        h->graphs.clear();
    }
    if (enable && h->P % 16 == 0 && h->D % 256 == 0 && h->maxT >= 5) {
        // first enable: head-major copies of the temporal to_qkv weights (3 D^2 halves per block); filled here if the weights are
        // already final, otherwise by gtav_dit_finalize
        for (int l = 0; l < h->L; ++l) {
            gtav_dit::Half& w = h->halves[l * 2 + 1];
            if (w.w_qkv_hm) continue;
            RET_IF(h->arena.alloc_t(&w.w_qkv_hm, (size_t)3 * h->D * h->D));
            if (h->finalized) RET_IF(launch_qkv_head_major(w.w_qkv, w.w_qkv_hm, h->D, nullptr));
        }
        if (h->finalized) GTAV_CHECK_HIP(hipDeviceSynchronize());
    }
    h->fuse_tattn = enable != 0;
    return 0;
}

int gtav_dit_profile(gtav_dit* h, int32_t enable) {
    GTAV_REQUIRE(h, "dit_profile: null handle");
    h->prof.on = enable != 0;
    h->prof.used = 0;
    for (int i = 0; i < PC_COUNT; ++i) { h->prof.ms[i] = 0; h->prof.n[i] = 0; }
    return 0;
}
int gtav_dit_profile_read(gtav_dit* h, double* ms_by_class, int64_t* launches_by_class) {
    GTAV_REQUIRE(h && ms_by_class && launches_by_class, "dit_profile_read: null argument");
    for (int i = 0; i < PC_COUNT; ++i) { ms_by_class[i] = h->prof.ms[i]; launches_by_class[i] = h->prof.n[i]; }
    return 0;
}


// ================================================================================================
// DiT training step: forward with saved activations, backward, AdamW.
// Reference: train_dit.py:649-650 (forward + mse), :680 accelerator.backward, :232-238 AdamW(betas 0.9 / 0.999, eps 1e-7),
// :965-970 clip_grad_norm_ / optimizer.step / zero_grad.  Mixed precision like the reference's bf16 autocast + fp32 master
// weights, with fp16 operands and a loss scale in place of bf16's exponent range: activation gradients travel as fp16 GEMM
// operands multiplied by tr.loss_scale, weight gradients / LayerNorm statistics / the residual-stream gradient are fp32.
// ================================================================================================
int gtav_dit_train_enable(gtav_dit* h, float* grad_arena_dev, int64_t grad_arena_numel) {
    GTAV_REQUIRE(h, "train_enable: null handle");
    GTAV_REQUIRE(!h->tr.on, "train_enable: already enabled");
    GTAV_REQUIRE(!h->any_bf16, "train_enable: the training step runs on fp16 operands (gtav_dit_set_operand_dtype(h, -1, GTAV_OPERAND_F16) first)");
    for (auto& kv : h->wt.slots) GTAV_REQUIRE(!kv.second.set, "train_enable: call it before any gtav_dit_set_weight (the fp32 masters are filled by set_weight)");
    gtav_dit::Train& t = h->tr;
    Arena& a = h->arena;
    const int D = h->D, L = h->L, Hp = h->Hm_pad;
    GTAV_REQUIRE(h->Hm == h->Hm_pad && h->Kpe == h->C * h->p * h->p, "train_enable: padded MLP width / patch size are not implemented for training");
    size_t count = 0;
    std::vector<std::string> names;
    for (auto& kv : h->wt.slots) {
        const std::string& n = kv.first;
        if (n.rfind("tables.", 0) == 0 || n.find("rotary_emb.freqs") != std::string::npos) continue;   // constants (requires_grad False upstream)
        kv.second.trainable = true;
        t.params.push_back(&kv.second);
        names.push_back(n);
        count += (size_t)kv.second.R * kv.second.C;
    }
    t.grad_count = count;
    if (grad_arena_dev) {
        GTAV_REQUIRE(grad_arena_numel == (int64_t)count, "train_enable: the gradient arena has %lld elements, the model has %lld trainable parameters",
                     (long long)grad_arena_numel, (long long)count);
        t.grad_arena = grad_arena_dev;
    } else {
        RET_IF(a.alloc_t(&t.grad_arena, count));
    }
    size_t off = 0;
    for (size_t pi = 0; pi < t.params.size(); ++pi) {
        Slot* sl = t.params[pi];
        const size_t n = (size_t)sl->R * sl->C;
        sl->grad = t.grad_arena + off;
        off += n;
        RET_IF(a.alloc_t(&sl->am, n));
        RET_IF(a.alloc_t(&sl->av, n));
        if (sl->kind == SLOT_F16_PAD) {
            RET_IF(a.alloc_t(&sl->master, n));
            if (names[pi] != "x_embedder.proj.weight")   // every GEMM weight but the patch embedding needs W^T for dX
                RET_IF(a.alloc_t(&sl->wT, (size_t)round_up(sl->C, 128) * round_up(sl->R, 64)));
        } else {
            sl->master = (float*)sl->dst + sl->c0;
        }
    }
    RET_IF(a.alloc_t(&t.ctl, 8));
    RET_IF(a.alloc_t(&t.red_ws, colsum_workspace(h->Mmax > h->max_rows ? h->Mmax : h->max_rows, h->Hm_pad > 6 * D ? h->Hm_pad : 6 * D)));
    RET_IF(a.alloc_t(&t.sumsq_part, (size_t)sumsq_parts(count)));
    {
        std::vector<AdamParam> ap;
        std::vector<AdamItem> ai;
        for (size_t pi = 0; pi < t.params.size(); ++pi) {
            Slot* sl = t.params[pi];
            AdamParam d;
            memset(&d, 0, sizeof(d));
            const bool f16w = sl->kind == SLOT_F16_PAD;
            d.p = sl->master; d.ldp = f16w ? sl->C : sl->Cp; d.R = sl->R; d.C = sl->C; d.g = sl->grad; d.m = sl->am; d.v = sl->av;
            if (f16w) { d.w16 = (f16*)sl->dst; d.Cp16 = sl->Cp; d.wT = sl->wT; d.RpT = round_up(sl->R, 64); }
            ap.push_back(d);
            if (f16w) {
                const unsigned nt = (unsigned)(cdiv(sl->R, 64) * cdiv(sl->C, 64));
                for (unsigned i = 0; i < nt; ++i) ai.push_back(AdamItem{(int)pi, i});
            } else {
                const size_t n = (size_t)sl->R * sl->C;
                for (size_t st = 0; st < n; st += 4096) ai.push_back(AdamItem{(int)pi, (unsigned)st});
            }
        }
        RET_IF(a.alloc_t(&t.adam_params, ap.size()));
        RET_IF(a.alloc_t(&t.adam_items, ai.size()));
        GTAV_CHECK_HIP(hipMemcpy(t.adam_params, ap.data(), ap.size() * sizeof(AdamParam), hipMemcpyHostToDevice));
        GTAV_CHECK_HIP(hipMemcpy(t.adam_items, ai.data(), ai.size() * sizeof(AdamItem), hipMemcpyHostToDevice));
        t.adam_n_items = (int)ai.size();
    }
    const size_t Mx = round_up(h->Mmax, 128), Mp = round_up(h->Mmax, 64), Mm = h->Mmax;
    t.res.resize(4 * L + 1);
    for (auto& r : t.res) RET_IF(a.alloc_t(&r, Mx * D));
    t.hb.resize(2 * L);
    for (int i = 0; i < 2 * L; ++i) {
        gtav_dit::Train::HB& b = t.hb[i];
        RET_IF(a.alloc_t(&b.xnA, Mx * D)); RET_IF(a.alloc_t(&b.ao, Mx * D)); RET_IF(a.alloc_t(&b.y1, Mx * D)); RET_IF(a.alloc_t(&b.xnB, Mx * D));
        RET_IF(a.alloc_t(&b.u, Mx * Hp)); RET_IF(a.alloc_t(&b.hh, Mx * Hp)); RET_IF(a.alloc_t(&b.y2, Mx * D));
        RET_IF(a.alloc_t(&b.q, Mx * D));
        if (i % 2 == 0) { RET_IF(a.alloc_t(&b.k, Mx * D)); RET_IF(a.alloc_t(&b.v, Mx * D)); }
        else { RET_IF(a.alloc_t(&b.k, Mx * 2 * D)); b.v = b.k; }
    }
    RET_IF(a.alloc_t(&t.xnF, Mx * D)); RET_IF(a.alloc_t(&t.xp, Mx * h->Kpe));
    const size_t R = h->max_rows;
    RET_IF(a.alloc_t(&t.z0, R * D)); RET_IF(a.alloc_t(&t.cpre, R * D));
    RET_IF(a.alloc_t(&t.dres, Mx * D)); RET_IF(a.alloc_t(&t.dtmp, Mx * D)); RET_IF(a.alloc_t(&t.stats, 2 * Mx));
    if (ln_bwd_fused_ok(D)) RET_IF(a.alloc_t(&t.ln_part, ln_bwd_fused_workspace((int)R, h->P, D)));
    RET_IF(a.alloc_t(&t.dmod, R * h->MODW)); RET_IF(a.alloc_t(&t.dSc, R * D)); RET_IF(a.alloc_t(&t.ada_part, ada_bwd_dx_workspace(h->MODW, D, (int)R))); RET_IF(a.alloc_t(&t.dc, R * D)); RET_IF(a.alloc_t(&t.dh0, R * D));
    RET_IF(a.alloc_t(&t.dz0, R * D));
    RET_IF(a.alloc_t(&t.g_d, Mx * D)); RET_IF(a.alloc_t(&t.g_d2, Mx * D)); RET_IF(a.alloc_t(&t.g_h, Mx * Hp)); RET_IF(a.alloc_t(&t.g_u, Mx * Hp)); RET_IF(a.alloc_t(&t.g_qkv, Mx * 3 * D));
    RET_IF(a.alloc_t(&t.dao, Mm * D)); RET_IF(a.alloc_t(&t.dfo, Mx * 64));
    const size_t widest = (size_t)(Hp > 3 * D ? Hp : 3 * D);
    RET_IF(a.alloc_t(&t.tA, widest * Mp)); RET_IF(a.alloc_t(&t.tB, widest * Mp));
    if (D % 256 == 0 && Hp % 256 == 0) {   // (rows of the transposed images: fc2 dY / X, fc1, out-proj, QKV)
        const size_t ra[4] = {(size_t)D, (size_t)Hp, (size_t)D, (size_t)3 * D}, rb[4] = {(size_t)Hp, (size_t)D, (size_t)D, (size_t)D};
        for (int i = 0; i < 4; ++i) { RET_IF(a.alloc_t(&t.tAg[i], ra[i] * Mp)); RET_IF(a.alloc_t(&t.tBg[i], rb[i] * Mp)); }
    }
    t.on = true;
    return 0;
}

int gtav_dit_train_param_count(gtav_dit* h, int64_t* numel) {
    GTAV_REQUIRE(h && numel, "train_param_count: null argument");
    int64_t c = 0;
    for (auto& kv : h->wt.slots) {
        const std::string& n = kv.first;
        if (n.rfind("tables.", 0) == 0 || n.find("rotary_emb.freqs") != std::string::npos) continue;
        c += (int64_t)kv.second.R * kv.second.C;
    }
    *numel = c;
    return 0;
}

int gtav_dit_set_loss_scale(gtav_dit* h, float scale) {
    GTAV_REQUIRE(h && scale > 0.f, "set_loss_scale: bad argument");
    h->tr.loss_scale = scale;
    return 0;
}

int gtav_dit_set_grad_divisor(gtav_dit* h, float divisor) {
    GTAV_REQUIRE(h && h->tr.on && divisor >= 1.0f, "set_grad_divisor: bad argument");
    h->tr.grad_div = divisor;
    return 0;
}
int gtav_dit_zero_grad(gtav_dit* h, void* stream) {
    GTAV_REQUIRE(h && h->tr.on, "zero_grad: training is not enabled");
    // (hipMemsetAsync splits 2.4 GB into ~600 fill launches of 4 MB: 3.8 ms per step in the rocprofv3 trace; one grid-stride kernel: 0.5 ms)
    RET_IF(launch_fill_f32(h->tr.grad_arena, h->tr.grad_count, 0.f, (hipStream_t)stream));
    // a training step starts here: saturation / non-finite bits raised by an earlier forward on this handle (validation, predict) are not this step's
    // overflow — clear them so that only the step's own stores can make the optimizer skip (gtav_dit_check reports inference saturation before that)
    return launch_err_clear(h->err_flag, ERR_F16_SAT | ERR_NONFINITE, (hipStream_t)stream);
}

// raw (loss-scaled) gradient of one parameter, torch layout; the caller divides by the loss scale
int gtav_dit_get_grad(gtav_dit* h, const char* name, float* dst, int64_t numel, void* stream) {
    GTAV_REQUIRE(h && name && dst && h->tr.on, "get_grad: bad argument / training is not enabled");
    auto it = h->wt.slots.find(name);
    GTAV_REQUIRE(it != h->wt.slots.end() && it->second.grad, "get_grad: '%s' is not a trainable parameter", name);
    GTAV_REQUIRE(numel == (int64_t)it->second.R * it->second.C, "get_grad: '%s' size mismatch", name);
    GTAV_CHECK_HIP(hipMemcpyAsync(dst, it->second.grad, numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int gtav_dit_train_forward(gtav_dit* h, const float* x, const int64_t* t64, const float* actions, float* out, int32_t B, int32_t T, void* stream) {
    GTAV_REQUIRE(h && x && t64 && out, "train_forward: null argument");
    GTAV_REQUIRE(h->tr.on && h->finalized, "train_forward: call gtav_dit_train_enable, load the weights and finalize first");
    GTAV_REQUIRE(B >= 1 && B <= h->maxB && T >= 1 && T <= h->maxT, "train_forward: B=%d T=%d outside capacity (%d, %d)", B, T, h->maxB, h->maxT);
    GTAV_REQUIRE(!actions || h->A > 0, "train_forward: model has no external_cond");
    hipStream_t s = (hipStream_t)stream;
    gtav_dit::Train& tr = h->tr;
    const int D = h->D, P = h->P, NB = B * T, M = NB * P, L = h->L, rows = NB, ldhc = D + h->Apad;
    h->prepared.valid = false;
    h->kvrec.valid = false;
    // conditioning path with its pre-activations kept (dit_cond applies SiLU inside the skinny GEMM)
    RET_IF(launch_cond_inputs(t64, rows, 1, nullptr, 0, h->sincos, h->E, actions, h->A, 0, h->A, h->HC, ldhc, D, h->Apad, h->err_flag, s));
    RET_IF(launch_skinny_f32(h->E, 256, h->w_t0, h->b_t0, tr.z0, D, rows, D, 256, 0, s));
    RET_IF(launch_silu(tr.z0, D, h->HC, ldhc, rows, D, s));
    RET_IF(launch_skinny_f32(h->HC, ldhc, h->w_t2cat, actions ? h->b_t2a : h->b_t2, tr.cpre, D, rows, D, ldhc, 0, s));
    RET_IF(launch_silu(tr.cpre, D, h->Sc, D, rows, D, s));
    RET_IF(launch_skinny_f32(h->Sc, D, h->w_ada, h->b_ada, h->mod, h->MODW, rows, h->MODW, D, 0, s));
    const float* mod = h->mod;
    RET_IF(launch_patchify(x, nullptr, NB, h->C, h->H, h->W, h->p, tr.xp, h->Kpe, 1.f, 0.f, h->err_flag, s));
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.X = tr.xp; g.ldx = h->Kpe; g.W = h->w_pe; g.M = M; g.N = D; g.K = h->Kpe; g.bias = h->b_pe; g.out = tr.res[0]; g.ldo = D;
    RET_IF(launch_gemm(g, EPI_F32, s));
    LnPending pend;
    bool have_pend = false;
    auto resid_gemm = [&](const f16* X, const f16* Wt, int K, const float* bias, const float* gate, float* x_out, f16* y_save) -> int {
        GemmParams q;
        memset(&q, 0, sizeof(q));
        q.X = X; q.ldx = K; q.W = Wt; q.M = M; q.N = D; q.K = K; q.out = h->parts; q.ldo = D;
        q.splitk = gemm_choose_splitk(M, D, K);
        GTAV_REQUIRE((size_t)q.splitk * M <= h->parts_rows, "split-K slabs exceed workspace");
        RET_IF(launch_gemm(q, EPI_PARTIAL, s));
        memset(&pend, 0, sizeof(pend));
        pend.parts = h->parts; pend.nsplit = q.splitk; pend.slab_stride = (size_t)M * D; pend.ld = D; pend.bias = bias;
        pend.gate = gate; pend.gate_stride = h->MODW; pend.gate_rows = nullptr; pend.rows_per_gate = P;
        pend.x_out = x_out; pend.y_save = y_save;
        have_pend = true;
        return 0;
    };
    for (int l = 0; l < L; ++l)
        for (int hf = 0; hf < 2; ++hf) {
            const int i = l * 2 + hf;
            const gtav_dit::Half& w = h->halves[i];
            gtav_dit::Train::HB& b = tr.hb[i];
            const float* mb = mod + (size_t)i * 6 * D;
            // LN1 normalises r_{2i} (= r_{2i-1} + gate (fc2 of the previous half-block), written to res[2i] by this launch)
            RET_IF(launch_ln_modulate(i == 0 ? tr.res[0] : tr.res[2 * i - 1], D, b.xnA, D, M, D, mb, mb + D, h->MODW, nullptr, P, have_pend ? &pend : nullptr, h->err_flag, s));
            have_pend = false;
            memset(&g, 0, sizeof(g));
            g.X = b.xnA; g.ldx = D; g.W = w.w_qkv; g.M = M; g.N = 3 * D; g.K = D; g.D = D; g.S = P; g.err_flag = h->err_flag;
            if (hf == 0) { g.qkv_mode = QKV_SPATIAL; g.q = b.q; g.k = b.k; g.v = b.v; g.rope_cs = h->rope_s.cs_dev; }
            else { g.qkv_mode = QKV_TEMPORAL; g.q = b.q; g.k = b.k; g.v = b.k; g.Tq = T; g.t0 = 0; g.Tmax = h->maxT; g.rope_cs = h->rope_t.cs_dev; }
            RET_IF(launch_gemm(g, EPI_QKV, s));
            if (hf == 0) RET_IF(launch_attn_spatial(b.q, b.k, b.v, b.ao, NB, h->heads, P, s));
            else RET_IF(launch_attn_temporal(b.q, b.k, b.ao, B, P, D, T, 0, h->maxT, s));
            RET_IF(resid_gemm(b.ao, w.w_out, D, w.b_out, mb + 2 * D, tr.res[2 * i + 1], b.y1));
            RET_IF(launch_ln_modulate(tr.res[2 * i], D, b.xnB, D, M, D, mb + 3 * D, mb + 4 * D, h->MODW, nullptr, P, &pend, h->err_flag, s));
            have_pend = false;
            memset(&g, 0, sizeof(g));
            g.X = b.xnB; g.ldx = D; g.W = w.w_fc1; g.M = M; g.N = h->Hm; g.K = D; g.bias = w.b_fc1; g.out = b.u; g.ldo = h->Hm_pad; g.err_flag = h->err_flag;
            if (g_fuse_gelu_fwd) g.out2 = b.hh;             // h = GELU(u) as a second image of the same epilogue (gemm.h out2)
            RET_IF(launch_gemm(g, EPI_F16_TILED, s));       // the pre-activation is kept: gelu'(u) in the backward pass
            if (!g_fuse_gelu_fwd) RET_IF(launch_gelu_tiled(b.u, b.hh, (size_t)round_up(M, 128) * h->Hm_pad, s));
            RET_IF(resid_gemm(b.hh, w.w_fc2, h->Hm_pad, w.b_fc2, mb + 5 * D, tr.res[2 * i + 2], b.y2));
        }
    const float* mf = mod + (size_t)L * 12 * D;
    RET_IF(launch_ln_modulate(tr.res[4 * L - 1], D, tr.xnF, D, M, D, mf, mf + D, h->MODW, nullptr, P, &pend, h->err_flag, s));
    memset(&g, 0, sizeof(g));
    g.X = tr.xnF; g.ldx = D; g.W = h->w_final; g.M = M; g.N = h->Nfin; g.K = D; g.bias = h->b_final; g.out = h->fo; g.ldo = h->Nfin;
    RET_IF(launch_gemm(g, EPI_F32, s));
    RET_IF(launch_unpatchify(h->fo, h->Nfin, out, NB, h->C, h->H, h->W, h->p, 0, 1.f, 0.f, s));
    tr.B = B; tr.T = T; tr.M = M; tr.Mp = round_up(M, 64); tr.rows = rows; tr.have_actions = actions != nullptr; tr.have_fwd = true;
    return 0;
}

// Residual stream of the last training forward after k branch additions (every block adds four branches: spatial attention, spatial
// MLP, temporal attention, temporal MLP): k = 0 is the patch embedding output, k = 4 (l + 1) the output of block l, k = 4 L the input of
// the final layer.  fp32 [B T P][D] in token order (b, t, p): per-block parity taps (model/dit.py:370-372).
int gtav_dit_train_get_residual(gtav_dit* h, int32_t k, float* dst, int64_t numel, void* stream) {
    GTAV_REQUIRE(h && dst && h->tr.on && h->tr.have_fwd, "train_get_residual: no saved forward");
    GTAV_REQUIRE(k >= 0 && k <= 4 * h->L, "train_get_residual: k=%d must be in [0, %d]", k, 4 * h->L);
    GTAV_REQUIRE(numel == (int64_t)h->tr.M * h->D, "train_get_residual: expected %lld elements", (long long)h->tr.M * h->D);
    GTAV_CHECK_HIP(hipMemcpyAsync(dst, h->tr.res[k], numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// Backward of loss = mean((v_pred[:, -1] - v_target)^2) through the forward saved by gtav_dit_train_forward.  Gradients are ADDED to the
// gradient arena (gtav_dit_zero_grad first), multiplied by the loss scale.
// Phases of the backward pass (gtav_dit_train_backward_phases): 0 = loss, final projection, final LayerNorm; 1 .. L = the blocks in
// reverse, phase p = block L - p (both half-blocks and the block's adaLN projection: after phase p every gradient named "blocks.<L-p>.*" is
// complete, so its slice of the arena can be all-reduced while the earlier blocks are still being differentiated); L + 1 = patch embedding
// and the shared conditioning path (t_embedder, external_cond).  tr.dres / tr.dmod carry the state from one phase to the next.
int gtav_dit_train_backward_phases(gtav_dit* h, const float* v_pred, const float* v_target, int32_t phase_begin, int32_t phase_end, void* stream) {
    GTAV_REQUIRE(h && v_pred && v_target, "train_backward: null argument");
    GTAV_REQUIRE(h->tr.on && h->tr.have_fwd, "train_backward: no saved forward (gtav_dit_train_forward)");
    GTAV_REQUIRE(phase_begin >= 0 && phase_begin <= phase_end && phase_end <= h->L + 2, "train_backward: phases [%d, %d) outside [0, %d]", phase_begin, phase_end,
                 h->L + 2);
    hipStream_t s = (hipStream_t)stream;
    gtav_dit::Train& tr = h->tr;
    const int D = h->D, P = h->P, L = h->L, B = tr.B, T = tr.T, M = tr.M, Mp = tr.Mp, NB = B * T, rows = tr.rows, Hp = h->Hm_pad, MODW = h->MODW;
    const int ldhc = D + h->Apad;
    auto slot = [&](const std::string& n) -> Slot& { return h->wt.slots[n]; };
    // dX = dY W: A = dY tile-major [M][Kc], WT = tile-major W^T [N][Kc]
    auto gemm_dx = [&](const f16* A, const f16* WT, int N, int Kc, int epi, void* out, int ldo) -> int {
        GemmParams q;
        memset(&q, 0, sizeof(q));
        q.X = A; q.ldx = Kc; q.W = WT; q.M = M; q.N = N; q.K = Kc; q.out = out; q.ldo = ldo; q.err_flag = h->err_flag;
        return launch_gemm(q, epi, s);
    };
    // dW[n][k] += sum_m dY[m][n] X[m][k]: both operands transposed to [.][Mp] (tokens are the contraction), accumulating epilogue
    // Half-blocks of production widths defer their four dW GEMMs into ONE grouped launch of 256 x 256 tiles (flush_dw; gemm.h)
    GemmDwGroup dwg[GEMM_DW_MAX_GROUPS];
    int ndw = 0;
    bool defer_dw = false;
    if (tr.tAg[0] && g_dw_grouped) {
        const GemmDwGroup probe[4] = {{tr.tAg[0], tr.tBg[0], tr.dres, D, Hp, Hp}, {tr.tAg[1], tr.tBg[1], tr.dres, Hp, D, D}, {tr.tAg[2], tr.tBg[2], tr.dres, D, D, D},
                                      {tr.tAg[3], tr.tBg[3], tr.dres, 3 * D, D, D}};
        defer_dw = gemm_dw_grouped_ok(probe, 4, Mp);
    }
    // Whole 128-token row tiles: the grouped launch contracts over the rows of the tile-major operands THEMSELVES (transposing LDS reads, gemm.hip
    // mainloop256_tn) — no transposed copies (8 of the 17 us transposes per half-block).  The operands must then live until flush_dw: the out-projection's
    // dY gets a buffer of its own (g_d2), the saved activations and g_u / g_qkv are not rewritten inside a half-block.
    const bool tn_dw = defer_dw && g_dw_tn && M % 128 == 0;
    // gate backward, the gate's own gradient and the bias gradient of the Linear in front of it in one pass over dres (train.hip gate_bwd_fused_kernel): the
    // per-frame partial sums of the bias gradient (NB x D floats) must fit the reduction workspace
    const bool fuse_ln = g_fuse_ln && tr.ln_part && M == NB * P && NB <= rows;
    auto ln_bwd = [&](const float* dxn, const float* x, const float* scale, int accumulate, float* dshift, float* dscale) -> int {
        if (fuse_ln) return launch_ln_mod_bwd_fused(dxn, x, scale, MODW, NB, P, D, tr.dres, accumulate, dshift, dscale, tr.ln_part, s);
        RET_IF(launch_ln_mod_bwd(dxn, x, scale, MODW, P, M, D, tr.dres, accumulate, tr.stats, s));
        return launch_frame_reduce_ln(dxn, x, tr.stats, NB, P, D, dshift, dscale, MODW, s);
    };
    const size_t ws_cap = colsum_workspace(h->Mmax > h->max_rows ? h->Mmax : h->max_rows, h->Hm_pad > 6 * D ? h->Hm_pad : 6 * D);   // floats of tr.red_ws
    const bool fuse_gate = g_fuse_gate && M == NB * P && (size_t)NB * D <= ws_cap;
    const bool defer_bias = fuse_gate && g_fuse_gelu && (size_t)2 * NB * D + (size_t)gelu_bwd_colsum_splits(M) * Hp <= ws_cap;
    auto flush_dw = [&]() -> int {
        if (!ndw) return 0;
        const int n = ndw;
        ndw = 0;
        return launch_gemm_dw_grouped(dwg, n, tn_dw ? M : Mp, h->err_flag, s, tn_dw);
    };
    auto gemm_dw = [&](const f16* dY, int N, const f16* X, int K, float* grad, int slot_i = -1) -> int {
        if (tn_dw && slot_i >= 0) {
            dwg[ndw++] = GemmDwGroup{dY, X, grad, N, K, K};
            return 0;
        }
        if (defer_dw && slot_i >= 0) {
            RET_IF(launch_transpose_tiled_f16(dY, M, N, tr.tAg[slot_i], s));
            RET_IF(launch_transpose_tiled_f16(X, M, K, tr.tBg[slot_i], s));
            dwg[ndw++] = GemmDwGroup{tr.tAg[slot_i], tr.tBg[slot_i], grad, N, K, K};
            return 0;
        }
        if (gemm_tn_pays(N, K, M)) {   // contraction over the rows of the tile-major operands themselves (transposing LDS reads): no transposes
            GemmParams q;
            memset(&q, 0, sizeof(q));
            q.X = dY; q.ldx = N; q.W = X; q.M = N; q.N = K; q.K = M; q.out = grad; q.ldo = K;
            return launch_gemm_tn(q, s);
        }
        RET_IF(launch_transpose_tiled_f16(dY, M, N, tr.tA, s));
        RET_IF(launch_transpose_tiled_f16(X, M, K, tr.tB, s));
        GemmParams q;
        memset(&q, 0, sizeof(q));
        q.X = tr.tA; q.ldx = Mp; q.W = tr.tB; q.M = N; q.N = K; q.K = Mp; q.out = grad; q.ldo = K;
        return launch_gemm(q, EPI_RESID, s);
    };
    const float scale = 2.0f * tr.loss_scale / ((float)B * (float)(h->C * h->H * h->W));
    GTAV_REQUIRE(h->Nfin <= 64, "train_backward: a final projection wider than 64 features is not implemented");
    const float* mod = h->mod;
    float* dmod = tr.dmod;
    // gradient of one adaLN projection (rows [row0, row0 + n) of W_ada / b_ada) from the dmod columns its LayerNorm / gate backward filled
    auto ada_grads = [&](size_t row0, int n, const std::string& wn, const std::string& bn) -> int {
        RET_IF(launch_gemm_tn_f32(dmod + row0, MODW, h->Sc, D, rows, n, D, slot(wn).grad, D, s));
        return launch_colsum_f32(dmod + row0, MODW, rows, n, slot(bn).grad, tr.red_ws, s);
    };
    // ---- phase 0: loss -> final projection -> final LayerNorm ----
    if (phase_begin <= 0 && 0 < phase_end) {
    RET_IF(launch_mse_bwd_patch(v_pred, v_target, B, T, h->C, h->H, h->W, h->p, scale, tr.dfo, 64, h->err_flag, s));
    {
        Slot& wf = slot("final_layer.linear.weight");
        // db: column sums over the 64-wide (zero-padded) dfo, only the first Nfin belong to the bias: sum into a scratch row first
        GTAV_CHECK_HIP(hipMemsetAsync(tr.dSc, 0, 64 * sizeof(float), s));
        RET_IF(launch_colsum_tiled_f16(tr.dfo, M, 64, tr.dSc, tr.red_ws, s));
        RET_IF(launch_add_f32(slot("final_layer.linear.bias").grad, tr.dSc, slot("final_layer.linear.bias").grad, h->Nfin, s));
        // dW_final [Nfin][D] += dfo^T xnF   (M = Nfin rows of the 64-row transposed operand)
        RET_IF(launch_transpose_tiled_f16(tr.dfo, M, 64, tr.tA, s));
        RET_IF(launch_transpose_tiled_f16(tr.xnF, M, D, tr.tB, s));
        GemmParams q;
        memset(&q, 0, sizeof(q));
        q.X = tr.tA; q.ldx = Mp; q.W = tr.tB; q.M = h->Nfin; q.N = D; q.K = Mp; q.out = wf.grad; q.ldo = D;
        RET_IF(launch_gemm(q, EPI_RESID, s));
        // d xnF = dfo W_final  -> fp32 [M][D]
        RET_IF(gemm_dx(tr.dfo, wf.wT, D, 64, EPI_F32, tr.dtmp, D));
        const float* mf = mod + (size_t)L * 12 * D;
        float* dmf = dmod + (size_t)L * 12 * D;
        RET_IF(ln_bwd(tr.dtmp, tr.res[4 * L], mf + D, 0, dmf, dmf + D));
        RET_IF(ada_grads((size_t)L * 12 * D, 2 * D, "final_layer.adaLN_modulation.1.weight", "final_layer.adaLN_modulation.1.bias"));
    }
    }
    // ---- phases 1 .. L: the blocks in reverse (temporal half-block, then spatial); tr.dres = d loss / d (residual state) ----
    for (int i = 2 * L - 1; i >= 0; --i) {
        const int l = i / 2, hf = i % 2;
        const int phase = L - l;
        if (phase < phase_begin || phase >= phase_end) continue;
        gtav_dit::Train::HB& b = tr.hb[i];
        char pre[64];
        snprintf(pre, sizeof(pre), "blocks.%d.%c_", l, hf == 0 ? 's' : 't');
        const std::string P_(pre);
        const float* mb = mod + (size_t)i * 6 * D;
        float* dmb = dmod + (size_t)i * 6 * D;
        // r_{2i+2} = r_{2i+1} + gate_mlp y2
        // (defer_bias: the partial sums of the half-block's three bias gradients go to three regions of the workspace and ONE launch adds them at the end of the half-block)
        float* const ws_fc2 = tr.red_ws, *const ws_out = tr.red_ws + (defer_bias ? (size_t)NB * D : 0), *const ws_fc1 = tr.red_ws + (defer_bias ? (size_t)2 * NB * D : 0);   // (not deferred: every reduction follows its partial sums at once and the regions may coincide)
        if (fuse_gate) {
            RET_IF(launch_gate_bwd_fused(tr.dres, b.y2, mb + 5 * D, MODW, NB, P, D, tr.g_d, dmb + 5 * D, defer_bias ? nullptr : slot(P_ + "mlp.fc2.bias").grad, ws_fc2, h->err_flag, s));
        } else {
            RET_IF(launch_gate_bwd(tr.dres, mb + 5 * D, MODW, P, M, D, tr.g_d, h->err_flag, s));
            RET_IF(launch_frame_reduce_gate(tr.dres, b.y2, NB, P, D, dmb + 5 * D, MODW, s));
            RET_IF(launch_colsum_tiled_f16(tr.g_d, M, D, slot(P_ + "mlp.fc2.bias").grad, tr.red_ws, s));
        }
        RET_IF(gemm_dw(tr.g_d, D, b.hh, Hp, slot(P_ + "mlp.fc2.weight").grad, 0));
        RET_IF(gemm_dx(tr.g_d, slot(P_ + "mlp.fc2.weight").wT, Hp, D, EPI_F16_TILED, tr.g_h, Hp));
        if (defer_bias) {
            RET_IF(launch_gelu_bwd_tiled_colsum(tr.g_h, b.u, tr.g_u, M, Hp, nullptr, ws_fc1, h->err_flag, s));
        } else if (g_fuse_gelu && colsum_workspace(round_up(M, 128), Hp) <= ws_cap) {
            RET_IF(launch_gelu_bwd_tiled_colsum(tr.g_h, b.u, tr.g_u, M, Hp, slot(P_ + "mlp.fc1.bias").grad, tr.red_ws, h->err_flag, s));
        } else {
            RET_IF(launch_gelu_bwd_tiled(tr.g_h, b.u, tr.g_u, (size_t)round_up(M, 128) * Hp, h->err_flag, s));
            RET_IF(launch_colsum_tiled_f16(tr.g_u, M, Hp, slot(P_ + "mlp.fc1.bias").grad, tr.red_ws, s));
        }
        RET_IF(gemm_dw(tr.g_u, Hp, b.xnB, D, slot(P_ + "mlp.fc1.weight").grad, 1));
        RET_IF(gemm_dx(tr.g_u, slot(P_ + "mlp.fc1.weight").wT, D, Hp, EPI_F32, tr.dtmp, D));
        RET_IF(ln_bwd(tr.dtmp, tr.res[2 * i + 1], mb + 4 * D, 1, dmb + 3 * D, dmb + 4 * D));
        // r_{2i+1} = r_{2i} + gate_msa y1
        f16* const g_o = tn_dw ? tr.g_d2 : tr.g_d;   // (the fc2 weight gradient above still reads g_d when the grouped launch is deferred without copies)
        if (fuse_gate) {
            RET_IF(launch_gate_bwd_fused(tr.dres, b.y1, mb + 2 * D, MODW, NB, P, D, g_o, dmb + 2 * D, defer_bias ? nullptr : slot(P_ + "attn.to_out.bias").grad, ws_out, h->err_flag, s));
        } else {
            RET_IF(launch_gate_bwd(tr.dres, mb + 2 * D, MODW, P, M, D, g_o, h->err_flag, s));
            RET_IF(launch_frame_reduce_gate(tr.dres, b.y1, NB, P, D, dmb + 2 * D, MODW, s));
            RET_IF(launch_colsum_tiled_f16(g_o, M, D, slot(P_ + "attn.to_out.bias").grad, tr.red_ws, s));
        }
        RET_IF(gemm_dw(g_o, D, b.ao, D, slot(P_ + "attn.to_out.weight").grad, 2));
        RET_IF(gemm_dx(g_o, slot(P_ + "attn.to_out.weight").wT, D, D, EPI_F16, tr.dao, D));
        if (hf == 0) RET_IF(launch_attn_spatial_bwd(b.q, b.k, b.v, tr.dao, NB, h->heads, P, D, h->rope_s.cs_dev, tr.g_qkv, h->err_flag, s));
        else RET_IF(launch_attn_temporal_bwd(b.q, b.k, tr.dao, B, P, D, T, h->maxT, h->rope_t.cs_dev, tr.g_qkv, h->err_flag, s));
        RET_IF(gemm_dw(tr.g_qkv, 3 * D, b.xnA, D, slot(P_ + "attn.to_qkv.weight").grad, 3));
        if (defer_bias) {
            const float* wsv[3] = {ws_fc2, ws_out, ws_fc1};
            float* dbv[3] = {slot(P_ + "mlp.fc2.bias").grad, slot(P_ + "attn.to_out.bias").grad, slot(P_ + "mlp.fc1.bias").grad};
            const int spv[3] = {NB, NB, gelu_bwd_colsum_splits(M)}, nv[3] = {D, D, Hp};
            RET_IF(launch_colsum_reduce_multi(wsv, dbv, spv, nv, 3, s));
        }
        RET_IF(flush_dw());
        RET_IF(gemm_dx(tr.g_qkv, slot(P_ + "attn.to_qkv.weight").wT, D, 3 * D, EPI_F32, tr.dtmp, D));
        RET_IF(ln_bwd(tr.dtmp, tr.res[2 * i], mb + D, 1, dmb, dmb + D));
        // all six dmod chunks of this half-block are in place: its adaLN projection's gradients
        RET_IF(ada_grads((size_t)i * 6 * D, 6 * D, P_ + "adaLN_modulation.1.weight", P_ + "adaLN_modulation.1.bias"));
    }
    if (!(phase_begin <= L + 1 && L + 1 < phase_end)) return 0;
    // ---- phase L + 1: patch embedding: r_0 = xp W_pe^T + b_pe ----
    RET_IF(launch_colsum_f32(tr.dres, D, M, D, slot("x_embedder.proj.bias").grad, tr.red_ws, s));
    RET_IF(launch_to_tiled_f16(tr.dres, M, D, tr.g_d, h->err_flag, s));
    {
        Slot& wpe = slot("x_embedder.proj.weight");
        GTAV_REQUIRE(wpe.C == h->Kpe, "train_backward: a patch embedding with padded K (%d of %d) is not implemented", wpe.C, h->Kpe);
        RET_IF(gemm_dw(tr.g_d, D, tr.xp, h->Kpe, wpe.grad));
    }
    // ---- the shared conditioning path (fp32, `rows` = B T rows): c = W_2 SiLU(W_0 e + b_0) + b_2 (+ W_ext a + b_ext), SiLU(c) feeds every adaLN
    // projection (their own gradients were taken block by block above) ----
    RET_IF(launch_ada_bwd_dx(dmod, MODW, h->w_ada, D, rows, tr.dSc, tr.ada_part, s));
    RET_IF(launch_silu_bwd(tr.dSc, D, tr.cpre, D, tr.dc, D, rows, D, s));
    RET_IF(launch_colsum_f32(tr.dc, D, rows, D, slot("t_embedder.mlp.2.bias").grad, tr.red_ws, s));
    RET_IF(launch_gemm_tn_f32(tr.dc, D, h->HC, ldhc, rows, D, D, slot("t_embedder.mlp.2.weight").grad, D, s));
    if (tr.have_actions) {
        RET_IF(launch_colsum_f32(tr.dc, D, rows, D, slot("external_cond.bias").grad, tr.red_ws, s));
        RET_IF(launch_gemm_tn_f32(tr.dc, D, h->HC + D, ldhc, rows, D, h->A, slot("external_cond.weight").grad, h->A, s));
    }
    RET_IF(launch_gemm_nn_f32(tr.dc, D, h->w_t2cat, ldhc, rows, D, D, tr.dh0, D, s));
    RET_IF(launch_silu_bwd(tr.dh0, D, tr.z0, D, tr.dz0, D, rows, D, s));
    RET_IF(launch_colsum_f32(tr.dz0, D, rows, D, slot("t_embedder.mlp.0.bias").grad, tr.red_ws, s));
    RET_IF(launch_gemm_tn_f32(tr.dz0, D, h->E, 256, rows, D, 256, slot("t_embedder.mlp.0.weight").grad, 256, s));
    // Last kernel of the backward pass: a saturated / non-finite fp16 store on THIS rank becomes +inf in the embedder bucket (the one the data-parallel
    // harness all-reduces last, train.gradient_buckets), so the skip decision of the optimizer step is the same on every rank (ops.h)
    RET_IF(launch_overflow_publish(h->err_flag, slot("x_embedder.proj.bias").grad, s));
    return 0;
}

int gtav_dit_train_backward(gtav_dit* h, const float* v_pred, const float* v_target, void* stream) {
    GTAV_REQUIRE(h, "train_backward: null handle");
    return gtav_dit_train_backward_phases(h, v_pred, v_target, 0, h->L + 2, stream);
}

// Slice [offset, offset + count) of the gradient arena that holds the parameters whose names start with `prefix` (names are laid out in
// lexicographic order, so "blocks.7." is one contiguous slice): the buckets of an all-reduce overlapped with the backward pass.
int gtav_dit_train_param_range(gtav_dit* h, const char* prefix, int64_t* offset, int64_t* count) {
    GTAV_REQUIRE(h && prefix && offset && count && h->tr.on, "train_param_range: bad argument / training is not enabled");
    const size_t plen = strlen(prefix);
    int64_t off = -1, cnt = 0, last_end = -1;
    for (auto& kv : h->wt.slots) {
        Slot& sl = kv.second;
        if (!sl.grad || kv.first.compare(0, plen, prefix) != 0) continue;
        const int64_t o = sl.grad - h->tr.grad_arena, n = (int64_t)sl.R * sl.C;
        if (off < 0) off = o;
        GTAV_REQUIRE(last_end < 0 || o == last_end, "train_param_range: parameters with prefix '%s' are not contiguous in the arena", prefix);
        last_end = o + n;
        cnt += n;
    }
    GTAV_REQUIRE(off >= 0, "train_param_range: no trainable parameter starts with '%s'", prefix);
    *offset = off;
    *count = cnt;
    return 0;
}

// One optimizer step over every trainable parameter: global gradient norm -> clipping coefficient (folded with 1 / loss_scale; a
// non-finite norm skips the step) -> AdamW -> refreshed fp16 operand copies (W and W^T) of the GEMM weights.
int gtav_dit_adamw_step(gtav_dit* h, float lr, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm, void* stream) {
    GTAV_REQUIRE(h && h->tr.on, "adamw_step: training is not enabled");
    hipStream_t s = (hipStream_t)stream;
    gtav_dit::Train& tr = h->tr;
    RET_IF(launch_sumsq(tr.grad_arena, tr.grad_count, tr.sumsq_part, s));
    // overflow (non-finite norm, or a saturated fp16 gradient / activation recorded in the error word) skips the step on the device; the
    // Adam step count and its bias corrections live in ctl[4..6] and advance only with applied steps
    RET_IF(launch_clip_coef(tr.ctl, tr.sumsq_part, sumsq_parts(tr.grad_count), 1.0f / (tr.loss_scale * tr.grad_div), max_grad_norm, beta1, beta2, h->err_flag, s));
    // one launch: AdamW on every parameter + the fp16 W / W^T operands of the GEMM weights rewritten from the updated masters
    RET_IF(launch_adamw_multi(tr.adam_params, tr.adam_items, tr.adam_n_items, tr.ctl, lr, beta1, beta2, eps, weight_decay, s));
    RET_IF(launch_add_f32(h->b_t2, h->b_ext, h->b_t2a, h->D, s));   // fused bias of c when actions are given (gtav_dit_finalize)
    h->prepared.valid = false;
    h->kvrec.valid = false;
    return 0;
}

// ctl: [0] sum of squares of the scaled gradients, [1] step coefficient (0 = the step was skipped), [2] skipped steps so far,
// [3] unscaled global gradient norm of the last step (torch.nn.utils.clip_grad_norm_'s return value)
// Optimizer state of one parameter (AdamW first / second moments, contiguous in the parameter's state-dict shape) and the step counters:
// with gtav_dit_get_weight / set_weight (the fp32 masters) this is everything `accelerator.save_state` / `load_state` keep for the
// optimizer (train_dit.py:765-849).
static int opt_slot(gtav_dit* h, const char* name, int64_t numel, Slot** out) {
    GTAV_REQUIRE(h && name && h->tr.on, "opt_state: training is not enabled");
    auto it = h->wt.slots.find(name);
    GTAV_REQUIRE(it != h->wt.slots.end() && it->second.trainable && it->second.am && it->second.av, "opt_state: '%s' is not a trainable parameter", name);
    GTAV_REQUIRE(numel == (int64_t)it->second.R * it->second.C, "opt_state: '%s' has %d x %d elements, got %lld", name, it->second.R, it->second.C, (long long)numel);
    *out = &it->second;
    return 0;
}
int gtav_dit_get_opt_state(gtav_dit* h, const char* name, float* m_dst, float* v_dst, int64_t numel, void* stream) {
    Slot* sl = nullptr;
    RET_IF(opt_slot(h, name, numel, &sl));
    GTAV_REQUIRE(m_dst && v_dst, "get_opt_state: null destination");
    GTAV_CHECK_HIP(hipMemcpyAsync(m_dst, sl->am, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    GTAV_CHECK_HIP(hipMemcpyAsync(v_dst, sl->av, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
int gtav_dit_set_opt_state(gtav_dit* h, const char* name, const float* m_src, const float* v_src, int64_t numel, void* stream) {
    Slot* sl = nullptr;
    RET_IF(opt_slot(h, name, numel, &sl));
    GTAV_REQUIRE(m_src && v_src, "set_opt_state: null source");
    GTAV_CHECK_HIP(hipMemcpyAsync(sl->am, m_src, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    GTAV_CHECK_HIP(hipMemcpyAsync(sl->av, v_src, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
int gtav_dit_get_opt_step(gtav_dit* h, int64_t* applied_steps, int64_t* skipped_steps, void* stream) {
    GTAV_REQUIRE(h && h->tr.on && applied_steps && skipped_steps, "get_opt_step: bad argument");
    float c[8];
    GTAV_CHECK_HIP(hipMemcpyAsync(c, h->tr.ctl, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GTAV_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    *applied_steps = (int64_t)c[4];
    *skipped_steps = (int64_t)c[2];
    return 0;
}
int gtav_dit_set_opt_step(gtav_dit* h, int64_t applied_steps, int64_t skipped_steps, void* stream) {
    GTAV_REQUIRE(h && h->tr.on && applied_steps >= 0 && applied_steps < (1 << 24) && skipped_steps >= 0, "set_opt_step: bad argument (the step count is kept as an exact fp32 integer: < 2^24)");
    float c[8];
    GTAV_CHECK_HIP(hipMemcpyAsync(c, h->tr.ctl, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GTAV_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    c[4] = (float)applied_steps; c[2] = (float)skipped_steps;
    GTAV_CHECK_HIP(hipMemcpy(h->tr.ctl, c, sizeof(c), hipMemcpyHostToDevice));
    return 0;
}
int gtav_dit_train_stats(gtav_dit* h, float* out4_host, void* stream) {
    GTAV_REQUIRE(h && out4_host && h->tr.on, "train_stats: bad argument");
    GTAV_CHECK_HIP(hipMemcpyAsync(out4_host, h->tr.ctl, 4 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GTAV_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// the handle's error words (gtav_dit::err_flag): copied back, cleared on the device; `words` gets 4 + n_groups ints
static int dit_read_err_words(gtav_dit* h, std::vector<int>& words, hipStream_t s) {
    words.assign(4 + h->n_groups, 0);
    GTAV_CHECK_HIP(hipMemcpyAsync(words.data(), h->err_flag, words.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    GTAV_CHECK_HIP(hipStreamSynchronize(s));
    GTAV_CHECK_HIP(hipMemsetAsync(h->err_flag, 0, words.size() * sizeof(int), s));
    return 0;
}

int gtav_dit_check(gtav_dit* h, void* stream) {
    GTAV_REQUIRE(h, "dit_check: null handle");
    std::vector<int> w;
    RET_IF(dit_read_err_words(h, w, (hipStream_t)stream));
    int flag = w[0];
    for (int g = 0; g < h->n_groups; ++g) flag |= w[4 + g];
    return report_err_flag(flag, "DiT");
}

static void dit_drop_graphs(gtav_dit* h) {
    for (auto& kv : h->graphs)
        if (kv.second) (void)hipGraphExecDestroy(kv.second);
    h->graphs.clear();
}

int gtav_dit_set_operand_dtype(gtav_dit* h, int32_t group, int32_t dtype) {
    GTAV_REQUIRE(h, "dit_set_operand_dtype: null handle");
    GTAV_REQUIRE(dtype == GTAV_OPERAND_F16 || dtype == GTAV_OPERAND_BF16, "dit_set_operand_dtype: dtype %d (0 = fp16, 1 = bf16)", dtype);
    GTAV_REQUIRE(group >= -1 && group < h->n_groups, "dit_set_operand_dtype: group %d outside [-1, %d)", group, h->n_groups);
    GTAV_REQUIRE(!h->tr.on || dtype == GTAV_OPERAND_F16, "dit_set_operand_dtype: a training handle keeps fp16 operands (its backward pass and loss scaling are fp16)");
    int changed = 0;
    for (int g = (group < 0 ? 0 : group); g < (group < 0 ? h->n_groups : group + 1); ++g) {
        if ((h->grp_bf16[g] != 0) == (dtype == GTAV_OPERAND_BF16)) continue;
        h->grp_bf16[g] = dtype == GTAV_OPERAND_BF16;
        changed += 1 + h->wt.set_dtype(g, dtype == GTAV_OPERAND_BF16);
    }
    if (changed) {
        // the weight images of the changed groups are of the other type now: the caller sends those weights again (gtav_dit_set_weight) and finalizes;
        // captured steps hold the other kernels; the temporal K/V caches of a switched half hold the other encoding
        h->finalized = false;
        h->kvrec.valid = false;
        dit_drop_graphs(h);
    }
    h->any_bf16 = false;
    for (unsigned char b : h->grp_bf16) h->any_bf16 |= b != 0;
    return 0;
}

int gtav_dit_get_operand_dtype(gtav_dit* h, int32_t group, int32_t* dtype) {
    GTAV_REQUIRE(h && dtype && group >= 0 && group < h->n_groups, "dit_get_operand_dtype: bad argument (group %d of %d)", group, h ? h->n_groups : 0);
    *dtype = h->grp_bf16[group] ? GTAV_OPERAND_BF16 : GTAV_OPERAND_F16;
    return 0;
}

int gtav_dit_operand_groups(gtav_dit* h, int32_t* n_groups) {
    GTAV_REQUIRE(h && n_groups, "dit_operand_groups: null argument");
    *n_groups = h->n_groups;
    return 0;
}

int gtav_dit_autorange(gtav_dit* h, int32_t* n_switched, void* stream) {
    GTAV_REQUIRE(h && n_switched, "dit_autorange: null argument");
    *n_switched = 0;
    std::vector<int> w;
    RET_IF(dit_read_err_words(h, w, (hipStream_t)stream));
    int other = w[0];
    for (int g = 0; g < h->n_groups; ++g) {
        other |= w[4 + g] & ~ERR_F16_SAT;
        if ((w[4 + g] & ERR_F16_SAT) && !h->grp_bf16[g]) {
            RET_IF(gtav_dit_set_operand_dtype(h, g, GTAV_OPERAND_BF16));
            *n_switched += 1;
        }
    }
    return report_err_flag(other, "DiT");
}

}  // extern "C"

// ================================================================================================
// ViT-VAE
// ================================================================================================
struct gtav_vae {
    gtav_vae_config cfg;
    int S, gh, gw, p, H, W, Kp, Npred, Lat, Mom, maxN, Mmax, Dmax, Hmax;
    Arena arena;
    WeightTable wt;
    struct Block { float *g1, *b1, *g2, *b2, *b_qkv, *b_proj, *b_fc1, *b_fc2; f16 *w_qkv, *w_proj, *w_fc1, *w_fc2; };
    std::vector<Block> enc, dec;
    f16 *w_patch, *w_quant, *w_post, *w_pred;
    float *b_patch, *b_quant, *b_post, *b_pred, *g_enc, *be_enc, *g_dec, *be_dec;
    RopeTable rope_e, rope_d;
    f16 *xp, *xn, *q, *k, *vt, *ao, *hbuf, *zin;
    float *resid, *po, *parts;
    int* err_flag = nullptr;
    size_t parts_rows = 0;
    bool finalized = false;
    const OperandOps* ops = &operand_ops(false);   // operand type of every 2-byte tensor of the handle (gtav_vae_set_operand_dtype; common.h "operand type")
    Profiler prof;   // gtav_vae_profile: per-class dispatch-attached events (bench.py's config4 roofline)
};

static int vae_blocks(gtav_vae* h, std::vector<gtav_vae::Block>& blocks, int dim, int heads, const RopeTable& rope, int N,
                      const float* g_last, const float* b_last, hipStream_t s) {
    // pre-LN blocks (model/vae.py:154-157); residual GEMMs are deferred into the next LayerNorm (see dit_forward_core),
    // the trailing enc_norm / dec_norm (g_last, b_last) consumes the last one and leaves LN(x) in h->xn.
    const int M = N * h->S, Hm = (int)(dim * h->cfg.mlp_ratio), Hm_pad = round_up(Hm, 128);
    GemmParams g;
    LnPending pend;
    bool have_pend = false;
    auto resid_gemm = [&](int cls, const f16* X, int ldx, const f16* Wt, int K, const float* bias) -> int {
        GemmParams q;
        memset(&q, 0, sizeof(q));
        q.X = X; q.ldx = ldx; q.W = Wt; q.M = M; q.N = dim; q.K = K; q.out = h->parts; q.ldo = dim;
        if (gemm_resid_inplace_ok(M, dim, K, 0)) {   // large M: in-place residual epilogue of the persistent loader-wave kernel (see dit_forward_core): no slab round trip
            q.out = h->resid; q.bias = bias;
            PROF(h, cls, s, h->ops->gemm(q, EPI_RESID, s));
            have_pend = false;
            return 0;
        }
        q.splitk = gemm_choose_splitk(M, dim, K);
        GTAV_REQUIRE((size_t)q.splitk * M * dim <= h->parts_rows * (size_t)h->Dmax, "split-K slabs exceed workspace");
        PROF(h, cls, s, h->ops->gemm(q, EPI_PARTIAL, s));
        memset(&pend, 0, sizeof(pend));
        pend.parts = h->parts; pend.nsplit = q.splitk; pend.slab_stride = (size_t)M * dim; pend.ld = dim; pend.bias = bias;
        have_pend = true;
        return 0;
    };
    for (auto& b : blocks) {
        PROF(h, PC_LN, s, h->ops->ln_affine(h->resid, dim, h->xn, dim, M, dim, b.g1, b.b1, have_pend ? &pend : nullptr, h->err_flag, s));
        have_pend = false;
        memset(&g, 0, sizeof(g));
        g.X = h->xn; g.ldx = dim; g.W = b.w_qkv; g.M = M; g.N = 3 * dim; g.K = dim; g.bias = b.b_qkv; g.D = dim; g.S = h->S;
        g.qkv_mode = QKV_SPATIAL; g.q = h->q; g.k = h->k; g.v = h->vt; g.rope_cs = rope.cs_dev; g.err_flag = h->err_flag;
        const bool qps = attn_spatial_wants_prescaled_q(h->S);   // long sequences: q leaves the epilogue in the exponent's unit of the flash attention kernel
        g.rope_cs_q = qps ? rope.csq_dev : nullptr;
        PROF(h, PC_QKV, s, h->ops->gemm(g, EPI_QKV, s));
        PROF(h, PC_ATTN_S, s, h->ops->attn_spatial(h->q, h->k, h->vt, h->ao, N, heads, h->S, s, qps));
        RET_IF(resid_gemm(PC_OUT, h->ao, dim, b.w_proj, dim, b.b_proj));
        PROF(h, PC_LN, s, h->ops->ln_affine(h->resid, dim, h->xn, dim, M, dim, b.g2, b.b2, have_pend ? &pend : nullptr, h->err_flag, s));
        have_pend = false;
        memset(&g, 0, sizeof(g));
        g.X = h->xn; g.ldx = dim; g.W = b.w_fc1; g.M = M; g.N = Hm; g.K = dim; g.bias = b.b_fc1; g.out = h->hbuf; g.ldo = Hm_pad; g.err_flag = h->err_flag;
        PROF(h, PC_FC1, s, h->ops->gemm(g, EPI_GELU_ERF, s));
        RET_IF(resid_gemm(PC_FC2, h->hbuf, Hm_pad, b.w_fc2, Hm_pad, b.b_fc2));
    }
    PROF(h, PC_LN, s, h->ops->ln_affine(h->resid, dim, h->xn, dim, M, dim, g_last, b_last, have_pend ? &pend : nullptr, h->err_flag, s));
    return 0;
}

extern "C" {

int gtav_vae_create(const gtav_vae_config* c, gtav_vae** out) {
    GTAV_REQUIRE(c && out, "vae_create: null argument");
    GTAV_REQUIRE(c->enc_dim % 128 == 0 && c->dec_dim % 128 == 0 && c->enc_dim / c->enc_heads == 64 && c->dec_dim / c->dec_heads == 64,
                 "VAE widths must be multiples of 128 with head_dim 64");
    GTAV_REQUIRE(c->input_height % c->patch_size == 0 && c->input_width % c->patch_size == 0, "VAE input not divisible by patch");
    GTAV_REQUIRE(c->latent_dim % 4 == 0 && c->latent_dim <= 64, "latent_dim=%d must be a multiple of 4, <= 64", c->latent_dim);
    gtav_vae* h = new gtav_vae();
    h->cfg = *c;
    h->p = c->patch_size; h->H = c->input_height; h->W = c->input_width; h->gh = h->H / h->p; h->gw = h->W / h->p; h->S = h->gh * h->gw;
    if (h->S % 8 != 0) {
        set_error("VAE seq_len=%d must be a multiple of 8", h->S);
        delete h;
        return 2;
    }
    h->Npred = 3 * h->p * h->p; h->Kp = round_up(h->Npred, 64); h->Lat = c->latent_dim; h->Mom = (c->use_variational ? 2 : 1) * h->Lat;
    h->maxN = c->max_frames_per_call > 0 ? c->max_frames_per_call : 8; h->Mmax = h->maxN * h->S;
    h->Dmax = c->enc_dim > c->dec_dim ? c->enc_dim : c->dec_dim;
    h->Hmax = round_up((int)(h->Dmax * c->mlp_ratio), 128);
    Arena& a = h->arena;
    WeightTable& wt = h->wt;
    int rc = 0;
#define A_(expr) do { if (!rc) rc = (expr); } while (0)
    const int De = c->enc_dim, Dd = c->dec_dim;
    A_(a.alloc_t(&h->w_patch, (size_t)round_up(De, 128) * h->Kp)); wt.add_f16("patch_embed.proj.weight", De, h->Npred, h->w_patch, round_up(De, 128), h->Kp);
    A_(a.alloc_t(&h->b_patch, De)); wt.add_f32("patch_embed.proj.bias", 1, De, h->b_patch, De);
    auto mk = [&](std::vector<gtav_vae::Block>& v, const char* prefix, int depth, int dim) {
        const int Hm = (int)(dim * c->mlp_ratio), Hm_pad = round_up(Hm, 128);
        v.resize(depth);
        for (int i = 0; i < depth && !rc; ++i) {
            gtav_vae::Block& b = v[i];
            char pre[64];
            snprintf(pre, sizeof(pre), "%s.%d.", prefix, i);
            std::string P_(pre);
            A_(a.alloc_t(&b.g1, dim)); wt.add_f32(P_ + "norm1.weight", 1, dim, b.g1, dim);
            A_(a.alloc_t(&b.b1, dim)); wt.add_f32(P_ + "norm1.bias", 1, dim, b.b1, dim);
            A_(a.alloc_t(&b.w_qkv, (size_t)round_up(3 * dim, 128) * dim)); wt.add_f16(P_ + "attn.qkv.weight", 3 * dim, dim, b.w_qkv, round_up(3 * dim, 128), dim);
            A_(a.alloc_t(&b.b_qkv, 3 * dim)); wt.add_f32(P_ + "attn.qkv.bias", 1, 3 * dim, b.b_qkv, 3 * dim);
            A_(a.alloc_t(&b.w_proj, (size_t)dim * dim)); wt.add_f16(P_ + "attn.proj.weight", dim, dim, b.w_proj, dim, dim);
            A_(a.alloc_t(&b.b_proj, dim)); wt.add_f32(P_ + "attn.proj.bias", 1, dim, b.b_proj, dim);
            A_(a.alloc_t(&b.g2, dim)); wt.add_f32(P_ + "norm2.weight", 1, dim, b.g2, dim);
            A_(a.alloc_t(&b.b2, dim)); wt.add_f32(P_ + "norm2.bias", 1, dim, b.b2, dim);
            A_(a.alloc_t(&b.w_fc1, (size_t)Hm_pad * dim)); wt.add_f16(P_ + "mlp.fc1.weight", Hm, dim, b.w_fc1, Hm_pad, dim);
            A_(a.alloc_t(&b.b_fc1, Hm_pad)); wt.add_f32(P_ + "mlp.fc1.bias", 1, Hm, b.b_fc1, Hm);
            A_(a.alloc_t(&b.w_fc2, (size_t)dim * Hm_pad)); wt.add_f16(P_ + "mlp.fc2.weight", dim, Hm, b.w_fc2, dim, Hm_pad);
            A_(a.alloc_t(&b.b_fc2, dim)); wt.add_f32(P_ + "mlp.fc2.bias", 1, dim, b.b_fc2, dim);
        }
    };
    mk(h->enc, "encoder", c->enc_depth, De);
    A_(a.alloc_t(&h->g_enc, De)); wt.add_f32("enc_norm.weight", 1, De, h->g_enc, De);
    A_(a.alloc_t(&h->be_enc, De)); wt.add_f32("enc_norm.bias", 1, De, h->be_enc, De);
    A_(a.alloc_t(&h->w_quant, (size_t)128 * De)); wt.add_f16("quant_conv.weight", h->Mom, De, h->w_quant, 128, De);
    A_(a.alloc_t(&h->b_quant, 128)); wt.add_f32("quant_conv.bias", 1, h->Mom, h->b_quant, h->Mom);
    A_(a.alloc_t(&h->w_post, (size_t)round_up(Dd, 128) * 64)); wt.add_f16("post_quant_conv.weight", Dd, h->Lat, h->w_post, round_up(Dd, 128), 64);
    A_(a.alloc_t(&h->b_post, Dd)); wt.add_f32("post_quant_conv.bias", 1, Dd, h->b_post, Dd);
    mk(h->dec, "decoder", c->dec_depth, Dd);
    A_(a.alloc_t(&h->g_dec, Dd)); wt.add_f32("dec_norm.weight", 1, Dd, h->g_dec, Dd);
    A_(a.alloc_t(&h->be_dec, Dd)); wt.add_f32("dec_norm.bias", 1, Dd, h->be_dec, Dd);
    A_(a.alloc_t(&h->w_pred, (size_t)round_up(h->Npred, 128) * Dd)); wt.add_f16("predictor.weight", h->Npred, Dd, h->w_pred, round_up(h->Npred, 128), Dd);
    A_(a.alloc_t(&h->b_pred, round_up(h->Npred, 128))); wt.add_f32("predictor.bias", 1, h->Npred, h->b_pred, h->Npred);
    h->rope_e.npos = h->rope_d.npos = h->S;
    A_(a.alloc_t(&h->rope_e.cos_dev, (size_t)h->S * 64)); wt.add_f32("tables.rope_enc_cos", h->S, 64, h->rope_e.cos_dev, 64, 0, false);
    A_(a.alloc_t(&h->rope_e.sin_dev, (size_t)h->S * 64)); wt.add_f32("tables.rope_enc_sin", h->S, 64, h->rope_e.sin_dev, 64, 0, false);
    A_(a.alloc_t(&h->rope_d.cos_dev, (size_t)h->S * 64)); wt.add_f32("tables.rope_dec_cos", h->S, 64, h->rope_d.cos_dev, 64, 0, false);
    A_(a.alloc_t(&h->rope_d.sin_dev, (size_t)h->S * 64)); wt.add_f32("tables.rope_dec_sin", h->S, 64, h->rope_d.sin_dev, 64, 0, false);
    A_(a.alloc_t(&h->rope_e.cs_dev, (size_t)h->S * 64)); A_(a.alloc_t(&h->rope_d.cs_dev, (size_t)h->S * 64));
    A_(a.alloc_t(&h->rope_e.csq_dev, (size_t)h->S * 64)); A_(a.alloc_t(&h->rope_d.csq_dev, (size_t)h->S * 64));
    const size_t Mx = round_up(h->Mmax, 128), Dm = h->Dmax;
    A_(a.alloc_t(&h->xp, Mx * h->Kp)); A_(a.alloc_t(&h->xn, Mx * Dm)); A_(a.alloc_t(&h->q, Mx * Dm)); A_(a.alloc_t(&h->k, Mx * Dm));
    A_(a.alloc_t(&h->vt, Mx * Dm)); A_(a.alloc_t(&h->ao, Mx * Dm)); A_(a.alloc_t(&h->hbuf, Mx * h->Hmax)); A_(a.alloc_t(&h->zin, Mx * 64));
    A_(a.alloc_t(&h->resid, Mx * Dm)); A_(a.alloc_t(&h->po, Mx * h->Npred));
    h->parts_rows = (2 * Mx * Dm > (size_t)(8u << 20) ? 2 * Mx * Dm : (size_t)(8u << 20)) / Dm;   // in rows of Dmax floats; two slabs at the largest M
    A_(a.alloc_t(&h->parts, h->parts_rows * Dm));
    A_(a.alloc_t(&h->err_flag, 4));
#undef A_
    if (rc) {
        delete h;
        return rc;
    }
    *out = h;
    return 0;
}

void gtav_vae_destroy(gtav_vae* h) { delete h; }

int gtav_vae_set_weight(gtav_vae* h, const char* name, const float* src, int64_t numel, void* stream) {
    GTAV_REQUIRE(h && name && src, "vae_set_weight: null argument");
    h->finalized = false;
    return h->wt.set(name, src, numel, (hipStream_t)stream);
}
int gtav_vae_get_weight(gtav_vae* h, const char* name, float* dst, int64_t numel, void* stream) {
    GTAV_REQUIRE(h && name && dst, "vae_get_weight: null argument");
    return h->wt.get(name, dst, numel, (hipStream_t)stream);
}

int gtav_vae_finalize(gtav_vae* h, void* stream) {
    GTAV_REQUIRE(h, "vae_finalize: null handle");
    RET_IF(h->wt.check_complete());
    GTAV_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    // model/vae.py:71-76: RotaryEmbedding(dim = head_dim // 4 = 16, pixel, max_freq = H*W) -> 8 freqs, 32 rotated dims
    auto build = [&](RopeTable& r, const char* cn, const char* sn_) -> int {
        if (h->wt.slots[cn].set && h->wt.slots[sn_].set) return 0;
        std::vector<float> l = linspace_f32(1.0f, (float)(h->S) / 2.0f, 8), fr(8), c, sn;
        for (int i = 0; i < 8; ++i) fr[i] = l[i] * (float)M_PI;
        build_axial_table(fr, h->gh, h->gw, c, sn);
        RET_IF(upload(r.cos_dev, c));
        return upload(r.sin_dev, sn);
    };
    RET_IF(build(h->rope_e, "tables.rope_enc_cos", "tables.rope_enc_sin"));
    RET_IF(build(h->rope_d, "tables.rope_dec_cos", "tables.rope_dec_sin"));
    RET_IF(launch_rope_interleave(h->rope_e.cos_dev, h->rope_e.sin_dev, h->rope_e.cs_dev, h->S, (hipStream_t)stream));
    RET_IF(launch_rope_interleave(h->rope_d.cos_dev, h->rope_d.sin_dev, h->rope_d.cs_dev, h->S, (hipStream_t)stream));
    for (RopeTable* r : {&h->rope_e, &h->rope_d}) {   // csq = 0 + (1/8 log2 e) cs
        GTAV_CHECK_HIP(hipMemsetAsync(r->csq_dev, 0, (size_t)h->S * 64 * sizeof(float), (hipStream_t)stream));
        RET_IF(launch_axpy_f32(r->csq_dev, r->cs_dev, kAttnQScale, (size_t)h->S * 64, (hipStream_t)stream));
    }
    GTAV_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    h->finalized = true;
    return 0;
}

int gtav_vae_encode(gtav_vae* h, const float* img, float in_scale, float in_shift, float* moments, int32_t N, void* stream) {
    GTAV_REQUIRE(h && img && moments, "vae_encode: null argument");
    GTAV_REQUIRE(h->finalized, "vae_encode: call gtav_vae_finalize first");
    GTAV_REQUIRE(N >= 1 && N <= h->maxN, "vae_encode: N=%d exceeds max_frames_per_call=%d", N, h->maxN);
    hipStream_t s = (hipStream_t)stream;
    const int De = h->cfg.enc_dim, M = N * h->S;
    PROF(h, PC_OTHER, s, h->ops->patchify(img, nullptr, N, 3, h->H, h->W, h->p, h->xp, h->Kp, in_scale, in_shift, h->err_flag, s));
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.X = h->xp; g.ldx = h->Kp; g.W = h->w_patch; g.M = M; g.N = De; g.K = h->Kp; g.bias = h->b_patch; g.out = h->resid; g.ldo = De;
    PROF(h, PC_OTHER, s, h->ops->gemm(g, EPI_F32, s));
    RET_IF(vae_blocks(h, h->enc, De, h->cfg.enc_heads, h->rope_e, N, h->g_enc, h->be_enc, s));
    memset(&g, 0, sizeof(g));
    g.X = h->xn; g.ldx = De; g.W = h->w_quant; g.M = M; g.N = h->Mom; g.K = De; g.bias = h->b_quant; g.out = moments; g.ldo = h->Mom;
    PROF(h, PC_OTHER, s, h->ops->gemm(g, EPI_F32, s));
    if (h->cfg.use_variational) PROF(h, PC_OTHER, s, launch_clamp_cols(moments, M, h->Mom, h->Lat, h->Mom, -30.f, 20.f, s));
    if (h->prof.on) {
        RET_IF(h->prof.begin(PC_EMPTY, s));
        RET_IF(h->prof.end(s));
    }
    return h->prof.collect(s);
}

int gtav_vae_decode(gtav_vae* h, const float* z, float z_scale, float* img, float out_scale, float out_shift, int32_t N,
                    void* stream) {
    GTAV_REQUIRE(h && z && img, "vae_decode: null argument");
    GTAV_REQUIRE(h->finalized, "vae_decode: call gtav_vae_finalize first");
    GTAV_REQUIRE(N >= 1 && N <= h->maxN, "vae_decode: N=%d exceeds max_frames_per_call=%d", N, h->maxN);
    hipStream_t s = (hipStream_t)stream;
    const int Dd = h->cfg.dec_dim, M = N * h->S;
    PROF(h, PC_OTHER, s, h->ops->convert_pad(z, h->Lat, M, h->Lat, h->zin, round_up(M, 128), 64, z_scale, 1, s));
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.X = h->zin; g.ldx = 64; g.W = h->w_post; g.M = M; g.N = Dd; g.K = 64; g.bias = h->b_post; g.out = h->resid; g.ldo = Dd;
    PROF(h, PC_OTHER, s, h->ops->gemm(g, EPI_F32, s));
    RET_IF(vae_blocks(h, h->dec, Dd, h->cfg.dec_heads, h->rope_d, N, h->g_dec, h->be_dec, s));
    memset(&g, 0, sizeof(g));
    g.X = h->xn; g.ldx = Dd; g.W = h->w_pred; g.M = M; g.N = h->Npred; g.K = Dd; g.bias = h->b_pred; g.out = h->po; g.ldo = h->Npred;
    PROF(h, PC_OTHER, s, h->ops->gemm(g, EPI_F32, s));
    PROF(h, PC_OTHER, s, launch_unpatchify(h->po, h->Npred, img, N, 3, h->H, h->W, h->p, 1, out_scale, out_shift, s));
    if (h->prof.on) {
        RET_IF(h->prof.begin(PC_EMPTY, s));
        RET_IF(h->prof.end(s));
    }
    return h->prof.collect(s);
}

int gtav_vae_profile(gtav_vae* h, int32_t enable) {
    GTAV_REQUIRE(h, "vae_profile: null handle");
    h->prof.on = enable != 0;
    h->prof.used = 0;
    for (int i = 0; i < PC_COUNT; ++i) { h->prof.ms[i] = 0; h->prof.n[i] = 0; }
    return 0;
}
int gtav_vae_profile_read(gtav_vae* h, double* ms_by_class, int64_t* launches_by_class) {
    GTAV_REQUIRE(h && ms_by_class && launches_by_class, "vae_profile_read: null argument");
    for (int i = 0; i < PC_COUNT; ++i) { ms_by_class[i] = h->prof.ms[i]; launches_by_class[i] = h->prof.n[i]; }
    return 0;
}

int gtav_vae_set_operand_dtype(gtav_vae* h, int32_t dtype) {
    GTAV_REQUIRE(h, "vae_set_operand_dtype: null handle");
    GTAV_REQUIRE(dtype == GTAV_OPERAND_F16 || dtype == GTAV_OPERAND_BF16, "vae_set_operand_dtype: dtype %d (0 = fp16, 1 = bf16)", dtype);
    const bool bf = dtype == GTAV_OPERAND_BF16;
    if (h->ops->bf16 != bf) {
        h->ops = &operand_ops(bf);
        h->wt.set_dtype(-1, bf);      // every weight image is of the other type now: send the weights again, then finalize
        h->finalized = false;
    }
    return 0;
}
int gtav_vae_get_operand_dtype(gtav_vae* h, int32_t* dtype) {
    GTAV_REQUIRE(h && dtype, "vae_get_operand_dtype: null argument");
    *dtype = h->ops->bf16 ? GTAV_OPERAND_BF16 : GTAV_OPERAND_F16;
    return 0;
}

int gtav_vae_check(gtav_vae* h, void* stream) {
    GTAV_REQUIRE(h, "vae_check: null handle");
    int flag = 0;
    GTAV_CHECK_HIP(hipMemcpyAsync(&flag, h->err_flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GTAV_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    GTAV_CHECK_HIP(hipMemsetAsync(h->err_flag, 0, sizeof(int), (hipStream_t)stream));
    return report_err_flag(flag, "VAE");
}

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
