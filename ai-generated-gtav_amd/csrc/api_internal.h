// Internals shared by the C-ABI translation units (api.hip: error plumbing, operand-type launcher tables, elementwise / kernel-level entry points, timer
// calibration; api_dit.hip: the DiT handle — create, weights, forward, sampler step, profile, check / operand type; api_train.hip: the DiT training step;
// api_vae.hip: the VAE handle).  Everything here is inline / header-only: handle bookkeeping (Arena, WeightTable), host-side table builders, the in-situ
// profiler and the DiT handle itself, which the inference and the training file both work on.
#pragma once
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

using namespace gtav;

#define RET_IF(expr)            \
    do {                        \
        int _rc = (expr);       \
        if (_rc) return _rc;    \
    } while (0)

namespace gtav {
namespace api {

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
    // device error words of the owning handle (word 0; a DiT handle: + 4 + group for the f16 slots of an operand group): a GEMM weight beyond the fp16 range is
    // clamped by the conversion AND raises ERR_F16_SAT there, so that check() reports it and gtav_dit_autorange moves the group to bf16
    int* err_words = nullptr;
    bool err_per_group = false;
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
        if (sl.kind == SLOT_F16_PAD) {
            int* ef = err_words ? err_words + (err_per_group && sl.group >= 0 ? 4 + sl.group : 0) : nullptr;
            RET_IF(operand_ops(sl.bf16).convert_pad(src, sl.C, sl.R, sl.C, (f16*)sl.dst, sl.Rp, sl.Cp, 1.0f, 1, s, ef));
        }
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
inline std::vector<float> linspace_f32(float start, float end, int steps) {
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

inline int upload(float* dst, const std::vector<float>& v) {
    GTAV_CHECK_HIP(hipMemcpy(dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// axial "pixel" RoPE table (rotary_embedding_torch.py:290-317): per position (r, c) of a gh x gw grid,
// head dims [0, 2F) rotate with the row angle, [2F, 4F) with the column angle (each freq repeated twice),
// remaining dims are identity.
inline void build_axial_table(const std::vector<float>& freqs, int gh, int gw, std::vector<float>& c, std::vector<float>& s) {
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
inline int report_err_flag(int flag, const char* who) {
    GTAV_REQUIRE(!(flag & ERR_TIMESTEP), "%s: a timestep outside [0, 999] was passed", who);
    GTAV_REQUIRE(!(flag & ERR_NONFINITE), "%s: a NaN or inf was found in the input tensor", who);
    GTAV_REQUIRE(!(flag & ERR_F16_SAT), "%s: an activation exceeded the fp16 range (|x| > 65504) and was saturated; results since the "
                 "last check are finite but clipped (the reference runs this path in bf16, which has fp32 range)", who);
    return 0;
}

}  // namespace api
}  // namespace gtav
using namespace gtav::api;

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
    struct Half { f16 *w_qkv, *w_out, *w_fc1, *w_fc2; float *b_out, *b_fc1, *b_fc2; f16* w_qkv_hm; };   // w_qkv_hm: head-major rows for the fused QKV + attention GEMMs (temporal halves: launch_qkv_head_major mode 0, spatial halves: mode 1), made by finalize / the set_fused switches
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
    // steps of 5 or more frames of 144 tokens: spatial QKV projection + spatial attention in one launch (gemm.hip gemm_qkvs_attn_kernel; bit-identical to the split path;
    // round 6: faster at every size measured, 80 ... 1280 blocks).  ON by default where the geometry allows (gtav_dit_create allocates the head-major weight copies of the
    // spatial halves); gtav_dit_set_fused_spatial() is the switch; training handles and bf16 half-blocks keep the split path.
    bool fuse_sattn = false;
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

