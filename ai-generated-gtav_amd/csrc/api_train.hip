// C-ABI of the DiT training step (include/gtav_amd.h): forward with saved activations, backward, AdamW, optimizer state.
#include "api_internal.h"

static int g_dw_grouped = GTAV_ENV_INT("GTAV_DW_GROUPED", 1);   // experiments build: 0 = one launch per weight gradient (A/B runs)
static int g_fuse_gelu_fwd = GTAV_ENV_INT("GTAV_FUSE_GELU_FWD", 1); // experiments build: 0 = h = GELU(u) by the flat elementwise kernel behind fc1 (A/B runs)
static int g_fuse_gelu = GTAV_ENV_INT("GTAV_FUSE_GELU_BWD", 1); // experiments build: 0 = gelu_bwd and the fc1 bias column sums as two launches (A/B runs)
static int g_fuse_ln = GTAV_ENV_INT("GTAV_FUSE_LN_BWD", 1);     // experiments build: 0 = ln_mod_bwd and frame_reduce_ln as two launches (A/B runs)
static int g_fuse_gate = GTAV_ENV_INT("GTAV_FUSE_GATE", 1);     // experiments build: 0 = gate_bwd, frame_reduce_gate and the bias column sums as three launches (A/B runs)
static int g_dw_tn = GTAV_ENV_INT("GTAV_DW_TN", 1);             // experiments build: 0 = transposed operand copies in front of the grouped launch (A/B runs)

extern "C" {

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

}  // extern "C"
