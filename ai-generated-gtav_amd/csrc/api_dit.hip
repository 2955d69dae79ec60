// C-ABI of the DiT handle (include/gtav_amd.h): create / weights / finalize, forward, the fused sampler step and its captured graph, profile, check, operand type.
#include "api_internal.h"

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

// the fused spatial to_qkv + attention launch: from 5 frames on (measured from 80 blocks up; a context-cached batch-1 step is 16 blocks: the split path's skinny kernels serve it)
static bool fused_spatial_ok(int M, int D, int P) { return gemm_qkvs_attn_ok(M, D, P) && M / P >= 5 && (M / P) * (D / 64) >= 80; }

// LayerNorm fold: which seams run folded at M tokens (seam A = out-proj -> fc1, seam B = fc2 -> next to_qkv / final projection)
static void fold_policy(const gtav_dit* h, int M, bool& fa, bool& fb) {
    const gtav_dit::Fold& f = h->fold;
    fa = fb = false;
    if (!f.ok || f.mode == 0 || h->tr.on || h->fuse_tattn || h->any_bf16) return;   // (the fused spatial launch steps aside per half-block: !folded_in)
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
            const bool fused2 = !h->tr.on && w2.w_qkv_hm && !h->ops(hb2).bf16 &&
                                ((hb2 & 1) ? h->fuse_tattn && gemm_qkvt_attn_ok(M, D, P, Tq, t0) : h->fuse_sattn && fused_spatial_ok(M, D, P));
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
            const bool fused_s = hf == 0 && h->fuse_sattn && !h->tr.on && !ops.bf16 && !folded_in && w.w_qkv_hm && fused_spatial_ok(M, D, P);
            if (fused_t) {
                g.W = w.w_qkv_hm; g.qkv_mode = QKV_TEMPORAL; g.k = h->kvcache[l]; g.v = h->kvcache[l]; g.out = h->ao; g.ldo = D;
                g.Tq = Tq; g.t0 = t0; g.Tmax = h->maxT; g.rope_cs = h->rope_t.cs_dev;
                PROF(h, PC_ATTN_T, s, launch_gemm_qkvt_attn(g, s));   // profiled as the attention class: one class = one kernel (gtav_dit_fused_launches tells a reader which it was)
            } else if (fused_s) {
                g.W = w.w_qkv_hm; g.qkv_mode = QKV_SPATIAL; g.out = h->ao; g.ldo = D; g.rope_cs = h->rope_s.cs_dev;
                PROF(h, PC_ATTN_S, s, launch_gemm_qkvs_attn(g, s));
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
    h->fuse_sattn = h->P == 144 && D % 256 == 0;   // gtav_dit_set_fused_spatial(h, 0) selects the two-kernel path
    for (int l = 0; l < h->L && !rc; ++l)
        for (int hf = 0; hf < 2; ++hf) {
            gtav_dit::Half& w = h->halves[l * 2 + hf];
            char pre[64];
            snprintf(pre, sizeof(pre), "blocks.%d.%c_", l, hf == 0 ? 's' : 't');
            std::string P_(pre);
            const int grp = l * 2 + hf;
            A_(a.alloc_t(&w.w_qkv, (size_t)3 * D * D)); wt.add_f16(P_ + "attn.to_qkv.weight", 3 * D, D, w.w_qkv, 3 * D, D, grp);
            w.w_qkv_hm = nullptr;   // temporal halves: allocated by gtav_dit_set_fused_temporal(h, 1)
            if (hf == 0 && h->P == 144 && D % 256 == 0) A_(a.alloc_t(&w.w_qkv_hm, (size_t)3 * D * D));   // spatial halves at 144 tokens per frame: the fused launch is the default
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
    wt.err_words = h->err_flag; wt.err_per_group = true;   // a clamped GEMM weight is reported in its operand group's word
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
    for (size_t hb = 0; hb < h->halves.size(); ++hb) {
        gtav_dit::Half& w = h->halves[hb];
        if (w.w_qkv_hm) RET_IF(launch_qkv_head_major(w.w_qkv, w.w_qkv_hm, D, s, (hb & 1) ? 0 : 1));
    }
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

int gtav_dit_set_fused_spatial(gtav_dit* h, int32_t enable) {
    GTAV_REQUIRE(h, "dit_set_fused_spatial: null handle");
    if (h->fuse_sattn != (enable != 0)) {   // captured sampler steps contain the other kernel sequence
        for (auto& kv : h->graphs)
            if (kv.second) (void)hipGraphExecDestroy(kv.second);
        h->graphs.clear();
    }
    if (enable && h->P == 144 && h->D % 256 == 0) {
        // first enable: head-major copies of the spatial to_qkv weights (3 D^2 halves per block), as for the temporal switch
        for (int l = 0; l < h->L; ++l) {
            gtav_dit::Half& w = h->halves[l * 2];
            if (w.w_qkv_hm) continue;
            RET_IF(h->arena.alloc_t(&w.w_qkv_hm, (size_t)3 * h->D * h->D));
            if (h->finalized) RET_IF(launch_qkv_head_major(w.w_qkv, w.w_qkv_hm, h->D, nullptr, 1));
        }
        if (h->finalized) GTAV_CHECK_HIP(hipDeviceSynchronize());
    }
    h->fuse_sattn = enable != 0;
    return 0;
}

int gtav_dit_fused_launches(gtav_dit* h, int32_t B, int32_t T, int32_t t0) {
    if (!h || B < 1 || T < 1) return 0;
    const int M = B * T * h->P;
    int mask = 0;
    for (int hb = 0; hb < 2 * h->L; ++hb) {
        const gtav_dit::Half& w = h->halves[hb];
        if (h->tr.on || h->ops(hb).bf16 || !w.w_qkv_hm) continue;
        if (!(hb & 1) && h->fuse_sattn && fused_spatial_ok(M, h->D, h->P)) mask |= 1;
        if ((hb & 1) && h->fuse_tattn && gemm_qkvt_attn_ok(M, h->D, h->P, T, t0)) mask |= 2;
    }
    return mask;
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
