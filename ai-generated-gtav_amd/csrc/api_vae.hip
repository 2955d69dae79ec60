// C-ABI of the ViT-VAE handle (include/gtav_amd.h): create / weights / finalize, encode, decode, profile, check, operand type.
#include "api_internal.h"

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
    A_(a.alloc_t(&h->err_flag, 4)); wt.err_words = h->err_flag;
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
    PROF(h, PC_OTHER, s, h->ops->convert_pad(z, h->Lat, M, h->Lat, h->zin, round_up(M, 128), 64, z_scale, 1, s, h->err_flag));
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

}  // extern "C"
