// Torch-free driver for rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) on the four GEMM classes of a DiT half-block and the fused spatial to_qkv + attention launch, through the
// C-ABI exactly as the model launches them: to_qkv (spatial layout + RoPE epilogue), out-proj and fc2 (split-K slabs, the model's K-slice
// heuristic), fc1 (GELU epilogue); M = 720 (batch-1 window step), M = 1152 (batch-8 context-cached step), M = 5760 (batch-8 window step) and M = 11 520 (the training batch);
// then the ViT-VAE's four at M = 46 080 (encode of the trainer's 80 frames: qkv with bias at S = 576, un-gated in-place projection / fc2, erf-GELU fc1); rotating weight
// buffers.  Dispatch order is fixed — for M in {720, 1152, 5760, 11520}: qkv, out, fc1, fc2, qkvs (the fused launch), then the VAE's qkv, proj, fc1, fc2, `iters` launches each — and
// tools/gemm_traffic.py segments the counter rows by that order.
//   hipcc -O2 tools/gemm_pmc.cpp -Iinclude -L ai-generated-gtav_amd -lgtav_amd -Wl,-rpath,'$ORIGIN/../ai-generated-gtav_amd' -o tools/gemm_pmc
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./tools/gemm_pmc 16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gtav_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define GK(x) do { if (x) { fprintf(stderr, "%s failed: %s\n", #x, gtav_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const int D = 1024, H = 4096, copies = 8, iters = argc > 1 ? atoi(argv[1]) : 16;
    const int NM = 4;
    const int Ms[NM] = {720, 1152, 5760, 11520};   // 11 520: the training batch's window (config4)
    std::vector<unsigned short> host((size_t)H * D);
    for (size_t i = 0; i < host.size(); ++i) host[i] = 0x2000 + (unsigned short)((i * 2654435761u) >> 20 & 0x3ff);  // small fp16 values
    std::vector<void*> w(copies);
    for (int c = 0; c < copies; ++c) {
        CK(hipMalloc(&w[c], (size_t)H * D * 2));
        CK(hipMemcpy(w[c], host.data(), (size_t)H * D * 2, hipMemcpyHostToDevice));
    }
    std::vector<void*> w_hm(copies);   // the to_qkv weight ([3072][1024] of each buffer) in the fused spatial launch's row order
    for (int c = 0; c < copies; ++c) {
        CK(hipMalloc(&w_hm[c], (size_t)3 * D * D * 2));
        GK(gtav_op_qkv_head_major_spatial(w[c], w_hm[c], D, nullptr));
    }
    CK(hipDeviceSynchronize());
    float *bias, *cs;
    CK(hipMalloc((void**)&bias, H * 4));
    CK(hipMemset(bias, 0, H * 4));
    CK(hipMalloc((void**)&cs, 144 * 64 * 4));
    {
        std::vector<float> one(144 * 64);
        for (size_t i = 0; i < one.size(); ++i) one[i] = (i & 1) ? 0.f : 1.f;     // (cos, sin) = (1, 0): identity rotation
        CK(hipMemcpy(cs, one.data(), one.size() * 4, hipMemcpyHostToDevice));
    }
    for (int mi = 0; mi < NM; ++mi) {
        const int M = Ms[mi], Mp = (M + 127) / 128 * 128;
        void *x, *xh, *q, *k, *v, *hb, *parts;
        CK(hipMalloc(&x, (size_t)Mp * D * 2));
        CK(hipMalloc(&xh, (size_t)Mp * H * 2));
        CK(hipMemcpy(x, host.data(), (size_t)Mp * D * 2 < host.size() * 2 ? (size_t)Mp * D * 2 : host.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(xh, 0, (size_t)Mp * H * 2));
        CK(hipMalloc(&q, (size_t)Mp * D * 2)); CK(hipMalloc(&k, (size_t)Mp * D * 2)); CK(hipMalloc(&v, (size_t)Mp * D * 2));
        CK(hipMalloc(&hb, (size_t)Mp * H * 2));
        CK(hipMalloc(&parts, (size_t)8 * Mp * D * 4));
        const int sk_out = gtav_op_gemm_choose_splitk(M, D, D), sk_fc2 = gtav_op_gemm_choose_splitk(M, D, H);
        const bool ip_out = gtav_op_gemm_resid_inplace(M, D, D), ip_fc2 = gtav_op_gemm_resid_inplace(M, D, H);   // large M: in-place gated residual epilogue
        float* gate;
        CK(hipMalloc((void**)&gate, (size_t)(M / 144 + 1) * D * 4));
        CK(hipMemset(gate, 0, (size_t)(M / 144 + 1) * D * 4));
        for (int it = 0; it < iters; ++it) GK(gtav_op_gemm_qkv(x, D, w[it % copies], nullptr, M, D, 0, q, k, v, 144, 0, 0, 0, cs, nullptr));
        for (int it = 0; it < iters; ++it)
            GK(ip_out ? gtav_op_gemm_f16(x, D, w[it % copies], bias, parts, D, M, D, D, 4, gate, D, 144, nullptr)
                      : gtav_op_gemm_f16(x, D, w[it % copies], nullptr, parts, D, M, D, D, 6, nullptr, sk_out, 1, nullptr));
        for (int it = 0; it < iters; ++it) GK(gtav_op_gemm_f16(x, D, w[it % copies], bias, hb, H, M, H, D, 2, nullptr, 0, 1, nullptr));
        for (int it = 0; it < iters; ++it)
            GK(ip_fc2 ? gtav_op_gemm_f16(xh, H, w[it % copies], bias, parts, D, M, D, H, 4, gate, D, 144, nullptr)
                      : gtav_op_gemm_f16(xh, H, w[it % copies], nullptr, parts, D, M, D, H, 6, nullptr, sk_fc2, 1, nullptr));
        // the fused spatial to_qkv + attention launch (frames of 144 tokens: every M here is a whole number of them, 5 or more): weight rows in its own order
        for (int it = 0; it < iters; ++it) GK(gtav_op_gemm_qkvs_attn(x, w_hm[it % copies], M, D, 144, cs, q, nullptr));
        CK(hipDeviceSynchronize());
        printf("M=%d: %d launches each of qkv (N=3072 K=1024), out (N=1024 K=1024, %d K slices), fc1 (N=4096 K=1024), fc2 (N=1024 K=4096, %d K slices), qkvs (fused spatial to_qkv + attention)\n", M, iters,
               ip_out ? 0 : sk_out, ip_fc2 ? 0 : sk_fc2);
        CK(hipFree(gate));
        CK(hipFree(x)); CK(hipFree(xh)); CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(v)); CK(hipFree(hb)); CK(hipFree(parts));
    }
    {   // ---- the ViT-VAE encoder's GEMMs at the trainer's batch (80 frames x 576 tokens) ----
        const int M = 46080, Mp = M;
        void *x, *xh, *q, *k, *v, *hb;
        float *res, *cs576;
        CK(hipMalloc(&x, (size_t)Mp * D * 2)); CK(hipMalloc(&xh, (size_t)Mp * H * 2));
        CK(hipMemset(x, 0x20, (size_t)Mp * D * 2)); CK(hipMemset(xh, 0, (size_t)Mp * H * 2));
        CK(hipMalloc(&q, (size_t)Mp * D * 2)); CK(hipMalloc(&k, (size_t)Mp * D * 2)); CK(hipMalloc(&v, (size_t)Mp * D * 2));
        CK(hipMalloc(&hb, (size_t)Mp * H * 2));
        CK(hipMalloc((void**)&res, (size_t)Mp * D * 4)); CK(hipMemset(res, 0, (size_t)Mp * D * 4));
        CK(hipMalloc((void**)&cs576, 576 * 64 * 4));
        {
            std::vector<float> one(576 * 64);
            for (size_t i = 0; i < one.size(); ++i) one[i] = (i & 1) ? 0.f : 1.f;
            CK(hipMemcpy(cs576, one.data(), one.size() * 4, hipMemcpyHostToDevice));
        }
        for (int it = 0; it < iters; ++it) GK(gtav_op_gemm_qkv(x, D, w[it % copies], bias, M, D, 0, q, k, v, 576, 0, 0, 0, cs576, nullptr));
        for (int it = 0; it < iters; ++it) GK(gtav_op_gemm_f16(x, D, w[it % copies], bias, res, D, M, D, D, 4, nullptr, 0, 0, nullptr));
        for (int it = 0; it < iters; ++it) GK(gtav_op_gemm_f16(x, D, w[it % copies], bias, hb, H, M, H, D, 3, nullptr, 0, 1, nullptr));
        for (int it = 0; it < iters; ++it) GK(gtav_op_gemm_f16(xh, H, w[it % copies], bias, res, D, M, D, H, 4, nullptr, 0, 0, nullptr));
        CK(hipDeviceSynchronize());
        printf("VAE M=%d: %d launches each of qkv, proj (in place), fc1 (erf-GELU), fc2 (in place)\n", M, iters);
    }
    return 0;
}
