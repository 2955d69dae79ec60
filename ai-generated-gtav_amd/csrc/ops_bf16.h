// The bf16-operand twins of the launchers api_dit.hip / api_vae.hip dispatch per operand type (common.h "operand type"): gemm.hip, attention.hip and elementwise.hip compiled a second
// time with -DGTAV_BF16_OPERANDS -Dgtav=gtav_bf16 (csrc/build.sh).  Same kernels, same argument meaning as gemm.h / ops.h; `f16*` there is `__bf16*` here and the
// parameter structs are the twin namespace's own (identical layout: api.hip passes its gtav::GemmParams / gtav::LnPending through a reference cast).
// Keep the signatures in step with ops.h / gemm.h: a mismatch is a link error, never a silent one.
#pragma once
#include "ops.h"

namespace gtav_bf16 {
struct GemmParams;
struct LnPending;
int launch_gemm(const GemmParams& p, int epi, hipStream_t stream);
int launch_ln_modulate(float* x, int ldx, __bf16* out, int ldo, int M, int D, const float* shift, const float* scale, int mod_stride, const int* rows,
                       int rows_per_mod, const LnPending* pend, int* err_flag, hipStream_t stream);
int launch_ln_affine(float* x, int ldx, __bf16* out, int ldo, int M, int D, const float* gamma, const float* beta, const LnPending* pend, int* err_flag,
                     hipStream_t stream);
int launch_patchify(const float* img, const int* frame_index, int NB, int C, int H, int W, int p, __bf16* out, int ldo, float a, float b, int* err_flag,
                    hipStream_t stream);
int launch_convert_pad_f16(const float* src, int lds, int R, int C, __bf16* dst, int Rp, int Cp, float scale, int tiled, hipStream_t stream, int* err_flag);
int launch_unpad_f16_to_f32(const __bf16* src, int lds, int R, int C, float* dst, int tiled, hipStream_t stream);
int launch_attn_spatial(const __bf16* Q, const __bf16* K, const __bf16* Vt, __bf16* O, int NB, int heads, int S, hipStream_t stream, bool q_prescaled);
int launch_attn_temporal(const __bf16* q, const __bf16* kv, __bf16* O, int B, int P, int D, int Tq, int t0, int Tmax, hipStream_t stream);
void set_error(const char* fmt, ...);     // defined in api.hip: forwards to gtav::set_error (the twin objects report through the same thread-local string)
const char* last_error();
}  // namespace gtav_bf16

namespace gtav {

// One set of launchers per operand type; the signatures are the fp16 ones (the bf16 set casts the 2-byte pointers).
struct OperandOps {
    int (*gemm)(const GemmParams& p, int epi, hipStream_t stream);
    int (*ln_modulate)(float* x, int ldx, f16* out, int ldo, int M, int D, const float* shift, const float* scale, int mod_stride, const int* rows, int rows_per_mod,
                       const LnPending* pend, int* err_flag, hipStream_t stream);
    int (*ln_affine)(float* x, int ldx, f16* out, int ldo, int M, int D, const float* gamma, const float* beta, const LnPending* pend, int* err_flag, hipStream_t stream);
    int (*patchify)(const float* img, const int* frame_index, int NB, int C, int H, int W, int p, f16* out, int ldo, float a, float b, int* err_flag, hipStream_t stream);
    int (*convert_pad)(const float* src, int lds, int R, int C, f16* dst, int Rp, int Cp, float scale, int tiled, hipStream_t stream, int* err_flag);
    int (*unpad)(const f16* src, int lds, int R, int C, float* dst, int tiled, hipStream_t stream);
    int (*attn_spatial)(const f16* Q, const f16* K, const f16* Vt, f16* O, int NB, int heads, int S, hipStream_t stream, bool q_prescaled);
    int (*attn_temporal)(const f16* q, const f16* kv, f16* O, int B, int P, int D, int Tq, int t0, int Tmax, hipStream_t stream);
    bool bf16;
};
const OperandOps& operand_ops(bool bf16);   // api.hip

}  // namespace gtav
