// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of gtav_amd.
// Wavefront = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_ext.h>
#include <stdint.h>

// ---- operand type of the 2-byte GEMM / attention operands -------------------------------------------------------------------
// The product's default is fp16 (10 mantissa bits: 6-9e-4 relative L2 per forward against the fp32 oracle).  The reference itself runs this path
// under bf16 autocast (generate.py:125-127, train_dit.py:190-198), whose exponent range is fp32's: a checkpoint with activations beyond +-65504 is clamped
// (and flagged) in fp16.  So gemm.hip, attention.hip and elementwise.hip are compiled a SECOND time with -DGTAV_BF16_OPERANDS -Dgtav=gtav_bf16 (csrc/build.sh):
// the same kernels, tiles, layouts and launch heuristics with `f16` = __bf16 — v_mfma_f32_16x16x32_bf16, v_cvt_pk_bf16_f32 (round to nearest even),
// v_dot2c_f32_bf16 — in a namespace of their own (the macro renames `gtav` for the whole translation unit).  api_dit.hip picks a set of launchers per half-block
// (gtav_dit_set_operand_dtype / the automatic switch of gtav_dit_autorange); the fp16 objects are unchanged by this, bit for bit.
#ifdef GTAV_BF16_OPERANDS
#define GTAV_F16_T __bf16
#else
#define GTAV_F16_T _Float16
#endif

namespace gtav {

typedef GTAV_F16_T f16;
typedef GTAV_F16_T f16x4 __attribute__((ext_vector_type(4)));
typedef GTAV_F16_T f16x8 __attribute__((ext_vector_type(8)));
typedef GTAV_F16_T f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// D = A (16 x 32) x B (32 x 16) + C on the matrix pipe, fp32 accumulate: the one MFMA shape of the 2-byte operand kernels
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c, int, int, int) {
#ifdef GTAV_BF16_OPERANDS
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}
// c + a[0] b[0] + a[1] b[1] in fp32 (v_dot2_f32_f16 / v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2acc(f16x2 a, f16x2 b, float c, bool) {
#ifdef GTAV_BF16_OPERANDS
    return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false);
#else
    return __builtin_amdgcn_fdot2(a, b, c, false);
#endif
}

constexpr int WAVE = 64;

// ---- kernel launch with optional dispatch-attached timing events (profiler hook) ----
// When g_launch_ev[0] is set (api_internal.h's Profiler, one class at a time), the next launch of this thread attaches the start /
// stop events to its own dispatch packet (hipExtLaunchKernel) and clears the slot: in-situ kernel times then carry no
// marker-packet overhead and agree with rocprofv3's kernel trace (round 6, gtav_timer_calibrate on a spin kernel of known device duration: the pair reads
// device time + 0.60 us, rocprofv3 device time + 0.62 us — profiles/timer_calibration.json).
}  // namespace gtav
namespace gtav_shared { extern thread_local hipEvent_t g_launch_ev[2]; }   // one slot for the fp16 objects and their bf16 twins (defined in api.hip)
namespace gtav {
using gtav_shared::g_launch_ev;
#define GTAV_LAUNCH(kern, grid, block, shmem, stream, ...)                                                              \
    do {                                                                                                                \
        if (gtav_shared::g_launch_ev[0]) {                                                                                     \
            hipExtLaunchKernelGGL(kern, grid, block, shmem, stream, gtav_shared::g_launch_ev[0], gtav_shared::g_launch_ev[1], 0, __VA_ARGS__); \
            gtav_shared::g_launch_ev[0] = nullptr;                                                                             \
        } else {                                                                                                        \
            hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);                                          \
        }                                                                                                               \
    } while (0)

// ---- error plumbing (C-ABI functions never throw; they return an int and set this string) ----
void set_error(const char* fmt, ...);
const char* last_error();

#define GTAV_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            gtav::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

#define GTAV_REQUIRE(cond, ...)                                                           \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            gtav::set_error(__VA_ARGS__);                                                 \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

// ---- experiment knobs ---------------------------------------------------------------------------
// The shipped library (build.sh) reads NO environment variables and carries no result-changing switches.  A second
// build with -DGTAV_EXPERIMENTS (build.sh exp -> libgtav_amd_exp.so, used by tools/ only) compiles in the timing
// experiments: GEMM debug bits (skip fills / skip MFMA: WRONG results), forced shapes, LayerNorm / attention variants.
#ifdef GTAV_EXPERIMENTS
#define GTAV_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define GTAV_DBG(p, bits) ((p).debug & (bits))
#else
#define GTAV_ENV_INT(name, dflt) (dflt)
#define GTAV_DBG(p, bits) false
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---- tile-major fp16 operand layout of the GEMM ---------------------------------------------
// A [rows][K] fp16 GEMM operand (activations X or weights W) is stored as contiguous 16 KiB tiles of
// 128 rows x 64 k, tile (r / 128, k / 64) at ((r / 128) * (K / 64) + k / 64) * 8192 halves, and INSIDE a tile in
// the exact LDS image the kernel wants: row-major 128-byte rows whose 16-byte chunk c sits at c ^ (row & 7)
// (the ds_read_b128 bank swizzle).  Every direct-to-LDS instruction then copies 1 KiB of contiguous memory
// (no power-of-two row stride -> no L2-channel hot spots) and a K-panel of a tile row is one linear stream.
// rows are padded to 128, K to 64.  Offset in halves of element (r, k):
__host__ __device__ __forceinline__ size_t tiled_off(int r, int k, int K) {
    return ((size_t)(r >> 7) * (size_t)(K >> 6) + (size_t)(k >> 6)) * 8192 + (size_t)((r & 127) * 64) +
           (size_t)(((((k >> 3) & 7) ^ (r & 7)) << 3) + (k & 7));
}

// ---- L2 prefetch of a weight the NEXT launch will stream (docs/LABNOTES.md 4.10) ---------------------------------------------
// A batch-1 step streams all 1.2 GB of weights from HBM, so every GEMM launch starts on a cold W.  The launch in front of it (whose own memory
// traffic is small) touches ONE dword per 128-byte line of the slice of W that the blocks of each XCD will stream: blocks with equal
// blockIdx % 8 share an XCD and its L2 (tools/xcd_start.hip), clean lines survive a kernel boundary (tools/l2_persist.hip: 0.56 us per 32 KiB
// block from the same XCD's L2, 1.3 us from another XCD's / the Infinity Cache, 3.1 us from HBM), and the consumer's tile map gives XCD x a
// contiguous eighth of its (K slice, row panel) order.  W is tile-major [rt row tiles of 128][nkt K tiles of 64] x 16 KiB; with `splitk` K
// slices the XCDs split K first, then the row tiles (gemm.hip tile_map).  Speed only: nothing depends on where a block really runs.
struct PrefetchDesc {
    const void* next;     // nullptr = off
    int rt, nkt, splitk;  // splitk >= 1, nkt % splitk == 0, splitk divides 8 or is a multiple of 8
    int kt_limit;         // > 0: only the first kt_limit K tiles of every (row tile, K slice) — what the consumer's prologue waits for; 0 = the whole slice
};
// The caller is thread `t` of the `nt` threads of the j-th of `nb` prefetching blocks of XCD `xcd`.  `sink` receives every load: keep it alive
// (asm volatile("" :: "v"(sink))) until a later wait proves the loads returned, or until the wave ends.
__device__ __forceinline__ void l2_prefetch_slice(const PrefetchDesc& d, int xcd, int j, int nb, int t, int nt, unsigned& sink) {
    const int sk = d.splitk;
    int ks0, ksn, rt0, rtn;
    if (sk >= 8) {
        ksn = sk >> 3; ks0 = xcd * ksn; rt0 = 0; rtn = d.rt;
    } else {
        const int per = 8 / sk, part = xcd % per;
        ks0 = xcd / per; ksn = 1;
        rt0 = part * d.rt / per; rtn = (part + 1) * d.rt / per - rt0;
    }
    const int kpt = d.nkt / sk;                                                  // K tiles per K slice
    const int lim = d.kt_limit > 0 && d.kt_limit < kpt ? d.kt_limit : kpt;       // ... of which the first `lim` are touched
    const int per_rt = ksn * lim;
    const int lines = rtn * per_rt * 128;
    const int l0 = (int)((long long)lines * j / nb), l1 = (int)((long long)lines * (j + 1) / nb);
    for (int l = l0 + t; l < l1; l += nt) {
        const int tile = l >> 7, r = tile / per_rt, rem = tile - r * per_rt, sl = rem / lim, k = rem - sl * lim;
        const char* a = (const char*)d.next + ((size_t)(rt0 + r) * d.nkt + (size_t)(ks0 + sl) * kpt + k) * 16384 + (l & 127) * 128;
        asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(a) : "memory");
    }
}
static inline bool prefetch_desc_ok(const PrefetchDesc& d) {
    return !d.next || (d.rt > 0 && d.nkt > 0 && d.splitk >= 1 && d.nkt % d.splitk == 0 && (d.splitk >= 8 ? d.splitk % 8 == 0 : 8 % d.splitk == 0));
}

// ---- fp32 -> fp16 with saturation -------------------------------------------------------------
// The reference runs this path under bf16 autocast (fp32 exponent range); our inter-kernel activations are fp16, so a
// plain conversion would turn |x| > 65504 into inf and the next GEMM into NaN.  Every fp16 store of an activation goes
// through sat4(): one v_med3_f32 per value clamps to +-65504 and a running |x| maximum (two v_max3_f32 per four values)
// lets the kernel raise a device flag (bit 1 of the handle's error word, reported by gtav_dit_check / gtav_vae_check)
// when anything was clamped: results stay finite and the caller learns that the fp16 range was exceeded.
#ifdef GTAV_BF16_OPERANDS
constexpr float F16_MAX = 3.3895313892515355e38f;   // largest finite bf16: the clamp never bites below fp32's own overflow, ERR_F16_SAT is never raised
#else
constexpr float F16_MAX = 65504.0f;
#endif
constexpr int ERR_TIMESTEP = 1, ERR_F16_SAT = 2, ERR_NONFINITE = 4;
__device__ __forceinline__ f16x4 sat4(float a, float b, float c, float d, float& amax) {
    amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), amax);
    amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(c), __builtin_fabsf(d)), amax);
    return f16x4{(f16)__builtin_amdgcn_fmed3f(a, -F16_MAX, F16_MAX), (f16)__builtin_amdgcn_fmed3f(b, -F16_MAX, F16_MAX),
                 (f16)__builtin_amdgcn_fmed3f(c, -F16_MAX, F16_MAX), (f16)__builtin_amdgcn_fmed3f(d, -F16_MAX, F16_MAX)};
}
// NaN-safe: a NaN input compares false and is not reported here (it stays a NaN in the output)
__device__ __forceinline__ void sat_report(float amax, int* err_flag) {
    if (err_flag && amax > F16_MAX) atomicOr(err_flag, ERR_F16_SAT);
}

// Sum over the 8 lanes of an aligned lane group with DPP moves (quad xor 1, quad xor 2, half-row mirror) instead of three
// ds_bpermute-based shuffles: every lane of the group ends up with the group's total (temporal attention: 8 lanes per head row).
__device__ __forceinline__ float group8_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    return v;
}

// ---- small device math ------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// GELU(approximate="tanh") as torch computes it: 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).
// 0.5 (1 + tanh u) == sigmoid(2u) == 1 / (1 + 2^(-2 u log2 e)): one v_exp_f32 + one v_rcp_f32 (1 ulp each; the
// result is rounded to fp16 right after), exact limits at both tails (2^+inf -> rcp(inf) = 0, 2^-inf -> 1).
__device__ __forceinline__ float gelu_tanh_f(float x) {
    // -2 log2(e) u = x (a + b x^2) with the constants folded: 3 multiplies / FMAs ahead of the two transcendentals instead of 5
    const float a = -2.0f * 1.4426950408889634f * 0.7978845608028654f, b = a * 0.044715f;
    const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, b, a));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// The same on four values with the full-rate arithmetic as packed fp32 operations (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two lanes
// of work per instruction, IEEE-identical results to gelu_tanh_f): the GELU epilogue is VALU-bound (GEMM epilogue of fc1: 5.7 us of a 25.7 us
// residency round at M = 5760), and the five full-rate operations per value are half of its issue slots beside the two transcendentals.
typedef float f32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_tanh_f4(const float (&x)[4], float (&y)[4]) {
    const float a = -2.0f * 1.4426950408889634f * 0.7978845608028654f, b = a * 0.044715f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2_ v = f32x2_{x[2 * h], x[2 * h + 1]};
        const f32x2_ t = __builtin_elementwise_fma(v * v, f32x2_{b, b}, f32x2_{a, a});
        const f32x2_ arg = v * t;
        const f32x2_ e = f32x2_{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
        const f32x2_ d = e + f32x2_{1.0f, 1.0f};
        const f32x2_ r = f32x2_{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const f32x2_ o = v * r;
        y[2 * h] = o[0];
        y[2 * h + 1] = o[1];
    }
}

// d/dx of gelu_tanh_f for two values (training backward): g = s + x s (1 - s) 2k (1 + 3c x^2) with s = 1 / (1 + 2^(x (a + b x^2))) as above, k = sqrt(2/pi),
// c = 0.044715.  The same two transcendentals and eight packed fp32 operations per PAIR; with the IEEE division and the scalar arithmetic it had before,
// gelu_bwd_colsum_kernel needed ~40 issue slots per value and was VALU-bound (57 us for the 283 MB of an fc1 backward at M = 11 520, round 5).
__device__ __forceinline__ f32x2_ gelu_tanh_grad_f2(f32x2_ v) {
    const float k = 0.7978845608028654f, c = 0.044715f;
    const float a = -2.0f * 1.4426950408889634f * k, b = a * c, k2 = 2.0f * k, k6 = 6.0f * k * c;
    const f32x2_ z = v * v;
    const f32x2_ arg = v * __builtin_elementwise_fma(z, f32x2_{b, b}, f32x2_{a, a});
    const f32x2_ d = f32x2_{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])} + f32x2_{1.0f, 1.0f};
    const f32x2_ s = f32x2_{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2_ t = (v * s) * __builtin_elementwise_fma(z, f32x2_{k6, k6}, f32x2_{k2, k2});
    return __builtin_elementwise_fma(t, f32x2_{1.0f, 1.0f} - s, s);
}

// exact (erf) GELU: 0.5 x (1 + erf(x / sqrt(2)))  (model/vae.py:128, torch.nn.GELU()) through libm's erff: ~36 VALU instructions per value
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }

// The same function for the GEMM epilogues (round 5): erf(x / sqrt 2) = z Q(z^2 - 1) with z = x sqrt 2 / 4.5 clamped to [-sqrt 2, sqrt 2] (beyond |x| = 4.5 the
// polynomial's end value, 1 to 3e-8, stands for erf) and Q a degree-9 minimax polynomial in a variable that spans [-1, 1] (coefficients of order 1: no
// cancellation in fp32), fitted to the ABSOLUTE error of the GELU value: |error| <= 3.0e-5 for every fp32 x in [-65504, 65504] (tools/gelu_poly_fit.py prints
// the table and re-measures it; 7e-6 of it is the fit, the rest the cancellation in hx f + hx where f is near -1) — no sigmoid shortcut, the definition
// itself.  That is an order of magnitude under the fp16 rounding of the stored activation (4.9e-4 relative) wherever |GELU| >~ 0.06; in the negative
// tail (x < -4, |GELU| < 1e-4) only the absolute bound holds: the relative error there can exceed 100 % and the sign may flip at magnitudes below 3e-5
// (tests/test_gpu_ops.py::test_gelu_erf_epilogue_absolute_error pins the bound).  14 operations per value, all but the clamp as packed fp32 (v_pk_mul /
// v_pk_fma: two values per instruction): ~7.5 issue slots per value.
constexpr float kGeluErfL = 4.5f;
constexpr float kGeluErfC[10] = {9.985491633e-01f, -4.911968410e-01f, 3.479329944e-01f, -2.564433515e-01f, 1.821636558e-01f,
                                 -1.155658439e-01f, 6.453407556e-02f, -4.252957553e-02f, 2.962661162e-02f, -9.964072146e-03f};
__device__ __forceinline__ void gelu_erf_f4(const float (&x)[4], float (&y)[4]) {
    constexpr float r2 = 1.4142135623730951f, zs = r2 / kGeluErfL;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2_ v = f32x2_{x[2 * h], x[2 * h + 1]};
        const f32x2_ z = v * f32x2_{zs, zs};
        const f32x2_ zc = f32x2_{__builtin_amdgcn_fmed3f(z[0], -r2, r2), __builtin_amdgcn_fmed3f(z[1], -r2, r2)};
        const f32x2_ sm = __builtin_elementwise_fma(zc, zc, f32x2_{-1.0f, -1.0f});
        f32x2_ q = f32x2_{kGeluErfC[9], kGeluErfC[9]};
#pragma unroll
        for (int k = 8; k >= 0; --k) q = __builtin_elementwise_fma(q, sm, f32x2_{kGeluErfC[k], kGeluErfC[k]});
        const f32x2_ f = zc * q;
        const f32x2_ hx = v * f32x2_{0.5f, 0.5f};
        const f32x2_ o = __builtin_elementwise_fma(hx, f, hx);
        y[2 * h] = o[0];
        y[2 * h + 1] = o[1];
    }
}

// Wave-wide sum with DPP moves instead of six ds_bpermute shuffles (each of those is an LDS round trip on the critical path):
// quad xor 1, quad xor 2, half-row mirror, row mirror leave every 16-lane row's total in all its lanes; row_bcast15 /
// row_bcast31 then chain the four rows into lane 63, which is broadcast with v_readlane.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define GTAV_DPP_ADD(ctrl, rmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, true))
    GTAV_DPP_ADD(0xB1, 0xF);    // quad_perm [1,0,3,2]
    GTAV_DPP_ADD(0x4E, 0xF);    // quad_perm [2,3,0,1]
    GTAV_DPP_ADD(0x141, 0xF);   // row_half_mirror
    GTAV_DPP_ADD(0x140, 0xF);   // row_mirror
    GTAV_DPP_ADD(0x142, 0xA);   // row_bcast15 into rows 1 and 3
    GTAV_DPP_ADD(0x143, 0xC);   // row_bcast31 into rows 2 and 3
#undef GTAV_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 16-byte store that is written through and dropped from the XCD's L2 (sc1).  A kernel's plain stores stay dirty in L2
// until the end-of-kernel release writes them back; with sc1 they have already left when the last wave ends, and outputs
// that the same launch never re-reads do not evict its operands.  Only for 16-byte stores: narrower sc1 stores are one
// fabric write each and slower than plain ones (MI355X_MICROARCH.md, stores of each flavour).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_sc1(void* dst, u32x4 v) {
    // the trailing s_nop 1 is part of the instruction's contract: the store reads its four data registers over several cycles
    // after issue, hipcc pads that hazard only for stores IT emits, and without the wait states the next VALU instruction may
    // overwrite the last data dword before the last 16 lanes have read it (seen in round 2 as one wrong fp16 per 16-byte chunk in
    // lanes 48-63, a few times per million stores: cdna_hip_programming.md 5.7 "Stores")
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void store16_sc1(void* dst, f32x4 v) {
    union { f32x4 f; u32x4 u; } cv;
    cv.f = v;
    store16_sc1(dst, cv.u);
}

// fp16x4 (8 bytes) per lane where lanes l and l ^ XM hold adjacent 4-column groups of one row (the lower group in the lane
// with bit XM clear): either a plain 8-byte store per lane, or the lower lane collects its neighbour's half and writes 16
// bytes write-through (sc1).  Both lanes of a pair must be active.
template <int XM>
__device__ __forceinline__ void store_f16x4_paired(f16* dst, f16x4 o, int lane, int paired) {
    if (!paired) {
        *(f16x4*)dst = o;
        return;
    }
    union { f16x4 h; unsigned u[2]; } mine, other;
    mine.h = o;
    if constexpr (XM == 1) {   // neighbour lane: a DPP quad permute [1,0,3,2] instead of an LDS-crossbar shuffle
        other.u[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine.u[0], 0xB1, 0xF, 0xF, true);
        other.u[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine.u[1], 0xB1, 0xF, 0xF, true);
    } else {
        other.u[0] = __shfl_xor(mine.u[0], XM, 64);
        other.u[1] = __shfl_xor(mine.u[1], XM, 64);
    }
    if (!(lane & XM)) store16_sc1(dst, u32x4{mine.u[0], mine.u[1], other.u[0], other.u[1]});
}

}  // namespace gtav
