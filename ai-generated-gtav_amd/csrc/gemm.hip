// fp16 MFMA GEMMs for gfx950 with fused epilogues (see gemm.h).
//
// One family of kernels, all K-step 64, v_mfma_f32_16x16x32_f16, operands staged with 16-byte direct-to-LDS loads
// (global_load_lds_dwordx4, 1 KiB = 8 rows per wave-instruction) into an LDS ring with counted s_waitcnt vmcnt(N) and ONE raw
// s_barrier per K-step.  Both operands are K-contiguous and TILE-MAJOR in memory (common.h tiled_off), so a 1-KiB piece is
// one linear copy; the LDS image is lane-linear and the bank-conflict swizzle (16-B chunk ^= row & 7) is already in the
// source layout and is applied again to the ds_read_b128 address (same involution).
//   gemm_kernel     128 x 128 / 128 x 256 tiles, 4 or 8 waves (+ optional loader waves)      mainloop / mainloop_ls
//   gemm_g_kernel   piece-granular tiles that may start on any 8-row boundary: 128 x 96, 96 x 96, 64 x 48 (small M),
//                   128 x 192 (large M, two blocks per CU)                                     mainloop_g
//   gemm256_kernel  256 x 256 tile, K-tile in four quadrant phases                             mainloop256
//   gemm_l_kernel   loader-wave tiles (small and medium M: 128 x 96, 64 x 48 / 96, 128 x 144)   mainloop_l
//   gemm_lp_kernel  persistent loader-wave kernel, 128 x 192 tiles (large-M residual GEMMs, in-place residual epilogue)
// launch_gemm() picks the shape per launch from a cost model / measured thresholds (bottom of this file).  Kernels, launchers and block shapes that only
// ever measured slower live in gemm_experiments.inc, which only `build.sh exp` compiles.
//
// Orientation: the MFMA "A" operand is the W tile and "B" the X tile, so D[row = feature][col = token]:
// every lane owns 4 CONSECUTIVE FEATURES of one token -> RoPE pairs, float4 bias/gate/residual and
// packed fp16x4 stores are lane-local.  Blocks that produce V for the spatial/VAE attention swap the
// operands (D[row = token][col = feature]) so that V is written already transposed (Vt[d][s]), which
// is the layout the PV product wants as its MFMA A operand.
#include "gemm.h"
#include "attn_tile.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <hip/hip_ext.h>

namespace gtav {

namespace {

constexpr int TN = 128, TK = 64;
constexpr int TILE_BYTES = 128 * TK * 2;  // 16 KiB per operand tile

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// Direct-to-LDS load in the scalar-base form: address = sbase (wave-uniform, an SGPR pair) + voff (32-bit per-lane offset) —
// no 64-bit per-lane pointer and no vector address arithmetic per K-step.  hipcc does not emit this form for the builtin, so it
// is inline asm; M0 (the LDS destination base of the DMA) is written in the same statement and restored (cdna_hip_programming.md
// 5.7).  The statement has no VGPR destination; completion is counted by hand (vmcnt) like every other fill.
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
// 16-byte store at a wave-uniform base + a 32-bit per-lane offset (no 64-bit per-lane address in VGPRs)
__device__ __forceinline__ void store16_soff(const void* sbase, unsigned voff, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
// one dword per lane (64 separate lines if the lanes say so) into 256 bytes of LDS: an L2 "touch" that writes no VGPR
__device__ __forceinline__ void glds4_s(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_offset(const void* p) { return (unsigned)(size_t)(lptr_t)p; }

__device__ __forceinline__ void store16q_sc1(void* dst, uint4 v) { store16_sc1(dst, u32x4{v.x, v.y, v.z, v.w}); }

// fp32 x4 -> fp16 x4, saturating (common.h sat4): amax collects the largest magnitude seen by this lane
__device__ __forceinline__ uint2 pack4(float& amax, float a, float b, float c, float d) {
    union { f16x4 h; uint2 u; } cv;
    cv.h = sat4(a, b, c, d, amax);
    return cv.u;
}


// K-step synchronisation: the wave's LDS reads have drained, all but its VM youngest vector-memory operations (the LDS-DMA fills of
// newer tiles) have completed, then the workgroup barrier.  The waits are the BUILTIN, not inline asm, and the lgkmcnt(0) is
// UNCONDITIONAL (ahead of the run-time choice of the vmcnt count): hipcc's own wait-count pass then knows on every path that no LDS
// read is outstanding after this point.  With the asm form (and with the lgkmcnt inside the branches) it did not, and in the
// software-pipelined loops it put an `s_waitcnt lgkmcnt(0)` between the fragment reads of tile t and the MFMAs of tile t - 1 in
// every second K-step: the reads' full LDS round trip exposed (gemm.s of round 2).  The barrier stays inline asm with a memory
// clobber: no LDS or global access may move across it.
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }
// Interleaved-pair RoPE of four consecutive features (two pairs) by (cos, sin, cos, sin): explicit fused multiply-adds in ONE fixed
// order.  Left as `a * c - b * s` the compiler picks which product to fuse per call site (the -DGTAV_EXPERIMENTS build fused the other
// one in the fused temporal kernel: 23 of 737 280 k values one fp16 ulp apart), and every QKV epilogue must produce the same bits
// (window step == context-cached step == fused temporal kernel).
__device__ __forceinline__ f32x4 rope4(const f32x4 v, const f32x4 cs) {
    f32x4 r;
    r[0] = __builtin_fmaf(v[0], cs[0], -(v[1] * cs[1]));
    r[1] = __builtin_fmaf(v[1], cs[0], v[0] * cs[1]);
    r[2] = __builtin_fmaf(v[2], cs[2], -(v[3] * cs[3]));
    r[3] = __builtin_fmaf(v[3], cs[2], v[2] * cs[3]);
    return r;
}

// the (cos, sin) table feature n of a QKV output rotates by: q features may have a table of their own (GemmParams::rope_cs_q)
__device__ __forceinline__ const float* rope_tab(const GemmParams& p, int n) { return (p.rope_cs_q && n < p.D) ? p.rope_cs_q : p.rope_cs; }

template <int VM>
__device__ __forceinline__ void wait_vm() {
    static_assert(VM >= 0 && VM < 64, "vmcnt is a 6-bit field");
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(VM, 15));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0)); }
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int VM>
__device__ __forceinline__ void kstep_sync() {
    wait_lgkm0();
    wait_vm<VM>();
    wg_barrier();
}
// the same for an NS-stage ring with G fills per wave and K-step: min(NS - 2, rem) newer tiles are in flight (rem = K-steps after this one)
template <int NS, int G>
__device__ __forceinline__ void wait_vm_ring(int rem) {
    if constexpr (NS >= 6) { if (rem >= 4) { wait_vm<4 * G>(); return; } }
    if constexpr (NS >= 5) { if (rem >= 3) { wait_vm<3 * G>(); return; } }
    if constexpr (NS >= 4) { if (rem >= 2) { wait_vm<2 * G>(); return; } }
    if constexpr (NS >= 3) { if (rem >= 1) { wait_vm<G>(); return; } }
    wait_vm<0>();
}
template <int NS, int G>
__device__ __forceinline__ void kstep_sync_ring(int rem) {
    wait_lgkm0();
    wait_vm_ring<NS, G>(rem);
    wg_barrier();
}

#ifdef GTAV_EXPERIMENTS
// s_memrealtime (100 MHz, one counter for the whole chip) orders the phases of different blocks; s_memtime (shader cycles) is
// a per-XCD counter with unrelated offsets and only gives this block's own cycle count (-> its clock)
#define GTAV_STAMP(var) do { if (p.stamps) var = __builtin_amdgcn_s_memrealtime(); } while (0)
// one lane's stamp straight into its block's row of the stamp buffer (the loader-wave kernels' per-wave / per-K-step timeline: 8 slots per block, 64 with debug bit 5)
#define GTAV_STAMP_ROW(p) ((p).stamps + (size_t)blockIdx.x * (((p).debug & 32) ? 64 : 8))
#define GTAV_STAMP_SLOT(cond, slot) do { if ((cond) && p.stamps) GTAV_STAMP_ROW(p)[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GTAV_STAMP_SLOT_HI(cond, slot) do { if ((cond) && p.stamps) GTAV_STAMP_ROW(p)[slot] |= (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFFFFFull) << 8; } while (0)
struct BlockStamps {
    unsigned long long t[4] = {0, 0, 0, 0}, c0 = 0;
    unsigned pf_sink = 0;   // destination register of l2_prefetch_next's loads: stays allocated until end()
    __device__ __forceinline__ void begin(const GemmParams& p) {
        if (p.stamps) { t[0] = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime(); }
    }
    template <bool PF = false>
    __device__ __forceinline__ void end(const GemmParams& p) {
        if constexpr (PF) asm volatile("" ::"v"(pf_sink));
        if (p.stamps && threadIdx.x == 0) {
            t[3] = __builtin_amdgcn_s_memrealtime();
            unsigned long long* d = p.stamps + (size_t)blockIdx.x * ((p.debug & 32) ? 64 : 8);
            d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = t[3]; d[4] = c0; d[5] = __builtin_amdgcn_s_memtime();
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            atomicOr(&d[6], (unsigned long long)(xcc & 0xF));   // bits 8.. may hold a loader wave's stamp (mainloop_l); the tool zeroes the buffer

        }
    }
};
#else
#define GTAV_STAMP(var) do { } while (0)
#define GTAV_STAMP_SLOT(cond, slot) do { } while (0)
#define GTAV_STAMP_SLOT_HI(cond, slot) do { } while (0)
struct BlockStamps {
    unsigned long long t[4];
    unsigned pf_sink = 0;   // destination register of l2_prefetch_next's loads: stays allocated until end()
    __device__ __forceinline__ void begin(const GemmParams&) {}
    template <bool PF = false>
    __device__ __forceinline__ void end(const GemmParams&) {
        if constexpr (PF) asm volatile("" ::"v"(pf_sink));
    }
};
#endif

// L2 prefetch of the next GEMM's weight (GemmParams::pf, common.h l2_prefetch_slice).  Called by the compute waves of the loader-wave kernels
// before their first barrier — they issue no other vector-memory instruction until the epilogue, whose first wait on a (younger) bias / RoPE
// load also proves these loads have returned (loads retire in order) — `sink`, their destination register, stays allocated until
// BlockStamps::end.  64 lines per wave-instruction, so a block's share (a few hundred lines) is one instruction per compute wave: nothing
// beside the ~450 fill instructions of its K loop.
__device__ __forceinline__ void l2_prefetch_next(const GemmParams& p, int cw, int ncw, int lane, unsigned& sink) {
    if (!p.pf.next) return;
    const int xcd = blockIdx.x & 7;
    l2_prefetch_slice(p.pf, xcd, (int)blockIdx.x >> 3, ((int)gridDim.x + 7 - xcd) >> 3, 64 * cw + lane, 64 * ncw, sink);
}

// Scheduling directive for a region that holds NM MFMAs and ND independent ds_reads: emit them as MFMA, RPM reads, MFMA, RPM reads,
// ... until the reads are out, the remaining MFMAs last (sched_group_barrier masks: 0x008 = MFMA, 0x100 = DS read).  The reads go
// into the issue shadows of the FIRST MFMAs so that the rest of the MFMA block covers their LDS round trip before the K-step's
// lgkmcnt(0) + barrier.
template <int NM, int ND, int RPM = 1>
__device__ __forceinline__ void interleave_mfma_dsread() {
    constexpr int NG = (ND + RPM - 1) / RPM;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
    }
    if constexpr (NM > NG) __builtin_amdgcn_sched_group_barrier(0x008, NM - NG, 0);
}

// NS-stage LDS ring.  Tile t lives in stage t % NS.  Steady state: NS-1 tiles are in flight when iteration t starts;
// the wave waits (counted vmcnt, never 0 in the main loop) until ITS OWN share of tile t has landed, the raw
// s_barrier then (a) makes every wave's share of tile t visible and (b) proves every wave has finished reading
// tile t-1, whose stage is refilled right after the barrier with tile t+NS-1.  One barrier per K-step.
//
// Block shape: 2 waves along N (64 features each) x WM waves along M (16 FJ tokens each):
//   WM = 2, FJ = 4: 128 x 128 tile, 4 waves, 32 KiB / stage
//   WM = 4, FJ = 2: 128 x 128 tile, 8 waves (two per SIMD), 32 KiB / stage   (small M: overlap fills with MFMA)
//   WM = 4, FJ = 4: 128 x 256 tile, 8 waves, 48 KiB / stage                  (large M: 1.5x fewer fill bytes per FLOP)
// With two waves per SIMD the second half of the waves (4..7, the SIMD partners of 0..3) issues its share of the
// next tile AFTER its MFMAs instead of before: a wave's direct-to-LDS loads back-pressure its in-order instruction
// stream at the ~63 GB/s/CU fill rate (profiles/round1/v5_gemm_8wave_microbench.txt: fills-only and MFMA-only loops
// cost the same and used to add up), so the partners' fill and MFMA phases now run beside each other.
struct NoStepHook { __device__ __forceinline__ void operator()(int) const {} };
// per_step(t): called once per K-step behind that step's refill (EPI_RESID_FOLD: one residual load per lane and step)
template <bool TR, int NS, int WM, int FJ, typename AfterPrologue, typename PerStep = NoStepHook>
__device__ __forceinline__ void mainloop(const GemmParams& p, char* smem, int n0, int m0, int kt0, int nkt,
                                         f32x4 (&acc)[4][FJ], BlockStamps& bs, AfterPrologue after_prologue, PerStep per_step = PerStep()) {
    constexpr int NWAVE = 2 * WM;
    constexpr int TMB = WM * 16 * FJ;                  // tokens per block tile
    constexpr int XT = TMB / 128;                      // 128-row X tiles per stage
    constexpr int STAGE_BYTES = (1 + XT) * TILE_BYTES;
    constexpr int WP = 16 / NWAVE;                     // W pieces (1 KiB) per wave per stage
    constexpr int XP = 16 * XT / NWAVE;                // X pieces per wave per stage
    constexpr int G = WP + XP;                         // direct-to-LDS loads per wave per stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1, wm = w >> 1;
    const bool late = NWAVE == 8 && w >= 4;             // SIMD partner of wave w - 4: fills after its MFMAs

    // ---- staging: operands are tile-major (common.h tiled_off): the 16 KiB tile (row-tile, k-tile) is contiguous and
    // already in LDS-image order, so each 1 KiB piece is one linear direct-to-LDS instruction ----
    const int nktot = p.K / TK;
    const char* wsrc = (const char*)p.W + ((size_t)(n0 >> 7) * nktot + kt0) * TILE_BYTES + (w * WP) * 1024 + lane * 16;
    const int xq = w * XP;                              // first X piece of this wave (pieces 0 .. 16 XT - 1)
    int xrt = (m0 >> 7) + (xq >> 4);
    const int last_rt = (p.M - 1) >> 7;
    xrt = xrt < last_rt ? xrt : last_rt;                // ragged last tile: re-read a valid row tile (results are masked)
    const char* xsrc = (const char*)p.X + ((size_t)xrt * nktot + kt0) * TILE_BYTES + (xq & 15) * 1024 + lane * 16;
    auto stage = [&](int t) {
        char* base = smem + (t % NS) * STAGE_BYTES;
        const char* ws = wsrc + (size_t)t * TILE_BYTES;
        const char* xs = xsrc + (size_t)t * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < WP; ++i) glds16(ws + i * 1024, base + (w * WP + i) * 1024);
#pragma unroll
        for (int i = 0; i < XP; ++i) glds16(xs + i * 1024, base + TILE_BYTES + (xq + i) * 1024);
    };

    // ---- fragment read offsets ----
    const int li = lane & 15, g = lane >> 4;
    const int xrow0 = wm * 16 * FJ;                     // first token row of this wave inside the block tile
    int woff[2], xoff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ch = ((4 * s + g) ^ (li & 7)) << 4;
        woff[s] = (64 * wn + li) * 128 + ch;
        xoff[s] = TILE_BYTES + (xrow0 >> 7) * TILE_BYTES + ((xrow0 & 127) + li) * 128 + ch;
    }

    const int npro = nkt < NS - 1 ? nkt : NS - 1;
    for (int t = 0; t < npro; ++t) stage(t);
    after_prologue();   // register loads the epilogue wants early (bias): behind the first fills, not in front of them
    for (int t = 0; t < nkt; ++t) {
        // tiles newer than t already issued: min(NS - 2, nkt - 1 - t), G loads each
        const int rem = nkt - 1 - t;
        kstep_sync_ring<NS, G>(rem);
        if (t == 0) GTAV_STAMP(bs.t[1]);
        const bool refill = t + NS - 1 < nkt && !GTAV_DBG(p, 1);
        if (refill && !late) stage(t + NS - 1);
        per_step(t);
        const char* b = smem + (t % NS) * STAGE_BYTES;
        if (!GTAV_DBG(p, 2)) {
            // both 32-deep halves of the K-step are fetched up front: the second half's fragments arrive under the
            // first half's MFMAs (the compiler emits the counted lgkmcnt waits)
            f16x8 wf[2][4], xf[2][FJ];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[s][i] = *(const f16x8*)(b + woff[s] + i * 16 * 128);
#pragma unroll
                for (int j = 0; j < FJ; ++j) xf[s][j] = *(const f16x8*)(b + xoff[s] + j * 16 * 128);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < FJ; ++j) {
                        if (TR)
                            acc[i][j] = mfma16(xf[s][j], wf[s][i], acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = mfma16(wf[s][i], xf[s][j], acc[i][j], 0, 0, 0);
                    }
        }
        if (refill && late) {
            asm volatile("" ::: "memory");   // keep the loads behind the MFMA block in program order
            stage(t + NS - 1);
        }
    }
}

// 256 features x 256 tokens per block, 8 waves (2 along N x 4 along M), each wave 128 x 64 = 8 x 4 MFMA tiles: half the
// L2->LDS fill bytes per FLOP of the 128 x 128 tile, which is what bounds the large-M GEMMs (755 MB of fills per fc1 launch
// at M = 5760 = 46 GB/s per CU over the whole launch, against a 66-73 GB/s per-CU LDS-DMA ceiling).
// LDS: two K-tile parities x four 16 KiB half-tiles {W0, W1, X0, X1} = 128 KiB.  A K-tile (64 MFMAs per wave) runs as four
// quadrant phases of 16 MFMAs; one half-tile (two 1-KiB pieces per wave) of a later K-tile is issued per phase, so fills,
// fragment reads and MFMAs interleave finely:
//   phase 1: issue W0(t+1) | read W rows 0-63 (8) + X cols 0-31 (4) + X cols 32-63 (4) | MFMA W[0:4] x X[0:2]
//   phase 2: issue W1(t+1) | read W rows 64-127 (8, for phase 3)                        | MFMA W[0:4] x X[2:4]
//   -- barrier (b): every wave has finished reading the X half-tiles of this parity --
//   phase 3: issue X0(t+2) into this parity's X0                                       | MFMA W[4:8] x X[2:4]
//   phase 4: issue X1(t+2)                                                             | MFMA W[4:8] x X[0:2]
//   -- vmcnt(4): everything but X0/X1(t+2) landed = K-tile t+1 complete; barrier (a) --
// WAR: W slots of parity p are last read in phase 2 (prefetched) of K-tile t and refilled in phases 1-2 of t+1, after
// barrier (a); X slots are last read in phase 1 and refilled after barrier (b).  RAW: a half-tile is read only after the
// issuing waves' counted vmcnt and a barrier (MI355X_MICROARCH.md: LDS-DMA ordering).
template <bool TR, typename AfterPrologue>
__device__ __forceinline__ void mainloop256(const GemmParams& p, char* smem, int n0, int m0, int kt0, int nkt,
                                            f32x4 (&acc)[8][4], BlockStamps& bs, AfterPrologue after_prologue) {
    constexpr int PAR = 4 * TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1, wm = w >> 1;
    const int nktot = p.K / TK;
    const int last_wt = ((p.N + 127) >> 7) - 1, last_rt = (p.M - 1) >> 7;
    const int wt0 = n0 >> 7, wt1 = wt0 + 1 < last_wt ? wt0 + 1 : last_wt;     // ragged edges re-read a valid tile (masked)
    const int rt0 = (m0 >> 7) < last_rt ? (m0 >> 7) : last_rt, rt1 = (m0 >> 7) + 1 < last_rt ? (m0 >> 7) + 1 : last_rt;
    const size_t po = (size_t)(2 * w) * 1024 + lane * 16;
    const char* const sw0 = (const char*)p.W + ((size_t)wt0 * nktot + kt0) * TILE_BYTES + po;
    const char* const sw1 = (const char*)p.W + ((size_t)wt1 * nktot + kt0) * TILE_BYTES + po;
    const char* const sx0 = (const char*)p.X + ((size_t)rt0 * nktot + kt0) * TILE_BYTES + po;
    const char* const sx1 = (const char*)p.X + ((size_t)rt1 * nktot + kt0) * TILE_BYTES + po;
    char* const dst0 = smem + (2 * w) * 1024;
    auto stage = [&](const char* src, int h, int t) {
        const char* s = src + (size_t)t * TILE_BYTES;
        char* d = dst0 + (t & 1) * PAR + h * TILE_BYTES;
        glds16(s, d);
        glds16(s + 1024, d + 1024);
    };

    const int li = lane & 15, g = lane >> 4;
    int woff[2], xoff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ch = ((4 * s + g) ^ (li & 7)) << 4;
        woff[s] = wn * TILE_BYTES + li * 128 + ch;
        xoff[s] = (2 + (wm >> 1)) * TILE_BYTES + (64 * (wm & 1) + li) * 128 + ch;
    }
    auto mma = [&](const f16x8& wv, const f16x8& xv, f32x4& c) {
        if (TR) c = mfma16(xv, wv, c, 0, 0, 0);
        else c = mfma16(wv, xv, c, 0, 0, 0);
    };

    stage(sx0, 2, 0); stage(sx1, 3, 0); stage(sw0, 0, 0); stage(sw1, 1, 0);
    if (nkt > 1) {
        stage(sx0, 2, 1); stage(sx1, 3, 1);
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    after_prologue();
    const bool fills = !GTAV_DBG(p, 1);
    GTAV_STAMP(bs.t[1]);
    for (int t = 0; t < nkt; ++t) {
        const char* b = smem + (t & 1) * PAR;
        const bool n1 = t + 1 < nkt && fills, n2 = t + 2 < nkt && fills;
        f16x8 wa[2][4], wb[2][4], xa[2][2], xb[2][2];
        // ---- phase 1 ----
        if (n1) stage(sw0, 0, t + 1);
        {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wa[s][i] = *(const f16x8*)(b + woff[s] + i * 16 * 128);
#pragma unroll
                for (int j = 0; j < 2; ++j) xa[s][j] = *(const f16x8*)(b + xoff[s] + j * 16 * 128);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) xb[s][j] = *(const f16x8*)(b + xoff[s] + (2 + j) * 16 * 128);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma(wa[s][i], xa[s][j], acc[i][j]);
        }
        // ---- phase 2 ----
        if (n1) stage(sw1, 1, t + 1);
        {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) wb[s][i] = *(const f16x8*)(b + woff[s] + (4 + i) * 16 * 128);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma(wa[s][i], xb[s][j], acc[i][2 + j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (b)
        // ---- phase 3 ----
        if (n2) stage(sx0, 2, t + 2);
        {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma(wb[s][i], xb[s][j], acc[4 + i][2 + j]);
        }
        // ---- phase 4 ----
        if (n2) stage(sx1, 3, t + 2);
        {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma(wb[s][i], xa[s][j], acc[4 + i][j]);
        }
        if (n2) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");   // (a)
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// The same 256 x 256 tile, LDS image, fill schedule and accumulator layout as mainloop256, run as TWO WAVE GROUPS IN ANTIPHASE (round 6; cdna_hip_programming.md
// "The 256^2 8-phase template"): waves 0-3 and waves 4-7 — one wave of each group per SIMD — execute the same phase sequence one barrier interval apart, so that on
// every SIMD one wave issues its 16 MFMAs of a phase (s_setprio 1) while its partner issues the fragment reads and the LDS-DMA fills of its next phase and waits for
// the reads to land.  In the lockstep loop both waves of a SIMD reach their reads at the same time and the matrix pipe waits out the LDS round trip with them
// (docs/LABNOTES.md 4.11: reads + MFMAs without fills 1.18 us per K-tile against 0.77 us of MFMA issue); here that latency and the address-pipe time of a fill
// (a wave that issues an LDS-DMA instruction is held until the pipe accepts it) lie under the partner's MFMAs.  Price: two barriers per phase.
//   phase p of a wave:   [L]  fill one half-tile | fragment reads of the phase | (phase 4: counted vmcnt) | lgkmcnt(0) | barrier
//                        [M]  16 MFMAs                                                                                | barrier
//   K-tile t (parity b):  L1 W0(t+1) | wa (W rows 0-63: 8 reads), xa (tokens 0-31: 4)     M1 wa x xa
//                         L2 W1(t+1) | xb (tokens 32-63: 4)                               M2 wa x xb
//                         L3 X0(t+2) | wb (W rows 64-127: 8)                              M3 wb x xb
//                         L4 X1(t+2) | vmcnt(4): all but X0 / X1(t+2) landed              M4 wb x xa
// Group 1 starts one barrier late and group 0 ends one barrier late: barrier instance k closes group 0's section k and group 1's section k - 1.
// WAR: a slot is refilled at least one barrier after every wave's last read of it HAS RETURNED — the lgkmcnt(0) stands in FRONT of the barrier that ends a read section
// (X slots of parity b: last read L2(t), refilled L3(t) by group 0 while group 1 is in M2(t), its L2 reads retired before the barrier between; W slots: last read
// L3(t), refilled L1(t+1)).  RAW: a wave reads K-tile t+1 (from L1(t+1) on) only behind BOTH groups' vmcnt(4) of L4(t): group 0's own stands two barriers back, group
// 1's one barrier back (its L4(t) ends at the barrier that ends group 0's M4(t)); group 1 reads after group 0's by construction.  LDS-DMA data is ordered for a
// ds_read by exactly that: the issuing wave's counted vmcnt, then a barrier the reader has passed (MI355X_MICROARCH.md).
template <bool TR, typename AfterPrologue>
__device__ __forceinline__ void mainloop256_pp(const GemmParams& p, char* smem, int n0, int m0, int kt0, int nkt,
                                               f32x4 (&acc)[8][4], BlockStamps& bs, AfterPrologue after_prologue) {
    constexpr int PAR = 4 * TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1, wm = w >> 1;
    const bool late = w >= 4;                        // group 1: one barrier interval behind group 0
    const int nktot = p.K / TK;
    const int last_wt = ((p.N + 127) >> 7) - 1, last_rt = (p.M - 1) >> 7;
    const int wt0 = n0 >> 7, wt1 = wt0 + 1 < last_wt ? wt0 + 1 : last_wt;     // ragged edges re-read a valid tile (masked)
    const int rt0 = (m0 >> 7) < last_rt ? (m0 >> 7) : last_rt, rt1 = (m0 >> 7) + 1 < last_rt ? (m0 >> 7) + 1 : last_rt;
    const size_t po = (size_t)(2 * w) * 1024 + lane * 16;
    const char* const sw0 = (const char*)p.W + ((size_t)wt0 * nktot + kt0) * TILE_BYTES + po;
    const char* const sw1 = (const char*)p.W + ((size_t)wt1 * nktot + kt0) * TILE_BYTES + po;
    const char* const sx0 = (const char*)p.X + ((size_t)rt0 * nktot + kt0) * TILE_BYTES + po;
    const char* const sx1 = (const char*)p.X + ((size_t)rt1 * nktot + kt0) * TILE_BYTES + po;
    char* const dst0 = smem + (2 * w) * 1024;
    auto stage = [&](const char* src, int h, int t) {
        const char* s = src + (size_t)t * TILE_BYTES;
        char* d = dst0 + (t & 1) * PAR + h * TILE_BYTES;
        glds16(s, d);
        glds16(s + 1024, d + 1024);
    };
    const int li = lane & 15, g = lane >> 4;
    int woff[2], xoff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ch = ((4 * s + g) ^ (li & 7)) << 4;
        woff[s] = wn * TILE_BYTES + li * 128 + ch;
        xoff[s] = (2 + (wm >> 1)) * TILE_BYTES + (64 * (wm & 1) + li) * 128 + ch;
    }
    auto mma = [&](const f16x8& wv, const f16x8& xv, f32x4& c) {
        if (TR) c = mfma16(xv, wv, c, 0, 0, 0);
        else c = mfma16(wv, xv, c, 0, 0, 0);
    };
    // end of a read / fill section: this wave's LDS reads have returned (the WAR rule above), then the barrier; nothing may be scheduled across
    auto end_l = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto end_m = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    stage(sx0, 2, 0); stage(sx1, 3, 0); stage(sw0, 0, 0); stage(sw1, 1, 0);
    if (nkt > 1) {
        stage(sx0, 2, 1); stage(sx1, 3, 1);
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    after_prologue();
    const bool fills = !GTAV_DBG(p, 1);
    GTAV_STAMP(bs.t[1]);
    if (late) end_m();                               // group 1 enters the loop one interval behind
    for (int t = 0; t < nkt; ++t) {
        const char* b = smem + (t & 1) * PAR;
        const bool n1 = t + 1 < nkt && fills, n2 = t + 2 < nkt && fills;
        f16x8 wa[2][4], wb[2][4], xa[2][2], xb[2][2];
        // ---- L1 / M1 ----
        if (n1) stage(sw0, 0, t + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 2; ++j) xa[s][j] = *(const f16x8*)(b + xoff[s] + j * 16 * 128);
#pragma unroll
            for (int i = 0; i < 4; ++i) wa[s][i] = *(const f16x8*)(b + woff[s] + i * 16 * 128);
        }
        end_l();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma(wa[s][i], xa[s][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        end_m();
        // ---- L2 / M2 ----
        if (n1) stage(sw1, 1, t + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) xb[s][j] = *(const f16x8*)(b + xoff[s] + (2 + j) * 16 * 128);
        end_l();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma(wa[s][i], xb[s][j], acc[i][2 + j]);
        __builtin_amdgcn_s_setprio(0);
        end_m();
        // ---- L3 / M3 ----
        if (n2) stage(sx0, 2, t + 2);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) wb[s][i] = *(const f16x8*)(b + woff[s] + (4 + i) * 16 * 128);
        end_l();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma(wb[s][i], xb[s][j], acc[4 + i][2 + j]);
        __builtin_amdgcn_s_setprio(0);
        end_m();
        // ---- L4 / M4 ----
        if (n2) stage(sx1, 3, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (n2) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");   // K-tile t + 1 complete (this wave's share): everything but X0 / X1(t+2)
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma(wb[s][i], xa[s][j], acc[4 + i][j]);
        __builtin_amdgcn_s_setprio(0);
        end_m();
    }
    if (!late) end_m();                              // group 0 leaves one interval late: every wave has executed the same number of barriers
}

// Piece-granular main loop for block tiles that are not multiples of the 128-row operand tiles: TNB = 32 FI features x
// TMB = 16 FJ WM tokens, staged as 1-KiB pieces (8 rows x 64 k) whose source is located per piece, so a 96-row tile may
// start anywhere on an 8-row boundary of the tile-major operand.  At M = 720 the 128 x 128 grid covers only 144-192 of the
// 256 CUs and every block fills 512 KB through a ~65 GB/s per-CU LDS-DMA path; 128 x 96 / 96 x 96 tiles give 256 blocks of
// 448 / 384 KB.  2 x WM waves; every wave issues G pieces per stage (the last waves repeat the final piece when the
// piece count does not divide evenly: same bytes to the same place), NS-stage ring with counted vmcnt as in mainloop().
template <bool TR, int NS, int FI, int FJ, int WM, typename AfterPrologue, typename PerStep = NoStepHook>
__device__ __forceinline__ void mainloop_g(const GemmParams& p, char* smem, int n0, int m0, int kt0, int nkt,
                                           f32x4 (&acc)[FI][FJ], BlockStamps& bs, AfterPrologue after_prologue, PerStep per_step = PerStep()) {
    constexpr int NWAVE = 2 * WM;
    constexpr int WPC = 4 * FI, XPC = 2 * FJ * WM, NP = WPC + XPC;   // pieces per stage
    constexpr int G = (NP + NWAVE - 1) / NWAVE;
    constexpr int STAGE_BYTES = NP * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1, wm = w >> 1;
    const int nktot = p.K / TK;
    const int last_wt = ((p.N + 127) >> 7) - 1, last_rt = (p.M - 1) >> 7;
    const char* src[G];
    int dsto[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        int q = w * G + i;
        q = q < NP ? q : NP - 1;
        const bool isw = q < WPC;
        const int row = isw ? n0 + 8 * q : m0 + 8 * (q - WPC);
        int rt = row >> 7;
        const int lim = isw ? last_wt : last_rt;
        rt = rt < lim ? rt : lim;                        // ragged edges re-read a valid tile (results are masked)
        src[i] = (const char*)(isw ? p.W : p.X) + ((size_t)rt * nktot + kt0) * TILE_BYTES + ((row & 127) >> 3) * 1024 + lane * 16;
        dsto[i] = q * 1024;
    }
    auto stage = [&](int t) {
        char* base = smem + (t % NS) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < G; ++i) glds16(src[i] + (size_t)t * TILE_BYTES, base + dsto[i]);
    };
    const int li = lane & 15, g = lane >> 4;
    int woff[2], xoff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ch = ((4 * s + g) ^ (li & 7)) << 4;
        woff[s] = (16 * FI * wn + li) * 128 + ch;
        xoff[s] = WPC * 1024 + (16 * FJ * wm + li) * 128 + ch;
    }
    // second wave of a SIMD: fills after its MFMAs (small-M shapes, deep ring: the partners' fill / MFMA phases overlap).  With the
    // two-stage ring of the large-M tile (two blocks per CU) a late fill sits on the K-step's critical path — it is awaited at the
    // very next barrier — so there every wave fills right after the barrier: QKV 57.6 -> 48.5 us, out-proj 20.5 -> 18.2 back to back
    // at M = 5760, B = 8 forward -2.5 % (profiles/round2/gemm_M5760_l2_prefetch_wave_and_early_fills.txt; debug bit 11 of the
    // experiments build restores the late fills for A/B runs).  An L2 prefetch wave (one load per 128-byte line of K-step t + 3)
    // changed nothing in the same runs: the K-step is not waiting for the fabric.
    const bool late = w >= 4 && (NS > 2 || GTAV_DBG(p, 2048));
    // (an L2 prefetch of the next GEMM's weight, l2_prefetch_next, in front of the prologue fills was measured here for the skinny shapes of the
    // context-cached step, M = 144: 1.49 -> 1.57 ms per step — every wave of these kernels waits on its own vmcnt ring, so the prefetch sits in front of
    // tile 0; only the loader-wave kernels, whose compute waves never wait on vmcnt in the K loop, carry it)
    const int npro = nkt < NS - 1 ? nkt : NS - 1;
    for (int t = 0; t < npro; ++t) stage(t);
    after_prologue();   // register loads the epilogue wants early (bias): behind the first fills, not in front of them
    // wait until this wave's share of tile t has landed (tiles newer than t already issued: min(NS - 2, rem), G loads each),
    // drain this wave's LDS reads, barrier: tile t is visible to everyone and nobody reads tile t - 1 any more
    auto sync = [&](int t) {
        const int rem = nkt - 1 - t;
        kstep_sync_ring<NS, G>(rem);
        if (t == 0) GTAV_STAMP(bs.t[1]);
    };
    auto rd = [&](int t, f16x8 (&wf)[2][FI], f16x8 (&xf)[2][FJ]) {
        const char* b = smem + (t % NS) * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < FI; ++i) wf[s][i] = *(const f16x8*)(b + woff[s] + i * 16 * 128);
#pragma unroll
            for (int j = 0; j < FJ; ++j) xf[s][j] = *(const f16x8*)(b + xoff[s] + j * 16 * 128);
        }
    };
    auto mm = [&](const f16x8 (&wf)[2][FI], const f16x8 (&xf)[2][FJ]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    if (TR) acc[i][j] = mfma16(xf[s][j], wf[s][i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = mfma16(wf[s][i], xf[s][j], acc[i][j], 0, 0, 0);
                }
    };
    if constexpr (FI * FJ >= 16) {
        // Large wave tiles (128 x 48, 96 x 48: 18-24 MFMAs per 32-deep half of a K-step), one block per CU, two waves per SIMD:
        // pipelined by HALF K-steps with one register set per half (fragments (FI + FJ) x 8 registers, the accumulators take 72-96):
        //   barrier t | early fills | read (t, half 0) -> A | MFMA (t - 1, half 1) from B | read (t, half 1) -> B | MFMA (t, half 0) | late fills
        // with one fragment read in each of the first MFMAs' issue shadows.
        f16x8 wa[FI], xa[FJ], wb[FI], xb[FJ];
        auto rdh = [&](int t, int sh, f16x8 (&wf)[FI], f16x8 (&xf)[FJ]) {
            const char* b = smem + (t % NS) * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < FI; ++i) wf[i] = *(const f16x8*)(b + woff[sh] + i * 16 * 128);
#pragma unroll
            for (int j = 0; j < FJ; ++j) xf[j] = *(const f16x8*)(b + xoff[sh] + j * 16 * 128);
        };
        auto mmh = [&](const f16x8 (&wf)[FI], const f16x8 (&xf)[FJ]) {
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    if (TR) acc[i][j] = mfma16(xf[j], wf[i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = mfma16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                }
        };
        const bool domm = !GTAV_DBG(p, 2);
        for (int t = 0; t < nkt; ++t) {
            sync(t);
            const bool refill = t + NS - 1 < nkt && !GTAV_DBG(p, 1);
            if (refill && !late) stage(t + NS - 1);
            per_step(t);
            if (t > 0) {
                rdh(t, 0, wa, xa);
                if (domm) mmh(wb, xb);
                interleave_mfma_dsread<FI * FJ, FI + FJ>();
            } else {
                rdh(t, 0, wa, xa);
            }
            __builtin_amdgcn_sched_barrier(0);
            rdh(t, 1, wb, xb);
            if (domm) mmh(wa, xa);
            interleave_mfma_dsread<FI * FJ, FI + FJ>();
            __builtin_amdgcn_sched_barrier(0);
            if (refill && late) {
                asm volatile("" ::: "memory");
                stage(t + NS - 1);
            }
        }
        if (domm) mmh(wb, xb);
    } else if constexpr (NS >= 3) {
        // Small-M shapes (one block per CU, 1.5 waves per SIMD): nothing else on the SIMD covers a wave's LDS-read latency, and
        // with reads and MFMAs of the same K-step in one iteration every K-step exposed it four times (the compiler interleaves
        // 4 reads / wait / 4 MFMAs: ~0.58 us per K-step at M = 720, profiles/round2 stamps).  One-step software pipeline instead:
        // iteration t issues the fragment reads of tile t into one register set and runs the MFMAs of tile t - 1 from the other,
        // so the reads are in flight under 16-24 MFMAs.  Tile t - 1's stage is refilled after barrier t: every wave drained its
        // reads of it (lgkmcnt(0)) before that barrier.  Two register sets, loop unrolled by two for static indexing.
        f16x8 wA[2][FI], xA[2][FJ], wB[2][FI], xB[2][FJ];
        auto step = [&](int t, f16x8 (&wr)[2][FI], f16x8 (&xr)[2][FJ], const f16x8 (&wm_)[2][FI], const f16x8 (&xm_)[2][FJ]) {
            sync(t);
            const bool refill = t + NS - 1 < nkt && !GTAV_DBG(p, 1);
            if (refill && !late) stage(t + NS - 1);
            per_step(t);
            if (t > 0 && !GTAV_DBG(p, 2)) {
                rd(t, wr, xr);
                mm(wm_, xm_);
                interleave_mfma_dsread<2 * FI * FJ, 2 * (FI + FJ), 1>();   // one read in each of the first MFMAs' issue shadows (not a block of reads first)
            } else {
                rd(t, wr, xr);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (refill && late) {
                asm volatile("" ::: "memory");
                stage(t + NS - 1);
            }
        };
        int t = 0;
        for (; t + 1 < nkt; t += 2) {
            step(t, wA, xA, wB, xB);
            step(t + 1, wB, xB, wA, xA);
        }
        if (t < nkt) {
            step(t, wA, xA, wB, xB);
            if (!GTAV_DBG(p, 2)) mm(wA, xA);
        } else if (!GTAV_DBG(p, 2)) {
            mm(wB, xB);
        }
    } else if (NS == 2 && !GTAV_DBG(p, 8192)) {
        // Two-stage ring (large M, two blocks per CU).  A wave that issues LDS-DMA is held until the address pipe takes the instruction; with the
        // whole block's 40 KiB issued in one burst right after the barrier, every wave queued for up to 640 cycles BEFORE its fragment reads
        // and MFMAs (alone on a CU a block ran its K-step in 1 590 cycles for 768 cycles of MFMA issue, tools/gemm_stamps.py).  Here the G fills
        // of a wave are spread between its MFMAs (sched_group_barrier: 4 MFMAs, 1 fill, ...): the queue never backs up, the fill costs issue
        // slots inside the MFMA shadows.  The last K-step has no refill and is peeled so that the loop body is branch-free.
        constexpr int NMF = 2 * FI * FJ, PERF = NMF / (G + 1) > 0 ? NMF / (G + 1) : 1;
        for (int t = 0; t + 1 < nkt; ++t) {
            sync(t);
            f16x8 wf[2][FI], xf[2][FJ];
            rd(t, wf, xf);
            if (!GTAV_DBG(p, 1)) stage(t + 1);
            per_step(t);
            mm(wf, xf);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, PERF, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            if constexpr (NMF > PERF * G) __builtin_amdgcn_sched_group_barrier(0x008, NMF - PERF * G, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            sync(nkt - 1);
            f16x8 wf[2][FI], xf[2][FJ];
            rd(nkt - 1, wf, xf);
            per_step(nkt - 1);
            mm(wf, xf);
        }
    } else {
        for (int t = 0; t < nkt; ++t) {
            sync(t);
            const bool refill = t + NS - 1 < nkt && !GTAV_DBG(p, 1);
            if (refill && !late) stage(t + NS - 1);
            per_step(t);
            f16x8 wf[2][FI], xf[2][FJ];
            rd(t, wf, xf);
            mm(wf, xf);
            if (refill && late) {
                asm volatile("" ::: "memory");
                stage(t + NS - 1);
            }
        }
    }
}

// XCD-aware, bijective block -> tile map: blocks that share an XCD (equal bid % 8) get a contiguous run of tiles.
// Inside an XCD's run the order is (m, n) with n FASTEST over a group of `gn` n-panels: co-resident blocks then share X
// row-tiles as well as W panels in that XCD's 4 MiB L2.  With m fastest, X (11.8 MB at M = 5760) was re-streamed from the
// fabric once per n-panel: rocprofv3 FETCH_SIZE 301 MB per fc1 launch against 20 MB algorithmic (profiles/round1/pmc).
// Split-K: the K slice is the slowest index, so the slices of one W panel stay on one XCD.  gn is chosen on the host
// (choose_gn): round 1 fixed it at tiles_n / 8 — every XCD then owns an eighth of the n-panels and streams ALL of X, which for
// the N = 1024 GEMMs at large M is 8 x X from the fabric (fc2 at M = 5760: 385 MB per launch against 55 MB algorithmic).
template <bool SPLITK, int TNB, int TMB>
__device__ __forceinline__ void tile_map(const GemmParams& p, int& n0, int& m0, int& ks, int& kt0, int& nkt) {
    const int tiles_m = (p.M + TMB - 1) / TMB, tiles_n = (p.N + TNB - 1) / TNB;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int swz = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    int tile_id = swz;
    ks = 0, kt0 = 0, nkt = p.K / TK;
    if constexpr (SPLITK) {
        const int tiles = tiles_m * tiles_n;
        ks = swz / tiles;
        tile_id = swz - ks * tiles;
        nkt = nkt / p.splitk;
        kt0 = ks * nkt;
    }
    int gn = p.tm.gn;                                      // host: choose_gn() (every launcher fills it)
    const int group = tiles_m * gn;
    const int ng = tile_id / group, rem = tile_id - ng * group;
    const int n_first = ng * gn;
    if (n_first + gn > tiles_n) gn = tiles_n - n_first;   // last, partial group
    const int tile_m = rem / gn, tile_n = n_first + (rem - tile_m * gn);
    n0 = tile_n * TNB;
    m0 = tile_m * TMB;
}

// The same map from host-precomputed constants (GemmParams::tm, filled by launch_l): no run-time integer division.
__device__ __forceinline__ int div_rcp(int a, unsigned rcp) { return rcp ? (int)__umulhi((unsigned)a, rcp) : a; }   // rcp == 0 encodes a divisor of 1
template <bool SPLITK, int TNB, int TMB>
__device__ __forceinline__ void tile_map_fast(const GemmParams& p, int& n0, int& m0, int& ks, int& kt0, int& nkt) {
    const GemmParams::TileMap& tm = p.tm;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int swz = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    int tile_id = swz;
    ks = 0, kt0 = 0, nkt = p.K / TK;
    if constexpr (SPLITK) {
        ks = div_rcp(swz, tm.rcp_tiles);
        tile_id = swz - ks * tm.tiles;
        nkt = nkt / p.splitk;       // a power of two or a small divisor: one division, off the loaders' critical path
        kt0 = ks * nkt;
    }
    const int ng = div_rcp(tile_id, tm.rcp_group), rem = tile_id - ng * tm.group;
    const int n_first = ng * tm.gn;
    const bool last = n_first + tm.gn > tm.tiles_n;      // last, partial group of n-panels
    const int gn = last ? tm.tiles_n - n_first : tm.gn;
    const int tile_m = div_rcp(rem, last ? tm.rcp_gnlast : tm.rcp_gn), tile_n = n_first + (rem - tile_m * gn);
    n0 = tile_n * TNB;
    m0 = tile_m * TMB;
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm fold, consumer side (EPI_*_FOLD; docs/LABNOTES.md 4.7).  The X operand is A[m][k] = x[m][k] (1 + scale[f][k]) of the residual row x
// the LayerNorm would have normalised (f = frame of token m), so with mean / rstd of the row and the per-frame tables
//   c1[f][n] = sum_k (1 + scale[f][k]) W[n][k],   c2[f][n] = sum_k shift[f][k] W[n][k] + bias[n]
// the GEMM of the modulated LayerNorm output is   y[m][n] = (acc[m][n] - mean_m c1[f][n]) rstd_m + c2[f][n]   (model/dit.py:19-27).
// mean / rstd come from the producer's partial sums (sum x, sum x^2) per 64-feature slot: lane (li, g) adds quarter g of its token's
// slots in slot order, two xor-shuffles combine the quarters — ONE fixed order for every consumer block of that row.
// A wave's 16-token groups never straddle a frame (f_P % 16 == 0, tiles start on multiples of 16).  The c1 / c2 slices of the block tile
// — TNB features of the (at most FOLD_NFR) frames its tokens belong to — are staged in LDS once per block (fold_stage_tables, behind the
// main loop: the ring is dead) and read from there per MFMA tile: no table registers are carried and no wave waits on its own global loads.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int FOLD_NFR = 4;                          // frames a block tile can touch: TM <= 192 tokens, f_P >= 64 (host-checked)
constexpr int fold_lds_bytes(int tnb) { return FOLD_NFR * 2 * tnb * 4; }
template <int FJ>
struct FoldTok {
    float mu[FJ], rs[FJ];       // mean / 1 / sqrt(var + eps) of token column j (token li of the lane)
    int b1, b2, b3;             // first tokens of the block tile's 2nd / 3rd / 4th frame
    int kmax;                   // last staged frame (token groups past M — ragged last tile — use it: their results are dropped, but must stay finite)
    int nchunk;                 // 16-byte chunks of the block's table slice: [frame k][c1 | c2][TNB] floats
    f32x4 tpre;                 // this thread's chunk of it, fetched before the main loop
    const char* cb;             // LDS copy, written in the epilogue (fold_tables_to_lds)
};
// frame (inside the block tile) of the 16-token group that starts at token m
template <int FJ>
__device__ __forceinline__ int fold_frame_of(const FoldTok<FJ>& ft, int m) {
    const int k = (m >= ft.b1 ? 1 : 0) + (m >= ft.b2 ? 1 : 0) + (m >= ft.b3 ? 1 : 0);
    return k < ft.kmax ? k : ft.kmax;
}
// Everything the folded epilogue needs from memory is fetched BEFORE the main loop, behind the first fills (like the bias): the block's
// table slice, one 16-byte chunk per compute thread `ct` (kept in 4 registers through the K loop, written to LDS when the ring is dead),
// and the row statistics, reduced to (mean, rstd) at once (2 FJ registers).  In the epilogue the same loads were two dependent memory round
// trips in front of every tile's tail — 10 us per launch at M = 5760, where the 512 resident blocks reach their epilogues together
// (profiles/round3/fold_v1_ab_B8.txt).
template <int TNB, int TM, int FJ>
__device__ __forceinline__ void fold_prefetch(const GemmParams& p, int n0, int m0, int ct /* compute-thread index */, int mw /* first token of the wave */,
                                              bool compute_wave, FoldTok<FJ>& ft) {
    const int mlast = m0 + TM - 1 < p.M ? m0 + TM - 1 : p.M - 1;
    const int f_first = m0 / p.f_P, nfr = mlast / p.f_P - f_first + 1;
    ft.b1 = (f_first + 1) * p.f_P; ft.b2 = ft.b1 + p.f_P; ft.b3 = ft.b2 + p.f_P;
    ft.kmax = nfr - 1;
    constexpr int Q = TNB / 4;                        // 16-byte chunks per table row slice
    ft.nchunk = nfr * 2 * Q;
    ft.tpre = f32x4{0.f, 0.f, 0.f, 0.f};
    ft.cb = nullptr;
    if (ct >= 0 && ct < ft.nchunk) {
        const int k = ct / (2 * Q), r = ct - k * 2 * Q, t = r / Q, n4 = r - t * Q;
        int n = n0 + 4 * n4;
        n = n < p.N ? n : p.N - 4;
        const int row = p.f_rows ? p.f_rows[f_first + k] : f_first + k;
        ft.tpre = *(const f32x4*)((t ? p.f_c2 : p.f_c1) + (size_t)row * p.f_ldc + n);
    }
    if (!compute_wave) return;
    // statistics: 64-feature slots (sum x, sum x^2); lane (li, g) adds quarter g of its token's slots in slot order, two xor-shuffles
    // combine the quarters: ONE fixed order for every consumer block of the row
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int q = p.f_nslot >> 2;                       // slots per lane quarter
    const float inv_d = 1.0f / (float)(p.f_nslot * 64);
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
        int m = mw + 16 * j + li;
        m = m < p.M ? m : p.M - 1;
        const float* sp = p.f_stats + ((size_t)m * p.f_nslot + g * q) * 2;
        float a1 = 0.f, a2 = 0.f;
        if (q & 1) {
            for (int s = 0; s < q; ++s) {
                const float2 v = *(const float2*)(sp + 2 * s);
                a1 += v.x; a2 += v.y;
            }
        } else {
            for (int s = 0; s < q; s += 2) {
                const f32x4 v = *(const f32x4*)(sp + 2 * s);
                a1 += v[0]; a2 += v[1];
                a1 += v[2]; a2 += v[3];
            }
        }
        a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64);
        a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
        const float mu = a1 * inv_d;
        ft.mu[j] = mu;
        ft.rs[j] = 1.0f / sqrtf(fmaxf(a2 * inv_d - mu * mu, 0.f) + 1e-6f);
    }
}
// epilogue, ring dead: the prefetched table chunks into LDS (the caller barriers afterwards)
template <int FJ>
__device__ __forceinline__ void fold_tables_to_lds(FoldTok<FJ>& ft, char* cb, int ct) {
    ft.cb = cb;
    if (ct >= 0 && ct < ft.nchunk) *(f32x4*)(cb + (size_t)ct * 16) = ft.tpre;
}
// producer side: a wave's token span (<= 64) is at most one frame long, so it meets at most two frames — the one of its first token (A)
// and the next (B)
__device__ __forceinline__ void fold_frames(const GemmParams& p, int mw, int& rowA, int& rowB, int& boundary) {
    const int mf = mw < p.M ? mw : p.M - 1;
    const int f0 = mf / p.f_P, flast = (p.M - 1) / p.f_P;
    const int f1 = f0 < flast ? f0 + 1 : flast;
    boundary = (f0 + 1) * p.f_P;
    rowA = p.f_rows ? p.f_rows[f0] : f0;
    rowB = p.f_rows ? p.f_rows[f1] : f1;
}
__device__ __forceinline__ f32x4 fold_apply(const f32x4 acc, float mu, float rs, const f32x4 c1, const f32x4 c2) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(__builtin_fmaf(-mu, c1[e], acc[e]), rs, c2[e]);
    return r;
}

// QKV epilogue staged through LDS: bias + RoPE happen in registers (pairs are lane-local), the block's fp16 result is laid
// out in LDS in the shape of its DESTINATION rows (q/k: [token][feature], V^T: [feature][token]) and leaves as 16-byte
// stores — a whole 128-byte head row of one token (or 8 consecutive tokens of one V^T row) per 8 lanes — instead of 8-byte
// stores scattered over 16 rows per wave-instruction.  tab[] holds the per-token destination coordinates (one integer
// division per token instead of one per lane and token).
// The (cos, sin) values of the RoPE rotation a compute wave will apply in qkv_staged, fetched BEFORE the main loop (loader-wave kernels:
// behind the first fills, like the bias): in the epilogue those loads were a memory round trip at the head of every QKV launch's tail.
template <int FI, int FJ, int WM, int WN, int WOFF>
__device__ __forceinline__ void prefetch_rope(const GemmParams& p, int n0, int m0, f32x4 (&prope)[FI][FJ]) {
    const int lane = threadIdx.x & 63, w = (int)(threadIdx.x >> 6) - WOFF;
    const int wn = w % WN, wm = w / WN, li = lane & 15, g = lane >> 4;
    const bool spatial = p.qkv_mode == QKV_SPATIAL;
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
        int m = m0 + 16 * FJ * wm + 16 * j + li;
        m = m < p.M ? m : p.M - 1;
        const int fr = m / p.S;
        const int pos = spatial ? m - fr * p.S : p.t0 + fr % p.Tq;
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int n = n0 + 16 * FI * wn + 16 * i + 4 * g;
            prope[i][j] = f32x4{1.f, 0.f, 1.f, 0.f};
            if (w >= 0 && w < WN * WM && n < 2 * p.D) prope[i][j] = *(const f32x4*)(rope_tab(p, n) + pos * 64 + (n & 63));
        }
    }
}

template <int FI, int FJ, int WM, int WN = 2, int WOFF = 0, bool PRE = false, bool FOLD = false>   // PRE: prope holds the prefetched (cos, sin) values (registers, static indexing)
__device__ __forceinline__ void qkv_staged(const GemmParams& p, f32x4 (&acc)[FI][FJ], const f32x4 (&pbias)[FI], char* smem, int n0, int m0, bool tr,
                                           const f32x4 (&prope)[FI][FJ], FoldTok<FJ>& ft) {
    constexpr int TM = WM * 16 * FJ, TNB = 16 * FI * WN;
    const int lane = threadIdx.x & 63, w = (int)(threadIdx.x >> 6) - WOFF;   // WOFF leading waves are loader waves (mainloop_l)
    const int wn = w % WN, wm = w / WN, li = lane & 15, g = lane >> 4;
    const bool compute_wave = w >= 0 && w < WN * WM;
    const bool spatial = p.qkv_mode == QKV_SPATIAL;
    constexpr int PN = (TNB / 8 + 7) / 8 * 8 * 16;   // LDS bytes per token row (q/k image): 16-byte chunks rounded up to 8
    constexpr int PT = (TM / 8 + 7) / 8 * 8 * 16;    // LDS bytes per feature row (V^T image)
    int2* tab = (int2*)(smem + (TM * PN > TNB * PT ? TM * PN : TNB * PT));
    float amax = 0.f;
    __syncthreads();   // every wave is done reading the last K-step's stage
    if constexpr (FOLD) fold_tables_to_lds<FJ>(ft, (char*)(tab + TM), (int)threadIdx.x - 64 * WOFF);
    for (int r = threadIdx.x; r < TM; r += (int)blockDim.x) {
        const int m = m0 + r;
        int a = -1, b = 0;
        if (m < p.M) {
            const int fr = m / p.S;                     // spatial: attention item; temporal: frame counter over (b, tl)
            if (spatial) {
                a = fr;
                b = m - fr * p.S;
            } else {
                const int bb = fr / p.Tq;
                const int tfr = p.t0 + (fr - bb * p.Tq);
                a = (bb * p.Tmax + tfr) * p.S + (m - fr * p.S);   // kv-cache token slot
                b = tfr;
            }
        }
        tab[r] = int2{a, b};
    }
    if constexpr (FOLD) __syncthreads();   // the staged c1 / c2 slices are visible
    if (compute_wave) {
        if (tr) {
            // D[row = token][col = feature]: the lane owns tokens ml..ml+3 of feature nl -> V^T image [feature][token]
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int nl = 16 * FI * wn + 16 * i + li;
                const float bv = (p.bias && n0 + nl < p.N) ? p.bias[n0 + nl] : 0.f;
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    const int ml = 16 * FJ * wm + 16 * j + 4 * g;
                    char* dst = smem + nl * PT + (((ml >> 3) ^ (nl & 7)) << 4) + ((ml >> 2) & 1) * 8;
                    const f32x4 a = acc[i][j];
                    if constexpr (FOLD) {
                        // the lane owns tokens 4 g .. 4 g + 3 of the group: their statistics live in lanes li = 4 g + e
                        const float* cf = (const float*)ft.cb + fold_frame_of(ft, m0 + 16 * FJ * wm + 16 * j) * 2 * TNB + nl;
                        const float c1 = cf[0], c2 = cf[TNB];
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float mu = __shfl(ft.mu[j], 4 * g + e, 64), rs = __shfl(ft.rs[j], 4 * g + e, 64);
                            o[e] = __builtin_fmaf(__builtin_fmaf(-mu, c1, a[e]), rs, c2);
                        }
                        *(uint2*)dst = pack4(amax, o[0], o[1], o[2], o[3]);
                    } else {
                        *(uint2*)dst = pack4(amax, a[0] + bv, a[1] + bv, a[2] + bv, a[3] + bv);
                    }
                }
            }
        } else {
            int pos[FJ];
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                int m = m0 + 16 * FJ * wm + 16 * j + li;
                m = m < p.M ? m : p.M - 1;
                const int fr = m / p.S;
                pos[j] = spatial ? m - fr * p.S : p.t0 + fr % p.Tq;
            }
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int nl = 16 * FI * wn + 16 * i + 4 * g;
                const int n = n0 + nl;
                const f32x4 bv = pbias[i];
                const bool rope = n < 2 * p.D;
                const int d = n & 63;                    // D % 64 == 0: the offset inside the head
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    const int ml = 16 * FJ * wm + 16 * j + li;
                    f32x4 v;
                    if constexpr (FOLD) {
                        const float* cf = (const float*)ft.cb + fold_frame_of(ft, m0 + 16 * FJ * wm + 16 * j) * 2 * TNB + nl;
                        v = fold_apply(acc[i][j], ft.mu[j], ft.rs[j], *(const f32x4*)cf, *(const f32x4*)(cf + TNB));
                    } else {
                        v = acc[i][j] + bv;
                    }
                    if (rope) {
                        f32x4 cs;
                        if constexpr (PRE) cs = prope[i][j];
                        else cs = *(const f32x4*)(rope_tab(p, n) + pos[j] * 64 + d);
                        v = rope4(v, cs);
                    }
                    char* dst = smem + ml * PN + (((nl >> 3) ^ (ml & 7)) << 4) + ((nl >> 2) & 1) * 8;
                    *(uint2*)dst = pack4(amax, v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    sat_report(amax, p.err_flag);
    __syncthreads();
    const int heads = p.D >> 6;
    if (tr) {
        constexpr int CPR = TM / 8;                     // 16-byte chunks (8 tokens) per feature row
        for (int q = threadIdx.x; q < TNB * CPR; q += (int)blockDim.x) {
            const int nl = q / CPR, tc = q % CPR;
            const int n = n0 + nl;
            const int2 t = tab[tc * 8];
            if (n >= p.N || t.x < 0) continue;
            const uint4 val = *(const uint4*)(smem + nl * PT + ((tc ^ (nl & 7)) << 4));
            const int nn = n - 2 * p.D;
            f16* dst = p.v + ((size_t)(t.x * heads + (nn >> 6)) * 64 + (nn & 63)) * p.S + t.y;
            if (p.out_sc1) store16q_sc1(dst, val);
            else *(uint4*)dst = val;
        }
    } else {
        constexpr int CPR = TNB / 8;                    // 16-byte chunks (8 features) per token row
        for (int q = threadIdx.x; q < TM * CPR; q += (int)blockDim.x) {
            const int ml = q / CPR, c = q % CPR;
            const int n = n0 + 8 * c;
            const int2 t = tab[ml];
            if (n >= p.N || t.x < 0) continue;
            const uint4 val = *(const uint4*)(smem + ml * PN + ((c ^ (ml & 7)) << 4));
            const int which = n >= 2 * p.D ? 2 : (n >= p.D ? 1 : 0);
            const int nn = n - which * p.D;
            f16* dst;
            if (spatial) dst = (which == 0 ? p.q : p.k) + ((size_t)(t.x * heads + (nn >> 6)) * p.S + t.y) * 64 + (nn & 63);
            else if (which == 0) dst = p.q + (size_t)(m0 + ml) * p.D + nn;
            else dst = p.k + (size_t)t.x * 2 * p.D + (which == 2 ? p.D : 0) + nn;
            if (p.out_sc1) store16q_sc1(dst, val);
            else *(uint4*)dst = val;
        }
    }
}

// Epilogue shared by every block shape.  The block tile is TNB = 32 FI features x TM = 16 FJ WM tokens; wave (wn, wm) owns
// features 16 FI wn .. and tokens 16 FJ wm ..; acc[i][j] is the 16 x 16 MFMA tile (feature group i, token group j).
// The bias of the non-transposed epilogues is fetched BEFORE the main loop (prefetch_bias): its L2 / HBM round trip used to
// sit at the head of every epilogue.
template <int EPIX, int FI, int WM, int WN = 2, int WOFF = 0>
__device__ __forceinline__ void prefetch_bias(const GemmParams& p, int n0, f32x4 (&pbias)[FI]) {
    constexpr int EPI = epi_base(EPIX);
    const int w = (int)(threadIdx.x >> 6) - WOFF;
    const int lane = threadIdx.x & 63, wn = w % WN, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        const int n = n0 + 16 * FI * wn + 16 * i + 4 * g;
        pbias[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI != EPI_PARTIAL && p.bias && n < p.N && w >= 0 && w < WN * WM) pbias[i] = *(const f32x4*)(p.bias + n);
    }
}

// EPI_RESID_FOLD: the residual tile the epilogue will update is fetched INSIDE the main loop, one 16-byte load per lane and K-step (MFMA tile
// q = i FJ + j at K-step q; tiles [qlo, qhi) here), into FI x FJ x 4 registers.  Not in front of the loop: the two-stage rings wait vmcnt(0) at
// every K-step, so a burst of FI FJ loads issued behind the prologue fills is awaited in full — 64 KB per block from HBM — before the first
// MFMA (fold v2: out-proj at M = 5760 33.5 us against 22.3 us for the slab GEMM, 6 us of it this read; profiles/round3).
template <int FI, int FJ, int WM, int WN = 2, int WOFF = 0>
__device__ __forceinline__ void prefetch_resid(const GemmParams& p, int n0, int m0, f32x4 (&xpre)[FI][FJ], int qlo, int qhi) {
    const int lane = threadIdx.x & 63, w = (int)(threadIdx.x >> 6) - WOFF;
    const int wn = w % WN, wm = w / WN, li = lane & 15, g = lane >> 4;
    const bool cw = w >= 0 && w < WN * WM;
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        const int n = n0 + 16 * FI * wn + 16 * i + 4 * g;
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int q = i * FJ + j;
            if (q < qlo || q >= qhi) continue;
            const int m = m0 + 16 * FJ * wm + 16 * j + li;
            xpre[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (cw && n < p.N && m < p.M) xpre[i][j] = *(const f32x4*)((const float*)p.out + (size_t)m * p.ldo + n);
        }
    }
}

// prope: registers fetched before the main loop when PRE — the RoPE (cos, sin) values of a QKV epilogue, or the residual tile of EPI_RESID_FOLD;
// ft: the fold consumer's prefetched statistics / table chunk (fold_prefetch), unused otherwise
template <int EPIX, int FI, int FJ, int WM, int WN = 2, int WOFF = 0, bool PRE = false>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x4 (&acc)[FI][FJ], const f32x4 (&pbias)[FI], char* smem, int n0, int m0, int ks, bool tr,
                                         const f32x4 (&prope)[FI][FJ], FoldTok<FJ>& ft) {
    constexpr int EPI = epi_base(EPIX);
    constexpr bool FOLD = epi_is_fold_consumer(EPIX);   // LayerNorm fold, consumer side: v = (acc - mean c1) rstd + c2 instead of acc + bias
    constexpr int TM = WM * 16 * FJ, TNB = 16 * FI * WN;
    constexpr int CT = 16 * FI * WN / 64;           // 64-feature sub-tiles per block tile row
    const int lane = threadIdx.x & 63, wraw = threadIdx.x >> 6, w = wraw - WOFF;   // WOFF leading waves are loader waves (mainloop_l)
    const int wn = w % WN, wm = w / WN, li = lane & 15, g = lane >> 4;
    const bool compute_wave = w >= 0 && w < WN * WM;
    float amax = 0.f;

    if constexpr (EPI == EPI_RESID_FOLD) {
        // LayerNorm fold, producer side: the gated residual update in place, the next GEMM's operand A = x (1 + scale_next + 1e-6)
        // (fp16, assembled in LDS in the tile-major image and copied out as 1-KiB pieces like the GELU output) and the row's partial
        // sums per 64-feature slot for the consumer's mean / rstd.  PRE: the residual tile was fetched before the main loop (prope) — the
        // read half of the read-modify-write is then off the tile's tail (it cost 14 us per launch at M = 5760, where every block reaches
        // its epilogue at the same time: profiles/round3/fold_v1_ab_B8.txt).
        static_assert(EPI != EPI_RESID_FOLD || (FI == 2 || FI == 4), "a 64-feature statistics slot is one wave (FI = 4) or two (FI = 2)");
        constexpr int WPS = 4 / FI;                     // waves per statistics slot
        float2* sc = (float2*)(smem + CT * TM * 128);   // [wn][token of the tile]: per-wave partial sums, combined after the barrier
        __syncthreads();   // every wave is done reading the last K-step's stage
        if (compute_wave) {
            int rowA, rowB, boundary;
            fold_frames(p, __builtin_amdgcn_readfirstlane(m0 + 16 * FJ * wm), rowA, rowB, boundary);
            const float *gA = p.gate + (size_t)rowA * p.gate_stride, *gB = p.gate + (size_t)rowB * p.gate_stride;
            const float *sA = p.f_scale + (size_t)rowA * p.gate_stride, *sB = p.f_scale + (size_t)rowB * p.gate_stride;
            float s1[FJ], s2[FJ];
#pragma unroll
            for (int j = 0; j < FJ; ++j) s1[j] = 0.f, s2[j] = 0.f;
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int nl = 16 * FI * wn + 16 * i + 4 * g;
                const int n = n0 + nl;
                const bool nok = n < p.N;
                const int nc = nok ? n : 0;
                const f32x4 bv = pbias[i];
                f32x4 gA4, gB4, sA4, sB4;
                if (GTAV_DBG(p, 0x100000)) {   // (experiments build: timing without the gate / scale loads)
                    gA4 = gB4 = f32x4{0.5f, 0.5f, 0.5f, 0.5f}; sA4 = sB4 = f32x4{0.1f, 0.1f, 0.1f, 0.1f};
                } else {
                    gA4 = *(const f32x4*)(gA + nc); gB4 = *(const f32x4*)(gB + nc);
                    sA4 = *(const f32x4*)(sA + nc); sB4 = *(const f32x4*)(sB + nc);
                }
                const int c = nl & 63;
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    const int ml = 16 * FJ * wm + 16 * j + li;
                    const int m = m0 + ml;
                    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (nok && m < p.M) {
                        float* dst = (float*)p.out + (size_t)m * p.ldo + n;
                        const bool inA = m0 + 16 * FJ * wm + 16 * j < boundary;
                        const f32x4 gt = inA ? gA4 : gB4, sc4 = inA ? sA4 : sB4;
                        f32x4 x;
                        if constexpr (PRE) x = prope[i][j];
                        else x = *(const f32x4*)dst;
                        x = x + gt * (acc[i][j] + bv);
                        if (!GTAV_DBG(p, 0x10000)) {
                            if (p.out_sc1 && !GTAV_DBG(p, 0x80000)) store16_sc1(dst, x);
                            else *(f32x4*)dst = x;
                        }
                        s1[j] += (x[0] + x[1]) + (x[2] + x[3]);
                        s2[j] += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[e] = x[e] * (1.0f + (sc4[e] + 1e-6f));
                    }
                    char* ld = smem + ((nl >> 6) * TM + ml) * 128 + (((c >> 3) ^ (ml & 7)) << 4) + (c & 7) * 2;
                    *(uint2*)ld = pack4(amax, a[0], a[1], a[2], a[3]);
                }
            }
#pragma unroll
            for (int j = 0; j < FJ; ++j) {      // the wave's 16 FI features of token li: the four lane groups g hold a quarter each
                float t1 = s1[j], t2 = s2[j];
                t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
                t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
                if (g == 0) sc[wn * TM + 16 * FJ * wm + 16 * j + li] = float2{t1, t2};
            }
        }
        sat_report(amax, p.err_flag);
        __syncthreads();
        {   // statistics slots of the tile: the waves that share a slot, added in wave order
            const int nslot = p.N >> 6;
            for (int q = threadIdx.x; q < TM * (TNB / 64); q += (int)blockDim.x) {
                const int sl = q / TM, tok = q - sl * TM;
                float2 v = sc[sl * WPS * TM + tok];
                if constexpr (WPS == 2) { const float2 u = sc[(sl * WPS + 1) * TM + tok]; v.x += u.x; v.y += u.y; }
                const int m = m0 + tok, slot = (n0 >> 6) + sl;
                if (m < p.M && slot < nslot && !GTAV_DBG(p, 0x40000)) *(float2*)(p.f_stats_out + ((size_t)m * nslot + slot) * 2) = v;
            }
        }
        const int nkt_out = p.N >> 6, last_rt = (p.M - 1) >> 7;
        constexpr int PR = TM / 8;
        for (int q = wraw; q < CT * PR; q += (int)(blockDim.x >> 6)) {
            const int cs = q / PR, pq = q - cs * PR;
            const int gr = m0 + 8 * pq;
            const int rt = gr >> 7;
            if (rt > last_rt || (n0 >> 6) + cs >= nkt_out || GTAV_DBG(p, 0x20000)) continue;
            const uint4 val = *(const uint4*)(smem + (cs * TM + 8 * pq) * 128 + lane * 16);
            char* dst = (char*)p.f_a + ((size_t)rt * nkt_out + (n0 >> 6) + cs) * TILE_BYTES + ((gr & 127) >> 3) * 1024 + lane * 16;
            if (p.out_sc1) store16q_sc1(dst, val);
            else *(uint4*)dst = val;
        }
        return;
    }

    if constexpr (EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF || EPI == EPI_F16_TILED) {
        // The fp16 output is the next GEMM's A operand (tile-major).  The block's TM x 128 result is assembled in LDS in
        // that image's row format — CT sub-tile columns of [TM tokens][64 features], rows swizzled like the destination —
        // and then copied out as fully contiguous 1 KiB pieces (8 token rows, 16 B per lane) instead of 16 scattered 8-byte
        // stores per lane.  Works for any block tile whose first token row is a multiple of 8 (96-token tiles included).
        __syncthreads();   // every wave is done reading the last K-step's stage
        if constexpr (FOLD) {
            fold_tables_to_lds<FJ>(ft, smem + CT * TM * 128, (int)threadIdx.x - 64 * WOFF);
            __syncthreads();
        }
        if (compute_wave) {
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int nl = 16 * FI * wn + 16 * i + 4 * g;           // feature inside the block tile (4 consecutive)
                const f32x4 bv = pbias[i];
                const int c = nl & 63;                              // feature inside the 64-wide sub-tile (= 16 i + 4 g)
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    const int ml = 16 * FJ * wm + 16 * j + li;      // token inside the block tile
                    char* dst = smem + ((nl >> 6) * TM + ml) * 128 + (((c >> 3) ^ (ml & 7)) << 4) + (c & 7) * 2;
                    f32x4 v;
                    if constexpr (FOLD) {
                        const float* cf = (const float*)ft.cb + fold_frame_of(ft, m0 + 16 * FJ * wm + 16 * j) * 2 * TNB + nl;
                        v = fold_apply(acc[i][j], ft.mu[j], ft.rs[j], *(const f32x4*)cf, *(const f32x4*)(cf + TNB));
                    } else {
                        v = acc[i][j] + bv;
                    }
                    if constexpr (EPI == EPI_GELU_TANH)
                        {
                            if (GTAV_DBG(p, 16384)) {   // experiments build: the scalar form, for A/B runs
                                *(uint2*)dst = pack4(amax, gelu_tanh_f(v[0]), gelu_tanh_f(v[1]), gelu_tanh_f(v[2]), gelu_tanh_f(v[3]));
                            } else {
                                const float xin[4] = {v[0], v[1], v[2], v[3]};
                                float yo[4];
                                gelu_tanh_f4(xin, yo);
                                *(uint2*)dst = pack4(amax, yo[0], yo[1], yo[2], yo[3]);
                            }
                        }
                    else if constexpr (EPI == EPI_GELU_ERF) {
                        const float xin[4] = {v[0], v[1], v[2], v[3]};
                        float yo[4];
                        gelu_erf_f4(xin, yo);
                        *(uint2*)dst = pack4(amax, yo[0], yo[1], yo[2], yo[3]);
                    } else
                        *(uint2*)dst = pack4(amax, v[0], v[1], v[2], v[3]);
                }
            }
        }
        sat_report(amax, p.err_flag);
        __syncthreads();
        const int nkt_out = p.ldo >> 6, last_rt = (p.M - 1) >> 7;
        constexpr int PR = TM / 8;                      // 1-KiB pieces (8 token rows) per 64-feature sub-tile column
        auto copy_out = [&](void* out) {
            for (int q = wraw; q < CT * PR; q += (int)(blockDim.x >> 6)) {
                const int cs = q / PR, pq = q - cs * PR;
                const int gr = m0 + 8 * pq;                 // first token row of the piece (m0 % 8 == 0)
                const int rt = gr >> 7;
                if (rt > last_rt || (n0 >> 6) + cs >= nkt_out) continue;   // ragged last block tile (tokens / features)
                const uint4 val = *(const uint4*)(smem + (cs * TM + 8 * pq) * 128 + lane * 16);
                char* dst = (char*)out + ((size_t)rt * nkt_out + (n0 >> 6) + cs) * TILE_BYTES + ((gr & 127) >> 3) * 1024 + lane * 16;
                if (p.out_sc1) store16q_sc1(dst, val);
                else *(uint4*)dst = val;
            }
        };
        copy_out(p.out);
        if constexpr (EPI == EPI_F16_TILED && !FOLD) {
            // Training forward (fc1): a SECOND tile-major image, GELU-tanh of the fp16 values just written (p.out2: h = GELU(u) beside the pre-activation u the
            // backward pass needs) — from the accumulators, rounded to fp16 first, so h is bit-identical to what the flat elementwise kernel computed from u;
            // that kernel read and wrote 94 MB per launch at batch 16.
            if (p.out2) {
                __syncthreads();   // the first image has been copied out
                if (compute_wave) {
                    float dummy = 0.f;
#pragma unroll
                    for (int i = 0; i < FI; ++i) {
                        const int nl = 16 * FI * wn + 16 * i + 4 * g;
                        const f32x4 bv = pbias[i];
                        const int c = nl & 63;
#pragma unroll
                        for (int j = 0; j < FJ; ++j) {
                            const int ml = 16 * FJ * wm + 16 * j + li;
                            char* dst = smem + ((nl >> 6) * TM + ml) * 128 + (((c >> 3) ^ (ml & 7)) << 4) + (c & 7) * 2;
                            const f32x4 v = acc[i][j] + bv;
                            const f16x4 u16 = sat4(v[0], v[1], v[2], v[3], dummy);
                            *(uint2*)dst = pack4(dummy, gelu_tanh_f((float)u16[0]), gelu_tanh_f((float)u16[1]), gelu_tanh_f((float)u16[2]), gelu_tanh_f((float)u16[3]));
                        }
                    }
                }
                __syncthreads();
                copy_out(p.out2);
            }
        }
        return;
    }
    if constexpr (EPI == EPI_QKV) {
        // block-uniform: 8-token groups of a V^T row must not straddle attention items
        if (!GTAV_DBG(p, 16) && (p.qkv_mode == QKV_TEMPORAL || p.S % 8 == 0)) {
            qkv_staged<FI, FJ, WM, WN, WOFF, PRE, FOLD>(p, acc, pbias, smem, n0, m0, tr, prope, ft);
            return;
        }
    }
    if constexpr (FOLD) {   // (EPI_F32: the final projection)
        __syncthreads();   // every wave is done reading the last K-step's stage
        fold_tables_to_lds<FJ>(ft, smem, (int)threadIdx.x - 64 * WOFF);
        __syncthreads();
    }
    if (!compute_wave) return;

    if constexpr (EPI == EPI_QKV) {
        if (tr) {
            // D[row = token][col = feature]: lane owns tokens m..m+3 of feature n (V part, spatial mode)
            const int heads = p.D >> 6;
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int n = n0 + 16 * FI * wn + 16 * i + li;
                const int nn = n - 2 * p.D;
                const float bv = p.bias ? p.bias[n] : 0.f;
                const int head = nn >> 6, d = nn & 63;
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    const int m = m0 + 16 * FJ * wm + 16 * j + 4 * g;
                    if (m >= p.M) continue;
                    const int nb = m / p.S, s = m - nb * p.S;
                    f16* dst = p.v + ((size_t)(nb * heads + head) * 64 + d) * p.S + s;
                    const f32x4 a = acc[i][j];
                    *(uint2*)dst = pack4(amax, a[0] + bv, a[1] + bv, a[2] + bv, a[3] + bv);
                }
            }
            sat_report(amax, p.err_flag);
            return;
        }
    }

    // token-side index math once per token column (integer divisions by runtime S / Tq are ~40 instructions each)
    int tok_m[FJ], tok_a[FJ], tok_b[FJ], tok_pos[FJ];
    if constexpr (EPI == EPI_QKV) {
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int m = m0 + 16 * FJ * wm + 16 * j + li;
            tok_m[j] = m;
            if (p.qkv_mode == QKV_SPATIAL) {
                const int nb = m / p.S;
                tok_a[j] = nb;                 // frame
                tok_b[j] = m - nb * p.S;       // token in frame
                tok_pos[j] = tok_b[j];
            } else {
                const int fr = m / p.S;        // frame counter over (b, tl)
                const int b = fr / p.Tq;
                const int tfr = p.t0 + (fr - b * p.Tq);
                tok_a[j] = (b * p.Tmax + tfr) * p.S + (m - fr * p.S);   // kv-cache token slot
                tok_b[j] = 0;
                tok_pos[j] = tfr;
            }
        }
    }
    if constexpr (EPI == EPI_RESID && !FOLD) {
        // Read-modify-write of the fp32 output, software-pipelined over the feature groups: the loads of group i + 1 (output tile, gate vector) are issued BEFORE
        // the stores of group i.  vmcnt retires in order, so a load issued behind a store waits for that store's round trip — the straightforward load / add /
        // store loop paid that FI x FJ times per wave (32 times on the 256 x 256 tile of the grouped weight-gradient launch).
        int grow[FJ];
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            int mm = m0 + 16 * FJ * wm + 16 * j + li;
            mm = mm < p.M ? mm : p.M - 1;
            grow[j] = 0;
            if (p.gate) {
                grow[j] = mm / p.rows_per_gate;
                if (p.gate_rows) grow[j] = p.gate_rows[grow[j]];
            }
        }
        f32x4 xo[2][FJ], gt[2][FJ];
        auto loads = [&](int i, f32x4 (&xv)[FJ], f32x4 (&gv)[FJ]) {
            const int n = n0 + 16 * FI * wn + 16 * i + 4 * g;
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const int m = m0 + 16 * FJ * wm + 16 * j + li;
                xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                gv[j] = f32x4{1.f, 1.f, 1.f, 1.f};
                if (n < p.N && m < p.M) {
                    xv[j] = *(const f32x4*)((const float*)p.out + (size_t)m * p.ldo + n);
                    if (p.gate) gv[j] = *(const f32x4*)(p.gate + (size_t)grow[j] * p.gate_stride + n);
                }
            }
        };
        loads(0, xo[0], gt[0]);
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            if (i + 1 < FI) loads(i + 1, xo[(i + 1) & 1], gt[(i + 1) & 1]);
            const int n = n0 + 16 * FI * wn + 16 * i + 4 * g;
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const int m = m0 + 16 * FJ * wm + 16 * j + li;
                if (n < p.N && m < p.M) {
                    const f32x4 x = xo[i & 1][j] + gt[i & 1][j] * (acc[i][j] + pbias[i]);
                    float* dst = (float*)p.out + (size_t)m * p.ldo + n;
                    if (p.out_sc1) store16_sc1(dst, x);
                    else *(f32x4*)dst = x;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        const int n = n0 + 16 * FI * wn + 16 * i + 4 * g;  // 4 consecutive features n..n+3
        if (n >= p.N) continue;
        const f32x4 bv = pbias[i];
        int which = 0, nn = n, head = 0, d = 0;
        if constexpr (EPI == EPI_QKV) {
            which = n >= 2 * p.D ? 2 : (n >= p.D ? 1 : 0);
            nn = n - which * p.D;
            head = nn >> 6;
            d = nn & 63;
        }
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int m = m0 + 16 * FJ * wm + 16 * j + li;
            if (m >= p.M) continue;
            f32x4 v;
            if constexpr (FOLD) {
                const float* cf = (const float*)ft.cb + fold_frame_of(ft, m0 + 16 * FJ * wm + 16 * j) * 2 * TNB + (n - n0);
                v = fold_apply(acc[i][j], ft.mu[j], ft.rs[j], *(const f32x4*)cf, *(const f32x4*)(cf + TNB));
            } else {
                v = acc[i][j] + bv;
            }
            if constexpr (EPI == EPI_PARTIAL) {
                // (fp16 slabs were tried for the large-M launches — half the slab bytes in the GEMM's tail and in the LayerNorm behind it — and
                // changed nothing: B = 8 kernel time 7.43 ms either way, profiles/round3/fp16_slabs_B8_{on,off}.txt)
                float* dst = (float*)p.out + ((size_t)ks * p.M + m) * p.ldo + n;
                if (p.out_sc1) store16_sc1(dst, v);
                else *(f32x4*)dst = v;
            } else if constexpr (EPI == EPI_F32) {
                *(f32x4*)((float*)p.out + (size_t)m * p.ldo + n) = v;
            } else if constexpr (EPI == EPI_F16) {
                *(uint2*)((f16*)p.out + (size_t)m * p.ldo + n) = pack4(amax, v[0], v[1], v[2], v[3]);
            } else if constexpr (EPI == EPI_QKV) {
                if (which < 2) {
                    // interleaved table: (cos, sin) of pair d/2 and of pair d/2+1 in one 16-byte load
                    const f32x4 cs = *(const f32x4*)(rope_tab(p, n) + tok_pos[j] * 64 + d);
                    v = rope4(v, cs);
                }
                const uint2 pk = pack4(amax, v[0], v[1], v[2], v[3]);
                if (p.qkv_mode == QKV_SPATIAL) {
                    const int heads = p.D >> 6;
                    f16* base = which == 0 ? p.q : p.k;
                    f16* dst = base + ((size_t)(tok_a[j] * heads + head) * p.S + tok_b[j]) * 64 + d;
                    *(uint2*)dst = pk;   // 8-byte sc1 stores are slower (one fabric write each): QKV 64 -> 73 us at M = 5760
                } else {
                    if (which == 0) {
                        *(uint2*)(p.q + (size_t)m * p.D + nn) = pk;
                    } else {
                        f16* base = p.k + (size_t)tok_a[j] * 2 * p.D + (which == 2 ? p.D : 0);
                        *(uint2*)(base + nn) = pk;
                    }
                }
            }
        }
    }
    if constexpr (EPI == EPI_F16 || EPI == EPI_QKV) sat_report(amax, p.err_flag);
}

template <int EPIX, int FI, int FJ, int WM, int WN = 2, int WOFF = 0>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x4 (&acc)[FI][FJ], const f32x4 (&pbias)[FI], char* smem, int n0, int m0, int ks, bool tr) {
    static_assert(!epi_is_fold_consumer(EPIX), "a fold consumer passes its prefetched FoldTok");
    FoldTok<FJ> ft;
    epilogue<EPIX, FI, FJ, WM, WN, WOFF, false>(p, acc, pbias, smem, n0, m0, ks, tr, acc, ft);   // (prope / ft are not read without PRE / FOLD)
}

template <int EPIX, int NS, int WM, int FJ>
__global__ __launch_bounds__(128 * WM, (WM == 2 && NS <= 2) ? 2 : (WM == 4 ? 2 : 1)) void gemm_kernel(GemmParams p) {
    constexpr int EPI = epi_base(EPIX);
    constexpr int TM = WM * 16 * FJ;
    __shared__ __attribute__((aligned(16))) char smem[NS * (1 + TM / 128) * TILE_BYTES];
    static_assert(NS * (1 + TM / 128) * TILE_BYTES >= TM * 256 + TM * 8 + fold_lds_bytes(TN), "the ring must cover the epilogue's LDS image");
    BlockStamps bs;
    bs.begin(p);
    int n0, m0, ks, kt0, nkt;
    tile_map<EPI == EPI_PARTIAL, TN, TM>(p, n0, m0, ks, kt0, nkt);

    f32x4 acc[4][FJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bool tr = false;
    if constexpr (EPI == EPI_QKV) tr = (p.qkv_mode == QKV_SPATIAL) && (n0 >= 2 * p.D);
    f32x4 pbias[4];
    // (residual prefetch inside the K loop only with the counted-wait rings: the two-stage ring waits vmcnt(0) at every K-step, where one more
    // load per step — an HBM miss — made every K-step as long as that miss: out-proj at M = 5760 33.5 -> 37.1 us, fc2 69.8 -> 94.8 us)
    constexpr bool FOLDC = epi_is_fold_consumer(EPIX), FOLDP = EPI == EPI_RESID_FOLD, PREX = FOLDP && NS >= 3;
    FoldTok<FJ> ft;
    f32x4 xpre[PREX ? 4 : 1][PREX ? FJ : 1];
    auto pf = [&]() {
        prefetch_bias<EPI, 4, WM>(p, n0, pbias);
        if constexpr (FOLDC) fold_prefetch<TN, TM, FJ>(p, n0, m0, (int)threadIdx.x, __builtin_amdgcn_readfirstlane(m0 + 16 * FJ * (int)(threadIdx.x >> 7)), true, ft);
    };
    if constexpr (EPI == EPI_QKV) {
        if (tr) mainloop<true, NS, WM, FJ>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
        else mainloop<false, NS, WM, FJ>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
    } else if constexpr (PREX) {
        mainloop<false, NS, WM, FJ>(p, smem, n0, m0, kt0, nkt, acc, bs, pf, [&](int t) { prefetch_resid<4, FJ, WM>(p, n0, m0, xpre, t, t + 1); });
        prefetch_resid<4, FJ, WM>(p, n0, m0, xpre, nkt, 4 * FJ);      // short K: the tiles the loop did not reach
    } else {
        mainloop<false, NS, WM, FJ>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
    }
    GTAV_STAMP(bs.t[2]);
    if constexpr (PREX) epilogue<EPIX, 4, FJ, WM, 2, 0, true>(p, acc, pbias, smem, n0, m0, ks, tr, xpre, ft);
    else if constexpr (FOLDC || FOLDP) epilogue<EPIX, 4, FJ, WM, 2, 0, false>(p, acc, pbias, smem, n0, m0, ks, tr, acc, ft);
    else epilogue<EPIX, 4, FJ, WM>(p, acc, pbias, smem, n0, m0, ks, tr);
    bs.end(p);
}

// ---------------------------------------------------------------------------------------------------------------------
// "TN" GEMM for weight gradients:  out[m][n] += sum_t Xop[t][m] * Wop[t][n]  — both operands are the ordinary tile-major activations
// [tokens][features]; the contraction runs over their ROWS.  (The NT kernels above need K contiguous, so dW = dY^T X used to transpose both
// operands first: 260 transposes = 4.5 ms of a 63 ms training step.)  128 x 128 output tile, 8 waves of 64 x 32, K-step = 64 tokens:
//   * a stage is the 64-token slab of the two 64-feature column tiles of each operand: four contiguous 8 KiB runs of the tile-major image
//     (8 pieces of 8 rows each), copied verbatim by LDS-DMA like every other fill — the LDS image is [64 tokens][64 features] per column
//     tile with the 16-byte chunks XOR-swizzled by (row & 7);
//   * an MFMA fragment needs 8 consecutive TOKENS of one feature per lane — a column of that image: two transposing LDS reads
//     (ds_read_b64_tr_b16: lane 4 q + p of a 16-lane group passes the address of token row q, features 4 p .. 4 p + 3, and receives its own
//     feature's four tokens) per fragment instead of one ds_read_b128; same bytes through the LDS, twice the read instructions.
// p.X = Xop (its features are the output rows m: the "token" role of the epilogue), p.W = Wop (features = output columns n), p.K = tokens.
// ---------------------------------------------------------------------------------------------------------------------
// The transposing reads are inline asm: behind the builtin hipcc's wait-count pass assumes they may alias the LDS-DMA fills in flight and puts
// an s_waitcnt vmcnt(0) in front of the first read of every K-step — the ring's prefetch was gone and the K-step cost a whole fill round trip
// (1.05 us against 0.6 us).  With asm the LDS waits are counted by hand: LDS returns in order, so lgkmcnt(N) retires all but the newest N reads.
typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_read_tr_asm(u32x2_& dst, unsigned lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(lds_addr));
}
union TnFrag { f16x8 h; u32x2_ u[2]; };
template <int NS>
__global__ __launch_bounds__(512, 1) void gemm_tn_kernel(GemmParams p) {
    // stage: 32 pieces of 8 token rows x 64 features [W ct0 | W ct1 | X ct0 | X ct1], each piece at a stride of 1088 bytes: the 64 bytes of
    // padding shift consecutive pieces by 16 banks, so the two 16-lane groups of a transposing read's 32-lane half (token rows 8 apart =
    // adjacent pieces, same swizzle) do not meet on the same banks (with a 1024-byte stride every such read was 2-way conflicted and the
    // K-step LDS-bound: 195 us per 4096 x 1024 x 11520 launch against 110 us for transposes + the NT kernel)
    constexpr int WM = 4, FJ = 2, HALF = 8192, PSTR = 1088, STAGE_BYTES = 32 * PSTR, G = 4;
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE_BYTES];
    BlockStamps bs;
    bs.begin(p);
    int n0, m0, ks, kt0, nkt;
    tile_map<false, 128, 128>(p, n0, m0, ks, kt0, nkt);   // nkt = p.K / 64 token steps
    f32x4 acc[4][FJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1, wm = w >> 1;
    // ---- fills: wave w copies pieces 4 w .. 4 w + 3 of the stage's 32 (piece q: operand q / 16, column tile (q / 8) & 1, 8-row piece q & 7) ----
    const int nct_w = p.N >> 6, nct_x = p.M >> 6;        // 64-feature column tiles per row tile of each operand
    const char* src[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int q = 4 * w + i, isx = q >> 4, ct = (q >> 3) & 1, pc = q & 7;
        const char* base = (const char*)(isx ? (const void*)p.X : (const void*)p.W);
        const int ct0 = ((isx ? m0 : n0) >> 6) + ct, nct = isx ? nct_x : nct_w;
        src[i] = base + (size_t)ct0 * TILE_BYTES + pc * 1024 + lane * 16 + (size_t)0 * nct;
    }
    auto stage = [&](int t) {   // token step t: rows 64 t .. 64 t + 63 = row tile t / 2, pieces 8 (t & 1) ..
        char* base = smem + (t % NS) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int q = 4 * w + i, isx = q >> 4;
            const size_t rowtile = (size_t)(t >> 1) * (isx ? nct_x : nct_w) * TILE_BYTES + (size_t)(t & 1) * HALF;
            glds16(src[i] + rowtile, base + q * PSTR);
        }
    };
    // ---- transposing fragment reads: lane (li = 4 q + pp, group g): token row 32 s + 8 g + 4 h + q, features 16 tile + 4 pp .. ----
    const int li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
    int woff[2][2][4], xoff[2][2][FJ];                   // [half-step s][h][tile]
#pragma unroll
    for (int sh = 0; sh < 2; ++sh)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int row = 32 * sh + 8 * g + 4 * hh + qq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = 2 * i + (pp >> 1);        // 16-byte chunk of the 64-feature row: features 16 i + 4 pp ..
                woff[sh][hh][i] = (wn * 8 + (row >> 3)) * PSTR + (row & 7) * 128 + ((ch ^ (row & 7)) << 4) + (pp & 1) * 8;
            }
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const int f = 32 * (wm & 1) + 16 * j + 4 * pp, ch = f >> 3;
                xoff[sh][hh][j] = ((2 + (wm >> 1)) * 8 + (row >> 3)) * PSTR + (row & 7) * 128 + ((ch ^ (row & 7)) << 4) + (pp & 1) * 8;
            }
        }
    const bool late = w >= 4;
    const int npro = nkt < NS - 1 ? nkt : NS - 1;
    for (int t = 0; t < npro; ++t) stage(t);
    for (int t = 0; t < nkt; ++t) {
        const int rem = nkt - 1 - t;
        kstep_sync_ring<NS, G>(rem);
        if (t == 0) GTAV_STAMP(bs.t[1]);
        const bool refill = t + NS - 1 < nkt;
        if (refill && !late) stage(t + NS - 1);
        const unsigned b = lds_offset(smem) + (unsigned)(t % NS) * STAGE_BYTES;
        TnFrag wf[2][4], xf[2][FJ];
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {   // 12 reads per half-step, all 24 issued up front
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lds_read_tr_asm(wf[sh][i].u[0], b + woff[sh][0][i]);
                lds_read_tr_asm(wf[sh][i].u[1], b + woff[sh][1][i]);
            }
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                lds_read_tr_asm(xf[sh][j].u[0], b + xoff[sh][0][j]);
                lds_read_tr_asm(xf[sh][j].u[1], b + xoff[sh][1][j]);
            }
        }
        // the waits name the fragments they retire as in / out operands, so no MFMA that reads them can be scheduled above its wait
#define GTAV_TN_FRAGS(sh) "+v"(wf[sh][0].u[0]), "+v"(wf[sh][0].u[1]), "+v"(wf[sh][1].u[0]), "+v"(wf[sh][1].u[1]), "+v"(wf[sh][2].u[0]), "+v"(wf[sh][2].u[1]), \
                          "+v"(wf[sh][3].u[0]), "+v"(wf[sh][3].u[1]), "+v"(xf[sh][0].u[0]), "+v"(xf[sh][0].u[1]), "+v"(xf[sh][1].u[0]), "+v"(xf[sh][1].u[1])
        static_assert(FJ == 2, "the wait statements list the fragments of FJ == 2");
        asm volatile("s_waitcnt lgkmcnt(12)" : GTAV_TN_FRAGS(0));   // LDS returns in order: the first half-step's 12 reads have landed
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[i][j] = mfma16(wf[0][i].h, xf[0][j].h, acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : GTAV_TN_FRAGS(1));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[i][j] = mfma16(wf[1][i].h, xf[1][j].h, acc[i][j], 0, 0, 0);
#undef GTAV_TN_FRAGS
        if (refill && late) {
            asm volatile("" ::: "memory");
            stage(t + NS - 1);
        }
    }
    GTAV_STAMP(bs.t[2]);
    f32x4 pbias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pbias[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    epilogue<EPI_RESID, 4, FJ, WM>(p, acc, pbias, smem, n0, m0, ks, false);
    bs.end(p);
}

template <int EPI, bool PP = false>   // PP: the antiphase main loop (mainloop256_pp, block shape 17)
__global__ __launch_bounds__(512, 1) void gemm256_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[8 * TILE_BYTES + (EPI == EPI_QKV ? 2048 : 0)];   // + qkv_staged's token table
    BlockStamps bs;
    bs.begin(p);
    int n0, m0, ks, kt0, nkt;
    tile_map<EPI == EPI_PARTIAL, 256, 256>(p, n0, m0, ks, kt0, nkt);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool tr = false;
    f32x4 pbias[8];
    auto pf = [&]() { prefetch_bias<EPI, 8, 4>(p, n0, pbias); };
    auto nopf = []() {};
    if constexpr (EPI == EPI_QKV) {
        tr = (p.qkv_mode == QKV_SPATIAL) && (n0 >= 2 * p.D);
        if constexpr (PP) {
            if (tr) mainloop256_pp<true>(p, smem, n0, m0, kt0, nkt, acc, bs, nopf);
            else mainloop256_pp<false>(p, smem, n0, m0, kt0, nkt, acc, bs, nopf);
        } else {
            if (tr) mainloop256<true>(p, smem, n0, m0, kt0, nkt, acc, bs, nopf);
            else mainloop256<false>(p, smem, n0, m0, kt0, nkt, acc, bs, nopf);
        }
        pf();   // the QKV variant is at the 256-register limit: fetch after the main loop
    } else {
        if constexpr (PP) mainloop256_pp<false>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
        else mainloop256<false>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
    }
    GTAV_STAMP(bs.t[2]);
    epilogue<EPI, 8, 4, 4>(p, acc, pbias, smem, n0, m0, ks, tr);
    bs.end(p);
}

// mainloop256 for operands that are tile-major [tokens][features] with the contraction over the TOKENS (weight gradients: out[m][n] += sum_t X[t][m] W[t][n];
// p.M = X features, p.N = W features, p.K = tokens; nct_x / nct_w = 64-feature column tiles per row tile of X / W) — no operand transposes.  Same four phases,
// same fill schedule and barriers as mainloop256; what changes is the stage image and the fragment reads:
//   * a K-tile is 64 tokens of 256 + 256 features: 64 pieces of [8 tokens][64 features] (1 KiB, copied verbatim: half of a source tile's column), pieces
//     0-31 = W column tiles 0-3, 32-63 = X column tiles 0-3, at a stride of 1088 bytes (gemm_tn_kernel: with 1024 the two 16-lane groups of a transposing
//     read's 32-lane half meet on the same banks); the four fill regions of mainloop256 are column-tile pairs (W 0-1, W 2-3, X 0-1, X 2-3), wave w copies
//     two adjacent token pieces of column tile w / 4 of each;
//   * a fragment (16 features x 32 tokens) is two ds_read_b64_tr_b16: lane (16 g + 4 q + pp) reads token row 32 sh + 8 g + 4 h + q, features 16 i + 4 pp ..,
//     h = 0 / 1 (TnFrag, as gemm_tn_kernel).  Eight per-lane offsets (h, 16-byte chunk pair) serve every fragment of both operands: column tile and
//     K half enter as immediates, the operand's piece base as one add.  The reads are inline asm (behind the builtin hipcc waits vmcnt(0) before every
//     first read), so the LDS waits are counted by hand and name the fragments they retire.
__device__ __forceinline__ void mainloop256_tn(const GemmParams& p, char* smem, int n0, int m0, int nkt, int nct_w, int nct_x, f32x4 (&acc)[8][4], BlockStamps& bs) {
    constexpr int PSTR = 1088, PAR = 64 * PSTR, HALF = 8192;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1, wm = w >> 1;
    // fills: region h (0, 1: W column tiles 2 h, 2 h + 1; 2, 3: X column tiles 2 (h - 2), + 1), wave w -> column tile w / 4 of the region, token pieces 2 w & 7, + 1
    const size_t po = (size_t)((2 * w) & 7) * 1024 + lane * 16;
    const char* const sw0 = (const char*)p.W + (size_t)((n0 >> 6) + (w >> 2)) * TILE_BYTES + po;
    const char* const sw1 = sw0 + (size_t)2 * TILE_BYTES;
    const char* const sx0 = (const char*)p.X + (size_t)((m0 >> 6) + (w >> 2)) * TILE_BYTES + po;
    const char* const sx1 = sx0 + (size_t)2 * TILE_BYTES;
    const size_t rtw = (size_t)nct_w * TILE_BYTES, rtx = (size_t)nct_x * TILE_BYTES;   // bytes per 128-token row tile
    char* const dst0 = smem + (2 * w) * PSTR + lane * 16;
    auto stage = [&](const char* src, size_t rt_bytes, int h, int t) {
        const char* s = src + (size_t)(t >> 1) * rt_bytes + (size_t)(t & 1) * HALF;
        char* d = dst0 + (t & 1) * PAR + h * 16 * PSTR;
        glds16(s, d);
        glds16(s + 1024, d + PSTR);
    };
    const int li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
    unsigned aw[2][4], ax[2][4];   // [h][chunk pair c]: W / X fragment (column tile 0 of the wave, K half 0) of tile c
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int r7 = 4 * hh + qq;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned o = (unsigned)(g * PSTR + r7 * 128 + (((2 * c + (pp >> 1)) ^ r7) << 4) + (pp & 1) * 8);
            aw[hh][c] = o + (unsigned)(2 * wn * 8 * PSTR);
            ax[hh][c] = o + (unsigned)((32 + wm * 8) * PSTR);
        }
    }
    // W tile i (16 features) of the wave: column tile 2 wn + i / 4 -> + (i / 4) * 8 pieces; K half sh: + 4 sh pieces
    auto rdw = [&](TnFrag& f, unsigned b, int i, int sh) {
        lds_read_tr_asm(f.u[0], b + aw[0][i & 3] + (unsigned)(((i >> 2) * 8 + 4 * sh) * PSTR));
        lds_read_tr_asm(f.u[1], b + aw[1][i & 3] + (unsigned)(((i >> 2) * 8 + 4 * sh) * PSTR));
    };
    auto rdx = [&](TnFrag& f, unsigned b, int j, int sh) {
        lds_read_tr_asm(f.u[0], b + ax[0][j] + (unsigned)(4 * sh * PSTR));
        lds_read_tr_asm(f.u[1], b + ax[1][j] + (unsigned)(4 * sh * PSTR));
    };
    auto mma = [&](const TnFrag& wv, const TnFrag& xv, f32x4& c) { c = mfma16(wv.h, xv.h, c, 0, 0, 0); };

    stage(sx0, rtx, 2, 0); stage(sx1, rtx, 3, 0); stage(sw0, rtw, 0, 0); stage(sw1, rtw, 1, 0);
    if (nkt > 1) {
        stage(sx0, rtx, 2, 1); stage(sx1, rtx, 3, 1);
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    GTAV_STAMP(bs.t[1]);
    const unsigned smem0 = lds_offset(smem);
#define GTAV_TNF(f) "+v"((f).u[0]), "+v"((f).u[1])
    for (int t = 0; t < nkt; ++t) {
        const unsigned b = smem0 + (unsigned)(t & 1) * PAR;
        const bool n1 = t + 1 < nkt, n2 = t + 2 < nkt;
        TnFrag wa[2][4], wb[2][4], xa[2][2], xb[2][2];
        // (lgkmcnt is a 4-bit counter: at most 15 reads may be outstanding — every group below is 8 or 12 reads, issued in front of the MFMAs of the group before)
        // ---- phase 1: W[0:4] x X[0:2] ----
        if (n1) stage(sw0, rtw, 0, t + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) rdw(wa[0][i], b, i, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) rdx(xa[0][j], b, j, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : GTAV_TNF(wa[0][0]), GTAV_TNF(wa[0][1]), GTAV_TNF(wa[0][2]), GTAV_TNF(wa[0][3]), GTAV_TNF(xa[0][0]), GTAV_TNF(xa[0][1]));
#pragma unroll
        for (int i = 0; i < 4; ++i) rdw(wa[1][i], b, i, 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) rdx(xa[1][j], b, j, 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma(wa[0][i], xa[0][j], acc[i][j]);
        asm volatile("s_waitcnt lgkmcnt(0)" : GTAV_TNF(wa[1][0]), GTAV_TNF(wa[1][1]), GTAV_TNF(wa[1][2]), GTAV_TNF(wa[1][3]), GTAV_TNF(xa[1][0]), GTAV_TNF(xa[1][1]));
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
#pragma unroll
            for (int j = 0; j < 2; ++j) rdx(xb[sh][j], b, 2 + j, sh);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma(wa[1][i], xa[1][j], acc[i][j]);
        // ---- phase 2: W[0:4] x X[2:4]; the fragments of W[4:8] for phases 3 / 4 ----
        if (n1) stage(sw1, rtw, 1, t + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" : GTAV_TNF(xb[0][0]), GTAV_TNF(xb[0][1]), GTAV_TNF(xb[1][0]), GTAV_TNF(xb[1][1]));
#pragma unroll
        for (int i = 0; i < 4; ++i) rdw(wb[0][i], b, 4 + i, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma(wa[0][i], xb[0][j], acc[i][2 + j]);
        asm volatile("s_waitcnt lgkmcnt(0)" : GTAV_TNF(wb[0][0]), GTAV_TNF(wb[0][1]), GTAV_TNF(wb[0][2]), GTAV_TNF(wb[0][3]));
#pragma unroll
        for (int i = 0; i < 4; ++i) rdw(wb[1][i], b, 4 + i, 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma(wa[1][i], xb[1][j], acc[i][2 + j]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : GTAV_TNF(wb[1][0]), GTAV_TNF(wb[1][1]), GTAV_TNF(wb[1][2]), GTAV_TNF(wb[1][3]) : : "memory");   // (b)
        // ---- phase 3 ----
        if (n2) stage(sx0, rtx, 2, t + 2);
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma(wb[sh][i], xb[sh][j], acc[4 + i][2 + j]);
        // ---- phase 4 ----
        if (n2) stage(sx1, rtx, 3, t + 2);
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma(wb[sh][i], xa[sh][j], acc[4 + i][j]);
        if (n2) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");   // (a)
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
#undef GTAV_TNF
}

// Grouped weight-gradient launch (launch_gemm_dw_grouped; training): the four dW GEMMs of a half-block — outputs
// 4096 x 1024, 1024 x 4096, 3072 x 1024 and 1024 x 1024, contraction over the 11 520 tokens of a batch-16 step — as ONE grid of 256 x 256
// tiles (64 + 64 + 48 + 16 = 192 tiles: one per CU, one round).  Alone each of them fills the chip only with 128 x 128 tiles, whose
// L2 -> LDS fill per FLOP is twice that of the 256 x 256 tile, and the long K loop is bound by exactly that fill (tools/dw_bench.py:
// 137 + 119 + 92 + 56 us one after the other against ~250 us for ANY number of 256 x 256 tiles up to one per CU).  Blocks of one XCD take a
// contiguous run of the (group, m-tile, n-tile) order, n fastest, so they share the rows of X and the W panels of their group in that
// XCD's L2.  Epilogue: out += acc (EPI_RESID without gate or bias), every output tile owned by exactly one block.
struct GemmDwGroups {
    GemmDwGroup g[GEMM_DW_MAX_GROUPS];
    int first[GEMM_DW_MAX_GROUPS + 1];   // first[i] = tiles of groups 0 .. i-1
    int n;
};
// TN = true: the groups' operands are the activations themselves, tile-major [tokens][features] (X: M features, W: N features, K tokens): mainloop256_tn
template <bool TN>
__global__ __launch_bounds__(512, 1) void gemm256_dw_grouped_kernel(GemmParams p0, GemmDwGroups gs) {
    __shared__ __attribute__((aligned(16))) char smem[TN ? 2 * 64 * 1088 : 8 * TILE_BYTES];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int swz = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    int gi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_DW_MAX_GROUPS; ++i)
        if (i < gs.n && swz >= gs.first[i]) gi = i;
    GemmParams p = p0;
    p.X = gs.g[gi].X; p.W = gs.g[gi].W; p.out = gs.g[gi].out; p.M = gs.g[gi].M; p.N = gs.g[gi].N; p.ldo = gs.g[gi].ldo;
    const int t = swz - gs.first[gi], tiles_n = p.N >> 8;
    const int tile_m = t / tiles_n, tile_n = t - tile_m * tiles_n;
    const int n0 = tile_n * 256, m0 = tile_m * 256;
    BlockStamps bs;
    bs.begin(p);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 pbias[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pbias[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (TN) mainloop256_tn(p, smem, n0, m0, p.K / TK, p.N >> 6, p.M >> 6, acc, bs);
    else mainloop256<false>(p, smem, n0, m0, 0, p.K / TK, acc, bs, []() {});
    GTAV_STAMP(bs.t[2]);
    epilogue<EPI_RESID, 8, 4, 4>(p, acc, pbias, smem, n0, m0, 0, false);
    bs.end(p);
}

// Experiment of round 2, NOT kept — de-phasing the two co-resident blocks of a CU (shape 12, large M).  The per-block timeline
// (tools/gemm_stamps.py, profiles/round2/stamps_lockstep.txt) shows the 512 blocks of a residency round in lockstep: prologue 2 us,
// main loop 21.6 us, epilogue 6.1 us, all at the same time on every CU.  Making the second block to arrive on a CU (a ticket from a
// per-CU counter indexed by XCC id / HW_ID) wait 3-16 us did shift its phases (stamps_dephased.txt) but bought nothing: fc1 at
// M = 5760 59.4 us without, 60.5-65.4 us with; M = 11 520 115.7 vs 113.1-115.7 (profiles/round2/dephase_sweep.txt).  A block does not
// run faster while its partner idles — its K-step is bound by its own barrier-synchronised fill / read / MFMA chain, not by a fill
// rate shared with the partner — so an offset only moves the idle time.  The same measurement explains the ping-pong kernel below.
template <int EPIX, int NS, int FI, int FJ, int WM>
// (second launch-bound = waves per SIMD: the 8-wave large-M tile runs two blocks per CU = 4 waves per SIMD = 128 registers)
__global__ __launch_bounds__(128 * WM, (WM == 4 && NS == 2 && FI <= 4) ? 4 : (WM == 2 && FI <= 4) ? 2 : 1) void gemm_g_kernel(GemmParams p) {
    constexpr int EPI = epi_base(EPIX);
    constexpr int TNB = 32 * FI, TM = 16 * FJ * WM;
    constexpr int STAGE = (4 * FI + 2 * FJ * WM) * 1024;
    // ring, or the QKV epilogue's pitched LDS image + token table (qkv_staged), whichever is larger
    constexpr int PNB = (TNB / 8 + 7) / 8 * 8 * 16, PTB = (TM / 8 + 7) / 8 * 8 * 16;
    constexpr int EPIB = (TM * PNB > TNB * PTB ? TM * PNB : TNB * PTB) + TM * 8 + (epi_is_fold_consumer(EPIX) ? fold_lds_bytes(TNB) : 0);
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE > EPIB ? NS * STAGE : EPIB];
    BlockStamps bs;
    bs.begin(p);
    int n0, m0, ks, kt0, nkt;
    tile_map<EPI == EPI_PARTIAL, TNB, TM>(p, n0, m0, ks, kt0, nkt);
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool tr = false;
    f32x4 pbias[FI];
    constexpr bool FOLDC = epi_is_fold_consumer(EPIX), FOLDP = EPI == EPI_RESID_FOLD, PREX = FOLDP && NS >= 3;   // (see gemm_kernel)
    FoldTok<FJ> ft;
    f32x4 xpre[PREX ? FI : 1][PREX ? FJ : 1];
    auto pf = [&]() {
        prefetch_bias<EPI, FI, WM>(p, n0, pbias);
        if constexpr (FOLDC) fold_prefetch<TNB, TM, FJ>(p, n0, m0, (int)threadIdx.x, __builtin_amdgcn_readfirstlane(m0 + 16 * FJ * (int)(threadIdx.x >> 7)), true, ft);
    };
    if constexpr (EPI == EPI_QKV) {
        tr = (p.qkv_mode == QKV_SPATIAL) && (n0 >= 2 * p.D);
        if (tr) mainloop_g<true, NS, FI, FJ, WM>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
        else mainloop_g<false, NS, FI, FJ, WM>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
    } else if constexpr (PREX) {
        mainloop_g<false, NS, FI, FJ, WM>(p, smem, n0, m0, kt0, nkt, acc, bs, pf, [&](int t) { prefetch_resid<FI, FJ, WM>(p, n0, m0, xpre, t, t + 1); });
        prefetch_resid<FI, FJ, WM>(p, n0, m0, xpre, nkt, FI * FJ);
    } else {
        mainloop_g<false, NS, FI, FJ, WM>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
    }
    GTAV_STAMP(bs.t[2]);
    if constexpr (PREX) epilogue<EPIX, FI, FJ, WM, 2, 0, true>(p, acc, pbias, smem, n0, m0, ks, tr, xpre, ft);
    else if constexpr (FOLDC || FOLDP) epilogue<EPIX, FI, FJ, WM, 2, 0, false>(p, acc, pbias, smem, n0, m0, ks, tr, acc, ft);
    else epilogue<EPIX, FI, FJ, WM>(p, acc, pbias, smem, n0, m0, ks, tr);
    bs.end(p);
}

// Grouped launch (launch_gemm_grouped): blockIdx.y selects one of several independent GEMMs that share M and K — the c1 / c2 tables of
// the LayerNorm fold, 2 x 65 products of the per-frame (1 + scale) and shift vectors with every to_qkv / fc1 / final weight, in ONE launch
// per forward (or per generated frame in the sampler).  128 x 96 tiles on the piece-granular main loop, plain f32 epilogue; a group with
// fewer tiles than gridDim.x lets the surplus blocks exit.
template <int NS, int FI, int FJ, int WM>
__global__ __launch_bounds__(128 * WM, 1) void gemm_grouped_kernel(GemmParams p0, const GemmGroup* __restrict__ groups) {
    constexpr int TNB = 32 * FI, TM = 16 * FJ * WM;
    constexpr int STAGE = (4 * FI + 2 * FJ * WM) * 1024;
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
    GemmParams p = p0;
    {
        const GemmGroup* gr = groups + blockIdx.y;
        p.X = gr->X; p.W = gr->W; p.out = gr->out; p.bias = gr->bias; p.N = gr->N; p.ldo = gr->ldo;
    }
    const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TNB - 1) / TNB, tiles = tiles_m * tiles_n;
    // XCD-aware order over this group's own tile count, n fastest (co-resident blocks share the X row tile)
    const int bid = blockIdx.x;
    if (bid >= tiles) return;
    const int xcd = bid & 7, qq = tiles >> 3, rr = tiles & 7;
    const bool inrange = (bid >> 3) < qq + (xcd < rr ? 1 : 0);   // (always true for bid < tiles; keeps the map a bijection on [0, tiles))
    const int tile_id = inrange ? (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3) : bid;
    const int tile_m = tile_id / tiles_n, tile_n = tile_id - tile_m * tiles_n;
    const int n0 = tile_n * TNB, m0 = tile_m * TM;
    BlockStamps bs;
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 pbias[FI];
    auto pf = [&]() { prefetch_bias<EPI_F32, FI, WM>(p, n0, pbias); };
    mainloop_g<false, NS, FI, FJ, WM>(p, smem, n0, m0, 0, p.K / TK, acc, bs, pf);
    epilogue<EPI_F32, FI, FJ, WM>(p, acc, pbias, smem, n0, m0, 0, false);
}

// ---------------------------------------------------------------------------------------------------------------------
// Loader-wave GEMM (shapes 20+): the fills are issued by DEDICATED loader waves, the compute waves never touch global memory in
// the K loop.
//
// Why (round-2 measurements, profiles/round2/ingest_bw.txt, gemm_dbg_r2b.txt): a CU takes L2-resident bytes in at ~64 B/clk
// (124-134 GB/s) whether by LDS-DMA or by register loads — the two paths share that limit and do not add — and a wave that issues
// an LDS-DMA instruction is held until the texture-address unit has accepted it.  In the all-waves-fill kernels above every wave
// issues its share of the next tile right after the K-step barrier, so all waves sit in the address queue together (28 KB at
// 64 B/clk = 450 cycles at M = 720) and only then start their 256-512 cycles of MFMAs: the two phases ADD (fc1 at M = 720: 10.2 us
// with the MFMAs skipped, 13.7 us with them; main loop 8.5 us where either phase alone needs 3.5).  Here NL loader waves (one per
// SIMD) own the fill stream and block in that queue on their own; WN x WM compute waves (two per SIMD, balanced) run
// barrier -> fragment reads of tile t -> MFMAs of tile t - 1 (the one-step software pipeline of mainloop_g) and the K-step costs
// max(ingest, MFMA) instead of their sum.  All waves meet at ONE s_barrier per K-step:
//   loader  t: s_waitcnt vmcnt: its own pieces of tile t have landed            | barrier t | issue tile t + NS - 1 into slot (t - 1) % NS
//   compute t: s_waitcnt lgkmcnt(0): its fragment reads of tile t - 1 are done  | barrier t | read tile t, MFMA tile t - 1
// so after barrier t tile t is complete and visible and nobody reads tile t - 1 any more.  Loader waves take part in the epilogue's
// barriers and in its copy-out loops (they have nothing else to do).
// ---------------------------------------------------------------------------------------------------------------------
// TRI: feature groups i >= TRI of every wave are accumulated TRANSPOSED (token rows, feature columns — the V^T part of the fused spatial to_qkv + attention kernel)
template <bool TR, int NS, int FI, int FJ, int WN, int WM, int NL, int TRI = FI, bool HALFSTEP = (FI * FJ >= 16), typename AfterPrologue>   // HALFSTEP: the half-K-step pipeline (one fragment set per half) also for smaller wave tiles that are short of registers
__device__ __forceinline__ void mainloop_l(const GemmParams& p, char* smem, int n0, int m0, int kt0, int nkt,
                                           f32x4 (&acc)[FI][FJ], BlockStamps& bs, AfterPrologue after_prologue) {
    static_assert(NS >= 3 && NL >= 1, "loader-wave ring: at least 3 stages and one loader wave");
    constexpr int NCW = WN * WM;
    constexpr int WPC = 2 * FI * WN, XPC = 2 * FJ * WM, NP = WPC + XPC;   // 1-KiB pieces per stage
    constexpr int G = (NP + NL - 1) / NL;
    constexpr int STAGE_BYTES = NP * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wraw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nktot = p.K / TK;
    if (wraw < NL) {
        // ------------------------------------------------ loader wave ------------------------------------------------
        // The loader waves are the FIRST waves of the workgroup: a CU starts the waves of a workgroup one after the other (the
        // twelfth wave of these blocks issued its first instruction 0.8-1.0 us after the first one, profiles/round2 stamps), and
        // nothing can be computed before tile 0 is in LDS.  At this scale instructions count: a wave issues ~250 instructions in
        // 0.5 us, so every piece address is a wave-uniform SGPR pair (operand base + tile offsets, scalar ALU) plus ONE constant
        // per-lane offset (lane * 16) — the scalar-base form of the direct-to-LDS load — instead of a 64-bit per-lane pointer
        // recomputed per piece.
        const int lw = wraw;
        const int last_wt = ((p.N + 127) >> 7) - 1, last_rt = (p.M - 1) >> 7;
        const unsigned voff = (unsigned)lane * 16u;
        const unsigned smem0 = lds_offset(smem);
        const char* sb[G];
        unsigned ldso[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            int q = lw * G + i;
            q = q < NP ? q : NP - 1;                         // the last wave repeats the final piece (same bytes, same place)
            const bool isw = q < WPC;
            const int row = isw ? n0 + 8 * q : m0 + 8 * (q - WPC);
            int rt = row >> 7;
            const int lim = isw ? last_wt : last_rt;
            rt = rt < lim ? rt : lim;                        // ragged edges re-read a valid tile (results are masked)
            sb[i] = (const char*)(isw ? p.W : p.X) + ((size_t)rt * nktot + kt0) * TILE_BYTES + ((row & 127) >> 3) * 1024;
            ldso[i] = smem0 + q * 1024;
        }
        auto stage = [&](int t) {
            const unsigned so = (unsigned)(t % NS) * STAGE_BYTES;
            const size_t go = (size_t)t * TILE_BYTES;
#pragma unroll
            for (int i = 0; i < G; ++i) glds16_s(sb[i] + go, voff, ldso[i] + so);
        };
        const int npro = nkt < NS - 1 ? nkt : NS - 1;
        GTAV_STAMP_SLOT(lw == 0 && lane == 0, 7);   // first loader: first fill issued now
        for (int t = 0; t < npro; ++t) stage(t);
        for (int t = 0; t < nkt; ++t) {
            wait_vm_ring<NS, G>(nkt - 1 - t);                // this wave's share of tile t has landed
            GTAV_STAMP_SLOT_HI(t == 0 && lw == 0 && lane == 0, 6);   // its tile-0 share landed
            wg_barrier();
            if (t == 0) GTAV_STAMP(bs.t[1]);
            GTAV_STAMP_SLOT(GTAV_DBG(p, 32) && lw == 0 && lane == 0 && t < 24, 8 + t);   // detailed mode (debug bit 5: 64 slots per block): loader 0 past barrier t
            if (t + NS - 1 < nkt && !GTAV_DBG(p, 1)) stage(t + NS - 1);
        }
        return;
    }
    // ------------------------------------------------ compute wave ------------------------------------------------
    const int w = wraw - NL;
    const int wn = w % WN, wm = w / WN;
    const int li = lane & 15, g = lane >> 4;
    int woff[2], xoff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ch = ((4 * s + g) ^ (li & 7)) << 4;
        woff[s] = (16 * FI * wn + li) * 128 + ch;
        xoff[s] = WPC * 1024 + (16 * FJ * wm + li) * 128 + ch;
    }
    // (issued a few K-steps into the loop instead, past the launch's own first fills, the prefetch made the step 4 % SLOWER than no prefetch:
    // profiles/round3/sampler_ab_window_prefetch_issued_mid_loop.txt)
    l2_prefetch_next(p, w, NCW, lane, bs.pf_sink);   // (before the epilogue's own early loads: older, so the epilogue's first wait covers them)
    after_prologue();   // register loads the epilogue wants early (bias)
    auto rd = [&](int t, f16x8 (&wf)[2][FI], f16x8 (&xf)[2][FJ]) {
        const char* b = smem + (t % NS) * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < FI; ++i) wf[s][i] = *(const f16x8*)(b + woff[s] + i * 16 * 128);
#pragma unroll
            for (int j = 0; j < FJ; ++j) xf[s][j] = *(const f16x8*)(b + xoff[s] + j * 16 * 128);
        }
    };
    auto mm = [&](const f16x8 (&wf)[2][FI], const f16x8 (&xf)[2][FJ]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    if (TR || i >= TRI) acc[i][j] = mfma16(xf[s][j], wf[s][i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = mfma16(wf[s][i], xf[s][j], acc[i][j], 0, 0, 0);
                }
    };
    if constexpr (HALFSTEP) {
        // Large wave tiles (64 x 64: 16 MFMAs per 32-deep half of a K-step): pipeline by HALF K-steps with one register set per half —
        //   barrier t | read (t, half 0) -> A | MFMA (t - 1, half 1) from B | read (t, half 1) -> B | MFMA (t, half 0) from A
        // every batch of reads is in flight under 16 MFMAs (256 cycles), the fragments cost (FI + FJ) x 8 registers instead of the
        // x 16 of two whole-K-step sets (the 256 x 128 tile then fits the 168 registers three waves per SIMD leave).
        f16x8 wa[FI], xa[FJ], wb[FI], xb[FJ];
        auto rdh = [&](int t, int sh, f16x8 (&wf)[FI], f16x8 (&xf)[FJ]) {
            const char* b = smem + (t % NS) * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < FI; ++i) wf[i] = *(const f16x8*)(b + woff[sh] + i * 16 * 128);
#pragma unroll
            for (int j = 0; j < FJ; ++j) xf[j] = *(const f16x8*)(b + xoff[sh] + j * 16 * 128);
        };
        auto mmh = [&](const f16x8 (&wf)[FI], const f16x8 (&xf)[FJ]) {
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    if (TR || i >= TRI) acc[i][j] = mfma16(xf[j], wf[i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = mfma16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                }
        };
        const bool domm = !GTAV_DBG(p, 2);
        for (int t = 0; t < nkt; ++t) {
            wait_lgkm0();
            wg_barrier();
            if (t == 0) GTAV_STAMP(bs.t[1]);
            // the FI + FJ fragment reads of one half are issued BETWEEN the FI x FJ MFMAs of the other half (one read per MFMA, in
            // the MFMA's issue shadow) instead of as a block in front of them: a block of 7-8 ds_read_b128 holds the wave's issue
            // for ~60 cycles in which the matrix pipe drains, twice per K-step
            if (t > 0) {
                rdh(t, 0, wa, xa);
                if (domm) mmh(wb, xb);
                interleave_mfma_dsread<FI * FJ, FI + FJ>();
            } else {
                rdh(t, 0, wa, xa);
            }
            __builtin_amdgcn_sched_barrier(0);
            rdh(t, 1, wb, xb);
            if (domm) mmh(wa, xa);
            interleave_mfma_dsread<FI * FJ, FI + FJ>();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (domm) mmh(wb, xb);
    } else {
        f16x8 wA[2][FI], xA[2][FJ], wB[2][FI], xB[2][FJ];
        auto step = [&](int t, f16x8 (&wr)[2][FI], f16x8 (&xr)[2][FJ], const f16x8 (&wm_)[2][FI], const f16x8 (&xm_)[2][FJ]) {
            GTAV_STAMP_SLOT(GTAV_DBG(p, 32) && w == 0 && lane == 0 && t < 24, 32 + t);   // compute wave 0 arrives at barrier t
            wait_lgkm0();
            wg_barrier();
            if (t == 0) GTAV_STAMP(bs.t[1]);
            if (t > 0 && !GTAV_DBG(p, 2)) {
                rd(t, wr, xr);
                mm(wm_, xm_);
                interleave_mfma_dsread<2 * FI * FJ, 2 * (FI + FJ), 1>();   // one read in each of the first MFMAs' issue shadows
            } else {
                rd(t, wr, xr);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        int t = 0;
        for (; t + 1 < nkt; t += 2) {
            step(t, wA, xA, wB, xB);
            step(t + 1, wB, xB, wA, xA);
        }
        if (t < nkt) {
            step(t, wA, xA, wB, xB);
            if (!GTAV_DBG(p, 2)) mm(wA, xA);
        } else if (!GTAV_DBG(p, 2)) {
            mm(wB, xB);
        }
    }
}

template <int EPIX, int NS, int FI, int FJ, int WN, int WM, int NL>
__global__ __launch_bounds__(64 * (WN * WM + NL), 1) void gemm_l_kernel(GemmParams p) {
    constexpr int EPI = epi_base(EPIX);
    constexpr int TNB = 16 * FI * WN, TM = 16 * FJ * WM;
    constexpr int STAGE = (2 * FI * WN + 2 * FJ * WM) * 1024;
    // ring, or the QKV epilogue's pitched LDS image + token table (qkv_staged), whichever is larger
    constexpr int PNB = (TNB / 8 + 7) / 8 * 8 * 16, PTB = (TM / 8 + 7) / 8 * 8 * 16;
    constexpr int EPIB = (TM * PNB > TNB * PTB ? TM * PNB : TNB * PTB) + TM * 8 + (epi_is_fold_consumer(EPIX) ? fold_lds_bytes(TNB) : 0);
    extern __shared__ __attribute__((aligned(16))) char smem_l[];
    static_assert(NS * STAGE >= EPIB, "the ring must cover the epilogue's LDS image");
    char* smem = smem_l;
    // Every kernel argument the prologue needs is fetched HERE, in one batch of scalar loads behind one wait.  Left to itself hipcc
    // loads GemmParams field by field in the basic blocks that use them: five to six dependent s_load / s_waitcnt lgkmcnt(0) round
    // trips (100-200 ns each from a cold scalar cache) in front of the loaders' first fill (first fill 0.5-0.6 us after block entry,
    // profiles/round2 stamps).  The empty asm statements are uses that pin the loads to this point.
#define GTAV_PIN_S(x) asm volatile("" ::"s"(x))
    GTAV_PIN_S(p.X); GTAV_PIN_S(p.W); GTAV_PIN_S(p.M); GTAV_PIN_S(p.N); GTAV_PIN_S(p.K); GTAV_PIN_S(p.splitk);
    GTAV_PIN_S(p.tm.tiles_m); GTAV_PIN_S(p.tm.tiles_n); GTAV_PIN_S(p.tm.gn); GTAV_PIN_S(p.tm.group); GTAV_PIN_S(p.tm.tiles);
    GTAV_PIN_S(p.tm.rcp_tiles); GTAV_PIN_S(p.tm.rcp_group); GTAV_PIN_S(p.tm.rcp_gn); GTAV_PIN_S(p.tm.rcp_gnlast);
    GTAV_PIN_S(p.bias); GTAV_PIN_S(p.out); GTAV_PIN_S(p.ldo); GTAV_PIN_S(p.qkv_mode); GTAV_PIN_S(p.D);
#undef GTAV_PIN_S
    BlockStamps bs;
    bs.begin(p);
    int n0, m0, ks, kt0, nkt;
    tile_map_fast<EPI == EPI_PARTIAL, TNB, TM>(p, n0, m0, ks, kt0, nkt);
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool tr = false;
    f32x4 pbias[FI];
    constexpr bool FOLDC = epi_is_fold_consumer(EPIX), FOLDP = EPI == EPI_RESID_FOLD;
    FoldTok<FJ> ft;
    // fold consumer: its table chunk and row statistics, fetched by the compute waves behind their first fragment reads' wait (after_prologue runs
    // on the compute waves only: compute-thread index = threadIdx.x - 64 NL)
    auto pff = [&]() {
        if constexpr (FOLDC) {
            const int cw = (int)(threadIdx.x >> 6) - NL;
            fold_prefetch<TNB, TM, FJ>(p, n0, m0, (int)threadIdx.x - 64 * NL, __builtin_amdgcn_readfirstlane(m0 + 16 * FJ * (cw / WN)), cw >= 0, ft);
        }
    };
    if constexpr (EPI == EPI_QKV && FI * FJ <= 6 && WN * WM + NL <= 12) {   // small wave tiles: the RoPE values of the epilogue are fetched before the main loop (24 registers; not at the 128-register budget of the 16-wave block)
        tr = (p.qkv_mode == QKV_SPATIAL) && (n0 >= 2 * p.D);
        f32x4 prope[FI][FJ];
        auto pfq = [&]() {
            prefetch_bias<EPI, FI, WM, WN, NL>(p, n0, pbias);
            prefetch_rope<FI, FJ, WM, WN, NL>(p, n0, m0, prope);
            pff();
        };
        if (tr) mainloop_l<true, NS, FI, FJ, WN, WM, NL>(p, smem, n0, m0, kt0, nkt, acc, bs, pfq);
        else mainloop_l<false, NS, FI, FJ, WN, WM, NL>(p, smem, n0, m0, kt0, nkt, acc, bs, pfq);
        GTAV_STAMP(bs.t[2]);
        epilogue<EPIX, FI, FJ, WM, WN, NL, true>(p, acc, pbias, smem, n0, m0, ks, tr, prope, ft);
    } else if constexpr (EPI == EPI_QKV) {
        tr = (p.qkv_mode == QKV_SPATIAL) && (n0 >= 2 * p.D);
        auto pf = [&]() { prefetch_bias<EPI, FI, WM, WN, NL>(p, n0, pbias); pff(); };
        if (tr) mainloop_l<true, NS, FI, FJ, WN, WM, NL>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
        else mainloop_l<false, NS, FI, FJ, WN, WM, NL>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
        GTAV_STAMP(bs.t[2]);
        epilogue<EPIX, FI, FJ, WM, WN, NL, false>(p, acc, pbias, smem, n0, m0, ks, tr, acc, ft);
    } else if constexpr (FOLDP) {
        f32x4 xpre[FI][FJ];
        // (compute waves of the loader-wave kernel issue no other vector-memory operation in the K loop and never wait on vmcnt there: the whole
        // residual tile can be requested up front)
        auto pf = [&]() { prefetch_bias<EPI, FI, WM, WN, NL>(p, n0, pbias); prefetch_resid<FI, FJ, WM, WN, NL>(p, n0, m0, xpre, 0, FI * FJ); };
        mainloop_l<false, NS, FI, FJ, WN, WM, NL>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
        GTAV_STAMP(bs.t[2]);
        epilogue<EPIX, FI, FJ, WM, WN, NL, true>(p, acc, pbias, smem, n0, m0, ks, tr, xpre, ft);
    } else {
        auto pf = [&]() { prefetch_bias<EPI, FI, WM, WN, NL>(p, n0, pbias); pff(); };
        mainloop_l<false, NS, FI, FJ, WN, WM, NL>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
        GTAV_STAMP(bs.t[2]);
        epilogue<EPIX, FI, FJ, WM, WN, NL, false>(p, acc, pbias, smem, n0, m0, ks, tr, acc, ft);
    }
    bs.template end<true>(p);
}

// ---------------------------------------------------------------------------------------------------------------------
// Loader-wave GEMM with the WEIGHT operand straight into registers (shape 22, round 6).
//
// What bounds the loader-wave kernel at a few hundred tokens (the batch-1 window step: one 128 x 96 tile per CU, 16 K-steps) is how many fill bytes a CU keeps in
// flight against the latency of a weight that comes from HBM or the Infinity Cache: the ring holds NS - 1 = 3 stages of 28 KiB, of which 16 KiB are W — 48 KiB of W
// in flight per CU — and the K-step costs 0.45-0.5 us where ingest (64 B/clk) and the matrix pipes need 0.2-0.25 (profiles/round6/*per_class*: out-proj with 8
// K-steps 6.1 us, fc2's slices with 16 K-steps 9.9 us).  LDS capacity caps the ring; the register file does not: 8 compute waves x D = 4 K-steps x 4 KiB of W
// fragments = 128 KiB in flight per CU.  So here
//   * the compute waves load their OWN W fragments (16 features x 32 k per 16-byte lane load: the tile-major image makes a fragment half of eight 128-byte lines)
//     D K-steps ahead into a register ring (plain global loads: hipcc counts their vmcnt itself; the waves issue no other vector-memory operation in the loop);
//   * the loader waves move only X (the activations: L2-hot, 12 KiB per K-step) through the LDS ring, as before;
//   * W never touches LDS: the ring is 12 KiB per stage and the LDS array serves X fragment reads only.
// Price: the WM = 2 waves that share a W row tile each load it (W crosses the CU's texture path twice: 32 + 12 = 44 KiB of ingest per K-step against 28).
// ---------------------------------------------------------------------------------------------------------------------
template <bool TR, int NS, int FI, int FJ, int WN, int WM, int NL, int D, typename AfterPrologue>
__device__ __forceinline__ void mainloop_lw(const GemmParams& p, char* smem, int n0, int m0, int kt0, int nkt,
                                            f32x4 (&acc)[FI][FJ], BlockStamps& bs, AfterPrologue after_prologue) {
    static_assert(NS >= 3 && NL >= 1 && D >= 2 && D % 2 == 0 && (FI == 1 || FI == 2), "X ring of at least 3 stages; an even W register ring; 1 or 2 feature tiles per wave");
    constexpr int XPC = 2 * FJ * WM;                         // 1-KiB X pieces per stage
    constexpr int G = (XPC + NL - 1) / NL;
    constexpr int STAGE_BYTES = XPC * 1024;
    constexpr bool XDB = FI * FJ * 2 + 2 * FJ * 8 <= 96;     // X fragments of two K-steps in registers (one-step software pipeline) only while they fit
    const int tid = threadIdx.x, lane = tid & 63;
    const int wraw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nktot = p.K / TK;
    if (wraw < NL) {
        // ------------------------------------------------ loader wave: X only ------------------------------------------------
        const int lw = wraw;
        const int last_rt = (p.M - 1) >> 7;
        const unsigned voff = (unsigned)lane * 16u;
        const unsigned smem0 = lds_offset(smem);
        const char* sb[G];
        unsigned ldso[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            int q = lw * G + i;
            q = q < XPC ? q : XPC - 1;                       // the last wave repeats the final piece (same bytes, same place)
            const int row = m0 + 8 * q;
            int rt = row >> 7;
            rt = rt < last_rt ? rt : last_rt;                // ragged edge: re-read a valid tile (results are masked)
            sb[i] = (const char*)p.X + ((size_t)rt * nktot + kt0) * TILE_BYTES + ((row & 127) >> 3) * 1024;
            ldso[i] = smem0 + q * 1024;
        }
        auto stage = [&](int t) {
            const unsigned so = (unsigned)(t % NS) * STAGE_BYTES;
            const size_t go = (size_t)t * TILE_BYTES;
#pragma unroll
            for (int i = 0; i < G; ++i) glds16_s(sb[i] + go, voff, ldso[i] + so);
        };
        const int npro = nkt < NS - 1 ? nkt : NS - 1;
        for (int t = 0; t < npro; ++t) stage(t);
        for (int t = 0; t < nkt; ++t) {
            wait_vm_ring<NS, G>(nkt - 1 - t);                // this wave's share of tile t has landed
            wg_barrier();
            if (t == 0) GTAV_STAMP(bs.t[1]);
            if (t + NS - 1 < nkt && !GTAV_DBG(p, 1)) stage(t + NS - 1);
        }
        return;
    }
    // ------------------------------------------------ compute wave ------------------------------------------------
    const int w = wraw - NL;
    const int wn = w % WN, wm = w / WN;
    const int li = lane & 15, g = lane >> 4;
    int xoff[2];
    // this lane's W fragment source: row n0 + 16 FI wn + 16 i + li of the tile-major weight, 16-byte chunk 4 s + g of its 128-byte row (common.h tiled_off)
    const int last_wt = ((p.N + 127) >> 7) - 1;
    const int wrow = n0 + 16 * FI * wn + li;                 // (+ 16 i: 16 FI WN <= 128 and n0 % 128 == 0 keep every i inside one row tile)
    int wrt = wrow >> 7;
    wrt = wrt < last_wt ? wrt : last_wt;
    int soff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ch = ((4 * s + g) ^ (li & 7)) << 4;
        xoff[s] = (16 * FJ * wm + li) * 128 + ch;
        soff[s] = ch;                                        // (row & 7) == (li & 7): n0 + 16 FI wn + 16 i is a multiple of 8
    }
    after_prologue();   // register loads the epilogue wants early (bias): OLDER than every W load, so the first W wait covers them
    // The W loads and their waits are inline asm with hand-counted vmcnt: behind plain loads hipcc's wait-count pass loses the ring at the loop's back edge and
    // drains to vmcnt(3) ... vmcnt(0) inside every K-step (seen in the first build of this loop: no load stayed in flight across a step).  A wait statement names
    // the registers it retires as in-out operands, so no use of them can be scheduled above it (the pattern of mainloop256_tn).  vmcnt retires in issue order
    // and these waves issue nothing else in the loop: when K-step u is consumed, the loads issued after its own are those of steps u + 1 .. min(u + D - 1, nkt - 1).
    f16x8 wr[D][2][FI];
    const unsigned wvoff0 = (unsigned)((wrow & 127) * 128 + soff[0]), wvoff1 = (unsigned)((wrow & 127) * 128 + soff[1]);
    const char* const wbase = (const char*)p.W + ((size_t)wrt * nktot + kt0) * TILE_BYTES;    // wave-uniform
    auto ldw = [&](int t, f16x8 (&dst)[2][FI]) {
        const char* sb = wbase + (size_t)t * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst[0][i]) : "v"(wvoff0), "s"(sb), "n"(i * 2048) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst[1][i]) : "v"(wvoff1), "s"(sb), "n"(i * 2048) : "memory");
        }
    };
    // slot `a` is consumed while the loads of YOUNGER later K-steps (2 FI each) may still be in flight.  ONE asm statement per call site and a compile-time count:
    // a run-time choice between wait statements makes the in-out registers phis of several definitions, and hipcc then copies them — in one of the paths AHEAD of
    // the wait, i.e. before the data has landed (seen in the second build of this loop).  Hence the loop below has no branch around a wait: a steady part in which
    // every step has D - 1 younger steps in flight, and a tail of exactly D steps with D - 1, ..., 0 (host-checked: K-steps per slice a multiple of D).
    auto wait_w = [&](f16x8 (&a)[2][FI], auto younger) {
        constexpr int CNT = decltype(younger)::value * 2 * FI;
        if constexpr (FI == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]) : "n"(CNT));
        else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[0][0]), "+v"(a[1][0]) : "n"(CNT));
    };
#pragma unroll
    for (int d = 0; d < D; ++d) ldw(d, wr[d]);
    auto mmh = [&](const f16x8 (&wf)[FI], const f16x8 (&xv)[FJ]) {      // one 32-deep half of a K-step
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                if (TR) acc[i][j] = mfma16(xv[j], wf[i], acc[i][j], 0, 0, 0);
                else acc[i][j] = mfma16(wf[i], xv[j], acc[i][j], 0, 0, 0);
            }
    };
    const bool domm = !GTAV_DBG(p, 2);
    auto rdx = [&](int t, f16x8 (&dst)[2][FJ]) {
        const char* b = smem + (t % NS) * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < FJ; ++j) dst[s][j] = *(const f16x8*)(b + xoff[s] + j * 16 * 128);
    };
    if constexpr (XDB) {
        // one-step software pipeline on the X side (as mainloop_l): barrier u + 1 | read X(u + 1) | MFMAs of K-step u (W from register slot u % D) | refill the slot with step u + D
        f16x8 xf[2][2][FJ];
        auto next_x = [&](int t, f16x8 (&dst)[2][FJ]) {    // barrier t (X stage t complete and visible), then its fragment reads
            wait_lgkm0();
            wg_barrier();
            rdx(t, dst);
        };
        auto mm = [&](const f16x8 (&wf)[2][FI], const f16x8 (&xv)[2][FJ]) { mmh(wf[0], xv[0]); mmh(wf[1], xv[1]); };
        next_x(0, xf[0]);
        GTAV_STAMP(bs.t[1]);
        auto steady = [&](int u0, auto dc) {
            constexpr int d = decltype(dc)::value;
            next_x(u0 + d + 1, xf[(d + 1) & 1]);
            wait_w(wr[d], std::integral_constant<int, D - 1>{});
            if (domm) mm(wr[d], xf[d & 1]);
            __builtin_amdgcn_sched_barrier(0);
            ldw(u0 + d + D, wr[d]);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto tail = [&](int u0, auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d + 1 < D) next_x(u0 + d + 1, xf[(d + 1) & 1]);
            wait_w(wr[d], std::integral_constant<int, D - 1 - d>{});
            if (domm) mm(wr[d], xf[d & 1]);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto run = [&](auto&& f, int u0) {
            if constexpr (D == 4) { f(u0, std::integral_constant<int, 0>{}); f(u0, std::integral_constant<int, 1>{}); f(u0, std::integral_constant<int, 2>{}); f(u0, std::integral_constant<int, 3>{}); }
            else { static_assert(D == 8, "rings of 4 or 8 K-steps");
                   f(u0, std::integral_constant<int, 0>{}); f(u0, std::integral_constant<int, 1>{}); f(u0, std::integral_constant<int, 2>{}); f(u0, std::integral_constant<int, 3>{});
                   f(u0, std::integral_constant<int, 4>{}); f(u0, std::integral_constant<int, 5>{}); f(u0, std::integral_constant<int, 6>{}); f(u0, std::integral_constant<int, 7>{}); }
        };
        for (int u0 = 0; u0 < nkt - D; u0 += D) run(steady, u0);     // (nkt % D == 0: slot and X set of step u0 + d are d and d & 1)
        run(tail, nkt - D);
    } else {
        // wide token tiles (FJ = 6: 48 fragment registers per K-step): no second X set — barrier u | read both halves of X(u) | MFMAs half 0 behind lgkmcnt(FJ),
        // half 1 behind lgkmcnt(0) | refill.  The co-resident compute wave of the SIMD covers the LDS round trip.
        f16x8 xf[2][FJ];
        auto step = [&](int u, auto dc, auto yc, bool refill) {
            constexpr int d = decltype(dc)::value;
            wait_lgkm0();
            wg_barrier();
            if (u == 0) GTAV_STAMP(bs.t[1]);
            rdx(u, xf);
            wait_w(wr[d], yc);
            if (domm) { mmh(wr[d][0], xf[0]); mmh(wr[d][1], xf[1]); }
            __builtin_amdgcn_sched_barrier(0);
            if (refill) ldw(u + D, wr[d]);
            __builtin_amdgcn_sched_barrier(0);
        };
        static_assert(D == 8 || D == 4, "rings of 4 or 8 K-steps");
#define GTAV_LW_STEADY(dd) step(u0 + dd, std::integral_constant<int, dd>{}, std::integral_constant<int, D - 1>{}, true)
#define GTAV_LW_TAIL(dd) step(nkt - D + dd, std::integral_constant<int, dd>{}, std::integral_constant<int, D - 1 - dd>{}, false)
        for (int u0 = 0; u0 < nkt - D; u0 += D) {
            GTAV_LW_STEADY(0); GTAV_LW_STEADY(1); GTAV_LW_STEADY(2); GTAV_LW_STEADY(3);
            if constexpr (D == 8) { GTAV_LW_STEADY(4); GTAV_LW_STEADY(5); GTAV_LW_STEADY(6); GTAV_LW_STEADY(7); }
        }
        GTAV_LW_TAIL(0); GTAV_LW_TAIL(1); GTAV_LW_TAIL(2); GTAV_LW_TAIL(3);
        if constexpr (D == 8) { GTAV_LW_TAIL(4); GTAV_LW_TAIL(5); GTAV_LW_TAIL(6); GTAV_LW_TAIL(7); }
#undef GTAV_LW_STEADY
#undef GTAV_LW_TAIL
    }
}

template <int EPIX, int NS, int FI, int FJ, int WN, int WM, int NL, int D>
__global__ __launch_bounds__(64 * (WN * WM + NL), 1) void gemm_lw_kernel(GemmParams p) {
    constexpr int EPI = epi_base(EPIX);
    static_assert(!epi_is_fold_consumer(EPIX) && EPI != EPI_RESID_FOLD, "no LayerNorm-fold epilogues on this kernel");
    constexpr int TNB = 16 * FI * WN, TM = 16 * FJ * WM;
    static_assert(TNB <= 128 && 128 % TNB == 0, "a block's features stay inside one 128-row weight tile");
    extern __shared__ __attribute__((aligned(16))) char smem_l[];
    char* smem = smem_l;
#define GTAV_PIN_S(x) asm volatile("" ::"s"(x))
    GTAV_PIN_S(p.X); GTAV_PIN_S(p.W); GTAV_PIN_S(p.M); GTAV_PIN_S(p.N); GTAV_PIN_S(p.K); GTAV_PIN_S(p.splitk);
    GTAV_PIN_S(p.tm.tiles_m); GTAV_PIN_S(p.tm.tiles_n); GTAV_PIN_S(p.tm.gn); GTAV_PIN_S(p.tm.group); GTAV_PIN_S(p.tm.tiles);
    GTAV_PIN_S(p.tm.rcp_tiles); GTAV_PIN_S(p.tm.rcp_group); GTAV_PIN_S(p.tm.rcp_gn); GTAV_PIN_S(p.tm.rcp_gnlast);
    GTAV_PIN_S(p.bias); GTAV_PIN_S(p.out); GTAV_PIN_S(p.ldo); GTAV_PIN_S(p.qkv_mode); GTAV_PIN_S(p.D);
#undef GTAV_PIN_S
    BlockStamps bs;
    bs.begin(p);
    int n0, m0, ks, kt0, nkt;
    tile_map_fast<EPI == EPI_PARTIAL, TNB, TM>(p, n0, m0, ks, kt0, nkt);
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool tr = false;
    f32x4 pbias[FI];
    FoldTok<FJ> ft;
    auto pf = [&]() { prefetch_bias<EPI, FI, WM, WN, NL>(p, n0, pbias); };
    if constexpr (EPI == EPI_QKV) {
        tr = (p.qkv_mode == QKV_SPATIAL) && (n0 >= 2 * p.D);
        if (tr) mainloop_lw<true, NS, FI, FJ, WN, WM, NL, D>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
        else mainloop_lw<false, NS, FI, FJ, WN, WM, NL, D>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
    } else {
        mainloop_lw<false, NS, FI, FJ, WN, WM, NL, D>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
    }
    GTAV_STAMP(bs.t[2]);
    epilogue<EPIX, FI, FJ, WM, WN, NL, false>(p, acc, pbias, smem, n0, m0, ks, tr, acc, ft);
    bs.end(p);
}

// ---------------------------------------------------------------------------------------------------------------------
// Temporal QKV projection + causal temporal attention in ONE launch (the window step at batch 1).
//
// The temporal attention of one (batch item, position, head) needs q, k, v of that head for the <= 5 frames of the window at that
// position and nothing else (model/attention.py:41-71).  So with
//   * the activation rows in (b, position group of 16, frame, position in group) order — written that way by the LayerNorm in
//     front (launch_ln_modulate's tperm): an 80-row tile holds all 5 frames of 16 positions;
//   * the to_qkv weight rows in head-major order [head][q 64 | k 64 | v 64] (a copy made by gtav_dit_finalize): a 192-feature
//     tile holds one head's q, k and v;
// a block of the loader-wave GEMM (4 loader waves + 4 compute waves of 48 features x 80 tokens) owns every operand of 16 x 5
// attention rows of its head.  Epilogue: RoPE in registers -> fp16 q | k | v image in LDS (exactly the values the split path stores
// to memory) -> K / V rows to the temporal cache (a later context-cached step reads them) -> the attention with the arithmetic of
// attn_temporal_kernel, operand for operand (same fp16 inputs, same order of the fdot2 / DPP / exp2 / fma chain: bit-identical
// outputs) -> the out-projection's tile-major X operand.  One launch and the q / kv round trip through memory fewer per block:
// 16 launches per forward.  Grid = (M / 80) x heads = 144 blocks at batch 1; larger batches keep the split path (several rounds
// of one-block-per-CU tiles lose to the two-blocks-per-CU QKV GEMM there).
// ---------------------------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(512, 1) void gemm_qkvt_attn_kernel(GemmParams p) {
    constexpr int FI = 3, FJ = 5, WN = 4, WM = 1, NL = 4, TNB = 192, TM = 80, TMAX = 5;
    constexpr int PN = 384;   // LDS bytes per token row of the q | k | v image: 24 16-byte chunks, XOR-swizzled inside groups of 8
    extern __shared__ __attribute__((aligned(16))) char smem_l[];
    static_assert(NS * (2 * FI * WN + 2 * FJ * WM) * 1024 >= TM * PN, "the ring must cover the epilogue's LDS image");
    char* smem = smem_l;
#define GTAV_PIN_S(x) asm volatile("" ::"s"(x))
    GTAV_PIN_S(p.X); GTAV_PIN_S(p.W); GTAV_PIN_S(p.M); GTAV_PIN_S(p.N); GTAV_PIN_S(p.K);
    GTAV_PIN_S(p.tm.tiles_m); GTAV_PIN_S(p.tm.tiles_n); GTAV_PIN_S(p.tm.gn); GTAV_PIN_S(p.tm.group); GTAV_PIN_S(p.tm.tiles);
    GTAV_PIN_S(p.tm.rcp_group); GTAV_PIN_S(p.tm.rcp_gn); GTAV_PIN_S(p.tm.rcp_gnlast);
#undef GTAV_PIN_S
    BlockStamps bs;
    bs.begin(p);
    int n0, m0, ks, kt0, nkt;
    tile_map_fast<false, TNB, TM>(p, n0, m0, ks, kt0, nkt);
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    mainloop_l<false, NS, FI, FJ, WN, WM, NL>(p, smem, n0, m0, kt0, nkt, acc, bs, []() {});
    GTAV_STAMP(bs.t[2]);
    const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) - NL, li = lane & 15, g = lane >> 4;
    const int head = n0 / TNB;
    const int pgs = p.S >> 4, tile_m = m0 / TM, b = tile_m / pgs, pg = tile_m - b * pgs;   // the tile = (batch item b, positions 16 pg ..)
    float amax = 0.f;
    __syncthreads();   // every wave is done reading the last K-step's stage
    if (w >= 0) {
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int nl = 16 * FI * w + 16 * i + 4 * g;          // feature inside [q | k | v] of this head
            const bool rope = nl < 128;
            const int d = nl & 63;
#pragma unroll
            for (int j = 0; j < FJ; ++j) {                      // token group j = frame j of the window, lane li = position in the group
                const int ml = 16 * j + li;
                f32x4 v = acc[i][j];
                if (rope) {
                    const f32x4 cs = *(const f32x4*)(p.rope_cs + (p.t0 + j) * 64 + d);
                    v = rope4(v, cs);
                }
                char* dst = smem + ml * PN + (((nl >> 3) ^ (ml & 7)) << 4) + ((nl >> 2) & 1) * 8;
                *(uint2*)dst = pack4(amax, v[0], v[1], v[2], v[3]);
            }
        }
    }
    sat_report(amax, p.err_flag);
    __syncthreads();
    // K / V rows of this head into the temporal cache ([b][frame][position][k D | v D]): 80 rows x 16 chunks of 16 bytes
    for (int q = tid; q < TM * 16 && !GTAV_DBG(p, 512); q += 512) {   // (experiments build, debug bit 9: no cache rows, timing only)
        const int r = q >> 4, cc = q & 15, t = r >> 4, pl = r & 15;
        const uint4 val = *(const uint4*)(smem + r * PN + (((8 + cc) ^ (r & 7)) << 4));
        const size_t slot = ((size_t)b * p.Tmax + p.t0 + t) * p.S + 16 * pg + pl;
        f16* dst = p.k + slot * 2 * p.D + (cc >= 8 ? p.D : 0) + head * 64 + (cc & 7) * 8;
        if (p.out_sc1) store16q_sc1(dst, val);
        else *(uint4*)dst = val;
    }
    // attention: one thread per (position pl, 8-feature chunk c) and query-frame PAIR — frames (0, 4), (1, 3), (2): 6 + 6 + 3 keys, one
    // round of 384 threads instead of 512 + 128; the 8 lanes of a head row are an aligned lane group (DPP reduction)
    union H8 { f16x8 v; f16x2 h[4]; };
    if (tid < 3 * 128) {
        const int c = tid & 7, pl = (tid >> 3) & 15, grp = tid >> 7;     // wave-uniform grp
        H8 k8[TMAX], v8[TMAX];
        const int thi = TMAX - 1 - grp;                                    // the later query frame of the pair: keys 0 .. thi
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t <= thi) {
                const int r = 16 * t + pl;
                k8[t].v = *(const f16x8*)(smem + r * PN + (((8 + c) ^ (r & 7)) << 4));
                v8[t].v = *(const f16x8*)(smem + r * PN + (((16 + c) ^ (r & 7)) << 4));
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tl = half == 0 ? thi : grp;
            if (half == 1 && grp == thi) break;                            // the middle frame is its own pair
            H8 q8;
            {
                const int r = 16 * tl + pl;
                q8.v = *(const f16x8*)(smem + r * PN + ((c ^ (r & 7)) << 4));
            }
            float sc[TMAX];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                sc[t] = -INFINITY;
                if (t <= tl) {   // causal (model/attention.py:62-64)
                    float dsum = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) dsum = dot2acc(q8.h[e], k8[t].h[e], dsum, false);
                    sc[t] = group8_sum(dsum) * 0.125f;
                    mx = fmaxf(mx, sc[t]);
                }
            }
            float den = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                if (t <= tl) {
                    const float pr = __builtin_amdgcn_exp2f((sc[t] - mx) * 1.4426950408889634f);
                    den += pr;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(pr, (float)v8[t].v[e], o[e]);
                }
            }
            const float inv = 1.0f / den;
            union { f16x8 h; u32x4 u; } o8;
#pragma unroll
            for (int e = 0; e < 8; ++e) o8.h[e] = (f16)(o[e] * inv);
            const int row = (b * p.Tq + tl) * p.S + 16 * pg + pl;   // the token's row in (b, frame, position) order: the out-projection's X
            store16_sc1((f16*)p.out + tiled_off(row, head * 64 + c * 8, p.D), o8.u);
        }
    }
    bs.template end<true>(p);
}


// ---------------------------------------------------------------------------------------------------------------------
// Spatial QKV projection + spatial attention in ONE launch (round 6; frames of 144 tokens, the default from 5 frames per step on).
//
// The spatial attention of one (frame, head) needs q, k, v of that head for the 144 tokens of the frame and nothing else (model/attention.py:16-38), and
// nothing downstream reads the spatial q / k / v again.  So a block of the loader-wave GEMM whose tile is (one head's 192 q | k | v features) x (one frame's 144
// tokens) owns every operand of that attention item: 4 loader waves + 8 compute waves (4 along the features x 2 along the tokens, 48 x 80 each; the X tile is
// padded to 160 rows with whatever follows the frame, computed and dropped).  The weight rows come in an order of their own (gtav_op_qkv_head_major_spatial):
// the compute waves of feature quarter wn hold features 16 wn .. 16 wn + 15 of q, of k AND of v (feature groups i = 0, 1, 2), so
//   * q and k of a wave rotate by the SAME (cos, sin) values — one set of 5 RoPE loads per lane, fetched before the main loop;
//   * the v group is accumulated transposed (mainloop_l TRI = 2: tokens on accumulator rows), exactly the arithmetic of the split path's transposed V tiles,
//     and leaves as 8-byte rows of the V^T image.
// Epilogue: RoPE in registers -> fp16 Q | K | V^T images in LDS, in the layouts attn_spatial_1p_kernel stages from memory (the same fp16 values the split path
// stores) -> attn_1p_tile (attn_tile.h, the body that kernel runs), one 16-query tile per wave (nine tiles on twelve waves: one round) -> the out-projection's
// tile-major X operand.  One launch and the q / k / v^T round trip through memory (2 x 4.4 MB at 720 tokens) fewer per block: 16 launches per step.
// Grid = frames x heads: 80 blocks at batch 1 (-1.3 us per spatial half-block against the two launches), 128 at the context-cached step of batch 8 (-4.9 us),
// 640 at the window step of batch 8 (2.5 residency rounds: -12.8 us) — profiles/round6/fused_spatial_*.txt.  The K-step (0.60 us) is the LDS port's: 128 KB of
// fragment reads + 44 KB of fills per step against 0.42 us of MFMA issue per SIMD.
// ---------------------------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(768, 1) void gemm_qkvs_attn_kernel(GemmParams p) {
    // 4 loader + 8 compute waves (4 along the features x 2 along the tokens, 48 x 80 each): the X tile is 160 rows — the frame's 144 tokens and 16 rows of whatever
    // follows, computed and dropped — so that the tokens divide over two waves per SIMD (with ONE compute wave per SIMD, 48 x 144, a fragment read stuck behind the
    // LDS port held that SIMD's MFMA issue too: K loop 10.0 us for 6.0 us of MFMA issue, profiles/round6/fused_spatial_*_block_stamps*.txt)
    constexpr int FI = 3, FJ = 5, WN = 4, WM = 2, NL = 4, TNB = 192, TM = 144, NQT = TM / 16, NK = 10;
    constexpr int S_pad = 16 * NK, vstride = (S_pad + 8) * 2;                                     // attn_tile.h's K / V^T image geometry
    constexpr int QS = 0, KS = TM * 128, VS = KS + S_pad * 128, IMG = VS + 64 * vstride;          // LDS image behind the main loop: Q | K | V^T
    extern __shared__ __attribute__((aligned(16))) char smem_l[];
    static_assert(NS * (2 * FI * WN + 2 * FJ * WM) * 1024 >= IMG, "the ring must cover the epilogue's LDS image");
    static_assert(16 * FJ * WM >= TM && 16 * FJ * WM == S_pad, "two token halves cover the frame");
    char* smem = smem_l;
#define GTAV_PIN_S(x) asm volatile("" ::"s"(x))
    GTAV_PIN_S(p.X); GTAV_PIN_S(p.W); GTAV_PIN_S(p.M); GTAV_PIN_S(p.N); GTAV_PIN_S(p.K);
    GTAV_PIN_S(p.tm.tiles_m); GTAV_PIN_S(p.tm.tiles_n); GTAV_PIN_S(p.tm.gn); GTAV_PIN_S(p.tm.group); GTAV_PIN_S(p.tm.tiles);
    GTAV_PIN_S(p.tm.rcp_group); GTAV_PIN_S(p.tm.rcp_gn); GTAV_PIN_S(p.tm.rcp_gnlast);
#undef GTAV_PIN_S
    BlockStamps bs;
    bs.begin(p);
    int n0, m0, ks, kt0, nkt;
    tile_map_fast<false, TNB, TM>(p, n0, m0, ks, kt0, nkt);   // tiles step by the frame's 144 rows; the main loop reads 160 from m0 (past the last frame: a valid tile again, dropped)
    const int tid = threadIdx.x, lane = tid & 63, wraw = tid >> 6, w = wraw - NL, li = lane & 15, g = lane >> 4;
    const int wn = w & (WN - 1), wm = w >> 2;   // compute waves: feature quarter, token half
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 cs[FJ];   // (cos, sin, cos, sin) of head features 16 wn + 4 g .. + 3 at positions 16 (5 wm + j) + li: q and k alike
    auto pf = [&]() {
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int pos = 16 * (FJ * wm + j) + li;
            cs[j] = *(const f32x4*)(p.rope_cs + (pos < TM ? pos : TM - 1) * 64 + 16 * wn + 4 * g);
        }
    };
    mainloop_l<false, NS, FI, FJ, WN, WM, NL, 2, true>(p, smem, n0, m0, kt0, nkt, acc, bs, pf);
    GTAV_STAMP(bs.t[2]);
    const int head = n0 / TNB, frame = m0 / TM;
    float amax = 0.f;
    __syncthreads();   // every wave is done reading the last K-step's stage
    if (w >= 0) {
        const int d = 16 * wn + 4 * g;
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int jj = FJ * wm + j;          // token group of the frame; group 9 (rows 144 .. 159) belongs to whatever follows the frame
            if (jj < NQT) {
                const int ml = 16 * jj + li;
                const int off = ml * 128 + (((d >> 3) ^ (ml & 7)) << 4) + ((d >> 2) & 1) * 8;
                const f32x4 qv = rope4(acc[0][j], cs[j]), kv = rope4(acc[1][j], cs[j]);
                *(uint2*)(smem + QS + off) = pack4(amax, qv[0], qv[1], qv[2], qv[3]);
                *(uint2*)(smem + KS + off) = pack4(amax, kv[0], kv[1], kv[2], kv[3]);
                const f32x4 vv = acc[2][j];   // transposed tile: feature 16 wn + li, tokens 16 jj + 4 g .. + 3
                *(uint2*)(smem + VS + (16 * wn + li) * vstride + (16 * jj + 4 * g) * 2) = pack4(amax, vv[0], vv[1], vv[2], vv[3]);
            }
        }
    } else {
        // the loader waves clear what the tile body reads beyond the 144 tokens: key rows 144 .. 159 (masked, but kept as attn_spatial_1p_kernel has them)
        // and V^T columns 144 .. 159 (probability 0 times whatever the ring left there must be 0)
        for (int q = tid; q < (S_pad - TM) * 8; q += 64 * NL) *(uint4*)(smem + KS + TM * 128 + q * 16) = make_uint4(0, 0, 0, 0);
        for (int q = tid; q < 64 * (S_pad - TM) / 8; q += 64 * NL) {
            const int dd = q / ((S_pad - TM) / 8), c = q - dd * ((S_pad - TM) / 8);
            *(uint4*)(smem + VS + dd * vstride + TM * 2 + c * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    sat_report(amax, p.err_flag);
    __syncthreads();
    if (GTAV_DBG(p, 64)) GTAV_STAMP(bs.t[2]);   // experiments build, debug bit 6: the "main loop end" stamp moves behind the image (tools/gemm_stamps.py then splits the epilogue)
    if (wraw < NQT) {   // nine query tiles, twelve waves: one round
        const int qr = wraw * 16 + li;
        const char* qrow = smem + QS + qr * 128;
        f16x8 qf[2];
        qf[0] = *(const f16x8*)(qrow + ((g ^ (qr & 7)) << 4));
        qf[1] = *(const f16x8*)(qrow + (((4 + g) ^ (qr & 7)) << 4));
        attn_1p_tile<NK>(smem + KS, smem + VS, qf, TM, wraw * 16, (f16*)p.out, frame * TM + wraw * 16, head * 64, p.D, lane, 0);
    }
    bs.template end<true>(p);
}


// ---------------------------------------------------------------------------------------------------------------------
// Persistent loader-wave GEMM for large M (shape 30, round 3).
//
// What the round-3 measurements say about the large-M launches (docs/LABNOTES.md 4.7 / 4.8): the 512 resident blocks of a launch are in the same phase, so
// the prologue (first tiles from HBM: 2.2 us) and the epilogue (24 MB of stores in one burst: 5.7 us) of every tile are exposed — 31 % of a residency
// round — and nothing of it can hide under the NEXT tile of the same block while the waves that store are the waves that wait for fills: vmcnt retires in
// issue order, so a wave with stores in flight cannot see its younger fills land.  Here the roles are split for good:
//   * 4 loader waves own every LDS-DMA fill and run AHEAD across tile boundaries (the ring never drains: a tile's first K-steps are already in flight
//     while the compute waves finish the previous tile);
//   * 8 compute waves (64 x 48 each, 128 features x 192 tokens per block) never wait on vmcnt: MFMAs from LDS fragments, then the epilogue of the tile
//     STRAIGHT FROM REGISTERS (pairs of lanes exchange halves for 16-byte stores; no LDS staging, no barrier), whose stores drain under the next tile's K loop;
//   * one block per CU, persistent over tiles blockIdx + i gridDim of the XCD-aware order; ONE s_barrier per K-step shared by all 12 waves.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int LP_MAXF = 8;   // EPI_RESID: frames (gate rows) a 192-token tile may touch: tokens per gate row >= 32 (host-checked)
__device__ __forceinline__ void lp_tile_of(const GemmParams& p, int v, int tiles_m, int tiles_n, int TNB, int TM, int& n0, int& m0) {
    const int T = tiles_m * tiles_n;
    const int xcd = v & 7, qq = T >> 3, rr = T & 7;
    const int tile_id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (v >> 3);
    int gn = p.tm.gn;
    const int group = tiles_m * gn;
    const int ng = tile_id / group, rem = tile_id - ng * group;
    const int n_first = ng * gn;
    if (n_first + gn > tiles_n) gn = tiles_n - n_first;
    const int tile_m = rem / gn, tile_n = n_first + (rem - tile_m * gn);
    n0 = tile_n * TNB;
    m0 = tile_m * TM;
}

template <int EPI, int NS, int FI = 4, int FJ = 3, int WN = 2, int WM = 4>
__global__ __launch_bounds__(768, 1) void gemm_lp_kernel(GemmParams p) {
    static_assert(WN * WM == 8, "8 compute waves");
    constexpr int NL = 4, TNB = 16 * FI * WN, TM = 16 * FJ * WM;
    constexpr int WPC = 2 * FI * WN, XPC = 2 * FJ * WM, NP = WPC + XPC, G = NP / NL, STAGE_BYTES = NP * 1024;
    static_assert(NP % NL == 0, "pieces divide over the loader waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wraw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nktot = p.K / TK, nkt = nktot;
    const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TNB - 1) / TNB, T = tiles_m * tiles_n;
    const int nblk = gridDim.x;
    const int ntile = (T - (int)blockIdx.x + nblk - 1) / nblk;       // tiles of this block (grid <= T)
    if (wraw < NL) {
        // ------------------------------------------------ loader wave ------------------------------------------------
        const int last_wt = ((p.N + 127) >> 7) - 1, last_rt = (p.M - 1) >> 7;
        const unsigned voff = (unsigned)lane * 16u;
        const unsigned smem0 = lds_offset(smem);
        const char* sb[G];
        auto setup = [&](int ti) {
            int n0, m0;
            lp_tile_of(p, (int)blockIdx.x + ti * nblk, tiles_m, tiles_n, TNB, TM, n0, m0);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int q = wraw * G + i;
                const bool isw = q < WPC;
                const int row = isw ? n0 + 8 * q : m0 + 8 * (q - WPC);
                int rt = row >> 7;
                const int lim = isw ? last_wt : last_rt;
                rt = rt < lim ? rt : lim;
                sb[i] = (const char*)(isw ? p.W : p.X) + (size_t)rt * nktot * TILE_BYTES + ((row & 127) >> 3) * 1024;
            }
        };
        const int S = ntile * nkt;
        int it = 0, ik = 0, islot = 0;                                // next K-step to issue: tile it, step ik, ring slot islot
        auto issue = [&]() {
            if (ik == 0) setup(it);
            const unsigned so = smem0 + (unsigned)islot * STAGE_BYTES + (unsigned)(wraw * G) * 1024u;
            const size_t go = (size_t)ik * TILE_BYTES;
#pragma unroll
            for (int i = 0; i < G; ++i) glds16_s(sb[i] + go, voff, so + i * 1024);
            if (++ik == nkt) { ik = 0; ++it; }
            if (++islot == NS) islot = 0;
        };
        const int npro = S < NS - 1 ? S : NS - 1;
        for (int g = 0; g < npro; ++g) issue();
        for (int g = 0; g < S; ++g) {
            wait_vm_ring<NS, G>(S - 1 - g);                           // this wave's share of step g has landed
            wg_barrier();
            if (g + NS - 1 < S) issue();
        }
        return;
    }
    // ------------------------------------------------ compute wave ------------------------------------------------
    const int w = wraw - NL, wn = w % WN, wm = w / WN, li = lane & 15, g4 = lane >> 4;
    int woff[2], xoff[2];
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
        const int ch = ((4 * sh + g4) ^ (li & 7)) << 4;
        woff[sh] = (16 * FI * wn + li) * 128 + ch;
        xoff[sh] = WPC * 1024 + (16 * FJ * wm + li) * 128 + ch;
    }
    int slot = 0;
    float amax = 0.f;
    // (the next-weight L2 prefetch of the small-M loader-wave kernels, l2_prefetch_next, was tried here for the to_qkv / fc1 launch behind this one at
    // M = 5760: out-proj 16.9 -> 18.8 us, fc2 52.5 -> 54.2, consumers unchanged — profiles/round3/forward_ab_B8_next_weight_prefetch_persistent_kernel.txt)
    for (int ti = 0; ti < ntile; ++ti) {
        int n0, m0;
        lp_tile_of(p, (int)blockIdx.x + ti * nblk, tiles_m, tiles_n, TNB, TM, n0, m0);
        f32x4 acc[FI][FJ];
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 pbias[FI];
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int n = n0 + 16 * FI * wn + 16 * i + 4 * g4;
            pbias[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EPI != EPI_PARTIAL && EPI != EPI_RESID && p.bias && n < p.N) pbias[i] = *(const f32x4*)(p.bias + n);   // (EPI_RESID: in the epilogue, see there)
        }
        // EPI_RESID (in-place gated residual update, large M): the residual tile is requested HERE, at the head of the tile's K loop — the compute waves issue
        // no other vector-memory load and never wait on vmcnt in the loop, so the 48 registers per lane arrive under the MFMAs and the epilogue finds them
        constexpr bool RES = EPI == EPI_RESID;
        f32x4 xres[RES ? FI : 1][RES ? FJ : 1];
        // ... and the bias / gate vectors of the tile (its TNB features of the <= LP_MAXF frames its tokens belong to: a few KiB) are staged in LDS behind the
        // ring by the compute waves themselves, one 16-byte chunk per thread, requested in front of the residual tile.  The epilogue then reads them with
        // ds_read: it issues NO vector-memory load, and a load behind its stores would wait for their write-through (vmcnt retires in order: measured
        // twelve times per tile, +14 us per launch).  Two areas alternate by tile parity: a wave may stage tile t + 1 while a slower one still reads tile t,
        // and the >= 1 K-step barriers of tile t + 1 lie between the last read of an area and its next write.
        const int gfirst = RES ? m0 / p.rows_per_gate : 0;
        float* const garea = (float*)(smem + NS * STAGE_BYTES) + (ti & 1) * (LP_MAXF + 1) * TNB;
        if constexpr (RES) {
            // (the lane number is re-derived here and in the epilogue — v_mbcnt, opaque to CSE — instead of living across the K loop: at the 168-register
            // budget of a 12-wave block hipcc spilled it to scratch and reloaded it once per tile behind a vmcnt(0))
            int lane_s = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(lane_s));
            const int ct = 64 * w + lane_s;                                 // compute-thread index, 0 .. 511
            if (ct < (LP_MAXF + 1) * (TNB / 4)) {
                const int r = ct / (TNB / 4), cq = ct - r * (TNB / 4);     // staged row (frame slot, LP_MAXF = bias), 4-feature chunk
                int n = n0 + 4 * cq;
                n = n < p.N ? n : 0;
                f32x4 val = r < LP_MAXF ? f32x4{1.f, 1.f, 1.f, 1.f} : f32x4{0.f, 0.f, 0.f, 0.f};
                if (r < LP_MAXF) {
                    if (p.gate) {
                        const int nfr = (p.M - 1) / p.rows_per_gate;
                        int fr = gfirst + r;
                        fr = fr < nfr ? fr : nfr;
                        if (p.gate_rows) fr = p.gate_rows[fr];
                        val = *(const f32x4*)(p.gate + (size_t)fr * p.gate_stride + n);
                    }
                } else if (p.bias) {
                    val = *(const f32x4*)(p.bias + n);
                }
                *(f32x4*)(garea + r * TNB + 4 * cq) = val;
            }
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int n = n0 + 16 * FI * wn + 16 * i + 4 * g4;
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    const int m = m0 + 16 * FJ * wm + 16 * j + li;
                    xres[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (n < p.N && m < p.M) xres[i][j] = *(const f32x4*)((const float*)p.out + (size_t)m * p.ldo + n);
                }
            }
        }
        f16x8 wa[FI], xa[FJ], wb[FI], xb[FJ];
        auto rdh = [&](const char* b, int sh, f16x8 (&wf)[FI], f16x8 (&xf)[FJ]) {
#pragma unroll
            for (int i = 0; i < FI; ++i) wf[i] = *(const f16x8*)(b + woff[sh] + i * 16 * 128);
#pragma unroll
            for (int j = 0; j < FJ; ++j) xf[j] = *(const f16x8*)(b + xoff[sh] + j * 16 * 128);
        };
        auto mmh = [&](const f16x8 (&wf)[FI], const f16x8 (&xf)[FJ]) {
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[i][j] = mfma16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        };
        for (int t = 0; t < nkt; ++t) {
            wait_lgkm0();
            wg_barrier();
            const char* b = smem + slot * STAGE_BYTES;
            if (t > 0) {
                rdh(b, 0, wa, xa);
                mmh(wb, xb);
                interleave_mfma_dsread<FI * FJ, FI + FJ>();
            } else {
                rdh(b, 0, wa, xa);
            }
            __builtin_amdgcn_sched_barrier(0);
            rdh(b, 1, wb, xb);
            mmh(wa, xa);
            interleave_mfma_dsread<FI * FJ, FI + FJ>();
            __builtin_amdgcn_sched_barrier(0);
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
        mmh(wb, xb);
        // ---- epilogue straight from the accumulators (no LDS, no barrier): the stores drain under the next tile's K loop ----
        int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // re-derived (see the staging above): li / g4 of the prologue die with the K loop's offsets
        asm volatile("" : "+v"(lane_e));
        const int li_e = lane_e & 15, g4_e = lane_e >> 4;
        int gslot[RES ? FJ : 1];   // EPI_RESID: staged gate row (frame slot) of this lane's token in each of its FJ token groups
        if constexpr (RES) {
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                int mm = m0 + 16 * FJ * wm + 16 * j + li_e;
                mm = mm < p.M ? mm : p.M - 1;
                gslot[j] = mm / p.rows_per_gate - gfirst;
            }
#pragma unroll
            for (int i = 0; i < FI; ++i) pbias[i] = *(const f32x4*)(garea + LP_MAXF * TNB + 16 * FI * wn + 16 * i + 4 * g4_e);
        }
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int n = n0 + 16 * FI * wn + 16 * i + 4 * g4_e;
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const int m = m0 + 16 * FJ * wm + 16 * j + li_e;
                const bool ok = n < p.N && m < p.M;
                const f32x4 v = acc[i][j] + pbias[i];
                if constexpr (RES) {
                    if (ok) {
                        const f32x4 gq = *(const f32x4*)(garea + gslot[j] * TNB + 16 * FI * wn + 16 * i + 4 * g4_e);
                        const f32x4 r = xres[i][j] + gq * v;
                        float* dst = (float*)p.out + (size_t)m * p.ldo + n;
                        if (p.out_sc1) store16_sc1(dst, r);
                        else *(f32x4*)dst = r;
                    }
                } else if constexpr (EPI == EPI_PARTIAL) {
                    if (ok) {
                        float* dst = (float*)p.out + (size_t)m * p.ldo + n;
                        if (p.out_sc1) store16_sc1(dst, v);
                        else *(f32x4*)dst = v;
                    }
                } else {
                    const float xin[4] = {v[0], v[1], v[2], v[3]};
                    float yo[4];
                    if constexpr (EPI == EPI_GELU_TANH) gelu_tanh_f4(xin, yo);
                    else { yo[0] = xin[0]; yo[1] = xin[1]; yo[2] = xin[2]; yo[3] = xin[3]; }
                    const uint2 mine = pack4(amax, yo[0], yo[1], yo[2], yo[3]);
                    // lanes (li, g) and (li, g ^ 1) hold adjacent 4-feature groups of one token: one 16-byte store per pair
                    const unsigned o0 = __shfl_xor(mine.x, 16, 64), o1 = __shfl_xor(mine.y, 16, 64);
                    if (ok && !(lane_e & 16)) {
                        f16* dst = (f16*)p.out + tiled_off(m, n, p.ldo);
                        if (p.out_sc1) store16_sc1(dst, u32x4{mine.x, mine.y, o0, o1});
                        else *(uint4*)dst = uint4{mine.x, mine.y, o0, o1};
                    }
                }
            }
        }
    }
    if constexpr (EPI != EPI_PARTIAL) sat_report(amax, p.err_flag);
}

// (the persistent 256-token-tile kernel of round 4, shapes 40 / 41 — correct, race-screened, slower than shape 12 at every size tried — lives in gemm_experiments.inc)
// (the persistent ping-pong kernel of round 2, shape 16 — measured slower than shape 12 — lives in gemm_experiments.inc: experiments build only)

}  // namespace

// Shape / ring-depth overrides: they select among kernels that all compute the same result (the parity tests force every
// shape through them).  The debug bits (skip fills / skip MFMA: WRONG results, timing only) exist only in the
// -DGTAV_EXPERIMENTS build, which also reads GTAV_GEMM_DEBUG / GTAV_GEMM_SHAPE for whole-bench A/B runs.
// (per THREAD, like the handles: a test that forces a shape does not change what another thread's handle launches)
static thread_local int g_force_stages = 0, g_force_wm = GTAV_ENV_INT("GTAV_GEMM_SHAPE", 0);
static int g_debug = GTAV_ENV_INT("GTAV_GEMM_DEBUG", 0);
void gemm_set_stages(int ns) { g_force_stages = ns; }
void gemm_set_wm(int wm) { g_force_wm = wm; }
#ifdef GTAV_EXPERIMENTS
void gemm_set_debug(int bits) { g_debug = bits; }
static unsigned long long* g_stamps = nullptr;
static int g_stamp_blocks = 0;
void gemm_set_stamps(unsigned long long* buf, int max_blocks) { g_stamps = buf; g_stamp_blocks = max_blocks; }
#endif

// The persistent loader-wave kernel (shape 31) takes the narrow residual GEMMs (N <= 2048: out-proj, fc2) in ONE K slice once its 128 x 192 tiles cover
// most of the chip (>= 160 tiles): measured against the heuristic's own choice, split-K included (profiles/round3/persistent_loader_kernel_vs_heuristic_*.txt) —
// M = 4320: out-proj 17.2 -> 13.2 us, fc2 47.1 -> 36.3 us; M = 5760: 17.5 -> 14.3, 51.0 -> 45.1; M = 8640: 27.5 -> 24.0, 78.6 -> 75.0; M = 11 520: 27.9 -> 25.1,
// 87.6 -> 84.5; at 120 tiles (M = 2880) it loses (12.0 -> 12.8, 31.7 -> 34.0) and the split-K shapes stay.  One slab also halves what the LayerNorm behind fc2 reads.
static int g_lp_enable = GTAV_ENV_INT("GTAV_LP", 1);   // experiments build: 0 = round-2 selection, for A/B runs
static int g_resid_inplace = GTAV_ENV_INT("GTAV_RESID_INPLACE", 1);   // experiments build: 0 = slab + LayerNorm reduction at every M (A/B runs)
static bool lp_takes(int M, int N, int K) { return g_lp_enable && N <= 2048 && N % 8 == 0 && K >= 512 && cdiv(M, 192) * cdiv(N, 128) >= 160; }
// (rows_per_gate: tokens per gate vector — the persistent kernel stages at most LP_MAXF gate rows per 192-token tile; 0 = no gate.  A forced block shape
// other than 31 (tests) runs the one-shot kernels, where the in-place epilogue was measured slower than slab + LayerNorm: keep the slabs then.)
bool gemm_resid_inplace_ok(int M, int N, int K, int rows_per_gate) {
    return g_resid_inplace && lp_takes(M, N, K) && (rows_per_gate == 0 || rows_per_gate >= 32) && (g_force_wm == 0 || g_force_wm == 31);
}
int gemm_choose_splitk(int M, int N, int K) {
    if (lp_takes(M, N, K)) return 1;
#ifdef GTAV_EXPERIMENTS
    if (g_debug & 256) return 1;   // A/B with GTAV_RESID_INPLACE_MIN_M=1: full-K residual GEMMs on 64 x 48 tiles + in-place epilogue (measured slower, profiles/round2/forward_ab_B1_inplace_fullK.txt)
#endif
    const int tiles = cdiv(M, 128) * cdiv(N, TN);
    int s = 1;
    // Every slab is one more 16-byte load per thread in the LayerNorm behind the GEMM (~0.4 us per slab and launch at M = 720) and 3 MB more through the L2s that
    // hold the next GEMM's prefetched weight: from 48 tiles on (the batch-1 window step) a K slice is at least 512 deep — out-proj in two slices instead of four:
    // GEMM unchanged (6.7-7.0 us), LayerNorm 5.45 -> 4.7 us, forward -1.3 ... -3 % (profiles/round3/forward_ab_B1_splitk_slices.txt); fc2 keeps four slices of 1024
    // (two: +1.6 us).  Below 48 tiles (the 144-token cached step) the CUs matter more: 256 as before.
    static const int old_rule = GTAV_ENV_INT("GTAV_SPLITK_OLD", 0);   // experiments build: A/B
    const int min_slice = (tiles >= 48 && !old_rule) ? 512 : 256;
    while (tiles * s < 192 && s < 8 && (K / TK) % (s * 2) == 0 && K / (s * 2) >= min_slice) s *= 2;
    // long-K GEMMs whose 128 x 192 grid would fill the 512 block slots unevenly (160-320 tiles): two K slices make it
    // 320-640 blocks of half the length — fc2 at M = 5760: 69.5 -> 57.6 us, for one more slab (+5 us) in the next LayerNorm
    if (s == 1 && K >= 4096 && (K / TK) % 2 == 0) {
        const int t192 = cdiv(M, 192) * cdiv(N, TN);
        if (t192 >= 160 && t192 < 320) s = 2;
    }
    return s;
}

#define GEMM_LAUNCH(kern, grid, block) GTAV_LAUNCH(kern, grid, block, 0, stream, p)

// shape: 2 = 128x128 / 4 waves, 3 = 128x128 / 8 waves, 7 = 256x256 / 8 waves, phased K-tile (mainloop256),
//        8 = 96x96 / 6 waves, 9 = 128x96 / 6 waves, 11 = 64x48 / 6 waves, 14 = 64x96 / 6 waves (piece-granular mainloop_g; small M),
//        12 = 128x192 / 8 waves, two blocks per CU (mainloop_g; large M).
// (Shapes 4, 5, 6, 10 of round 1 — 128x256, loader-wave variants, 4-wave 128x192 — measured slower and were removed.)
// ---- persistent ping-pong kernel (shape 16) ----
static int g_num_cus[64] = {0};
static int device_cus(int* dev_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    *dev_out = dev;
    if (!g_num_cus[dev & 63]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_num_cus[dev & 63] = n;
    }
    return g_num_cus[dev & 63];
}
// Round-2 measurement (profiles/round2/pp_stamps_v1.txt): correct and bit-identical to the one-shot kernels, the overlap works (MAIN
// phases run back to back), but with only 8 of the 16 waves computing, a K-step costs 1.15-1.25 us against 0.68 us of fill time: the
// barrier-synchronised waves serialise into fill / read / MFMA phases, where two independent co-resident blocks interleave them.
// fc1 at M = 5760: 71-76 us against 57 us for shape 12.  Not selected by the heuristic (GTAV_PP=1 in the experiments build or a
// forced shape 16 run it); the de-phased shape 12 above gets the same overlap with 16 computing waves.
static int g_l_for_8 = GTAV_ENV_INT("GTAV_L_FOR_8", 1);   // experiments build: 0 keeps the 96 x 96 tile (shape 8) where the cost model picks it
#ifndef GTAV_EXPERIMENTS
bool gemm_pp_ok(int, int, int, int) { return false; }   // the ping-pong kernel (shape 16) is compiled into the experiments build only (gemm_experiments.inc)
#endif

// loader-wave kernels: dynamic LDS above 64 KiB needs the per-device opt-in once per instantiation
// block -> tile map constants of the loader-wave kernels (tile_map_fast)
// Width (in n-panels) of the tile groups of the block -> tile map (tile_map): the linear tile order is (K slice, group of gn
// n-panels, m-tile, panel in group) and every XCD takes a contiguous eighth of it.  One pass over a group streams the K slice of X
// once (its co-resident blocks share X row-tiles and W panels in the XCD's L2), so with G = tiles_n / gn groups and C = 8 / splitk
// XCDs per K slice the fabric sees  G x X  +  max(1, C / G) x W  bytes per slice: narrow groups re-stream X, wide groups make
// several XCDs fetch the same W panels.  Pick the minimum, with the W panels of a group capped at 2 MiB (they must survive in the
// 4 MiB L2 while the group's m-tiles sweep past).  Round 1's fixed gn = tiles_n / 8 is the G = 8 corner: right at M = 720 (X is
// small), 3.5x too much traffic for the N = 1024 GEMMs at M = 5760.
static int g_force_gn = GTAV_ENV_INT("GTAV_GEMM_GN", 0);   // experiments build: > 0 forces the group width, < 0 = round-1 rule (A/B runs)
static int choose_gn(int M, int N, int K, int tmb, int tnb, int splitk) {
    const int tiles_n = cdiv(N, tnb);
#ifdef GTAV_EXPERIMENTS
    if (g_debug & 1024) return tiles_n >= 8 ? tiles_n >> 3 : 1;   // debug bit 10: round-1 rule, for A/B runs in one process
#endif
    if (g_force_gn < 0) return tiles_n >= 8 ? tiles_n >> 3 : 1;
    if (g_force_gn > 0) return g_force_gn < tiles_n ? g_force_gn : tiles_n;
    const double ks = (double)K / (splitk > 0 ? splitk : 1);
    const double xs = 2.0 * M * ks, ws = 2.0 * N * ks, wpanel = 2.0 * tnb * ks;
    const double c = splitk >= 8 ? 1.0 : 8.0 / (splitk > 0 ? splitk : 1);
    auto cost_of = [&](int gn) {
        const int g = cdiv(tiles_n, gn);
        return g * xs + ws * (c > g ? c / g : 1.0);
    };
    int best = 1;
    double best_cost = 1e300;
    // the 2 MiB cap protects W panels that must SURVIVE in L2 while further m-tiles of the group sweep past; a grid that is resident all at
    // once (<= 512 block slots) has no sweep — its blocks walk K in step and share the current K window only — so the cap does not apply
    // (weight gradients of the training step, K = 11 520 tokens: a 2.9 MB W panel forced gn = 1 = every XCD streaming ALL of the 94 MB X
    // operand, 778 MB per launch at 7 TB/s; uncapped gn = 4: 282 MB)
    const bool one_round = (long long)cdiv(M, tmb) * tiles_n * (splitk > 0 ? splitk : 1) <= 512;
    for (int gn = 1; gn <= tiles_n; ++gn) {
        if (!one_round && gn > 1 && gn * wpanel > 2.0 * 1024 * 1024) break;
        const double cost = cost_of(gn);
        if (cost < best_cost * 0.999) best_cost = cost, best = gn;   // ties keep the narrower group (more XCD-local W)
    }
    // The model ignores how the groups line up with the XCDs' runs of tiles: where it predicts less than a 20 % saving keep round 1's
    // eighth-of-the-panels groups, which do line up (fc1 at M = 720: gn = 5 instead of 4 was predicted 1 % better and measured
    // 34.1 MB per launch against 27.4 MB)
    const int gn1 = tiles_n >= 8 ? tiles_n >> 3 : 1;
    if (best_cost > 0.8 * cost_of(gn1)) best = gn1;
    return best;
}

static int fill_tile_map(GemmParams::TileMap& tm, int M, int N, int K, int tmb, int tnb, int splitk) {
    tm.tiles_m = cdiv(M, tmb);
    tm.tiles_n = cdiv(N, tnb);
    tm.tiles = tm.tiles_m * tm.tiles_n;
    tm.gn = choose_gn(M, N, K, tmb, tnb, splitk);
    tm.group = tm.tiles_m * tm.gn;
    const int gnlast = tm.tiles_n % tm.gn ? tm.tiles_n % tm.gn : tm.gn;
    auto rcp = [](int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };   // ceil(2^32 / d), d >= 2
    GTAV_REQUIRE((long long)tm.tiles * splitk < 65536, "gemm: grid of %d tiles x %d slices is too large for the 32-bit reciprocal tile map", tm.tiles, splitk);
    // a divisor of 1 has no 32-bit reciprocal (it would be 2^32): 0 encodes it, div_rcp() then returns its argument
    tm.rcp_tiles = tm.tiles > 1 ? rcp(tm.tiles) : 0;
    tm.rcp_group = tm.group > 1 ? rcp(tm.group) : 0;
    tm.rcp_gn = tm.gn > 1 ? rcp(tm.gn) : 0;
    tm.rcp_gnlast = gnlast > 1 ? rcp(gnlast) : 0;
    return 0;
}

template <int EPI, int NS, int FI, int FJ, int WN, int WM, int NL>
static int launch_l(const GemmParams& p, int splitk, hipStream_t stream) {
    constexpr int LDS = NS * (2 * FI * WN + 2 * FJ * WM) * 1024;
    static unsigned long long attr_devs = 0;
    int dev = 0;
    GTAV_REQUIRE(device_cus(&dev) > 0, "gemm: no current device");
    if (!(attr_devs >> (dev & 63) & 1)) {
        GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_l_kernel<EPI, NS, FI, FJ, WN, WM, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_devs |= 1ull << (dev & 63);
    }
    GemmParams q = p;
    if (int rc_ = fill_tile_map(q.tm, p.M, p.N, p.K, 16 * FJ * WM, 16 * FI * WN, splitk)) return rc_;
    const dim3 grid(q.tm.tiles * splitk);
    GTAV_LAUNCH((gemm_l_kernel<EPI, NS, FI, FJ, WN, WM, NL>), grid, dim3(64 * (WN * WM + NL)), LDS, stream, q);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

// loader-wave kernel with the weight operand straight into registers (shape 22): dynamic LDS = the X ring, or the epilogue's image if that is larger
template <int EPI, int NS, int FI, int FJ, int WN, int WM, int NL, int D>
static int launch_lw(const GemmParams& p, int splitk, hipStream_t stream) {
    constexpr int TNB = 16 * FI * WN, TM = 16 * FJ * WM;
    constexpr int RING = NS * 2 * FJ * WM * 1024;
    constexpr int PNB = (TNB / 8 + 7) / 8 * 8 * 16, PTB = (TM / 8 + 7) / 8 * 8 * 16;
    constexpr int EPIB = (TM * PNB > TNB * PTB ? TM * PNB : TNB * PTB) + TM * 8;      // qkv_staged's pitched image + token table (gemm_l_kernel)
    constexpr int GELUB = (TNB / 64 > 0 ? TNB / 64 : 1) * TM * 128;                     // the tile-major GELU image
    constexpr int LDS = RING > EPIB ? (RING > GELUB ? RING : GELUB) : (EPIB > GELUB ? EPIB : GELUB);
    static unsigned long long attr_devs = 0;
    int dev = 0;
    GTAV_REQUIRE(device_cus(&dev) > 0, "gemm: no current device");
    if (!(attr_devs >> (dev & 63) & 1)) {
        GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_lw_kernel<EPI, NS, FI, FJ, WN, WM, NL, D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_devs |= 1ull << (dev & 63);
    }
    GemmParams q = p;
    if (int rc_ = fill_tile_map(q.tm, p.M, p.N, p.K, TM, TNB, splitk)) return rc_;
    const dim3 grid(q.tm.tiles * splitk);
    GTAV_LAUNCH((gemm_lw_kernel<EPI, NS, FI, FJ, WN, WM, NL, D>), grid, dim3(64 * (WN * WM + NL)), LDS, stream, q);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

// persistent loader-wave kernel (shape 30): one block per CU, dynamic LDS = ring of NS x 40 KiB
template <int EPI, int NS, int FI = 4, int FJ = 3, int WN = 2, int WM = 4>
static int launch_lp(const GemmParams& p, hipStream_t stream) {
    constexpr int TNB = 16 * FI * WN, TM = 16 * FJ * WM;
    constexpr int LDS = NS * (TNB + TM) * 128 + (EPI == EPI_RESID ? 2 * (LP_MAXF + 1) * TNB * 4 : 0);   // + the staged bias / gate rows, two tile parities
    if constexpr (EPI == EPI_RESID) GTAV_REQUIRE(!p.gate || (p.rows_per_gate > 0 && (TM - 1) / p.rows_per_gate + 2 <= LP_MAXF), "gemm/resid on the persistent kernel: %d tokens per gate row are too few (a tile may touch at most %d rows)", p.rows_per_gate, LP_MAXF);
    static unsigned long long attr_devs = 0;
    int dev = 0;
    const int cus = device_cus(&dev);
    GTAV_REQUIRE(cus > 0, "gemm: no current device");
    if (!(attr_devs >> (dev & 63) & 1)) {
        GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_lp_kernel<EPI, NS, FI, FJ, WN, WM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_devs |= 1ull << (dev & 63);
    }
    GemmParams q = p;
    if (EPI == EPI_RESID && !q.gate) q.rows_per_gate = 1 << 30;   // no gate: one staged row of ones
    const int T = cdiv(p.M, TM) * cdiv(p.N, TNB);
    q.tm.gn = choose_gn(p.M, p.N, p.K, TM, TNB, 1);
    GTAV_LAUNCH((gemm_lp_kernel<EPI, NS, FI, FJ, WN, WM>), dim3(T < cus ? T : cus), dim3(768), LDS, stream, q);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

#ifdef GTAV_EXPERIMENTS
#include "gemm_experiments.inc"   // laboratory: kernels, launchers and block shapes that measured slower than what the heuristic picks (tools/ only)
#endif

template <int EPI>
static int launch_epi(const GemmParams& p_in, int ns, int shape, int splitk, hipStream_t stream) {
    GemmParams p = p_in;
    auto set_gn = [&](int tmb, int tnb) { p.tm.gn = choose_gn(p.M, p.N, p.K, tmb, tnb, splitk); };
    // the LayerNorm-fold epilogues are instantiated for the shapes the heuristic can pick for them (2, 3, 11, 12, 14, 20)
    constexpr bool FOLDISH = EPI == EPI_RESID_FOLD || epi_is_fold_consumer(EPI);
    if constexpr (FOLDISH) GTAV_REQUIRE(shape == 2 || shape == 3 || shape == 11 || shape == 12 || shape == 13 || shape == 14 || shape == 20 || shape == 24 || shape == 26 || shape == 27 || shape == 28,
                                        "gemm: block shape %d has no LayerNorm-fold epilogue", shape);
    if (shape == 20) return launch_l<EPI, 4, 2, 3, 4, 2, 4>(p, splitk, stream);   // 128 x 96, 8 compute + 4 loader waves
#ifdef GTAV_EXPERIMENTS
    // Round 6, measured SLOWER, experiments build only (profiles/round6/weight_operand_straight_into_registers_shapes_22_18.txt): the captured batch-1 step 2.05 -> 2.26 ms
    // (shape 22) / 2.12 -> 2.23 ms (shape 18), bit-identical results.  More weight bytes in flight per CU (128 KiB in registers against 48 KiB of the LDS ring) buy
    // nothing at M = 720: the K-step of the loader-wave kernel is not waiting for the weight.
    if constexpr (!FOLDISH) {
        // the same tile, the weight operand straight into registers 4 K-steps ahead (its K loop is written for K-steps per slice that are a multiple of 4: else shape 20)
        if (shape == 22) {
            if ((p.K / TK / (splitk > 0 ? splitk : 1)) % 4 == 0) return launch_lw<EPI, 4, 2, 3, 4, 2, 4, 4>(p, splitk, stream);
            return launch_l<EPI, 4, 2, 3, 4, 2, 4>(p, splitk, stream);
        }
        // ... and with the eight compute waves side by side along the features (16 features x all 96 tokens each): every W fragment is loaded by exactly one wave,
        // 8 K-steps ahead
        if (shape == 18) {
            if ((p.K / TK / (splitk > 0 ? splitk : 1)) % 8 == 0) return launch_lw<EPI, 4, 1, 6, 8, 1, 4, 8>(p, splitk, stream);
            return launch_l<EPI, 4, 2, 3, 4, 2, 4>(p, splitk, stream);
        }
    }
#else
    GTAV_REQUIRE(shape != 22 && shape != 18, "gemm: block shape %d (weight operand straight into registers) exists only in the experiments build (csrc/build.sh exp)", shape);
#endif
    if (shape == 24) return launch_l<EPI, 4, 2, 1, 2, 3, 2>(p, splitk, stream);   // 64 x 48, 6 compute + 2 loader waves (the skinny shape 11 on the loader-wave kernel)
    if (shape == 26) return launch_l<EPI, 4, 2, 2, 2, 3, 2>(p, splitk, stream);   // 64 x 96, 6 compute + 2 loader waves (shape 14 likewise)
#ifdef GTAV_EXPERIMENTS
    {   // shapes that exist only in the experiments build (8, 9, 16, 21, 23, 25, 27, 28, 30, 32, 33, 40, 41, 42): gemm_experiments.inc
        bool handled = false;
        const int rc_ = launch_experiment_shape<EPI>(p, ns, shape, splitk, stream, handled);
        if (handled) return rc_;
    }
#else
    GTAV_REQUIRE(shape != 8 && shape != 9 && shape != 16 && shape != 21 && shape != 23 && shape != 25 && shape != 27 && shape != 28 && shape != 30 && shape != 32 &&
                 shape != 33 && shape != 40 && shape != 41 && shape != 42, "gemm: block shape %d exists only in the experiments build (csrc/build.sh exp)", shape);
#endif
    GTAV_REQUIRE(!p.out2 || (EPI == EPI_F16_TILED && shape != 31), "gemm: a second output image (out2) exists for EPI_F16_TILED on the staged epilogue only (epilogue %d, block shape %d)", (int)EPI, shape);
    if (shape == 31) {   // persistent loader-wave kernel: 128 x 192 tiles, 3-stage ring
        if constexpr (EPI == EPI_GELU_TANH || EPI == EPI_PARTIAL || EPI == EPI_F16_TILED || EPI == EPI_RESID) {
            GTAV_REQUIRE(splitk == 1 && p.N % 8 == 0, "gemm: the persistent loader-wave kernel runs the whole K in one slice (N %% 8 == 0)");
            return launch_lp<EPI, 3>(p, stream);   // (30 / 32 / 33 — 4-stage ring, 256 x 128 / 128 x 256 tiles — were handled above in the experiments build)
        } else {
            GTAV_REQUIRE(false, "gemm: the persistent loader-wave kernel (shape %d) has no epilogue %d", shape, (int)EPI);
        }
    }
    if constexpr (!FOLDISH) {
    // 128 x 144 (nine 16-token groups: one 144-token frame per row tile), 12 compute waves of 32 x 48 + 4 loader waves (round 4): M = 1152, the context-cached
    // step at batch 8, is 8 x 32 = 256 tiles for fc1 / 4-slice fc2 where the 128 x 128 grid has 288 (1.1 rounds)
    if (shape == 29) return launch_l<EPI, 4, 2, 3, 4, 3, 4>(p, splitk, stream);
    }
    // (round 5: a 128 x 144 two-blocks-per-CU tile — shape 12 with three token groups, 6 waves — to turn to_qkv's 720 tiles at M = 5760 (2 rounds at 70 %) into
    // 960 (94 %): correct, race-clean and 40 % SLOWER at every M from 2880 to 11 520 (66 against 47 us; six-wave blocks do not pair up on the SIMDs):
    // profiles/round5/gemm_128x144_two_blocks_per_cu.txt; a 128 x 160 tile of 4 waves of 64 x 80 (864 tiles): 61.5 against 56-58 us on another box; both removed again)
    // (a 128 x 128 loader-wave tile — 8 compute waves of 32 x 64 — was 12 % faster than shape 3 back to back and equal in the training step's
    // weight-gradient GEMMs, which are bound by the fabric traffic of their 118 MB of operands: not kept)
    // (Round 2 also measured one-block-per-CU large tiles on mainloop_g's half-K-step pipeline — 256 x 192, 192 x 192 and
    // 256 x 144 with 8 / 6 waves of 128 x 48 / 96 x 48 — and the 256 x 128 loader-wave tile, shape 21: all correct, all 5-30 %
    // SLOWER than the two-blocks-per-CU 128 x 192 tile at M = 5760 / 11 520, profiles/round2/gemm_large_tile_*.txt: without a
    // co-resident block the prologue and epilogue of every tile are exposed.  The 1-block shapes 30-32 were removed again.)
    if (shape == 14) {         // 64 features x 96 tokens, 6 waves: a few hundred tokens (M = 288-320)
        set_gn(96, 64);
        const dim3 grid(cdiv(p.M, 96) * cdiv(p.N, 64) * splitk);
        GEMM_LAUNCH((gemm_g_kernel<EPI, 4, 2, 2, 3>), grid, dim3(384));
    } else if (shape == 13) {  // 128 features x 96 tokens, 4 waves of 64 x 48, two-stage ring, two blocks per CU: long-K N = 1024 GEMMs at a few thousand
                               // tokens in ONE K slice (480 tiles at M = 5760 fill the 512 block slots; 128 x 128 has 360, 128 x 192 only 240)
        set_gn(96, 128);
        const dim3 grid(cdiv(p.M, 96) * cdiv(p.N, 128) * splitk);
        GEMM_LAUNCH((gemm_g_kernel<EPI, 2, 4, 3, 2>), grid, dim3(256));
    } else if (shape == 12) {  // 128 features x 192 tokens, 8 waves of 64 x 48, two blocks per CU (4 waves per SIMD)
        set_gn(192, 128);
        const dim3 grid(cdiv(p.M, 192) * cdiv(p.N, 128) * splitk);
        GEMM_LAUNCH((gemm_g_kernel<EPI, 2, 4, 3, 4>), grid, dim3(512));
    } else if (shape == 11) {  // 64 features x 48 tokens, 6 waves: skinny M (context-cached sampling, M = 144)
        set_gn(48, 64);
        const dim3 grid(cdiv(p.M, 48) * cdiv(p.N, 64) * splitk);
        GEMM_LAUNCH((gemm_g_kernel<EPI, 4, 2, 1, 3>), grid, dim3(384));   // 6 stages measured 6-8 % slower
    } else if (shape == 7 || shape == 17) {   // 17 (experiments build only): shape 7 with the two wave groups of a block in antiphase (mainloop256_pp)
#ifndef GTAV_EXPERIMENTS
        // round 6: correct, race-screened (profiles/round6/race_screen_shape17.txt) and within +-1 % of shape 7 on the shapes that select it (fc2 / projection at M >= 11 520,
        // fc1 -7 % at M = 11 520 where shape 12 is still ahead of both at the model level): the K-tile is bound by how many fill bytes a CU keeps in flight, not by
        // how its two waves per SIMD interleave reads and MFMAs (profiles/round6/antiphase_256x256_main_loop_shape17_vs_shape7_vs_heuristic.txt: main loop 26.9 vs 27.2 us)
        GTAV_REQUIRE(shape != 17, "gemm: block shape 17 (antiphase 256 x 256 main loop) exists only in the experiments build (csrc/build.sh exp)");
#endif
#ifndef GTAV_EXPERIMENTS
        if constexpr (EPI == EPI_QKV) {   // 15 spilled registers and never selected by the heuristic
            GTAV_REQUIRE(false, "gemm: the 256 x 256 tile has no QKV epilogue in the product build");
        } else
#endif
        if constexpr (!FOLDISH) {
            set_gn(256, 256);
            // A narrow output with far more activations than weights (the VAE's fc2 at M = 46 080: X 377 MB, W 8 MB): keep ALL n-tiles of an m-tile on one XCD,
            // beyond choose_gn's 2 MiB cap — X then crosses the fabric once and W, which stays in the Infinity Cache, is what gets re-read: 1.94 GB -> less per
            // launch by the PMC counters (profiles/traffic.json vae_fc2), 426 -> 408 us (profiles/round5/gemm_tile_group_width_large_M.txt)
            if (!g_force_gn && splitk == 1 && p.N <= 1024 && (size_t)p.M >= 8 * (size_t)p.N) p.tm.gn = cdiv(p.N, 256);
            const dim3 grid(cdiv(p.M, 256) * cdiv(p.N, 256) * splitk);
#ifdef GTAV_EXPERIMENTS
            if (shape == 17) GEMM_LAUNCH((gemm256_kernel<EPI, true>), grid, dim3(512));
            else
#endif
            GEMM_LAUNCH((gemm256_kernel<EPI>), grid, dim3(512));
        }
    } else if (shape == 3) {
        set_gn(128, TN);
        const dim3 grid(cdiv(p.M, 128) * cdiv(p.N, TN) * splitk);
        if (ns <= 2) GEMM_LAUNCH((gemm_kernel<EPI, 2, 4, 2>), grid, dim3(512));
        else GEMM_LAUNCH((gemm_kernel<EPI, 4, 4, 2>), grid, dim3(512));
    } else {
        GTAV_REQUIRE(shape == 2, "gemm: unknown block shape %d", shape);
        set_gn(128, TN);
        const dim3 grid(cdiv(p.M, 128) * cdiv(p.N, TN) * splitk);
        if (ns <= 2) GEMM_LAUNCH((gemm_kernel<EPI, 2, 2, 4>), grid, dim3(256));
        else GEMM_LAUNCH((gemm_kernel<EPI, 4, 2, 4>), grid, dim3(256));
    }
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

// Fused temporal QKV GEMM + attention (gemm_qkvt_attn_kernel).  p: X = the LayerNorm output in tperm order, W = head-major to_qkv
// weight, M tokens = B * 5 * S, N = 3 D, K = D, D, S (positions per frame, % 16 == 0), Tq = 5, t0 = 0, Tmax, rope_cs, k = the
// temporal K/V cache, out = the attention output (f16, tile-major, logical row length D).
static int fill_tile_map(GemmParams::TileMap& tm, int M, int N, int K, int tmb, int tnb, int splitk);
bool gemm_qkvt_attn_ok(int M, int D, int S, int Tq, int t0) {
    static const int max_blocks = GTAV_ENV_INT("GTAV_QKVT_MAX_BLOCKS", 256);   // experiments build: A/B of the fused kernel on larger grids (batch 8: 1152 blocks)
    return Tq == 5 && t0 == 0 && S % 16 == 0 && D % 64 == 0 && M % (Tq * S) == 0 && (M / 80) * (D / 64) <= max_blocks;
}
int launch_gemm_qkvt_attn(const GemmParams& p_in, hipStream_t stream) {
    GemmParams q = p_in;
    GTAV_REQUIRE(gemm_qkvt_attn_ok(q.M, q.D, q.S, q.Tq, q.t0), "gemm/qkvt_attn: unsupported geometry M=%d D=%d S=%d Tq=%d t0=%d", q.M, q.D, q.S, q.Tq, q.t0);
    GTAV_REQUIRE(q.N == 3 * q.D && q.K % TK == 0 && q.k && q.out && q.rope_cs && q.Tmax >= q.Tq, "gemm/qkvt_attn: missing buffers");
    GTAV_REQUIRE(((uintptr_t)q.X & 15) == 0 && ((uintptr_t)q.W & 15) == 0, "gemm: operands must be 16-byte aligned");
    constexpr int NS = 4, LDS = NS * 34 * 1024;
    static unsigned long long attr_devs = 0;
    int dev = 0;
    GTAV_REQUIRE(device_cus(&dev) > 0, "gemm: no current device");
    if (!(attr_devs >> (dev & 63) & 1)) {
        GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_qkvt_attn_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_devs |= 1ull << (dev & 63);
    }
#ifdef GTAV_EXPERIMENTS
    q.debug = g_debug & (3 | 32 | 512);
    q.stamps = g_stamps;
#endif
    q.out_sc1 = 1;
    q.splitk = 1;
    if (int rc_ = fill_tile_map(q.tm, q.M, q.N, q.K, 80, 192, 1)) return rc_;
    GTAV_LAUNCH((gemm_qkvt_attn_kernel<NS>), dim3(q.tm.tiles), dim3(512), LDS, stream, q);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

// Fused spatial QKV GEMM + attention (gemm_qkvs_attn_kernel).  p: X = the LayerNorm output (rows in (b, frame, position) order), W = the to_qkv weight in
// the wave-interleaved head-major row order (launch_qkv_head_major mode 1), M tokens = frames * 144, N = 3 D, K = D, D, S = 144, rope_cs = the spatial table
// [144][64], out = the attention output (f16, tile-major, logical row length D).
bool gemm_qkvs_attn_ok(int M, int D, int S) {
    static const int max_blocks = GTAV_ENV_INT("GTAV_QKVS_MAX_BLOCKS", 65535);   // experiments build: A/B against the split path per grid size (65535: the tile map's limit)
    return S == 144 && D % 64 == 0 && M > 0 && M % S == 0 && (long long)(M / S) * (D / 64) <= max_blocks;
}
int launch_gemm_qkvs_attn(const GemmParams& p_in, hipStream_t stream) {
    GemmParams q = p_in;
    GTAV_REQUIRE(gemm_qkvs_attn_ok(q.M, q.D, q.S), "gemm/qkvs_attn: unsupported geometry M=%d D=%d S=%d", q.M, q.D, q.S);
    GTAV_REQUIRE(q.N == 3 * q.D && q.K % TK == 0 && q.out && q.rope_cs && !q.rope_cs_q && !q.bias, "gemm/qkvs_attn: missing buffers (or a bias / a separate q table)");
    GTAV_REQUIRE(((uintptr_t)q.X & 15) == 0 && ((uintptr_t)q.W & 15) == 0, "gemm: operands must be 16-byte aligned");
    constexpr int NS = 3, LDS = NS * 44 * 1024;
    static unsigned long long attr_devs = 0;
    int dev = 0;
    GTAV_REQUIRE(device_cus(&dev) > 0, "gemm: no current device");
    if (!(attr_devs >> (dev & 63) & 1)) {
        GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_qkvs_attn_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_devs |= 1ull << (dev & 63);
    }
#ifdef GTAV_EXPERIMENTS
    q.debug = g_debug & (3 | 32 | 64);
    q.stamps = g_stamps;
#endif
    q.splitk = 1;
    q.qkv_mode = QKV_SPATIAL;
    if (int rc_ = fill_tile_map(q.tm, q.M, q.N, q.K, 144, 192, 1)) return rc_;
    // (Two blocks per (frame, head) on grids of at most half the chip — both run the whole projection, each walks half of the query tiles in one round of its eight
    // waves — was measured and is not kept: 19.2 -> 19.4 us at 80 blocks, 19.8 -> 20.9 at 128.)
    GTAV_LAUNCH((gemm_qkvs_attn_kernel<NS>), dim3(q.tm.tiles), dim3(768), LDS, stream, q);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

// out[m][n] += sum_t X[t][m] W[t][n] (gemm_tn_kernel): X tile-major [tokens][M features], W tile-major [tokens][N features], K = tokens.
bool gemm_tn_ok(int M, int N, int K) { return M > 0 && N > 0 && K > 0 && M % 128 == 0 && N % 128 == 0 && K % 64 == 0; }
// at least 128 tiles of 128 x 128: smaller grids stay on transposes + the NT kernels, which pick smaller tiles for them
static int g_tn_enable = GTAV_ENV_INT("GTAV_GEMM_TN", 0);   // experiments build: 1 = use it for the weight gradients (A/B runs)
bool gemm_tn_pays(int M, int N, int K) { return g_tn_enable && gemm_tn_ok(M, N, K) && (M / 128) * (N / 128) >= 128; }
int launch_gemm_tn(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    GTAV_REQUIRE(gemm_tn_ok(p.M, p.N, p.K), "gemm_tn: M=%d, N=%d must be multiples of 128 and K=%d of 64", p.M, p.N, p.K);
    GTAV_REQUIRE(p.out && p.ldo >= p.N && p.ldo % 4 == 0, "gemm_tn: bad output ldo=%d", p.ldo);
    GTAV_REQUIRE(((uintptr_t)p.X & 15) == 0 && ((uintptr_t)p.W & 15) == 0, "gemm_tn: operands must be 16-byte aligned");
#ifdef GTAV_EXPERIMENTS
    p.debug = 0;
    p.stamps = g_stamps;
#endif
    p.splitk = 1;
    p.bias = nullptr; p.gate = nullptr;
    p.tm.gn = choose_gn(p.M, p.N, p.K, 128, 128, 1);
    const dim3 grid((p.M / 128) * (p.N / 128));
    GTAV_LAUNCH((gemm_tn_kernel<4>), grid, dim3(512), 0, stream, p);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

bool gemm_dw_grouped_ok(const GemmDwGroup* g, int n, int K) {
    if (!g || n < 1 || n > GEMM_DW_MAX_GROUPS || K <= 0 || K % TK) return false;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (g[i].M <= 0 || g[i].N <= 0 || g[i].M % 256 || g[i].N % 256 || g[i].ldo < g[i].N || g[i].ldo % 4) return false;
        tiles += (g[i].M >> 8) * (g[i].N >> 8);
    }
    // worth it from half a round of 256 x 256 tiles on (below that the 128 x 128 grids of the single launches cover more CUs) up to
    // two rounds; K long enough that the one-tile-per-CU prologue / epilogue do not matter
    int dev = 0;
    const int cus = device_cus(&dev);
    return cus > 0 && 2 * tiles >= cus && tiles <= 2 * cus && K >= 2048;
}

int launch_gemm_dw_grouped(const GemmDwGroup* g, int n, int K, int* err_flag, hipStream_t stream, bool tn) {
    GTAV_REQUIRE(g && n >= 1 && n <= GEMM_DW_MAX_GROUPS && K > 0 && K % TK == 0, "gemm_dw_grouped: bad arguments (n=%d K=%d)", n, K);
    GemmDwGroups gs;
    memset(&gs, 0, sizeof(gs));
    gs.n = n;
    for (int i = 0; i < n; ++i) {
        GTAV_REQUIRE(g[i].X && g[i].W && g[i].out && g[i].M > 0 && g[i].N > 0 && g[i].M % 256 == 0 && g[i].N % 256 == 0 && g[i].ldo >= g[i].N && g[i].ldo % 4 == 0,
                     "gemm_dw_grouped: group %d: M=%d, N=%d must be multiples of 256, ldo=%d >= N", i, g[i].M, g[i].N, g[i].ldo);
        GTAV_REQUIRE(((uintptr_t)g[i].X & 15) == 0 && ((uintptr_t)g[i].W & 15) == 0 && ((uintptr_t)g[i].out & 15) == 0, "gemm_dw_grouped: group %d: operands must be 16-byte aligned", i);
        gs.g[i] = g[i];
        gs.first[i + 1] = gs.first[i] + (g[i].M >> 8) * (g[i].N >> 8);
    }
    for (int i = n + 1; i <= GEMM_DW_MAX_GROUPS; ++i) gs.first[i] = gs.first[n];
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.K = K; p.splitk = 1; p.err_flag = err_flag; p.rows_per_gate = 1;
#ifdef GTAV_EXPERIMENTS
    p.stamps = g_stamps;
#endif
    if (tn) {   // operands tile-major [tokens][features]: K tokens in whole 128-row tiles (the pad rows of a ragged last tile are not zero)
        GTAV_REQUIRE(K % 128 == 0, "gemm_dw_grouped: the transpose-free form contracts over whole 128-token row tiles (K=%d)", K);
        GTAV_LAUNCH(gemm256_dw_grouped_kernel<true>, dim3(gs.first[n]), dim3(512), 0, stream, p, gs);
    } else {
        GTAV_LAUNCH(gemm256_dw_grouped_kernel<false>, dim3(gs.first[n]), dim3(512), 0, stream, p, gs);
    }
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_grouped(const GemmGroup* groups_dev, int n_groups, int max_N, int M, int K, hipStream_t stream) {
    GTAV_REQUIRE(groups_dev && n_groups > 0 && n_groups < 65536 && M > 0 && max_N > 0 && K > 0 && K % TK == 0, "gemm_grouped: bad arguments");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.N = max_N; p.K = K; p.splitk = 1;
    const dim3 grid(cdiv(M, 96) * cdiv(max_N, 128), n_groups);
    hipLaunchKernelGGL((gemm_grouped_kernel<4, 4, 2, 3>), grid, dim3(384), 0, stream, p, groups_dev);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}


int launch_gemm(const GemmParams& p_in, int epi_x, hipStream_t stream) {
    GemmParams p = p_in;
    const int epi = epi_base(epi_x);
    const bool fold_c = epi_is_fold_consumer(epi_x), fold_p = epi_x == EPI_RESID_FOLD;
    if (fold_c || fold_p) {
        GTAV_REQUIRE(p.f_P > 0 && p.f_P % 16 == 0 && p.M % p.f_P == 0, "gemm/fold: %d tokens per frame must be a multiple of 16 and divide M=%d", p.f_P, p.M);
        if (fold_c) {
            GTAV_REQUIRE(p.f_stats && p.f_c1 && p.f_c2 && p.f_ldc >= p.N && p.f_nslot == p.K / 64 && p.f_nslot % 4 == 0,
                         "gemm/fold: consumer needs the row statistics (K/64 = %d slots, a multiple of 4) and the c1 / c2 tables", p.K / 64);
            GTAV_REQUIRE(!p.bias, "gemm/fold: the bias of a fold consumer lives in its c2 table");
            if (epi == EPI_QKV) GTAV_REQUIRE(p.qkv_mode == QKV_TEMPORAL || p.S % 16 == 0, "gemm/fold: spatial QKV needs S %% 16 == 0");
        } else {
            GTAV_REQUIRE(p.gate && p.f_scale && p.f_stats_out && p.f_a && p.N % 64 == 0 && p.ldo >= p.N,
                         "gemm/fold: producer needs gate, next scale, statistics and operand buffers, N %% 64 == 0");
        }
    }
    p.debug = g_debug & (3 | 16 | 32 | 2048 | 8192 | 16384 | 0x1F0000);   // 0x10000.. : pieces of the fold producer epilogue off (timing, experiments build)   // bit 4: direct (unstaged) QKV epilogue; bit 5 (experiments): per-K-step stamps of the loader-wave kernels
    p.stamps = nullptr;
#ifdef GTAV_EXPERIMENTS
    p.stamps = g_stamps;            // the tool sizes the buffer for the largest grid it launches (g_stamp_blocks)
#endif
    // The 16-byte output stores of the split-K slabs, the tile-major GELU output and the staged QKV epilogue go out
    // write-through-and-drop (sc1): nothing is left dirty in L2 for the end-of-kernel write-back, and the output does not
    // evict operand tiles.  Measured in situ at B = 1 (every GEMM class 3-9 % shorter, consumers unchanged: 3.94 -> 4.10
    // frames/s) and back-to-back (profiles/round1/v13_gemm_sc1_stores_microbench.txt).  8-byte sc1 stores are SLOWER (one
    // fabric write each), so the unstaged epilogues keep plain stores.  Debug bit 3 turns it off (experiments).
    p.out_sc1 = (g_debug & 8) ? 0 : 1;
    GTAV_REQUIRE(prefetch_desc_ok(p.pf), "gemm: bad prefetch descriptor (row tiles %d, K tiles %d, K slices %d)", p.pf.rt, p.pf.nkt, p.pf.splitk);
    if (g_debug & 0x800000) p.pf.next = nullptr;   // experiments build: A/B
    GTAV_REQUIRE(p.K > 0 && p.K % TK == 0, "gemm: K=%d must be a positive multiple of %d", p.K, TK);
    GTAV_REQUIRE(p.M > 0 && p.N > 0 && p.N % 4 == 0, "gemm: bad M=%d N=%d", p.M, p.N);
    GTAV_REQUIRE(((uintptr_t)p.X & 15) == 0 && ((uintptr_t)p.W & 15) == 0, "gemm: operands must be 16-byte aligned");
    int splitk = 1;
    if (epi == EPI_QKV) {
        GTAV_REQUIRE(p.N == 3 * p.D && p.D % 128 == 0, "gemm/qkv: N=%d must equal 3*D, D=%d %% 128 == 0", p.N, p.D);
        GTAV_REQUIRE(p.S > 0 && p.q && p.k && p.rope_cs, "gemm/qkv: missing buffers");
        if (p.qkv_mode == QKV_SPATIAL) {
            GTAV_REQUIRE(p.S % 4 == 0 && p.M % p.S == 0 && p.v, "gemm/qkv spatial: S=%d must divide M=%d, S %% 4 == 0", p.S, p.M);
        } else {
            GTAV_REQUIRE(p.Tq > 0 && p.M % (p.S * p.Tq) == 0 && p.t0 + p.Tq <= p.Tmax, "gemm/qkv temporal: bad frame geometry");
        }
    } else {
        GTAV_REQUIRE(p.out && p.ldo >= p.N && p.ldo % 4 == 0, "gemm: bad output ldo=%d", p.ldo);
        if (epi == EPI_RESID && p.gate) GTAV_REQUIRE(p.rows_per_gate > 0, "gemm/resid: rows_per_gate");
        if (epi == EPI_GELU_TANH || epi == EPI_GELU_ERF || epi == EPI_F16_TILED) GTAV_REQUIRE(p.ldo % 64 == 0, "gemm/gelu: tile-major output needs ldo %% 64 == 0");
        if (fold_p) GTAV_REQUIRE(p.splitk <= 1, "gemm/fold: the producer epilogue needs the complete sum (no split-K)");
        if (epi == EPI_PARTIAL) {
            splitk = p.splitk;
            GTAV_REQUIRE(splitk >= 1 && (p.K / TK) % splitk == 0, "gemm/partial: splitk=%d must divide K/64=%d", splitk, p.K / TK);
        }
    }
    // Block shape (measured, profiles/round1/v6_gemm_shapes_microbench.txt):
    //   grid < 256 blocks of 128 x 128 (one block per CU at most): 8 waves per block (two staggered waves per SIMD),
    //     4-stage ring — 17-25 % faster than 4 waves at M = 720;
    //   larger grids: 4 waves, 2-stage ring, two co-resident blocks per CU; the 128 x 256 / 8-wave tile (shape 4) ties it.
    const int blocks128 = cdiv(p.M, 128) * cdiv(p.N, TN) * splitk;
    const bool foldish = fold_c || fold_p;
    int wm = g_force_wm ? g_force_wm : (blocks128 <= 256 ? 3 : 2);
    if (!g_force_wm && blocks128 <= 256) {
        // small M: every block is bound by its own L2->LDS fill (~65 GB/s per CU), i.e. by (TN + TM) x K bytes, and the grid
        // by how many of the 256 CUs it covers.  Candidates: 128 x 128 (shape 3), 128 x 96 (shape 9), 96 x 96 (shape 8).
        auto cost = [&](int tn, int tm) { return cdiv(cdiv(p.M, tm) * cdiv(p.N, tn) * splitk, 256) * (tn + tm); };
        int best = cost(128, 128);
        if (cost(128, 96) < best) best = cost(128, 96), wm = 9;
        const bool ok96 = p.N % 96 == 0 && epi != EPI_GELU_TANH && epi != EPI_GELU_ERF && epi != EPI_F16_TILED && !(epi == EPI_QKV && p.qkv_mode == QKV_SPATIAL);
        if (ok96 && cost(96, 96) < best) best = cost(96, 96), wm = 8;
        // skinny M (M = 144: the context-cached sampler step): 64 x 48 tiles put 144-192 blocks of 224 KB where the
        // 128 x 96 grid has 48-64 blocks of 448 KB
        if (cost(64, 96) < best) best = cost(64, 96), wm = 14;   // M = 288-320 (g256 window step, batch-2 cached step)
        if (cost(64, 48) < best) best = cost(64, 48), wm = 11;
    }
    // Round 2: the 128 x 96 tile runs on the loader-wave kernel (shape 20: 4 loader + 8 compute waves, fills and MFMAs overlap by
    // construction): QKV 14.1 -> 10.7 us, fc1 13.3 -> 10.9, fc2 12.0 -> 10.2, out-proj 6.6 -> 6.1 at M = 720 (profiles/round2).
    if (!g_force_wm && !(g_debug & 64) && (wm == 9 || (wm == 8 && g_l_for_8))) wm = (g_debug & 128) ? 25 : 20;   // debug bit 6 (experiments build): round-1 shapes, for A/B runs in one process
#ifdef GTAV_EXPERIMENTS
    // round 6 A/B (debug bit 24): the 128 x 96 loader-wave tile with the weight operand straight into registers (shape 22) wherever the heuristic picks shape 20
    if (!g_force_wm && wm == 20 && (g_debug & 0x1000000) && !foldish && (p.K / TK / (splitk > 0 ? splitk : 1)) % 4 == 0) wm = 22;
    if (!g_force_wm && wm == 20 && (g_debug & 0x2000000) && !foldish && (p.K / TK / (splitk > 0 ? splitk : 1)) % 8 == 0) wm = 18;   // debug bit 25: shape 18
#endif
    // Round 3: the skinny tiles on the loader-wave kernel too (6 compute + 2 loader waves): M = 144 QKV 8.45 -> 7.34 us, fc1 7.75 -> 6.83, out-proj 5.00 -> 4.68;
    // M = 288 QKV 10.3 -> 8.4, fc1 9.7 -> 8.4 (profiles/round3/skinny_loader_wave_shapes_M144_M288.txt); their compute waves carry the next-weight L2 prefetch.
    // The 8-slice fc2 at M = 288 stays on the all-waves-fill kernel (7.4 against 8.3 us).
    if (!g_force_wm && !(g_debug & 64)) {
        if (wm == 11) wm = 24;
        else if (wm == 14 && !(epi_x == EPI_PARTIAL && splitk >= 8)) wm = 26;
    }
    // Round 4: a little over a thousand tokens (M = 1152: the context-cached step at batch 8) sits between the small-M tiles (128 x 96: 12 row tiles, 1.1-1.5
    // rounds of one-block-per-CU tiles) and the large-M ones (128 x 128, two blocks per CU: 216-288 blocks on 512 slots).  128 x 144 tiles on the loader-wave
    // kernel (shape 29) are one round there: to_qkv 192, fc1 256, four-slice fc2 256 tiles (profiles/round4/step_B8_cached_*.json: to_qkv 14.9, fc1 18.5, fc2 19.7 us before).
    bool frame_tile = false;
    if (!g_force_wm && !foldish && !(g_debug & 64)) {
        const int t29 = cdiv(p.M, 144) * cdiv(p.N, 128) * splitk, t20 = cdiv(p.M, 96) * cdiv(p.N, 128) * splitk;
        if (t29 > 128 && t29 <= 256 && t20 > 256 && blocks128 <= 320) wm = 29, frame_tile = true;
    }
    if (!g_force_wm && wm == 2 && cdiv(p.M, 192) * cdiv(p.N, 128) * splitk >= 320) {
        // large M: 128 x 192 tiles (4 waves of 64 x 96, still two blocks per CU) move 17 % fewer fill bytes per FLOP than
        // 128 x 128: QKV 62.6 -> 55.5 us, fc1 63.8 -> 61.1 us at M = 5760 (profiles/round1/v17_gemm_128x192_microbench.txt);
        // below ~320 tiles the 512 block slots are too unevenly filled (fc2 at M = 5760: 240 tiles, 65.7 -> 73.1 us).
        wm = 12;   // 8 waves of 64 x 48 (four per SIMD with the co-resident block): QKV 60.3 -> 54.8, fc1 62.1 -> 59.4 vs the 4-wave form (shape 10)
    }
    // Narrow outputs (the N = 1024 residual GEMMs) at a few thousand tokens: both two-blocks-per-CU tiles, 128 x 192 (shape 12) and 128 x 96
    // (shape 13), run a grid that FILLS the 512 block slots once faster than 128 x 128 runs 360-720 blocks: out-proj at M = 5760 20.6 -> 16.8 us
    // (shape 13: 480 tiles), at M = 11 520 33.4 -> 30.8 us and fc2 97.6 -> 90.2 us (shape 12: 480 tiles; the 256 x 256 tile lost to it),
    // M = 2880 out-proj 13.7 -> 11.4 us, fc2 34.6 -> 27.9 us (shape 13, two K slices) — profiles/round3/gemm_shapes_*.txt.  The LayerNorm-fold
    // producer takes the larger tile whenever its grid is more than half full (27.2 us against 33.5 us at M = 5760).
    bool narrow_pick = false;
    if (!g_force_wm && !frame_tile && blocks128 > 256 && p.N <= 2048 && epi != EPI_QKV) {
        const int t12 = cdiv(p.M, 192) * cdiv(p.N, 128) * splitk, t13 = cdiv(p.M, 96) * cdiv(p.N, 128) * splitk;
        if (t12 > 256 && t12 <= 512) wm = 12, narrow_pick = true;
        else if (t13 > 256 && t13 <= 512 && !fold_p) wm = 13, narrow_pick = true;
        else if (fold_p && t12 > 128) wm = 12, narrow_pick = true;
    }
    if (!g_force_wm && splitk == 1 && ((epi_x == EPI_PARTIAL && lp_takes(p.M, p.N, p.K)) || (epi_x == EPI_RESID && gemm_resid_inplace_ok(p.M, p.N, p.K, p.gate ? p.rows_per_gate : 0)))) wm = 31, narrow_pick = true;
    // Tens of thousands of tokens (the VAE encode of a training batch: M = 46 080): the in-place residual GEMMs on whole rounds of 256 x 256 tiles
    // (>= 2 rounds, >= 90 % full) beat the persistent 128 x 192 kernel — fc2 453 -> 399 us, attention projection 182 -> 164 us back to back
    // (profiles/round5/gemm_vae_shapes_M23040_M46080.txt); at M = 23 040 (1.4 rounds) they lose (240 against 223 us) and shape 31 stays
    if (!g_force_wm && wm == 31 && epi_x == EPI_RESID && p.N % 256 == 0 && p.N <= 1024 && p.K >= 1024) {
        const int t256 = cdiv(p.M, 256) * (p.N / 256), rounds = cdiv(t256, 256);
        if (rounds >= 2 && t256 * 10 >= rounds * 256 * 9) wm = 7;
    }
#ifdef GTAV_EXPERIMENTS
    if (!g_force_wm && epi_x == EPI_GELU_TANH && (g_debug & 0x600000) && cdiv(p.M, 128) * cdiv(p.N, 256) >= 512) wm = (g_debug & 0x200000) ? 33 : 32;   // A/B of the persistent 128 x 256 / 256 x 128 tiles for fc1 (debug bits 21 / 22)
#endif
    // large M: the persistent ping-pong kernel (epilogue and next tile's prologue under the other wave group's MFMAs)
    if (!g_force_wm && splitk == 1 && gemm_pp_ok(p.M, p.N, p.K, epi) && !(epi == EPI_QKV && p.qkv_mode == QKV_SPATIAL && p.S % 8 != 0)) wm = 16;
    GTAV_REQUIRE(!(wm == 8 && epi == EPI_QKV && p.qkv_mode == QKV_SPATIAL), "gemm: 96-feature tiles straddle the K / V boundary (spatial QKV)");
    // 256 x 256 tiles (shape 7) halve the fill bytes per FLOP but run one block per CU, so a tile's epilogue (a 32 MB
    // store burst per round of 256 tiles) is not hidden by a co-resident block: they win only where the K loop is long
    // relative to the output and the grid is one well-filled round — the N = 1024 residual GEMMs at M >= 11 520
    // (profiles/round1/v12_gemm_256tile_microbench.txt: fc2 129 -> 98 us, out-proj 40 -> 35 us).
    if (!g_force_wm && splitk == 1 && epi != EPI_QKV && !foldish && !narrow_pick && p.N % 256 == 0 && p.N <= 1024 && p.K >= 1024) {
        const int t256 = cdiv(p.M, 256) * (p.N / 256), rounds = cdiv(t256, 256);
        if (t256 * 10 >= rounds * 256 * 7) wm = 7;
    }
    int ns = g_force_stages ? g_force_stages : ((wm == 12 || wm == 13) ? 2 : wm >= 8 ? 4 : wm == 3 ? 4 : 2);   // (shapes 20+ carry their ring depth in the template)
    // a wave's token span (16 FJ) must not exceed a frame: the skinny 64 x 48 / 64 x 96 tiles span 16 / 32 tokens, the others 48-64
    if (foldish) {
        const int span = wm == 2 ? 64 : (wm == 3 || wm == 14 || wm == 26) ? 32 : (wm == 11 || wm == 24) ? 16 : 48;   // (12, 13, 20: 48)
        GTAV_REQUIRE(p.f_P >= span, "gemm/fold: frames of %d tokens are shorter than the token span (%d) of a wave of block shape %d", p.f_P, span, wm);
    }
    switch (epi_x) {
#ifdef GTAV_EXPERIMENTS   // the LayerNorm fold (round 3: correct, measured slower at every size) is compiled into the experiments build only
        case EPI_RESID_FOLD: return launch_epi<EPI_RESID_FOLD>(p, ns, wm, splitk, stream);
        case EPI_QKV_FOLD: return launch_epi<EPI_QKV_FOLD>(p, ns, wm, splitk, stream);
        case EPI_GELU_TANH_FOLD: return launch_epi<EPI_GELU_TANH_FOLD>(p, ns, wm, splitk, stream);
        case EPI_F32_FOLD: return launch_epi<EPI_F32_FOLD>(p, ns, wm, splitk, stream);
#else
        case EPI_RESID_FOLD: case EPI_QKV_FOLD: case EPI_GELU_TANH_FOLD: case EPI_F32_FOLD:
            GTAV_REQUIRE(false, "gemm: the LayerNorm-fold epilogues exist only in the experiments build (csrc/build.sh exp)");
#endif
        case EPI_F32: return launch_epi<EPI_F32>(p, ns, wm, splitk, stream);
        case EPI_F16: return launch_epi<EPI_F16>(p, ns, wm, splitk, stream);
        case EPI_GELU_TANH: return launch_epi<EPI_GELU_TANH>(p, ns, wm, splitk, stream);
        case EPI_GELU_ERF: return launch_epi<EPI_GELU_ERF>(p, ns, wm, splitk, stream);
        case EPI_RESID: return launch_epi<EPI_RESID>(p, ns, wm, splitk, stream);
        case EPI_QKV: return launch_epi<EPI_QKV>(p, ns, wm, splitk, stream);
        case EPI_PARTIAL: return launch_epi<EPI_PARTIAL>(p, ns, wm, splitk, stream);
        case EPI_F16_TILED: return launch_epi<EPI_F16_TILED>(p, ns, wm, splitk, stream);
        default: GTAV_REQUIRE(false, "gemm: unknown epilogue %d", epi_x);
    }
    return 0;
}

}  // namespace gtav
