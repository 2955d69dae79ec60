// Attention kernels for gfx950.
//
// attn_spatial: non-causal attention over S tokens per (frame, head), head_dim 64, on v_mfma_f32_16x16x32_f16:
//     S^T = K . Q^T          (keys on accumulator ROWS, queries on lanes)
//     softmax                (row max / sum = in-lane + two cross-lane steps)
//     O^T += Vt . P^T        (P^T is taken straight from the S^T accumulators: the k-slot -> key map is applied to the Vt read instead,
//                             so P never moves between lanes or through LDS)
//   attn_spatial_1p_kernel   S <= 160 (the DiT's 144 frame tokens): K / Vt of the head resident in LDS, one pass per query tile
//   attn_flash_kernel        longer sequences (the VAE's 576 tokens): K / Vt streamed through an LDS ring, flash form (round 5)
//
// attn_temporal: causal attention over the <= 8 frames of the sliding window per (b, position, head);
// 25 dot products of length 64 per head: VALU + 16-lane xor-shuffle reductions, everything in registers.
#include "ops.h"
#include "attn_tile.h"

#include <cstdlib>

namespace gtav {

namespace {

constexpr float kScaleLog2e = kAttnQScale;  // 1/sqrt(64) * log2(e)

// SHORT sequences (S_pad = 16 NK <= 160 keys: the DiT's 144 patch tokens per frame; round 4): one block per (frame, head[, q-split]); K [S][64] (rows XOR-swizzled
// for conflict-free ds_read_b128) and Vt [64][S] (rows padded) of the head are staged once in LDS, every wave then walks 16-query tiles in ONE pass: the scores
// of all NK key tiles live in registers (4 NK floats per lane), one row maximum, one exponential per score, then the PV products — no online-softmax rescale and
// no dependency between key blocks (the online-softmax kernel of round 1 walked a query tile as three serial QK^T -> max -> exp -> rescale -> PV rounds: 17.7 us
// per launch at batch 8 against a 9.4 us traffic floor).  softmax(QK^T / 8) V is computed with one fp32 rounding order for every batch size and both sampling
// algorithms (the kernel is chosen by S alone).
template <int NW, int NK>
__global__ __launch_bounds__(64 * NW) void attn_spatial_1p_kernel(const f16* __restrict__ Q, const f16* __restrict__ K,
                                                              const f16* __restrict__ Vt, f16* __restrict__ O, int heads, int S,
                                                              int qsplit, int sc1) {
    constexpr int S_pad = 16 * NK;
    static_assert(NK % 2 == 0 && NK <= 10, "an even number of 16-key tiles, at most 160 keys");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                              // [S_pad][128 B], 16-B chunk c of row r stored at c ^ (r & 7)
    constexpr int vstride = (S_pad + 8) * 2;      // bytes per Vt row
    char* Vs = smem + (size_t)S_pad * 128;        // [64][S_pad + 8] halves
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x;                    // nb * heads + head
    const int nb = bh / heads, head = bh - nb * heads;
    const f16* Kg = K + (size_t)bh * S * 64;
    const f16* Vg = Vt + (size_t)bh * 64 * S;
    const f16* Qg = Q + (size_t)bh * S * 64;
    const int nqt = (S + 15) >> 4;
    constexpr int NT = 64 * NW;
    const int qt_first = blockIdx.y * NW + w;
    f16x8 qpre[2] = {};
    if (qt_first < nqt) {
        int qr = qt_first * 16 + li;
        qr = qr < S ? qr : S - 1;
        qpre[0] = *(const f16x8*)(Qg + (size_t)qr * 64 + 8 * g);
        qpre[1] = *(const f16x8*)(Qg + (size_t)qr * 64 + 32 + 8 * g);
    }
    {   // stage K (swizzled) and Vt (padded), zero the padding: every load of a batch before the first LDS write (one memory round trip)
        constexpr int nk = S_pad * 8, vchunks = (S_pad + 8) / 8, nv = 64 * vchunks, nmax = nk > nv ? nk : nv;
        for (int base = tid; base < nmax; base += NT * 6) {
            uint4 kv[6], vv[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int idx = base + u * NT;
                kv[u] = make_uint4(0, 0, 0, 0);
                if (idx < nk && (idx >> 3) < S) kv[u] = *(const uint4*)(Kg + (size_t)(idx >> 3) * 64 + (idx & 7) * 8);
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int idx = base + u * NT;
                vv[u] = make_uint4(0, 0, 0, 0);
                if (idx < nv) {
                    const int d = idx / vchunks, c = idx - d * vchunks;
                    if (c * 8 < S) vv[u] = *(const uint4*)(Vg + (size_t)d * S + c * 8);  // S % 8 == 0
                }
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int idx = base + u * NT;
                if (idx < nk) {
                    const int r = idx >> 3, c = idx & 7;
                    *(uint4*)(Ks + r * 128 + ((c ^ (r & 7)) << 4)) = kv[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int idx = base + u * NT;
                if (idx < nv) {
                    const int d = idx / vchunks, c = idx - d * vchunks;
                    *(uint4*)(Vs + d * vstride + c * 16) = vv[u];
                }
            }
        }
    }
    __syncthreads();
    const int Dm = heads * 64;
    for (int qt = qt_first; qt < nqt; qt += NW * qsplit) {
        const int q0 = qt * 16;
        f16x8 qf[2];
        if (qt == qt_first) {
            qf[0] = qpre[0];
            qf[1] = qpre[1];
        } else {
            int qr = q0 + li;
            qr = qr < S ? qr : S - 1;
            qf[0] = *(const f16x8*)(Qg + (size_t)qr * 64 + 8 * g);
            qf[1] = *(const f16x8*)(Qg + (size_t)qr * 64 + 32 + 8 * g);
        }
        attn_1p_tile<NK>(Ks, Vs, qf, S, q0, O, nb * S + q0, head * 64, Dm, lane, sc1);   // attn_tile.h: shared with the fused to_qkv + attention GEMM
    }
}

// ---- long sequences (S_pad > 160: the VAE's 576 tokens per frame; round 5) ---------------------------------------------------------------
// The kernel of rounds 1-4 kept ALL of K / Vt of a (frame, head) in LDS (148 KB at S = 576: one 4-wave block per CU, whose 144 KB prologue load nothing hid)
// and walked 16-query tiles one at a time with an online softmax, so every K / Vt fragment read from LDS fed ONE MFMA (64 B/clk/SIMD = the whole LDS bandwidth
// at the MFMA issue rate) and every 64-key block paid the rescale: 0.106-0.117 of the MFMA peak at 40-80 frames x 16 heads
// (profiles/round4/train_step_kernel_stats.csv, profiles/round5/attn_flash_S576_ab_first_version.txt; removed in round 5).  This kernel is the flash form of the same arithmetic:
//   * a block = 4 waves x NQT 16-query tiles (NQT = 3: 192 queries; S = 576 is three blocks per (frame, head), whose ids are 8 apart = one XCD's L2
//     under round-robin placement, speed only); several blocks per CU (48 KiB of LDS, <= 168 registers)
//   * K / Vt arrive in 64-key blocks by LDS-DMA (global_load_lds_dwordx4, 4 pieces per wave and block) into a 3-slot ring: counted vmcnt, ONE barrier
//     per key block, block kb + 2 in flight while block kb is consumed; the bank swizzles are applied to the per-lane SOURCE address (the LDS image
//     of an LDS-DMA is lane-linear): K rows chunk ^ kswz(key), Vt rows chunk ^ ((d >> 1) & 7): every fragment is one conflict-free ds_read_b128
//   * every K / Vt fragment feeds NQT MFMAs; S^T = K Q^T and O^T += Vt P^T with the operand maps of the kernels above (P^T straight from the score registers)
//   * scores live in the exponent's own unit: Q carries 1/8 log2 e (applied in fp32 by the to_qkv epilogue — GemmParams::rope_cs_q — or, for callers with
//     plain q, by this kernel on the fp16 fragments), and the reference maximum is the C INPUT of the score MFMAs (S' = K Q^T - mref comes out of the
//     matrix pipe), so a probability is ONE v_exp_f32; the reference maximum moves only when a score exceeds it by more than 8 (P <= 2^8: exact in fp16
//     to the same 11 bits; the sums are fp32): detection is lane-local (8 v_max3 + 1 compare per query tile), the cross-lane maximum (v_permlane16_swap /
//     v_permlane32_swap) and the rescale of O are a rare wave-uniform branch; the row sums are a fifth output tile of the PV product (a ones A operand:
//     6 more MFMAs per key block instead of 48 additions per lane, and no cross-lane sum at the end)
//   profiles/round5/attn_flash_*: 373 -> 159 us (first version) -> see there, per 80-frame x 16-head launch
constexpr float kLazyThr = 8.0f;   // in exponent units (log2)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int attn_waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

// 16-byte direct-to-LDS load, scalar-base form: address = sbase (SGPR pair) + voff (32-bit per-lane byte offset); LDS destination = lds_addr (-> M0) + lane * 16
__device__ __forceinline__ void glds16_s_attn(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
// bank swizzle of the K rows of the flash kernel (16-byte chunk c of key row r sits at c ^ kswz(r)): its lanes read the rows 8 (li >> 2) + (li & 3) (+ 4),
// and with (r & 7) the lanes li and li + 12 / li + 4 and li + 8 of a ds_read_b128 lane group would share banks; bits 1, 3, 4 of the row separate all sixteen
__device__ __forceinline__ int kswz(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }
// max over the four lanes li, li + 16, li + 32, li + 48 (every lane ends with the result)
__device__ __forceinline__ float max_over_g(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // rows 1 <-> 0, 3 <-> 2
    v = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);   // lanes 32-63 <-> 0-31
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}
template <int NQT, int OCC, bool RAGGED, bool PRESCALED>   // OCC = blocks per CU the register budget must admit; RAGGED: S % 64 != 0; PRESCALED: Q holds q / 8 log2 e
__global__ __launch_bounds__(256, OCC) void attn_flash_kernel(const f16* __restrict__ Q, const f16* __restrict__ K, const f16* __restrict__ Vt,
                                                            f16* __restrict__ O, int heads, int S, int nfh, int nqb, int sc1) {
    const bool dbg_force = (sc1 & 2) != 0;
    constexpr int SLOT = 16384, NSLOT = 3;          // per slot: K block [64 keys][128 B] | Vt block [64 d][128 B]
    __shared__ __attribute__((aligned(1024))) char smem[NSLOT * SLOT];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: the LDS-DMA base goes to M0 (no waterfall loop)
    const int li = lane & 15, g = lane >> 4;
    // block -> ((frame, head), query block).  The nqb query blocks of one (frame, head) re-read the same K / Vt: give them ids that are 8 apart
    int fh, qb;
    {
        const int bid = blockIdx.x;
        if ((nfh & 7) == 0) {
            const int x = bid & 7, j = bid >> 3, fl = j / nqb;
            qb = j - fl * nqb;
            fh = fl * 8 + x;
        } else {
            fh = bid / nqb;
            qb = bid - fh * nqb;
        }
    }
    const int nb = fh / heads, head = fh - nb * heads;
    const f16* Kg = K + (size_t)fh * S * 64;
    const f16* Vg = Vt + (size_t)fh * 64 * S;
    const f16* Qg = Q + (size_t)fh * S * 64;
    const int qt0 = (qb * 4 + w) * NQT;            // first 16-query tile of this wave

    // LDS-DMA of key block kb into ring slot `slot`: 8 K pieces + 8 Vt pieces of 1 KiB (8 rows of 128 B each); wave w owns pieces 2 w, 2 w + 1 of both.
    // Rows / columns beyond S are clamped to valid memory (finite values): their scores are masked and their probabilities are exactly 0.
    // The fills are inline asm in the scalar-base form (wave-uniform base + 32-bit lane offset): behind the builtin, hipcc's wait-count pass assumes
    // every later ds_read may alias a fill in flight and drains vmcnt(0) in front of the first fragment read of EVERY key block (the ring's prefetch gone).
    const int prow = lane >> 3, pcp = lane & 7;
    const unsigned smem_base = (unsigned)(size_t)(lptr_t)smem;
    unsigned kvo[2], vvo[2];          // per-lane source offsets (bytes) of this wave's two K / two Vt pieces inside key block 0
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * (2 * w + i) + prow;     // key row / feature row of the piece's lane
        kvo[i] = (unsigned)(r * 128 + ((pcp ^ kswz(r)) << 4));
        vvo[i] = (unsigned)(r * S * 2 + ((pcp ^ ((r >> 1) & 7)) << 4));
    }
    auto issue = [&](int kb, int slot) {
        const unsigned base = smem_base + slot * SLOT + 2 * w * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned off = kvo[i] + (unsigned)(kb * 8192);
            if constexpr (RAGGED) {               // clamp the ROW to the last valid key, keep the chunk
                const int rowoff = (int)(off & ~127u), klast = (S - 1) * 128;
                off = (unsigned)(rowoff < klast ? rowoff : klast) + (off & 127u);
            }
            glds16_s_attn(Kg, off, base + i * 1024);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned off = vvo[i] + (unsigned)(kb * 128);
            if constexpr (RAGGED) {               // clamp the 8-key chunk to the last valid one (S % 8 == 0)
                const int r = 8 * (2 * w + i) + prow, col = kb * 64 + ((pcp ^ ((r >> 1) & 7)) << 3);
                off = (unsigned)(r * S * 2) + (unsigned)((col < S ? col : S - 8) * 2);
            }
            glds16_s_attn(Vg, off, base + 8192 + i * 1024);
        }
    };

    const int nkb = (S + 63) >> 6;
    issue(0, 0);
    if (nkb > 1) issue(1, 1);

    f16x8 qf[NQT][2];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        int qr = (qt0 + qt) * 16 + li;
        qr = qr < S ? qr : S - 1;
        qf[qt][0] = *(const f16x8*)(Qg + (size_t)qr * 64 + 8 * g);
        qf[qt][1] = *(const f16x8*)(Qg + (size_t)qr * 64 + 32 + 8 * g);
    }
    // The compiler's own wait for the Q loads must sit HERE, ahead of the loop (the empty asm "uses" the registers): left to the first MFMA it lands INSIDE
    // the loop, and since the wait-count pass cannot see the inline-asm fills it is a vmcnt(0) that drains the ring's prefetch in every key block
    // (vmcnt retires in order: the first two key blocks, issued before Q, have landed here too)
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) asm volatile("" : "+v"(qf[qt][0]), "+v"(qf[qt][1]));
    if constexpr (!PRESCALED) {   // plain q: the exponent's unit goes onto the fp16 fragments here (one more fp16 rounding of q than the prescaled form)
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[qt][h][e] = (f16)((float)qf[qt][h][e] * kScaleLog2e);
    }

    // per-lane LDS offsets of the fragment reads inside a slot.  Key tile kt holds the keys 32 (kt >> 1) + 8 (i >> 2) + 4 (kt & 1) + (i & 3) on its rows i
    // (a lane simply reads that K row): the score registers of tiles 2 s and 2 s + 1 are then the P^T operand of the keys 32 s + 8 g .. + 7 in natural
    // order, and the Vt fragment of a lane is ONE 16-byte read (chunk 4 s + g of feature row d) instead of two 8-byte reads 32 bytes apart
    int koff[2][2];   // [kt & 1][k-step]  (+ (kt >> 1) * 4096)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 8 * (li >> 2) + 4 * e + (li & 3);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) koff[e][ks] = r * 128 + (((4 * ks + g) ^ kswz(r)) << 4);
    }
    int voff[2];      // [s2]  (+ dt * 2048)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) voff[s2] = 8192 + li * 128 + (((4 * s2 + g) ^ ((li >> 1) & 7)) << 4);

    f32x4 o[NQT][5];      // [4] = the row sums: O^T tile of an all-ones feature tile (every row of it holds sum_k P[k][q])
    float nmref[NQT];     // minus the reference maximum of the lane's query: splat, it is the C operand of the score MFMAs
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        nmref[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (f16)1.0f;

    int slot = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        // this wave's pieces of block kb have landed (block kb + 1 may still be in flight); the barrier then (a) makes every wave's pieces visible and
        // (b) proves every wave has finished reading block kb - 1, whose slot block kb + 2 overwrites
        if (kb + 1 < nkb) __builtin_amdgcn_s_waitcnt(attn_waitcnt_imm(4, 15));
        else __builtin_amdgcn_s_waitcnt(attn_waitcnt_imm(0, 15));
        asm volatile("s_barrier" ::: "memory");
        if (kb + 2 < nkb) issue(kb + 2, slot >= 1 ? slot - 1 : NSLOT - 1);
        const char* sb = smem + slot * SLOT;
        slot = slot + 1 < NSLOT ? slot + 1 : 0;

        // ---- S' = K Q^T - mref: 4 key tiles x NQT query tiles, every K fragment feeds NQT MFMAs ----
        f32x4 sc[NQT][4], cin[NQT];
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) cin[qt] = f32x4{nmref[qt], nmref[qt], nmref[qt], nmref[qt]};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f16x8 k0 = *(const f16x8*)(sb + (kt >> 1) * 4096 + koff[kt & 1][0]);
            const f16x8 k1 = *(const f16x8*)(sb + (kt >> 1) * 4096 + koff[kt & 1][1]);
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                sc[qt][kt] = mfma16(k0, qf[qt][0], cin[qt], 0, 0, 0);
                sc[qt][kt] = mfma16(k1, qf[qt][1], sc[qt][kt], 0, 0, 0);
            }
        }
        if constexpr (RAGGED) {
            if (kb * 64 + 64 > S) {   // padded keys exist only in the last block (wave-uniform)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kb * 64 + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r >= S) {
#pragma unroll
                            for (int qt = 0; qt < NQT; ++qt) sc[qt][kt][r] = -INFINITY;
                        }
            }
        }
        // ---- does any score exceed its reference maximum by more than the threshold?  Lane-local: every key sits in some lane ----
        // Signed-INTEGER maxima of the score bits: a positive float is its bit pattern in integer order and every negative one is a negative integer, which is
        // all that "some score > 8" needs (v_max3_i32; behind fmaxf hipcc canonicalises every MFMA result with a v_max_f32 x, x, x of its own, and inline
        // asm must not read MFMA results: the compiler pads that hazard only for instructions it emits itself)
        bool need = kb == 0 || dbg_force;   // the first block SETS the reference maximum (with cin = 0 a row of very negative scores would underflow to a zero sum)
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            int im = (int)0x80000000;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                // (every element through a scalar float first: __builtin_bit_cast applied to the vector subscript itself read element 0 four times — hipcc, ROCm 7.2)
                const float s0 = sc[qt][kt][0], s1 = sc[qt][kt][1], s2_ = sc[qt][kt][2], s3 = sc[qt][kt][3];
                im = max(max(__float_as_int(s0), __float_as_int(s1)), im);
                im = max(max(__float_as_int(s2_), __float_as_int(s3)), im);
            }
            need = need || (im > __float_as_int(kLazyThr));
        }
        if (__any(need)) {   // rare after the first block: a jump of the maximum.  Everything at the old reference is rescaled exactly once, the pending scores re-based
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    m = __builtin_fmaxf(__builtin_fmaxf(sc[qt][kt][0], sc[qt][kt][1]), m);
                    m = __builtin_fmaxf(__builtin_fmaxf(sc[qt][kt][2], sc[qt][kt][3]), m);
                }
                const float bm = max_over_g(m);
                const float delta = kb == 0 ? bm : __builtin_fmaxf(bm, 0.f);   // the reference only ever rises after the first block
                if (kb != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int dt = 0; dt < 5; ++dt) o[qt][dt] = o[qt][dt] * alpha;
                }
                nmref[qt] -= delta;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[qt][kt][r] -= delta;
            }
        }
        // ---- P = 2^S' (one v_exp_f32 per score), then O^T += Vt P^T: two 32-key steps x (4 feature tiles + the ones tile); every Vt fragment feeds NQT MFMAs ----
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f16x8 pf[NQT];
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[qt][r] = (f16)__builtin_amdgcn_exp2f(sc[qt][2 * s2][r]);
                    pf[qt][4 + r] = (f16)__builtin_amdgcn_exp2f(sc[qt][2 * s2 + 1][r]);
                }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f16x8 vf = *(const f16x8*)(sb + dt * 2048 + voff[s2]);
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt) o[qt][dt] = mfma16(vf, pf[qt], o[qt][dt], 0, 0, 0);
            }
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) o[qt][4] = mfma16(ones, pf[qt], o[qt][4], 0, 0, 0);
        }
    }

    const int Dm = heads * 64;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        const float inv = 1.0f / o[qt][4][0];
        const int qrow = (qt0 + qt) * 16 + li;
        if (qrow < S) {
            const int mrow = nb * S + qrow;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f16x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (f16)(o[qt][dt][r] * inv);
                store_f16x4_paired<16>(O + tiled_off(mrow, head * 64 + dt * 16 + 4 * g, Dm), h, lane, sc1 & 1);
            }
        }
    }
}

// One thread per 8 consecutive features (16-byte loads / stores; 8 lanes per head), D/8 threads per (b, p) column and as many
// columns per block as fit in 256 threads; grid = (column groups, query-frame group): with `split` every query frame of a column
// gets its own block, which loads only the K / V frames its causal mask admits — five times as many independent blocks for the
// 144-column batch-1 step (one block per column left 112 of 256 CUs idle and chained five softmaxes per thread).
// q.k uses v_dot2_f32_f16 (exact fp16 products, fp32 accumulation), the head reduction is DPP, exp is the raw v_exp_f32.
// TM = compile-time bound on the number of frames (5 for the DiT's window, 8 otherwise): the frame loops are fully unrolled.
template <int TM>
__global__ __launch_bounds__(256) void attn_temporal_kernel(const f16* __restrict__ q, const f16* __restrict__ kv,
                                                            f16* __restrict__ O, int ncol, int P, int D, int Tq, int t0, int Tmax,
                                                            int split) {
    const int tpc = D >> 3;                                     // threads per column
    const int ci = threadIdx.x / tpc, lc = threadIdx.x - ci * tpc;
    const int bp = blockIdx.x * ((int)blockDim.x / tpc) + ci;
    if (bp >= ncol) return;                                     // whole 8-lane head groups leave together (tpc % 8 == 0)
    const int b = bp / P, p = bp - b * P;
    const int c = lc * 8;
    const int tl_lo = split ? blockIdx.y : 0, tl_hi = split ? blockIdx.y + 1 : Tq;
    const int Tk = t0 + tl_hi;                    // frames 0 .. t0 + tl_hi - 1 are visible to the last query of this block
    union H8 { f16x8 v; f16x2 h[4]; };
    H8 k8[TM], v8[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        if (t < Tk) {
            const f16* base = kv + (((size_t)b * Tmax + t) * P + p) * 2 * D;
            k8[t].v = *(const f16x8*)(base + c);
            v8[t].v = *(const f16x8*)(base + D + c);
        }
    }
    // every query row of this block is fetched up front, together with K / V (one memory round trip for the block)
    H8 qall[TM];
#pragma unroll
    for (int tl = 0; tl < TM; ++tl)
        if (tl >= tl_lo && tl < tl_hi) qall[tl].v = *(const f16x8*)(q + (((size_t)b * Tq + tl) * P + p) * D + c);
#pragma unroll
    for (int tl = 0; tl < TM; ++tl) {
        if (tl >= tl_hi) break;
        if (tl < tl_lo) continue;
        const int tq = t0 + tl;
        const size_t row = ((size_t)b * Tq + tl) * P + p;
        float s[TM];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            s[t] = -INFINITY;
            if (t <= tq) {  // causal (model/attention.py:62-64), wave-uniform
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) d = dot2acc(qall[tl].h[e], k8[t].h[e], d, false);
                s[t] = group8_sum(d) * 0.125f;
                mx = fmaxf(mx, s[t]);
            }
        }
        float den = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if (t <= tq) {
                const float pr = __builtin_amdgcn_exp2f((s[t] - mx) * 1.4426950408889634f);   // argument <= 0
                den += pr;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(pr, (float)v8[t].v[e], acc[e]);   // explicit fma: the fused QKV + attention GEMM (gemm.hip) must round identically
            }
        }
        const float inv = 1.0f / den;
        f16x8 o8;
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = (f16)(acc[e] * inv);
        *(f16x8*)(O + tiled_off((int)row, c, D)) = o8;
    }
}

}  // namespace

// output as paired 16-byte write-through stores (common.h store_f16x4_paired); experiments build: GTAV_ATTN_SC1=1 enables it
static int g_attn_sc1 = GTAV_ENV_INT("GTAV_ATTN_SC1", 0) | (GTAV_ENV_INT("GTAV_ATTN_DBG_FORCE", 0) ? 2 : 0);   // measured neutral at B = 1 (attention re-reads nothing, writes little)

bool attn_spatial_wants_prescaled_q(int S) { return S > 0 && S % 8 == 0 && round_up(S, 32) > 160; }

int launch_attn_spatial(const f16* Q, const f16* K, const f16* Vt, f16* O, int NB, int heads, int S, hipStream_t stream, bool q_prescaled) {
    GTAV_REQUIRE(S > 0 && S % 8 == 0, "attn_spatial: S=%d must be a positive multiple of 8", S);
    GTAV_REQUIRE(!q_prescaled || attn_spatial_wants_prescaled_q(S), "attn_spatial: S=%d runs a kernel that takes plain q", S);
    const int S_pad = round_up(S, 32);
    const int nqt = cdiv(S, 16);
    static const int force_flash = GTAV_ENV_INT("GTAV_ATTN_FORCE_FLASH", 0);   // experiments build: the flash kernel for short sequences too (A/B runs)
    if (S_pad <= 160 && !force_flash) {   // short sequences (the DiT's frames): one pass per query tile, scores in registers
        const size_t lds = (size_t)S_pad * 128 + (size_t)64 * (S_pad + 8) * 2;
        int devid = 0;
        GTAV_CHECK_HIP(hipGetDevice(&devid));
        // enough blocks to fill 256 CUs when there are few (frame, head) pairs; each block re-stages K / Vt from L2
        int qsplit = 1;
        const int max_split = cdiv(nqt, 4);
        static const int blocks_target = GTAV_ENV_INT("GTAV_ATTN_S_BLOCKS", 512);
        while (NB * heads * qsplit < blocks_target && qsplit < max_split) ++qsplit;
        const dim3 grid(NB * heads, qsplit);
        const int nk = S_pad / 16;
#define GTAV_ATTN_1P(NK_)                                                                                                               \
        do {                                                                                                                          \
            static unsigned long long devs_ = 0;   /* hipFuncAttributeMaxDynamicSharedMemorySize is per device: one "raised" bit per ordinal */ \
            if (!(devs_ >> (devid & 63) & 1)) {                                                                                       \
                GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)attn_spatial_1p_kernel<4, NK_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                devs_ |= 1ull << (devid & 63);                                                                                        \
            }                                                                                                                         \
            GTAV_LAUNCH((attn_spatial_1p_kernel<4, NK_>), grid, dim3(256), lds, stream, Q, K, Vt, O, heads, S, qsplit, g_attn_sc1 & 1);   \
        } while (0)
#ifdef GTAV_EXPERIMENTS
        // experiments build: waves per block of the one-pass kernel at NK = 10 (A/B runs: GTAV_ATTN_1P_NW = 3, 5 or 9; the product runs 4)
        static const int nw_exp = GTAV_ENV_INT("GTAV_ATTN_1P_NW", 4);
#define GTAV_ATTN_1P_NW(NW_)                                                                                                            \
        do {                                                                                                                          \
            GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)attn_spatial_1p_kernel<NW_, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            int qs_ = 1;                                                                                                              \
            GTAV_LAUNCH((attn_spatial_1p_kernel<NW_, 10>), dim3(NB * heads, qs_), dim3(64 * NW_), lds, stream, Q, K, Vt, O, heads, S, qs_, g_attn_sc1 & 1); \
        } while (0)
        if (nk == 10 && nw_exp == 3) { GTAV_ATTN_1P_NW(3); GTAV_CHECK_HIP(hipGetLastError()); return 0; }
        if (nk == 10 && nw_exp == 5) { GTAV_ATTN_1P_NW(5); GTAV_CHECK_HIP(hipGetLastError()); return 0; }
        if (nk == 10 && nw_exp == 9) { GTAV_ATTN_1P_NW(9); GTAV_CHECK_HIP(hipGetLastError()); return 0; }
#undef GTAV_ATTN_1P_NW
#endif
        if (nk == 2) GTAV_ATTN_1P(2);
        else if (nk == 4) GTAV_ATTN_1P(4);
        else if (nk == 6) GTAV_ATTN_1P(6);
        else if (nk == 8) GTAV_ATTN_1P(8);
        else GTAV_ATTN_1P(10);
#undef GTAV_ATTN_1P
        GTAV_CHECK_HIP(hipGetLastError());
        return 0;
    }
    // long sequences (the VAE's 576 tokens): flash form, K / Vt streamed through a 3-slot LDS ring (48 KiB whatever S is), 4 waves x NQT query tiles per block.
    // NQT by the fewest padded query tiles; on a tie the larger tile once the grid fills the chip, else the smaller (more blocks)
    int best = 0, best_pad = 1 << 30, best_nqb = 0;
    for (int c = 2; c <= 4; ++c) {
        const int nqb_c = cdiv(nqt, 4 * c), pad = nqb_c * c;
        const bool better = pad < best_pad || (pad == best_pad && NB * heads * best_nqb >= 512);
        if (better) best = c, best_pad = pad, best_nqb = nqb_c;
    }
    static const int force_nqt = GTAV_ENV_INT("GTAV_ATTN_FLASH_NQT", 0), occ3 = GTAV_ENV_INT("GTAV_ATTN_FLASH_OCC3", 1);   // experiments build: A/B runs
    if (force_nqt >= 2 && force_nqt <= 4) best = force_nqt, best_nqb = cdiv(nqt, 4 * force_nqt);
    const int nfh = NB * heads;
    GTAV_REQUIRE((long long)nfh * best_nqb < (1ll << 31) && (long long)S * 64 * 2 < (1ll << 31), "attn_spatial: grid / sequence too large");
    const dim3 fgrid(nfh * best_nqb);
#define GTAV_ATTN_FLASH_(NQT_, OCC_, RG_, PS_) GTAV_LAUNCH((attn_flash_kernel<NQT_, OCC_, RG_, PS_>), fgrid, dim3(256), 0, stream, Q, K, Vt, O, heads, S, nfh, best_nqb, g_attn_sc1)
#define GTAV_ATTN_FLASH(NQT_, OCC_)                                                                    \
    do {                                                                                               \
        if (S % 64 != 0) { if (q_prescaled) GTAV_ATTN_FLASH_(NQT_, OCC_, true, true); else GTAV_ATTN_FLASH_(NQT_, OCC_, true, false); }   \
        else { if (q_prescaled) GTAV_ATTN_FLASH_(NQT_, OCC_, false, true); else GTAV_ATTN_FLASH_(NQT_, OCC_, false, false); }             \
    } while (0)
    if (best == 2) GTAV_ATTN_FLASH(2, 3);
    else if (best == 3 && occ3) GTAV_ATTN_FLASH(3, 3);
    else if (best == 3) GTAV_ATTN_FLASH(3, 2);
    else GTAV_ATTN_FLASH(4, 2);
#undef GTAV_ATTN_FLASH
#undef GTAV_ATTN_FLASH_
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_attn_temporal(const f16* q, const f16* kv, f16* O, int B, int P, int D, int Tq, int t0, int Tmax,
                         hipStream_t stream) {
    GTAV_REQUIRE(D % 256 == 0 && D <= 2048, "attn_temporal: D=%d must be a multiple of 256 and <= 2048", D);
    GTAV_REQUIRE(Tq > 0 && t0 >= 0 && t0 + Tq <= Tmax && Tmax <= 8, "attn_temporal: window t0=%d Tq=%d Tmax=%d (max 8)", t0, Tq, Tmax);
    static const int split_max = GTAV_ENV_INT("GTAV_ATTN_T_SPLIT_MAX", 1024);
    const int split = (Tq > 1 && B * P < split_max) ? 1 : 0;   // few columns: one block per (column group, query frame)
    const int tpc = D / 8, cpb = tpc >= 256 ? 1 : 256 / tpc;  // threads per column, columns per block
    const dim3 grid(cdiv(B * P, cpb), split ? Tq : 1), block(tpc * cpb);
    if (Tmax <= 5) GTAV_LAUNCH(attn_temporal_kernel<5>, grid, block, 0, stream, q, kv, O, B * P, P, D, Tq, t0, Tmax, split);
    else GTAV_LAUNCH(attn_temporal_kernel<8>, grid, block, 0, stream, q, kv, O, B * P, P, D, Tq, t0, Tmax, split);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gtav
