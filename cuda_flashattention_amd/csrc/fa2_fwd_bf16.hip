// fa2_fwd_bf16.hip -- FlashAttention-2 forward for gfx950 (MI355X), bf16 in / fp32 accumulate.
//
// Replaces flash_attention_2_forward_kernel (reference
// src/02_flash_attention_v2_forward/flash_attention_kernel.cu:37-297, cleaned copy
// src/03_flash_attention_v2_ring/common/flash_attention_kernel.cu:13-130) and, through the
// resume/finalize switches, ring_attention_forward_kernel
// (src/03_flash_attention_v2_ring/common/ring_attention_kernel.cu:13-140).  Same maths --
// m' = max(m, rowmax), l = e^{m-m'} l + sum e^{s-m'}, O = e^{m-m'} O + P V, O/l, L = m + ln l
// (attention_helper.h:76-110) -- on a different machine mapping:
//
//   * workgroup = 4 waves = 256 query rows of one head, ONE wave per SIMD so a wave may use the
//     whole 512-entry register file: it owns 64 rows (two 32-row blocks), keeps their O^T
//     accumulators pinned in AGPRs (2 x D/32 tiles = 128 registers at D = 128) and their Q
//     fragments in VGPRs, and every K / V^T fragment it reads from LDS feeds BOTH row blocks;
//   * S^T = K Q^T on v_mfma_f32_32x32x16_bf16 ("swapped" product): the accumulator column
//     (= lane & 31) is the QUERY row, its 16 registers are keys, so a row's max and sum are
//     in-lane reductions plus one permlane32_swap -- the reference's (Bc + d) shuffle
//     butterflies per row per tile are gone;
//   * P stays in registers: the S^T accumulator, packed to bf16, IS the B operand of
//     O^T += V^T P^T (contraction over the accumulator's row index), V^T fragments come from
//     LDS through ds_read_b64_tr_b16;
//   * with one wave per SIMD nothing but the wave's own instruction stream can overlap the
//     softmax arithmetic with the matrix pipe, so the loop is a three-stage software pipeline
//     over 32-key half-tiles u:   A(u+1): S^T of the next half-tile   (16 MFMAs)
//                                 X(u)  : max / exp2 / sum / pack of the current one (VALU)
//                                 B(u-1): O^T += V^T P^T of the previous one (16 MFMAs)
//     A(u+1) runs beside the row-max of u, B(u-1) beside its exponentials; a rescale of O (rare:
//     only when some row's max moved) is applied after B(u-1), so everything accumulated at the
//     old max is scaled exactly once;
//   * K/V tiles (64 keys) arrive by LDS-DMA (buffer_load ... lds, 16 B per lane, issued two tiles
//     ahead) into a ring of three XOR-swizzled images (fa2_common.h: lds_off; the swizzle is
//     applied to the SOURCE address, the LDS write is linear); the buffer resource's range check
//     zero-fills keys past the end; one barrier per 64-key tile;
//   * exp2 domain (v_exp_f32), running max kept in natural units so L matches the reference's
//     natural-log LSE;
//   * work mapping is XCD-aware (fa2_common.h: map_block).
#include <type_traits>

#include "fa2_common.h"
#include "fa2_launch.h"

namespace fa2 {

constexpr int kFwdWaves = 4;
constexpr int kFwdRows = 64 * kFwdWaves;   // query rows per workgroup
constexpr int kFwdKV = 64;                  // keys per DMA tile (two 32-key half-tiles)
constexpr int kFwdBufs = 3;                 // LDS ring depth
constexpr float kRescaleThr = 6.0f;         // natural-log units of the scaled score

typedef __attribute__((address_space(3))) void* lds_ptr_t;
#ifdef FA2_SUM_PK
typedef f32x2 LSum;
#else
typedef float LSum;
#endif

template <int D, bool CAUSAL, bool STATE>
__global__ void __launch_bounds__(256, 1) fa2_fwd_bf16_kernel(FwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = D * 2;               // bytes per tile row
    constexpr int TILEB = kFwdKV * ROWB;      // bytes per K (or V) tile
    constexpr int VREG = kFwdBufs * TILEB;    // LDS: [3 K tiles][3 V tiles]; every read offset < 64 KiB from its region base
    constexpr int HALFB = 32 * ROWB;          // one 32-key half-tile
    constexpr int CPR = D / 8;                // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;             // rows per DMA wave-instruction (1 KiB)
    constexpr int NP = kFwdKV / RPI;          // DMA pieces per tensor per tile: 16 or 8
    constexpr int KS = D / 16;                // k-steps of QK^T
    constexpr int DT = D / 32;                // 32-column tiles of O
    constexpr int NG = 2 * DT;                // PV groups per half-tile: (dt, sp)
    constexpr int RPG = 16 / NG;              // S registers exponentiated beside each PV group
    // AGPR map (literal ranges owned by asm, fa2_common.h): O^T tile (qb, dt) at A_O + (qb*DT+dt)*16,
    // Q fragment (qb, s) at A_QF + (qb*KS+s)*4 -- the B operand of S^T = K Q^T, taken straight from AGPRs.
    constexpr int A_O = 0;
    constexpr int A_QF = 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31;
    const int h = lane >> 5;

    const int nrb = (p.Nq + kFwdRows - 1) / kFwdRows;
    int head, rb;
    map_block(blockIdx.x, p.BH, nrb, head, rb);
    if (CAUSAL) rb = nrb - 1 - rb;            // heaviest row-blocks first

    const int Nq = p.Nq, Nk = p.Nk;
    const char* Qh = (const char*)p.Q + (size_t)head * Nq * ROWB;
    const char* Kh = (const char*)p.K + (size_t)head * Nk * ROWB;
    const char* Vh = (const char*)p.V + (size_t)head * Nk * ROWB;

    const int q0 = rb * kFwdRows + wave * 64;          // first query row of this wave

    // Number of real K/V tiles; the loop runs whole triples of tiles (ring of three buffers,
    // unrolled so that every LDS offset is an immediate) and at least one tile more than the real
    // ones, so that the pipeline drains inside the loop: the extra tiles are fully masked (keys
    // >= Nk read as zeros and are masked; causal: keys above every row of the workgroup).
    int ntiles = (Nk + kFwdKV - 1) / kFwdKV;
    if (CAUSAL) {
        const int last_q = min(rb * kFwdRows + kFwdRows - 1, Nq - 1);
        const int last_key = last_q + p.causal_shift;             // last visible key index
        const int lim = last_key < 0 ? 0 : last_key / kFwdKV + 1;
        ntiles = min(ntiles, lim);
    }
    const int niter = ((ntiles + 1 + 2) / 3) * 3;

    // ---- Q fragments of both row blocks -> AGPRs; lane holds Q[q][16s + 8h .. +7].
    int qrow[2];
    static_for<2>([&](auto QB) {
        constexpr int qb = decltype(QB)::value;
        qrow[qb] = q0 + 32 * qb + qi;
        const int qld = qrow[qb] < Nq ? qrow[qb] : Nq - 1;     // clamped for loads (pad, don't mask)
        static_for<KS>([&](auto S) {
            constexpr int sidx = decltype(S)::value;
            acc_write_frag<A_QF + (qb * KS + sidx) * 4>(
                *reinterpret_cast<const bf16x8*>(Qh + (size_t)qld * ROWB + 16 * (2 * sidx + h)));
        });
    });

    // ---- running state.  O^T lives in literal AGPRs (fa2_common.h: acc_*): tile (qb, dt) is
    // a[(qb * DT + dt) * 16 .. +15], register 4g + e of it is O[q][32 dt + 8 g + 4 h + e].
    // m_run: the reference maximum (natural units), mb = m_run * log2(e) (0 while -inf), thr = the raw
    // (unscaled) score above which a row asks for a new reference; l_run: this lane's share of the row sum.
    const float inv_scale = 1.0f / p.scale;
    float m_run[2], mb[2], thr[2];
#ifdef FA2_SUM_PK
    f32x2 l_run[2];
#else
    float l_run[2];
#endif
    static_for<2>([&](auto QB) {
        constexpr int qb = decltype(QB)::value;
        if (STATE && p.resume) {
            const int qld = qrow[qb] < Nq ? qrow[qb] : Nq - 1;
            const float* Oa = p.Oacc + ((size_t)head * Nq + qld) * D;
            static_for<4 * DT>([&](auto G) {
                constexpr int dt = decltype(G)::value / 4, g = decltype(G)::value % 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(Oa + 32 * dt + 8 * g + 4 * h);
                static_for<4>([&](auto E) {
                    constexpr int e = decltype(E)::value;
                    acc_write<A_O + (qb * DT + dt) * 16 + 4 * g + e>(v[e]);
                });
            });
            m_run[qb] = p.M[(size_t)head * Nq + qld];
            l_run[qb] = LSum{h == 0 ? p.L[(size_t)head * Nq + qld] : 0.0f};
            mb[qb] = m_run[qb] == -INFINITY ? 0.0f : m_run[qb] * kLog2e;
            thr[qb] = (m_run[qb] + kRescaleThr) * inv_scale;
        } else {
            static_for<16 * DT>([&](auto R) { acc_write<A_O + qb * DT * 16 + decltype(R)::value>(0.0f); });
            m_run[qb] = -INFINITY;
            l_run[qb] = LSum{0.0f};
            mb[qb] = 0.0f;
            thr[qb] = -INFINITY;
        }
    });

    // ---- LDS-DMA staging: wave w issues pieces w, w + 4, ... of the 2 * NP pieces of a tile.  The
    // swizzle term depends on the row modulo 16 only, hence is the same for all pieces of a wave:
    // one per-lane voffset, everything else wave-uniform (soffset); rows >= Nk read as zeros.
    const int drow = lane / CPR;
    const int dslot = lane % CPR;
    const int prow = wave * RPI + drow;
    const int doff = drow * ROWB + 16 * ((lds_off<D>(prow, dslot) - ROWB * prow) >> 4);
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, Nk * ROWB, 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, Nk * ROWB, 0x00020000);
    auto stage = [&](int t, int buf) {
        char* b = smem + buf * TILEB;
#pragma unroll
        for (int j = wave; j < 2 * NP; j += kFwdWaves) {
            const int which = j / NP, piece = j % NP;
            const int soff = (t * kFwdKV + piece * RPI) * ROWB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(which ? v_rsrc : k_rsrc, (lds_ptr_t)(b + which * VREG + piece * 1024),
                                                     16, doff, soff, 0, 0);
        }
    };

    const float c2 = p.scale * kLog2e;   // exp(s * scale - m) = exp2(s * c2 - m * log2e)

    // ---- loop-invariant per-lane LDS offsets
    const int trq = (lane & 15) >> 2;       // row inside the 4-row block of a transposed read
    const int trp = lane & 3;               // 4-column group inside the 16-column block
    const int trcb = (lane >> 4) & 1;       // which 16-column half of the 32-column tile
    int roff[KS], toff[DT][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) roff[s] = lds_off<D>(qi, 2 * s + h);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)      // V region base folded in; +16 rows: sp = 1
            toff[dt][jj] = VREG + lds_off<D>(8 * jj + 4 * h + trq, 4 * dt + 2 * trcb + (trp >> 1)) + 8 * (trp & 1);
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;

    // ---- pipeline registers
    f32x16 scur[2];          // S^T of half-tile u     (keys on registers, query on the lane)
    bf16x8 pprev[2][2];      // packed P of half-tile u-1: [row block][k-step]

    // A stage: S^T of the 32 keys whose K rows start KOFF bytes into the K region, masked.  key0 is the
    // index of the first of those keys.
    // It also takes the in-lane maximum of the CURRENT half-tile (scur) between its MFMA statements
    // (rmax[qb], over this lane's 16 keys): scur is threaded through the statements so the partial
    // maxima sit beside the MFMAs instead of after them.
    float rmax[2] = {-INFINITY, -INFINITY};
    auto stage_a = [&](auto KOFF_, int key0, f32x16 (&snext)[2], bool with_max) {
        constexpr int KOFF = decltype(KOFF_)::value;
        const char* Kt = smem + KOFF;
        constexpr int RPS = 16 / KS;                 // scur registers folded into the maximum per k-step
        bf16x8 ka = lds_read_frag(Kt, roff[0]);
        bf16x8 kb1 = lds_read_frag(Kt, roff[1]);
        rmax[0] = -INFINITY; rmax[1] = -INFINITY;
        static_for<KS>([&](auto S) {
            constexpr int sidx = decltype(S)::value;
            bf16x8 kn = kb1;
            if constexpr (sidx + 2 < KS) kn = lds_read_frag(Kt, roff[sidx + 2]);
            if constexpr (sidx == 0)
                mfma2_bagpr_init<A_QF + (0 * KS + sidx) * 4, A_QF + (1 * KS + sidx) * 4>(snext[0], snext[1], ka, scur[0], scur[1],
                                                                                         rmax[0], rmax[1]);
            else
                mfma2_bagpr<A_QF + (0 * KS + sidx) * 4, A_QF + (1 * KS + sidx) * 4>(snext[0], snext[1], ka, scur[0], scur[1],
                                                                                    rmax[0], rmax[1]);
            if (with_max) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int r = sidx * RPS; r < (sidx + 1) * RPS; ++r) rmax[qb] = fmaxf(rmax[qb], scur[qb][r]);
            }
            ka = kb1; kb1 = kn;
        });
        const bool tail = key0 + 32 > Nk;
        bool diag = false;
        if (CAUSAL) diag = key0 + 31 > q0 + p.causal_shift;
        if (tail || diag) {
            mfma_vgpr_settle(snext[1]);          // the products were issued a moment ago
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + acc_row(r, h);
                    bool dead = key >= Nk;
                    if (CAUSAL) dead = dead || key > qrow[qb] + p.causal_shift;
                    if (dead) snext[qb][r] = -INFINITY;
                }
        }
    };

    // X + B stages of one half-tile step: O^T += V^T P^T of half-tile u-1 (V rows start VOFF bytes
    // into the V region, P = pprev) with the exponentials of half-tile u (scur) issued between its
    // MFMA groups, packed to bf16 as they appear (they are the next step's pprev) and summed from the
    // packed values.  The running maximum is a LAZY reference: the rows of a block move to their
    // current maximum only when some row's score has risen more than kRescaleThr above its reference
    // (or has no reference yet) -- one compare per row block on the common path.  In between,
    // P = exp(s - m_ref) may exceed 1 (by at most e^kRescaleThr): harmless in fp32 sums and in bf16 P,
    // whose relative precision does not depend on magnitude -- and O, l and L come out the same.
    auto stage_xb = [&](auto VOFF_) {
        constexpr int VOFF = decltype(VOFF_)::value;
        // first V^T fragment goes out before the arithmetic
        bf16x4 va0, va1;
        lds_read_tr2_asm<VOFF>(va0, va1, lbase + toff[0][0], lbase + toff[0][1]);

        float alpha[2] = {1.0f, 1.0f};
        bool need[2] = {false, false};
        if (__any(rmax[0] > thr[0] || rmax[1] > thr[1])) {      // rmax: taken beside the S^T MFMAs (stage_a)
            asm volatile("; fa2-cold: new softmax reference");
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float mx = half_max(rmax[qb]) * p.scale;
                const bool grow = mx > m_run[qb] + kRescaleThr;        // also true from m_run = -inf
                const bool any_grow = __any(grow);
                const float m_new = any_grow ? fmaxf(m_run[qb], mx) : m_run[qb];
                // O only needs scaling if some row already accumulated something at an older reference
                need[qb] = any_grow && __any(m_run[qb] != -INFINITY && m_new != m_run[qb]);
                // first visible key of a row: m_run = -inf -> alpha = 0 (its accumulators are 0 anyway)
                alpha[qb] = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f((m_run[qb] - m_new) * kLog2e);
                m_run[qb] = m_new;
                // a row that has seen no visible key yet keeps p = 0 (avoid inf - inf)
                mb[qb] = m_new == -INFINITY ? 0.0f : m_new * kLog2e;
                thr[qb] = (m_new + kRescaleThr) * inv_scale;
                l_run[qb] *= alpha[qb];
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        uint32_t pw[2][8];       // packed P of half-tile u: word 4s + i = registers 8s + 2i, 8s + 2i + 1
        static_for<NG>([&](auto G) {
            constexpr int g = decltype(G)::value;
            constexpr int dt = g >> 1, sp = g & 1;
            bf16x8 vf;
#pragma unroll
            for (int e = 0; e < 4; ++e) { vf[e] = va0[e]; vf[4 + e] = va1[e]; }
            // the S tiles of the current half-tile and the running sums are threaded through the
            // statement: the exponentials below stay between this group's MFMAs and the next group's
            if constexpr (g + 1 < NG) {
                constexpr int dtn = (g + 1) >> 1;
                constexpr int spo = ((g + 1) & 1) * 16 * ROWB;
                bf16x4 vn0, vn1;
                pv_group_next<VOFF + spo, A_O + (0 * DT + dt) * 16, A_O + (1 * DT + dt) * 16>(
                    vn0, vn1, lbase + toff[dtn][0], lbase + toff[dtn][1], vf, pprev[0][sp], pprev[1][sp], scur[0], scur[1],
                    l_run[0], l_run[1]);
                va0 = vn0; va1 = vn1;
            } else {
                pv_group_last<A_O + (0 * DT + dt) * 16, A_O + (1 * DT + dt) * 16>(vf, pprev[0][sp], pprev[1][sp], scur[0],
                                                                                  scur[1], l_run[0], l_run[1]);
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = g * RPG; r < (g + 1) * RPG; r += 2) {
                    const float e0 = __builtin_amdgcn_exp2f(scur[qb][r] * c2 - mb[qb]);
                    const float e1 = __builtin_amdgcn_exp2f(scur[qb][r + 1] * c2 - mb[qb]);
                    pw[qb][r >> 1] = pack_bf16_pair(e0, e1);
#if defined(FA2_SUM_DOT2)
                    sum_bf16_pair(l_run[qb], pw[qb][r >> 1]);
#elif defined(FA2_SUM_PK)
                    l_run[qb] += f32x2{e0, e1};
#else
                    l_run[qb] += e0 + e1;
#endif
                }
        });
        thread2f(scur[0], scur[1], l_run[0], l_run[1]);

        if (need[0] || need[1]) {   // everything accumulated so far (through half-tile u-1) is at the old reference
            mfma_acc_settle();
            static_for<2>([&](auto QB) {
                constexpr int qb = decltype(QB)::value;
                if (need[qb])
                    static_for<4 * DT>([&](auto R4) { acc_scale4<A_O + qb * DT * 16 + 4 * decltype(R4)::value>(alpha[qb]); });
            });
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const u32x4 w = {pw[qb][4 * sp], pw[qb][4 * sp + 1], pw[qb][4 * sp + 2], pw[qb][4 * sp + 3]};
                pprev[qb][sp] = __builtin_bit_cast(bf16x8, w);
            }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: tiles 0 and 1 into buffers 0 and 1; buffer 2 (read by the first, all-zero-P
    // PV step) must hold finite data: tile 0 again.
    stage(0, 0);
    stage(1, 1);
    stage(0, 2);
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int e = 0; e < 8; ++e) pprev[qb][sp][e] = (__bf16)0.0f;
    {
        f32x16 s0[2];
        stage_a(std::integral_constant<int, 0>{}, 0, s0, false);
        mfma_vgpr_settle(s0[1]);
        scur[0] = s0[0]; scur[1] = s0[1];
    }

    // One 64-key tile T living in ring buffer B (= T mod 3):
    //   step 1 (u = 2T)  : A on K[T] second half       ; X(u) ; B on V[T-1] second half (buffer B+2)
    //   barrier          : buffer B+2 is free, tile T+1 has landed -> DMA tile T+2 into B+2
    //   step 2 (u = 2T+1): A on K[T+1] first half (B+1) ; X(u) ; B on V[T] first half
#ifdef FA2_DIAG_STAMPS
    unsigned long long dg_a = 0, dg_xb = 0, dg_sync = 0, dg_t = 0, dg_loop = 0;
#define FA2_STAMP(acc) { unsigned long long ts_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) :: "memory"); acc += ts_ - dg_t; dg_t = ts_; }
#else
#define FA2_STAMP(acc)
#endif
    auto tile = [&](auto B_, int T) {
        constexpr int B = decltype(B_)::value;
        constexpr int B1 = (B + 1) % kFwdBufs, B2 = (B + 2) % kFwdBufs;
        f32x16 snext[2];
        stage_a(std::integral_constant<int, B * TILEB + HALFB>{}, T * kFwdKV + 32, snext, true);
        FA2_STAMP(dg_a)
        stage_xb(std::integral_constant<int, B2 * TILEB + HALFB>{});
        scur[0] = snext[0]; scur[1] = snext[1];
        FA2_STAMP(dg_xb)
#ifndef FA2_ABL_NOBAR
        __syncthreads();
#endif
#ifndef FA2_ABL_NODMA
        stage(T + 2, B2);
#endif
        FA2_STAMP(dg_sync)
        stage_a(std::integral_constant<int, B1 * TILEB>{}, (T + 1) * kFwdKV, snext, true);
        FA2_STAMP(dg_a)
        stage_xb(std::integral_constant<int, B * TILEB>{});
        scur[0] = snext[0]; scur[1] = snext[1];
        FA2_STAMP(dg_xb)
    };

#ifdef FA2_DIAG_STAMPS
    { unsigned long long ts_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) :: "memory"); dg_t = ts_; dg_loop = ts_; }
#endif
    for (int T = 0; T < niter; T += 3) {
        tile(std::integral_constant<int, 0>{}, T);
        tile(std::integral_constant<int, 1>{}, T + 1);
        tile(std::integral_constant<int, 2>{}, T + 2);
    }
#ifdef FA2_DIAG_STAMPS
    if (lane == 0) {   // diagnostic build only: cycle sums overwrite the head of L
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.L) + ((size_t)blockIdx.x * kFwdWaves + wave) * 4;
        dbg[0] = dg_a; dbg[1] = dg_xb; dbg[2] = dg_sync; dbg[3] = dg_t - dg_loop;
    }
    return;
#endif

    // ---- epilogue
    mfma_acc_settle();
    static_for<2>([&](auto QB) {
        constexpr int qb = decltype(QB)::value;
#ifdef FA2_SUM_PK
        const float l_tot = half_sum(l_run[qb][0] + l_run[qb][1]);
#else
        const float l_tot = half_sum(l_run[qb]);
#endif
        const size_t qoff = (size_t)head * Nq + qrow[qb];
        const bool fin = !STATE || p.finalize;
        const float inv = fin ? (l_tot > 0.0f ? 1.0f / l_tot : 0.0f) : 1.0f;
        static_for<4 * DT>([&](auto G) {
            constexpr int dt = decltype(G)::value / 4, g = decltype(G)::value % 4;
            constexpr int R = A_O + (qb * DT + dt) * 16 + 4 * g;
            f32x4 v;
            v[0] = acc_read<R>() * inv; v[1] = acc_read<R + 1>() * inv;
            v[2] = acc_read<R + 2>() * inv; v[3] = acc_read<R + 3>() * inv;
            if (qrow[qb] < Nq) {
                if (fin) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x4*>((char*)p.O + qoff * ROWB + 2 * (32 * dt + 8 * g + 4 * h)) = o;
                } else {
                    *reinterpret_cast<f32x4*>(p.Oacc + qoff * D + 32 * dt + 8 * g + 4 * h) = v;
                }
            }
        });
        if (qrow[qb] < Nq && h == 0) {
            if (fin) {
                p.L[qoff] = m_run[qb] + __builtin_logf(l_tot);
            } else {
                p.L[qoff] = l_tot;
                p.M[qoff] = m_run[qb];
            }
        }
    });
}

template <int D, bool CAUSAL, bool STATE>
static hipError_t launch_one(const FwdArgs& a, hipStream_t stream)
{
    constexpr int lds = kFwdBufs * 2 * kFwdKV * D * 2;
    auto kern = fa2_fwd_bf16_kernel<D, CAUSAL, STATE>;
    static bool attr_set[64] = {};
    hipError_t e = ensure_dynamic_lds(kern, lds, attr_set);
    if (e != hipSuccess) return e;
    const int nrb = (a.Nq + kFwdRows - 1) / kFwdRows;
    const dim3 grid((unsigned)(nrb * a.BH));
    hipLaunchKernelGGL(kern, grid, dim3(64 * kFwdWaves), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_fwd_bf16(const FwdArgs& a, hipStream_t stream)
{
    const bool state = a.resume || !a.finalize;
    if (a.d == 128) {
        if (state) return a.causal ? launch_one<128, true, true>(a, stream) : launch_one<128, false, true>(a, stream);
        return a.causal ? launch_one<128, true, false>(a, stream) : launch_one<128, false, false>(a, stream);
    }
    if (a.d == 64) {
        if (state) return a.causal ? launch_one<64, true, true>(a, stream) : launch_one<64, false, true>(a, stream);
        return a.causal ? launch_one<64, true, false>(a, stream) : launch_one<64, false, false>(a, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace fa2
