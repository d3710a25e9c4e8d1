// fa2_bwd_bf16.hip -- FlashAttention-2 backward for gfx950 (MI355X), bf16 in / fp32 accumulate.
//
// Replaces flash_attention_2_backward_kernel (reference
// src/02_flash_attention_v2_backward/flash_attention_backward_kernel.cu:47-246): the same
// maths -- D = rowsum(dO o O) (:97-120), P = exp(scale QK^T - L) (:157-173), dP = dO V^T
// (:176-187), dS = P o (dP - D) (:190-193), dQ = scale dS K (:196-205), dV = P^T dO,
// dK = scale dS^T Q (:208-221) -- but NOT its work split.  The reference is row-parallel and
// sums dK/dV with shared-memory and global atomicAdd (:208-231), which serialises and is
// order-dependent.  Here every gradient element is owned by exactly one wave:
//
//   kernel 0  fa2_bwd_delta      D[q] = sum_c dO[q][c] O[q][c]                (HBM-bound)
//   kernel 1  fa2_bwd_dq         row-parallel (the forward's mapping): a workgroup owns 256
//                                query rows, streams K/V tiles, keeps dQ^T in accumulators
//   kernel 2  fa2_bwd_dkdv       column-parallel: a workgroup owns 256 keys (a wave 32 of
//                                them, K/V fragments resident in registers), streams Q/dO
//                                tiles, keeps dK^T and dV^T in accumulators
//
// No atomics, no memset, bitwise reproducible.  The price is that S and dP are formed in both
// kernels (7 block products instead of 5); on MI355X the alternative -- fp32 atomics for dQ --
// is bounded by the chip-wide float-atomic rate (~1.3 TB/s): dQ alone would be
// B H (N/256) N d 4 B = 8.6 GB at (4,16,8192,128), a 6.6 ms floor, more than the two extra
// products cost on the matrix cores.
//
// MFMA orientation (fa2_common.h): in kernel 1 the accumulator column (lane) is the query,
// as in the forward; in kernel 2 it is the KEY, so that P[q][key] and dS[q][key], packed to
// bf16 in registers, are directly the B operands of dV^T += dO^T P and dK^T += Q^T dS (both
// contract over the accumulator's row index q) and dO^T / Q^T come from LDS through
// ds_read_b64_tr_b16.  Row constants ride in the accumulators: kernel 2 starts the S chain
// from -L/scale and the dP chain from -D, so p = exp2(c S') and dS = p dP' need no
// per-element subtraction (rows are registers there, not lanes).
#include <type_traits>

#include "fa2_common.h"
#include "fa2_launch.h"

namespace fa2 {

// --------------------------------------------------------------------------- kernel 0: D
// 16 lanes per row (16 bytes of dO and of O each per step), 4 rows per wave.  Besides D it
// writes the two row constants kernel 2 preloads into its accumulators, already transformed:
// RC[0][row] = -L/scale (so that exp2(c (S - L/scale)) = P) and RC[1][row] = -D.
template <int D>
__global__ void __launch_bounds__(256) fa2_bwd_delta_kernel(const __bf16* __restrict__ dO,
                                                            const __bf16* __restrict__ O,
                                                            const float* __restrict__ L,
                                                            float* __restrict__ Dv, float* __restrict__ RC,
                                                            size_t rows, int nq, int q_hs, int q_row0, size_t rc_plane,
                                                            float inv_scale)
{
    const size_t r = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    const size_t head = r / (size_t)nq, i = r % (size_t)nq;
    const size_t row = head * (size_t)q_hs + i;              // in the tensors (their pointers address row q_row0 of head 0)
    const size_t prow = row + (size_t)q_row0;                // in the dense workspace planes
    float acc = 0.0f;
    if (r < rows) {
#pragma unroll
        for (int c = sub * 8; c < D; c += 128) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(dO + row * D + c);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(O + row * D + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)b[e];
        }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 16);
    if (r < rows && sub == 0) {
        Dv[prow] = acc;
        RC[prow] = -L[row] * inv_scale;
        RC[rc_plane + prow] = -acc;
    }
}

// --------------------------------------------------------------------------- kernel 1: dQ
// Workgroup = 4 waves = 256 query rows of one head, ONE wave per SIMD; a wave owns 64 rows (two
// 32-row blocks).  Everything long-lived sits in the accumulator half of the register file as
// literal AGPR ranges owned by asm (fa2_common.h: acc_*): the dQ^T accumulators
// (2 x D/32 tiles), and the resident Q and dO fragments, which the S^T and dP^T products take
// straight from AGPRs as their B operand (mfma_bagpr).  The architectural VGPRs are left to the
// S^T/dP^T tiles of a 64-key tile, the packed dS and the streamed fragments.  K/V tiles (64 keys)
// arrive by LDS-DMA into a double-buffered swizzled image; every K/V row fragment and every K^T
// transposed fragment read from LDS feeds both row blocks.
constexpr int kBwdRows = 256;   // query rows per workgroup (kernel 1)
constexpr int kDqWaves = 4;
constexpr int kDqKV = 64;       // keys per streamed tile in kernel 1
constexpr int kDkQ = 32;        // query rows per sub-tile in kernel 2

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#include "fa2_bwd_dq_body.inc"

#define FA2_DQ_CLOBBERS "memory", "vcc", "v255", FA2_ACC_CLOBBERS
#define FA2_DQ_OPS_128 [r0] "v"(roff[0]), [r1] "v"(roff[1]), [r2] "v"(roff[2]), [r3] "v"(roff[3]), [r4] "v"(roff[4]), [r5] "v"(roff[5]), \
    [r6] "v"(roff[6]), [r7] "v"(roff[7]), [t0] "v"(toff[0]), [t1] "v"(toff[1]), [t2] "v"(toff[2]), [t3] "v"(toff[3]), [t4] "v"(toff[4]),     \
    [t5] "v"(toff[5]), [t6] "v"(toff[6]), [t7] "v"(toff[7])
#define FA2_DQ_OPS_64 [r0] "v"(roff[0]), [r1] "v"(roff[1]), [r2] "v"(roff[2]), [r3] "v"(roff[3]), [t0] "v"(toff[0]), [t1] "v"(toff[1]),    \
    [t2] "v"(toff[2]), [t3] "v"(toff[3])

// One literal VGPR move (see dkdv_vset below): seeds registers the generated bodies own.
template <int R>
__device__ __forceinline__ void dq_vset(float x)
{
    asm volatile("v_mov_b32 v%c1, %0" : : "v"(x), "i"(R) : "v255");
}

template <int D, int BUF, bool MASKED>
__device__ __forceinline__ void dq_body(const uint32_t (&roff)[D / 16], const uint32_t (&toff)[D / 16], float c2, float lq0, float lq1,
                                        const f32x16& nd0, const f32x16& nd1, int hi0, int hi1, int hp0, int hp1)
{
#define FA2_DQ_CASE(DD, B, M)                                                                                                      \
    if constexpr (D == DD && BUF == B && MASKED == bool(M))                                                                          \
        asm volatile(FA2_DQ_BODY_D##DD##_B##B##_M##M : : FA2_DQ_OPS_##DD, [c2] "s"(c2), [lq0] "v"(lq0), [lq1] "v"(lq1), [nd0] "v"(nd0), \
                     [nd1] "v"(nd1), [hi0] "v"(hi0), [hi1] "v"(hi1), [hp0] "v"(hp0), [hp1] "v"(hp1) : FA2_DQ_CLOBBERS);
    FA2_DQ_CASE(128, 0, 0) FA2_DQ_CASE(128, 1, 0) FA2_DQ_CASE(128, 2, 0) FA2_DQ_CASE(128, 0, 1) FA2_DQ_CASE(128, 1, 1) FA2_DQ_CASE(128, 2, 1)
    FA2_DQ_CASE(64, 0, 0) FA2_DQ_CASE(64, 1, 0) FA2_DQ_CASE(64, 2, 0) FA2_DQ_CASE(64, 0, 1) FA2_DQ_CASE(64, 1, 1) FA2_DQ_CASE(64, 2, 1)
#undef FA2_DQ_CASE
}

template <int D>
__device__ __forceinline__ void dq_prologue(const uint32_t (&roff)[D / 16], const uint32_t (&toff)[D / 16], float c2, float lq0, float lq1)
{
    // the early work of the first tile: its first fragment reads, and the (harmless) VALU pass over the seeded SET1
    const f32x16 z = {};
    if constexpr (D == 128)
        asm volatile(FA2_DQ_PRO_D128_M0 : : FA2_DQ_OPS_128, [c2] "s"(c2), [lq0] "v"(lq0), [lq1] "v"(lq1), [nd0] "v"(z), [nd1] "v"(z),
                     [hi0] "v"(0), [hi1] "v"(0), [hp0] "v"(0), [hp1] "v"(0) : FA2_DQ_CLOBBERS);
    else
        asm volatile(FA2_DQ_PRO_D64_M0 : : FA2_DQ_OPS_64, [c2] "s"(c2), [lq0] "v"(lq0), [lq1] "v"(lq1), [nd0] "v"(z), [nd1] "v"(z),
                     [hi0] "v"(0), [hi1] "v"(0), [hp0] "v"(0), [hp1] "v"(0) : FA2_DQ_CLOBBERS);
}

// The main loop is NOT scheduled by hipcc (as in kernel 2 below): amdgpu_num_vgpr(64) leaves v64..v255 and the whole
// accumulator file to the generated bodies of fa2_bwd_dq_body.inc (tools/gen_dq_body.py: register map and schedule).
// One body = one 64-key tile = 8 KS + 8 DT MFMAs; the two 32-key blocks of a tile are half a tile apart in a software
// pipeline (S^T/dP^T of one block beside the exponentials, products and packs of the other), so the dQ^T products of
// a tile's second block run in the NEXT body, against the K image that is still in the ring of three LDS buffers.
// One barrier per tile, inside the body (vmcnt(0) + s_barrier in front of its first read of the next tile): every wave
// past it has also finished with the tile before the previous one, so its buffer may take the next DMA.
template <int D, bool CAUSAL>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(64))) fa2_bwd_dq_kernel(BwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = D * 2;
    constexpr int TILEB = kDqKV * ROWB;
    constexpr int VRING = 3 * TILEB;          // LDS: [3 K tiles][3 V tiles]; every offset from its address register < 64 KiB
    constexpr int CPR = D / 8;
    constexpr int RPI = 64 / CPR;
    constexpr int NP = kDqKV / RPI;           // DMA pieces per tensor per tile
    constexpr int KS = D / 16;
    constexpr int DT = D / 32;
    // AGPR map (the bodies use the same numbers)
    constexpr int A_DQ = 0;                   // dQ^T tile (qb, dt): a[A_DQ + (qb*DT + dt)*16 ..+15]
    constexpr int A_QF = 128;                 // Q fragment (qb, s): a[A_QF + (qb*KS + s)*4 ..+3]
    constexpr int A_GF = 192;                 // dO fragment (qb, s)
    constexpr int SET1 = D == 128 ? FA2_DQ_D128_SET1 : FA2_DQ_D64_SET1;
    constexpr int ROFFV = D == 128 ? FA2_DQ_D128_ROFFV : FA2_DQ_D64_ROFFV;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31;
    const int h = lane >> 5;
    const int Nq = p.Nq, N = p.Nk;           // N: keys of the block (every key-side bound below)

    const int nrb = (Nq + kBwdRows - 1) / kBwdRows;
    int head, rb;
    map_block(blockIdx.x, p.BH, nrb, head, rb);
    if (CAUSAL) rb = nrb - 1 - rb;

    const size_t slab = (size_t)head * p.q_hs * ROWB;          // Q, dO, dQ
    const size_t kslab = (size_t)head * p.k_hs * ROWB;         // K, V
    const char* Qh = (const char*)p.Q + slab;
    const char* Kh = (const char*)p.K + kslab;
    const char* Vh = (const char*)p.V + kslab;
    const char* Gh = (const char*)p.dO + slab;

    const int q0 = rb * kBwdRows + wave * 64;

    int ntiles = (N + kDqKV - 1) / kDqKV;
    if (CAUSAL) {
        const int last_key = min(rb * kBwdRows + kBwdRows - 1, Nq - 1) + p.causal_shift;     // last visible key
        ntiles = min(ntiles, last_key < 0 ? 0 : last_key / kDqKV + 1);
    }
    // whole triples of tiles (ring of three buffers: every LDS offset is an immediate) and at least one past the real ones,
    // whose body finishes the second key block of the last real tile; the extra tiles are fully masked
    const int niter = ((ntiles + 1 + 2) / 3) * 3;

    // ---- resident operands -> AGPRs; per-row constants
    int qrow[2];
    float Lq[2], Dq[2];
    static_for<2>([&](auto QB) {
        constexpr int qb = decltype(QB)::value;
        qrow[qb] = q0 + 32 * qb + qi;
        const int qld = qrow[qb] < Nq ? qrow[qb] : Nq - 1;
        static_for<KS>([&](auto S) {
            constexpr int sidx = decltype(S)::value;
            acc_write_frag<A_QF + (qb * KS + sidx) * 4>(
                *reinterpret_cast<const bf16x8*>(Qh + (size_t)qld * ROWB + 16 * (2 * sidx + h)));
            acc_write_frag<A_GF + (qb * KS + sidx) * 4>(
                *reinterpret_cast<const bf16x8*>(Gh + (size_t)qld * ROWB + 16 * (2 * sidx + h)));
        });
        static_for<16 * DT>([&](auto R) { acc_write<A_DQ + qb * DT * 16 + decltype(R)::value>(0.0f); });
        Lq[qb] = p.L[(size_t)head * p.q_hs + qld] * kLog2e;
        Dq[qb] = p.D[(size_t)head * p.q_hs + p.q_row0 + qld];
    });
    const float c2 = p.scale * kLog2e;
    // -D of the lane's query row as a whole accumulator tile: the dP^T chains start from it, so
    // dP' = dP - D comes out of the MFMA and dS = P dP' needs no subtraction.
    f32x16 negD[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) negD[qb][r] = -Dq[qb];
    // SET1 (S^T / dP^T of "key block 1 of the tile before the first"): P = exp2(-huge) = 0 and dP' = 0, so the first body's
    // Q1 stage adds exactly zero
    static_for<32>([&](auto R) {
        dq_vset<SET1 + decltype(R)::value>(-1.0e30f);
        dq_vset<SET1 + 32 + decltype(R)::value>(0.0f);
    });

    // ---- LDS-DMA staging (as in fa2_fwd: one per-lane voffset, wave-uniform soffset, range-checked)
    const int drow = lane / CPR;
    const int dslot = lane % CPR;
    const int prow = wave * RPI + drow;
    const int doff = drow * ROWB + 16 * ((lds_off<D>(prow, dslot) - ROWB * prow) >> 4);
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, N * ROWB, 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, N * ROWB, 0x00020000);
    auto stage = [&](int t, int buf) {
        char* b = smem + buf * TILEB;
#pragma unroll
        for (int j = wave; j < 2 * NP; j += kDqWaves) {
            const int which = j / NP, piece = j % NP;
            const int soff = (t * kDqKV + piece * RPI) * ROWB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(which ? v_rsrc : k_rsrc, (lptr_t)(b + which * VRING + piece * 1024),
                                                     16, doff, soff, 0, 0);
        }
    };
    stage(0, 0);
    stage(0, 2);          // the "previous" slot of the first body must hold finite data: tile 0 again

    // ---- loop-invariant per-lane LDS addresses (byte addresses; ring-buffer parts are immediates in the bodies)
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;
    const int trq = (lane & 15) >> 2;
    const int trp = lane & 3;
    const int trcb = (lane >> 4) & 1;
    uint32_t roff[KS], toff[2 * DT];
#pragma unroll
    for (int s = 0; s < KS; ++s) roff[s] = lbase + lds_off<D>(qi, 2 * s + h);            // +32 rows: kb = 1
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)      // +16 rows: sp = 1, +32 rows: kb = 1
            toff[2 * dt + jj] = lbase + lds_off<D>(8 * jj + 4 * h + trq, 4 * dt + 2 * trcb + (trp >> 1)) + 8 * (trp & 1);

    static_for<KS>([&](auto S) {               // row-read addresses of the V ring
        constexpr int sidx = decltype(S)::value;
        dq_vset<ROFFV + sidx>(__uint_as_float(roff[sidx] + VRING));
    });

    __syncthreads();                         // tile 0 has landed (vmcnt(0) inside)
    dq_prologue<D>(roff, toff, c2, Lq[0], Lq[1]);

    auto tile = [&](auto BUF, int t) {
        constexpr int buf = decltype(BUF)::value;
        stage(t + 1, (buf + 1) % 3);         // its buffer was last read two bodies ago, before that body's barrier
        const int key0 = t * kDqKV;
        // this body holds the arithmetic of tile t's first key block and of tile t-1's second one
        bool masked = key0 + kDqKV > N;
        if (CAUSAL) masked = masked || key0 + kDqKV - 1 > q0 + p.causal_shift;
#ifdef FA2_DQ_FORCE_MASKED
        masked = true;                       // diagnostic build: every tile through the masked bodies
#endif
        if (masked) {                        // wave-uniform and rare
            int hi[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
                hi[qb] = (CAUSAL ? min(N, qrow[qb] + p.causal_shift + 1) : N) - key0 - 4 * h;
            dq_body<D, buf, true>(roff, toff, c2, Lq[0], Lq[1], negD[0], negD[1], hi[0], hi[1], hi[0] + kDqKV, hi[1] + kDqKV);
        } else {
            dq_body<D, buf, false>(roff, toff, c2, Lq[0], Lq[1], negD[0], negD[1], 0, 0, 0, 0);
        }
    };
    for (int t = 0; t < niter; t += 3) {
        tile(std::integral_constant<int, 0>{}, t);
        tile(std::integral_constant<int, 1>{}, t + 1);
        tile(std::integral_constant<int, 2>{}, t + 2);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the last body's look-ahead DMA and reads

    mfma_acc_settle();
    static_for<2>([&](auto QB) {
        constexpr int qb = decltype(QB)::value;
        static_for<4 * DT>([&](auto G) {
            constexpr int dt = decltype(G)::value / 4, g = decltype(G)::value % 4;
            constexpr int R = A_DQ + (qb * DT + dt) * 16 + 4 * g;
            bf16x4 o;
            o[0] = (__bf16)(acc_read<R>() * p.scale); o[1] = (__bf16)(acc_read<R + 1>() * p.scale);
            o[2] = (__bf16)(acc_read<R + 2>() * p.scale); o[3] = (__bf16)(acc_read<R + 3>() * p.scale);
            if (qrow[qb] < Nq)
                *reinterpret_cast<bf16x4*>((char*)p.dQ + slab + (size_t)qrow[qb] * ROWB + 2 * (32 * dt + 8 * g + 4 * h)) = o;
        });
    });
}

// --------------------------------------------------------------------------- kernel 2: dK, dV
// Workgroup = 4 waves = 256 keys of one head, ONE wave per SIMD so that each wave may use the whole 512-entry
// register file: a wave owns 64 keys (two 32-key blocks) and keeps dK^T and dV^T of those keys -- 2 x 2 x (D/32)
// accumulator tiles = 256 registers at D = 128 -- plus its K fragments resident in VGPRs.  V of the workgroup's 256
// keys sits in LDS (read as the B operand of dP); Q/dO tiles of 64 query rows stream through a double-buffered LDS
// image by LDS-DMA (source address pre-swizzled so the linear LDS write produces the swizzled image), the row
// constants (pre-transformed by kernel 0) with them.
//
// The main loop is NOT scheduled by hipcc.  The kernel is compiled with amdgpu_num_vgpr(60): hipcc may allocate
// v0..v59 only (loop control, DMA issue, the epilogue's temporaries, the loop-invariant LDS addresses it passes
// in as operands); v60..v255 and the whole accumulator file are named by the generated bodies of
// fa2_bwd_dkdv_body.inc (tools/gen_dkdv_body.py: register map, stage order, per-gap issue budget).  One body =
// one 32-row sub-tile = 4 KS + 8 DT MFMAs with every LDS read, wait, exponential, product and pack assigned to
// an MFMA gap; bodies chain cyclically (each ends with the first reads of the next), so the matrix pipe does
// not drain at sub-tile or tile seams.  Two barriers per 64-row tile: B1 at its start (everyone is done with the
// previous tile's buffer -> the DMA of the next tile may overwrite it) and B2 inside the body of sub-tile 1
// (vmcnt(0) + barrier: the next tile has landed) in front of its first read of the other buffer.
// Masking (sequence tail, causal diagonal) is a second body variant that zeroes P behind each exp; dS then uses
// v_mul_legacy (0 x anything = 0), so rows past the end contribute nothing whatever their row constants hold.
constexpr int kDkWaves = 4;
constexpr int kDkKeys = 64 * kDkWaves;      // keys per workgroup

#include "fa2_bwd_dkdv_body.inc"

// One literal VGPR move / accumulator access per call; the clobber of v255 is what makes the kernel descriptor
// allocate the registers the bodies name (a reserved register to hipcc: it never allocates it).
template <int R>
__device__ __forceinline__ void dkdv_vset(uint32_t x)
{
    asm volatile("v_mov_b32 v%c1, %0" : : "v"(x), "i"(R) : "v255");
}

#define FA2_DKDV_CLOBBERS "memory", "vcc", "s10", "s11", "v255", FA2_ACC_CLOBBERS

template <int D, int BUF, int SH, bool MASKED>
__device__ __forceinline__ void dkdv_body(const uint32_t (&roff)[D / 16], const uint32_t (&toff)[D / 16], uint32_t rc, float c2,
                                          int hi, int lo0, int lo1)
{
#define FA2_DKDV_CASE(DD, B, S, M)                                                                                             \
    if constexpr (D == DD && BUF == B && SH == S && MASKED == bool(M))                                                           \
        asm volatile(FA2_DKDV_BODY_D##DD##_B##B##_S##S##_M##M : : FA2_DKDV_OPS_##DD, [rc] "v"(rc), [c2] "s"(c2), [hi] "v"(hi),    \
                     [lo0] "v"(lo0), [lo1] "v"(lo1) : FA2_DKDV_CLOBBERS);
    FA2_DKDV_CASE(128, 0, 0, 0) FA2_DKDV_CASE(128, 0, 1, 0) FA2_DKDV_CASE(128, 1, 0, 0) FA2_DKDV_CASE(128, 1, 1, 0)
    FA2_DKDV_CASE(128, 0, 0, 1) FA2_DKDV_CASE(128, 0, 1, 1) FA2_DKDV_CASE(128, 1, 0, 1) FA2_DKDV_CASE(128, 1, 1, 1)
    FA2_DKDV_CASE(64, 0, 0, 0) FA2_DKDV_CASE(64, 0, 1, 0) FA2_DKDV_CASE(64, 1, 0, 0) FA2_DKDV_CASE(64, 1, 1, 0)
    FA2_DKDV_CASE(64, 0, 0, 1) FA2_DKDV_CASE(64, 0, 1, 1) FA2_DKDV_CASE(64, 1, 0, 1) FA2_DKDV_CASE(64, 1, 1, 1)
#undef FA2_DKDV_CASE
}

template <int D>
__device__ __forceinline__ void dkdv_prologue(const uint32_t (&roff)[D / 16], const uint32_t (&toff)[D / 16], uint32_t rc)
{
    // the early reads of the first sub-tile (buffer 0, sub-tile 0); the masked and plain variants issue the same ones
    if constexpr (D == 128) asm volatile(FA2_DKDV_PRO_D128_M0 : : FA2_DKDV_OPS_128, [rc] "v"(rc) : FA2_DKDV_CLOBBERS);
    else asm volatile(FA2_DKDV_PRO_D64_M0 : : FA2_DKDV_OPS_64, [rc] "v"(rc) : FA2_DKDV_CLOBBERS);
}

template <int D, bool CAUSAL>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(60))) fa2_bwd_dkdv_kernel(BwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = D * 2;
    constexpr int TROWS = 64;                   // rows per DMA tile: two 32-row sub-tiles
    constexpr int TILEB = TROWS * ROWB;         // Q (or dO) tile
    constexpr int BUFB = 2 * TILEB + 512;       // Q tile, dO tile, 64 x (-L/scale), 64 x (-D)
    constexpr int CPR = D / 8;                  // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;               // rows one DMA wave-instruction covers (1 KiB)
    constexpr int NINS = TROWS / RPI;           // DMA instructions per tensor per tile: 16 or 8
    constexpr int KS = D / 16;
    constexpr int DT = D / 32;
    constexpr int KF = D == 128 ? FA2_DKDV_D128_KF : FA2_DKDV_D64_KF;
    constexpr int ROFFV = D == 128 ? FA2_DKDV_D128_ROFFV : FA2_DKDV_D64_ROFFV;
    constexpr int A_DK = D == 128 ? FA2_DKDV_D128_A_DK : FA2_DKDV_D64_A_DK;
    constexpr int A_DV = D == 128 ? FA2_DKDV_D128_A_DV : FA2_DKDV_D64_A_DV;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ki = lane & 31;
    const int h = lane >> 5;
    const int N = p.Nk, Nq = p.Nq;          // N: keys of the block

    const int ncb = (N + kDkKeys - 1) / kDkKeys;
    int head, cb;
    map_block(blockIdx.x, p.BH, ncb, head, cb);     // causal: key block 0 is the heaviest, already first

    const size_t slab = (size_t)head * p.k_hs * ROWB;          // K, V, dK, dV
    const size_t qslab = (size_t)head * p.q_hs * ROWB;         // Q, dO
    const char* Qh = (const char*)p.Q + qslab;
    const char* Kh = (const char*)p.K + slab;
    const char* Vh = (const char*)p.V + slab;
    const char* Gh = (const char*)p.dO + qslab;
    const size_t rc_plane = (size_t)p.BH * p.q_hs;

    const int kw0 = cb * kDkKeys + wave * 64;       // first key of this wave

    char* const bufs = smem;                         // [2][Q tile | dO tile | row constants]
    char* const Vimg = smem + 2 * BUFB;              // after them: every tile offset fits a 16-bit immediate

    // K fragments of both key blocks: B operands of S' = Q K^T (lane = key column) -> v[KF ...]
    static_for<2>([&](auto KB) {
        constexpr int kb = decltype(KB)::value;
        int kr = kw0 + 32 * kb + ki;
        kr = kr < N ? kr : N - 1;
        static_for<KS>([&](auto S) {
            constexpr int sidx = decltype(S)::value;
            const u32x4 w = *reinterpret_cast<const u32x4*>(Kh + (size_t)kr * ROWB + 16 * (2 * sidx + h));
            dkdv_vset<KF + 4 * (kb * KS + sidx) + 0>(w[0]);
            dkdv_vset<KF + 4 * (kb * KS + sidx) + 1>(w[1]);
            dkdv_vset<KF + 4 * (kb * KS + sidx) + 2>(w[2]);
            dkdv_vset<KF + 4 * (kb * KS + sidx) + 3>(w[3]);
        });
    });
    // V image: the workgroup's 256 keys, swizzled like every other tile.
    for (int c = tid; c < kDkKeys * CPR; c += 256) {
        const int row = c / CPR, ch = c % CPR;
        int kr = cb * kDkKeys + row;
        kr = kr < N ? kr : N - 1;
        *reinterpret_cast<u32x4*>(Vimg + lds_off<D>(row, ch)) =
            *reinterpret_cast<const u32x4*>(Vh + (size_t)kr * ROWB + 16 * ch);
    }
    static_for<32 * DT>([&](auto R) {
        acc_write<A_DK + decltype(R)::value>(0.0f);
        acc_write<A_DV + decltype(R)::value>(0.0f);
    });

    const int ntiles = (Nq + TROWS - 1) / TROWS;
    int t0 = 0;
    if (CAUSAL) t0 = min(ntiles, max(0, cb * kDkKeys - p.causal_shift) / TROWS);      // earlier query rows see none of these keys
    // Always whole pairs of tiles (the loop is unrolled by two: every LDS offset is an immediate).  An odd count is
    // padded with one tile past the sequence end: all of its rows are masked.
    const int tend = t0 + ((ntiles - t0 + 1) & ~1);

    const float c2 = p.scale * kLog2e;

    // ---- LDS-DMA staging.  Wave w issues pieces w, w+4, ... of the 2*NINS pieces of a tile; a piece is RPI rows =
    // 1 KiB written linearly, lane l -> row l / CPR, slot l % CPR; the slot must hold chunk (slot ^ f(row)), f being
    // lds_off's swizzle: so the SOURCE address is permuted, the LDS side stays linear.  The swizzle term depends on the
    // row modulo 16 only, so ONE per-lane byte offset (voffset) serves every DMA of the wave; the tile/piece part is
    // wave-uniform (soffset) and the slab is described by a buffer resource whose range check turns rows past the end
    // of the sequence into zeros.
    const int drow = lane / CPR;
    const int dslot = lane % CPR;
    const int prow = wave * RPI + drow;                                   // row inside the tile (first piece)
    const int doff = drow * ROWB + 16 * ((lds_off<D>(prow, dslot) - ROWB * prow) >> 4);
    const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Qh, 0, Nq * ROWB, 0x00020000);
    const auto g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Gh, 0, Nq * ROWB, 0x00020000);
    const auto rc_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.RC, 0, (int)(2 * rc_plane * 4), 0x00020000);
    const int rcoff = (int)(((wave == 0 ? 0 : rc_plane) + (size_t)head * p.q_hs + p.q_row0 + lane) * 4);
    auto stage = [&](int t, int buf) {
        char* b = bufs + buf * BUFB;
#pragma unroll
        for (int j = wave; j < 2 * NINS; j += kDkWaves) {
            const int which = j / NINS, piece = j % NINS;
            const int soff = (t * TROWS + piece * RPI) * ROWB;            // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(which ? g_rsrc : q_rsrc, (lptr_t)(b + which * TILEB + piece * 1024),
                                                     16, doff, soff, 0, 0);
        }
        if (wave < 2)                                // row constants: wave 0 the 64 x -L/scale, wave 1 the 64 x -D
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rc_rsrc, (lptr_t)(b + 2 * TILEB + wave * 256), 4, rcoff, t * TROWS * 4, 0, 0);
    };

    if (t0 < tend) stage(t0, 0);                     // tile t lives in buffer (t - t0) & 1

    // ---- loop-invariant per-lane LDS addresses (byte addresses; buffer / sub-tile parts are immediates in the bodies)
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;
    const int trq = (lane & 15) >> 2;
    const int trp = lane & 3;
    const int trcb = (lane >> 4) & 1;
    uint32_t roff[KS], toff[2 * DT];
#pragma unroll
    for (int s = 0; s < KS; ++s) roff[s] = lbase + lds_off<D>(ki, 2 * s + h);                 // A-operand row = query = lane & 31
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)                                                        // +16 rows: sp = 1
            toff[2 * dt + jj] = lbase + lds_off<D>(8 * jj + 4 * h + trq, 4 * dt + 2 * trcb + (trp >> 1)) + 8 * (trp & 1);
    const uint32_t rcadr = lbase + 2 * TILEB + 16 * h;          // rows 8g + 4h .. of a constant plane (the bodies add buffer / sub-tile)
    // this wave's 64 V rows (same swizzle phase as row ki)
    static_for<KS>([&](auto S) {
        constexpr int sidx = decltype(S)::value;
        dkdv_vset<ROFFV + sidx>(roff[sidx] + 2 * BUFB + 64 * wave * ROWB);
    });

    __syncthreads();                                 // V image written, first tile landed (vmcnt(0) inside)
    if (t0 < tend) dkdv_prologue<D>(roff, toff, rcadr);

    auto tile = [&](auto BUF, int t) {
        constexpr int buf = decltype(BUF)::value;
        __builtin_amdgcn_s_barrier();                // B1: nobody reads the other buffer any more
        if (t + 1 < tend) stage(t + 1, buf ^ 1);
        const int q0t = t * TROWS;
        bool masked = q0t + TROWS > Nq;
        if (CAUSAL) masked = masked || q0t < kw0 + 63 - p.causal_shift;
        if (masked) {                                // wave-uniform and rare
            const int hi = Nq - q0t - 4 * h;
            const int lo0 = CAUSAL ? kw0 + ki - p.causal_shift - q0t - 4 * h : -(1 << 30);
            const int lo1 = CAUSAL ? lo0 + 32 : lo0;
            dkdv_body<D, buf, 0, true>(roff, toff, rcadr, c2, hi, lo0, lo1);
            dkdv_body<D, buf, 1, true>(roff, toff, rcadr, c2, hi, lo0, lo1);
        } else {
            dkdv_body<D, buf, 0, false>(roff, toff, rcadr, c2, 0, 0, 0);
            dkdv_body<D, buf, 1, false>(roff, toff, rcadr, c2, 0, 0, 0);
        }
    };

    for (int t = t0; t < tend; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        tile(std::integral_constant<int, 1>{}, t + 1);
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the last body's look-ahead reads
    mfma_acc_settle();
    static_for<2>([&](auto KB) {
        constexpr int kb = decltype(KB)::value;
        const int key = kw0 + 32 * kb + ki;
        char* dKk = (char*)p.dK + slab + (size_t)(key < N ? key : 0) * ROWB;
        char* dVk = (char*)p.dV + slab + (size_t)(key < N ? key : 0) * ROWB;
        static_for<4 * DT>([&](auto G) {
            constexpr int dt = decltype(G)::value / 4, g = decltype(G)::value % 4;
            constexpr int RK = A_DK + 16 * (kb * DT + dt) + 4 * g, RV = A_DV + 16 * (kb * DT + dt) + 4 * g;
            bf16x4 a, b;
            a[0] = (__bf16)(acc_read<RK>() * p.scale); a[1] = (__bf16)(acc_read<RK + 1>() * p.scale);
            a[2] = (__bf16)(acc_read<RK + 2>() * p.scale); a[3] = (__bf16)(acc_read<RK + 3>() * p.scale);
            b[0] = (__bf16)acc_read<RV>(); b[1] = (__bf16)acc_read<RV + 1>();
            b[2] = (__bf16)acc_read<RV + 2>(); b[3] = (__bf16)acc_read<RV + 3>();
            if (key < N) {
                *reinterpret_cast<bf16x4*>(dKk + 2 * (32 * dt + 8 * g + 4 * h)) = a;
                *reinterpret_cast<bf16x4*>(dVk + 2 * (32 * dt + 8 * g + 4 * h)) = b;
            }
        });
    });
}

// --------------------------------------------------------------------------- launch
template <int D, bool CAUSAL>
static hipError_t launch_bwd_one(const BwdArgs& a, hipStream_t stream)
{
    const size_t rows = (size_t)a.BH * a.Nq;
    hipError_t e = hipSuccess;
    if (a.phases & 1) {
        hipLaunchKernelGGL((fa2_bwd_delta_kernel<D>), dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0,
                           stream, (const __bf16*)a.dO, (const __bf16*)a.O, a.L, a.D, a.RC, rows, a.Nq, a.q_hs, a.q_row0,
                           (size_t)a.BH * a.q_hs, 1.0f / a.scale);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const int nb = (a.Nq + kBwdRows - 1) / kBwdRows;
    constexpr int lds_dq = 3 * 2 * kDqKV * D * 2;      // ring of three (K tile | V tile)
    constexpr int lds_dk = kDkKeys * D * 2 + 2 * (2 * 2 * kDkQ * D * 2 + 512);
    static bool set_dq[64] = {}, set_dk[64] = {};
    e = ensure_dynamic_lds(fa2_bwd_dq_kernel<D, CAUSAL>, lds_dq, set_dq);
    if (e != hipSuccess) return e;
    e = ensure_dynamic_lds(fa2_bwd_dkdv_kernel<D, CAUSAL>, lds_dk, set_dk);
    if (e != hipSuccess) return e;
    if (a.phases & 2) {
        hipLaunchKernelGGL((fa2_bwd_dq_kernel<D, CAUSAL>), dim3((unsigned)(nb * a.BH)), dim3(256), lds_dq, stream, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (a.phases & 4) {
        const int ncb = (a.Nk + kDkKeys - 1) / kDkKeys;
        hipLaunchKernelGGL((fa2_bwd_dkdv_kernel<D, CAUSAL>), dim3((unsigned)(ncb * a.BH)), dim3(256), lds_dk, stream, a);
        e = hipGetLastError();
    }
    return e;
}

hipError_t launch_bwd_bf16(const BwdArgs& a, hipStream_t stream)
{
    if (a.d == 128) return a.causal ? launch_bwd_one<128, true>(a, stream) : launch_bwd_one<128, false>(a, stream);
    if (a.d == 64) return a.causal ? launch_bwd_one<64, true>(a, stream) : launch_bwd_one<64, false>(a, stream);
    return hipErrorInvalidValue;
}

}  // namespace fa2
