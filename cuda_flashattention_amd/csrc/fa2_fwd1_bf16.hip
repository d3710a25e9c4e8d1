// fa2_fwd1_bf16.hip -- FlashAttention-2 forward for gfx950, bf16 in / fp32 accumulate, ONE wave per SIMD with a generated
// main loop.  Replaces flash_attention_2_forward_kernel (reference src/02_flash_attention_v2_forward/
// flash_attention_kernel.cu:37-297) and, through the resume / finalize switches, ring_attention_forward_kernel
// (src/03_flash_attention_v2_ring/common/ring_attention_kernel.cu:13-140).  Same maths as fa2_fwd_bf16.hip (the two-waves-
// per-SIMD kernel of rounds 1-2, whose stages were hand-written asm with hipcc glue between them): S^T = K Q^T so that a
// row's max and sum are in-lane reductions, P stays in registers, lazy softmax reference, exp2 domain, natural-log L.
//
// What is different (round 3; tools/probes/softmax_port.hip, profiles/r3_*): the forward is co-limited by VALU issue (4 VALU
// instructions per S element: fma, exp, add, half a max3 and half a pack) and by the MFMA pipe; with two waves per SIMD
// hipcc-scheduled stages put 8 of the 64 VALU instructions of a 32-key step beside the 8 S^T products and 56 beside the 8 PV
// products, and both waves of a SIMD want the VALU at the same time.  Here a workgroup is 4 waves = 256 query rows, one
// wave per SIMD (512 registers), a wave owns 64 rows (two 32-row blocks that share every K and V^T fragment read: 0.75 LDS
// reads per MFMA instead of 1.5), and the main loop is ONE asm body per 32-key block produced by tools/gen_fwd_body.py:
//   A  S^T(j)  = K(j) Q^T               P  O^T += V^T(j-2) P^T(j-2)            VALU: softmax of block j-1, maxima of block j
// -- every VALU instruction has a whole body of independent MFMAs to hide behind, every LDS read is issued 4-7 MFMAs ahead
// of its use behind a counted lgkmcnt, the next tile's LDS-DMA is part of the body.  hipcc owns v0..v63 only
// (amdgpu_num_vgpr(64)); v64.. and the accumulator file are named by the bodies.
//
// The softmax reference is lazy (as before): a body ends with "does any lane's maximum exceed its threshold", returned in
// an SGPR; the rare update is compiler code between two bodies.  With the pipeline two blocks deep, at that point O^T holds
// the products through block j-2, the row sums through block j-1, and P(j-1) is packed and waiting: the sums are rescaled
// at once, O^T one body later (after P(j-1), formed at the old reference, has been added to it).
//
// Lane maxima in the first tile only (round 3, second half).  The running maximum of the online softmax does two jobs: it
// gives the exponentials a reference, and it keeps them in range.  In fp32 sums and bf16 P -- whose relative precision does
// not depend on magnitude -- a reference that lags the true maximum by tens of units costs nothing: O / l comes out the same
// to rounding.  So a workgroup takes the lane maxima during its first tile (which establishes the reference, with the
// lazy update above) and then runs the X variant of the bodies -- no v_max3_f32 (16 of a body-pair's ~146
// VALU instructions), no compare.  What the maxima guarded against, a score more than ~60 above the reference, shows in the row
// sums.  Between two rounds of the ring a row whose sum has passed 2^30 moves its reference up by 64 ln 2 (a slow rise of
// any size costs nothing); and if any row of the workgroup ends with l not below 2^80 (or not finite), or was seen there by
// one of those moves -- a jump of more than ~48 units inside one round -- the workgroup runs its row block again with the
// maxima in every body (`safe`), as before.  (What stays outside the contract: |V| beyond ~2^47, where P < 2^80 against a
// lagging reference can take P V out of fp32 although O itself would fit.)
// Tail, diagonal and drain bodies keep their maxima either way.
// Measured same-box: (4,16,4096,64) 945 -> ~990 TFLOP/s, (4,16,8192,128) 1.692 -> ~1.65 ms.
#include <type_traits>

#include "fa2_common.h"
#include "fa2_launch.h"

namespace fa2 {

#include "fa2_fwd_body.inc"

constexpr int kF1Rows = 256;                 // query rows per workgroup (4 waves x 64 or 8 waves x 32)
constexpr int kF1Bufs = 4;                  // LDS ring depth (tools/gen_fwd_body.py: NBUF)
constexpr float kF1RescaleThr = 6.0f;       // natural-log units of the scaled score
constexpr float kF1SumLimit = 1.2089258e24f; // 2^80: a row sum at or above it sends the workgroup through its row block again, with maxima
constexpr float kF1LiftAt = 1.0737418e9f;    // 2^30: between two rounds without maxima, a row whose sum has passed it moves its reference up ...
constexpr float kF1LiftBy = 44.3614196f;     // ... by 64 ln 2 (its sums and O^T scale by 2^-64)
enum { F1_PLAIN = 0, F1_MASKED = 1, F1_NOMAX = 2 };

typedef __attribute__((address_space(3))) void* f1_lptr_t;

// Two shapes of the same kernel (tools/gen_fwd_body.py: CONFIGS):
//   QBS = 2  one wave per SIMD: 4 waves x 64 rows, 512 registers per wave (hipcc: v0..v63), a[0:256)
//   QBS = 1  two waves per SIMD: 8 waves x 32 rows, 128 + 128 registers per wave (hipcc: v0..v39), a[0:128) -- for d = 64,
//            where the VALU (4 instructions per S element against half the MFMAs of d = 128) is the bound and two waves
//            issue VALU instructions at ~4.4 clocks each against ~7 for one wave alone (tools/probes/softmax_port.hip)
#define FA2_ACC128_LIST                                                                                                        \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", \
    "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37",     \
    "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55",     \
    "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73",     \
    "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91",     \
    "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108",   \
    "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124",   \
    "a125", "a126", "a127"
#define FA2_F1_REGS2 "v255", FA2_ACC_CLOBBERS
#define FA2_F1_REGS1 "v127", FA2_ACC128_LIST
#define FA2_F1_MISC "memory", "vcc", "scc", "s10", "s11", "s12", "m0"

// every asm statement below exists twice: with the register-file clobbers of the one-wave and of the two-wave shape
#define FA2_F1_ASM(QBS, TEXT, OUTS, INS)                                                  \
    do {                                                                                   \
        if constexpr (QBS == 2) asm volatile(TEXT : OUTS : INS : FA2_F1_MISC, FA2_F1_REGS2); \
        else asm volatile(TEXT : OUTS : INS : FA2_F1_MISC, FA2_F1_REGS1);                  \
    } while (0)
#define FA2_F1_COMMA ,

template <int QBS, int R>
__device__ __forceinline__ void f1_vset(uint32_t x)
{
    FA2_F1_ASM(QBS, "v_mov_b32 v%c1, %0", , "v"(x) FA2_F1_COMMA "i"(R));
}
template <int QBS, int R>
__device__ __forceinline__ void f1_vsetf(float x)
{
    FA2_F1_ASM(QBS, "v_mov_b32 v%c1, %0", , "v"(x) FA2_F1_COMMA "i"(R));
}
template <int R>
__device__ __forceinline__ float f1_vget()
{
    float x;
    asm volatile("v_mov_b32 %0, v%c1" : "=v"(x) : "i"(R));
    return x;
}
template <int QBS, int R>
__device__ __forceinline__ void f1_awrite(float x)
{
    FA2_F1_ASM(QBS, "v_accvgpr_write_b32 a[%c1], %0", , "v"(x) FA2_F1_COMMA "i"(R));
}
template <int R>
__device__ __forceinline__ float f1_aread()
{
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(R));
    return x;
}
template <int QBS, int LO>
__device__ __forceinline__ void f1_awrite_frag(bf16x8 f)
{
    const u32x4 w = __builtin_bit_cast(u32x4, f);
    FA2_F1_ASM(QBS, "v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\t"
                    "v_accvgpr_write_b32 a[%c7], %3", ,
               "v"(w[0]) FA2_F1_COMMA "v"(w[1]) FA2_F1_COMMA "v"(w[2]) FA2_F1_COMMA "v"(w[3]) FA2_F1_COMMA "i"(LO) FA2_F1_COMMA "i"(LO + 1)
                   FA2_F1_COMMA "i"(LO + 2) FA2_F1_COMMA "i"(LO + 3));
}
// four accumulator registers *= alpha (per lane = per query row)
template <int QBS, int R>
__device__ __forceinline__ void f1_scale4(float alpha)
{
    float t0, t1, t2, t3;
    FA2_F1_ASM(QBS, "v_accvgpr_read_b32 %0, a[%c5]\n\tv_accvgpr_read_b32 %1, a[%c6]\n\t"
                    "v_accvgpr_read_b32 %2, a[%c7]\n\tv_accvgpr_read_b32 %3, a[%c8]\n\t"
                    "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4\n\t"
                    "v_accvgpr_write_b32 a[%c5], %0\n\tv_accvgpr_write_b32 a[%c6], %1\n\t"
                    "v_accvgpr_write_b32 a[%c7], %2\n\tv_accvgpr_write_b32 a[%c8], %3",
               "=&v"(t0) FA2_F1_COMMA "=&v"(t1) FA2_F1_COMMA "=&v"(t2) FA2_F1_COMMA "=&v"(t3),
               "v"(alpha) FA2_F1_COMMA "i"(R) FA2_F1_COMMA "i"(R + 1) FA2_F1_COMMA "i"(R + 2) FA2_F1_COMMA "i"(R + 3));
}
template <int QBS, int R>
__device__ __forceinline__ void f1_acc_zero(u32x4 z)
{
    FA2_F1_ASM(QBS, "s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, %0, 0", , "v"(z) FA2_F1_COMMA "i"(R) FA2_F1_COMMA "i"(R + 15));
}

struct F1Dma {
    __amdgpu_buffer_rsrc_t krs, vrs;
    uint32_t mw, dvo, kso;
};

// One body.  KB < NH - 1: plain; KB == NH - 1: starts with the tile barrier and issues the DMA of tile t + 2.  MODE: with
// maxima (F1_PLAIN), with maxima and the mask (F1_MASKED), without maxima (F1_NOMAX: `need` comes back 0).
template <int D, int QBS, int BUF, int KB, int MODE>
__device__ __forceinline__ void f1_body(float c2, int& need, const int (&hi)[2], const F1Dma& dma)
{
#define FA2_F1_CASE(TAG, DD, QQ, B, K, MV, M)                                                                                      \
    if constexpr (D == DD && QBS == QQ && BUF == B && KB == K && MODE == MV)                                                         \
        FA2_F1_ASM(QQ, FA2_FWD_BODY_##TAG##_B##B##_K##K##_##M, [need] "=&s"(need),                                                     \
                   [c2] "s"(c2) FA2_F1_COMMA [hi0] "v"(hi[0]) FA2_F1_COMMA [hi1] "v"(hi[1]) FA2_F1_COMMA [ninf] "v"(-INFINITY)           \
                       FA2_F1_COMMA [mw] "s"(dma.mw) FA2_F1_COMMA [dvo] "v"(dma.dvo) FA2_F1_COMMA [krs] "s"(dma.krs)                     \
                       FA2_F1_COMMA [vrs] "s"(dma.vrs) FA2_F1_COMMA [kso] "s"(dma.kso));
#define FA2_F1_CASES_B(TAG, DD, QQ, K, MV, M) \
    FA2_F1_CASE(TAG, DD, QQ, 0, K, MV, M) FA2_F1_CASE(TAG, DD, QQ, 1, K, MV, M) FA2_F1_CASE(TAG, DD, QQ, 2, K, MV, M) FA2_F1_CASE(TAG, DD, QQ, 3, K, MV, M)
#define FA2_F1_CASES_V(TAG, DD, QQ, K) \
    FA2_F1_CASES_B(TAG, DD, QQ, K, F1_PLAIN, M0) FA2_F1_CASES_B(TAG, DD, QQ, K, F1_MASKED, M1) FA2_F1_CASES_B(TAG, DD, QQ, K, F1_NOMAX, X)
    FA2_F1_CASES_V(D128Q2, 128, 2, 0) FA2_F1_CASES_V(D128Q2, 128, 2, 1)
    FA2_F1_CASES_V(D64Q2, 64, 2, 0) FA2_F1_CASES_V(D64Q2, 64, 2, 1) FA2_F1_CASES_V(D64Q2, 64, 2, 2) FA2_F1_CASES_V(D64Q2, 64, 2, 3)
    FA2_F1_CASES_V(D64Q1, 64, 1, 0) FA2_F1_CASES_V(D64Q1, 64, 1, 1) FA2_F1_CASES_V(D64Q1, 64, 1, 2) FA2_F1_CASES_V(D64Q1, 64, 1, 3)
#undef FA2_F1_CASES_V
#undef FA2_F1_CASES_B
#undef FA2_F1_CASE
}

template <int D, int QBS>
__device__ __forceinline__ void f1_prologue()
{
    if constexpr (D == 128) FA2_F1_ASM(2, FA2_FWD_PRO_D128Q2, , );
    else if constexpr (QBS == 2) FA2_F1_ASM(2, FA2_FWD_PRO_D64Q2, , );
    else FA2_F1_ASM(1, FA2_FWD_PRO_D64Q1, , );
}

// generator constants of a configuration
template <int D, int QBS> struct F1Map;
template <> struct F1Map<128, 2> {
    static constexpr int KV = FA2_FWD_D128Q2_KV, SET0 = FA2_FWD_D128Q2_SET0, SET1 = FA2_FWD_D128Q2_SET1, PF0 = FA2_FWD_D128Q2_PF0,
                         ROFF = FA2_FWD_D128Q2_ROFF, TOFFV = FA2_FWD_D128Q2_TOFFV, ST = FA2_FWD_D128Q2_STATE, V0 = FA2_FWD_D128Q2_V0,
                         A_QF = FA2_FWD_D128Q2_A_QF;
};
template <> struct F1Map<64, 2> {
    static constexpr int KV = FA2_FWD_D64Q2_KV, SET0 = FA2_FWD_D64Q2_SET0, SET1 = FA2_FWD_D64Q2_SET1, PF0 = FA2_FWD_D64Q2_PF0,
                         ROFF = FA2_FWD_D64Q2_ROFF, TOFFV = FA2_FWD_D64Q2_TOFFV, ST = FA2_FWD_D64Q2_STATE, V0 = FA2_FWD_D64Q2_V0,
                         A_QF = FA2_FWD_D64Q2_A_QF;
};
template <> struct F1Map<64, 1> {
    static constexpr int KV = FA2_FWD_D64Q1_KV, SET0 = FA2_FWD_D64Q1_SET0, SET1 = FA2_FWD_D64Q1_SET1, PF0 = FA2_FWD_D64Q1_PF0,
                         ROFF = FA2_FWD_D64Q1_ROFF, TOFFV = FA2_FWD_D64Q1_TOFFV, ST = FA2_FWD_D64Q1_STATE, V0 = FA2_FWD_D64Q1_V0,
                         A_QF = FA2_FWD_D64Q1_A_QF;
};

template <int D, int QBS, bool CAUSAL, bool STATE>
__device__ __forceinline__ void fa2_fwd1_impl(const FwdArgs& p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = D * 2;
    using M = F1Map<D, QBS>;
    constexpr int kF1Waves = 8 / QBS;                                    // 4 waves x 64 rows or 8 waves x 32 rows
    constexpr int WROWS = 32 * QBS;                                      // query rows per wave
    constexpr int KV = M::KV;                                            // keys per tile
    constexpr int NH = KV / 32;                                          // key blocks (bodies) per tile
    constexpr int TILEB = KV * ROWB;                                     // 16 KiB either way
    constexpr int KRING = kF1Bufs * TILEB;                               // LDS: [4 K tiles][4 V tiles]
    constexpr int CPR = D / 8;
    constexpr int RPI = 64 / CPR;                                        // rows per 1-KiB DMA piece
    constexpr int NP = KV / RPI;                                         // pieces per tensor per tile (16)
    constexpr int KS = D / 16, DT = D / 32;
    constexpr int A_O = 0, A_QF = M::A_QF;
    constexpr int SET0 = M::SET0, SET1 = M::SET1, PF0 = M::PF0, ROFF = M::ROFF, TOFFV = M::TOFFV;
    constexpr int ST = M::ST;                                            // l (2 per row block) | rm | mb | th (1 per row block each)
    constexpr int ST_RM = ST + 2 * QBS, ST_MB = ST + 3 * QBS, ST_TH = ST + 4 * QBS;
    static_assert(NP % kF1Waves == 0 && TILEB == 16384, "DMA piece split");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31;
    const int h = lane >> 5;

    // Causal (plain launches; the ring's resumable steps keep one row block per workgroup): a workgroup takes TWO row blocks,
    // nrb - 1 - i and i -- every workgroup the same number of key blocks.  With one row block per workgroup, heaviest first
    // within a head, the last head of every XCD still ends with its heaviest blocks (a quarter of a CU's whole share each)
    // started late: 16 % between a launch's first and last finishing CU (round 3).  The pairs keep a head's workgroups
    // together on one XCD (K / V through its L2), as before.
    constexpr bool PAIRED = CAUSAL && !STATE;
    const int nrb = (p.Nq + kF1Rows - 1) / kF1Rows;
    const int nwg = PAIRED ? (nrb + 1) / 2 : nrb;          // workgroups per head
    int head, wi;
    map_block(blockIdx.x, p.BH, nwg, head, wi);
#pragma nounroll
    for (int half = 0; half < (PAIRED ? 2 : 1); ++half) {
    int rb = CAUSAL ? nrb - 1 - wi : wi;              // (not paired) heaviest row blocks first
    if constexpr (PAIRED) {
        if (half == 1) {
            if (wi == nrb - 1 - wi) break;            // odd count: the middle block is its own pair
            rb = wi;
            __syncthreads();                          // every wave is done with the first block's last tiles (LDS is re-staged)
        }
    }

    const int Nq = p.Nq, Nk = p.Nk;
    const size_t qhs = p.q_hs ? p.q_hs : Nq, khs = p.k_hs ? p.k_hs : Nk;
    const char* Qh = (const char*)p.Q + (size_t)head * qhs * ROWB;
    const char* Kh = (const char*)p.K + (size_t)head * khs * ROWB;
    const char* Vh = (const char*)p.V + (size_t)head * khs * ROWB;
    const int q0 = rb * kF1Rows + wave * WROWS;           // first query row of this wave

    // key blocks that hold a visible key for some row of the WORKGROUP (the tile barriers need every wave in every body);
    // two more bodies drain the pipeline (their S^T is masked completely: P = 0)
    int J = (Nk + 31) / 32;
    if (CAUSAL) {
        const int last_key = min(rb * kF1Rows + kF1Rows - 1, Nq - 1) + p.causal_shift;
        J = min(J, last_key < 0 ? 0 : last_key / 32 + 1);
    }
    const int JB = J + 2;

    // ---- LDS-DMA staging (per-lane source offset pre-swizzled, wave-uniform soffset, rows >= Nk read as zeros)
    const int drow = lane / CPR, dslot = lane % CPR;
    const int prow = wave * RPI + drow;
    const int doff = drow * ROWB + 16 * ((lds_off<D>(prow, dslot) - ROWB * prow) >> 4);
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, Nk * ROWB, 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, Nk * ROWB, 0x00020000);
    auto stage = [&](int t, int buf) {
        char* b = smem + buf * TILEB;
#pragma unroll
        for (int j = wave; j < 2 * NP; j += kF1Waves) {
            const int which = j / NP, piece = j % NP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(which ? v_rsrc : k_rsrc, (f1_lptr_t)(b + which * KRING + piece * 1024), 16, doff,
                                                     (t * KV + piece * RPI) * ROWB, 0, 0);
        }
    };

    // ---- Q fragments -> AGPRs; lane holds Q[q][16 s + 8 h .. +7] of rows q0 + 32 qb + qi
    int qrow[QBS];
    static_for<QBS>([&](auto QB) {
        constexpr int qb = decltype(QB)::value;
        qrow[qb] = q0 + 32 * qb + qi;
        const int qld = qrow[qb] < Nq ? qrow[qb] : Nq - 1;
        static_for<KS>([&](auto S) {
            constexpr int sidx = decltype(S)::value;
            f1_awrite_frag<QBS, A_QF + (qb * KS + sidx) * 4>(*reinterpret_cast<const bf16x8*>(Qh + (size_t)qld * ROWB + 16 * (2 * sidx + h)));
        });
    });

    // ---- running state per row block.  m_run (the reference, natural units) and the deferred O scale are hipcc's; the row
    // sums l (two partial sums per block), mb = m_run log2 e (0 while -inf) and thr = the raw score above which the lane asks
    // for a new reference live in the registers the bodies name (ST ...) and are rewritten only by the rare update below.
    const float inv_scale = 1.0f / p.scale;
    float m_run[QBS], pend[QBS];
    bool have_pend = false;
    // (re)starts a pass over the row block: the first tiles' DMA, O^T, the softmax state, the "blocks before the first"
    auto init_pass = [&]() {
        stage(0, 0);
        stage(1, 1);
        stage(0, 3);          // "the tile before the first": read by the first bodies' P stage (against P = 0): must be finite
#ifdef FA2_F1_PREFILL         // timing builds whose bodies issue no DMA (tools/gen_fwd_body.py, FA2_GEN_FWD_ABL=noDMA): every ring slot holds real rows
        stage(2, 2);
#endif
#pragma unroll
        for (int qb = 0; qb < QBS; ++qb) pend[qb] = 1.0f;
        have_pend = false;
        if (STATE && p.resume) {
            static_for<QBS>([&](auto QB) {
                constexpr int qb = decltype(QB)::value;
                const int qld = qrow[qb] < Nq ? qrow[qb] : Nq - 1;
                const float* Oa = p.Oacc + ((size_t)head * qhs + qld) * D;
                static_for<4 * DT>([&](auto G) {
                    constexpr int dt = decltype(G)::value / 4, g = decltype(G)::value % 4;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(Oa + 32 * dt + 8 * g + 4 * h);
                    static_for<4>([&](auto E) { f1_awrite<QBS, A_O + (qb * DT + dt) * 16 + 4 * g + decltype(E)::value>(v[decltype(E)::value]); });
                });
                m_run[qb] = p.M[(size_t)head * qhs + qld];
                f1_vsetf<QBS, ST + 2 * qb>(h == 0 ? p.L[(size_t)head * qhs + qld] : 0.0f);
                f1_vsetf<QBS, ST + 2 * qb + 1>(0.0f);
                f1_vsetf<QBS, ST_MB + qb>(m_run[qb] == -INFINITY ? 0.0f : m_run[qb] * kLog2e);
                f1_vsetf<QBS, ST_TH + qb>((m_run[qb] + kF1RescaleThr) * inv_scale);
            });
        } else {
            const u32x4 z = {0u, 0u, 0u, 0u};
            static_for<QBS * DT>([&](auto T) { f1_acc_zero<QBS, A_O + 16 * decltype(T)::value>(z); });
            static_for<QBS>([&](auto QB) {
                constexpr int qb = decltype(QB)::value;
                m_run[qb] = -INFINITY;
                f1_vsetf<QBS, ST + 2 * qb>(0.0f);
                f1_vsetf<QBS, ST + 2 * qb + 1>(0.0f);
                f1_vsetf<QBS, ST_MB + qb>(0.0f);
                f1_vsetf<QBS, ST_TH + qb>(-INFINITY);
            });
        }
        // S sets and packed P of "the blocks before the first": exp2(-huge) = 0 and P = 0, so the first two bodies add exactly zero
        static_for<16 * QBS>([&](auto R) {
            f1_vsetf<QBS, SET0 + decltype(R)::value>(-1.0e30f);
            f1_vsetf<QBS, SET1 + decltype(R)::value>(-1.0e30f);
            f1_vset<QBS, PF0 + decltype(R)::value>(0u);
        });
    };

    // ---- loop-invariant LDS addresses into the registers the bodies name
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;
    {
        const int trq = (lane & 15) >> 2, trp = lane & 3, trcb = (lane >> 4) & 1;
        static_for<KS>([&](auto S) {
            constexpr int sidx = decltype(S)::value;
            f1_vset<QBS, ROFF + sidx>(lbase + lds_off<D>(qi, 2 * sidx + h));
        });
        static_for<2 * DT>([&](auto I) {
            constexpr int dt = decltype(I)::value / 2, jj = decltype(I)::value % 2;
            f1_vset<QBS, TOFFV + decltype(I)::value>(lbase + KRING + lds_off<D>(8 * jj + 4 * h + trq, 4 * dt + 2 * trcb + (trp >> 1)) + 8 * (trp & 1));
        });
    }
    const float c2 = p.scale * kLog2e;
    F1Dma dma;
    dma.krs = k_rsrc; dma.vrs = v_rsrc;
    dma.mw = lbase + (uint32_t)wave * 1024u;
    dma.dvo = (uint32_t)doff;
    dma.kso = 0;

#ifndef FA2_F1_NO_SETPRIO
    // two waves per SIMD: the later-dispatched half of the workgroup loses every issue arbitration to the older half
    // (priority, then age); one static priority bump for that half evens them out (MI355X_MICROARCH.md, 'Two waves per SIMD')
    if constexpr (QBS == 1)
        if (wave >= kF1Waves / 2) __builtin_amdgcn_s_setprio(1);
#endif

    // ---- the rare path between two bodies: first the O^T rescale left over from the previous update, then a new reference
    auto update = [&](int need) {
        if (have_pend) {
            asm volatile("; fa2-cold: deferred O rescale");
            mfma_acc_settle();
            static_for<QBS>([&](auto QB) {
                constexpr int qb = decltype(QB)::value;
                static_for<4 * DT>([&](auto R4) { f1_scale4<QBS, A_O + qb * DT * 16 + 4 * decltype(R4)::value>(pend[qb]); });
                pend[qb] = 1.0f;
            });
            have_pend = false;
        }
        if (need) {
            asm volatile("; fa2-cold: new softmax reference");
            bool any_scale = false;
            static_for<QBS>([&](auto QB) {
                constexpr int qb = decltype(QB)::value;
                const float mx = half_max(f1_vget<ST_RM + qb>()) * p.scale;
                const bool grow = mx > m_run[qb] + kF1RescaleThr;        // also true from m_run = -inf
                const bool any_grow = __any(grow);
                const float m_new = any_grow ? fmaxf(m_run[qb], mx) : m_run[qb];
                // O only needs scaling if some row already accumulated something at an older reference
                const bool sc = any_grow && __any(m_run[qb] != -INFINITY && m_new != m_run[qb]);
                const float alpha = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f((m_run[qb] - m_new) * kLog2e);
                m_run[qb] = m_new;
                f1_vsetf<QBS, ST_MB + qb>(m_new == -INFINITY ? 0.0f : m_new * kLog2e);      // a row with no visible key yet keeps p = 0
                f1_vsetf<QBS, ST_TH + qb>((m_new + kF1RescaleThr) * inv_scale);
                f1_vsetf<QBS, ST + 2 * qb>(f1_vget<ST + 2 * qb>() * alpha);
                f1_vsetf<QBS, ST + 2 * qb + 1>(f1_vget<ST + 2 * qb + 1>() * alpha);
                pend[qb] = sc ? alpha : 1.0f;
                any_scale = any_scale || sc;
            });
            have_pend = any_scale;
        }
    };

    // Between two rounds without maxima: rows whose sums have outgrown 2^30 -- their scores have risen ~20 units above the
    // reference the first tile left them -- move their reference up by 64 ln 2, through the same deferred rescale as above.
    // A slow rise of any size therefore costs nothing; only a jump of more than ~48 units inside one round of the ring ends
    // in the restart.
    // `out_of_range` (per lane, sticky for the pass): a lift scales the sums by 2^-64, which would hide from the check at the
    // end of the pass that a sum had been at or above 2^80 (or not finite) in between -- P of a key 85 - 88 units above the
    // lagging reference is ~2^122 - 2^127, finite, and P V leaves fp32 for |V| > 2 while l does not.  So the lift looks
    // first, and what it sees goes into the restart decision.
    bool out_of_range = false;
    auto lift = [&]() {
        bool over[QBS];
        bool any = false;
#pragma unroll
        for (int qb = 0; qb < QBS; ++qb) over[qb] = false;
        static_for<QBS>([&](auto QB) {
            constexpr int qb = decltype(QB)::value;
            const float lt = f1_vget<ST + 2 * qb>() + f1_vget<ST + 2 * qb + 1>();      // this lane's half of the row's sum
            out_of_range = out_of_range || !(lt < kF1SumLimit);                          // also true for NaN / inf
            over[qb] = half_max(lt) > kF1LiftAt;      // both lanes of a row agree
            any = any || over[qb];
        });
        if (!__any(any)) return;
        asm volatile("; fa2-cold: row sums past 2^30: references up");
        update(0);                                        // an O^T rescale still pending goes first
        static_for<QBS>([&](auto QB) {
            constexpr int qb = decltype(QB)::value;
            const float m_new = over[qb] ? m_run[qb] + kF1LiftBy : m_run[qb];
            const float alpha = over[qb] ? __builtin_amdgcn_exp2f((m_run[qb] - m_new) * kLog2e) : 1.0f;
            m_run[qb] = m_new;
            f1_vsetf<QBS, ST_MB + qb>(m_new == -INFINITY ? 0.0f : m_new * kLog2e);
            f1_vsetf<QBS, ST_TH + qb>((m_new + kF1RescaleThr) * inv_scale);
            f1_vsetf<QBS, ST + 2 * qb>(f1_vget<ST + 2 * qb>() * alpha);
            f1_vsetf<QBS, ST + 2 * qb + 1>(f1_vget<ST + 2 * qb + 1>() * alpha);
            pend[qb] = alpha;
        });
        have_pend = true;
    };

    // FLAVOUR 0 / 2: every body of the tile is known to exist and to be unmasked for this wave (no per-body decisions: the
    // only code between two bodies is the test of the flag the body returns), with (0) or without (2) the lane maxima;
    // FLAVOUR 1 = GENERAL: tail, diagonal and drain tiles.
    auto run_tile = [&](auto B_, auto FLAVOUR_, int t) {
        constexpr int B = decltype(B_)::value;
        constexpr bool GENERAL = decltype(FLAVOUR_)::value == 1;
        constexpr int FAST = decltype(FLAVOUR_)::value == 2 ? F1_NOMAX : F1_PLAIN;
        dma.kso = (uint32_t)(((t + 2) * KV + wave * RPI) * ROWB);
        static_for<NH>([&](auto KB_) {
            constexpr int kb = decltype(KB_)::value;
            const int j = t * NH + kb;
            int need = 0;
            if constexpr (GENERAL) {
                if (j >= JB) return;                                  // wave- and workgroup-uniform
                const int key0 = j * 32;
                bool masked = key0 + 32 > Nk;
                if (CAUSAL) masked = masked || key0 + 31 > q0 + p.causal_shift;
                if (masked) {
                    int hi[2] = {0, 0};
#pragma unroll
                    for (int qb = 0; qb < QBS; ++qb) hi[qb] = (CAUSAL ? min(Nk, qrow[qb] + p.causal_shift + 1) : Nk) - key0 - 4 * h;
                    f1_body<D, QBS, B, kb, F1_MASKED>(c2, need, hi, dma);
                } else {
                    const int hi[2] = {0, 0};
                    f1_body<D, QBS, B, kb, F1_PLAIN>(c2, need, hi, dma);
                }
            } else {
                const int hi[2] = {0, 0};
                f1_body<D, QBS, B, kb, FAST>(c2, need, hi, dma);
            }
            // (both are SGPR values already; the readfirstlane tells hipcc that the branch is uniform)
            if (__builtin_amdgcn_readfirstlane(need | (int)have_pend)) update(need);
        });
    };
    const int ntl = (JB + NH - 1) / NH;                    // tiles with a body to run (the last ones partly)
    // tiles whose every key is visible to every row of this WAVE: plain bodies, nothing to decide
    int nfull = Nk / KV;
    if (CAUSAL) nfull = min(nfull, max(0, (q0 + p.causal_shift + 1) / KV));
    nfull = min(nfull, J / NH) & ~3;                       // whole rounds of the ring of four
    int nfull_wg = Nk / KV;
    if (CAUSAL) nfull_wg = min(nfull_wg, max(0, (rb * kF1Rows + (kF1Waves - 1) * WROWS + p.causal_shift + 1) / KV));
    nfull_wg = min(nfull_wg, J / NH) & ~3;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // Pass 0: lane maxima in the first tile only (see the header).  Pass 1 (`safe`), entered only when a row
    // sum of the workgroup left the range pass 0 vouches for: maxima in every body.
    int* const redo_flag = reinterpret_cast<int*>(smem + 2 * KRING);
    bool safe = false;
    for (;;) {
        init_pass();
        out_of_range = false;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // tiles 0 and 1 have landed
        f1_prologue<D, QBS>();
        int t = 0;
        for (; t < nfull; t += 4) {
            if (safe) {
                run_tile(I0{}, I0{}, t); run_tile(I1{}, I0{}, t + 1); run_tile(I2{}, I0{}, t + 2); run_tile(I3{}, I0{}, t + 3);
            } else if (t == 0) {             // the first tile establishes the reference
                run_tile(I0{}, I0{}, t); run_tile(I1{}, I2{}, t + 1); run_tile(I2{}, I2{}, t + 2); run_tile(I3{}, I2{}, t + 3);
            } else {
                lift();
                run_tile(I0{}, I2{}, t); run_tile(I1{}, I2{}, t + 1); run_tile(I2{}, I2{}, t + 2); run_tile(I3{}, I2{}, t + 3);
            }
        }
        for (; t < ntl; t += 4) {
            run_tile(I0{}, I1{}, t);
            if (t + 1 >= ntl) break;
            run_tile(I1{}, I1{}, t + 1);
            if (t + 2 >= ntl) break;
            run_tile(I2{}, I1{}, t + 2);
            if (t + 3 >= ntl) break;
            run_tile(I3{}, I1{}, t + 3);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the last bodies' look-ahead DMA and reads
        // Did any row of the WORKGROUP leave the range the X rounds vouch for?  (Every decision here is workgroup-uniform:
        // the restart goes through the tile barriers again.  nfull_wg: the workgroup's last wave has the most full rounds.)
        if (safe || nfull_wg == 0) break;
        bool bad = out_of_range;         // a sum seen out of range by a lift, although lifted back since
        static_for<QBS>([&](auto QB) {
            constexpr int qb = decltype(QB)::value;
            bad = bad || !(half_sum(f1_vget<ST + 2 * qb>() + f1_vget<ST + 2 * qb + 1>()) < kF1SumLimit);      // also true for NaN
        });
        if (tid == 0) *redo_flag = 0;
        __syncthreads();
        if (__any(bad) && lane == 0) *redo_flag = 1;
        __syncthreads();
        if (*redo_flag == 0) break;
        asm volatile("; fa2-cold: a row sum left the range of the rounds without maxima: the row block again, with maxima");
        safe = true;
    }
    // ---- epilogue
    mfma_acc_settle();
    const bool fin = !STATE || p.finalize;
    static_for<QBS>([&](auto QB) {
        constexpr int qb = decltype(QB)::value;
        const float l_tot = half_sum(f1_vget<ST + 2 * qb>() + f1_vget<ST + 2 * qb + 1>());
        const size_t qoff = (size_t)head * qhs + qrow[qb];
        const float pa = pend[qb];                 // an O rescale still pending from the last update (1 otherwise)
        const float inv = (fin ? (l_tot > 0.0f ? 1.0f / l_tot : 0.0f) : 1.0f) * pa;
        // a lane holds 4 consecutive columns of its row per register quad, its partner lane (+32) the next 4: for the bf16
        // output one v_permlane32_swap per packed dword pairs them up, so that every lane stores 16 contiguous bytes
        static_for<2 * DT>([&](auto G) {
            constexpr int dt = decltype(G)::value / 2, gp = decltype(G)::value % 2;
            constexpr int R = A_O + (qb * DT + dt) * 16 + 8 * gp;
            f32x4 v, w;
            v[0] = f1_aread<R>() * inv; v[1] = f1_aread<R + 1>() * inv; v[2] = f1_aread<R + 2>() * inv; v[3] = f1_aread<R + 3>() * inv;
            w[0] = f1_aread<R + 4>() * inv; w[1] = f1_aread<R + 5>() * inv; w[2] = f1_aread<R + 6>() * inv; w[3] = f1_aread<R + 7>() * inv;
            if (fin) {
                bf16x4 x, y;
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] = (__bf16)v[e]; y[e] = (__bf16)w[e]; }
                const u32x2 xu = __builtin_bit_cast(u32x2, x), yu = __builtin_bit_cast(u32x2, y);
                const auto s0 = __builtin_amdgcn_permlane32_swap(xu[0], yu[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(xu[1], yu[1], false, false);
                const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                if (qrow[qb] < Nq) *reinterpret_cast<u32x4*>((char*)p.O + qoff * ROWB + 2 * (32 * dt + 16 * gp + 8 * h)) = o;
            } else if (qrow[qb] < Nq) {
                *reinterpret_cast<f32x4*>(p.Oacc + qoff * D + 32 * dt + 16 * gp + 4 * h) = v;
                *reinterpret_cast<f32x4*>(p.Oacc + qoff * D + 32 * dt + 16 * gp + 8 + 4 * h) = w;
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
    }      // the pair's second row block
}

// the two shapes as kernels: the register budgets differ (one wave per SIMD: hipcc keeps to v0..v63 of 512 registers; two waves
// per SIMD: to v0..v39 of 128 + 128)
template <int D, bool CAUSAL, bool STATE>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(64))) fa2_fwd1_bf16_kernel(FwdArgs p)
{
    fa2_fwd1_impl<D, 2, CAUSAL, STATE>(p);
}
template <int D, bool CAUSAL, bool STATE>
__global__ void __launch_bounds__(512, 1) __attribute__((amdgpu_num_vgpr(40))) fa2_fwd1x2_bf16_kernel(FwdArgs p)
{
    fa2_fwd1_impl<D, 1, CAUSAL, STATE>(p);
}

#ifndef FA2_FWD64_QBS
#define FA2_FWD64_QBS 1          // d = 64: 1 = two waves per SIMD (VALU-bound shape), 2 = one wave per SIMD
#endif

template <int D, bool CAUSAL, bool STATE>
static hipError_t launch_one1(const FwdArgs& a, hipStream_t stream)
{
    constexpr int lds = 2 * kF1Bufs * 16384 + 16;          // the two rings + the workgroup's restart flag
    const int nrb1 = (a.Nq + kF1Rows - 1) / kF1Rows;
    const int nrb = (CAUSAL && !STATE) ? (nrb1 + 1) / 2 : nrb1;      // plain causal launches: two row blocks per workgroup (fa2_fwd1_impl)
    static bool attr_set[64] = {};
    if constexpr (D == 64 && FA2_FWD64_QBS == 1) {
        auto kern = fa2_fwd1x2_bf16_kernel<D, CAUSAL, STATE>;
        hipError_t e = ensure_dynamic_lds(kern, lds, attr_set);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)(nrb * a.BH)), dim3(512), lds, stream, a);
    } else {
        auto kern = fa2_fwd1_bf16_kernel<D, CAUSAL, STATE>;
        hipError_t e = ensure_dynamic_lds(kern, lds, attr_set);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)(nrb * a.BH)), dim3(256), lds, stream, a);
    }
    return hipGetLastError();
}
hipError_t launch_fwd1_bf16(const FwdArgs& a, hipStream_t stream)
{
    const bool state = a.resume || !a.finalize;
    if (a.d == 128) {
        if (state) return a.causal ? launch_one1<128, true, true>(a, stream) : launch_one1<128, false, true>(a, stream);
        return a.causal ? launch_one1<128, true, false>(a, stream) : launch_one1<128, false, false>(a, stream);
    }
    if (a.d == 64) {
        if (state) return a.causal ? launch_one1<64, true, true>(a, stream) : launch_one1<64, false, true>(a, stream);
        return a.causal ? launch_one1<64, true, false>(a, stream) : launch_one1<64, false, false>(a, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace fa2
