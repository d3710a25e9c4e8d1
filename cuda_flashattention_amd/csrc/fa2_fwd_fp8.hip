// fa2_fwd_fp8.hip -- FlashAttention-2 forward with fp8 (OCP e4m3) Q/K/V for gfx950 (MI355X):
// BASELINE configs[4] ("FA2 fwd causal long-context, fp8 Q/K/V, N=32768 d=128").  Same algorithm as the bf16 forward
// (fa2_fwd1_bf16.hip: S^T = K Q^T so that a row's max and sum are in-lane reductions, P stays in registers, lazy softmax
// reference, exp2 domain, natural-log L) and, since round 3, the same construction: the main loop is ONE generated asm body
// per 64 keys (tools/gen_fwd_fp8_body.py -> fa2_fwd_fp8_body.inc), a pipeline two bodies deep --
//   A  S^T(j) = K(j) Q^T  (4 MFMAs)      P  O^T += V^T(j-2) P^T(j-2)  (4 MFMAs)      VALU: softmax of the keys j-1, maxima of the keys j
// -- with every LDS read issued ahead of its use behind a counted lgkmcnt, the next tile's LDS-DMA issued from inside the
// bodies and the softmax state in registers the bodies name (hipcc owns v0..v31 only).  Workgroup = 8 waves x 32 query rows,
// two waves per SIMD: the kernel is bound by VALU issue (4 VALU instructions per S element against a quarter of the MFMA
// cycles of bf16 d = 128), and two waves issue VALU instructions at ~4 clocks each against ~7 for a wave alone.
//
// What follows from the matrix instruction:
//   * v_mfma_f32_32x32x64_f8f6f4 contracts 64 values per instruction (32 bytes per lane and operand): S^T = K Q^T over
//     d = 128 is two MFMAs per 32x32 tile, O^T += V^T P^T over 64 keys is ONE per 32-column tile of O.
//   * Both operands of an MFMA pair lane-half h, slot j with lane-half h, slot j; which k index the hardware calls that
//     does not matter as long as both operands are gathered the same way.  For S^T both K and Q fragments take bytes
//     64 s + 32 h .. + 31 of their row.  For P V the B operand is the exponentiated S^T accumulator, whose register r in
//     lane-half h is accumulator row (r & 3) + 8 (r >> 2) + 4 h; the K rows are therefore fed in a permuted order (pi
//     below) that makes those 16 registers the CONSECUTIVE keys 16 h .. 16 h + 15 of the 32-key block.  Packed four to a
//     register they are k-slots 0..15 (first key block) and 16..31 (second) of the B operand, and the matching A operand
//     is two plain 16-byte row reads of a V^T tile -- which is why V is transposed once per call into a workspace
//     ([d][Npad] per head, fa2_fp8_transpose_kernel: 34 us of 1.95 ms at the BASELINE shape).  Reading V through
//     ds_read_b64_tr_b8 instead would take four LDS instructions per fragment where the V^T image takes two: eight more
//     issue slots per 64 keys and wave in a kernel whose bound is issue slots -- more than the 1.7 % the pass costs.
//   * P is rounded to e4m3 (v_cvt_pk_fp8_f32) for the second product; with the lazy reference P never exceeds
//     e^6 = 403 < 448, the largest e4m3 value.  The row sum is taken from the unrounded fp32 p.
//   * O is written in bf16, L in fp32.  d = 128 only.
//
// Lane maxima only where they can matter.  The online softmax needs a row's maximum over new keys only to notice that a score
// passes the row's threshold (reference + 6).  By Cauchy-Schwarz scale q.k <= scale |q| |k|: the pre-pass that transposes V also
// writes, per 64 keys, the largest |k| (kn), a wave knows the largest scale |q| of its 32 rows (qn) and the smallest reference
// among them (m_min, refreshed by the rare update); for a round of the LDS ring (512 keys) with qn max(kn) <= m_min + 6 no
// score can pass any threshold, and the wave runs the X variant of the bodies: no v_max3 (16 of a body's 130 VALU
// instructions), no compare.  Same bits as with the maxima -- they would have triggered nothing.  Keys with an outlier norm,
// the first round (no reference yet), the sequence tail and the causal diagonal run the bodies with maxima.
#include <type_traits>

#include "fa2_common.h"
#include "fa2_launch.h"

namespace fa2 {

#include "fa2_fwd_fp8_body.inc"

constexpr int kF8Waves = 8;
constexpr int kF8Rows = 32 * kF8Waves;
constexpr int kF8KV = FA2_F8_KV;            // keys per LDS tile (two bodies of 64)
constexpr int kF8Bufs = FA2_F8_NBUF;
constexpr int kF8D = 128;
constexpr float kF8RescaleThr = 6.0f;

typedef __attribute__((address_space(3))) void* f8_lds_ptr_t;

#define FA2_F8_CLOBBERS \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
    "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", \
    "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", \
    "a124", "a125", "a126", "a127"
// every statement that names a body-owned register: the whole accumulator file and v127 (so that the kernel is allocated all
// 128 VGPRs: hipcc itself stays below FA2_F8_V0, amdgpu_num_vgpr)
#define FA2_F8_REGS "v127", FA2_F8_CLOBBERS
#define FA2_F8_MISC "memory", "vcc", "scc", "s10", "s11", "s12", "m0"

template <int R>
__device__ __forceinline__ void f8_vset(uint32_t x)
{
    asm volatile("v_mov_b32 v%c1, %0" : : "v"(x), "i"(R) : FA2_F8_REGS);
}
template <int R>
__device__ __forceinline__ void f8_vsetf(float x)
{
    asm volatile("v_mov_b32 v%c1, %0" : : "v"(x), "i"(R) : FA2_F8_REGS);
}
template <int R>
__device__ __forceinline__ float f8_vget()
{
    float x;
    asm volatile("v_mov_b32 %0, v%c1" : "=v"(x) : "i"(R));
    return x;
}
template <int R>
__device__ __forceinline__ void f8_acc_write(uint32_t x)
{
    asm volatile("v_accvgpr_write_b32 a[%c1], %0" : : "v"(x), "i"(R) : FA2_F8_REGS);
}
template <int R>
__device__ __forceinline__ float f8_acc_read()
{
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(R));
    return x;
}
template <int R>
__device__ __forceinline__ void f8_acc_scale4(float alpha)
{
    float t0, t1, t2, t3;
    asm volatile("v_accvgpr_read_b32 %0, a[%c5]\n\tv_accvgpr_read_b32 %1, a[%c6]\n\t"
                 "v_accvgpr_read_b32 %2, a[%c7]\n\tv_accvgpr_read_b32 %3, a[%c8]\n\t"
                 "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4\n\t"
                 "v_accvgpr_write_b32 a[%c5], %0\n\tv_accvgpr_write_b32 a[%c6], %1\n\t"
                 "v_accvgpr_write_b32 a[%c7], %2\n\tv_accvgpr_write_b32 a[%c8], %3"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                 : "v"(alpha), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3) : FA2_F8_REGS);
}
template <int R>
__device__ __forceinline__ void f8_acc_zero(u32x4 z)
{
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, %0, 0" : : "v"(z), "i"(R), "i"(R + 15) : FA2_F8_REGS);
}

// a wave-uniform float into an SGPR (the builtin is typed int: the value goes through it as bits)
__device__ __forceinline__ float f8_uniform(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

// LDS images.  K tile: [128 keys][128 B], 16-byte chunk index XORed with fK(row); V^T tile: two halves (64 keys each) of
// [128 d][64 B], chunk index XORed with fV(row).  Both make a 16-lane group of a ds_read_b128 (16 different rows, same
// logical chunk) hit 16 different 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int f8_fk(int row) { return ((row >> 1) & 3) | (((row >> 4) & 1) << 2); }
__device__ __forceinline__ int f8_fv(int row) { return (row >> 2) & 3; }
// K row fed as accumulator row m of a 32-key block (see the header): m = 8 j + 4 b + i -> 16 b + 4 j + i.
__device__ __forceinline__ int f8_pi(int m) { return (m & 3) + 4 * ((m >> 3) & 3) + 16 * ((m >> 2) & 1); }

// ---- pre-pass, one block per 64 keys of a head: V [N][128] -> V^T [128][Npad] (keys >= N written as zeros), and
// kn[head][block] = the largest |k| among the block's keys, rounded up (see the header)
__global__ void __launch_bounds__(256) fa2_fp8_transpose_kernel(const unsigned char* V, unsigned char* Vt, const unsigned char* K, float* kn,
                                                                int N, int Npad)
{
    constexpr int TS = kF8D + 16;                    // tile row pitch: 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) unsigned char tile[64 * TS];
    __shared__ float wmax[4];
    const int head = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const int key = t * 64 + (tid >> 2), part = tid & 3;        // four lanes per key: 32 bytes each
    {
        float ss = 0.0f;
        if (key < N) {
            const u32x4* kp = reinterpret_cast<const u32x4*>(K + ((size_t)head * N + key) * kF8D + 32 * part);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const u32x4 w = kp[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[e], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[e], true);
                    ss += lo[0] * lo[0] + lo[1] * lo[1] + hi[0] * hi[0] + hi[1] * hi[1];
                }
            }
        }
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) ss = fmaxf(ss, __shfl_xor(ss, o));
        if ((tid & 63) == 0) wmax[tid >> 6] = ss;
    }
    // V rows -> LDS, 16 bytes at a time
    {
        u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = v0;
        if (key < N) {
            const u32x4* vp = reinterpret_cast<const u32x4*>(V + ((size_t)head * N + key) * kF8D + 32 * part);
            v0 = vp[0]; v1 = vp[1];
        }
        u32x4* dst = reinterpret_cast<u32x4*>(tile + (tid >> 2) * TS + 32 * part);
        dst[0] = v0; dst[1] = v1;
    }
    __syncthreads();
    // out: 16 keys of one d row per task and 16-byte store; four consecutive lanes write one 64-byte run of a V^T row
    unsigned char* Vth = Vt + (size_t)head * kF8D * Npad;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i, chunk = q & 3, dcol = q >> 2;
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t v = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) v |= (uint32_t)tile[(16 * chunk + 4 * w + e) * TS + dcol] << (8 * e);
            o[w] = v;
        }
        *reinterpret_cast<u32x4*>(Vth + (size_t)dcol * Npad + t * 64 + 16 * chunk) = o;
    }
    if (tid == 0)
        kn[(size_t)head * (Npad / 64) + t] = __builtin_sqrtf(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))) * 1.0002f;
}

struct F8Dma {
    __amdgpu_buffer_rsrc_t krs, vrs;
    uint32_t mw, dvk, dvv, kso, vso;
};

// One body: half KB of the tile in ring buffer BUF.  KB == 1 starts with the tile barrier and issues the DMA of tile t + 2.
enum { F8_PLAIN = 0, F8_MASKED = 1, F8_NOMAX = 2 };
template <int BUF, int KB, int MODE>
__device__ __forceinline__ void f8_body(float c2, int& need, int hi, const F8Dma& dma)
{
#define FA2_F8_CASE(B, K, MV, M)                                                                                              \
    if constexpr (BUF == B && KB == K && MODE == MV)                                                                           \
        asm volatile(FA2_F8_BODY_B##B##_K##K##_##M                                                                             \
                     : [need] "=&s"(need)                                                                                      \
                     : [c2] "s"(c2), [hi] "v"(hi), [ninf] "v"(-INFINITY), [mw] "s"(dma.mw), [dvk] "v"(dma.dvk), [dvv] "v"(dma.dvv), \
                       [krs] "s"(dma.krs), [vrs] "s"(dma.vrs), [kso] "s"(dma.kso), [vso] "s"(dma.vso)                           \
                     : FA2_F8_MISC, FA2_F8_REGS);
#define FA2_F8_CASES(K, MV, M) FA2_F8_CASE(0, K, MV, M) FA2_F8_CASE(1, K, MV, M) FA2_F8_CASE(2, K, MV, M) FA2_F8_CASE(3, K, MV, M)
    FA2_F8_CASES(0, F8_PLAIN, M0) FA2_F8_CASES(1, F8_PLAIN, M0) FA2_F8_CASES(0, F8_MASKED, M1) FA2_F8_CASES(1, F8_MASKED, M1)
    FA2_F8_CASES(0, F8_NOMAX, X) FA2_F8_CASES(1, F8_NOMAX, X)
#undef FA2_F8_CASES
#undef FA2_F8_CASE
}

template <bool CAUSAL>
__global__ void __launch_bounds__(64 * kF8Waves, 1) __attribute__((amdgpu_num_vgpr(FA2_F8_V0))) fa2_fwd_fp8_kernel(FwdFp8Args p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = kF8D;                 // bytes per K row
    constexpr int KV = kF8KV, NH = KV / 64;    // keys per tile, bodies per tile
    constexpr int TILEB = KV * ROWB;           // 16 KiB: K tile; V^T tile = two halves of [128 d][64 B]
    constexpr int KRING = kF8Bufs * TILEB;     // LDS: [4 K tiles][4 V^T tiles]
    constexpr int DT = kF8D / 32;
    constexpr int SET0 = FA2_F8_SET0, SET1 = FA2_F8_SET1, PF0 = FA2_F8_PF0, KA = FA2_F8_KA, VA = FA2_F8_VA, ST = FA2_F8_STATE;
    constexpr int ST_RM = ST + 2, ST_MB = ST + 3, ST_TH = ST + 4, A_QF = FA2_F8_A_QF;
    static_assert(TILEB == 16384 && NH == 2 && kF8Bufs == 4, "LDS ring as the generator lays it out");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31;
    const int h = lane >> 5;

    const int N = p.N, Npad = p.Npad;
    const int nrb = (N + kF8Rows - 1) / kF8Rows;
    int head, rb;
    map_block(blockIdx.x, p.BH, nrb, head, rb);
    if (CAUSAL) {
        rb = nrb - 1 - rb;                    // heaviest row blocks first
        // Round 4: ... over TWO heads of an XCD at a time (row-block-major inside the pair), not head after head: with one head
        // after the other an XCD's last head still began with its heaviest blocks (a quarter of a CU's whole share each at
        // BASELINE configs[4]) when the CUs were about to run dry; longest-first over the pair ends with the lightest blocks of
        // both.  (This kernel has 32 compiler registers and none to spare for the bf16 forward's two-blocks-per-workgroup loop.)
        if ((p.BH & 7) == 0) {
            const int x = blockIdx.x & 7, j = blockIdx.x >> 3, hx = p.BH >> 3;
            const int g = j / (2 * nrb);                       // pair of heads of this XCD
            const bool full = 2 * g + 1 < hx;                  // (an odd head count per XCD: the last one goes alone)
            const int r = j - g * 2 * nrb;
            rb = nrb - 1 - (full ? r >> 1 : r);
            head = (2 * g + (full ? (r & 1) : 0)) * 8 + x;
        }
    }

    const char* Qh = (const char*)p.Q + (size_t)head * N * ROWB;
    const char* Kh = (const char*)p.K + (size_t)head * N * ROWB;
    const char* Vth = (const char*)p.Vt + (size_t)head * kF8D * Npad;

    const int q0 = rb * kF8Rows + wave * 32;      // first query row of this wave
    const int qrow = q0 + qi;
    const int qld = qrow < N ? qrow : N - 1;

    // bodies (64 keys) that hold a visible key for some row of the WORKGROUP (the tile barriers need every wave in every
    // body); two more bodies drain the pipeline (their S^T is masked completely: P = 0)
    int J = (N + 63) / 64;
    if (CAUSAL) {
        const int last_key = min(rb * kF8Rows + kF8Rows - 1, N - 1);
        J = min(J, last_key / 64 + 1);
    }
    const int JB = J + 2;

    // ---- LDS-DMA staging: the swizzle is applied to the SOURCE chunk, the LDS write is linear.  K piece pc = rows 8 pc .. + 7
    // of the tile (wave w: pieces w and w + 8); V^T piece (half hf, w) = d rows 16 w .. + 15 of keys 64 hf .. + 63.
    const int krow = lane >> 3, kslot = lane & 7;
    const int doffK = krow * ROWB + 16 * (kslot ^ f8_fk(8 * wave + krow));
    const int vrow = lane >> 2, vslot = lane & 3;
    const int doffV = vrow * Npad + 16 * (vslot ^ f8_fv(vrow));
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, N * ROWB, 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vth, 0, kF8D * Npad, 0x00020000);
    auto stage = [&](int t, int buf) {
        char* b = smem + buf * TILEB + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (f8_lds_ptr_t)(b + i * 8192), 16, doffK, (t * KV + 8 * wave + 64 * i) * ROWB, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (f8_lds_ptr_t)(b + KRING + i * 8192), 16, doffV,
                                                     16 * wave * Npad + t * KV + 64 * i, 0, 0);
        }
    };
    stage(0, 0);
    stage(1, 1);
    stage(0, 3);          // "the tile before the first": read by the first bodies' P stage (against P = 0): must be finite

    // ---- Q fragments -> AGPRs: k-step s takes bytes 64 s + 32 h .. + 31 of the row
    // (on the way: qn = the largest scale |q| among the wave's rows, rounded up -- see the header)
    float qss = 0.0f;
    static_for<2>([&](auto S) {
        constexpr int s = decltype(S)::value;
        const u32x4 lo = *reinterpret_cast<const u32x4*>(Qh + (size_t)qld * ROWB + 64 * s + 32 * h);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(Qh + (size_t)qld * ROWB + 64 * s + 32 * h + 16);
        static_for<4>([&](auto E) {
            constexpr int e = decltype(E)::value;
            f8_acc_write<A_QF + 8 * s + e>(lo[e]);
            f8_acc_write<A_QF + 8 * s + 4 + e>(hi[e]);
            const auto a = __builtin_amdgcn_cvt_pk_f32_fp8(lo[e], false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo[e], true);
            const auto c = __builtin_amdgcn_cvt_pk_f32_fp8(hi[e], false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi[e], true);
            qss += a[0] * a[0] + a[1] * a[1] + b[0] * b[0] + b[1] * b[1] + c[0] * c[0] + c[1] * c[1] + d[0] * d[0] + d[1] * d[1];
        });
    });
    qss = half_sum(qss);                              // the two lane halves hold the two halves of each 64-byte k-step
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) qss = fmaxf(qss, __shfl_xor(qss, o));
    const float qn = f8_uniform(__builtin_sqrtf(qss) * p.scale * 1.0002f);
    const float* knh = p.kn + (size_t)head * (Npad / 64);
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        static_for<DT>([&](auto T) { f8_acc_zero<16 * decltype(T)::value>(z); });
    }

    // ---- running state.  m_run (the reference, natural units) and the deferred O scale are hipcc's; the row sums (two partial
    // sums), mb = m_run log2 e (0 while -inf) and thr = the raw score above which the lane asks for a new reference live in
    // the registers the bodies name and are rewritten only by the rare update below.
    const float inv_scale = 1.0f / p.scale;
    float m_run = -INFINITY, pend = 1.0f;
    float m_min6 = -INFINITY;             // wave-uniform: the smallest threshold (reference + 6, natural units) among the wave's rows
    bool have_pend = false;
    f8_vsetf<ST>(0.0f);
    f8_vsetf<ST + 1>(0.0f);
    f8_vsetf<ST_MB>(0.0f);
    f8_vsetf<ST_TH>(-INFINITY);
    // S sets and packed P of "the keys before the first": exp2(-huge) = 0 and P = 0, so the first two bodies add exactly zero
    static_for<32>([&](auto R) {
        f8_vsetf<SET0 + decltype(R)::value>(-1.0e30f);
        f8_vsetf<SET1 + decltype(R)::value>(-1.0e30f);
    });
    static_for<16>([&](auto R) { f8_vset<PF0 + decltype(R)::value>(0u); });

    // ---- loop-invariant LDS addresses into the registers the bodies name.  K fragment (blk, s), half i: row pi(qi) + 32 blk
    // (+ 64 per body), chunk 4 s + 2 h + i; V^T fragment dt: row 32 dt + qi of a half, chunks h (keys 16 h ..) and 2 + h
    // (keys 32 + 16 h ..).
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;
    {
        const int prow = f8_pi(qi);
        static_for<4>([&](auto I) {
            constexpr int s = decltype(I)::value / 2, i = decltype(I)::value % 2;
            f8_vset<KA + decltype(I)::value>(lbase + prow * ROWB + 16 * ((4 * s + 2 * h + i) ^ f8_fk(prow)));
        });
        f8_vset<VA>(lbase + KRING + qi * 64 + 16 * (h ^ f8_fv(qi)));
        f8_vset<VA + 1>(lbase + KRING + qi * 64 + 16 * ((2 + h) ^ f8_fv(qi)));
    }
    const float c2 = p.scale * kLog2e;
    F8Dma dma;
    dma.krs = k_rsrc; dma.vrs = v_rsrc;
    dma.mw = lbase + (uint32_t)wave * 1024u;
    dma.dvk = (uint32_t)doffK;
    dma.dvv = (uint32_t)doffV;
    dma.kso = 0; dma.vso = 0;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                         // tiles 0 and 1 have landed
    // two waves per SIMD: the later-dispatched half of the workgroup loses every issue arbitration to the older half
    // (priority, then age); one static priority bump for that half evens them out (MI355X_MICROARCH.md, 'Two waves per SIMD')
#ifndef FA2_F8_NO_SETPRIO
    if (wave >= kF8Waves / 2) __builtin_amdgcn_s_setprio(1);
#endif
    asm volatile(FA2_F8_PRO : : : FA2_F8_MISC, FA2_F8_REGS);

    // ---- the rare path between two bodies: first the O^T rescale left over from the previous update, then a new reference
    // (fa2_fwd1_bf16.hip: at that point O^T holds the products through the keys j - 2, the row sums through j - 1, and P(j-1)
    // is packed and waiting: the sums are rescaled at once, O^T one body later)
    auto update = [&](int need) {
        if (have_pend) {
            asm volatile("; fa2-cold: deferred O rescale");
            mfma_acc_settle();
            static_for<4 * DT>([&](auto R4) { f8_acc_scale4<4 * decltype(R4)::value>(pend); });
            pend = 1.0f;
            have_pend = false;
        }
        if (need) {
            asm volatile("; fa2-cold: new softmax reference");
            const float mx = half_max(f8_vget<ST_RM>()) * p.scale;
            const bool grow = mx > m_run + kF8RescaleThr;        // also true from m_run = -inf
            const bool any_grow = __any(grow);
            const float m_new = any_grow ? fmaxf(m_run, mx) : m_run;
            // O only needs scaling if some row already accumulated something at an older reference
            const bool sc = any_grow && __any(m_run != -INFINITY && m_new != m_run);
            const float alpha = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
            m_run = m_new;
            f8_vsetf<ST_MB>(m_new == -INFINITY ? 0.0f : m_new * kLog2e);      // a row with no visible key yet keeps p = 0
            f8_vsetf<ST_TH>((m_new + kF8RescaleThr) * inv_scale);
            f8_vsetf<ST>(f8_vget<ST>() * alpha);
            f8_vsetf<ST + 1>(f8_vget<ST + 1>() * alpha);
            pend = sc ? alpha : 1.0f;
            have_pend = sc;
            float mm = m_new;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) mm = fminf(mm, __shfl_xor(mm, o));
            m_min6 = f8_uniform(mm) + kF8RescaleThr;
        }
    };

    // FLAVOUR 0 / 2: both bodies of the tile are known to exist and to be unmasked for this wave (the only code between two
    // bodies is the test of the flag the body returns), with (0) or without (2) the lane maxima; 1: GENERAL -- tail, diagonal
    // and drain tiles.
    auto run_tile = [&](auto B_, auto FLAVOUR_, int t) {
        constexpr int B = decltype(B_)::value;
        constexpr bool GENERAL = decltype(FLAVOUR_)::value == 1;
        constexpr int FAST = decltype(FLAVOUR_)::value == 2 ? F8_NOMAX : F8_PLAIN;
        dma.kso = (uint32_t)(((t + 2) * KV + 8 * wave) * ROWB);
        dma.vso = (uint32_t)(16 * wave * Npad + (t + 2) * KV);
        static_for<NH>([&](auto KB_) {
            constexpr int kb = decltype(KB_)::value;
            const int j = t * NH + kb;
            int need = 0;
            if constexpr (GENERAL) {
                if (j >= JB) return;                                  // wave- and workgroup-uniform
                const int key0 = j * 64;
                bool masked = key0 + 64 > N;
                if (CAUSAL) masked = masked || key0 + 63 > q0;
                if (masked) {
                    // register r of half h of block blk is key key0 + 32 blk + 16 h + r: alive iff 32 blk + r < hi
                    const int hi = (CAUSAL ? min(N, qrow + 1) : N) - key0 - 16 * h;
                    f8_body<B, kb, F8_MASKED>(c2, need, hi, dma);
                } else {
                    f8_body<B, kb, F8_PLAIN>(c2, need, 0, dma);
                }
            } else {
                f8_body<B, kb, FAST>(c2, need, 0, dma);
            }
            // (both are SGPR values already; the readfirstlane tells hipcc that the branch is uniform)
            if (__builtin_amdgcn_readfirstlane(need | (int)have_pend)) update(need);
        });
    };
    const int ntl = (JB + NH - 1) / NH;                    // tiles with a body to run (the last one maybe partly)
    // tiles whose every key is visible to every row of this WAVE: plain bodies, nothing to decide
    int nfull = N / KV;
    if (CAUSAL) nfull = min(nfull, (q0 + 1) / KV);
    nfull = min(nfull, J / NH) & ~3;                       // whole rounds of the ring of four
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    int t = 0;
    for (; t < nfull; t += 4) {
        // one round of the ring = 512 keys = 8 entries of kn: can any of their scores pass any row's threshold?
        float kmax = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) kmax = fmaxf(kmax, knh[2 * t + i]);
#ifdef FA2_F8_FORCE_MAX
        kmax = INFINITY;
#endif
        if (qn * kmax <= m_min6) {
            run_tile(I0{}, I2{}, t); run_tile(I1{}, I2{}, t + 1); run_tile(I2{}, I2{}, t + 2); run_tile(I3{}, I2{}, t + 3);
        } else {
            run_tile(I0{}, I0{}, t); run_tile(I1{}, I0{}, t + 1); run_tile(I2{}, I0{}, t + 2); run_tile(I3{}, I0{}, t + 3);
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

    // ---- epilogue (as fa2_fwd1_bf16.hip: a lane holds 4 consecutive columns of its row per register quad, its partner lane
    // (+32) the next 4; one v_permlane32_swap per packed dword pairs them up so that every lane stores 16 contiguous bytes)
    mfma_acc_settle();
    const float l_tot = half_sum(f8_vget<ST>() + f8_vget<ST + 1>());
    const size_t qoff = (size_t)head * N + qrow;
    const float inv = (l_tot > 0.0f ? 1.0f / l_tot : 0.0f) * pend * p.o_scale;      // pend: an O rescale still pending from the last update; o_scale: V's descale
    static_for<2 * DT>([&](auto G) {
        constexpr int dt = decltype(G)::value / 2, gp = decltype(G)::value % 2;
        constexpr int R = dt * 16 + 8 * gp;
        f32x4 v, w;
        v[0] = f8_acc_read<R>() * inv; v[1] = f8_acc_read<R + 1>() * inv; v[2] = f8_acc_read<R + 2>() * inv; v[3] = f8_acc_read<R + 3>() * inv;
        w[0] = f8_acc_read<R + 4>() * inv; w[1] = f8_acc_read<R + 5>() * inv; w[2] = f8_acc_read<R + 6>() * inv; w[3] = f8_acc_read<R + 7>() * inv;
        bf16x4 x, y;
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = (__bf16)v[e]; y[e] = (__bf16)w[e]; }
        const u32x2 xu = __builtin_bit_cast(u32x2, x), yu = __builtin_bit_cast(u32x2, y);
        const auto s0 = __builtin_amdgcn_permlane32_swap(xu[0], yu[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(xu[1], yu[1], false, false);
        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
        if (qrow < N) *reinterpret_cast<u32x4*>((char*)p.O + qoff * (kF8D * 2) + 2 * (32 * dt + 16 * gp + 8 * h)) = o;
    });
    if (qrow < N && h == 0) p.L[qoff] = m_run + __builtin_logf(l_tot);
}

hipError_t launch_fwd_fp8(const FwdFp8Args& a, hipStream_t stream)
{
    if (a.d != kF8D || a.Npad % 64 != 0 || a.Npad < a.N || !a.kn) return hipErrorInvalidValue;
    constexpr int lds = 2 * kF8Bufs * kF8KV * kF8D;
    hipLaunchKernelGGL(fa2_fp8_transpose_kernel, dim3((unsigned)(a.Npad / 64), (unsigned)a.BH), dim3(256), 0, stream,
                       (const unsigned char*)a.V, (unsigned char*)a.Vt, (const unsigned char*)a.K, a.kn, a.N, a.Npad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int nrb = (a.N + kF8Rows - 1) / kF8Rows;
    const dim3 grid((unsigned)(nrb * a.BH));
    static bool set_c[64] = {}, set_n[64] = {};
    if (a.causal) {
        e = ensure_dynamic_lds(fa2_fwd_fp8_kernel<true>, lds, set_c);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fa2_fwd_fp8_kernel<true>, grid, dim3(64 * kF8Waves), lds, stream, a);
    } else {
        e = ensure_dynamic_lds(fa2_fwd_fp8_kernel<false>, lds, set_n);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fa2_fwd_fp8_kernel<false>, grid, dim3(64 * kF8Waves), lds, stream, a);
    }
    return hipGetLastError();
}

}  // namespace fa2
