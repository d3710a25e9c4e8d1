// fa2_fwd_fp8.hip -- FlashAttention-2 forward with fp8 (OCP e4m3) Q/K/V for gfx950 (MI355X):
// BASELINE configs[4] ("FA2 fwd causal long-context, fp8 Q/K/V, N=32768 d=128").  Same algorithm
// and the same machine mapping as fa2_fwd_bf16.hip (8 waves x 32 query rows, two waves per SIMD,
// kernel-owned AGPRs, stages as long asm statements, lazy softmax reference, LDS-DMA ring of three
// 64-key tiles); what differs is the matrix instruction and everything that follows from it:
//
//   * v_mfma_f32_32x32x64_f8f6f4 contracts 64 values per instruction (32 bytes per lane and
//     operand): S^T = K Q^T over d = 128 is two MFMAs per 32x32 tile, O^T += V^T P^T over a 64-key
//     tile is ONE per 32-column tile of O.
//   * Both operands of an MFMA pair lane-half h, slot j with lane-half h, slot j; which k index the
//     hardware calls that does not matter as long as both operands are gathered the same way.  For
//     S^T both K and Q fragments take bytes 64 s + 32 h .. + 31 of their row.  For P V the B operand
//     is the exponentiated S^T accumulator, whose register r in lane-half h is accumulator row
//     (r & 3) + 8 (r >> 2) + 4 h; the K rows are therefore fed in a permuted order (pi below) that
//     makes those 16 registers the CONSECUTIVE keys 16 h .. 16 h + 15 of the 32-key block.  Packed
//     four to a register they are k-slots 0..15 (first key block) and 16..31 (second) of the B
//     operand, and the matching A operand is two plain 16-byte row reads of a V^T tile -- which is why
//     V is transposed once per call into a workspace ([d][N] per head, fa2_fp8_transpose_kernel)
//     instead of being read through transposed LDS loads.
//   * P is rounded to e4m3 (v_cvt_pk_fp8_f32) for the second product; with the lazy reference P never
//     exceeds e^6 = 403 < 448, the largest e4m3 value.  The row sum is taken from the unrounded fp32 p.
//   * O is written in bf16, L in fp32.  d = 128 only.
#include <type_traits>

#include "fa2_common.h"
#include "fa2_launch.h"

namespace fa2 {

constexpr int kF8Waves = 8;
constexpr int kF8Rows = 32 * kF8Waves;
constexpr int kF8KV = 64;
constexpr int kF8Bufs = 3;
constexpr int kF8D = 128;
constexpr float kF8RescaleThr = 6.0f;

typedef __attribute__((address_space(3))) void* f8_lds_ptr_t;
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8;

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

// AGPR map: O^T tile dt: a[16 dt ..+15]; Q fragment of k-step s: a[64 + 8 s ..+7]; packed P: a[80:87]; fragment slot i:
// a[96 + 8 i ..+7] (K fragments (kb, s) -> slot 2 kb + s in the A stage, V^T fragment dt -> slot dt in
// the B stage).
template <int R>
__device__ __forceinline__ void f8_acc_write(float x)
{
    asm volatile("v_accvgpr_write_b32 a[%c1], %0" : : "v"(x), "i"(R) : FA2_F8_CLOBBERS);
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
                 : "v"(alpha), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3) : FA2_F8_CLOBBERS);
}

// LDS images.  K tile: [64 keys][128 B], 16-byte chunk index XORed with fK(row); V^T tile:
// [128 d][64 B], chunk index XORed with fV(row).  Both make a 16-lane group of a ds_read_b128 (16
// different rows, same logical chunk) hit 16 different 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int f8_fk(int row) { return ((row >> 1) & 3) | (((row >> 4) & 1) << 2); }
__device__ __forceinline__ int f8_fv(int row) { return (row >> 2) & 3; }
// K row fed as accumulator row m of a 32-key block (see the header): m = 8 j + 4 b + i -> 16 b + 4 j + i.
__device__ __forceinline__ int f8_pi(int m) { return (m & 3) + 4 * ((m >> 3) & 3) + 16 * ((m >> 2) & 1); }

// ---- V [N][128] -> V^T [128][Npad] per head (keys >= N written as zeros)
__global__ void __launch_bounds__(256) fa2_fp8_transpose_kernel(const unsigned char* V, unsigned char* Vt, int N, int Npad)
{
    __shared__ unsigned char tile[64][kF8D + 4];
    const int head = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const unsigned char* Vh = V + (size_t)head * N * kF8D;
    unsigned char* Vth = Vt + (size_t)head * kF8D * Npad;
    for (int c = tid; c < 64 * (kF8D / 4); c += 256) {
        const int row = c / (kF8D / 4), w = c % (kF8D / 4);
        const int key = t * 64 + row;
        uint32_t v = 0;
        if (key < N) v = *reinterpret_cast<const uint32_t*>(Vh + (size_t)key * kF8D + 4 * w);
        *reinterpret_cast<uint32_t*>(&tile[row][4 * w]) = v;
    }
    __syncthreads();
    for (int c = tid; c < kF8D * 16; c += 256) {     // 16 words of 4 keys per d row
        const int dcol = c / 16, w = c % 16;
        uint32_t v = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) v |= (uint32_t)tile[4 * w + e][dcol] << (8 * e);
        *reinterpret_cast<uint32_t*>(Vth + (size_t)dcol * Npad + t * 64 + 4 * w) = v;
    }
}

// One softmax quad: t0..t3 = exp2(s * c2 - mb); l += t0 + t1 + t2 + t3; w = pack_e4m3(t0, t1, t2, t3).
#define FA2_F8_QUAD(sa, sb, sc, sd, w)                          \
    "v_fma_f32 %[t0], %[" sa "], %[c2], -%[mb]\n\t"            \
    "v_fma_f32 %[t1], %[" sb "], %[c2], -%[mb]\n\t"            \
    "v_exp_f32 %[t0], %[t0]\n\t"                                \
    "v_exp_f32 %[t1], %[t1]\n\t"                                \
    "v_add_f32 %[l], %[l], %[t0]\n\t"                           \
    "v_cvt_pk_fp8_f32 %[" w "], %[t0], %[t1]\n\t"               \
    "v_add_f32 %[l], %[l], %[t1]\n\t"                           \
    "v_fma_f32 %[t0], %[" sc "], %[c2], -%[mb]\n\t"            \
    "v_fma_f32 %[t1], %[" sd "], %[c2], -%[mb]\n\t"            \
    "v_exp_f32 %[t0], %[t0]\n\t"                                \
    "v_exp_f32 %[t1], %[t1]\n\t"                                \
    "v_add_f32 %[l], %[l], %[t0]\n\t"                           \
    "v_cvt_pk_fp8_f32 %[" w "], %[t0], %[t1] op_sel:[0,0,1]\n\t" \
    "v_add_f32 %[l], %[l], %[t1]\n\t"

// ---- A stage (asm part): n0, n1 = S^T of the two 32-key blocks of the tile whose K image starts KOFF
// bytes into LDS; rmax = this lane's maximum over the current tile (sc0, sc1), taken beside the MFMAs.
template <int KOFF>
__device__ __forceinline__ void f8_stage_a(f32x16& n0, f32x16& n1, float& rmax, const uint32_t (&ka)[2][2], const f32x16& sc0,
                                           const f32x16& sc1)
{
    constexpr int HALFK = 32 * kF8D;
    asm volatile(
        "ds_read_b128 a[96:99], %[k00] offset:%c[o0]\n\t"
        "ds_read_b128 a[100:103], %[k01] offset:%c[o0]\n\t"
        "ds_read_b128 a[104:107], %[k10] offset:%c[o0]\n\t"
        "ds_read_b128 a[108:111], %[k11] offset:%c[o0]\n\t"
        "ds_read_b128 a[112:115], %[k00] offset:%c[o1]\n\t"
        "ds_read_b128 a[116:119], %[k01] offset:%c[o1]\n\t"
        "ds_read_b128 a[120:123], %[k10] offset:%c[o1]\n\t"
        "ds_read_b128 a[124:127], %[k11] offset:%c[o1]\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        "v_mfma_f32_32x32x64_f8f6f4 %[sn], a[96:103], a[64:71], 0\n\t"
        "v_max3_f32 %[rm], %[s0], %[s1], %[s2]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s3], %[s4]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s5], %[s6]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s7], %[s8]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_mfma_f32_32x32x64_f8f6f4 %[sn], a[104:111], a[72:79], %[sn]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s9], %[s10]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s11], %[s12]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s13], %[s14]\n\t"
        "v_max_f32 %[rm], %[rm], %[s15]"
        : [sn] "=&v"(n0), [rm] "=&v"(rmax)
        : [k00] "v"(ka[0][0]), [k01] "v"(ka[0][1]), [k10] "v"(ka[1][0]), [k11] "v"(ka[1][1]), [o0] "i"(KOFF),
          [o1] "i"(KOFF + HALFK),
          [s0] "v"(sc0[0]), [s1] "v"(sc0[1]), [s2] "v"(sc0[2]), [s3] "v"(sc0[3]), [s4] "v"(sc0[4]), [s5] "v"(sc0[5]),
          [s6] "v"(sc0[6]), [s7] "v"(sc0[7]), [s8] "v"(sc0[8]), [s9] "v"(sc0[9]), [s10] "v"(sc0[10]), [s11] "v"(sc0[11]),
          [s12] "v"(sc0[12]), [s13] "v"(sc0[13]), [s14] "v"(sc0[14]), [s15] "v"(sc0[15])
        : FA2_F8_CLOBBERS);
    asm volatile(
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_mfma_f32_32x32x64_f8f6f4 %[sn], a[112:119], a[64:71], 0\n\t"
        "v_max3_f32 %[rm], %[rm], %[s0], %[s1]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s2], %[s3]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s4], %[s5]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s6], %[s7]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mfma_f32_32x32x64_f8f6f4 %[sn], a[120:127], a[72:79], %[sn]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s8], %[s9]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s10], %[s11]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s12], %[s13]\n\t"
        "v_max3_f32 %[rm], %[rm], %[s14], %[s15]"
        : [sn] "=&v"(n1), [rm] "+v"(rmax)
        : [s0] "v"(sc1[0]), [s1] "v"(sc1[1]), [s2] "v"(sc1[2]), [s3] "v"(sc1[3]), [s4] "v"(sc1[4]), [s5] "v"(sc1[5]),
          [s6] "v"(sc1[6]), [s7] "v"(sc1[7]), [s8] "v"(sc1[8]), [s9] "v"(sc1[9]), [s10] "v"(sc1[10]), [s11] "v"(sc1[11]),
          [s12] "v"(sc1[12]), [s13] "v"(sc1[13]), [s14] "v"(sc1[14]), [s15] "v"(sc1[15])
        : FA2_F8_CLOBBERS);
}

// ---- B stage: O^T += V^T P^T of the previous tile (V^T image starts VOFF bytes into the V region) with the
// exponentials, packing and sum of the current tile beside the four MFMAs.
template <int VOFF>
__device__ __forceinline__ void f8_stage_b(float& l_run, uint32_t (&pw)[8], const uint32_t (&va)[2], float c2, float mb,
                                           const f32x16& sc0, const f32x16& sc1)
{
    float t0, t1;
    asm volatile(
        "ds_read_b128 a[96:99], %[v0] offset:%c[o0]\n\t"
        "ds_read_b128 a[100:103], %[v1] offset:%c[o0]\n\t"
        "ds_read_b128 a[104:107], %[v0] offset:%c[o1]\n\t"
        "ds_read_b128 a[108:111], %[v1] offset:%c[o1]\n\t"
        "ds_read_b128 a[112:115], %[v0] offset:%c[o2]\n\t"
        "ds_read_b128 a[116:119], %[v1] offset:%c[o2]\n\t"
        "ds_read_b128 a[120:123], %[v0] offset:%c[o3]\n\t"
        "ds_read_b128 a[124:127], %[v1] offset:%c[o3]\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        "v_mfma_f32_32x32x64_f8f6f4 a[0:15], a[96:103], a[80:87], a[0:15]\n\t"
        FA2_F8_QUAD("s0", "s1", "s2", "s3", "w0")
        FA2_F8_QUAD("s4", "s5", "s6", "s7", "w1")
        : [l] "+v"(l_run), [t0] "=&v"(t0), [t1] "=&v"(t1), [w0] "=&v"(pw[0]), [w1] "=&v"(pw[1])
        : [v0] "v"(va[0]), [v1] "v"(va[1]), [o0] "i"(VOFF), [o1] "i"(VOFF + 2048), [o2] "i"(VOFF + 4096),
          [o3] "i"(VOFF + 6144), [c2] "v"(c2), [mb] "v"(mb),
          [s0] "v"(sc0[0]), [s1] "v"(sc0[1]), [s2] "v"(sc0[2]), [s3] "v"(sc0[3]), [s4] "v"(sc0[4]), [s5] "v"(sc0[5]),
          [s6] "v"(sc0[6]), [s7] "v"(sc0[7])
        : FA2_F8_CLOBBERS);
    asm volatile(
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_mfma_f32_32x32x64_f8f6f4 a[16:31], a[104:111], a[80:87], a[16:31]\n\t"
        FA2_F8_QUAD("s0", "s1", "s2", "s3", "w0")
        FA2_F8_QUAD("s4", "s5", "s6", "s7", "w1")
        : [l] "+v"(l_run), [t0] "=&v"(t0), [t1] "=&v"(t1), [w0] "=&v"(pw[2]), [w1] "=&v"(pw[3])
        : [c2] "v"(c2), [mb] "v"(mb),
          [s0] "v"(sc0[8]), [s1] "v"(sc0[9]), [s2] "v"(sc0[10]), [s3] "v"(sc0[11]), [s4] "v"(sc0[12]), [s5] "v"(sc0[13]),
          [s6] "v"(sc0[14]), [s7] "v"(sc0[15])
        : FA2_F8_CLOBBERS);
    asm volatile(
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_mfma_f32_32x32x64_f8f6f4 a[32:47], a[112:119], a[80:87], a[32:47]\n\t"
        FA2_F8_QUAD("s0", "s1", "s2", "s3", "w0")
        FA2_F8_QUAD("s4", "s5", "s6", "s7", "w1")
        : [l] "+v"(l_run), [t0] "=&v"(t0), [t1] "=&v"(t1), [w0] "=&v"(pw[4]), [w1] "=&v"(pw[5])
        : [c2] "v"(c2), [mb] "v"(mb),
          [s0] "v"(sc1[0]), [s1] "v"(sc1[1]), [s2] "v"(sc1[2]), [s3] "v"(sc1[3]), [s4] "v"(sc1[4]), [s5] "v"(sc1[5]),
          [s6] "v"(sc1[6]), [s7] "v"(sc1[7])
        : FA2_F8_CLOBBERS);
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mfma_f32_32x32x64_f8f6f4 a[48:63], a[120:127], a[80:87], a[48:63]\n\t"
        FA2_F8_QUAD("s0", "s1", "s2", "s3", "w0")
        FA2_F8_QUAD("s4", "s5", "s6", "s7", "w1")
        : [l] "+v"(l_run), [t0] "=&v"(t0), [t1] "=&v"(t1), [w0] "=&v"(pw[6]), [w1] "=&v"(pw[7])
        : [c2] "v"(c2), [mb] "v"(mb),
          [s0] "v"(sc1[8]), [s1] "v"(sc1[9]), [s2] "v"(sc1[10]), [s3] "v"(sc1[11]), [s4] "v"(sc1[12]), [s5] "v"(sc1[13]),
          [s6] "v"(sc1[14]), [s7] "v"(sc1[15])
        : FA2_F8_CLOBBERS);
}

template <bool CAUSAL>
__global__ void __launch_bounds__(64 * kF8Waves, 1) fa2_fwd_fp8_kernel(FwdFp8Args p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = kF8D;                 // bytes per K row
    constexpr int KT = kF8KV * ROWB;           // K tile: 8 KiB
    constexpr int VT = kF8D * kF8KV;           // V^T tile: 128 rows of 64 B
    constexpr int VREG = kF8Bufs * KT;         // LDS: [3 K tiles][3 V^T tiles]
    constexpr int HALFK = 32 * ROWB;           // second 32-key block of a K tile
    constexpr int DT = kF8D / 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31;
    const int h = lane >> 5;

    const int N = p.N, Npad = p.Npad;
    const int nrb = (N + kF8Rows - 1) / kF8Rows;
    int head, rb;
    map_block(blockIdx.x, p.BH, nrb, head, rb);
    if (CAUSAL) rb = nrb - 1 - rb;

    const char* Qh = (const char*)p.Q + (size_t)head * N * ROWB;
    const char* Kh = (const char*)p.K + (size_t)head * N * ROWB;
    const char* Vth = (const char*)p.Vt + (size_t)head * kF8D * Npad;

    const int q0 = rb * kF8Rows + wave * 32;
    const int qrow = q0 + qi;
    const int qld = qrow < N ? qrow : N - 1;

    int ntiles = (N + kF8KV - 1) / kF8KV;
    if (CAUSAL) {
        const int last_q = min(rb * kF8Rows + kF8Rows - 1, N - 1);
        ntiles = min(ntiles, last_q / kF8KV + 1);
    }
    const int niter = ((ntiles + 1 + 2) / 3) * 3;      // whole triples, at least one (fully masked) tile past the real ones

    // ---- Q fragments -> AGPRs: k-step s takes bytes 64 s + 32 h .. + 31 of the row
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(Qh + (size_t)qld * ROWB + 64 * s + 32 * h);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(Qh + (size_t)qld * ROWB + 64 * s + 32 * h + 16);
        if (s == 0)
            asm volatile("v_accvgpr_write_b32 a64, %0\n\tv_accvgpr_write_b32 a65, %1\n\tv_accvgpr_write_b32 a66, %2\n\t"
                         "v_accvgpr_write_b32 a67, %3\n\tv_accvgpr_write_b32 a68, %4\n\tv_accvgpr_write_b32 a69, %5\n\t"
                         "v_accvgpr_write_b32 a70, %6\n\tv_accvgpr_write_b32 a71, %7"
                         : : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3])
                         : FA2_F8_CLOBBERS);
        else
            asm volatile("v_accvgpr_write_b32 a72, %0\n\tv_accvgpr_write_b32 a73, %1\n\tv_accvgpr_write_b32 a74, %2\n\t"
                         "v_accvgpr_write_b32 a75, %3\n\tv_accvgpr_write_b32 a76, %4\n\tv_accvgpr_write_b32 a77, %5\n\t"
                         "v_accvgpr_write_b32 a78, %6\n\tv_accvgpr_write_b32 a79, %7"
                         : : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3])
                         : FA2_F8_CLOBBERS);
    }
    static_for<16 * DT>([&](auto R) { f8_acc_write<decltype(R)::value>(0.0f); });

    // ---- running state (see fa2_fwd_bf16.hip)
    const float inv_scale = 1.0f / p.scale;
    float m_run = -INFINITY, l_run = 0.0f, mb = 0.0f, thr = -INFINITY;

    // ---- LDS-DMA staging: wave w issues K piece w (rows 8 w .. + 7 of the tile) and V^T piece w (d rows
    // 16 w .. + 15); the swizzle is applied to the SOURCE chunk, the LDS write is linear.
    const int krow = lane >> 3, kslot = lane & 7;
    const int doffK = krow * ROWB + 16 * (kslot ^ f8_fk(8 * wave + krow));
    const int vrow = lane >> 2, vslot = lane & 3;
    const int doffV = vrow * Npad + 16 * (vslot ^ f8_fv(vrow));
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, N * ROWB, 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vth, 0, kF8D * Npad, 0x00020000);
    auto stage_k = [&](int t, int slot) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (f8_lds_ptr_t)(smem + slot * KT + wave * 1024), 16, doffK,
                                                 (t * kF8KV + 8 * wave) * ROWB, 0, 0);
    };
    auto stage_v = [&](int t, int slot) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (f8_lds_ptr_t)(smem + VREG + slot * VT + wave * 1024), 16, doffV,
                                                 16 * wave * Npad + t * kF8KV, 0, 0);
    };

    const float c2 = p.scale * kLog2e;

    // ---- per-lane LDS addresses.  K fragment (kb, s), half i: row pi(qi) + 32 kb, chunk 4 s + 2 h + i;
    // V^T fragment dt: row 32 dt + qi, chunks h (keys 16 h ..) and 2 + h (keys 32 + 16 h ..).
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;
    const int prow = f8_pi(qi);
    uint32_t ka[2][2], va[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) ka[s][i] = lbase + prow * ROWB + 16 * ((4 * s + 2 * h + i) ^ f8_fk(prow));
    va[0] = lbase + VREG + qi * 64 + 16 * (h ^ f8_fv(qi));
    va[1] = lbase + VREG + qi * 64 + 16 * ((2 + h) ^ f8_fv(qi));

    f32x16 sc0, sc1;              // S^T of the current 64-key tile: key blocks 0 and 1
    // (packed e4m3 P of the previous tile lives in a[80:87]: word j = key block j / 4, registers 4 (j % 4) ..+3)
    float rmax = -INFINITY;

    // ---- A stage: S^T of the tile whose K image starts KOFF bytes into LDS; rmax = max over sc0, sc1
    auto stage_a = [&](auto KOFF_, int key0, f32x16& n0, f32x16& n1) {
        f8_stage_a<decltype(KOFF_)::value>(n0, n1, rmax, ka, sc0, sc1);
        const bool tail = key0 + kF8KV > N;
        bool diag = false;
        if (CAUSAL) diag = key0 + kF8KV - 1 > q0;
        if (tail || diag) {
            mfma_vgpr_settle(n1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k0 = key0 + 16 * h + r, k1 = k0 + 32;       // register r of half h = key 16 h + r of its block
                bool d0 = k0 >= N, d1 = k1 >= N;
                if (CAUSAL) { d0 = d0 || k0 > qrow; d1 = d1 || k1 > qrow; }
                if (d0) n0[r] = -INFINITY;
                if (d1) n1[r] = -INFINITY;
            }
        }
    };

    // ---- X stage: lazy softmax reference (fa2_fwd_bf16.hip)
    float alpha = 1.0f;
    auto stage_x = [&]() -> bool {
        bool need = false;
        if (__any(rmax > thr)) {
            asm volatile("; fa2-cold: new softmax reference");
            const float mx = half_max(rmax) * p.scale;
            const bool grow = mx > m_run + kF8RescaleThr;
            const bool any_grow = __any(grow);
            const float m_new = any_grow ? fmaxf(m_run, mx) : m_run;
            need = any_grow && __any(m_run != -INFINITY && m_new != m_run);
            alpha = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
            m_run = m_new;
            mb = m_new == -INFINITY ? 0.0f : m_new * kLog2e;
            thr = (m_new + kF8RescaleThr) * inv_scale;
            l_run *= alpha;
        }
        return need;
    };

    // ---- B stage: O^T += V^T P^T of the previous tile (V^T image starts VOFF bytes into the V region) with
    // the exponentials, packing and sum of the current tile beside the four MFMAs
    uint32_t pw[8];
    auto stage_b = [&](auto VOFF_) { f8_stage_b<decltype(VOFF_)::value>(l_run, pw, va, c2, mb, sc0, sc1); };

    // One 64-key step: A(T+1) on K buffer KB, X(T), B(T-1) on V^T buffer VB.
    auto step = [&](auto KOFF_, int key0, auto VOFF_) {
        f32x16 n0, n1;
        stage_a(KOFF_, key0, n0, n1);
        const bool need = stage_x();
        stage_b(VOFF_);
        if (need) {
            mfma_acc_settle();
            static_for<4 * DT>([&](auto R4) { f8_acc_scale4<4 * decltype(R4)::value>(alpha); });
        }
        // P of this tile becomes the B operand of the next step (a[80:87]); this step's four products were
        // issued long ago and have read theirs
        asm volatile("v_accvgpr_write_b32 a80, %0\n\tv_accvgpr_write_b32 a81, %1\n\tv_accvgpr_write_b32 a82, %2\n\t"
                     "v_accvgpr_write_b32 a83, %3\n\tv_accvgpr_write_b32 a84, %4\n\tv_accvgpr_write_b32 a85, %5\n\t"
                     "v_accvgpr_write_b32 a86, %6\n\tv_accvgpr_write_b32 a87, %7"
                     : : "v"(pw[0]), "v"(pw[1]), "v"(pw[2]), "v"(pw[3]), "v"(pw[4]), "v"(pw[5]), "v"(pw[6]), "v"(pw[7])
                     : FA2_F8_CLOBBERS);
        sc0 = n0; sc1 = n1;
    };

    // ---- schedule.  Step T: A on K[T+1], X(T), B on V^T[T-1].  K[t] and V^T[t] live in slots t mod 3 of
    // their rings.  In front of step T: wait until all but this wave's two newest DMAs have landed (those
    // of the previous barrier may still be in flight: every tile gets two steps to arrive), barrier, then
    // request K[T+3] into K[T]'s slot (read for the last time in step T-1) and V^T[T+1] into V^T[T-2]'s.
    stage_k(0, 0); stage_k(1, 1); stage_k(2, 2);
    stage_v(0, 0); stage_v(0, 2);          // slot 2 stands in for V^T[-1]: finite data under an all-zero P
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    static_for<8>([&](auto J) { f8_acc_write<80 + decltype(J)::value>(0.0f); });      // P of "tile -1" = 0
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc0[r] = 0.0f; sc1[r] = 0.0f; }
    {
        f32x16 n0, n1;
        stage_a(std::integral_constant<int, 0>{}, 0, n0, n1);
        mfma_vgpr_settle(n1);
        sc0 = n0; sc1 = n1;
    }
    auto tile = [&](auto B_, int T) {
        constexpr int B = decltype(B_)::value;
        constexpr int B1 = (B + 1) % kF8Bufs, B2 = (B + 2) % kF8Bufs;
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __syncthreads();
        stage_k(T + 3, B);
        stage_v(T + 1, B1);
        step(std::integral_constant<int, B1 * KT>{}, (T + 1) * kF8KV, std::integral_constant<int, B2 * VT>{});
    };
    for (int T = 0; T < niter; T += 3) {
        tile(std::integral_constant<int, 0>{}, T);
        tile(std::integral_constant<int, 1>{}, T + 1);
        tile(std::integral_constant<int, 2>{}, T + 2);
    }

    // ---- epilogue.  The lane half is recomputed (v_mbcnt) rather than kept from kernel entry: a value that is live
    // across the whole loop only to be used here gets spilled to scratch at this register budget.
    mfma_acc_settle();
    const int h_ep = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 5);
    const float l_tot = half_sum(l_run);
    const size_t qoff = (size_t)head * N + qrow;
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
    static_for<4 * DT>([&](auto G) {
        constexpr int dt = decltype(G)::value / 4, g = decltype(G)::value % 4;
        constexpr int R = dt * 16 + 4 * g;
        f32x4 v;
        v[0] = f8_acc_read<R>() * inv; v[1] = f8_acc_read<R + 1>() * inv;
        v[2] = f8_acc_read<R + 2>() * inv; v[3] = f8_acc_read<R + 3>() * inv;
        if (qrow < N) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4*>((char*)p.O + qoff * (kF8D * 2) + 2 * (32 * dt + 8 * g + 4 * h_ep)) = o;
        }
    });
    if (qrow < N && h_ep == 0) p.L[qoff] = m_run + __builtin_logf(l_tot);
}
#undef FA2_F8_QUAD

hipError_t launch_fwd_fp8(const FwdFp8Args& a, hipStream_t stream)
{
    if (a.d != kF8D) return hipErrorInvalidValue;
    constexpr int lds = kF8Bufs * (kF8KV * kF8D + kF8D * kF8KV);
    hipLaunchKernelGGL(fa2_fp8_transpose_kernel, dim3((unsigned)(a.Npad / 64), (unsigned)a.BH), dim3(256), 0, stream,
                       (const unsigned char*)a.V, (unsigned char*)a.Vt, a.N, a.Npad);
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
