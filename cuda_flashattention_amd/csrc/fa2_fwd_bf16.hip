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
//   * A wave that is alone on its SIMD can issue only one instruction every ~7 clocks (any kind:
//     tools/probes/issue_cost.hip), while a 32x32x16 MFMA keeps the matrix pipe busy for 32; the
//     softmax needs ~8 instructions per MFMA, so a one-wave-per-SIMD kernel is issue-bound at about
//     half the matrix rate whatever its schedule.  Two waves per SIMD issue alternately (~4.7 clocks
//     per instruction together) and that is what this kernel is built around: workgroup = 8 waves
//     = 256 query rows of one head, each wave owns 32 rows and at most 128 VGPRs + 128 AGPRs.
//   * Everything only the matrix pipe touches lives in AGPRs with literal names that the kernel, not
//     hipcc, allocates: the wave's O^T accumulators (D/32 tiles), its Q fragments (B operand of
//     S^T = K Q^T) and eight 4-register slots that ds_read_b128 / ds_read_b64_tr_b16 fill with K
//     and V^T fragments straight from LDS.  The VGPRs hold what the VALU works on: S^T of the
//     current and of the next half-tile, packed P, addresses.
//   * S^T = K Q^T ("swapped" product): the accumulator column (= lane & 31) is the QUERY row, its
//     16 registers are keys, so a row's max and sum are in-lane reductions plus one
//     permlane32_swap -- the reference's (Bc + d) shuffle butterflies per row per tile are gone.
//     P stays in registers: the S^T accumulator, exponentiated and packed to bf16, IS the B operand
//     of O^T += V^T P^T.
//   * Software pipeline over 32-key half-tiles u:  A(u+1): S^T of the next half-tile (MFMAs, with the
//     in-lane maxima of half-tile u beside them), X(u): decision on the softmax reference, B(u-1):
//     O^T += V^T P^T of the previous one (MFMAs, with the exponentials / packing / sums of u beside
//     them).  Each stage is a few long asm statements, so the instruction order, the LDS waits and
//     the (absent) pad nops are exactly as written.
//   * The softmax reference (running max) is lazy: see stage_x.
//   * K/V tiles (64 keys) arrive by LDS-DMA (buffer_load ... lds, 16 B per lane, issued two tiles
//     ahead) into a ring of three XOR-swizzled images (fa2_common.h: lds_off; the swizzle is
//     applied to the SOURCE address, the LDS write is linear); the buffer resource's range check
//     zero-fills keys past the end; one barrier per 64-key tile.
//   * exp2 domain (v_exp_f32), running max kept in natural units so L matches the reference's
//     natural-log LSE; work mapping is XCD-aware (fa2_common.h: map_block).
#include <type_traits>

#include "fa2_common.h"
#include "fa2_launch.h"

namespace fa2 {

constexpr int kFwdWaves = 8;
constexpr int kFwdRows = 32 * kFwdWaves;   // query rows per workgroup
// keys per DMA tile (two or four 32-key half-tile steps).  64 everywhere; -DFA2_FWD_KV64=128 builds the d = 64 kernels with
// 128-key tiles (16 KiB per tile like d = 128).  Measured at (4,16,4096,64): no faster (866 vs 860 TFLOP/s on one box) --
// the d = 64 forward is bound by VALU issue, not by the tile barrier -- and slower with a causal mask (coarser diagonal).
#ifndef FA2_FWD_KV64
#define FA2_FWD_KV64 64
#endif
template <int D> constexpr int fwd_kv() { return D == 64 ? FA2_FWD_KV64 : 64; }
constexpr int kFwdBufs = 3;                 // LDS ring depth
constexpr float kRescaleThr = 6.0f;         // natural-log units of the scaled score

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// The wave's accumulator file: a0..a127, all of it named by the asm below.
#define FA2_ACC128_CLOBBERS \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
    "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", \
    "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", \
    "a124", "a125", "a126", "a127"

// AGPR map.  O^T tile dt: a[16 dt .. +15] (register 4g + e = O[q][32 dt + 8 g + 4 h + e]);
// Q fragment of k-step s: a[64 + 4 s .. +3]; fragment slot i: a[96 + 4 i .. +3].
constexpr int A_O = 0;
constexpr int A_QF = 64;
constexpr int A_F = 96;

template <int R>
__device__ __forceinline__ void a128_write(float x)
{
    asm volatile("v_accvgpr_write_b32 a[%c1], %0" : : "v"(x), "i"(R) : FA2_ACC128_CLOBBERS);
}
template <int R>
__device__ __forceinline__ float a128_read()
{
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(R));
    return x;
}
template <int LO>
__device__ __forceinline__ void a128_write_frag(bf16x8 f)
{
    const u32x4 w = __builtin_bit_cast(u32x4, f);
    asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\t"
                 "v_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
                 : : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "i"(LO), "i"(LO + 1), "i"(LO + 2), "i"(LO + 3)
                 : FA2_ACC128_CLOBBERS);
}
template <int R>
__device__ __forceinline__ void a128_scale4(float alpha)
{
    float t0, t1, t2, t3;
    asm volatile("v_accvgpr_read_b32 %0, a[%c5]\n\tv_accvgpr_read_b32 %1, a[%c6]\n\t"
                 "v_accvgpr_read_b32 %2, a[%c7]\n\tv_accvgpr_read_b32 %3, a[%c8]\n\t"
                 "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4\n\t"
                 "v_accvgpr_write_b32 a[%c5], %0\n\tv_accvgpr_write_b32 a[%c6], %1\n\t"
                 "v_accvgpr_write_b32 a[%c7], %2\n\tv_accvgpr_write_b32 a[%c8], %3"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                 : "v"(alpha), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3) : FA2_ACC128_CLOBBERS);
}

// ---- asm building blocks.  Operand numbering is spelled out at each statement; "%cN" prints an
// immediate operand bare (AGPR index / LDS offset).
//
// Hazards the statements are written around (hipcc does not look inside asm):
//   * LDS data is used only behind an s_waitcnt lgkmcnt(n) that counts the reads issued after it;
//   * a v_exp_f32 result is not read by the very next instruction (trans forwarding);
//   * back-to-back MFMAs accumulate into the same tile (exact SrcC = vDst overlap), which the
//     hardware interlocks; MFMA results in VGPRs (S^T) are read by the VALU a whole stage later, or
//     behind mfma_vgpr_settle();
//   * no MFMA source is written by the VALU within the two instructions before it.
// ---- A stage (asm part): snext = S^T of the 32 keys whose K rows start KOFF bytes into the K region
// (fragment addresses ka[s] + KOFF); rmax = this lane's maximum over scur, taken beside the MFMAs.
// Two statements of four k-steps at D = 128 (the first issues all eight fragment reads), one at D = 64.
template <int D, int KOFF>
__device__ __forceinline__ void fwd_stage_a(f32x16& snext, float& rmax, const uint32_t (&ka)[D / 16], const f32x16& scur)
{
    if constexpr (D == 128) {
        asm volatile(
            "ds_read_b128 a[96:99], %[k0] offset:%c[off]\n\t"
            "ds_read_b128 a[100:103], %[k1] offset:%c[off]\n\t"
            "ds_read_b128 a[104:107], %[k2] offset:%c[off]\n\t"
            "ds_read_b128 a[108:111], %[k3] offset:%c[off]\n\t"
            "ds_read_b128 a[112:115], %[k4] offset:%c[off]\n\t"
            "ds_read_b128 a[116:119], %[k5] offset:%c[off]\n\t"
            "ds_read_b128 a[120:123], %[k6] offset:%c[off]\n\t"
            "ds_read_b128 a[124:127], %[k7] offset:%c[off]\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[96:99], a[64:67], 0\n\t"
            "v_max_f32 %[rm], %[s0], %[s1]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[100:103], a[68:71], %[sn]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s2], %[s3]\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[104:107], a[72:75], %[sn]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s4], %[s5]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[108:111], a[76:79], %[sn]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s6], %[s7]"
            : [sn] "=&v"(snext), [rm] "=&v"(rmax)
            : [k0] "v"(ka[0]), [k1] "v"(ka[1]), [k2] "v"(ka[2]), [k3] "v"(ka[3]), [k4] "v"(ka[4]), [k5] "v"(ka[5]),
              [k6] "v"(ka[6]), [k7] "v"(ka[7]), [off] "i"(KOFF),
              [s0] "v"(scur[0]), [s1] "v"(scur[1]), [s2] "v"(scur[2]), [s3] "v"(scur[3]), [s4] "v"(scur[4]),
              [s5] "v"(scur[5]), [s6] "v"(scur[6]), [s7] "v"(scur[7])
            : FA2_ACC128_CLOBBERS);
        asm volatile(
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[112:115], a[80:83], %[sn]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s0], %[s1]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[116:119], a[84:87], %[sn]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s2], %[s3]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[120:123], a[88:91], %[sn]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s4], %[s5]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[124:127], a[92:95], %[sn]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s6], %[s7]"
            : [sn] "+v"(snext), [rm] "+v"(rmax)
            : [s0] "v"(scur[8]), [s1] "v"(scur[9]), [s2] "v"(scur[10]), [s3] "v"(scur[11]), [s4] "v"(scur[12]),
              [s5] "v"(scur[13]), [s6] "v"(scur[14]), [s7] "v"(scur[15])
            : FA2_ACC128_CLOBBERS);
    } else {
        asm volatile(
            "ds_read_b128 a[96:99], %[k0] offset:%c[off]\n\t"
            "ds_read_b128 a[100:103], %[k1] offset:%c[off]\n\t"
            "ds_read_b128 a[104:107], %[k2] offset:%c[off]\n\t"
            "ds_read_b128 a[108:111], %[k3] offset:%c[off]\n\t"
            "s_waitcnt lgkmcnt(3)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[96:99], a[64:67], 0\n\t"
            "v_max_f32 %[rm], %[s0], %[s1]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s2], %[s3]\n\t"
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[100:103], a[68:71], %[sn]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s4], %[s5]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s6], %[s7]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[104:107], a[72:75], %[sn]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s8], %[s9]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s10], %[s11]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[sn], a[108:111], a[76:79], %[sn]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s12], %[s13]\n\t"
            "v_max3_f32 %[rm], %[rm], %[s14], %[s15]"
            : [sn] "=&v"(snext), [rm] "=&v"(rmax)
            : [k0] "v"(ka[0]), [k1] "v"(ka[1]), [k2] "v"(ka[2]), [k3] "v"(ka[3]), [off] "i"(KOFF),
              [s0] "v"(scur[0]), [s1] "v"(scur[1]), [s2] "v"(scur[2]), [s3] "v"(scur[3]), [s4] "v"(scur[4]),
              [s5] "v"(scur[5]), [s6] "v"(scur[6]), [s7] "v"(scur[7]), [s8] "v"(scur[8]), [s9] "v"(scur[9]),
              [s10] "v"(scur[10]), [s11] "v"(scur[11]), [s12] "v"(scur[12]), [s13] "v"(scur[13]), [s14] "v"(scur[14]),
              [s15] "v"(scur[15])
            : FA2_ACC128_CLOBBERS);
    }
}

// ---- B stage: O^T += V^T P^T (V rows start VOFF bytes into the V region, fragment addresses
// va[dt][half] + VOFF, P = pprev) with the exponentials of scur, their packing to bf16 (pw: word
// 4s + i = registers 8s + 2i, 8s + 2i + 1 -- the next step's pprev) and their sum beside the MFMAs.
// One softmax pair: t0,t1 = exp2(s * c2 - mb); l += t0 + t1; w = pack(t0, t1).
#define FA2_SOFTMAX_PAIR(sa, sb, w)                         \
"v_fma_f32 %[t0], %[" sa "], %[c2], -%[mb]\n\t"        \
"v_fma_f32 %[t1], %[" sb "], %[c2], -%[mb]\n\t"        \
"v_exp_f32 %[t0], %[t0]\n\t"                            \
"v_exp_f32 %[t1], %[t1]\n\t"                            \
"v_add_f32 %[l], %[l], %[t0]\n\t"                       \
"v_cvt_pk_bf16_f32 %[" w "], %[t0], %[t1]\n\t"          \
"v_add_f32 %[l], %[l], %[t1]\n\t"
template <int D, int VOFF>
__device__ __forceinline__ void fwd_stage_b(float& l_run, uint32_t (&pw)[8], const uint32_t (&va)[D / 32][2],
                                            const bf16x8 (&pprev)[2], float c2, float mb, const f32x16& scur)
{
    constexpr int ROWB = D * 2;
    constexpr int SP1 = VOFF + 16 * ROWB;          // second k-step: 16 rows further
    float t0, t1;
    if constexpr (D == 128) {
        // fragments (dt, sp) -> slot 2 dt + sp; all sixteen transposed reads first
        asm volatile(
            "ds_read_b64_tr_b16 a[96:97], %[v00] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[98:99], %[v01] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[100:101], %[v00] offset:%c[o1]\n\t"
            "ds_read_b64_tr_b16 a[102:103], %[v01] offset:%c[o1]\n\t"
            "ds_read_b64_tr_b16 a[104:105], %[v10] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[106:107], %[v11] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[108:109], %[v10] offset:%c[o1]\n\t"
            "ds_read_b64_tr_b16 a[110:111], %[v11] offset:%c[o1]\n\t"
            "ds_read_b64_tr_b16 a[112:113], %[v20] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[114:115], %[v21] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[116:117], %[v20] offset:%c[o1]\n\t"
            "ds_read_b64_tr_b16 a[118:119], %[v21] offset:%c[o1]\n\t"
            "ds_read_b64_tr_b16 a[120:121], %[v30] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[122:123], %[v31] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[124:125], %[v30] offset:%c[o1]\n\t"
            "ds_read_b64_tr_b16 a[126:127], %[v31] offset:%c[o1]\n\t"
            "s_waitcnt lgkmcnt(12)\n\t"
            "v_mfma_f32_32x32x16_bf16 a[0:15], a[96:99], %[p0], a[0:15]\n\t"
            FA2_SOFTMAX_PAIR("s0", "s1", "w0")
            "v_mfma_f32_32x32x16_bf16 a[0:15], a[100:103], %[p1], a[0:15]\n\t"
            FA2_SOFTMAX_PAIR("s2", "s3", "w1")
            "s_waitcnt lgkmcnt(8)\n\t"
            "v_mfma_f32_32x32x16_bf16 a[16:31], a[104:107], %[p0], a[16:31]\n\t"
            FA2_SOFTMAX_PAIR("s4", "s5", "w2")
            "v_mfma_f32_32x32x16_bf16 a[16:31], a[108:111], %[p1], a[16:31]\n\t"
            FA2_SOFTMAX_PAIR("s6", "s7", "w3")
            : [l] "+v"(l_run), [t0] "=&v"(t0), [t1] "=&v"(t1), [w0] "=&v"(pw[0]), [w1] "=&v"(pw[1]), [w2] "=&v"(pw[2]),
              [w3] "=&v"(pw[3])
            : [v00] "v"(va[0][0]), [v01] "v"(va[0][1]), [v10] "v"(va[1][0]), [v11] "v"(va[1][1]), [v20] "v"(va[2][0]),
              [v21] "v"(va[2][1]), [v30] "v"(va[3][0]), [v31] "v"(va[3][1]), [o0] "i"(VOFF), [o1] "i"(SP1),
              [p0] "v"(pprev[0]), [p1] "v"(pprev[1]), [c2] "v"(c2), [mb] "v"(mb),
              [s0] "v"(scur[0]), [s1] "v"(scur[1]), [s2] "v"(scur[2]), [s3] "v"(scur[3]), [s4] "v"(scur[4]),
              [s5] "v"(scur[5]), [s6] "v"(scur[6]), [s7] "v"(scur[7])
            : FA2_ACC128_CLOBBERS);
        asm volatile(
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 a[32:47], a[112:115], %[p0], a[32:47]\n\t"
            FA2_SOFTMAX_PAIR("s0", "s1", "w0")
            "v_mfma_f32_32x32x16_bf16 a[32:47], a[116:119], %[p1], a[32:47]\n\t"
            FA2_SOFTMAX_PAIR("s2", "s3", "w1")
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_bf16 a[48:63], a[120:123], %[p0], a[48:63]\n\t"
            FA2_SOFTMAX_PAIR("s4", "s5", "w2")
            "v_mfma_f32_32x32x16_bf16 a[48:63], a[124:127], %[p1], a[48:63]\n\t"
            FA2_SOFTMAX_PAIR("s6", "s7", "w3")
            : [l] "+v"(l_run), [t0] "=&v"(t0), [t1] "=&v"(t1), [w0] "=&v"(pw[4]), [w1] "=&v"(pw[5]), [w2] "=&v"(pw[6]),
              [w3] "=&v"(pw[7])
            : [p0] "v"(pprev[0]), [p1] "v"(pprev[1]), [c2] "v"(c2), [mb] "v"(mb),
              [s0] "v"(scur[8]), [s1] "v"(scur[9]), [s2] "v"(scur[10]), [s3] "v"(scur[11]), [s4] "v"(scur[12]),
              [s5] "v"(scur[13]), [s6] "v"(scur[14]), [s7] "v"(scur[15])
            : FA2_ACC128_CLOBBERS);
    } else {
        asm volatile(
            "ds_read_b64_tr_b16 a[96:97], %[v00] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[98:99], %[v01] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[100:101], %[v00] offset:%c[o1]\n\t"
            "ds_read_b64_tr_b16 a[102:103], %[v01] offset:%c[o1]\n\t"
            "ds_read_b64_tr_b16 a[104:105], %[v10] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[106:107], %[v11] offset:%c[o0]\n\t"
            "ds_read_b64_tr_b16 a[108:109], %[v10] offset:%c[o1]\n\t"
            "ds_read_b64_tr_b16 a[110:111], %[v11] offset:%c[o1]\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "v_mfma_f32_32x32x16_bf16 a[0:15], a[96:99], %[p0], a[0:15]\n\t"
            FA2_SOFTMAX_PAIR("s0", "s1", "w0")
            FA2_SOFTMAX_PAIR("s2", "s3", "w1")
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 a[0:15], a[100:103], %[p1], a[0:15]\n\t"
            FA2_SOFTMAX_PAIR("s4", "s5", "w2")
            FA2_SOFTMAX_PAIR("s6", "s7", "w3")
            : [l] "+v"(l_run), [t0] "=&v"(t0), [t1] "=&v"(t1), [w0] "=&v"(pw[0]), [w1] "=&v"(pw[1]), [w2] "=&v"(pw[2]),
              [w3] "=&v"(pw[3])
            : [v00] "v"(va[0][0]), [v01] "v"(va[0][1]), [v10] "v"(va[1][0]), [v11] "v"(va[1][1]), [o0] "i"(VOFF),
              [o1] "i"(SP1), [p0] "v"(pprev[0]), [p1] "v"(pprev[1]), [c2] "v"(c2), [mb] "v"(mb),
              [s0] "v"(scur[0]), [s1] "v"(scur[1]), [s2] "v"(scur[2]), [s3] "v"(scur[3]), [s4] "v"(scur[4]),
              [s5] "v"(scur[5]), [s6] "v"(scur[6]), [s7] "v"(scur[7])
            : FA2_ACC128_CLOBBERS);
        asm volatile(
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_mfma_f32_32x32x16_bf16 a[16:31], a[104:107], %[p0], a[16:31]\n\t"
            FA2_SOFTMAX_PAIR("s0", "s1", "w0")
            FA2_SOFTMAX_PAIR("s2", "s3", "w1")
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_bf16 a[16:31], a[108:111], %[p1], a[16:31]\n\t"
            FA2_SOFTMAX_PAIR("s4", "s5", "w2")
            FA2_SOFTMAX_PAIR("s6", "s7", "w3")
            : [l] "+v"(l_run), [t0] "=&v"(t0), [t1] "=&v"(t1), [w0] "=&v"(pw[4]), [w1] "=&v"(pw[5]), [w2] "=&v"(pw[6]),
              [w3] "=&v"(pw[7])
            : [p0] "v"(pprev[0]), [p1] "v"(pprev[1]), [c2] "v"(c2), [mb] "v"(mb),
              [s0] "v"(scur[8]), [s1] "v"(scur[9]), [s2] "v"(scur[10]), [s3] "v"(scur[11]), [s4] "v"(scur[12]),
              [s5] "v"(scur[13]), [s6] "v"(scur[14]), [s7] "v"(scur[15])
            : FA2_ACC128_CLOBBERS);
    }
}
#undef FA2_SOFTMAX_PAIR

template <int R>
__device__ __forceinline__ void fwd_acc_zero(u32x4 z)
{
    // (the compiler does not know this statement is an MFMA: the wait states between its writes of z and the read are ours)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, %0, 0" : : "v"(z), "i"(R), "i"(R + 15) : FA2_ACC128_CLOBBERS);
}

template <int D, bool CAUSAL, bool STATE>
__global__ void __launch_bounds__(64 * kFwdWaves, 1) fa2_fwd_bf16_kernel(FwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = D * 2;               // bytes per tile row
    constexpr int kFwdKV = fwd_kv<D>();       // keys per DMA tile
    constexpr int NH = kFwdKV / 32;           // 32-key half-tile steps per tile: 2 or 4
    constexpr int TILEB = kFwdKV * ROWB;      // bytes per K (or V) tile
    constexpr int VREG = kFwdBufs * TILEB;    // LDS: [3 K tiles][3 V tiles]; every read offset < 64 KiB from its region base
    constexpr int HALFB = 32 * ROWB;          // one 32-key half-tile
    constexpr int CPR = D / 8;                // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;             // rows per DMA wave-instruction (1 KiB)
    constexpr int NP = kFwdKV / RPI;          // DMA pieces per tensor per tile (1 KiB each)
    constexpr int PPW = 2 * NP / kFwdWaves;   // DMA pieces per wave per tile: 4 or 2
    constexpr int KS = D / 16;                // k-steps of QK^T
    constexpr int DT = D / 32;                // 32-column tiles of O

#ifdef FA2_DIAG_BLOCKS
    unsigned long long bk_t0, bk_t1 = 0, bk_t2 = 0;
    unsigned bk_hw;
    asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_waitcnt lgkmcnt(0)" : "=s"(bk_t0), "=s"(bk_hw) :: "memory");
#endif
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
    const size_t qhs = p.q_hs ? p.q_hs : Nq, khs = p.k_hs ? p.k_hs : Nk;     // rows between heads
    const char* Qh = (const char*)p.Q + (size_t)head * qhs * ROWB;
    const char* Kh = (const char*)p.K + (size_t)head * khs * ROWB;
    const char* Vh = (const char*)p.V + (size_t)head * khs * ROWB;

    const int q0 = rb * kFwdRows + wave * 32;          // first query row of this wave
    const int qrow = q0 + qi;
    const int qld = qrow < Nq ? qrow : Nq - 1;         // clamped for loads (pad, don't mask)

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

    // ---- LDS-DMA staging: wave w issues pieces w, w + 8, ... of the 2 * NP pieces of a tile (the
    // first NP are K, the rest V).  The swizzle term depends on the row modulo 16 only, hence is the
    // same for all pieces of a wave: one per-lane voffset, everything else wave-uniform (soffset);
    // rows >= Nk read as zeros.
    const int drow = lane / CPR;
    const int dslot = lane % CPR;
    const int prow = wave * RPI + drow;
    const int doff = drow * ROWB + 16 * ((lds_off<D>(prow, dslot) - ROWB * prow) >> 4);
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, Nk * ROWB, 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, Nk * ROWB, 0x00020000);
    auto stage = [&](int t, int buf) {
        char* b = smem + buf * TILEB;
        static_for<PPW>([&](auto I) {
            constexpr int i = decltype(I)::value;
            constexpr bool isv = (kFwdWaves * i) >= NP;            // wave + 8 i < NP  <=>  8 i < NP
            const int piece = wave + kFwdWaves * i - (isv ? NP : 0);
            const int soff = (t * kFwdKV + piece * RPI) * ROWB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isv ? v_rsrc : k_rsrc, (lds_ptr_t)(b + (isv ? VREG : 0) + piece * 1024),
                                                     16, doff, soff, 0, 0);
        });
    };
    // ---- prologue DMA first: tiles 0 and 1 into buffers 0 and 1; buffer 2 (read by the first, all-zero-P PV step) must
    // hold finite data: tile 0 again.  Issued before the Q fragments and the accumulators are set up, so that their
    // latency runs beside that work instead of after it.
    stage(0, 0);
    stage(1, 1);
    stage(0, 2);

    // ---- Q fragments -> AGPRs; lane holds Q[q][16s + 8h .. +7].
    static_for<KS>([&](auto S) {
        constexpr int sidx = decltype(S)::value;
        a128_write_frag<A_QF + sidx * 4>(*reinterpret_cast<const bf16x8*>(Qh + (size_t)qld * ROWB + 16 * (2 * sidx + h)));
    });

    // ---- running state.  m_run: the reference maximum (natural units), mb = m_run * log2(e) (0 while
    // -inf), thr = the raw (unscaled) score above which the row asks for a new reference; l_run: this
    // lane's share of the row sum.
    const float inv_scale = 1.0f / p.scale;
    float m_run, l_run, mb, thr;
    if (STATE && p.resume) {
        const float* Oa = p.Oacc + ((size_t)head * qhs + qld) * D;
        static_for<4 * DT>([&](auto G) {
            constexpr int dt = decltype(G)::value / 4, g = decltype(G)::value % 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(Oa + 32 * dt + 8 * g + 4 * h);
            static_for<4>([&](auto E) {
                constexpr int e = decltype(E)::value;
                a128_write<A_O + dt * 16 + 4 * g + e>(v[e]);
            });
        });
        m_run = p.M[(size_t)head * qhs + qld];
        l_run = h == 0 ? p.L[(size_t)head * qhs + qld] : 0.0f;
        mb = m_run == -INFINITY ? 0.0f : m_run * kLog2e;
        thr = (m_run + kRescaleThr) * inv_scale;
    } else {
        const u32x4 z = {0u, 0u, 0u, 0u};            // O^T <- 0: DT MFMAs on a zero fragment instead of 16 DT accumulator writes
        static_for<DT>([&](auto T) { fwd_acc_zero<A_O + 16 * decltype(T)::value>(z); });
        m_run = -INFINITY;
        l_run = 0.0f;
        mb = 0.0f;
        thr = -INFINITY;
    }

    auto tile_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA has landed ...
        __syncthreads();                                      // ... and so has everyone's
    };

    const float c2 = p.scale * kLog2e;   // exp(s * scale - m) = exp2(s * c2 - m * log2e)

    // ---- loop-invariant per-lane LDS addresses
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;
    const int trq = (lane & 15) >> 2;       // row inside the 4-row block of a transposed read
    const int trp = lane & 3;               // 4-column group inside the 16-column block
    const int trcb = (lane >> 4) & 1;       // which 16-column half of the 32-column tile
    uint32_t ka[KS], va[DT][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) ka[s] = lbase + lds_off<D>(qi, 2 * s + h);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)      // V region base folded in; +16 rows: sp = 1
            va[dt][jj] = lbase + VREG + lds_off<D>(8 * jj + 4 * h + trq, 4 * dt + 2 * trcb + (trp >> 1)) + 8 * (trp & 1);

    // Two waves share each SIMD; the second-dispatched half of the workgroup (waves 4-7) loses every VALU arbitration to
    // the older half (priority, then age).  One static priority bump for that half, and no per-stage flips, removes its
    // start-of-stage penalty (MI355X_MICROARCH.md, 'Two waves per SIMD' item 4).  `wave` is wave-uniform by construction
    // (readfirstlane), so this is a scalar branch around one s_setprio.
#ifndef FA2_NO_SETPRIO
    if (wave >= kFwdWaves / 2) __builtin_amdgcn_s_setprio(1);
#endif

    // ---- pipeline registers
    f32x16 scur;             // S^T of half-tile u     (keys on registers, query on the lane)
    bf16x8 pprev[2];         // packed P of half-tile u-1, per k-step
    float rmax = -INFINITY;  // this lane's maximum over scur (taken beside the S^T MFMAs)

    // ---- A stage: snext = S^T of the 32 keys whose K rows start KOFF bytes into the K region;
    // rmax = max over scur.  Two statements of four k-steps at D = 128 (the first issues all eight
    // fragment reads), one at D = 64.
    auto stage_a = [&](auto KOFF_, int key0, f32x16& snext) {
        fwd_stage_a<D, decltype(KOFF_)::value>(snext, rmax, ka, scur);
        const bool tail = key0 + 32 > Nk;
        bool diag = false;
        if (CAUSAL) diag = key0 + 31 > q0 + p.causal_shift;
        if (tail || diag) {
            mfma_vgpr_settle(snext);             // the products were issued a moment ago
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + acc_row(r, h);
                bool dead = key >= Nk;
                if (CAUSAL) dead = dead || key > qrow + p.causal_shift;
                if (dead) snext[r] = -INFINITY;
            }
        }
    };

    // ---- X stage: the softmax reference.  It is LAZY: the rows of the wave move to their current
    // maximum only when some row's score has risen more than kRescaleThr above its reference (or has
    // no reference yet) -- one compare on the common path.  In between, P = exp(s - m_ref) may exceed 1
    // (by at most e^kRescaleThr): harmless in fp32 sums and in bf16 P, whose relative precision does
    // not depend on magnitude -- and O, l and L come out the same.  Returns whether O must be scaled
    // by alpha once the pending products (all at the old reference) have been accumulated.
    float alpha = 1.0f;
    auto stage_x = [&]() -> bool {
        bool need = false;
        if (__any(rmax > thr)) {
            asm volatile("; fa2-cold: new softmax reference");
            const float mx = half_max(rmax) * p.scale;
            const bool grow = mx > m_run + kRescaleThr;        // also true from m_run = -inf
            const bool any_grow = __any(grow);
            const float m_new = any_grow ? fmaxf(m_run, mx) : m_run;
            // O only needs scaling if some row already accumulated something at an older reference
            need = any_grow && __any(m_run != -INFINITY && m_new != m_run);
            // first visible key of a row: m_run = -inf -> alpha = 0 (its accumulators are 0 anyway)
            alpha = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
            m_run = m_new;
            // a row that has seen no visible key yet keeps p = 0 (avoid inf - inf)
            mb = m_new == -INFINITY ? 0.0f : m_new * kLog2e;
            thr = (m_new + kRescaleThr) * inv_scale;
            l_run *= alpha;
        }
        return need;
    };

    // ---- B stage: O^T += V^T P^T of half-tile u-1 (V rows start VOFF bytes into the V region,
    // P = pprev) with the exponentials of half-tile u (scur), their packing to bf16 (pw: word 4s + i =
    // registers 8s + 2i, 8s + 2i + 1, the next step's pprev) and their sum beside the MFMAs.
    uint32_t pw[8];
    auto stage_b = [&](auto VOFF_) { fwd_stage_b<D, decltype(VOFF_)::value>(l_run, pw, va, pprev, c2, mb, scur); };

#ifdef FA2_DIAG_STAMPS
    unsigned long long dg_a = 0, dg_x = 0, dg_b = 0, dg_r = 0, dg_vm = 0, dg_bar = 0, dg_sync = 0, dg_t = 0, dg_loop = 0;
#define FA2_STAMP(acc) { unsigned long long ts_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) :: "memory"); acc += ts_ - dg_t; dg_t = ts_; }
#else
#define FA2_STAMP(acc)
#endif
    // One half-tile step: A(u+1), X(u), B(u-1), then the (rare) rescale and the rotation
    // pprev <- P(u), scur <- S(u+1).
    auto step = [&](auto KOFF_, int key0, auto VOFF_) {
        f32x16 snext;
        stage_a(KOFF_, key0, snext);
        FA2_STAMP(dg_a)
        const bool need = stage_x();
        FA2_STAMP(dg_x)
        stage_b(VOFF_);
        FA2_STAMP(dg_b)
        if (need) {         // everything accumulated so far (through half-tile u-1) is at the old reference
            mfma_acc_settle();
            static_for<4 * DT>([&](auto R4) { a128_scale4<A_O + 4 * decltype(R4)::value>(alpha); });
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const u32x4 w = {pw[4 * sp], pw[4 * sp + 1], pw[4 * sp + 2], pw[4 * sp + 3]};
            pprev[sp] = __builtin_bit_cast(bf16x8, w);
        }
        scur = snext;
        FA2_STAMP(dg_r)
    };

    // ---- the prologue tiles (issued at the top) have landed
    tile_barrier();
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int e = 0; e < 8; ++e) pprev[sp][e] = (__bf16)0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) scur[r] = 0.0f;
    {
        f32x16 s0;
        stage_a(std::integral_constant<int, 0>{}, 0, s0);
        mfma_vgpr_settle(s0);
        scur = s0;
    }

    // One tile T (NH half-tiles of 32 keys) living in ring buffer B (= T mod 3); half-tile step u = NH T + j runs
    // A on half-tile u + 1 and B on half-tile u - 1:
    //   step 0           : A on K[T] half 1            ; X ; B on V[T-1] last half (buffer B+2)
    //   barrier          : buffer B+2 is free, tile T+1 has landed -> DMA tile T+2 into B+2
    //   steps 1 .. NH-2  : A on K[T] half j+1          ; X ; B on V[T] half j-1          (d = 64 only: NH = 4)
    //   step NH-1        : A on K[T+1] half 0 (B+1)    ; X ; B on V[T] half NH-2
    auto tile = [&](auto B_, int T) {
        constexpr int B = decltype(B_)::value;
        constexpr int B1 = (B + 1) % kFwdBufs, B2 = (B + 2) % kFwdBufs;
        step(std::integral_constant<int, B * TILEB + HALFB>{}, T * kFwdKV + 32, std::integral_constant<int, B2 * TILEB + (NH - 1) * HALFB>{});
#ifdef FA2_DIAG_STAMPS
        FA2_STAMP(dg_r)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FA2_STAMP(dg_vm)
        __syncthreads();
        FA2_STAMP(dg_bar)
#else
        tile_barrier();
#endif
        stage(T + 2, B2);
        FA2_STAMP(dg_sync)
        static_for<NH - 2>([&](auto J) {
            constexpr int j = decltype(J)::value;
            step(std::integral_constant<int, B * TILEB + (j + 2) * HALFB>{}, T * kFwdKV + 32 * (j + 2),
                 std::integral_constant<int, B * TILEB + j * HALFB>{});
        });
        step(std::integral_constant<int, B1 * TILEB>{}, (T + 1) * kFwdKV, std::integral_constant<int, B * TILEB + (NH - 2) * HALFB>{});
    };

#ifdef FA2_DIAG_STAMPS
    { unsigned long long ts_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) :: "memory"); dg_t = ts_; dg_loop = ts_; }
#endif
#ifdef FA2_DIAG_BLOCKS
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bk_t1) :: "memory");
#endif
    for (int T = 0; T < niter; T += 3) {
        tile(std::integral_constant<int, 0>{}, T);
        tile(std::integral_constant<int, 1>{}, T + 1);
        tile(std::integral_constant<int, 2>{}, T + 2);
    }
#ifdef FA2_DIAG_STAMPS
    if (lane == 0) {   // diagnostic build only: cycle sums overwrite the head of L
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.L) + ((size_t)blockIdx.x * kFwdWaves + wave) * 8;
        dbg[0] = dg_a; dbg[1] = dg_x; dbg[2] = dg_b; dbg[3] = dg_r; dbg[4] = dg_sync; dbg[5] = dg_t - dg_loop; dbg[6] = dg_vm; dbg[7] = dg_bar;
    }
    return;
#endif

#ifdef FA2_DIAG_BLOCKS
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bk_t2) :: "memory");
#endif
    // ---- epilogue
    mfma_acc_settle();
    const float l_tot = half_sum(l_run);
    const size_t qoff = (size_t)head * qhs + qrow;
    const bool fin = !STATE || p.finalize;
    const float inv = fin ? (l_tot > 0.0f ? 1.0f / l_tot : 0.0f) : 1.0f;
    // a lane holds 4 consecutive columns of its row per register quad, its partner lane (+32) the next 4: for the bf16
    // output one v_permlane32_swap per packed dword pairs them up, so that every lane stores 16 contiguous bytes
    static_for<2 * DT>([&](auto G) {
        constexpr int dt = decltype(G)::value / 2, gp = decltype(G)::value % 2;
        constexpr int R = A_O + dt * 16 + 8 * gp;
        f32x4 v, w;
        v[0] = a128_read<R>() * inv; v[1] = a128_read<R + 1>() * inv;
        v[2] = a128_read<R + 2>() * inv; v[3] = a128_read<R + 3>() * inv;
        w[0] = a128_read<R + 4>() * inv; w[1] = a128_read<R + 5>() * inv;
        w[2] = a128_read<R + 6>() * inv; w[3] = a128_read<R + 7>() * inv;
        if (fin) {
            bf16x4 x, y;
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = (__bf16)v[e]; y[e] = (__bf16)w[e]; }
            const u32x2 xu = __builtin_bit_cast(u32x2, x), yu = __builtin_bit_cast(u32x2, y);
            const auto s0 = __builtin_amdgcn_permlane32_swap(xu[0], yu[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(xu[1], yu[1], false, false);
            const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            if (qrow < Nq) *reinterpret_cast<u32x4*>((char*)p.O + qoff * ROWB + 2 * (32 * dt + 16 * gp + 8 * h)) = o;
        } else if (qrow < Nq) {
            *reinterpret_cast<f32x4*>(p.Oacc + qoff * D + 32 * dt + 16 * gp + 4 * h) = v;
            *reinterpret_cast<f32x4*>(p.Oacc + qoff * D + 32 * dt + 16 * gp + 8 + 4 * h) = w;
        }
    });
#ifdef FA2_DIAG_BLOCKS
    if (false) {
#else
    if (qrow < Nq && h == 0) {
#endif
        if (fin) {
            p.L[qoff] = m_run + __builtin_logf(l_tot);
        } else {
            p.L[qoff] = l_tot;
            p.M[qoff] = m_run;
        }
    }
#ifdef FA2_DIAG_BLOCKS
    __syncthreads();
    if (tid == 0) {     // diagnostic build only: block timeline (100 MHz ticks) over the tail of L
        unsigned long long bk_t3;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bk_t3) :: "memory");
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.L + (size_t)p.BH * Nq) - 8 * (size_t)(gridDim.x - blockIdx.x);
        dbg[0] = bk_t0; dbg[1] = bk_t1; dbg[2] = bk_t2; dbg[3] = bk_t3; dbg[4] = bk_hw;
    }
#endif
}

template <int D, bool CAUSAL, bool STATE>
static hipError_t launch_one(const FwdArgs& a, hipStream_t stream)
{
    constexpr int lds = kFwdBufs * 2 * fwd_kv<D>() * D * 2;
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
