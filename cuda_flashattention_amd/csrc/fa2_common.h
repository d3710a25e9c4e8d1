// fa2_common.h -- device-side building blocks shared by the gfx950 FA2 kernels.
//
// Everything here is written for CDNA4 / gfx950 only: wave64, v_mfma_f32_32x32x16_bf16,
// ds_read_b64_tr_b16, v_permlane32_swap, 160 KiB LDS per CU.  There is no other target.
//
// It replaces the reference's scalar device helpers -- warp_reduce_sum / warp_reduce_max
// (src/util/cuda_helper.h:21-37) and load_Q_tile / process_kv_block
// (src/util/attention_helper.h:6-132) -- with MFMA fragments: the per-row reductions those
// helpers do with 32-lane shuffle butterflies become in-lane reductions over an MFMA
// accumulator column plus ONE cross-half permlane32_swap.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

namespace fa2 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// ---------------------------------------------------------------------------------------
// LDS image of a [rows][D] bf16 tile that serves BOTH kinds of MFMA operand read:
//   * row reads   (ds_read_b128, 8 consecutive elements of one row), and
//   * transposed reads (ds_read_b64_tr_b16, 4 rows x 16 columns delivered column-major).
// `ch` is the 16-byte chunk index inside the row.  The XOR spreads a 16-lane b128 group
// (16 different rows, same chunk) over all 16 slots of the 256-B bank row, and a 32-lane
// tr-read half (4 rows x 4 chunks) over all 64 banks: both conflict-free under the gfx950
// bank rules (tools/lds_bank_sim.py checks exactly this function).
// ---------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ int lds_off(int row, int ch)
{
    static_assert(D == 64 || D == 128, "head_dim must be 64 or 128");
    if constexpr (D == 128)
        return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
    else
        return 128 * row + 16 * (ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)));
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// Row read: 8 consecutive bf16 of one row = one 32x32x16 MFMA A/B fragment for the lane.
__device__ __forceinline__ bf16x8 lds_read_frag(const char* base, int byte_off)
{
    return *reinterpret_cast<const bf16x8*>(base + byte_off);
}

template <int N>
__device__ __forceinline__ void lds_tr_wait2(bf16x4& a, bf16x4& b)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N));
}

// Accumulate into a tile that must live in the accumulator half of the register file.  In the
// one-wave-per-SIMD kernels the long-lived output accumulators (256 registers) are pinned to
// AGPRs this way, which leaves the 256 architectural VGPRs to the operands and to the
// short-lived S / dP tiles that the VALU has to read (hipcc otherwise parks operands in AGPRs
// and copies them back every iteration).  The asm is opaque to hipcc's hazard recogniser:
// `s_nop 1` covers a VALU-written operand, and whoever reads the tile with non-MFMA code must
// first execute mfma_acc_settle().
__device__ __forceinline__ void mfma32_acc(f32x16& c, bf16x8 a, bf16x8 b)
{
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_acc_settle()
{
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}

// ---------------------------------------------------------------------------------------
// Accumulator tiles owned by the kernel, not by hipcc: literal AGPR ranges a[LO : LO+15] that only
// these helpers touch (the clobber list reserves the whole accumulator file a0..a255 for them, so
// hipcc allocates nothing there).  Taking a long-lived MFMA accumulator out of C++ dataflow this way means no phi, no
// copy and no spill can ever involve it -- in particular a conditional rescale of the tile costs
// nothing when not taken.  The asm is opaque to hipcc's hazard recogniser: acc_mfma carries
// `s_nop 1` for a VALU-written operand, and any non-MFMA access to a tile an MFMA may still be
// writing must be preceded by mfma_acc_settle().
// ---------------------------------------------------------------------------------------
#define FA2_ACC_CLOBBERS \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
    "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
    "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
    "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
    "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
    "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
    "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
    "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
    "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
    "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
    "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Product whose B operand is a fragment resident in literal AGPRs a[BLO : BLO+3] (an operand that
// never changes during the kernel, e.g. the Q fragments of a query block) and whose accumulator is
// an ordinary VGPR tile the VALU reads afterwards: c += a * a[BLO:BLO+3].  Whoever reads c with
// non-MFMA code must first execute mfma_vgpr_settle(c).
template <int BLO>
__device__ __forceinline__ void mfma_bagpr(f32x16& c, bf16x8 a)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0"
                 : "+v"(c) : "v"(a), "i"(BLO), "i"(BLO + 3) : FA2_ACC_CLOBBERS);
}

// Two all-VGPR products sharing their A operand in one statement: c0 += a * b0, c1 += a * b1.
__device__ __forceinline__ void mfma2_vv(f32x16& c0, f32x16& c1, bf16x8 a, bf16x8 b0, bf16x8 b1)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1"
                 : "+v"(c0), "+v"(c1) : "v"(a), "v"(b0), "v"(b1));
}
// The same, threading two finished tiles through the statement (see mfma4_bagpr).
__device__ __forceinline__ void mfma2_vv(f32x16& c0, f32x16& c1, bf16x8 a, bf16x8 b0, bf16x8 b1, f32x16& dep0, f32x16& dep1)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %6, %1"
                 : "+v"(c0), "+v"(c1), "+v"(dep0), "+v"(dep1) : "v"(a), "v"(b0), "v"(b1));
}

// One k-step of two products that share their streamed A operands, in ONE asm statement (every
// asm boundary costs a pad s_nop from hipcc, and the loops that use these are issue-bound):
//   s0 += ka * a[Q0..]   s1 += ka * a[Q1..]   d0 += va * a[G0..]   d1 += va * a[G1..]
// The four `dep` tiles are not touched: naming them as read-write operands threads them through
// this statement, so VALU code that consumes them cannot be scheduled above it and VALU code that
// produces them cannot sink below the next such statement -- that is how arithmetic on a finished
// tile is placed between the MFMA groups of the next one at no instruction cost.
template <int Q0, int Q1, int G0, int G1>
__device__ __forceinline__ void mfma4_bagpr(f32x16& s0, f32x16& s1, f32x16& d0, f32x16& d1, bf16x8 ka, bf16x8 va,
                                            f32x16& dep0, f32x16& dep1, f32x16& dep2, f32x16& dep3)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %8, a[%c10:%c11], %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %1, %8, a[%c12:%c13], %1\n\t"
                 "v_mfma_f32_32x32x16_bf16 %2, %9, a[%c14:%c15], %2\n\t"
                 "v_mfma_f32_32x32x16_bf16 %3, %9, a[%c16:%c17], %3"
                 : "+v"(s0), "+v"(s1), "+v"(d0), "+v"(d1), "+v"(dep0), "+v"(dep1), "+v"(dep2), "+v"(dep3)
                 : "v"(ka), "v"(va), "i"(Q0), "i"(Q0 + 3), "i"(Q1), "i"(Q1 + 3), "i"(G0), "i"(G0 + 3), "i"(G1), "i"(G1 + 3)
                 : FA2_ACC_CLOBBERS);
}
template <int Q0, int Q1, int G0, int G1>
__device__ __forceinline__ void mfma4_bagpr(f32x16& s0, f32x16& s1, f32x16& d0, f32x16& d1, bf16x8 ka, bf16x8 va)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, a[%c6:%c7], %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %1, %4, a[%c8:%c9], %1\n\t"
                 "v_mfma_f32_32x32x16_bf16 %2, %5, a[%c10:%c11], %2\n\t"
                 "v_mfma_f32_32x32x16_bf16 %3, %5, a[%c12:%c13], %3"
                 : "+v"(s0), "+v"(s1), "+v"(d0), "+v"(d1)
                 : "v"(ka), "v"(va), "i"(Q0), "i"(Q0 + 3), "i"(Q1), "i"(Q1 + 3), "i"(G0), "i"(G0 + 3), "i"(G1), "i"(G1 + 3)
                 : FA2_ACC_CLOBBERS);
}
// The same for the first k-step of the chains: s = ka * q (C = 0), d = va * g + c (C = a constant
// tile, e.g. minus the row constant that would otherwise be subtracted element by element).
template <int Q0, int Q1, int G0, int G1>
__device__ __forceinline__ void mfma4_bagpr_init(f32x16& s0, f32x16& s1, f32x16& d0, f32x16& d1, bf16x8 ka, bf16x8 va,
                                                 const f32x16& c0, const f32x16& c1)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, a[%c8:%c9], 0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %1, %4, a[%c10:%c11], 0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %2, %5, a[%c12:%c13], %6\n\t"
                 "v_mfma_f32_32x32x16_bf16 %3, %5, a[%c14:%c15], %7"
                 : "=&v"(s0), "=&v"(s1), "=&v"(d0), "=&v"(d1)
                 : "v"(ka), "v"(va), "v"(c0), "v"(c1),
                   "i"(Q0), "i"(Q0 + 3), "i"(Q1), "i"(Q1 + 3), "i"(G0), "i"(G0 + 3), "i"(G1), "i"(G1 + 3)
                 : FA2_ACC_CLOBBERS);
}
// acc_mfma2 threading TWO dependent tiles.
template <int LO0, int LO1>
__device__ __forceinline__ void acc_mfma2(bf16x8 a, bf16x8 b0, bf16x8 b1, f32x16& dep0, f32x16& dep1)
{
    asm volatile("s_nop 0\n\tv_mfma_f32_32x32x16_bf16 a[%c5:%c6], %2, %3, a[%c5:%c6]\n\t"
                 "v_mfma_f32_32x32x16_bf16 a[%c7:%c8], %2, %4, a[%c7:%c8]"
                 : "+v"(dep0), "+v"(dep1)
                 : "v"(a), "v"(b0), "v"(b1), "i"(LO0), "i"(LO0 + 15), "i"(LO1), "i"(LO1 + 15) : FA2_ACC_CLOBBERS);
}
// acc_mfma2 threading two tiles and two scalars (running sums that must not be deferred).
template <int LO0, int LO1>
__device__ __forceinline__ void acc_mfma2(bf16x8 a, bf16x8 b0, bf16x8 b1, f32x16& dep0, f32x16& dep1, float& f0, float& f1)
{
    asm volatile("s_nop 0\n\tv_mfma_f32_32x32x16_bf16 a[%c7:%c8], %4, %5, a[%c7:%c8]\n\t"
                 "v_mfma_f32_32x32x16_bf16 a[%c9:%c10], %4, %6, a[%c9:%c10]"
                 : "+v"(dep0), "+v"(dep1), "+v"(f0), "+v"(f1)
                 : "v"(a), "v"(b0), "v"(b1), "i"(LO0), "i"(LO0 + 15), "i"(LO1), "i"(LO1 + 15) : FA2_ACC_CLOBBERS);
}
// Threads four tiles through the asm order (see mfma4_bagpr) without doing anything.
__device__ __forceinline__ void thread4(f32x16& a, f32x16& b, f32x16& c, f32x16& d)
{
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// Two accumulations into asm-owned tiles that share their A operand: a[LO0..] += a * b0, a[LO1..] += a * b1.
template <int LO0, int LO1>
__device__ __forceinline__ void acc_mfma2(bf16x8 a, bf16x8 b0, bf16x8 b1)
{
    asm volatile("s_nop 0\n\tv_mfma_f32_32x32x16_bf16 a[%c3:%c4], %0, %1, a[%c3:%c4]\n\t"
                 "v_mfma_f32_32x32x16_bf16 a[%c5:%c6], %0, %2, a[%c5:%c6]"
                 : : "v"(a), "v"(b0), "v"(b1), "i"(LO0), "i"(LO0 + 15), "i"(LO1), "i"(LO1 + 15) : FA2_ACC_CLOBBERS);
}
// The same, threading four dependent tiles through the statement (see mfma4_bagpr).
template <int LO0, int LO1>
__device__ __forceinline__ void acc_mfma2(bf16x8 a, bf16x8 b0, bf16x8 b1, f32x16& dep0, f32x16& dep1, f32x16& dep2,
                                          f32x16& dep3)
{
    asm volatile("s_nop 0\n\tv_mfma_f32_32x32x16_bf16 a[%c7:%c8], %4, %5, a[%c7:%c8]\n\t"
                 "v_mfma_f32_32x32x16_bf16 a[%c9:%c10], %4, %6, a[%c9:%c10]"
                 : "+v"(dep0), "+v"(dep1), "+v"(dep2), "+v"(dep3)
                 : "v"(a), "v"(b0), "v"(b1), "i"(LO0), "i"(LO0 + 15), "i"(LO1), "i"(LO1 + 15) : FA2_ACC_CLOBBERS);
}

// Two accumulations into hipcc-allocated AGPR tiles that share their A operand, fused with the two
// transposed reads of the NEXT fragment (n0, n1) and the wait that leaves exactly those in flight,
// in ONE statement and without a pad: a, b0, b1 must not have been written by the VALU within the
// two instructions before it (PAD = true adds the two wait states when that cannot be guaranteed).
template <int IMM, bool PAD = false>
__device__ __forceinline__ void tr_mfma2_acc_next(bf16x4& n0, bf16x4& n1, uint32_t addr0, uint32_t addr1, f32x16& c0,
                                                  f32x16& c1, bf16x8 a, bf16x8 b0, bf16x8 b1)
{
    if constexpr (PAD)
        asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%c6\n\tds_read_b64_tr_b16 %1, %5 offset:%c6\n\t"
                     "s_waitcnt lgkmcnt(2)\n\ts_nop 1\n\t"
                     "v_mfma_f32_32x32x16_bf16 %2, %7, %8, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %7, %9, %3"
                     : "=&v"(n0), "=&v"(n1), "+a"(c0), "+a"(c1)
                     : "v"(addr0), "v"(addr1), "i"(IMM), "v"(a), "v"(b0), "v"(b1));
    else
        asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%c6\n\tds_read_b64_tr_b16 %1, %5 offset:%c6\n\t"
                     "s_waitcnt lgkmcnt(2)\n\t"
                     "v_mfma_f32_32x32x16_bf16 %2, %7, %8, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %7, %9, %3"
                     : "=&v"(n0), "=&v"(n1), "+a"(c0), "+a"(c1)
                     : "v"(addr0), "v"(addr1), "i"(IMM), "v"(a), "v"(b0), "v"(b1));
}
// The last pair of a run: nothing further to read, everything outstanding must have landed.
__device__ __forceinline__ void tr_mfma2_acc_last(f32x16& c0, f32x16& c1, bf16x8 a, bf16x8 b0, bf16x8 b1)
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1"
                 : "+a"(c0), "+a"(c1) : "v"(a), "v"(b0), "v"(b1));
}
// First k-step of two chains that start from the SAME constant tile: c0 holds the constants on entry,
// c1 is written from them (C operand = c0) before c0 is accumulated in place -- no copy of the tile.
__device__ __forceinline__ void mfma2_vv_cinit(f32x16& c0, f32x16& c1, bf16x8 a, bf16x8 b0, bf16x8 b1)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %1, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %0"
                 : "+v"(c0), "=&v"(c1) : "v"(a), "v"(b0), "v"(b1));
}

// Both halves of one transposed fragment in one statement.
template <int IMM>
__device__ __forceinline__ void lds_read_tr2_asm(bf16x4& r0, bf16x4& r1, uint32_t addr0, uint32_t addr1)
{
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"
                 : "=&v"(r0), "=&v"(r1) : "v"(addr0), "v"(addr1), "i"(IMM));
}

__device__ __forceinline__ void mfma_vgpr_settle(f32x16& c)
{
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c));
}
// Parks a bf16x8 fragment in a[LO : LO+3].
template <int LO>
__device__ __forceinline__ void acc_write_frag(bf16x8 f)
{
    const u32x4 w = __builtin_bit_cast(u32x4, f);
    asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\t"
                 "v_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
                 : : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "i"(LO), "i"(LO + 1), "i"(LO + 2), "i"(LO + 3)
                 : FA2_ACC_CLOBBERS);
}

template <int R>
__device__ __forceinline__ float acc_read()
{
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(R));
    return x;
}
template <int R>
__device__ __forceinline__ void acc_write(float x)
{
    asm volatile("v_accvgpr_write_b32 a[%c1], %0" : : "v"(x), "i"(R) : FA2_ACC_CLOBBERS);
}

// Keeps an MFMA operand's registers allocated up to this point.  hipcc sees an asm-issued MFMA as
// an instruction that has read its operands once issued, and may hand a dead operand register to
// the very next VALU instruction as a temporary -- while the matrix pipe is still reading it
// (observed: wrong products at D = 64).  Place after the VALU work that follows the MFMA.
__device__ __forceinline__ void keep_alive(const bf16x8& x) { asm volatile("" : : "v"(x)); }

__device__ __forceinline__ float half_max(float x)
{
    const uint32_t u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float half_sum(float x)
{
    const uint32_t u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Pack 8 consecutive accumulator registers (rows 8s..8s+7 of a 32x32 result, in the
// accumulator's own row order) into the bf16 fragment of k-step s of a following MFMA that
// contracts over the accumulator's ROW index.  Element j of lane-half h is accumulator row
// 16s + 8(j>>2) + 4h + (j&3); the other operand must be gathered in the same k order.
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& x, int s)
{
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = (__bf16)x[8 * s + j];
    return out;
}

// Accumulator row (0..31) held in register r of lane-half h of a 32x32 MFMA result.
__device__ __forceinline__ constexpr int acc_row(int r, int h)
{
    return (r & 3) + 8 * (r >> 2) + 4 * h;
}

// XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs (bid % 8 labels
// the group sharing an L2).  All row-blocks of one head re-read that head's K and V, so the
// j-th workgroup of XCD-group x is given head (j / nrb) * 8 + x: the 32 CUs of one XCD work
// through the row-blocks of ONE head at a time and its K/V stay in that XCD's 4 MiB L2.
// Falls back to the plain order when the head count is not a multiple of 8 (still bijective).
__device__ __forceinline__ void map_block(int bid, int n_heads, int nrb, int& head, int& rb)
{
    if ((n_heads & 7) == 0) {
        const int x = bid & 7, j = bid >> 3;
        head = (j / nrb) * 8 + x;
        rb = j % nrb;
    } else {
        head = bid / nrb;
        rb = bid % nrb;
    }
}

}  // namespace fa2
