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

// Transposed read.  Within each group of 16 lanes, lane 4q+p supplies the address of row q,
// columns 4p..4p+3 of a 4-row x 16-column block; lane i of the group receives column i of
// the 4 rows (row q in element q).  EXEC must be all ones (never call under divergence).
__device__ __forceinline__ bf16x4 lds_read_tr(const char* base, int byte_off)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (lds_bf16x4*)(uintptr_t)(uint32_t)(uintptr_t)(base + byte_off));
}

// The same read issued from inline asm, for loops that keep an LDS-DMA (buffer_load ... lds) in
// flight: hipcc puts `s_waitcnt vmcnt(0)` in front of the tr-read BUILTIN whenever a DMA is
// pending (it cannot tell the two apart), which exposes the whole DMA latency.  The asm form is
// invisible to that logic; in exchange its completion must be waited for by hand with
// lds_tr_wait<N>() naming every destination (so no compiler copy can run ahead of the data).
// `addr` is the lane's LDS byte address, IMM a compile-time byte offset (< 65536).
template <int IMM>
__device__ __forceinline__ bf16x4 lds_read_tr_asm(uint32_t addr)
{
    bf16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(IMM));
    return r;
}
// Waits until at most N LDS operations of this wave are outstanding (they return in order).
template <int N>
__device__ __forceinline__ void lds_tr_wait(bf16x4& a, bf16x4& b, bf16x4& c, bf16x4& d)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N));
}

template <int N>
__device__ __forceinline__ void lds_tr_wait2(bf16x4& a, bf16x4& b)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N));
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
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

// Value held by the same lane index in the OTHER 32-lane half of the wave.
__device__ __forceinline__ float other_half(float x)
{
    const uint32_t u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // r[0]: lanes 32-63 now hold lanes 0-31's value; r[1]: lanes 0-31 hold lanes 32-63's.
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

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
