// fa2_bwd_fused.hip -- the single-kernel, five-product FlashAttention-2 backward for gfx950: what fa2_backward runs for
// bf16, causal or not, at d = 128 (every seq_len from 897 up and every multiple of 256; padded to the key block inside: RAGGED)
// and, since round 4, at d = 64 (HD = 64: every seq_len whose padding costs under 7 %), and what fa2_backward_block runs for the
// ring backward's dense square and unmasked rectangular blocks (RECT).  Everything else -- short ragged lengths, masked
// rectangular blocks -- runs the dQ and dK/dV kernels of fa2_bwd_bf16.hip.
// Replaces the reference's flash_attention_2_backward_kernel (02_flash_attention_v2_backward/
// flash_attention_backward_kernel.cu:47-246), one kernel for any N and d <= 128 there (dispatch :264-297).  Design and
// measurements: DESIGN.md section 3.  The text below describes d = 128; the d = 64 differences are at the kernel template.
//
// Work split as fa2_bwd_dkdv_kernel (a workgroup owns 256 keys, dK^T / dV^T in 256 accumulator registers per wave), plus the
// query gradient: the packed dS pairs each wave already forms for dK go to a [key][q] tile in LDS, and every wave
// contracts that tile over ALL 256 keys of the workgroup with K^T (transposed reads of the K image) for its own 32 of the
// 128 columns -- dQ[q][col] -- so S and dP are formed once (80 MFMAs per 32-row sub-tile and wave instead of 64 + 48).
// The main loop is a generated body (tools/gen_fused_body.py).  What remains is the sum of the dQ tiles over the N / 256
// workgroups of a head.  The shipped form (CHAIN) passes a running sum from key block to key block, in a fixed order,
// through the L2 of the XCD the head is worked on: deterministic.  The other form (fa2_backward_fused, mode 0) adds the
// tiles with fp32 atomics (memory-side on MI355X, ~1.3 TB/s chip-wide: B H (N/256) N d 4 bytes = 6.7 ms at the bench
// shape); it is kept as the measured alternative.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "fa2_common.h"
#include "fa2_launch.h"

namespace fa2 {

#include "fa2_bwd_fused_body.inc"

typedef __attribute__((address_space(3))) void* fused_lptr_t;

// dS tile [256 keys][32 q] bf16, 64-byte rows of eight 8-byte chunks: chunk c of row r sits at chunk c ^ key(r), key(r) = bits
// 1..3 of r.  A ds_write_b64 is served in four groups of 16 consecutive lanes (= 16 consecutive rows here) against 32 banks
// of 4 bytes: rows r and r + 2 lie 128 bytes apart, so the eight even (and the eight odd) rows of a group need eight
// different chunks -- bits 1..3.  The transposed reads (two groups of 32 lanes: four consecutive rows x all eight chunks each)
// are conflict-free under any per-row permutation of the chunks.  The E chain steps 16 rows per immediate offset and the
// packs 32, so the key cannot use row bits >= 4, and does not need to.  (Rounds 2-3 used bits 2..3 only: every store two-way
// conflicted, SQ_LDS_BANK_CONFLICT = 32 cycles per body.)  tools/lds_bank_sim.py checks both access patterns.
#ifndef FA2_FUSED_DSKEY
#define FA2_FUSED_DSKEY(row) (((row) >> 1) & 7)
#endif

#ifndef FA2_FUSED_DIAG        // diagnostic builds (wrong results, timing only): 1 = no waiting, 2 = no running-sum loads, 4 = no stores
#define FA2_FUSED_DIAG 0
#endif

template <int R>
__device__ __forceinline__ void fused_vset(uint32_t x)
{
    asm volatile("v_mov_b32 v%c1, %0" : : "v"(x), "i"(R) : "v255");
}

#define FA2_FUSED_CLOBBERS "memory", "vcc", "v255", FA2_ACC_CLOBBERS
#define FA2_FUSED_OPS [r0] "v"(roff[0]), [r1] "v"(roff[1]), [r2] "v"(roff[2]), [r3] "v"(roff[3]), [r4] "v"(roff[4]), [r5] "v"(roff[5]),  \
    [r6] "v"(roff[6]), [r7] "v"(roff[7]), [t0] "v"(toff[0]), [t1] "v"(toff[1]), [t2] "v"(toff[2]), [t3] "v"(toff[3]), [t4] "v"(toff[4]),   \
    [t5] "v"(toff[5]), [t6] "v"(toff[6]), [t7] "v"(toff[7]), [rc] "v"(rcv), [c2] "s"(c2)

// chained kernel: everything a step does is inside the body (tools/gen_fused_body.py lists the operands)
struct FusedStep {
    __amdgpu_buffer_rsrc_t drs, lrs, qrs, grs, rcrs, ctl;
    uint32_t dso, lso, qso, rcso, pvo, mso;
    int need, pval;
};
// head_dim 64 (round 4): the same bodies generated for KS = 4, DT = 2 (FA2_FUSED64_*): four row-read and four transposed-read
// addresses instead of eight
#define FA2_FUSED_OPS64 [r0] "v"(roff[0]), [r1] "v"(roff[1]), [r2] "v"(roff[2]), [r3] "v"(roff[3]), [t0] "v"(toff[0]), [t1] "v"(toff[1]),    \
    [t2] "v"(toff[2]), [t3] "v"(toff[3]), [rc] "v"(rcv), [c2] "s"(c2)
template <int BUF, int PAR, int VMW, bool MASKED>
__device__ __forceinline__ void fused_cbody(const uint32_t (&roff)[4], const uint32_t (&toff)[4], uint32_t rcv, float c2, uint32_t dqv,
                                            uint32_t dvo, uint32_t rcvo, uint32_t mw, uint32_t mw2, int wv, const FusedStep& f, int& err,
                                            int lo0 = 0, int lo1 = 0)
{
#define FA2_FUSED_CASE(B, P)                                                                                                          \
    if constexpr (BUF == B && PAR == P && MASKED)                                                                                     \
        asm volatile(FA2_FUSED64_MBODY_B##B##_P##P                                                                                    \
                     : [err] "+s"(err)                                                                                                \
                     : FA2_FUSED_OPS64, [vm] "i"(VMW), [dqv] "v"(dqv), [drs] "s"(f.drs), [dso] "s"(f.dso), [lrs] "s"(f.lrs), [lso] "s"(f.lso), \
                       [mw] "s"(mw), [mw2] "s"(mw2), [qrs] "s"(f.qrs), [grs] "s"(f.grs), [rcrs] "s"(f.rcrs), [qso] "s"(f.qso),           \
                       [rcso] "s"(f.rcso), [dvo] "v"(dvo), [rcvo] "v"(rcvo), [wv] "s"(wv), [ctl] "s"(f.ctl), [pvo] "s"(f.pvo),           \
                       [mso] "s"(f.mso), [need] "s"(f.need), [pval] "s"(f.pval), [lo0] "v"(lo0), [lo1] "v"(lo1)                        \
                     : FA2_FUSED_CLOBBERS, "s12", "s13", "scc", "exec", "m0", "v39");                                                 \
    if constexpr (BUF == B && PAR == P && !MASKED)                                                                                    \
        asm volatile(FA2_FUSED64_CBODY_B##B##_P##P                                                                                    \
                     : [err] "+s"(err)                                                                                                \
                     : FA2_FUSED_OPS64, [vm] "i"(VMW), [dqv] "v"(dqv), [drs] "s"(f.drs), [dso] "s"(f.dso), [lrs] "s"(f.lrs), [lso] "s"(f.lso), \
                       [mw] "s"(mw), [mw2] "s"(mw2), [qrs] "s"(f.qrs), [grs] "s"(f.grs), [rcrs] "s"(f.rcrs), [qso] "s"(f.qso),           \
                       [rcso] "s"(f.rcso), [dvo] "v"(dvo), [rcvo] "v"(rcvo), [wv] "s"(wv), [ctl] "s"(f.ctl), [pvo] "s"(f.pvo),           \
                       [mso] "s"(f.mso), [need] "s"(f.need), [pval] "s"(f.pval)                                                        \
                     : FA2_FUSED_CLOBBERS, "s12", "s13", "scc", "exec", "m0", "v39");
    FA2_FUSED_CASE(0, 0) FA2_FUSED_CASE(0, 1) FA2_FUSED_CASE(1, 0) FA2_FUSED_CASE(1, 1) FA2_FUSED_CASE(2, 0) FA2_FUSED_CASE(2, 1)
#undef FA2_FUSED_CASE
}

template <int BUF, int PAR, int VMW, bool MASKED>
__device__ __forceinline__ void fused_cbody(const uint32_t (&roff)[8], const uint32_t (&toff)[8], uint32_t rcv, float c2, uint32_t dqv,
                                            uint32_t dvo, uint32_t rcvo, uint32_t mw, uint32_t mw2, int wv, const FusedStep& f, int& err,
                                            int lo0 = 0, int lo1 = 0)
{
#define FA2_FUSED_CASE(B, P)                                                                                                          \
    if constexpr (BUF == B && PAR == P && MASKED)                                                                                     \
        asm volatile(FA2_FUSED_MBODY_B##B##_P##P                                                                                      \
                     : [err] "+s"(err)                                                                                                \
                     : FA2_FUSED_OPS, [vm] "i"(VMW), [dqv] "v"(dqv), [drs] "s"(f.drs), [dso] "s"(f.dso), [lrs] "s"(f.lrs), [lso] "s"(f.lso), \
                       [mw] "s"(mw), [mw2] "s"(mw2), [qrs] "s"(f.qrs), [grs] "s"(f.grs), [rcrs] "s"(f.rcrs), [qso] "s"(f.qso),           \
                       [rcso] "s"(f.rcso), [dvo] "v"(dvo), [rcvo] "v"(rcvo), [wv] "s"(wv), [ctl] "s"(f.ctl), [pvo] "s"(f.pvo),           \
                       [mso] "s"(f.mso), [need] "s"(f.need), [pval] "s"(f.pval), [lo0] "v"(lo0), [lo1] "v"(lo1)                        \
                     : FA2_FUSED_CLOBBERS, "s12", "s13", "scc", "exec", "m0", "v39");                                                 \
    if constexpr (BUF == B && PAR == P && !MASKED)                                                                                    \
        asm volatile(FA2_FUSED_CBODY_B##B##_P##P                                                                                      \
                     : [err] "+s"(err)                                                                                                \
                     : FA2_FUSED_OPS, [vm] "i"(VMW), [dqv] "v"(dqv), [drs] "s"(f.drs), [dso] "s"(f.dso), [lrs] "s"(f.lrs), [lso] "s"(f.lso), \
                       [mw] "s"(mw), [mw2] "s"(mw2), [qrs] "s"(f.qrs), [grs] "s"(f.grs), [rcrs] "s"(f.rcrs), [qso] "s"(f.qso),           \
                       [rcso] "s"(f.rcso), [dvo] "v"(dvo), [rcvo] "v"(rcvo), [wv] "s"(wv), [ctl] "s"(f.ctl), [pvo] "s"(f.pvo),           \
                       [mso] "s"(f.mso), [need] "s"(f.need), [pval] "s"(f.pval)                                                        \
                     : FA2_FUSED_CLOBBERS, "s12", "s13", "scc", "exec", "m0", "v39");
    FA2_FUSED_CASE(0, 0) FA2_FUSED_CASE(0, 1) FA2_FUSED_CASE(1, 0) FA2_FUSED_CASE(1, 1) FA2_FUSED_CASE(2, 0) FA2_FUSED_CASE(2, 1)
#undef FA2_FUSED_CASE
}

template <int BUF, int PAR, int VMW>
__device__ __forceinline__ void fused_body(const uint32_t (&roff)[8], const uint32_t (&toff)[8], uint32_t rcv, float c2)
{
#define FA2_FUSED_CASE(B, P) \
    if constexpr (BUF == B && PAR == P) asm volatile(FA2_FUSED_BODY_B##B##_P##P : : FA2_FUSED_OPS, [vm] "i"(VMW) : FA2_FUSED_CLOBBERS);
    FA2_FUSED_CASE(0, 0) FA2_FUSED_CASE(0, 1) FA2_FUSED_CASE(1, 0) FA2_FUSED_CASE(1, 1) FA2_FUSED_CASE(2, 0) FA2_FUSED_CASE(2, 1)
#undef FA2_FUSED_CASE
}

// this wave's dQ tile (16 registers: lane = column 32 w + (lane & 31), register r = row (r & 3) + 8 (r >> 2) + 4 h) -> dQacc
// rows [row0, row0 + 32) of the head: rsrc covers the head's fp32 slab, voff = ((4 h) * 128 + col) * 4, soff = row0 * 512.
template <int DQT>
__device__ __forceinline__ void fused_dq_atomic(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff)
{
    asm volatile(
        "s_add_u32 s12, %2, 4096\n\ts_add_u32 s13, %2, 8192\n\ts_add_u32 s14, %2, 12288\n\t"
        "buffer_atomic_add_f32 v%c3, %0, %1, %2 offen\n\tbuffer_atomic_add_f32 v%c4, %0, %1, %2 offen offset:512\n\t"
        "buffer_atomic_add_f32 v%c5, %0, %1, %2 offen offset:1024\n\tbuffer_atomic_add_f32 v%c6, %0, %1, %2 offen offset:1536\n\t"
        "buffer_atomic_add_f32 v%c7, %0, %1, s12 offen\n\tbuffer_atomic_add_f32 v%c8, %0, %1, s12 offen offset:512\n\t"
        "buffer_atomic_add_f32 v%c9, %0, %1, s12 offen offset:1024\n\tbuffer_atomic_add_f32 v%c10, %0, %1, s12 offen offset:1536\n\t"
        "buffer_atomic_add_f32 v%c11, %0, %1, s13 offen\n\tbuffer_atomic_add_f32 v%c12, %0, %1, s13 offen offset:512\n\t"
        "buffer_atomic_add_f32 v%c13, %0, %1, s13 offen offset:1024\n\tbuffer_atomic_add_f32 v%c14, %0, %1, s13 offen offset:1536\n\t"
        "buffer_atomic_add_f32 v%c15, %0, %1, s14 offen\n\tbuffer_atomic_add_f32 v%c16, %0, %1, s14 offen offset:512\n\t"
        "buffer_atomic_add_f32 v%c17, %0, %1, s14 offen offset:1024\n\tbuffer_atomic_add_f32 v%c18, %0, %1, s14 offen offset:1536\n\t"
        "s_nop 1"
        : : "v"(voff), "s"(rs), "s"(soff), "i"(DQT), "i"(DQT + 1), "i"(DQT + 2), "i"(DQT + 3), "i"(DQT + 4), "i"(DQT + 5), "i"(DQT + 6),
            "i"(DQT + 7), "i"(DQT + 8), "i"(DQT + 9), "i"(DQT + 10), "i"(DQT + 11), "i"(DQT + 12), "i"(DQT + 13), "i"(DQT + 14),
            "i"(DQT + 15)
        : "memory", "s12", "s13", "s14", "v255");
}

// 16 bytes of both key blocks' V rows (k-step S) into their fragment registers; completes at the caller's vmcnt(0)
template <int VF, int KS, int S>
__device__ __forceinline__ void fused_load_vfrag(const char* v0, const char* v1)
{
#ifndef FA2_FUSED_KV_NT       // experiment: once-read K / V rows (and the dK / dV stores) with the streaming cache policy
#define FA2_FUSED_KV_NT ""
#define FA2_FUSED_KV_AUX 0
#endif
    asm volatile("global_load_dwordx4 v[%c2:%c3], %0, off offset:%c4" FA2_FUSED_KV_NT "\n\tglobal_load_dwordx4 v[%c5:%c6], %1, off offset:%c4" FA2_FUSED_KV_NT
                 : : "v"(v0), "v"(v1), "i"(VF + 4 * S), "i"(VF + 4 * S + 3), "i"(32 * S), "i"(VF + 4 * (KS + S)), "i"(VF + 4 * (KS + S) + 3)
                 : "memory", "v255");
}

template <int T>
__device__ __forceinline__ void fused_acc_zero(u32x4 z)
{
    // (the compiler does not know this statement is an MFMA: the wait states between its writes of z and the read are ours)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, %0, 0" : : "v"(z), "i"(16 * T), "i"(16 * T + 15) : FA2_ACC_CLOBBERS);
}

template <int DQT>
__device__ __forceinline__ void fused_dq_zero()
{
    static_for<16>([&](auto R) { fused_vset<DQT + decltype(R)::value>(0u); });
}

// ---- the chained form's control block (ints, zeroed by the launcher before every launch)
//   ticket[x]  at 32 x            per-XCD unit queue (x = hardware XCC_ID, 16 possible values)
//   next_head  at 32 * 16         heads handed out so far, over all XCDs
//   error      at 32 * 17         set when a bounded spin ran out (the result is then poisoned by the output pass)
//   headmap[x][k] at 32 * 18 + x (BH + kHmPad) + k   1 + head the k-th chain of queue x works on; -1 = nothing left; 0 = not yet known
//   prog[head][j] behind it       query sub-tiles whose running dQ sum key block j has completely stored
constexpr int kCtlTicket = 0, kCtlNextHead = 32 * 16, kCtlError = 32 * 17, kCtlHeadmap = 32 * 18;
constexpr int kHmPad = 16;           // chains a queue may name past the last head (the causal form takes them in groups of up to 8)
constexpr int kSpinLimit = FA2_FUSED_SPIN_LIMIT;      // one bound for every wait: the bodies' (generated) and the unit queue's

struct FusedArgs {
    BwdArgs b;
    float* dQacc;       // fp32 [BH][N][128]: zeroed by the launcher and atomically added to (CHAIN = false);
                        // the running sums as [BH][N / 32][wave][g][lane] x 4 floats (CHAIN = true: fused_dq_out_chain_kernel)
    int* ctl;           // control block (CHAIN = true)
    int npad;           // seq_len rounded up to a multiple of 256 (== seq_len unless the launch is ragged: CHAIN only)
    const float* rc;    // the row-constant planes the kernel reads: b.RC ([2][BH][seq_len]), or for a ragged launch their padded
                        // copy [2][BH][npad] whose rows >= seq_len hold (-1e30, 0): P = 0 and dS = 0 there
    int fault;          // FA2_TEST_HOOKS builds only (tests/loopback/libfa2_mi355x_hooks.so): key block 1 of every head never
                        // publishes its progress.  The product build has neither the switch nor the code it guards.
};

// Test hooks.  The product library has none: no environment variable or call can turn dQ into NaNs or shrink the grid.
// tests/ link a second build of THIS file with -DFA2_TEST_HOOKS (csrc/Makefile, target `hooks`) that exports a setter.
#ifdef FA2_TEST_HOOKS
static int g_hook_fault = 0, g_hook_grid = 0, g_hook_last_grid = 0;
extern "C" void fa2_test_set_fused_hooks(int fault, int grid) { g_hook_fault = fault; g_hook_grid = grid; }
extern "C" int fa2_test_last_fused_grid(void) { return g_hook_last_grid; }      // workgroups of the last chained launch
#define FA2_HOOK_NOTE_GRID(n) (g_hook_last_grid = (n))
#else
constexpr int g_hook_fault = 0, g_hook_grid = 0;
#define FA2_HOOK_NOTE_GRID(n) ((void)0)
#endif

__device__ __forceinline__ int fused_ctl_ints(int BH, int ncb) { return kCtlHeadmap + 16 * (BH + kHmPad) + 9 * BH * ncb; }

__device__ __forceinline__ int fused_load_sc1(const int* p)
{
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// The progress of the previous key block is prefetched into v39, a register the compiler does not allocate (the kernel is
// limited to v0..v38) and the generated body owns: issued at the top of a body, read behind its barrier.
__device__ __forceinline__ void fused_seen_set(int v)
{
    asm volatile("v_mov_b32 v39, %0" : : "s"(v) : "v39");
}

// Inside the unit loop nothing per-lane is kept by the compiler (it has v0..v38): the lane index is recomputed where needed,
// and the mailbox is read through an address that lives in an SGPR.
__device__ __forceinline__ int fused_lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
template <int OFF>
__device__ __forceinline__ int fused_mail_read(uint32_t lds_addr)
{
    int v;
    asm volatile("v_mov_b32 %0, %1\n\tds_read_b32 %0, %0 offset:%c2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "s"(lds_addr), "i"(OFF) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}

// Flags and running sums are written with plain stores: they land in the L2 of this XCD, which is where the readers (same
// XCD by construction, sc1 loads that bypass their CU's vector cache) look.  An sc1 store would be written through to
// memory first -- measured: 9.2 ms instead of 5.6 ms for the whole backward, the consumers' polls then wait on HBM.
__device__ __forceinline__ void fused_store_flag(int* p, int v)
{
    asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

// Every wait in the chained kernel is bounded: a spin that runs out raises the error word, and a raised error word ends
// everybody else's spins within 1024 polls (the output pass then poisons dQ) -- a fault shows as NaNs, not as a hung GPU.
__device__ __forceinline__ bool fused_spin_over(int* ctl, int spins)
{
    if (spins > kSpinLimit) {
        atomicExch(ctl + kCtlError, 1);
        return true;
    }
    return (spins & 1023) == 0 && fused_load_sc1(ctl + kCtlError) != 0;
}

// One launch of the five-product backward.
//   CHAIN = false: one workgroup per (head, key block); dQ tiles are added to dQacc with fp32 atomics.
//   CHAIN = true:  a persistent grid (one workgroup per CU).  Workgroups take (head, key block j) units, in order, from the
//     queue of the XCD they run on (hardware XCC_ID), so that the key blocks of a head are worked on by CUs that share an L2.
//     Key block j takes the running sum of a query sub-tile from key block j - 1 (loaded straight into the accumulator the
//     E chain then adds to), stores its own, and publishes how far it is in prog[head][j]: the order of the additions is
//     fixed, so the result is deterministic.  A unit only ever waits for units taken from the same queue before it, and
//     the workgroup holding the oldest unfinished unit never waits: no deadlock for any number of resident workgroups.
//   CAUSAL (chained form only, square, no shift): key block cb sees the query sub-tiles t >= 8 cb, the first eight of them
//     through a mask.  Everything runs on the REVERSED sub-tile index: every key block starts at the last sub-tile and walks
//     down to its diagonal, so all of a head's workgroups start at once and key block cb follows key block cb - 1 a step or
//     two behind, exactly as in the non-causal form: the sum of a sub-tile starts at key block 0 and is handed UP to the key
//     block on whose diagonal it lies, which stores the final sum.  Units are taken in ascending key-block order (a unit
//     waits only for units taken before it) -- the longest first -- and the masked bodies are a unit's last.  (Round 2 handed
//     the sums DOWN to key block 0 and had to take the longest units last: the last head of every XCD then ended with one
//     workgroup walking 256 steps alone.)
// RAGGED (chained forms): seq_len is not a multiple of 256 -- loops, running-sum layout and row-constant planes are built on
// the padded length fp.npad; a separate instantiation, so that the kernels of the aligned shapes carry none of it.
// RECT (chained, non-causal, aligned; round 4): a RECTANGULAR, head-strided block -- Nq query rows (a multiple of 32) of every
// head, heads q_hs rows apart, the first of them row q_row0 of the dense row-constant planes, against Nk keys (a multiple of
// 256), heads k_hs rows apart: the unmasked half blocks of the zig-zag causal ring's backward.  Its own instantiation too: the
// square kernels keep one length in one register.
// HD = 64 (chained, aligned, square; round 4): the bodies generated for head_dim 64 (40 MFMAs per sub-tile).  The dQ tile of a
// sub-tile is 32 x 64: wave w forms columns 32 (w & 1) .. + 31 over the 128 keys of half w >> 1 of the workgroup's 256, so a
// sub-tile has TWO running sums per column block (one per key half), each handed from key block to key block exactly like the one
// of head_dim 128 -- the layout of dQacc ([tile][wave][g][lane] x 4 floats) and the bodies' eight loads / stores are the same --
// and fa2_bwd_fused_dq_out_chain64_kernel adds the two.
template <bool CHAIN, bool CAUSAL, bool RAGGED = false, bool RECT = false, int HD = 128>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(39))) fa2_bwd_fused_kernel(FusedArgs fp)
{
    static_assert(CHAIN || !RAGGED, "the atomics form has no ragged variant");
    static_assert(!RECT || (CHAIN && !CAUSAL && !RAGGED), "rectangular blocks: chained, unmasked, aligned");
    static_assert(HD == 128 || (HD == 64 && CHAIN && !RECT), "head_dim 64: chained, square");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const BwdArgs& p = fp.b;
    constexpr int D = HD, ROWB = 2 * D, KS = D / 16, DT = D / 32;
    constexpr int DQW = 128;                         // floats per row of the running-sum layout: 4 waves x 32 columns, either head_dim
    constexpr int TROWS = 32;
    constexpr bool W = D == 128;
    constexpr int QRING = W ? FA2_FUSED_QRING : FA2_FUSED64_QRING, BUFB = W ? FA2_FUSED_BUFB : FA2_FUSED64_BUFB,
                  DSB = W ? FA2_FUSED_DSB : FA2_FUSED64_DSB, LDSB = W ? FA2_FUSED_LDS : FA2_FUSED64_LDS;
    constexpr int CPR = D / 8, RPI = 64 / CPR, NINS = TROWS / RPI;      // one-KiB DMA pieces per tensor per 32-row tile: 8 (4 at head_dim 64)
    constexpr int VF = W ? FA2_FUSED_VF : FA2_FUSED64_VF, DQT = W ? FA2_FUSED_DQT : FA2_FUSED64_DQT,
                  ROFFK = W ? FA2_FUSED_ROFFK : FA2_FUSED64_ROFFK, DSWR = W ? FA2_FUSED_DSWR : FA2_FUSED64_DSWR,
                  DSRD = W ? FA2_FUSED_DSRD : FA2_FUSED64_DSRD, KT = W ? FA2_FUSED_KT : FA2_FUSED64_KT;
    constexpr int VMW = CHAIN ? (W ? 4 : 2) : 63;    // vector-memory operations issued behind the DQT loads and in front of the E chain: the wave's DMA pieces
    int* const mail = reinterpret_cast<int*>(smem + LDSB);      // 16 bytes behind the generated map: the unit taken
    const uint32_t mail_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem) + LDSB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ki = lane & 31;
    const int h = lane >> 5;
    const int N = p.Nk;                          // keys: the TRUE length (tensor ranges, stores).  Square: == Nq == q_hs == k_hs
    const int NP = RAGGED ? fp.npad : N;         // the length the loops, the running-sum layout and the planes are built on
    const int NQ = RECT ? p.Nq : N;              // query rows (RECT: a multiple of 32)
    const int NPQ = RECT ? NQ : NP;
    const int QHS = RECT ? p.q_hs : N, KHS = RECT ? p.k_hs : N;      // rows between consecutive heads, query side / key side
    const int QR0 = RECT ? p.q_row0 : 0;         // the block's first row in the dense row-constant planes
    const int ncb = NP / 256;
    const int ntiles = NPQ / TROWS;
    const int niter = ((ntiles + 1 + 5) / 6) * 6;        // bodies come in sixes (ring of 3 x dS parity); the extra ones see zero rows
    const float c2 = p.scale * kLog2e;
    const size_t rc_plane = (size_t)p.BH * (RECT ? QHS : NP);

    // ---- loop-invariant LDS addresses
    const uint32_t lbase = (uint32_t)(uintptr_t)smem;
    const int trq = (lane & 15) >> 2, trp = lane & 3, trcb = (lane >> 4) & 1;
    uint32_t roff[KS], toff[2 * DT];
#pragma unroll
    for (int s = 0; s < KS; ++s) roff[s] = lbase + QRING + lds_off<D>(ki, 2 * s + h);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            toff[2 * dt + jj] = lbase + QRING + lds_off<D>(8 * jj + 4 * h + trq, 4 * dt + 2 * trcb + (trp >> 1)) + 8 * (trp & 1);
    const uint32_t rcv = lbase + QRING + 2 * TROWS * ROWB + 16 * h;
    static_for<KS>([&](auto S) {                   // this wave's K rows in the K image
        constexpr int sidx = decltype(S)::value;
        fused_vset<ROFFK + sidx>(lbase + lds_off<D>(ki + 64 * wave, 2 * sidx + h));
    });
    // E: this wave's 32 columns of dQ over "its" keys -- head_dim 128: column block = wave, all 256 keys; head_dim 64: column
    // block wave & 1, the 128 keys of half wave >> 1 (the bodies step 16 keys per immediate from these bases)
    const int ecol = W ? wave : (wave & 1), ekey0 = W ? 0 : 128 * (wave >> 1);
    static_for<2>([&](auto JJ) {                   // K^T for this wave's 32 columns; dS^T of the workgroup's tile
        constexpr int jj = decltype(JJ)::value;
        fused_vset<KT + jj>(lbase + ekey0 * ROWB + lds_off<D>(8 * jj + 4 * h + trq, 4 * ecol + 2 * trcb + (trp >> 1)) + 8 * (trp & 1));
        const int row = 8 * jj + 4 * h + trq;
        fused_vset<DSRD + jj>(lbase + DSB + (ekey0 + row) * 64 + 8 * ((4 * trcb + trp) ^ FA2_FUSED_DSKEY(row)));
    });
    static_for<4>([&](auto C) {                    // dS write addresses: 8-byte chunk (4 sp + 2 jp + h) of row 64 w + ki
        constexpr int c = decltype(C)::value;
        fused_vset<DSWR + c>(lbase + DSB + (64 * wave + ki) * 64 + 8 * (((2 * c) | h) ^ FA2_FUSED_DSKEY(ki)));
    });
    const int drow = lane / CPR, dslot = lane % CPR;
    const int prow = wave * RPI + drow;
    const int doff = drow * ROWB + 16 * ((lds_off<D>(prow, dslot) - ROWB * prow) >> 4);
    const uint32_t dqv = CHAIN ? (uint32_t)(wave * 4096 + lane * 16) : (uint32_t)(((4 * h) * D + 32 * wave + ki) * 4);
    const auto rc_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)fp.rc, 0, (int)(2 * rc_plane * 4), 0x00020000);

    int xcc = 0;
    if constexpr (CHAIN) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    }
    int* const prog_base = fp.ctl + kCtlHeadmap + 16 * (p.BH + kHmPad);
    // Units of a queue come in GROUPS of G chains (heads), key-block-major inside a group: ticket u -> group u / (G ncb), key
    // block j = (u % (G ncb)) / G, chain G group + u % G.  Non-causal: G = 1, a head's key blocks one after the other (all
    // units are equally long).  Causal: a unit of key block j walks ntiles - 8 j sub-tiles, and with G = 1 the last head of
    // every queue ended with its LONGEST units started last (a quarter of a CU's whole share each); with a queue's heads taken
    // together the units come longest first over the group.  G = 2 (BH / 8 if that is less, at least 1 -- the first tickets of
    // the eight queues must not grab more heads than there are): a larger group balances better but spreads a queue's 32 CUs
    // over more heads, and a key block that starts long after its predecessor finds that one's running sums gone from the L2.
    // Measured at (4,16,8192,128) causal, same box: G = 1 2.476 ms, 2 2.406, 4 2.470, 8 2.548.  A unit still only waits for units taken from the same
    // queue before it (its predecessor (head, j - 1) is G tickets earlier): the deadlock argument is unchanged.
#ifndef FA2_FUSED_CAUSAL_G
#define FA2_FUSED_CAUSAL_G 2
#endif
    const int G = CAUSAL ? max(1, min(FA2_FUSED_CAUSAL_G, p.BH / 8)) : 1;
    const auto ctl_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)fp.ctl, 0, CHAIN ? fused_ctl_ints(p.BH, ncb) * 4 : 0, 0x00020000);

    // ---- taking a unit (chained form): tid 0 draws tickets from the queue of this XCD until one names a (head, key block) or
    // the queue is exhausted, and leaves (head | -2, key block, error word) in the mailbox; everybody reads it behind a barrier
    auto take_ticket = [&]() {
        if (wave == 0 && fused_lane_id() == 0) {
            int hd, j = 0;
            do {
                const int u = atomicAdd(fp.ctl + kCtlTicket + 32 * xcc, 1);
                const int r = u % (G * ncb);
                const int k = (u / (G * ncb)) * G + r % G;
                j = r / G;
                hd = -2;                                 // -2: the queue is exhausted; -1: this ticket's chain has no head, the next one may
                if (k < p.BH + kHmPad) {
                    int* hm = fp.ctl + kCtlHeadmap + xcc * (p.BH + kHmPad) + k;
                    int v = 0, spins = 0;
                    if (j == 0) {
                        // heads are handed to a queue's chains IN CHAIN ORDER (chain k asks after chain k - 1 has): "chain k
                        // got none" then means "no later chain of this queue gets one", which is what ends a workgroup
                        if (k > 0)
                            while ((v = fused_load_sc1(hm - 1)) == 0)
                                if (fused_spin_over(fp.ctl, ++spins)) break;
                        if (k > 0 && v == 0) {
                            v = -1;                      // a wait ran out (error word raised): nobody takes new work
                        } else {
                            const int g = atomicAdd(fp.ctl + kCtlNextHead, 1);
                            v = g < p.BH ? g + 1 : -1;
                        }
                        fused_store_flag(hm, v);
                    } else {
                        while ((v = fused_load_sc1(hm)) == 0)
                            if (fused_spin_over(fp.ctl, ++spins)) { v = -1; break; }
                    }
                    // a chain without a head: the first chain of a group ends the queue, any other only this ticket
                    hd = v > 0 ? v - 1 : (r % G == 0 ? -2 : -1);
                }
            } while (hd == -1);
            mail[0] = hd;
            mail[1] = j;
            mail[2] = hd >= 0 ? fused_load_sc1(fp.ctl + kCtlError) : 0;     // somebody's wait ran out: nobody waits any more
        }
    };
    // ---- a unit's operands that live for the whole unit: the V fragments of both key blocks straight into v[VF ...] (B
    // operands of dP' = dO V^T: lane = key column), and the K image (the workgroup's 256 keys, swizzled like every other
    // tile: rows for S', columns for dQ) by LDS-DMA: neither passes through compiler-allocated registers.  Issued as soon
    // as the unit is known -- for every unit but a workgroup's first that is BEFORE the previous unit's epilogue, whose
    // dK / dV stores then overlap these loads (round 4; the change-over was ticket, loads, stores one after the other).
    auto issue_unit_loads = [&](int head_, int cb_) {
        const size_t slab_ = (size_t)head_ * KHS * ROWB;
        const char* Kh_ = (const char*)p.K + slab_;
        const char* Vh_ = (const char*)p.V + slab_;
        const int kw0_ = cb_ * 256 + wave * 64;
        const int lane_u = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));    // recomputed: nothing per-lane
        // (keys past the end of a ragged sequence: any finite row will do -- their P is masked to zero -- so the address
        // is clamped; everything else reaches them through range-checked buffer resources)
        const int vk0 = RAGGED ? min(kw0_ + (lane_u & 31), N - 1) : kw0_ + (lane_u & 31);
        const int vk1 = RAGGED ? min(kw0_ + 32 + (lane_u & 31), N - 1) : kw0_ + 32 + (lane_u & 31);
        const char* v0 = Vh_ + (size_t)vk0 * ROWB + 16 * (lane_u >> 5);
        const char* v1 = Vh_ + (size_t)vk1 * ROWB + 16 * (lane_u >> 5);
        static_for<KS>([&](auto S) { fused_load_vfrag<VF, KS, decltype(S)::value>(v0, v1); });
        const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kh_, 0, N * ROWB, 0x00020000);
#pragma unroll
        for (int j = wave; j < 256 / RPI; j += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (fused_lptr_t)(smem + j * 1024), 16, doff, (cb_ * 256 + j * RPI) * ROWB, 0, FA2_FUSED_KV_AUX);
    };

    // the unit a workgroup is about to work on is what the mailbox says (written behind the previous unit's bodies; nothing is
    // carried in registers across the epilogue)
    auto next_unit_loads = [&]() {
        if constexpr (CHAIN) {
            take_ticket();
            __syncthreads();
            const int hn = fused_mail_read<0>(mail_addr);
            if (hn >= 0) issue_unit_loads(hn, fused_mail_read<4>(mail_addr));
        }
    };
    next_unit_loads();

    for (;;) {
#ifdef FA2_FUSED_STATS
        const uint64_t s_pull = __builtin_readcyclecounter();
#endif
        int head, cb, err0 = 0;
        if constexpr (CHAIN) {
            head = fused_mail_read<0>(mail_addr);
            cb = fused_mail_read<4>(mail_addr);
            err0 = fused_mail_read<8>(mail_addr);
            if (head < 0) break;                     // the queue is exhausted
        } else {
            map_block(blockIdx.x, p.BH, ncb, head, cb);
            issue_unit_loads(head, cb);
        }
        const size_t slab = (size_t)head * QHS * ROWB;
        const char* Qh = (const char*)p.Q + slab;
        const char* Gh = (const char*)p.dO + slab;
        const int kw0 = cb * 256 + wave * 64;
        // dK^T, dV^T <- 0: sixteen MFMAs on a zero fragment instead of 256 accumulator writes
        {
            const u32x4 z = {0u, 0u, 0u, 0u};
            static_for<4 * DT>([&](auto T) { fused_acc_zero<decltype(T)::value>(z); });
        }

        // ---- LDS-DMA staging of a 32-row Q / dO tile + its row constants into ring slot `buf`
        const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Qh, 0, NQ * ROWB, 0x00020000);
        const auto g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Gh, 0, NQ * ROWB, 0x00020000);
        const int rcoff = (int)(((wave == 0 ? 0 : rc_plane) + (size_t)head * (RECT ? QHS : NP) + QR0 + (fused_lane_id() & 31)) * 4);
        auto stage = [&](int t, int buf) {
            char* b = smem + QRING + buf * BUFB;
#pragma unroll
            for (int j = wave; j < 2 * NINS; j += 4) {
                const int which = j / NINS, piece = j % NINS;
                const int soff = (t * TROWS + piece * RPI) * ROWB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(which ? g_rsrc : q_rsrc, (fused_lptr_t)(b + which * (TROWS * ROWB) + piece * 1024),
                                                         16, doff, soff, 0, 0);
            }
            if (wave < 2 && fused_lane_id() < 32)        // 32 x -L/scale (wave 0), 32 x -D (wave 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rc_rsrc, (fused_lptr_t)(b + 2 * TROWS * ROWB + wave * 128), 4, rcoff, t * TROWS * 4, 0, 0);
        };
        // steps u = 0 .. n_u - 1 of this unit work on sub-tile tl(u); the bodies from `first_masked` on carry the causal mask
        const int n_u = CAUSAL ? ntiles - 8 * cb : ntiles;
        const int niter_u = CAUSAL ? ((n_u + 1 + 5) / 6) * 6 : niter;
        // non-causal, ragged: the LAST key block holds keys past the end (their K rows read as zeros, so S' = -L/scale and P
        // would not vanish): all of its bodies are the masked ones, the mask being "this lane's key exists".  (Causal: those
        // keys lie above every existing row's diagonal and the last key block's bodies are all masked anyway.)
        const bool ragged_unit = RAGGED && !CAUSAL && cb == ncb - 1;
        const int first_masked = CAUSAL ? ((n_u - 8) / 6) * 6 : (ragged_unit ? 0 : niter_u);
        auto tl = [&](int u) { return CAUSAL ? ntiles - 1 - u : u; };
        stage(tl(0), 0);
#ifdef FA2_FUSED_PREFILL      // timing builds whose bodies issue no DMA (tools/gen_fused_body.py, FA2_GEN_ABL=noDMA): every ring slot holds real rows
        stage(tl(1), 1);
        stage(tl(2), 2);
#endif
        fused_dq_zero<DQT>();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                 // V fragments, K image and the first tile have landed
        if constexpr (W) asm volatile(FA2_FUSED_PRO : : FA2_FUSED_OPS, [vm] "i"(VMW) : FA2_FUSED_CLOBBERS);
        else asm volatile(FA2_FUSED64_PRO : : FA2_FUSED_OPS64, [vm] "i"(VMW) : FA2_FUSED_CLOBBERS);

        const auto dq_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(fp.dQacc + (size_t)head * NPQ * DQW), 0, NPQ * DQW * 4, 0x00020000);
        int* const mine = prog_base + head * ncb + cb;
        const int* const prev = mine - 1;                             // the key block this one takes the running sums from
        const int prev_off = (int)((prev - fp.ctl) * 4);
        const int mine_off = (int)((mine - fp.ctl) * 4);
        if constexpr (CHAIN) fused_seen_set(0);          // what prev was last seen at
        int err = err0;                                  // raised by a body whose wait for the previous key block ran out
#ifdef FA2_FUSED_STATS
        int st_steps = 0, st_polls = 0, st_cycles = 0;
        const uint64_t u0 = __builtin_readcyclecounter();
#endif

        const auto null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)fp.dQacc, 0, 0, 0x00020000);     // every access out of range
        // FAST: a step in the middle of a unit (1 <= t, t + 1 < n_u, and for the causal form the previous key block exists for
        // sub-tile tl(t)): every selection below is decided, what is left between two bodies is a handful of scalar adds.
        // The general form costs ~25 scalar instructions per body (3 % of it: one wave per SIMD, nothing hides them).
        const bool hp_unit = cb > 0;
        const auto lrs_unit = hp_unit && !(FA2_FUSED_DIAG & 2) ? dq_rsrc : null_rsrc;
        auto step = [&](auto BUF, auto PAR, auto MASKED, auto FAST_, int t) {
            constexpr int buf = decltype(BUF)::value, par = decltype(PAR)::value;
            constexpr bool masked = decltype(MASKED)::value;
            constexpr bool FAST = decltype(FAST_)::value;
            if constexpr (CHAIN && FAST) {
                FusedStep f;
                f.drs = dq_rsrc;
                f.lrs = lrs_unit;
                f.dso = (uint32_t)(tl(t - 1) * TROWS * DQW * 4);
                f.lso = (uint32_t)(tl(t) * TROWS * DQW * 4);
                f.qrs = q_rsrc; f.grs = g_rsrc; f.rcrs = rc_rsrc; f.ctl = ctl_rsrc;
                f.qso = (uint32_t)(tl(t + 1) * TROWS * ROWB + wave * 1024);
                f.rcso = (uint32_t)(tl(t + 1) * TROWS * 4);
                f.pvo = (uint32_t)prev_off; f.mso = (uint32_t)mine_off;
                f.need = (hp_unit && !err && !(FA2_FUSED_DIAG & 1)) ? t + 1 : (int)0x80000000;
                f.pval = t;
#ifdef FA2_TEST_HOOKS
                if (fp.fault && cb == 1) f.pval = 0;
#endif
                fused_cbody<buf, par, VMW, false>(roff, toff, rcv, c2, dqv, (uint32_t)doff, (uint32_t)rcoff, lbase + QRING + wave * 1024,
                                                  lbase + QRING + wave * 128, wave, f, err, 0, 0);
            } else if constexpr (CHAIN) {
                // body of step t: DMA of the sub-tile of step t + 1; E forms the dQ tile of the sub-tile of step t - 1 on top of
                // the running sum loaded by body t - 1 and stores it; behind the barrier it publishes "t sub-tiles out", waits
                // until the previous key block of the chain has published t + 1, and loads the running sum for step t
                FusedStep f;
                const bool live = t >= 1 && t <= n_u;
                const bool more = t + 1 < n_u;                                   // step t + 1 has a sub-tile (else: zero rows)
                const bool has_prev = t < n_u && cb > 0;
                f.drs = live ? dq_rsrc : null_rsrc;
                f.lrs = has_prev && !(FA2_FUSED_DIAG & 2) ? dq_rsrc : null_rsrc;
                f.dso = (uint32_t)__builtin_amdgcn_readfirstlane(tl(t - 1) * TROWS * DQW * 4);
                f.lso = (uint32_t)__builtin_amdgcn_readfirstlane(tl(t) * TROWS * DQW * 4);
                f.qrs = q_rsrc; f.grs = g_rsrc; f.rcrs = rc_rsrc; f.ctl = ctl_rsrc;
                f.qso = more ? (uint32_t)(tl(t + 1) * TROWS * ROWB + wave * 1024) : 0x40000000u;      // out of range: zeros
                f.rcso = more ? (uint32_t)(tl(t + 1) * TROWS * 4) : 0x40000000u;
                f.pvo = (uint32_t)prev_off; f.mso = (uint32_t)mine_off;
                f.need = (has_prev && !err && !(FA2_FUSED_DIAG & 1)) ? t + 1 : (int)0x80000000;
                f.pval = t < n_u ? t : n_u;
#ifdef FA2_TEST_HOOKS
                if (fp.fault && cb == 1) f.pval = 0;      // tests: whoever waits for this key block must give up, not hang
#endif
                int lo0 = 0, lo1 = 0;
                if constexpr (masked) {      // key - 32 tile - 4 h for the lane's two keys (recomputed: nothing per-lane is kept)
                    const int lane_m = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                    if constexpr (CAUSAL) {
                        lo0 = kw0 + (lane_m & 31) - 4 * (lane_m >> 5) - TROWS * tl(t);
                        lo1 = lo0 + 32;
                    } else {                 // keep (lo <= row) everything of an existing key, nothing of a key past the end
                        lo0 = kw0 + (lane_m & 31) < N ? -0x40000000 : 0x40000000;
                        lo1 = kw0 + 32 + (lane_m & 31) < N ? -0x40000000 : 0x40000000;
                    }
                }
#ifdef FA2_FUSED_STATS2
                const uint64_t b0 = __builtin_readcyclecounter();
#endif
                fused_cbody<buf, par, VMW, masked>(roff, toff, rcv, c2, dqv, (uint32_t)doff, (uint32_t)rcoff, lbase + QRING + wave * 1024,
                                                   lbase + QRING + wave * 128, wave, f, err, lo0, lo1);
#ifdef FA2_FUSED_STATS2
                st_cycles += (int)(__builtin_readcyclecounter() - b0);       // cycles inside the bodies (reported as "waited")
#endif
            } else {
                static_assert(W || CHAIN, "head_dim 64: chained form only");
                stage(t + 1, (buf + 1) % 3);
                if constexpr (W) fused_body<buf, par, VMW>(roff, toff, rcv, c2);
#ifndef FA2_FUSED_NO_DQ                              // diagnostic build: how long the kernel takes without the dQ traffic
                // the body's E stage has finished the dQ tile of sub-tile t - 1
                if (t >= 1 && t <= ntiles) fused_dq_atomic<DQT>(dq_rsrc, dqv, (uint32_t)__builtin_amdgcn_readfirstlane((t - 1) * TROWS * D * 4));
#endif
                fused_dq_zero<DQT>();
            }
        };
        auto six = [&](auto MASKED, auto FAST_, int t) {
            step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, MASKED, FAST_, t);
            step(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, MASKED, FAST_, t + 1);
            step(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, MASKED, FAST_, t + 2);
            step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, MASKED, FAST_, t + 3);
            step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, MASKED, FAST_, t + 4);
            step(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, MASKED, FAST_, t + 5);
        };
        // sixes whose every step is FAST: t >= 6 (so t >= 1), t + 5 + 1 < n_u; the causal form's masked bodies come behind them
        const int fast_end = CHAIN ? n_u - 2 : -1;                            // last step that may be FAST
        {
            const int plain_end = first_masked;
            int t = 0;
            asm volatile("" : "+s"(t));          // not a literal: the bodies take values derived from it in SGPR operands
            if (t < plain_end) { six(std::false_type{}, std::false_type{}, t); t += 6; }
#pragma unroll 1
            for (; t + 5 <= fast_end && t < plain_end; t += 6) six(std::false_type{}, std::true_type{}, t);
#pragma unroll 1
            for (; t < plain_end; t += 6) six(std::false_type{}, std::false_type{}, t);
            if constexpr (CHAIN && (CAUSAL || RAGGED)) {
#pragma unroll 1
                for (; t < niter_u; t += 6) six(std::true_type{}, std::false_type{}, t);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (CHAIN)
            if (err && !err0 && wave == 0 && fused_lane_id() == 0) atomicExch(fp.ctl + kCtlError, 1);
#ifdef FA2_FUSED_STATS
        const uint64_t s_loop = __builtin_readcyclecounter();
#endif
        // ---- the NEXT unit, before this one's epilogue: its ticket, and (behind the barrier that also says every wave is done
        // with the K image and its V fragments) its K image and V fragments -- they load while dK / dV are being stored
        next_unit_loads();

        mfma_acc_settle();
        // lane indices recomputed here so that nothing per-lane has to live (or spill) across the loop
        const int lane_ep = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int ki_ep = lane_ep & 31, h_ep = lane_ep >> 5;
        // a lane holds 4 consecutive columns of its key per register quad, its partner lane (+32) the next 4: one
        // v_permlane32_swap per packed dword pairs them up, so that every lane stores 16 contiguous bytes
        static_for<2>([&](auto KB) {
            constexpr int kb = decltype(KB)::value;
            const int key = kw0 + 32 * kb + ki_ep;
            // (uniform 64-bit bases + ONE 32-bit per-lane offset for both tensors: two 64-bit per-lane pointers are four of
            // the compiler's 39 registers, and it has none to spare here)
            const size_t slab_k = RECT ? (size_t)head * KHS * ROWB : slab;
            char* const dKb = (char*)p.dK + slab_k;
            char* const dVb = (char*)p.dV + slab_k;
            const uint32_t koff = (uint32_t)key * ROWB + 16u * h_ep;
            char* dKk = dKb + koff;
            char* dVk = dVb + koff;
            static_for<2 * DT>([&](auto G) {
                constexpr int dt = decltype(G)::value / 2, gp = decltype(G)::value % 2;
                constexpr int RK = 16 * (kb * DT + dt) + 8 * gp, RV = 32 * DT + RK;
                auto pack4 = [&](float a, float b, float c, float d, float sc) {
                    bf16x4 v;
                    v[0] = (__bf16)(a * sc); v[1] = (__bf16)(b * sc); v[2] = (__bf16)(c * sc); v[3] = (__bf16)(d * sc);
                    return __builtin_bit_cast(u32x2, v);
                };
                auto emit = [&](u32x2 x, u32x2 y, char* dst) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(x[0], y[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(x[1], y[1], false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
#ifdef FA2_FUSED_ST_NT
                    if (!RAGGED || key < N) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst + 2 * (32 * dt + 16 * gp)));
#else
                    if (!RAGGED || key < N) *reinterpret_cast<u32x4*>(dst + 2 * (32 * dt + 16 * gp)) = o;      // (ragged: keys past the end)
#endif
                };
                emit(pack4(acc_read<RK>(), acc_read<RK + 1>(), acc_read<RK + 2>(), acc_read<RK + 3>(), p.scale),
                     pack4(acc_read<RK + 4>(), acc_read<RK + 5>(), acc_read<RK + 6>(), acc_read<RK + 7>(), p.scale), dKk);
                emit(pack4(acc_read<RV>(), acc_read<RV + 1>(), acc_read<RV + 2>(), acc_read<RV + 3>(), 1.0f),
                     pack4(acc_read<RV + 4>(), acc_read<RV + 5>(), acc_read<RV + 6>(), acc_read<RV + 7>(), 1.0f), dVk);
            });
        });
#ifdef FA2_FUSED_STATS
        if (CHAIN && wave == 0 && fused_lane_id() == 0) {      // debug: ctl[kCtlError + 1 ...] = steps that waited, polls, cycles waited, cycles total
            atomicAdd(fp.ctl + kCtlError + 1, st_steps);
            atomicAdd(fp.ctl + kCtlError + 2, st_polls);
            atomicAdd((unsigned long long*)(fp.ctl + kCtlError + 4), (unsigned long long)st_cycles);
            atomicAdd((unsigned long long*)(fp.ctl + kCtlError + 6), (unsigned long long)(__builtin_readcyclecounter() - u0));
            int* rec = prog_base + p.BH * ncb + 8 * (head * ncb + cb);       // per unit: xcc, cycles waited, steps waited, start, end
            rec[0] = xcc; rec[1] = st_cycles; rec[2] = (int)(u0 - s_pull); rec[3] = (int)(__builtin_readcyclecounter() - s_loop);
            *(unsigned long long*)(rec + 4) = u0; *(unsigned long long*)(rec + 6) = __builtin_readcyclecounter();
        }
#endif
        if constexpr (!CHAIN) break;
        // (no barrier here: the mailbox is next written behind the next unit's bodies, which are full of barriers)
    }
}

// dQ (bf16) = scale * dQacc (fp32); poisoned when the chained kernel reported a spin that ran out
__global__ void __launch_bounds__(256) fa2_bwd_fused_dq_out_kernel(const float* __restrict__ acc, __bf16* __restrict__ dQ, size_t n8, float scale,
                                                                    const int* __restrict__ err)
{
    const float poison = (err && *err) ? __builtin_nanf("") : 0.0f;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
        const f32x4 a = reinterpret_cast<const f32x4*>(acc)[2 * i], b = reinterpret_cast<const f32x4*>(acc)[2 * i + 1];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = (__bf16)(a[e] * scale + poison); o[4 + e] = (__bf16)(b[e] * scale + poison); }
        reinterpret_cast<bf16x8*>(dQ)[i] = o;
    }
}

// the same for the chained kernel's layout: slot i = (((head * NP / 32 + tile) * 4 + wave) * 4 + g) * 64 + lane holds rows
// 32 tile + 8 g + 4 (lane >> 5) + (0 .. 3) of column 32 wave + (lane & 31); NP = N rounded up to 256, rows >= N are not stored;
// consecutive heads of dQ are qhs rows apart (N, or a rectangular block's head stride)
__global__ void __launch_bounds__(256) fa2_bwd_fused_dq_out_chain_kernel(const float* __restrict__ acc, __bf16* __restrict__ dQ, size_t n4, float scale,
                                                                          const int* __restrict__ err, int N, int NP, int qhs)
{
    const float poison = *err ? __builtin_nanf("") : 0.0f;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t tiles = (size_t)(NP / 32);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 a = reinterpret_cast<const f32x4*>(acc)[i];
        const int lane = (int)(i & 63), g = (int)((i >> 6) & 3), wave = (int)((i >> 8) & 3);
        const size_t tg = i >> 10, head = tg / tiles;
        const int row = (int)(tg % tiles) * 32 + 8 * g + 4 * (lane >> 5);
        __bf16* o = dQ + (head * (size_t)qhs + row) * 128 + 32 * wave + (lane & 31);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (row + e < N) o[(size_t)e * 128] = (__bf16)(a[e] * scale + poison);
    }
}

// head_dim 64: slot (tile, wave, g, lane) holds rows 32 tile + 8 g + 4 (lane >> 5) + (0 .. 3) of column 32 (wave & 1) + (lane & 31),
// summed over the keys of half wave >> 1 of every key block: dQ = scale x (the sum of half 0 + the sum of half 1).  n4h = BH x
// (N / 32) x 2 column blocks x 4 x 64 threads' worth.
__global__ void __launch_bounds__(256) fa2_bwd_fused_dq_out_chain64_kernel(const float* __restrict__ acc, __bf16* __restrict__ dQ, size_t n4h, float scale,
                                                                            const int* __restrict__ err, int N, int NP)
{
    const float poison = *err ? __builtin_nanf("") : 0.0f;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t tiles = (size_t)(NP / 32);      // NP = N rounded up to 256: rows >= N are not stored
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4h; i += stride) {
        const int lane = (int)(i & 63), g = (int)((i >> 6) & 3), cbk = (int)((i >> 8) & 1);
        const size_t tg = i >> 9, head = tg / tiles;
        const size_t slot = ((tg * 4 + cbk) * 4 + g) * 64 + lane;
        const f32x4 a = reinterpret_cast<const f32x4*>(acc)[slot], b = reinterpret_cast<const f32x4*>(acc)[slot + 2 * 4 * 64];
        const int row = (int)(tg % tiles) * 32 + 8 * g + 4 * (lane >> 5);
        __bf16* o = dQ + (head * (size_t)N + row) * 64 + 32 * cbk + (lane & 31);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (row + e < N) o[(size_t)e * 64] = (__bf16)((a[e] + b[e]) * scale + poison);
    }
}

// ragged launches: the row-constant planes [2][BH][N] copied to [2][BH][NP] with (-1e30, 0) in the rows past the end -- a row
// that does not exist then has S' = -1e30, P = exp2(-huge) = 0 and dS = 0 whatever the (zero) Q / dO rows give
__global__ void __launch_bounds__(256) fa2_bwd_fused_rcpad_kernel(const float* __restrict__ rc, float* __restrict__ out, int BH, int N, int NP)
{
    const size_t per = (size_t)BH * NP, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * per) return;
    const size_t plane = i / per, r = i % per, head = r / NP;
    const int row = (int)(r % NP);
    out[i] = row < N ? rc[plane * (size_t)BH * N + head * (size_t)N + row] : (plane == 0 ? -1.0e30f : 0.0f);
}

// The ordered hand-off relies on one hardware property: a workgroup's plain stores land in the L2 of the XCC whose id it
// reads from HW_REG_XCC_ID, and sc1 loads issued on that XCC read that L2.  That was validated on gfx950 in SPX mode (one
// device = 8 XCCs x 32 CUs = 256 CUs); on anything else the launcher does not take the chained form (fa2_backward then runs
// the dQ and dK/dV kernels and fa2_backward_plan says why).
bool bwd_fused_device_ok(const char** why)
{
    static int verdict[64] = {};          // 0 = unknown, 1 = ok, 2 = not gfx950, 3 = not 256 CUs
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { if (why) *why = "no current device"; return false; }
    if (!verdict[dev]) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) != hipSuccess) { if (why) *why = "device properties unavailable"; return false; }
        verdict[dev] = strncmp(pr.gcnArchName, "gfx950", 6) != 0 ? 2 : pr.multiProcessorCount != 256 ? 3 : 1;
    }
    static const char* const text[] = {"", "single five-product kernel (gfx950, 256 CUs: the layout the ordered hand-off was validated on)",
                                       "two kernels: device is not gfx950", "two kernels: device does not expose 256 CUs (partitioned GPU): "
                                       "the ordered hand-off was validated in SPX mode only"};
    if (why) *why = text[verdict[dev]];
    return verdict[dev] == 1;
}

// the control block's error word of the last chained launch that used `ctl` (host copy; the caller has synchronised)
hipError_t bwd_fused_read_error(const int* ctl, int* err, hipStream_t stream)
{
    hipError_t e = hipMemcpyAsync(err, ctl + kCtlError, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(stream);
}

// a backward on this workspace that runs no hand-off (the two kernels, the atomics form): the error word then says so
hipError_t bwd_fused_clear_error(int* ctl, hipStream_t stream)
{
    return launch_fill_f32(reinterpret_cast<float*>(ctl + kCtlError), 1, 0.0f, stream);      // (a kernel, not hipMemsetAsync: see the launcher)
}

// (the last 8 ints per unit are only written by -DFA2_FUSED_STATS builds: tools/gpu_stats_fused.py)
size_t bwd_fused_ctl_bytes(int BH, int N) { return (size_t)(kCtlHeadmap + 16 * (BH + kHmPad) + 9 * BH * ((N + 255) / 256)) * sizeof(int); }

hipError_t launch_bwd_fused_bf16(const BwdArgs& a, float* dQacc, int* ctl, int mode, hipStream_t stream, float* rcpad)
{
    const int npad = (a.Nk + 255) / 256 * 256;
    const bool ragged = npad != a.Nk;            // chained form only: needs `rcpad` (2 BH npad floats)
    // a rectangular and / or head-strided block (fa2_backward_block: the causal ring's unmasked half blocks): chained form,
    // no mask, both lengths aligned; everything else is the dense square problem
    const bool rect = a.Nq != a.Nk || a.q_hs != a.Nq || a.k_hs != a.Nk || a.q_row0 != 0;
    if ((a.d != 128 && a.d != 64) || a.Nk < 1 || a.Nq < 1) return hipErrorInvalidValue;
    if (a.d == 64 && (mode != 1 || rect)) return hipErrorInvalidValue;          // head_dim 64: chained, square
    if (rect && (mode != 1 || a.causal || ragged || a.Nq % 32 != 0 || a.q_row0 < 0 || a.q_hs < a.q_row0 + a.Nq || a.k_hs < a.Nk))
        return hipErrorInvalidValue;
    if (ragged && (mode != 1 || !rcpad)) return hipErrorInvalidValue;
    if (a.causal && (mode != 1 || a.causal_shift != 0)) return hipErrorInvalidValue;
    if (mode == 1 && (a.phases & 8) && !bwd_fused_device_ok(nullptr)) return hipErrorNotSupported;
    hipError_t e = hipSuccess;
    if (a.phases & 1) {
        BwdArgs d = a;
        d.phases = 1;
        e = launch_bwd_bf16(d, stream);                   // D = rowsum(dO o O) and the row constants (kernel 0)
        if (e != hipSuccess) return e;
    }
    if (!(a.phases & 8)) return hipSuccess;
    const size_t elems = (size_t)a.BH * (rect ? a.Nq : npad) * 128;
    const int units = a.BH * (npad / 256);
    if (ragged) {
        const size_t n = (size_t)2 * a.BH * npad;
        hipLaunchKernelGGL(fa2_bwd_fused_rcpad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)a.RC, rcpad,
                           a.BH, a.Nk, npad);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    FusedArgs fa{a, dQacc, ctl, npad, ragged ? (const float*)rcpad : (const float*)a.RC, g_hook_fault};
    const int lds = (a.d == 64 ? FA2_FUSED64_LDS : FA2_FUSED_LDS) + 16;
    if (mode == 0) {
        e = launch_fill_f32(dQacc, elems, 0.0f, stream);
        if (e != hipSuccess) return e;
        e = bwd_fused_clear_error(ctl, stream);
        if (e != hipSuccess) return e;
        static bool set_f[64] = {};
        e = ensure_dynamic_lds(fa2_bwd_fused_kernel<false, false>, lds, set_f);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((fa2_bwd_fused_kernel<false, false>), dim3((unsigned)units), dim3(256), lds, stream, fa);
    } else {
        static int cus[64] = {};
        int dev = 0;
        e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!cus[dev]) {
            e = hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev);
            if (e != hipSuccess) return e;
        }
        // The control block is zeroed by a KERNEL (0.0f is the all-zero word), not by hipMemsetAsync: captured into a HIP graph the
        // memset node of the 81 KB block of the bench shape did not take effect on replay (round 4: the replayed step ran in
        // 1.8 ms -- the stale unit queues read "exhausted" and the grid left at once, dQ / dK / dV untouched -- while the 4 KB block of
        // the (1,4,1024,128) test replayed fine).  A kernel node replays like any other launch.
        e = launch_fill_f32(reinterpret_cast<float*>(ctl), bwd_fused_ctl_bytes(a.BH, a.Nk) / sizeof(float), 0.0f, stream);
        if (e != hipSuccess) return e;
        static bool set_t[64] = {}, set_c[64] = {};
        // a.reserve_cus: the ring backward's exchanges (RCCL kernels on the communication stream) must find a CU while this
        // grid runs -- its workgroups are persistent and fill a CU's register file, nothing else becomes resident beside them
        int avail = cus[dev] - (a.reserve_cus > 0 ? a.reserve_cus : 0);
        if (avail < 1) avail = 1;
        int wgs = units < avail ? units : avail;
        // test builds: fewer workgroups than CUs (the unit queues must drain with ANY number of resident workgroups, down
        // to one -- the claim the hand-off's deadlock freedom rests on)
        if (g_hook_grid >= 1 && g_hook_grid < wgs) wgs = g_hook_grid;
        FA2_HOOK_NOTE_GRID(wgs);
        const dim3 grid((unsigned)wgs);
        static bool set_cr[64] = {}, set_tr[64] = {}, set_re[64] = {}, set_64[64] = {}, set_64c[64] = {}, set_64r[64] = {}, set_64cr[64] = {};
        if (a.d == 64 && a.causal && ragged) {
            e = ensure_dynamic_lds(fa2_bwd_fused_kernel<true, true, true, false, 64>, lds, set_64cr);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fa2_bwd_fused_kernel<true, true, true, false, 64>), grid, dim3(256), lds, stream, fa);
        } else if (a.d == 64 && ragged) {
            e = ensure_dynamic_lds(fa2_bwd_fused_kernel<true, false, true, false, 64>, lds, set_64r);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fa2_bwd_fused_kernel<true, false, true, false, 64>), grid, dim3(256), lds, stream, fa);
        } else if (a.d == 64 && a.causal) {
            e = ensure_dynamic_lds(fa2_bwd_fused_kernel<true, true, false, false, 64>, lds, set_64c);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fa2_bwd_fused_kernel<true, true, false, false, 64>), grid, dim3(256), lds, stream, fa);
        } else if (a.d == 64) {
            e = ensure_dynamic_lds(fa2_bwd_fused_kernel<true, false, false, false, 64>, lds, set_64);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fa2_bwd_fused_kernel<true, false, false, false, 64>), grid, dim3(256), lds, stream, fa);
        } else if (rect) {
            e = ensure_dynamic_lds(fa2_bwd_fused_kernel<true, false, false, true>, lds, set_re);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fa2_bwd_fused_kernel<true, false, false, true>), grid, dim3(256), lds, stream, fa);
        } else if (a.causal && ragged) {
            e = ensure_dynamic_lds(fa2_bwd_fused_kernel<true, true, true>, lds, set_cr);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fa2_bwd_fused_kernel<true, true, true>), grid, dim3(256), lds, stream, fa);
        } else if (a.causal) {
            e = ensure_dynamic_lds(fa2_bwd_fused_kernel<true, true>, lds, set_c);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fa2_bwd_fused_kernel<true, true>), grid, dim3(256), lds, stream, fa);
        } else if (ragged) {
            e = ensure_dynamic_lds(fa2_bwd_fused_kernel<true, false, true>, lds, set_tr);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fa2_bwd_fused_kernel<true, false, true>), grid, dim3(256), lds, stream, fa);
        } else {
            e = ensure_dynamic_lds(fa2_bwd_fused_kernel<true, false>, lds, set_t);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fa2_bwd_fused_kernel<true, false>), grid, dim3(256), lds, stream, fa);
        }
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (mode == 0)
        hipLaunchKernelGGL(fa2_bwd_fused_dq_out_kernel, dim3(2048), dim3(256), 0, stream, dQacc, (__bf16*)a.dQ, elems / 8, a.scale,
                           (const int*)nullptr);
    else if (a.d == 64)
        hipLaunchKernelGGL(fa2_bwd_fused_dq_out_chain64_kernel, dim3(4096), dim3(256), 0, stream, dQacc, (__bf16*)a.dQ, elems / 8, a.scale,
                           ctl + kCtlError, a.Nk, npad);
    else
        hipLaunchKernelGGL(fa2_bwd_fused_dq_out_chain_kernel, dim3(4096), dim3(256), 0, stream, dQacc, (__bf16*)a.dQ, elems / 4, a.scale,
                           ctl + kCtlError, rect ? a.Nq : a.Nk, rect ? a.Nq : npad, rect ? a.q_hs : a.Nk);
    return hipGetLastError();
}

}  // namespace fa2
