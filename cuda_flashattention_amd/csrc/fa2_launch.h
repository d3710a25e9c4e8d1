// fa2_launch.h -- host-side launcher declarations (internal; the public surface is
// include/fa2_mi355x.h).  Each launcher validates nothing: fa2_capi.cpp owns argument
// checking and status codes, the launchers only pick a template instance and enqueue it.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace fa2 {

// One forward problem: BH independent [Nq, d] query slabs against [Nk, d] key/value slabs.
// Plain attention has Nq == Nk; a ring step runs Nq local query rows against the Nk rows of
// the resident K/V shard and carries (Oacc, Lrun, Mrun) between launches.
struct FwdArgs {
    const void* Q;    // [BH][Nq][d]   bf16
    const void* K;    // [BH][Nk][d]   bf16
    const void* V;    // [BH][Nk][d]   bf16
    void* O;          // [BH][Nq][d]   bf16 (written when finalize != 0)
    float* L;         // [BH][Nq]      natural-log LSE when finalize, running sum l otherwise
    float* Oacc;      // [BH][Nq][d]   fp32 un-normalised accumulator (ring state) or nullptr
    float* M;         // [BH][Nq]      running max in natural (scaled) units (ring state) or nullptr
    int BH, Nq, Nk, d;
    float scale;
    int causal;       // keys j > i + causal_shift are masked
    int causal_shift; // global key index of the shard's first key minus that of the first query
    int resume;       // 1: start from (Oacc, L, M); 0: start from (0, 0, -inf)
    int finalize;     // 1: write O = acc / l and L = m + ln l; 0: store state
    int q_hs, k_hs;   // rows between consecutive heads of Q/O/Oacc/L/M and of K/V (0: Nq, Nk -- dense slabs).
                      // Larger strides address a row range of every head (the zig-zag chunks of the causal ring).
};

hipError_t launch_fwd1_bf16(const FwdArgs& a, hipStream_t stream);     // generated main loop: one wave per SIMD (d = 128), two (d = 64)
// acc (fp32) = (init) or += src (bf16): `rows` runs of `cols` elements, `pitch` elements apart in both.
hipError_t launch_accumulate_bf16(float* acc, const void* src, size_t rows, size_t cols, size_t pitch, int init, hipStream_t stream);
// Ring epilogue: O = bf16(Oacc / l), L = m + ln l for `rows` consecutive rows (l arrives in L).
hipError_t launch_finalize_state(const float* Oacc, const float* M, float* L, void* O, size_t rows, int d, hipStream_t stream);

// fp8 (OCP e4m3) forward, d = 128: Q, K, V fp8 [BH][N][128]; O bf16; Vt: [BH][128][Npad] fp8 scratch the
// launcher fills with V transposed (Npad = N rounded up to 64); kn: [BH][Npad / 64] fp32 scratch it fills with the largest
// |k| of every 64 keys (fa2_fwd_fp8.hip: where the softmax needs no lane maxima).
struct FwdFp8Args {
    const void* Q; const void* K; const void* V;
    void* Vt;
    float* kn;
    void* O;
    float* L;
    int BH, N, Npad, d;
    float scale;      // what multiplies q . k of the STORED e4m3 values: softmax_scale x q_descale x k_descale
    int causal;
    float o_scale;    // what multiplies O: v_descale (1 = V holds its own values)
};
hipError_t launch_fwd_fp8(const FwdFp8Args& a, hipStream_t stream);

struct BwdArgs {
    const void* Q; const void* K; const void* V; const void* O; const void* dO;  // bf16
    const float* L;   // [BH][q_hs] natural-log LSE over ALL keys of the row (pointer at row q_row0 of head 0, like Q)
    void* dQ; void* dK; void* dV;   // bf16
    float* D;         // [BH][q_hs] workspace plane: rowsum(dO o O)                (dense, NOT offset by q_row0)
    float* RC;        // [2][BH][q_hs] workspace planes: -L/scale and -D, the row constants of the dK/dV kernel
    int BH, Nq, Nk, d;
    int q_hs, k_hs;   // rows between consecutive heads of Q/O/dO/dQ/L and of K/V/dK/dV (>= Nq, Nk; dense: == Nq, Nk)
    int q_row0;       // the block's rows are rows [q_row0, q_row0 + Nq) of every head (indexes the workspace planes)
    float scale;
    int causal;       // key j is visible to local row i iff j <= i + causal_shift
    int causal_shift;
    int phases;       // bit 0: D = rowsum(dO o O), bit 1: dQ kernel, bit 2: dK/dV kernel (7 = all)
    int reserve_cus;  // single-kernel form: CUs its persistent grid leaves free (for a communication kernel on another stream)
};

hipError_t launch_bwd_bf16(const BwdArgs& a, hipStream_t stream);
// Single-kernel five-product backward (fa2_bwd_fused.hip): d = 128, dense, square; causal with mode 1 only.
// dQacc: [BH][NP][128] fp32 scratch (NP = N rounded up to 256); ctl: bwd_fused_ctl_bytes of scratch; mode 0 = dQ by fp32
// atomics (N % 256 == 0 only), 1 = dQ handed from key block to key block in a fixed order (deterministic; any N: a ragged
// launch also needs rcpad, 2 BH NP floats, for the padded row-constant planes).  hipErrorInvalidValue for shapes it does not take.
size_t bwd_fused_ctl_bytes(int BH, int N);
hipError_t launch_bwd_fused_bf16(const BwdArgs& a, float* dQacc, int* ctl, int mode, hipStream_t stream, float* rcpad = nullptr);
// Whether the current device has the layout the ordered hand-off (mode 1) was validated on; *why = a static sentence.
bool bwd_fused_device_ok(const char** why);
// Synchronises `stream` and reads the error word a chained launch leaves in its control block (non-zero: a bounded wait
// ran out and the output pass turned dQ into NaNs).
hipError_t bwd_fused_read_error(const int* ctl, int* err, hipStream_t stream);
hipError_t bwd_fused_clear_error(int* ctl, hipStream_t stream);      // error word <- 0 (a backward on this workspace without a hand-off)

// fp32 family (exact f32 MFMA, any d <= 128, any N): the reference-signature drop-ins.
struct F32Args {
    const float* Q; const float* K; const float* V; float* O; float* L;
    const float* dO; float* dQ; float* dK; float* dV; float* D;
    int BH, N, d;
    float scale;
    int causal;
    int phases;       // backward only, as BwdArgs::phases
    // forward only -- resumable step (ring): Nk keys in the resident shard (0 = N), running max
    // in M, running sum in L and the un-normalised accumulator in O itself between steps,
    // exactly the reference's state layout (ring_attention_kernel.cu:67-79, :125-137).
    int Nk;
    float* M;
    int resume, finalize;   // plain forward: resume = 0, finalize = 1
};
hipError_t launch_fwd_f32(const F32Args& a, hipStream_t stream);
hipError_t launch_bwd_f32(const F32Args& a, hipStream_t stream);

// Raises the dynamic-LDS limit of `kern` on the current device once (per template instance:
// pass a function-local static flag array).  Benign race: the attribute is idempotent.
template <typename Kern>
inline hipError_t ensure_dynamic_lds(Kern kern, int bytes, bool (&done)[64])
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    return hipSuccess;
}

// FlashAttention-1 restatement (fa1_f32.hip): one head, fp32, didactic baseline row.
hipError_t launch_fa1_f32(const float* Q, const float* K, const float* V, float* O, float* l, float* m, int N, int d, int Bc,
                          hipStream_t stream);

// Element-wise helpers (fa2_util.hip).
hipError_t launch_fill_f32(float* p, size_t n, float value, hipStream_t stream);
hipError_t launch_mfma_probe(const void* in, float* out, int iters, int blocks, hipStream_t stream);
hipError_t launch_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t stream);
hipError_t launch_bf16_to_f32(const void* src, float* dst, size_t n, hipStream_t stream);
hipError_t launch_read_clocks(unsigned long long* out, hipStream_t stream);

}  // namespace fa2
