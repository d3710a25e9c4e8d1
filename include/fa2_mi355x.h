/*
 * fa2_mi355x.h -- C ABI of the MI355X-native FlashAttention-2 hot path (libfa2_mi355x.so).
 *
 * This is the drop-in boundary for the three host wrappers the reference
 * (terryye/cuda_FlashAttention) calls from its test mains -- there is no registry or FFI in
 * the reference, the wrappers ARE its operator interface:
 *
 *   flash_attention_2_forward   src/02_flash_attention_v2_forward/flash_attention_kernel.cu:300-309
 *                               (cleaned copy src/03_flash_attention_v2_ring/common/flash_attention_kernel.cu:133-172)
 *   flash_attention_2_backward  src/02_flash_attention_v2_backward/flash_attention_backward_kernel.cu:249-262
 *   ring_attention_forward      src/03_flash_attention_v2_ring/common/ring_attention_kernel.cu:143-156
 *                               (declared in fa2_ring_mi355x.h: it needs RCCL)
 *
 * Conventions (all entry points):
 *   - plain C, raw DEVICE pointers and sizes, no framework types;
 *   - tensors are contiguous row-major [B][H][N][d] ("head slabs" [N][d], each an independent
 *     instance of the reference's single-head problem); L is [B][H][N] fp32 and holds the
 *     NATURAL-log logsumexp of the scaled scores, exactly what the reference stores
 *     (flash_attention_kernel.cu:291-294);
 *   - the caller owns every tensor; the library never allocates in the extended entry points;
 *   - stream-ordered and asynchronous: work is enqueued on `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream) and the call returns; nothing synchronises;
 *   - return value: FA2_OK (0) or a negative status.  The library never aborts -- the
 *     reference's assert()/exit() policy (cuda_helper.h:74-82, nccl_utils.h:11-18,
 *     flash_attention_kernel.cu:317) becomes status codes.
 */
#ifndef FA2_MI355X_H
#define FA2_MI355X_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------- */
#define FA2_OK                     0
#define FA2_ERR_NULL_POINTER      -1
#define FA2_ERR_INVALID_SHAPE     -2   /* B, H, N <= 0, scale <= 0, ... */
#define FA2_ERR_UNSUPPORTED_HEAD_DIM -3 /* bf16/fp8: d must be 64 or 128; fp32: 1 <= d <= 128 */
#define FA2_ERR_UNSUPPORTED_DTYPE -4
#define FA2_ERR_WORKSPACE         -5   /* workspace NULL or smaller than *_workspace_bytes() */
#define FA2_ERR_UNSUPPORTED       -6   /* combination not implemented (e.g. causal ring) */
#define FA2_ERR_HANDOFF_TIMEOUT   -7   /* fa2_backward_status: a bounded wait of the single-kernel backward ran out; dQ is NaN */
#define FA2_ERR_HIP_BASE          -1000 /* -(1000 + hipError_t) */
#define FA2_ERR_RCCL_BASE         -2000 /* -(2000 + ncclResult_t) */

/* ---- element types of Q/K/V/O/dO/dQ/dK/dV -------------------------------------------- */
#define FA2_DTYPE_BF16 0   /* primary: bf16 in, fp32 accumulate on v_mfma_f32_32x32x16_bf16 */
#define FA2_DTYPE_F32  1   /* the reference's type: exact f32 MFMA (v_mfma_f32_32x32x2_f32)  */
#define FA2_DTYPE_FP8_E4M3 2 /* forward only: OCP e4m3 Q/K/V on v_mfma_f32_32x32x64_f8f6f4, O in bf16, d = 128 */

const char* fa2_version(void);
const char* fa2_status_string(int status);

/* ======================================================================================
 * Reference-signature drop-ins: single head, fp32, default stream -- argument for argument
 * the reference's wrappers, with `void` widened to an int status.  Any seq_len >= 1, any
 * 1 <= head_dim <= 128 (the reference asserts head_dim <= 64 / 128).
 * ==================================================================================== */

/* replaces flash_attention_2_forward (02_forward/flash_attention_kernel.cu:300-309) */
int flash_attention_2_forward(const float* Q, const float* K, const float* V,
                              float* O, float* L,
                              int seq_len, int head_dim, float softmax_scale);

/* replaces flash_attention_2_backward (02_backward/flash_attention_backward_kernel.cu:249-262).
 * Like the reference it overwrites dQ, dK, dV (the reference memsets dK, dV itself, :282-283).
 * Scratch for D = rowsum(dO o O) comes from a per-device cache owned by the library. */
int flash_attention_2_backward(const float* Q, const float* K, const float* V,
                               const float* O, const float* L, const float* dO,
                               float* dQ, float* dK, float* dV,
                               int seq_len, int head_dim, float softmax_scale);

/* replaces flash_attention (01_flash_attention_v1/main.cu:7-20), the FlashAttention-1 step of the reference's staircase:
 * O = softmax(Q K^T / sqrt(d)) V for one fp32 head with the running row sums l and row maxima m written beside it
 * (m + ln l is the row's log-sum-exp).  Bc is the number of keys per staged K/V tile, as in the reference (whose tests sweep
 * Bc in {1, 2, 4}, main.cu:300-345): 1 <= Bc, values above 64 are clamped to 64 (the kernel's LDS tile); M is accepted and,
 * as in the reference's own wrapper (main.cu:22), not used.  A didactic baseline (scalar fp32, no MFMA): the fast path is flash_attention_2_forward. */
int flash_attention(const float* Q, const float* K, const float* V, float* O, float* l, float* m,
                    int N, int d, int Bc, int M);

/* ======================================================================================
 * Extended entry points: (B, H), dtype, causal mask, stream.
 * ==================================================================================== */

/* O = softmax(scale * Q K^T [+ causal mask]) V ;  L = logsumexp rows.
 * dtype BF16: Q,K,V,O bf16, d in {64,128}.  dtype F32: Q,K,V,O fp32, 1 <= d <= 128.
 * bf16 numerics: a row's softmax reference is set by the first keys it sees (64 at d = 128, 128 at d = 64; lazily, as the
 * reference's running maximum would be) and afterwards only lifted in steps of 64 ln 2 when the row's sum has outgrown 2^30 --
 * in fp32 sums and bf16 probabilities a lagging reference changes nothing but rounding.  A 256-row block one of whose rows
 * would leave the safe range that way (scores jumping by more than ~48 natural units within 256 / 512 consecutive keys) is
 * computed a second time with the running maximum followed key block by key block: same results, twice the time for that
 * block.  NaN inputs give NaN outputs. */
int fa2_forward(const void* Q, const void* K, const void* V, void* O, float* L,
                int B, int H, int seq_len, int head_dim, float softmax_scale,
                int dtype, int causal, void* stream);

/* fp8 (OCP e4m3) forward, BASELINE configs[4]: Q, K, V are e4m3 [B][H][N][128], O is bf16, L fp32.  The
 * kernel reads V through a transposed copy ([B][H][128][N rounded up to 64], e4m3) that this call writes
 * into `workspace` first, followed by the largest key norm of every 64 keys ([B][H][N / 64] fp32: where scale |q| |k|
 * cannot reach a row's rescale threshold the kernel skips that row's running-maximum update -- same results)
 * (fa2_forward_fp8_workspace_bytes).  fa2_forward(..., FA2_DTYPE_FP8_E4M3, ...) is
 * the same call with the workspace taken from the stream-ordered allocator.  No counterpart in the
 * reference (fp32 end to end): parity is against the oracle fed the e4m3-rounded inputs. */
size_t fa2_forward_fp8_workspace_bytes(int B, int H, int seq_len, int head_dim);
int fa2_forward_fp8(const void* Q, const void* K, const void* V, void* O, float* L,
                    int B, int H, int seq_len, int head_dim, float softmax_scale, int causal,
                    void* workspace, size_t workspace_bytes, void* stream);
/* The same for tensors stored as x / descale (per-tensor scaling, what an fp8 producer that uses e4m3's range hands over):
 * the scores are softmax_scale (q_descale Q)(k_descale K)^T and O = P (v_descale V).  fa2_forward_fp8 is this call with the
 * three descales 1 -- i.e. for tensors whose own values fit e4m3 (|x| <= 448); a caller with scaled tensors who uses that
 * entry point must fold q_descale k_descale into softmax_scale and multiply O by v_descale himself.  L is the log-sum-exp
 * of the DESCALED scores either way.  Descales must be positive and finite. */
int fa2_forward_fp8_scaled(const void* Q, const void* K, const void* V, void* O, float* L,
                           int B, int H, int seq_len, int head_dim, float softmax_scale,
                           float q_descale, float k_descale, float v_descale, int causal,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Bytes of scratch fa2_backward needs for this problem: D and the row-constant planes, plus -- for the shapes the
 * single-kernel backward takes (bf16; head_dim 128, or head_dim 64 at aligned lengths: see fa2_backward) -- the fp32 running sums
 * of dQ (B H NP 128 x 4 bytes at either head_dim, NP = seq_len rounded up to a multiple of 256), a small control block and, when
 * NP != seq_len, padded row-constant planes. */
size_t fa2_backward_workspace_bytes(int B, int H, int seq_len, int head_dim, int dtype);

/* dQ, dK, dV from Q, K, V, O, L (forward outputs) and dO.  Deterministic: no floating-point atomics, every
 * gradient element is summed in a fixed order (the reference's smem + global atomicAdd scheme,
 * flash_attention_backward_kernel.cu:208-231, is not reproduced).  Two implementations behind this call:
 *   - bf16, head_dim 128 (causal or not): ONE kernel that forms the five block products once (csrc/fa2_bwd_fused.hip); a
 *     workgroup owns 256 keys (dK, dV in registers) and the dQ tiles are summed key block after key block in a fixed order
 *     through the L2 of the XCD the head is pinned to.  Its loops run on seq_len rounded up to a multiple of 256 (keys past
 *     the end masked, rows past the end given row constants that make P vanish, nothing past the end stored); it is taken
 *     whenever that padding costs less than the two extra products of the other form: 5 roundup(N, 256) <= 7 roundup(N, 64),
 *     i.e. every multiple of 256, every N >= 897 and the lengths just below a multiple of 256 under that;
 *   - bf16, head_dim 64 (causal or not; round 4): the same kernel built for head_dim 64 (a sub-tile's dQ tile is summed in two
 *     key halves, i.e. two running sums per column block), taken when 13 roundup(N, 256) <= 14 roundup(N, 64) -- it is ~10 %
 *     ahead of the other form there, so the padding may cost 7 %: every multiple of 256, the lengths just below, every N >= 3329;
 *   - everything else: a dQ kernel and a dK/dV kernel (seven products, csrc/fa2_bwd_bf16.hip).
 * The environment variable FA2_BACKWARD_PATH=two_kernel keeps every shape on the second form.
 * PLACEMENT ASSUMPTION of the first form: its workgroups read HW_REG_XCC_ID and hand running sums to each other through the
 * L2 of that XCC (plain stores, sc1 loads); validated on gfx950 in SPX mode (one device = 8 XCCs x 32 CUs).  On any other
 * device layout (a partitioned GPU, another architecture) this call runs the second form instead: fa2_backward_plan.
 * A hand-off that times out (every wait is bounded) leaves dQ all NaN and dK / dV complete: fa2_backward_status. */
int fa2_backward(const void* Q, const void* K, const void* V, const void* O, const float* L,
                 const void* dO, void* dQ, void* dK, void* dV,
                 int B, int H, int seq_len, int head_dim, float softmax_scale,
                 int dtype, int causal, void* workspace, size_t workspace_bytes, void* stream);

/* Which implementation fa2_backward runs for this problem ON THE CURRENT DEVICE: 1 = the single five-product kernel,
 * 2 = the dQ + dK/dV kernels (or the fp32 kernels); negative = status.  *reason (may be NULL) receives a static sentence.
 * The single kernel hands running dQ sums from workgroup to workgroup through the L2 of the XCC both run on (they read
 * HW_REG_XCC_ID and take work from that XCC's queue); that was validated on gfx950 exposing all 256 CUs (SPX mode).  On a
 * partitioned GPU or another architecture fa2_backward silently uses the two kernels instead -- this call says so. */
int fa2_backward_plan(int B, int H, int seq_len, int head_dim, int dtype, int causal, const char** reason);

/* Synchronises `stream` and reports how the last fa2_backward / fa2_backward_phases / fa2_backward_fused call on that
 * workspace ended: FA2_OK, or FA2_ERR_HANDOFF_TIMEOUT when a workgroup of the single-kernel backward gave up waiting for
 * the key block before it (every such wait is bounded, 2^22 polls).  CONTRACT: the launch itself returns FA2_OK either way
 * (it is asynchronous); after a time-out EVERY element of dQ is NaN (the output pass poisons it) while dK and dV are
 * complete and correct -- they never depend on the hand-off.  A caller that consumes dK / dV without looking at dQ should
 * call this once per step (it costs a stream synchronisation and a 4-byte copy).  Shapes that run the two kernels have no
 * hand-off: the call only synchronises. */
int fa2_backward_status(const void* workspace, size_t workspace_bytes, int B, int H, int seq_len, int head_dim, int dtype,
                        void* stream);

/* The same, restricted to some of its kernels -- bit 0: D = rowsum(dO o O) and the row constants into the
 * workspace, bit 1: the dQ kernel, bit 2: the dK/dV kernel, bit 3: the single five-product kernel and its output
 * pass (FA2_ERR_UNSUPPORTED for shapes or devices it does not take, and in combination with bits 1 or 2).  7 = fa2_backward (which picks the implementation);
 * 6 = the two-kernel form whatever the shape.  For profiling and for callers that overlap the two independent
 * kernels on separate streams; bits 1, 2 and 3 need what bit 0 produced in the same workspace. */
int fa2_backward_phases(const void* Q, const void* K, const void* V, const void* O, const float* L,
                        const void* dO, void* dQ, void* dK, void* dV,
                        int B, int H, int seq_len, int head_dim, float softmax_scale,
                        int dtype, int causal, void* workspace, size_t workspace_bytes, void* stream,
                        int phases);

/* The backward of a rectangular BLOCK of the score matrix: q_len query rows (a row range of every head,
 * consecutive heads q_head_stride rows apart in Q / O / dO / dQ / L; 0 = q_len) against kv_len keys (K / V /
 * dK / dV, heads kv_head_stride rows apart; 0 = kv_len).  dQ = this block's contribution to the rows' dQ,
 * dK / dV = the rows' contribution to the keys' gradients; L must be the log-sum-exp over ALL keys of the
 * row (not only this block's), which is what makes blocks add up -- the unit of work of the ring backward.
 * With causal != 0 key j is visible to local query i iff j <= i + causal_shift.  The workspace is
 * fa2_backward_workspace_bytes(B, H, q_head_stride ? q_head_stride : q_len, ...): its planes are dense
 * [B][H][q_head_stride] and the block's rows are rows [q_row0, q_row0 + q_len) of every head (the tensor
 * pointers address row q_row0 of head 0), so phase bit 0 (D = rowsum(dO o O)) may be run once over the
 * dense local tensors (q_row0 = 0, q_len = q_head_stride) and reused by every block of them.  bf16 only.
 * fa2_backward is the block q_len = kv_len = seq_len, strides 0, q_row0 0, shift 0.
 * phases as fa2_backward_phases bits 0..2.  With bits 1 and 2 both set (6 or 7) the library picks the implementation as
 * fa2_backward does: a DENSE SQUARE block (q_len = kv_len, dense strides, q_row0 = 0, no shift) of head_dim 128 and a
 * length that is a multiple of 256 runs the single five-product kernel -- provided the workspace is
 * fa2_backward_workspace_bytes(B, H, q_len, ...) (room for its running sums) and the device is the validated layout
 * (fa2_backward_plan) -- and so does an UNMASKED rectangular, head-strided block of head_dim 128 with q_len a multiple of 32
 * (at least 512) and kv_len a multiple of 256 (at most q_head_stride rounded up to 256), given the workspace of
 * fa2_backward_workspace_bytes(B, H, q_head_stride, ...): the zig-zag causal ring's half blocks.  Every other block runs the dQ
 * and dK/dV kernels.  A single bit (2 or 4) always runs that kernel.
 * FA2_PHASE_LEAVE_ROOM (bit 4) asks the single kernel to leave some of the device's CUs free: its persistent workgroups fill a
 * CU's register file for the whole launch, so a kernel on another stream that must run CONCURRENTLY (the ring backward's RCCL
 * exchange of the previous step's dK / dV pieces) would otherwise wait for the launch to end.  How many: bits 8..15 of
 * `phases`, FA2_PHASE_LEAVE_CUS(n) with 1 <= n <= 255; 0 there (plain FA2_PHASE_LEAVE_ROOM) = 16.  A run-time argument, so
 * that a multi-GPU run can tune it without a rebuild (fa2_ring_ctx_set_reserved_cus).  No effect on the results. */
#define FA2_PHASE_LEAVE_ROOM 16
#define FA2_PHASE_LEAVE_CUS(n) (FA2_PHASE_LEAVE_ROOM | (((n) & 0xff) << 8))
int fa2_backward_block(const void* Q, const void* K, const void* V, const void* O, const float* L,
                       const void* dO, void* dQ, void* dK, void* dV,
                       int B, int H, int q_len, int kv_len, int head_dim, float softmax_scale, int dtype,
                       int q_head_stride, int kv_head_stride, int q_row0, int causal, int causal_shift,
                       void* workspace, size_t workspace_bytes, void* stream, int phases);

/* The single-kernel five-product backward (csrc/fa2_bwd_fused.hip) with the way dQ is summed over the key-block
 * workgroups of a head chosen by the caller -- mode 1: handed from key block to key block in a fixed order by a
 * persistent grid (deterministic; what fa2_backward uses), mode 0: fp32 atomics (NOT bit-reproducible; kept as the
 * measured alternative; seq_len a multiple of 256 and d = 128 only).  bf16, d = 128 or (mode 1, aligned lengths) 64, non-causal (fa2_backward also takes the causal case), the seq_len rule of fa2_backward;
 * FA2_ERR_UNSUPPORTED otherwise.  Workspace: fa2_backward_fused_workspace_bytes (= fa2_backward_workspace_bytes). */
size_t fa2_backward_fused_workspace_bytes(int B, int H, int seq_len, int head_dim);
int fa2_backward_fused(const void* Q, const void* K, const void* V, const void* O, const float* L,
                       const void* dO, void* dQ, void* dK, void* dV,
                       int B, int H, int seq_len, int head_dim, float softmax_scale, int mode,
                       void* workspace, size_t workspace_bytes, void* stream);

/* One resumable forward step: folds the kv_len keys/values of a resident shard into the running
 * state of q_len local query rows -- the unit of work of ring_attention_forward_kernel
 * (ring_attention_kernel.cu:13-140).  State between steps: Mrun = running max (natural units),
 * L = running sum l, and the un-normalised accumulator -- for bf16 in Oacc (fp32
 * [B][H][q_len][d]), for fp32 in O itself (Oacc ignored), as the reference keeps it (:125-137).
 * first != 0 starts from (0, 0, -inf) instead of loading the state; last != 0 writes
 * O = acc / l and L = m + ln l (:112-124) instead of storing the state.  Non-causal. */
int fa2_forward_step(const void* Q, const void* K, const void* V,
                     void* O, float* L, float* Oacc, float* M,
                     int B, int H, int q_len, int kv_len, int head_dim, float softmax_scale,
                     int dtype, int first, int last, void* stream);

/* fa2_forward_step (bf16 only) over a ROW RANGE of every head: consecutive heads are q_head_stride rows
 * apart in Q / O / Oacc / L / M and kv_head_stride rows apart in K / V (0 = q_len / kv_len, dense); the
 * pointers address the first row of the range in head 0.  With causal != 0 key j is visible to local
 * query i iff j <= i + causal_shift.  This is the unit of work of the causal zig-zag ring, where a
 * step folds half of the local keys into all local rows, or all of them into half of the rows. */
int fa2_forward_step_strided(const void* Q, const void* K, const void* V,
                             void* O, float* L, float* Oacc, float* M,
                             int B, int H, int q_len, int kv_len, int head_dim, float softmax_scale,
                             int dtype, int first, int last, int q_head_stride, int kv_head_stride,
                             int causal, int causal_shift, void* stream);

/* Turns the bf16 step state of `rows` consecutive rows into results: O = acc / l (bf16), L = m + ln l
 * (on entry L holds l) -- what last != 0 does inside a step, for schedules whose last step does not
 * touch every row. */
int fa2_forward_state_finalize(void* O, float* L, const float* Oacc, const float* M,
                               size_t rows, int head_dim, int dtype, void* stream);

/* acc[i] = (init ? 0 : acc[i]) + src[i] for n bf16 values: fp32 running sums of bf16 contributions (the
 * ring backward adds each step's gradients this way). */
int fa2_accumulate_bf16(float* acc, const void* src, size_t n, int init, void* stream);
/* The same over `rows` runs of `cols` contiguous elements that start `pitch` elements apart in both acc and
 * src (a row range of every head of a [B][H][N][d] tensor). */
int fa2_accumulate_bf16_2d(float* acc, const void* src, size_t rows, size_t cols, size_t pitch, int init, void* stream);

/* Measurement aid: a small kernel on `stream` writes, for every XCC x of the device it reaches (x = HW_REG_XCC_ID < 16), two
 * 64-bit device counters to out32[2 x], out32[2 x + 1] (DEVICE memory, 32 x 8 = 256 bytes, 16-byte aligned, zeroed by the
 * caller: an XCC the device does not have keeps zeros): shader-clock ticks (s_memtime: an XCC's own counter, so only
 * differences taken on the SAME XCC mean anything) and ticks of the constant 100 MHz reference (s_memrealtime).  Bracket a
 * stretch of work with two calls into two buffers: per XCC, d(ticks) / d(reference ticks) x 100 MHz is the mean shader clock
 * it held over the stretch (cuda_flashattention_amd.ops.mean_shader_clock_mhz averages the XCCs present in both). */
int fa2_read_clocks(unsigned long long* out32, void* stream);

/* Measurement aid: `workgroups` workgroups of four waves (one per SIMD) each run 4 x iters back-to-back
 * v_mfma_f32_32x32x16_bf16 on the operands at `operands` (DEVICE memory, 1024 x 16 bytes of bf16 the caller fills -- random
 * values: the clock a device holds depends on the data) and write workgroups x 256 floats to `out`.  4 x iters x workgroups x 4 x
 * 32768 flops: timed by the caller with one workgroup per CU, it is what THIS device sustains on bare MFMAs -- the ceiling
 * bench.py reports beside the nominal peak, because devices of one pool differ by several per cent on power-limited kernels. */
int fa2_mfma_probe(const void* operands, float* out, int iters, int workgroups, void* stream);

/* Element-wise helpers (grid-stride, HBM-bound). */
int fa2_fill_f32(float* dst, size_t n, float value, void* stream);      /* init_array, cuda_helper.h:60-65 */
int fa2_convert_f32_to_bf16(const float* src, void* dst, size_t n, void* stream);
int fa2_convert_bf16_to_f32(const void* src, float* dst, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FA2_MI355X_H */
