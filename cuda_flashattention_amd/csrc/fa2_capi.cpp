// fa2_capi.cpp -- the C ABI of libfa2_mi355x.so (include/fa2_mi355x.h): argument checking,
// status codes and dispatch.  No kernels here and no framework types; the host wrappers of
// the reference (flash_attention_kernel.cu:300-343, flash_attention_backward_kernel.cu:249-299)
// become these functions.
#include "../../include/fa2_mi355x.h"
#include <cstdlib>
#include <cstring>

#include "fa2_launch.h"

#include <hip/hip_runtime.h>
#include <mutex>

namespace {

inline int hip_status(hipError_t e) { return e == hipSuccess ? FA2_OK : FA2_ERR_HIP_BASE - (int)e; }

// The kernels address one head slab through a buffer resource (32-bit byte count, 32-bit byte offsets): a slab of N
// rows must stay below 2 GiB (at 4 bytes per element, the widest type).  Larger problems get a status, not a silent
// wrap-around.
inline int check_common(int B, int H, int N, int d, float scale)
{
    if (B <= 0 || H <= 0 || N <= 0 || d <= 0) return FA2_ERR_INVALID_SHAPE;
    if (!(scale > 0.0f)) return FA2_ERR_INVALID_SHAPE;
    if ((long long)B * H > 0x7fffffffLL / 64) return FA2_ERR_INVALID_SHAPE;
    if ((long long)N * d * 4 > 0x7fffffffLL) return FA2_ERR_INVALID_SHAPE;
    return FA2_OK;
}

// bf16 backward only: its two row-constant planes (B H rows floats each) sit behind ONE buffer resource.  The single-kernel
// form builds that resource (and its int offsets) on the length PADDED to a multiple of 256 (fa2_bwd_fused.hip: rc_rsrc,
// rcoff), so the padded length is what must fit -- for every shape: the 255 rows of slack cost nobody a legitimate problem.
inline int check_bwd_planes(int B, int H, int rows)
{
    const long long padded = ((long long)rows + 255) / 256 * 256;
    return (long long)B * H * padded * 8 > 0x7fffffffLL ? FA2_ERR_INVALID_SHAPE : FA2_OK;
}

inline int check_dim(int d, int dtype)
{
    if (dtype == FA2_DTYPE_BF16) return (d == 64 || d == 128) ? FA2_OK : FA2_ERR_UNSUPPORTED_HEAD_DIM;
    if (dtype == FA2_DTYPE_F32) return (d >= 1 && d <= 128) ? FA2_OK : FA2_ERR_UNSUPPORTED_HEAD_DIM;
    if (dtype == FA2_DTYPE_FP8_E4M3) return d == 128 ? FA2_OK : FA2_ERR_UNSUPPORTED_HEAD_DIM;
    return FA2_ERR_UNSUPPORTED_DTYPE;
}

// Grow-only per-device scratch for the reference-signature backward, which has no workspace
// argument (the reference cudaMemsets inside its wrapper too, :282-283).
struct ScratchCache {
    std::mutex mu;
    void* ptr[64] = {};
    size_t cap[64] = {};
    int get(size_t bytes, void** out)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return hip_status(e);
        if (dev < 0 || dev >= 64) return FA2_ERR_UNSUPPORTED;
        std::lock_guard<std::mutex> g(mu);
        if (cap[dev] < bytes) {
            if (ptr[dev]) {
                e = hipDeviceSynchronize();      // nothing may still be using the old block
                if (e != hipSuccess) return hip_status(e);
                (void)hipFree(ptr[dev]);
                ptr[dev] = nullptr; cap[dev] = 0;
            }
            e = hipMalloc(&ptr[dev], bytes);
            if (e != hipSuccess) return hip_status(e);
            cap[dev] = bytes;
        }
        *out = ptr[dev];
        return FA2_OK;
    }
} g_scratch;

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

// The compiler that produced the code objects is part of the version: the kernels pin registers and count
// wait states by hand, and tests/test_isa_guard.py re-checks the generated code whenever this string changes.
#define FA2_STR2(x) #x
#define FA2_STR(x) FA2_STR2(x)
const char* fa2_version(void)
{
    return "fa2_mi355x 0.2 (gfx950; hip " FA2_STR(HIP_VERSION_MAJOR) "." FA2_STR(HIP_VERSION_MINOR) "." FA2_STR(HIP_VERSION_PATCH)
           "; clang " __clang_version__ ")";
}

const char* fa2_status_string(int s)
{
    switch (s) {
    case FA2_OK: return "ok";
    case FA2_ERR_NULL_POINTER: return "null pointer";
    case FA2_ERR_INVALID_SHAPE: return "invalid shape or scale";
    case FA2_ERR_UNSUPPORTED_HEAD_DIM: return "unsupported head_dim";
    case FA2_ERR_UNSUPPORTED_DTYPE: return "unsupported dtype";
    case FA2_ERR_WORKSPACE: return "workspace missing or too small";
    case FA2_ERR_UNSUPPORTED: return "unsupported combination";
    case FA2_ERR_HANDOFF_TIMEOUT: return "single-kernel backward: a bounded wait for the previous key block ran out (dQ is NaN)";
    default: break;
    }
    if (s <= FA2_ERR_RCCL_BASE) return "RCCL error (code = -(status) - 2000)";
    if (s <= FA2_ERR_HIP_BASE) return hipGetErrorString((hipError_t)(FA2_ERR_HIP_BASE - s));
    return "unknown status";
}

int fa2_forward(const void* Q, const void* K, const void* V, void* O, float* L,
                int B, int H, int seq_len, int head_dim, float softmax_scale,
                int dtype, int causal, void* stream)
{
    if (!Q || !K || !V || !O || !L) return FA2_ERR_NULL_POINTER;
    int st = check_common(B, H, seq_len, head_dim, softmax_scale);
    if (st) return st;
    st = check_dim(head_dim, dtype);
    if (st) return st;
    if (dtype == FA2_DTYPE_BF16) {
        fa2::FwdArgs a{};
        a.Q = Q; a.K = K; a.V = V; a.O = O; a.L = L; a.Oacc = nullptr; a.M = nullptr;
        a.BH = B * H; a.Nq = seq_len; a.Nk = seq_len; a.d = head_dim; a.scale = softmax_scale;
        a.causal = causal ? 1 : 0; a.causal_shift = 0; a.resume = 0; a.finalize = 1;
        return hip_status(fa2::launch_fwd1_bf16(a, (hipStream_t)stream));
    }
    if (dtype == FA2_DTYPE_FP8_E4M3) {      // workspace from the stream-ordered allocator
        const size_t need = fa2_forward_fp8_workspace_bytes(B, H, seq_len, head_dim);
        void* ws = nullptr;
        hipError_t e = hipMallocAsync(&ws, need, (hipStream_t)stream);
        if (e != hipSuccess) return hip_status(e);
        st = fa2_forward_fp8(Q, K, V, O, L, B, H, seq_len, head_dim, softmax_scale, causal, ws, need, stream);
        e = hipFreeAsync(ws, (hipStream_t)stream);
        return st ? st : hip_status(e);
    }
    fa2::F32Args a{};
    a.Q = (const float*)Q; a.K = (const float*)K; a.V = (const float*)V; a.O = (float*)O; a.L = L;
    a.BH = B * H; a.N = seq_len; a.d = head_dim; a.scale = softmax_scale; a.causal = causal ? 1 : 0;
    a.Nk = seq_len; a.M = nullptr; a.resume = 0; a.finalize = 1;
    return hip_status(fa2::launch_fwd_f32(a, (hipStream_t)stream));
}

size_t fa2_forward_fp8_workspace_bytes(int B, int H, int seq_len, int head_dim)
{
    if (B <= 0 || H <= 0 || seq_len <= 0 || head_dim != 128) return 0;
    const size_t npad = ((size_t)seq_len + 63) / 64 * 64;
    return (size_t)B * H * head_dim * npad + (size_t)B * H * (npad / 64) * sizeof(float);      // V^T | key-norm maxima (npad % 64 == 0: aligned)
}

int fa2_forward_fp8(const void* Q, const void* K, const void* V, void* O, float* L,
                    int B, int H, int seq_len, int head_dim, float softmax_scale, int causal,
                    void* workspace, size_t workspace_bytes, void* stream)
{
    return fa2_forward_fp8_scaled(Q, K, V, O, L, B, H, seq_len, head_dim, softmax_scale, 1.0f, 1.0f, 1.0f, causal, workspace,
                                  workspace_bytes, stream);
}

int fa2_forward_fp8_scaled(const void* Q, const void* K, const void* V, void* O, float* L,
                           int B, int H, int seq_len, int head_dim, float softmax_scale,
                           float q_descale, float k_descale, float v_descale, int causal,
                           void* workspace, size_t workspace_bytes, void* stream)
{
    if (!Q || !K || !V || !O || !L) return FA2_ERR_NULL_POINTER;
    if (!(q_descale > 0.0f) || !(k_descale > 0.0f) || !(v_descale > 0.0f)) return FA2_ERR_INVALID_SHAPE;
    int st = check_common(B, H, seq_len, head_dim, softmax_scale);
    if (st) return st;
    // scores are formed from the STORED values: the two descales belong to the softmax scale (the lazy reference, its
    // thresholds and the key-norm bound are all in units of that product); V's multiplies O once, in the epilogue
    softmax_scale *= q_descale * k_descale;
    if (!(softmax_scale > 0.0f) || !(softmax_scale < 3.0e38f)) return FA2_ERR_INVALID_SHAPE;
    st = check_dim(head_dim, FA2_DTYPE_FP8_E4M3);
    if (st) return st;
    if (!workspace || workspace_bytes < fa2_forward_fp8_workspace_bytes(B, H, seq_len, head_dim)) return FA2_ERR_WORKSPACE;
    fa2::FwdFp8Args a{};
    a.Q = Q; a.K = K; a.V = V; a.Vt = workspace; a.O = O; a.L = L;
    a.BH = B * H; a.N = seq_len; a.Npad = (seq_len + 63) / 64 * 64; a.d = head_dim;
    a.kn = reinterpret_cast<float*>((char*)workspace + (size_t)a.BH * head_dim * a.Npad);
    a.scale = softmax_scale; a.causal = causal ? 1 : 0; a.o_scale = v_descale;
    return hip_status(fa2::launch_fwd_fp8(a, (hipStream_t)stream));
}

// D = rowsum(dO o O), then the two row-constant planes (-L/scale, -D) of the kernels that own key blocks
static size_t bwd_base_ws(int B, int H, int rows) { return 3 * align256((size_t)B * H * rows * sizeof(float)); }

// shapes the single-kernel (five-product) backward takes: csrc/fa2_bwd_fused.hip.  Any seq_len: its loops run on the length
// rounded up to a multiple of 256 (keys past the end are masked, rows past the end get row constants that make P vanish), so
// it is taken whenever that padding costs less than the two extra block products of the two-kernel form, whose own tiles are
// 64 keys: 5 x roundup(N, 256) <= 7 x roundup(N, 64) -- every N >= 897, and the N just below a multiple of 256 under that.
static int fused_npad(int n) { return (n + 255) / 256 * 256; }
static bool bwd_fused_shape(int seq_len, int head_dim, int dtype)
{
    if (dtype != FA2_DTYPE_BF16 || seq_len < 1) return false;
    const long long np = fused_npad(seq_len), n64 = (seq_len + 63) / 64 * 64;
    // head_dim 64 (round 4): the single kernel is ~10 % ahead of the two kernels there (the same VALU and hand-off work beside half
    // the MFMAs), so padding to the key block may cost 7 % at most: every multiple of 256, the lengths just below one, every N >= 3329
    if (head_dim == 64) return 13 * np <= 14 * n64 && np * 128 * 4 <= 0x7fffffffLL;
    if (head_dim != 128) return false;
    return 5 * np <= 7 * n64 && np * head_dim * 4 <= 0x7fffffffLL;
}
// workspace behind the D / row-constant planes: fp32 dQ sums [BH][NP][d] | control block | (ragged only) padded row constants
struct FusedWs { float* acc; int* ctl; float* rcpad; size_t bytes; };
static FusedWs fused_ws(void* base, int B, int H, int seq_len, int head_dim)
{
    const int np = fused_npad(seq_len);
    (void)head_dim;      // the running sums are [np / 32 sub-tiles][4 waves][32 x 32] fp32 at either head_dim: 128 floats per row
    const size_t a = align256((size_t)B * H * np * 128 * 4), c = align256(fa2::bwd_fused_ctl_bytes(B * H, np));
    const size_t r = np != seq_len ? align256((size_t)2 * B * H * np * sizeof(float)) : 0;
    char* b = (char*)base;
    return FusedWs{(float*)b, (int*)(b + a), r ? (float*)(b + a + c) : nullptr, a + c + r};
}
static size_t bwd_fused_ws(int B, int H, int seq_len, int head_dim) { return fused_ws(nullptr, B, H, seq_len, head_dim).bytes; }

// FA2_BACKWARD_PATH=two_kernel keeps fa2_backward on the two deterministic kernels for every shape (A/B runs, triage)
static bool bwd_fused_allowed()
{
    static const bool allowed = [] {
        const char* e = getenv("FA2_BACKWARD_PATH");
        return !(e && strcmp(e, "two_kernel") == 0);
    }();
    return allowed;
}

size_t fa2_backward_workspace_bytes(int B, int H, int seq_len, int head_dim, int dtype)
{
    if (B <= 0 || H <= 0 || seq_len <= 0) return 0;
    return bwd_base_ws(B, H, seq_len) + (bwd_fused_shape(seq_len, head_dim, dtype) ? bwd_fused_ws(B, H, seq_len, head_dim) : 0);
}

static int backward_block_impl(const void* Q, const void* K, const void* V, const void* O, const float* L,
                               const void* dO, void* dQ, void* dK, void* dV,
                               int B, int H, int q_len, int kv_len, int head_dim, float softmax_scale, int dtype,
                               int q_head_stride, int kv_head_stride, int q_row0, int causal, int causal_shift,
                               void* workspace, size_t workspace_bytes, void* stream, int phases, bool allow_single);

int fa2_backward(const void* Q, const void* K, const void* V, const void* O, const float* L,
                 const void* dO, void* dQ, void* dK, void* dV,
                 int B, int H, int seq_len, int head_dim, float softmax_scale,
                 int dtype, int causal, void* workspace, size_t workspace_bytes, void* stream)
{
    return fa2_backward_phases(Q, K, V, O, L, dO, dQ, dK, dV, B, H, seq_len, head_dim, softmax_scale,
                               dtype, causal, workspace, workspace_bytes, stream, 7);
}

int fa2_backward_phases(const void* Q, const void* K, const void* V, const void* O, const float* L,
                        const void* dO, void* dQ, void* dK, void* dV,
                        int B, int H, int seq_len, int head_dim, float softmax_scale,
                        int dtype, int causal, void* workspace, size_t workspace_bytes, void* stream,
                        int phases)
{
    if (!Q || !K || !V || !O || !L || !dO || !dQ || !dK || !dV) return FA2_ERR_NULL_POINTER;
    int st = check_common(B, H, seq_len, head_dim, softmax_scale);
    if (st) return st;
    if (dtype == FA2_DTYPE_FP8_E4M3) return FA2_ERR_UNSUPPORTED_DTYPE;      // fp8 is forward only
    st = check_dim(head_dim, dtype);
    if (st) return st;
    if (!workspace || workspace_bytes < fa2_backward_workspace_bytes(B, H, seq_len, head_dim, dtype))
        return FA2_ERR_WORKSPACE;
    if (dtype == FA2_DTYPE_BF16) {
        st = check_bwd_planes(B, H, seq_len);
        if (st) return st;
        // phases: 1 = D and the row constants, 2 = dQ kernel, 4 = dK/dV kernel, 8 = the single five-product kernel + its
        // output pass.  7 ("all of it") takes the single kernel where the shape and the device allow (d = 128,
        // seq_len % 256 == 0, gfx950 with all 256 CUs); bit 3 does not combine with bits 1 and 2 (they are two ways of
        // computing the same outputs).
        const bool fused_ok = bwd_fused_shape(seq_len, head_dim, dtype);
        if ((phases & 8) && (phases & 6)) return FA2_ERR_UNSUPPORTED;
        if ((phases & 8) && (!fused_ok || !fa2::bwd_fused_device_ok(nullptr))) return FA2_ERR_UNSUPPORTED;
        if ((phases & 8) || (phases == 7 && fused_ok && bwd_fused_allowed() && fa2::bwd_fused_device_ok(nullptr))) {
            fa2::BwdArgs a{};
            a.Q = Q; a.K = K; a.V = V; a.O = O; a.dO = dO; a.L = L; a.dQ = dQ; a.dK = dK; a.dV = dV;
            a.D = (float*)workspace; a.BH = B * H; a.Nq = seq_len; a.Nk = seq_len; a.d = head_dim;
            a.RC = (float*)((char*)workspace + align256((size_t)B * H * seq_len * sizeof(float)));
            a.q_hs = seq_len; a.k_hs = seq_len; a.q_row0 = 0; a.scale = softmax_scale; a.causal = causal ? 1 : 0; a.causal_shift = 0;
            a.phases = phases == 7 ? 9 : (phases & 9);
            const FusedWs w = fused_ws((char*)workspace + bwd_base_ws(B, H, seq_len), B, H, seq_len, head_dim);
            return hip_status(fa2::launch_bwd_fused_bf16(a, w.acc, w.ctl, 1, (hipStream_t)stream, w.rcpad));
        }
        if (fused_ok) {
            // the workspace has a control block whose error word fa2_backward_status reads: it must describe THIS call, also
            // when the two kernels run it (FA2_BACKWARD_PATH, a device that is not the validated layout, phases 2 | 4)
            const FusedWs w = fused_ws((char*)workspace + bwd_base_ws(B, H, seq_len), B, H, seq_len, head_dim);
            const hipError_t e = fa2::bwd_fused_clear_error(w.ctl, (hipStream_t)stream);
            if (e != hipSuccess) return hip_status(e);
        }
        return backward_block_impl(Q, K, V, O, L, dO, dQ, dK, dV, B, H, seq_len, seq_len, head_dim, softmax_scale, dtype, 0, 0, 0,
                                   causal, 0, workspace, workspace_bytes, stream, phases, false);     // phases 6 here = the two kernels
    }
    fa2::F32Args a{};
    a.Q = (const float*)Q; a.K = (const float*)K; a.V = (const float*)V; a.O = (float*)O;
    a.L = (float*)L; a.dO = (const float*)dO; a.dQ = (float*)dQ; a.dK = (float*)dK; a.dV = (float*)dV;
    a.D = (float*)workspace; a.BH = B * H; a.N = seq_len; a.d = head_dim;
    a.scale = softmax_scale; a.causal = causal ? 1 : 0; a.phases = phases & 7;
    return hip_status(fa2::launch_bwd_f32(a, (hipStream_t)stream));
}

int fa2_backward_plan(int B, int H, int seq_len, int head_dim, int dtype, int causal, const char** reason)
{
    (void)causal;
    static const char* const kShape = "two kernels: the single kernel takes bf16 with head_dim 128 and a seq_len whose padding to a "
                                      "multiple of 256 costs less than two block products (5 roundup(N,256) <= 7 roundup(N,64)), or "
                                      "head_dim 64 and a seq_len whose padding costs less than 7 % (13 roundup(N,256) <= 14 roundup(N,64))";
    static const char* const kEnv = "two kernels: FA2_BACKWARD_PATH=two_kernel";
    static const char* const kF32 = "fp32 path (exact f32 MFMA kernels)";
    if (reason) *reason = "";
    if (B <= 0 || H <= 0 || seq_len <= 0) return FA2_ERR_INVALID_SHAPE;
    if (dtype == FA2_DTYPE_FP8_E4M3) return FA2_ERR_UNSUPPORTED_DTYPE;
    int st = check_dim(head_dim, dtype);
    if (st) return st;
    if (dtype == FA2_DTYPE_F32) { if (reason) *reason = kF32; return 2; }
    if (!bwd_fused_shape(seq_len, head_dim, dtype)) { if (reason) *reason = kShape; return 2; }
    if (!bwd_fused_allowed()) { if (reason) *reason = kEnv; return 2; }
    const char* why = "";
    const bool ok = fa2::bwd_fused_device_ok(&why);
    if (reason) *reason = why;
    return ok ? 1 : 2;
}

int fa2_backward_status(const void* workspace, size_t workspace_bytes, int B, int H, int seq_len, int head_dim, int dtype,
                        void* stream)
{
    if (!workspace) return FA2_ERR_NULL_POINTER;
    if (B <= 0 || H <= 0 || seq_len <= 0) return FA2_ERR_INVALID_SHAPE;
    // no hand-off, nothing that can time out: only drain the stream.  (A workspace of a single-kernel shape always carries an
    // error word written by the LAST backward on it -- every launch path clears or sets it -- but where this process never
    // runs the single kernel, environment or device, there is no reason to trust what the caller's buffer holds.)
    if (!bwd_fused_shape(seq_len, head_dim, dtype) || !bwd_fused_allowed() || !fa2::bwd_fused_device_ok(nullptr)) {
        return hip_status(hipStreamSynchronize((hipStream_t)stream));
    }
    if (workspace_bytes < fa2_backward_workspace_bytes(B, H, seq_len, head_dim, dtype)) return FA2_ERR_WORKSPACE;
    const FusedWs w = fused_ws((char*)const_cast<void*>(workspace) + bwd_base_ws(B, H, seq_len), B, H, seq_len, head_dim);
    int err = 0;
    hipError_t e = fa2::bwd_fused_read_error(w.ctl, &err, (hipStream_t)stream);
    if (e != hipSuccess) return hip_status(e);
    return err ? FA2_ERR_HANDOFF_TIMEOUT : FA2_OK;
}

static int backward_block_impl(const void* Q, const void* K, const void* V, const void* O, const float* L,
                       const void* dO, void* dQ, void* dK, void* dV,
                       int B, int H, int q_len, int kv_len, int head_dim, float softmax_scale, int dtype,
                       int q_head_stride, int kv_head_stride, int q_row0, int causal, int causal_shift,
                       void* workspace, size_t workspace_bytes, void* stream, int phases, bool allow_single)
{
    if (!Q || !K || !V || !O || !L || !dO || !dQ || !dK || !dV) return FA2_ERR_NULL_POINTER;
    const int q_hs = q_head_stride ? q_head_stride : q_len, k_hs = kv_head_stride ? kv_head_stride : kv_len;
    int st = check_common(B, H, q_len, head_dim, softmax_scale);
    if (!st) st = check_common(B, H, kv_len > 0 ? kv_len : 1, head_dim, softmax_scale);
    if (!st) st = check_common(B, H, q_hs > 0 ? q_hs : 1, head_dim, softmax_scale);
    if (st) return st;
    if (kv_len <= 0 || q_row0 < 0 || q_hs < q_row0 + q_len || k_hs < kv_len) return FA2_ERR_INVALID_SHAPE;
    st = check_bwd_planes(B, H, q_hs);
    if (st) return st;
    if (dtype != FA2_DTYPE_BF16) return FA2_ERR_UNSUPPORTED_DTYPE;
    st = check_dim(head_dim, dtype);
    if (st) return st;
    if (!workspace || workspace_bytes < bwd_base_ws(B, H, q_hs)) return FA2_ERR_WORKSPACE;
    // A dense square block (the ring backward's unit whenever the local length is a multiple of 256: every non-causal step,
    // and the local causal block) IS a problem the single five-product kernel takes -- L being the log-sum-exp over more
    // keys than the block's changes nothing for it.  Taken when both main kernels are asked for at once (phases 6 or 7),
    // the workspace has room for its running sums and the device is the validated layout; otherwise the two kernels.
    // Round 4: so is an UNMASKED rectangular, head-strided block whose lengths are aligned (q_len a multiple of 32 and at least
    // 512, kv_len a multiple of 256) -- the other two block shapes of the zig-zag causal ring.  The workspace layout is the
    // square problem's for q_hs rows (planes, running sums, control block -- fa2_backward_status finds its word in one place),
    // so kv_len must not exceed what its control block was sized for.
    const bool square = q_len == kv_len && q_hs == q_len && k_hs == kv_len && q_row0 == 0 && (!causal || causal_shift == 0) &&
                        bwd_fused_shape(q_len, head_dim, dtype);
    const bool rect = !square && !causal && head_dim == 128 && q_len % 32 == 0 && q_len >= 512 && kv_len % 256 == 0 &&
                      kv_len <= fused_npad(q_hs) && bwd_fused_shape(q_hs, head_dim, dtype);
    if (allow_single && (phases & 6) == 6 && (square || rect) && bwd_fused_allowed() &&
        workspace_bytes >= fa2_backward_workspace_bytes(B, H, q_hs, head_dim, dtype) && fa2::bwd_fused_device_ok(nullptr)) {
        fa2::BwdArgs f{};
        f.Q = Q; f.K = K; f.V = V; f.O = O; f.dO = dO; f.L = L; f.dQ = dQ; f.dK = dK; f.dV = dV;
        f.D = (float*)workspace; f.BH = B * H; f.Nq = q_len; f.Nk = kv_len; f.d = head_dim;
        f.RC = (float*)((char*)workspace + align256((size_t)B * H * q_hs * sizeof(float)));
        f.q_hs = q_hs; f.k_hs = k_hs; f.q_row0 = q_row0; f.scale = softmax_scale; f.causal = causal ? 1 : 0; f.causal_shift = 0;
        f.phases = 8 | (phases & 1);
        // bits 8..15: how many CUs to leave (FA2_PHASE_LEAVE_CUS(n)); 0 there = the default of 16
        f.reserve_cus = (phases & FA2_PHASE_LEAVE_ROOM) ? (((phases >> 8) & 0xff) ? ((phases >> 8) & 0xff) : 16) : 0;
        const FusedWs w = fused_ws((char*)workspace + bwd_base_ws(B, H, q_hs), B, H, q_hs, head_dim);
        return hip_status(fa2::launch_bwd_fused_bf16(f, w.acc, w.ctl, 1, (hipStream_t)stream, w.rcpad));
    }
    if (allow_single && bwd_fused_shape(q_hs, head_dim, dtype) &&
        workspace_bytes >= fa2_backward_workspace_bytes(B, H, q_hs, head_dim, dtype)) {
        // the two kernels on a workspace that carries a control block: its error word must describe this call (fa2_backward_status)
        const FusedWs w = fused_ws((char*)workspace + bwd_base_ws(B, H, q_hs), B, H, q_hs, head_dim);
        const hipError_t e = fa2::bwd_fused_clear_error(w.ctl, (hipStream_t)stream);
        if (e != hipSuccess) return hip_status(e);
    }
    fa2::BwdArgs a{};
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.dO = dO; a.L = L; a.dQ = dQ; a.dK = dK; a.dV = dV;
    a.D = (float*)workspace; a.BH = B * H; a.Nq = q_len; a.Nk = kv_len; a.d = head_dim;
    a.RC = (float*)((char*)workspace + align256((size_t)B * H * q_hs * sizeof(float)));
    a.q_hs = q_hs; a.k_hs = k_hs; a.q_row0 = q_row0;
    a.scale = softmax_scale; a.causal = causal ? 1 : 0; a.causal_shift = causal ? causal_shift : 0; a.phases = phases & 7;
    return hip_status(fa2::launch_bwd_bf16(a, (hipStream_t)stream));
}

int fa2_backward_block(const void* Q, const void* K, const void* V, const void* O, const float* L,
                       const void* dO, void* dQ, void* dK, void* dV,
                       int B, int H, int q_len, int kv_len, int head_dim, float softmax_scale, int dtype,
                       int q_head_stride, int kv_head_stride, int q_row0, int causal, int causal_shift,
                       void* workspace, size_t workspace_bytes, void* stream, int phases)
{
    return backward_block_impl(Q, K, V, O, L, dO, dQ, dK, dV, B, H, q_len, kv_len, head_dim, softmax_scale, dtype, q_head_stride,
                               kv_head_stride, q_row0, causal, causal_shift, workspace, workspace_bytes, stream, phases, true);
}

size_t fa2_backward_fused_workspace_bytes(int B, int H, int seq_len, int head_dim)
{
    if (B <= 0 || H <= 0 || seq_len <= 0 || !bwd_fused_shape(seq_len, head_dim, FA2_DTYPE_BF16)) return 0;
    return fa2_backward_workspace_bytes(B, H, seq_len, head_dim, FA2_DTYPE_BF16);
}

int fa2_backward_fused(const void* Q, const void* K, const void* V, const void* O, const float* L,
                       const void* dO, void* dQ, void* dK, void* dV,
                       int B, int H, int seq_len, int head_dim, float softmax_scale, int mode,
                       void* workspace, size_t workspace_bytes, void* stream)
{
    if (!Q || !K || !V || !O || !L || !dO || !dQ || !dK || !dV) return FA2_ERR_NULL_POINTER;
    int st = check_common(B, H, seq_len, head_dim, softmax_scale);
    if (st) return st;
    if (!st) st = check_bwd_planes(B, H, seq_len);
    if (st) return st;
    if (!bwd_fused_shape(seq_len, head_dim, FA2_DTYPE_BF16) || (mode != 0 && mode != 1)) return FA2_ERR_UNSUPPORTED;
    if (mode == 0 && (seq_len % 256 != 0 || head_dim != 128)) return FA2_ERR_UNSUPPORTED;      // the atomics form: head_dim 128, aligned
    if (mode == 1 && !fa2::bwd_fused_device_ok(nullptr)) return FA2_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < fa2_backward_fused_workspace_bytes(B, H, seq_len, head_dim)) return FA2_ERR_WORKSPACE;
    fa2::BwdArgs a{};
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.dO = dO; a.L = L; a.dQ = dQ; a.dK = dK; a.dV = dV;
    a.D = (float*)workspace; a.BH = B * H; a.Nq = seq_len; a.Nk = seq_len; a.d = head_dim;
    a.RC = (float*)((char*)workspace + align256((size_t)B * H * seq_len * sizeof(float)));
    a.q_hs = seq_len; a.k_hs = seq_len; a.q_row0 = 0; a.scale = softmax_scale; a.causal = 0; a.causal_shift = 0; a.phases = 9;
    const FusedWs w = fused_ws((char*)workspace + bwd_base_ws(B, H, seq_len), B, H, seq_len, head_dim);
    return hip_status(fa2::launch_bwd_fused_bf16(a, w.acc, w.ctl, mode, (hipStream_t)stream, w.rcpad));
}

int fa2_forward_step(const void* Q, const void* K, const void* V,
                     void* O, float* L, float* Oacc, float* M,
                     int B, int H, int q_len, int kv_len, int head_dim, float softmax_scale,
                     int dtype, int first, int last, void* stream)
{
    if (!Q || !K || !V || !L) return FA2_ERR_NULL_POINTER;
    int st = check_common(B, H, q_len, head_dim, softmax_scale);
    if (st) return st;
    if (kv_len <= 0) return FA2_ERR_INVALID_SHAPE;
    if (dtype == FA2_DTYPE_FP8_E4M3) return FA2_ERR_UNSUPPORTED_DTYPE;      // no resumable fp8 step
    st = check_dim(head_dim, dtype);
    if (st) return st;
    if (dtype == FA2_DTYPE_F32) {
        // fp32: O itself carries the un-normalised accumulator between steps (the reference's
        // layout); Oacc is ignored.
        if (!O) return FA2_ERR_NULL_POINTER;
        if ((!first || !last) && !M) return FA2_ERR_NULL_POINTER;
        fa2::F32Args a{};
        a.Q = (const float*)Q; a.K = (const float*)K; a.V = (const float*)V; a.O = (float*)O; a.L = L;
        a.BH = B * H; a.N = q_len; a.Nk = kv_len; a.d = head_dim; a.scale = softmax_scale; a.causal = 0;
        a.M = M; a.resume = first ? 0 : 1; a.finalize = last ? 1 : 0;
        return hip_status(fa2::launch_fwd_f32(a, (hipStream_t)stream));
    }
    return fa2_forward_step_strided(Q, K, V, O, L, Oacc, M, B, H, q_len, kv_len, head_dim, softmax_scale, dtype, first,
                                    last, 0, 0, 0, 0, stream);
}

int fa2_forward_step_strided(const void* Q, const void* K, const void* V,
                             void* O, float* L, float* Oacc, float* M,
                             int B, int H, int q_len, int kv_len, int head_dim, float softmax_scale,
                             int dtype, int first, int last, int q_head_stride, int kv_head_stride,
                             int causal, int causal_shift, void* stream)
{
    if (!Q || !K || !V || !L) return FA2_ERR_NULL_POINTER;
    int st = check_common(B, H, q_len, head_dim, softmax_scale);
    if (st) return st;
    if (kv_len <= 0) return FA2_ERR_INVALID_SHAPE;
    if (dtype != FA2_DTYPE_BF16) return FA2_ERR_UNSUPPORTED_DTYPE;
    st = check_dim(head_dim, dtype);
    if (st) return st;
    if ((q_head_stride && q_head_stride < q_len) || (kv_head_stride && kv_head_stride < kv_len)) return FA2_ERR_INVALID_SHAPE;
    if (last && !O) return FA2_ERR_NULL_POINTER;
    if ((!first || !last) && (!Oacc || !M)) return FA2_ERR_NULL_POINTER;
    fa2::FwdArgs a{};
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.L = L; a.Oacc = Oacc; a.M = M;
    a.BH = B * H; a.Nq = q_len; a.Nk = kv_len; a.d = head_dim; a.scale = softmax_scale;
    a.causal = causal ? 1 : 0; a.causal_shift = causal ? causal_shift : 0;
    a.resume = first ? 0 : 1; a.finalize = last ? 1 : 0;
    a.q_hs = q_head_stride; a.k_hs = kv_head_stride;
    return hip_status(fa2::launch_fwd1_bf16(a, (hipStream_t)stream));
}

int fa2_forward_state_finalize(void* O, float* L, const float* Oacc, const float* M,
                               size_t rows, int head_dim, int dtype, void* stream)
{
    if (!O || !L || !Oacc || !M) return FA2_ERR_NULL_POINTER;
    if (dtype != FA2_DTYPE_BF16) return FA2_ERR_UNSUPPORTED_DTYPE;
    int st = check_dim(head_dim, dtype);
    if (st) return st;
    return hip_status(fa2::launch_finalize_state(Oacc, M, L, O, rows, head_dim, (hipStream_t)stream));
}

int flash_attention_2_forward(const float* Q, const float* K, const float* V,
                              float* O, float* L, int seq_len, int head_dim, float softmax_scale)
{
    return fa2_forward(Q, K, V, O, L, 1, 1, seq_len, head_dim, softmax_scale, FA2_DTYPE_F32, 0, nullptr);
}

int flash_attention(const float* Q, const float* K, const float* V, float* O, float* l, float* m, int N, int d, int Bc, int M)
{
    (void)M;                 // the reference sizes nothing from M either (main.cu:22: "we set Bc directly")
    if (!Q || !K || !V || !O || !l || !m) return FA2_ERR_NULL_POINTER;
    if (N <= 0 || d <= 0 || Bc <= 0) return FA2_ERR_INVALID_SHAPE;
    if (d > 128) return FA2_ERR_UNSUPPORTED_HEAD_DIM;
    return hip_status(fa2::launch_fa1_f32(Q, K, V, O, l, m, N, d, Bc, nullptr));
}

int flash_attention_2_backward(const float* Q, const float* K, const float* V,
                               const float* O, const float* L, const float* dO,
                               float* dQ, float* dK, float* dV,
                               int seq_len, int head_dim, float softmax_scale)
{
    const size_t need = fa2_backward_workspace_bytes(1, 1, seq_len, head_dim, FA2_DTYPE_F32);
    if (need == 0) return FA2_ERR_INVALID_SHAPE;
    void* ws = nullptr;
    int st = g_scratch.get(need, &ws);
    if (st) return st;
    return fa2_backward(Q, K, V, O, L, dO, dQ, dK, dV, 1, 1, seq_len, head_dim, softmax_scale,
                        FA2_DTYPE_F32, 0, ws, need, nullptr);
}

int fa2_accumulate_bf16(float* acc, const void* src, size_t n, int init, void* stream)
{
    return fa2_accumulate_bf16_2d(acc, src, 1, n, n, init, stream);
}

int fa2_accumulate_bf16_2d(float* acc, const void* src, size_t rows, size_t cols, size_t pitch, int init, void* stream)
{
    if (!acc || !src) return FA2_ERR_NULL_POINTER;
    if (pitch < cols && rows > 1) return FA2_ERR_INVALID_SHAPE;
    return hip_status(fa2::launch_accumulate_bf16(acc, src, rows, cols, pitch, init, (hipStream_t)stream));
}

int fa2_read_clocks(unsigned long long* out32, void* stream)
{
    if (!out32) return FA2_ERR_NULL_POINTER;
    if ((uintptr_t)out32 & 15) return FA2_ERR_INVALID_SHAPE;
    return hip_status(fa2::launch_read_clocks(out32, (hipStream_t)stream));
}

int fa2_mfma_probe(const void* operands, float* out, int iters, int workgroups, void* stream)
{
    if (!operands || !out) return FA2_ERR_NULL_POINTER;
    if (iters < 1 || workgroups < 1) return FA2_ERR_INVALID_SHAPE;
    return hip_status(fa2::launch_mfma_probe(operands, out, iters, workgroups, (hipStream_t)stream));
}

int fa2_fill_f32(float* dst, size_t n, float value, void* stream)
{
    if (!dst && n) return FA2_ERR_NULL_POINTER;
    return hip_status(fa2::launch_fill_f32(dst, n, value, (hipStream_t)stream));
}

int fa2_convert_f32_to_bf16(const float* src, void* dst, size_t n, void* stream)
{
    if ((!src || !dst) && n) return FA2_ERR_NULL_POINTER;
    return hip_status(fa2::launch_f32_to_bf16(src, dst, n, (hipStream_t)stream));
}

int fa2_convert_bf16_to_f32(const void* src, float* dst, size_t n, void* stream)
{
    if ((!src || !dst) && n) return FA2_ERR_NULL_POINTER;
    return hip_status(fa2::launch_bf16_to_f32(src, dst, n, (hipStream_t)stream));
}

}  // extern "C"
