// fa2_ring.cpp -- sequence-sharded ring forward over RCCL/xGMI (include/fa2_ring_mi355x.h).
//
// Host-side schedule only: the per-step compute is fa2_forward_step (libfa2_mi355x.so), the
// transport is RCCL point-to-point.  Replaces ring_attention_forward
// (reference src/03_flash_attention_v2_ring/common/ring_attention_kernel.cu:143-239) and the
// exchange helpers of src/util/nccl_utils.h:115-142.
//
// Stream discipline (no device-wide sync anywhere, cf. ring_attention_kernel.cu:220):
//   compute stream  = the caller's stream: step kernels, in step order;
//   comm stream     = the context's: grouped ncclSend/ncclRecv, in step order;
//   ev_recv[s]      comm -> compute : "the shard step s+1 computes on has landed";
//   ev_comp[s]      compute -> comm : "step s no longer reads its shard buffer" (the buffer is
//                                     the receive target of a later exchange);
//   ev_in           compute -> comm : "the caller's inputs / the workspace are ready".
#include "../../../include/fa2_ring_mi355x.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <new>

namespace {

constexpr int kMaxRanks = 64;

inline int hip_status(hipError_t e) { return e == hipSuccess ? FA2_OK : FA2_ERR_HIP_BASE - (int)e; }
inline int nccl_status(ncclResult_t r) { return r == ncclSuccess ? FA2_OK : FA2_ERR_RCCL_BASE - (int)r; }

#define RING_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_status(e_); } while (0)
#define RING_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return nccl_status(r_); } while (0)

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline size_t elem_size(int dtype) { return dtype == FA2_DTYPE_F32 ? 4 : 2; }

}  // namespace

struct fa2_ring_ctx {
    ncclComm_t comm = nullptr;
    bool owns_comm = false;
    int rank = 0, nranks = 1, device = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_in = nullptr;
    hipEvent_t ev_recv[kMaxRanks] = {};
    hipEvent_t ev_comp[kMaxRanks] = {};
};

namespace {

int ctx_init_common(fa2_ring_ctx* c)
{
    RING_HIP(hipGetDevice(&c->device));
    RING_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    RING_HIP(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    for (int i = 0; i < c->nranks && i < kMaxRanks; ++i) {
        RING_HIP(hipEventCreateWithFlags(&c->ev_recv[i], hipEventDisableTiming));
        RING_HIP(hipEventCreateWithFlags(&c->ev_comp[i], hipEventDisableTiming));
    }
    return FA2_OK;
}

// One grouped exchange: own/resident shard to `to`, a shard from `from` (nccl_utils.h:123-131).
int exchange_pair(fa2_ring_ctx* c, const void* sk, const void* sv, void* rk, void* rv, size_t bytes,
                  int to, int from, hipStream_t s)
{
    RING_NCCL(ncclSend(sk, bytes, ncclInt8, to, c->comm, s));
    RING_NCCL(ncclRecv(rk, bytes, ncclInt8, from, c->comm, s));
    RING_NCCL(ncclSend(sv, bytes, ncclInt8, to, c->comm, s));
    RING_NCCL(ncclRecv(rv, bytes, ncclInt8, from, c->comm, s));
    return FA2_OK;
}

struct Plan {
    size_t shard;      // bytes of one K (or V) shard
    size_t off_acc;    // fp32 accumulator (bf16 path only)
    size_t off_m;      // running max
    size_t off_buf;    // receive slots: [slot][K|V]
    int slots;
    size_t total;
};

Plan make_plan(int B, int H, int nl, int d, int dtype, int nranks, int schedule)
{
    Plan p{};
    const size_t rows = (size_t)B * H * nl;
    p.shard = align256(rows * d * elem_size(dtype));
    p.slots = nranks <= 1 ? 0 : (schedule == FA2_RING_MESH ? nranks - 1 : (nranks == 2 ? 1 : 2));
    size_t off = 0;
    p.off_acc = off; off += dtype == FA2_DTYPE_BF16 ? align256(rows * d * 4) : 0;
    p.off_m = off;   off += align256(rows * 4);
    p.off_buf = off; off += (size_t)p.slots * 2 * p.shard;
    p.total = off;
    return p;
}

}  // namespace

extern "C" {

int fa2_ring_get_unique_id(void* id_out)
{
    if (!id_out) return FA2_ERR_NULL_POINTER;
    static_assert(sizeof(ncclUniqueId) == FA2_RING_UNIQUE_ID_BYTES, "unique id size");
    RING_NCCL(ncclGetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out)));
    return FA2_OK;
}

int fa2_ring_ctx_create(fa2_ring_ctx** out, const void* unique_id, int rank, int nranks)
{
    if (!out || !unique_id) return FA2_ERR_NULL_POINTER;
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return FA2_ERR_INVALID_SHAPE;
    fa2_ring_ctx* c = new (std::nothrow) fa2_ring_ctx();
    if (!c) return FA2_ERR_WORKSPACE;
    c->rank = rank; c->nranks = nranks; c->owns_comm = true;
    ncclUniqueId id;
    __builtin_memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { delete c; return nccl_status(r); }
    int st = ctx_init_common(c);
    if (st) { fa2_ring_ctx_destroy(c); return st; }
    *out = c;
    return FA2_OK;
}

int fa2_ring_ctx_create_from_comm(fa2_ring_ctx** out, void* nccl_comm, int rank, int nranks)
{
    if (!out || !nccl_comm) return FA2_ERR_NULL_POINTER;
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return FA2_ERR_INVALID_SHAPE;
    fa2_ring_ctx* c = new (std::nothrow) fa2_ring_ctx();
    if (!c) return FA2_ERR_WORKSPACE;
    c->rank = rank; c->nranks = nranks; c->owns_comm = false;
    c->comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    int st = ctx_init_common(c);
    if (st) { fa2_ring_ctx_destroy(c); return st; }
    *out = c;
    return FA2_OK;
}

int fa2_ring_ctx_destroy(fa2_ring_ctx* c)
{
    if (!c) return FA2_OK;
    if (c->comm_stream) { (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamDestroy(c->comm_stream); }
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    for (int i = 0; i < kMaxRanks; ++i) {
        if (c->ev_recv[i]) (void)hipEventDestroy(c->ev_recv[i]);
        if (c->ev_comp[i]) (void)hipEventDestroy(c->ev_comp[i]);
    }
    int st = FA2_OK;
    if (c->owns_comm && c->comm) st = nccl_status(ncclCommDestroy(c->comm));
    delete c;
    return st;
}

size_t fa2_ring_workspace_bytes(int B, int H, int local_seq_len, int head_dim, int dtype,
                                int nranks, int schedule)
{
    if (B <= 0 || H <= 0 || local_seq_len <= 0 || head_dim <= 0 || nranks < 1) return 0;
    return make_plan(B, H, local_seq_len, head_dim, dtype, nranks, schedule).total;
}

int fa2_ring_exchange_kv(fa2_ring_ctx* c, const void* send_k, void* recv_k,
                         const void* send_v, void* recv_v, size_t bytes, void* stream)
{
    if (!c || !send_k || !recv_k || !send_v || !recv_v) return FA2_ERR_NULL_POINTER;
    const int next = (c->rank + 1) % c->nranks, prev = (c->rank - 1 + c->nranks) % c->nranks;
    RING_NCCL(ncclGroupStart());
    int st = exchange_pair(c, send_k, send_v, recv_k, recv_v, bytes, next, prev, (hipStream_t)stream);
    ncclResult_t r = ncclGroupEnd();
    if (st) return st;
    return nccl_status(r);
}

}  // extern "C"

// Shared driver of the plain and the causal (zig-zag) ring.  Causal layout: the sequence is cut into 2 P
// chunks of c = local / 2 rows; rank r holds chunk r in its local rows [0, c) and chunk 2 P - 1 - r in rows
// [c, 2 c).  With the shard of owner o resident: o == r is plain causal attention over the local rows (chunk
// r precedes chunk 2 P - 1 - r); o < r means only the owner's FIRST chunk is visible, to every local
// row, unmasked; o > r means BOTH its chunks are visible, unmasked, to the local rows of the second chunk
// only.  Every step after the first therefore costs half a dense block on every rank -- the point of the
// zig-zag order.  The last step need not touch every row, so results are produced by a finalize pass.
static int ring_forward_impl(fa2_ring_ctx* c,
                             const void* Q_local, const void* K_local, const void* V_local,
                             void* O_local, float* L_local,
                             int B, int H, int total_seq_len, int local_seq_len, int head_dim,
                             float softmax_scale, int dtype, int schedule, int causal,
                             void* workspace, size_t workspace_bytes, void* stream_)
{
    if (!c || !Q_local || !K_local || !V_local || !O_local || !L_local) return FA2_ERR_NULL_POINTER;
    if (B <= 0 || H <= 0 || local_seq_len <= 0 || head_dim <= 0) return FA2_ERR_INVALID_SHAPE;
    const int P = c->nranks;
    if ((long long)local_seq_len * P != (long long)total_seq_len) return FA2_ERR_INVALID_SHAPE;  // 04_ring_attention.cu:55-63
    if (causal && dtype != FA2_DTYPE_BF16) return FA2_ERR_UNSUPPORTED_DTYPE;
    if (causal && (local_seq_len & 1)) return FA2_ERR_INVALID_SHAPE;
    if (dtype != FA2_DTYPE_BF16 && dtype != FA2_DTYPE_F32) return FA2_ERR_UNSUPPORTED_DTYPE;
    if (schedule != FA2_RING_RELAY && schedule != FA2_RING_MESH) return FA2_ERR_UNSUPPORTED;
    const Plan pl = make_plan(B, H, local_seq_len, head_dim, dtype, P, schedule);
    if (!workspace || workspace_bytes < pl.total) return FA2_ERR_WORKSPACE;

    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    float* acc = dtype == FA2_DTYPE_BF16 ? (float*)(ws + pl.off_acc) : nullptr;
    float* M = (float*)(ws + pl.off_m);
    auto slotK = [&](int i) { return (void*)(ws + pl.off_buf + (size_t)i * 2 * pl.shard); };
    auto slotV = [&](int i) { return (void*)(ws + pl.off_buf + (size_t)i * 2 * pl.shard + pl.shard); };
    const size_t bytes = (size_t)B * H * local_seq_len * head_dim * elem_size(dtype);
    const int rank = c->rank;

    const int half = local_seq_len / 2;
    auto step = [&](const void* Kc, const void* Vc, int s) -> int {
        if (!causal)
            return fa2_forward_step(Q_local, Kc, Vc, O_local, L_local, acc, M, B, H, local_seq_len,
                                    local_seq_len, head_dim, softmax_scale, dtype,
                                    s == 0 ? 1 : 0, s == P - 1 ? 1 : 0, stream);
        const int owner = (rank - s + P) % P;
        if (owner == rank)            // local block: causal over the local row order
            return fa2_forward_step_strided(Q_local, Kc, Vc, O_local, L_local, acc, M, B, H, local_seq_len, local_seq_len,
                                            head_dim, softmax_scale, dtype, 1, 0, 0, 0, 1, 0, stream);
        if (owner < rank)             // the owner's first chunk, visible to every local row
            return fa2_forward_step_strided(Q_local, Kc, Vc, O_local, L_local, acc, M, B, H, local_seq_len, half, head_dim,
                                            softmax_scale, dtype, 0, 0, 0, local_seq_len, 0, 0, stream);
        // both of the owner's chunks, visible to the rows of the local second chunk only
        const size_t ro = (size_t)half, eo = ro * head_dim;
        return fa2_forward_step_strided((const char*)Q_local + eo * 2, Kc, Vc, (char*)O_local + eo * 2, L_local + ro, acc + eo,
                                        M + ro, B, H, half, local_seq_len, head_dim, softmax_scale, dtype, 0, 0,
                                        local_seq_len, 0, 0, 0, stream);
    };
    auto finish = [&]() -> int {
        if (!causal) return FA2_OK;
        return fa2_forward_state_finalize(O_local, L_local, acc, M, (size_t)B * H * local_seq_len, head_dim, dtype, stream);
    };

    if (P == 1) { int st1 = step(K_local, V_local, 0); return st1 ? st1 : finish(); }

    // the comm stream may touch the workspace / read the inputs only after everything the
    // caller queued before this call
    RING_HIP(hipEventRecord(c->ev_in, stream));
    RING_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_in, 0));

    if (schedule == FA2_RING_RELAY) {
        const int next = (rank + 1) % P, prev = (rank - 1 + P) % P;
        // resident buffer of step s: the caller's shard at s = 0, then the two slots alternately
        auto curK = [&](int s) { return s == 0 ? K_local : (const void*)slotK(pl.slots == 1 ? 0 : (s - 1) & 1); };
        auto curV = [&](int s) { return s == 0 ? V_local : (const void*)slotV(pl.slots == 1 ? 0 : (s - 1) & 1); };
        for (int s = 0; s < P; ++s) {
            if (s < P - 1) {
                // the receive target cur(s+1) was last read by step s-1
                if (s >= 2) RING_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_comp[s - 1], 0));
                RING_NCCL(ncclGroupStart());
                int st = exchange_pair(c, curK(s), curV(s), (void*)curK(s + 1), (void*)curV(s + 1), bytes,
                                       next, prev, c->comm_stream);
                ncclResult_t r = ncclGroupEnd();
                if (st) return st;
                RING_NCCL(r);
                RING_HIP(hipEventRecord(c->ev_recv[s], c->comm_stream));
            }
            if (s >= 1) RING_HIP(hipStreamWaitEvent(stream, c->ev_recv[s - 1], 0));
            int st = step(curK(s), curV(s), s);
            if (st) return st;
            if (s < P - 1) RING_HIP(hipEventRecord(c->ev_comp[s], stream));
        }
        return finish();
    }

    // FA2_RING_MESH: step s computes on the shard owned by rank - s (the relay's order), fetched
    // directly from its owner into slot s-1.  Two grouped exchanges: the shard step 1 needs
    // first, then all the others at once over the remaining links.
    auto issue = [&](int s_lo, int s_hi, hipEvent_t done) -> int {
        RING_NCCL(ncclGroupStart());
        int st = FA2_OK;
        for (int s = s_lo; s <= s_hi && !st; ++s)
            st = exchange_pair(c, K_local, V_local, slotK(s - 1), slotV(s - 1), bytes,
                               (rank + s) % P, (rank - s + P) % P, c->comm_stream);
        ncclResult_t r = ncclGroupEnd();
        if (st) return st;
        RING_NCCL(r);
        RING_HIP(hipEventRecord(done, c->comm_stream));
        return FA2_OK;
    };
    int st = issue(1, 1, c->ev_recv[0]);
    if (st) return st;
    if (P > 2) {
        st = issue(2, P - 1, c->ev_recv[1]);
        if (st) return st;
    }
    st = step(K_local, V_local, 0);
    if (st) return st;
    for (int s = 1; s < P; ++s) {
        if (s == 1) RING_HIP(hipStreamWaitEvent(stream, c->ev_recv[0], 0));
        if (s == 2) RING_HIP(hipStreamWaitEvent(stream, c->ev_recv[1], 0));
        st = step(slotK(s - 1), slotV(s - 1), s);
        if (st) return st;
    }
    return finish();
}

extern "C" {

int fa2_ring_attention_forward(fa2_ring_ctx* c,
                               const void* Q_local, const void* K_local, const void* V_local,
                               void* O_local, float* L_local,
                               int B, int H, int total_seq_len, int local_seq_len, int head_dim,
                               float softmax_scale, int dtype, int schedule,
                               void* workspace, size_t workspace_bytes, void* stream)
{
    return ring_forward_impl(c, Q_local, K_local, V_local, O_local, L_local, B, H, total_seq_len, local_seq_len, head_dim,
                             softmax_scale, dtype, schedule, 0, workspace, workspace_bytes, stream);
}

int fa2_ring_attention_forward_causal(fa2_ring_ctx* c,
                                      const void* Q_local, const void* K_local, const void* V_local,
                                      void* O_local, float* L_local,
                                      int B, int H, int total_seq_len, int local_seq_len, int head_dim,
                                      float softmax_scale, int dtype, int schedule,
                                      void* workspace, size_t workspace_bytes, void* stream)
{
    return ring_forward_impl(c, Q_local, K_local, V_local, O_local, L_local, B, H, total_seq_len, local_seq_len, head_dim,
                             softmax_scale, dtype, schedule, 1, workspace, workspace_bytes, stream);
}

int ring_attention_forward(const float* Q_local, float* K_local, float* V_local,
                           float* O_local, float* L_local,
                           int total_seq_len, int local_seq_len, int head_dim, float softmax_scale,
                           void* comm, int rank, int nranks)
{
    fa2_ring_ctx* c = nullptr;
    int st = fa2_ring_ctx_create_from_comm(&c, comm, rank, nranks);
    if (st) return st;
    const size_t need = fa2_ring_workspace_bytes(1, 1, local_seq_len, head_dim, FA2_DTYPE_F32, nranks, FA2_RING_RELAY);
    void* ws = nullptr;
    hipError_t e = hipMalloc(&ws, need ? need : 256);
    if (e != hipSuccess) { fa2_ring_ctx_destroy(c); return hip_status(e); }
    for (int s = 0; s < nranks; ++s)   // the reference's progress line (ring_attention_kernel.cu:201-202)
        printf("Rank %d, Step %d: Processing K,V block from rank %d\n", rank, s, (rank - s + nranks) % nranks);
    st = fa2_ring_attention_forward(c, Q_local, K_local, V_local, O_local, L_local, 1, 1, total_seq_len,
                                    local_seq_len, head_dim, softmax_scale, FA2_DTYPE_F32, FA2_RING_RELAY,
                                    ws, need, nullptr);
    e = hipDeviceSynchronize();
    (void)hipFree(ws);
    int st2 = fa2_ring_ctx_destroy(c);
    if (st) return st;
    if (e != hipSuccess) return hip_status(e);
    return st2;
}

// ---------------------------------------------------------------------------------------------------
// Ring backward (past the reference, whose ring is forward-only: SURVEY 8f rank 2).  Q, dO, O, L stay
// put; the K/V shards are fetched from their owners as in the mesh forward.  While owner o's shard is
// resident a rank runs the ordinary two backward kernels on (its rows) x (o's keys): the dQ part is
// added to its own fp32 running sum, the dK/dV parts -- gradients of o's keys -- are sent to o, which adds
// what it receives to ITS fp32 running sums (send to rank - s, receive from rank + s: a permutation per
// step).  L must be the log-sum-exp over the WHOLE sequence (the ring forward's output), which is what
// makes the per-shard pieces add up.  First version: transfers and kernels of a step run one after the
// other (only the initial K/V fetch overlaps compute).
namespace {
struct BwdPlan {
    size_t shard, acc;                 // bytes of one bf16 shard / of one fp32 running sum
    size_t off_kv, off_ws, off_tmp, off_rcv, off_acc, total, ws_bytes;
};
BwdPlan make_bwd_plan(int B, int H, int nl, int d, int P)
{
    BwdPlan p{};
    const size_t elems = (size_t)B * H * nl * d;
    p.shard = align256(elems * 2);
    p.acc = align256(elems * 4);
    p.ws_bytes = align256(fa2_backward_workspace_bytes(B, H, nl, d, FA2_DTYPE_BF16));
    size_t off = 0;
    p.off_kv = off;  off += (size_t)(P > 1 ? P - 1 : 0) * 2 * p.shard;     // fetched K/V shards
    p.off_ws = off;  off += p.ws_bytes;                                    // the backward kernels' own scratch
    p.off_tmp = off; off += 3 * p.shard;                                   // this step's dQ, dK, dV (bf16)
    p.off_rcv = off; off += 2 * p.shard;                                   // received dK, dV contributions
    p.off_acc = off; off += 3 * p.acc;                                     // fp32 running sums
    p.total = off;
    return p;
}
}  // namespace

size_t fa2_ring_backward_workspace_bytes(int B, int H, int local_seq_len, int head_dim, int dtype, int nranks)
{
    if (B <= 0 || H <= 0 || local_seq_len <= 0 || head_dim <= 0 || nranks <= 0 || dtype != FA2_DTYPE_BF16) return 0;
    return make_bwd_plan(B, H, local_seq_len, head_dim, nranks).total;
}

int fa2_ring_attention_backward(fa2_ring_ctx* c,
                                const void* Q_local, const void* K_local, const void* V_local,
                                const void* O_local, const float* L_local, const void* dO_local,
                                void* dQ_local, void* dK_local, void* dV_local,
                                int B, int H, int total_seq_len, int local_seq_len, int head_dim,
                                float softmax_scale, int dtype,
                                void* workspace, size_t workspace_bytes, void* stream_)
{
    if (!c || !Q_local || !K_local || !V_local || !O_local || !L_local || !dO_local || !dQ_local || !dK_local || !dV_local)
        return FA2_ERR_NULL_POINTER;
    if (B <= 0 || H <= 0 || local_seq_len <= 0 || head_dim <= 0) return FA2_ERR_INVALID_SHAPE;
    const int P = c->nranks, rank = c->rank;
    if ((long long)local_seq_len * P != (long long)total_seq_len) return FA2_ERR_INVALID_SHAPE;
    if (dtype != FA2_DTYPE_BF16) return FA2_ERR_UNSUPPORTED_DTYPE;
    const BwdPlan pl = make_bwd_plan(B, H, local_seq_len, head_dim, P);
    if (!workspace || workspace_bytes < pl.total) return FA2_ERR_WORKSPACE;

    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    auto slotK = [&](int i) { return (void*)(ws + pl.off_kv + (size_t)i * 2 * pl.shard); };
    auto slotV = [&](int i) { return (void*)(ws + pl.off_kv + (size_t)i * 2 * pl.shard + pl.shard); };
    void* bws = ws + pl.off_ws;
    void* tq = ws + pl.off_tmp; void* tk = ws + pl.off_tmp + pl.shard; void* tv = ws + pl.off_tmp + 2 * pl.shard;
    void* rk = ws + pl.off_rcv; void* rv = ws + pl.off_rcv + pl.shard;
    float* aq = (float*)(ws + pl.off_acc); float* ak = (float*)(ws + pl.off_acc + pl.acc); float* av = (float*)(ws + pl.off_acc + 2 * pl.acc);
    const size_t elems = (size_t)B * H * local_seq_len * head_dim;
    const size_t bytes = elems * 2;

    // K/V shards from their owners (two grouped exchanges, as in the mesh forward)
    if (P > 1) {
        RING_HIP(hipEventRecord(c->ev_in, stream));
        RING_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_in, 0));
        for (int part = 0; part < 2; ++part) {
            const int lo = part == 0 ? 1 : 2, hi = part == 0 ? 1 : P - 1;
            if (lo > hi) continue;
            RING_NCCL(ncclGroupStart());
            int st = FA2_OK;
            for (int s = lo; s <= hi && !st; ++s)
                st = exchange_pair(c, K_local, V_local, slotK(s - 1), slotV(s - 1), bytes, (rank + s) % P, (rank - s + P) % P,
                                   c->comm_stream);
            ncclResult_t r = ncclGroupEnd();
            if (st) return st;
            RING_NCCL(r);
            RING_HIP(hipEventRecord(c->ev_recv[part], c->comm_stream));
        }
    }
    // D = rowsum(dO o O) and the row constants, once: they depend on local rows only
    int st = fa2_backward_phases(Q_local, K_local, V_local, O_local, L_local, dO_local, tq, tk, tv, B, H, local_seq_len, head_dim,
                                 softmax_scale, dtype, 0, bws, pl.ws_bytes, stream, 1);
    if (st) return st;
    for (int s = 0; s < P; ++s) {
        const void* Kc = s == 0 ? K_local : slotK(s - 1);
        const void* Vc = s == 0 ? V_local : slotV(s - 1);
        if (s == 1) RING_HIP(hipStreamWaitEvent(stream, c->ev_recv[0], 0));
        if (s == 2) RING_HIP(hipStreamWaitEvent(stream, c->ev_recv[1], 0));
        st = fa2_backward_phases(Q_local, Kc, Vc, O_local, L_local, dO_local, tq, tk, tv, B, H, local_seq_len, head_dim,
                                 softmax_scale, dtype, 0, bws, pl.ws_bytes, stream, 6);
        if (st) return st;
        st = fa2_accumulate_bf16(aq, tq, elems, s == 0, stream);
        if (st) return st;
        if (s == 0) {                               // own keys: the contribution stays here
            st = fa2_accumulate_bf16(ak, tk, elems, 1, stream);
            if (!st) st = fa2_accumulate_bf16(av, tv, elems, 1, stream);
            if (st) return st;
            continue;
        }
        // gradients of owner (rank - s)'s keys go to it; those of mine computed by rank + s come in
        RING_HIP(hipEventRecord(c->ev_comp[s], stream));
        RING_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_comp[s], 0));
        RING_NCCL(ncclGroupStart());
        st = exchange_pair(c, tk, tv, rk, rv, bytes, (rank - s + P) % P, (rank + s) % P, c->comm_stream);
        ncclResult_t r = ncclGroupEnd();
        if (st) return st;
        RING_NCCL(r);
        RING_HIP(hipEventRecord(c->ev_in, c->comm_stream));
        RING_HIP(hipStreamWaitEvent(stream, c->ev_in, 0));
        st = fa2_accumulate_bf16(ak, rk, elems, 0, stream);
        if (!st) st = fa2_accumulate_bf16(av, rv, elems, 0, stream);
        if (st) return st;
    }
    st = fa2_convert_f32_to_bf16(aq, dQ_local, elems, stream);
    if (!st) st = fa2_convert_f32_to_bf16(ak, dK_local, elems, stream);
    if (!st) st = fa2_convert_f32_to_bf16(av, dV_local, elems, stream);
    return st;
}

}  // extern "C"
