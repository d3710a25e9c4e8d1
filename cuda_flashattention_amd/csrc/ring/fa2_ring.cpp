// fa2_ring.cpp -- sequence-sharded ring attention over RCCL/xGMI (include/fa2_ring_mi355x.h).
//
// Host-side schedules only.  Replaces ring_attention_forward
// (reference src/03_flash_attention_v2_ring/common/ring_attention_kernel.cu:143-239) and the
// exchange helpers of src/util/nccl_utils.h:115-142.  Everything a schedule does to a device goes
// through a fa2_ring_backend table: the product table below is HIP streams/events + RCCL grouped
// ncclSend/ncclRecv + the kernels of libfa2_mi355x.so; tests/ install others (a one-GPU loopback
// transport, a CPU simulator) so that THIS code -- slot rotation, event fences, peer arithmetic --
// is what runs at P = 2, 4, 8 without eight GPUs.
//
// Stream discipline (no device-wide sync anywhere, cf. ring_attention_kernel.cu:220):
//   compute stream  = the caller's stream: step kernels, in step order;
//   comm stream     = the context's: grouped send/recv, in step order;
//   ev_recv[s]      comm -> compute : "the shard step s+1 computes on has landed";
//   ev_comp[s]      compute -> comm : "step s no longer reads its shard buffer / has produced its
//                                     gradients" (the buffer is the target of a later exchange);
//   ev_in           compute -> comm : "the caller's inputs / the workspace are ready";
//   ev_x[s] (backward) comm -> compute: "the gradient exchange of step s is done".
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

#define RING_TRY(call) do { int st_ = (call); if (st_ != FA2_OK) return st_; } while (0)

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline size_t elem_size(int dtype) { return dtype == FA2_DTYPE_F32 ? 4 : 2; }

}  // namespace

struct fa2_ring_ctx {
    fa2_ring_backend be{};
    ncclComm_t comm = nullptr;
    bool owns_comm = false;
    int rank = 0, nranks = 1;
    void* comm_stream = nullptr;
    void* ev_in = nullptr;
    void* ev_recv[kMaxRanks] = {};
    void* ev_comp[kMaxRanks] = {};
    void* ev_x[kMaxRanks] = {};
    int reserve_cus = 16;       // fa2_ring_ctx_set_reserved_cus
};

// ---------------------------------------------------------------------------------------------------
// The product backend: HIP + RCCL + libfa2_mi355x.so.
// ---------------------------------------------------------------------------------------------------
namespace {

int hb_stream_create(void*, void** out)
{
    // the ring's private streams carry the exchanges: highest priority, so that an RCCL kernel is dispatched as soon as a CU
    // frees up beside the step kernels of the caller's stream (they fill the chip)
    hipStream_t s = nullptr;
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest);
    if (e != hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    *out = (void*)s;
    return hip_status(e);
}
int hb_stream_destroy(void*, void* s)
{
    (void)hipStreamSynchronize((hipStream_t)s);
    return hip_status(hipStreamDestroy((hipStream_t)s));
}
int hb_event_create(void*, void** out)
{
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    *out = (void*)ev;
    return hip_status(e);
}
int hb_event_destroy(void*, void* ev) { return hip_status(hipEventDestroy((hipEvent_t)ev)); }
int hb_event_record(void*, void* ev, void* s) { return hip_status(hipEventRecord((hipEvent_t)ev, (hipStream_t)s)); }
int hb_stream_wait_event(void*, void* s, void* ev) { return hip_status(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)ev, 0)); }

// RCCL transport: `user` is the context (it holds the communicator).
int hb_group_start(void*) { return nccl_status(ncclGroupStart()); }
int hb_group_end(void*) { return nccl_status(ncclGroupEnd()); }
int hb_send(void* user, const void* buf, size_t bytes, int peer, void* s)
{
    fa2_ring_ctx* c = (fa2_ring_ctx*)user;
    if (!c || !c->comm) return FA2_ERR_NULL_POINTER;
    return nccl_status(ncclSend(buf, bytes, ncclInt8, peer, c->comm, (hipStream_t)s));
}
int hb_recv(void* user, void* buf, size_t bytes, int peer, void* s)
{
    fa2_ring_ctx* c = (fa2_ring_ctx*)user;
    if (!c || !c->comm) return FA2_ERR_NULL_POINTER;
    return nccl_status(ncclRecv(buf, bytes, ncclInt8, peer, c->comm, (hipStream_t)s));
}

int hb_forward_step(void*, const void* Q, const void* K, const void* V, void* O, float* L, float* Oacc, float* M, int B, int H,
                    int q_len, int kv_len, int d, float scale, int dtype, int first, int last, int q_hs, int kv_hs, int causal,
                    int causal_shift, void* stream)
{
    if (dtype == FA2_DTYPE_F32) {
        if (q_hs || kv_hs || causal) return FA2_ERR_UNSUPPORTED;
        return fa2_forward_step(Q, K, V, O, L, Oacc, M, B, H, q_len, kv_len, d, scale, dtype, first, last, stream);
    }
    return fa2_forward_step_strided(Q, K, V, O, L, Oacc, M, B, H, q_len, kv_len, d, scale, dtype, first, last, q_hs, kv_hs,
                                    causal, causal_shift, stream);
}
int hb_state_finalize(void*, void* O, float* L, const float* Oacc, const float* M, size_t rows, int d, int dtype, void* stream)
{
    return fa2_forward_state_finalize(O, L, Oacc, M, rows, d, dtype, stream);
}
int hb_backward_block(void*, const void* Q, const void* K, const void* V, const void* O, const float* L, const void* dO, void* dQ,
                      void* dK, void* dV, int B, int H, int q_len, int kv_len, int d, float scale, int dtype, int q_hs, int kv_hs,
                      int q_row0, int causal, int causal_shift, void* ws, size_t ws_bytes, void* stream, int phases)
{
    return fa2_backward_block(Q, K, V, O, L, dO, dQ, dK, dV, B, H, q_len, kv_len, d, scale, dtype, q_hs, kv_hs, q_row0, causal,
                              causal_shift, ws, ws_bytes, stream, phases);
}
int hb_accumulate(void*, float* acc, const void* src, size_t rows, size_t cols, size_t pitch, int init, void* stream)
{
    return fa2_accumulate_bf16_2d(acc, src, rows, cols, pitch, init, stream);
}
int hb_convert(void*, const float* src, void* dst, size_t n, void* stream) { return fa2_convert_f32_to_bf16(src, dst, n, stream); }

void fill_default_backend(fa2_ring_backend* b)
{
    b->user = nullptr;
    b->stream_create = hb_stream_create;
    b->stream_destroy = hb_stream_destroy;
    b->event_create = hb_event_create;
    b->event_destroy = hb_event_destroy;
    b->event_record = hb_event_record;
    b->stream_wait_event = hb_stream_wait_event;
    b->group_start = hb_group_start;
    b->send = hb_send;
    b->recv = hb_recv;
    b->group_end = hb_group_end;
    b->forward_step = hb_forward_step;
    b->state_finalize = hb_state_finalize;
    b->backward_block = hb_backward_block;
    b->accumulate_bf16_2d = hb_accumulate;
    b->convert_f32_to_bf16 = hb_convert;
}

bool backend_complete(const fa2_ring_backend* b)
{
    return b->stream_create && b->stream_destroy && b->event_create && b->event_destroy && b->event_record &&
           b->stream_wait_event && b->group_start && b->send && b->recv && b->group_end && b->forward_step &&
           b->state_finalize && b->backward_block && b->accumulate_bf16_2d && b->convert_f32_to_bf16;
}

int ctx_init_common(fa2_ring_ctx* c)
{
    const fa2_ring_backend& b = c->be;
    RING_TRY(b.stream_create(b.user, &c->comm_stream));
    RING_TRY(b.event_create(b.user, &c->ev_in));
    for (int i = 0; i < c->nranks && i < kMaxRanks; ++i) {
        RING_TRY(b.event_create(b.user, &c->ev_recv[i]));
        RING_TRY(b.event_create(b.user, &c->ev_comp[i]));
        RING_TRY(b.event_create(b.user, &c->ev_x[i]));
    }
    return FA2_OK;
}

// Thin views of the backend for the schedules below.
struct Dev {
    fa2_ring_ctx* c;
    int record(void* ev, void* s) const { return c->be.event_record(c->be.user, ev, s); }
    int wait(void* s, void* ev) const { return c->be.stream_wait_event(c->be.user, s, ev); }
    int group_start() const { return c->be.group_start(c->be.user); }
    int group_end() const { return c->be.group_end(c->be.user); }
    // One K+V pair: send (sk, sv) to `to`, receive (rk, rv) from `from` (nccl_utils.h:123-131).
    int exchange_pair(const void* sk, const void* sv, void* rk, void* rv, size_t bytes, int to, int from, void* s) const
    {
        const fa2_ring_backend& b = c->be;
        RING_TRY(b.send(b.user, sk, bytes, to, s));
        RING_TRY(b.recv(b.user, rk, bytes, from, s));
        RING_TRY(b.send(b.user, sv, bytes, to, s));
        RING_TRY(b.recv(b.user, rv, bytes, from, s));
        return FA2_OK;
    }
    // group_start ... group_end around `body`; the group is always closed, the first error wins.
    template <typename F>
    int grouped(F&& body) const
    {
        RING_TRY(group_start());
        const int st = body();
        const int en = group_end();
        return st ? st : en;
    }
};

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

int fa2_ring_default_backend(fa2_ring_backend* out)
{
    if (!out) return FA2_ERR_NULL_POINTER;
    fill_default_backend(out);
    return FA2_OK;
}

int fa2_ring_get_unique_id(void* id_out)
{
    if (!id_out) return FA2_ERR_NULL_POINTER;
    static_assert(sizeof(ncclUniqueId) == FA2_RING_UNIQUE_ID_BYTES, "unique id size");
    return nccl_status(ncclGetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out)));
}

int fa2_ring_ctx_create(fa2_ring_ctx** out, const void* unique_id, int rank, int nranks)
{
    if (!out || !unique_id) return FA2_ERR_NULL_POINTER;
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return FA2_ERR_INVALID_SHAPE;
    fa2_ring_ctx* c = new (std::nothrow) fa2_ring_ctx();
    if (!c) return FA2_ERR_WORKSPACE;
    fill_default_backend(&c->be);
    c->be.user = c;
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
    fill_default_backend(&c->be);
    c->be.user = c;
    c->rank = rank; c->nranks = nranks; c->owns_comm = false;
    c->comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    int st = ctx_init_common(c);
    if (st) { fa2_ring_ctx_destroy(c); return st; }
    *out = c;
    return FA2_OK;
}

int fa2_ring_ctx_create_with_backend(fa2_ring_ctx** out, const fa2_ring_backend* backend, int rank, int nranks)
{
    if (!out || !backend) return FA2_ERR_NULL_POINTER;
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return FA2_ERR_INVALID_SHAPE;
    if (!backend_complete(backend)) return FA2_ERR_NULL_POINTER;
    fa2_ring_ctx* c = new (std::nothrow) fa2_ring_ctx();
    if (!c) return FA2_ERR_WORKSPACE;
    c->be = *backend;
    c->rank = rank; c->nranks = nranks;
    int st = ctx_init_common(c);
    if (st) { fa2_ring_ctx_destroy(c); return st; }
    *out = c;
    return FA2_OK;
}

int fa2_ring_ctx_set_reserved_cus(fa2_ring_ctx* c, int n)
{
    if (!c) return FA2_ERR_NULL_POINTER;
    if (n < 1 || n > 255) return FA2_ERR_INVALID_SHAPE;
    c->reserve_cus = n;
    return FA2_OK;
}

int fa2_ring_ctx_destroy(fa2_ring_ctx* c)
{
    if (!c) return FA2_OK;
    const fa2_ring_backend& b = c->be;
    if (c->comm_stream) (void)b.stream_destroy(b.user, c->comm_stream);     // drains it first
    if (c->ev_in) (void)b.event_destroy(b.user, c->ev_in);
    for (int i = 0; i < kMaxRanks; ++i) {
        if (c->ev_recv[i]) (void)b.event_destroy(b.user, c->ev_recv[i]);
        if (c->ev_comp[i]) (void)b.event_destroy(b.user, c->ev_comp[i]);
        if (c->ev_x[i]) (void)b.event_destroy(b.user, c->ev_x[i]);
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
    const Dev dev{c};
    return dev.grouped([&] { return dev.exchange_pair(send_k, send_v, recv_k, recv_v, bytes, next, prev, stream); });
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
                             void* workspace, size_t workspace_bytes, void* stream)
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

    const Dev dev{c};
    const fa2_ring_backend& be = c->be;
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
            return be.forward_step(be.user, Q_local, Kc, Vc, O_local, L_local, acc, M, B, H, local_seq_len, local_seq_len,
                                   head_dim, softmax_scale, dtype, s == 0 ? 1 : 0, s == P - 1 ? 1 : 0, 0, 0, 0, 0, stream);
        const int owner = (rank - s + P) % P;
        if (owner == rank)            // local block: causal over the local row order
            return be.forward_step(be.user, Q_local, Kc, Vc, O_local, L_local, acc, M, B, H, local_seq_len, local_seq_len,
                                   head_dim, softmax_scale, dtype, 1, 0, 0, 0, 1, 0, stream);
        if (owner < rank)             // the owner's first chunk, visible to every local row
            return be.forward_step(be.user, Q_local, Kc, Vc, O_local, L_local, acc, M, B, H, local_seq_len, half, head_dim,
                                   softmax_scale, dtype, 0, 0, 0, local_seq_len, 0, 0, stream);
        // both of the owner's chunks, visible to the rows of the local second chunk only
        const size_t ro = (size_t)half, eo = ro * head_dim;
        return be.forward_step(be.user, (const char*)Q_local + eo * 2, Kc, Vc, (char*)O_local + eo * 2, L_local + ro, acc + eo,
                               M + ro, B, H, half, local_seq_len, head_dim, softmax_scale, dtype, 0, 0, local_seq_len, 0, 0, 0,
                               stream);
    };
    auto finish = [&]() -> int {
        if (!causal) return FA2_OK;
        return be.state_finalize(be.user, O_local, L_local, acc, M, (size_t)B * H * local_seq_len, head_dim, dtype, stream);
    };

    if (P == 1) { int st1 = step(K_local, V_local, 0); return st1 ? st1 : finish(); }

    // the comm stream may touch the workspace / read the inputs only after everything the
    // caller queued before this call
    RING_TRY(dev.record(c->ev_in, stream));
    RING_TRY(dev.wait(c->comm_stream, c->ev_in));

    if (schedule == FA2_RING_RELAY) {
        const int next = (rank + 1) % P, prev = (rank - 1 + P) % P;
        // resident buffer of step s: the caller's shard at s = 0, then the two slots alternately
        auto curK = [&](int s) { return s == 0 ? K_local : (const void*)slotK(pl.slots == 1 ? 0 : (s - 1) & 1); };
        auto curV = [&](int s) { return s == 0 ? V_local : (const void*)slotV(pl.slots == 1 ? 0 : (s - 1) & 1); };
        for (int s = 0; s < P; ++s) {
            if (s < P - 1) {
                // the receive target cur(s+1) was last read by step s-1
                if (s >= 2) RING_TRY(dev.wait(c->comm_stream, c->ev_comp[s - 1]));
                RING_TRY(dev.grouped([&] {
                    return dev.exchange_pair(curK(s), curV(s), (void*)curK(s + 1), (void*)curV(s + 1), bytes, next, prev,
                                             c->comm_stream);
                }));
                RING_TRY(dev.record(c->ev_recv[s], c->comm_stream));
            }
            if (s >= 1) RING_TRY(dev.wait(stream, c->ev_recv[s - 1]));
            RING_TRY(step(curK(s), curV(s), s));
            if (s < P - 1) RING_TRY(dev.record(c->ev_comp[s], stream));
        }
        return finish();
    }

    // FA2_RING_MESH: step s computes on the shard owned by rank - s (the relay's order), fetched
    // directly from its owner into slot s-1.  Two grouped exchanges: the shard step 1 needs
    // first, then all the others at once over the remaining links.
    auto issue = [&](int s_lo, int s_hi, void* done) -> int {
        RING_TRY(dev.grouped([&] {
            int st = FA2_OK;
            for (int s = s_lo; s <= s_hi && !st; ++s)
                st = dev.exchange_pair(K_local, V_local, slotK(s - 1), slotV(s - 1), bytes, (rank + s) % P, (rank - s + P) % P,
                                       c->comm_stream);
            return st;
        }));
        return dev.record(done, c->comm_stream);
    };
    RING_TRY(issue(1, 1, c->ev_recv[0]));
    if (P > 2) RING_TRY(issue(2, P - 1, c->ev_recv[1]));
    RING_TRY(step(K_local, V_local, 0));
    for (int s = 1; s < P; ++s) {
        if (s == 1) RING_TRY(dev.wait(stream, c->ev_recv[0]));
        if (s == 2) RING_TRY(dev.wait(stream, c->ev_recv[1]));
        RING_TRY(step(slotK(s - 1), slotV(s - 1), s));
    }
    return finish();
}

// ---------------------------------------------------------------------------------------------------
// Ring backward (past the reference, whose ring is forward-only: SURVEY 8f rank 2).  Q, dO, O, L stay
// put; the K/V shards are fetched from their owners as in the mesh forward.  While owner o's shard is
// resident a rank runs the ordinary two backward kernels on (its rows) x (o's keys): the dQ part is
// added to its own fp32 running sum, the dK/dV parts -- gradients of o's keys -- are sent to o, which adds
// what it receives to ITS fp32 running sums (send to rank - s, receive from rank + s: a permutation per
// step).  L must be the log-sum-exp over the WHOLE sequence (the ring forward's output), which is what
// makes the per-shard pieces add up.
//
// Overlap: the gradient pieces of step s travel while the kernels of step s+1 run.  Two sets of send
// buffers (step parity) and two of receive buffers; per step
//   compute stream: block kernels of step s, dQ sum, record ev_comp[s];  then the DEFERRED sums of step s-1:
//                   wait ev_x[s-1], add its receive buffers;
//   comm stream:    wait ev_comp[s], grouped exchange, record ev_x[s].
// These two fences are all the double buffering needs: the kernels of step s overwrite the send buffers that
// exchange s-2 read, and the compute stream has waited for ev_x[s-2] already (the deferred sums of step s-1);
// exchange s overwrites the receive buffers whose sums were enqueued at step s-1, before ev_comp[s].
// Causal (zig-zag sharding as in the causal forward): the block of owner o is the local causal block
// (o == r), o's first chunk of keys against every local row (o < r), or all of o's keys against the rows of
// the second local chunk (o > r); in the o < r case only the first half of every head's dK/dV piece is
// meaningful and only that half is added by the receiver (who knows the sender's case from the ranks).
// The pieces travel as bf16 (each is rounded once; the sums are fp32): checked against the whole-sequence gradients at P = 8
// in tests/test_gpu_ring_loopback.py.
// ---------------------------------------------------------------------------------------------------
namespace {
struct BwdPlan {
    size_t shard, acc;                 // bytes of one bf16 shard / of one fp32 running sum
    size_t off_kv, off_ws, off_tq, off_tmp, off_rcv, off_acc, total, ws_bytes;
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
    p.off_tq = off;  off += p.shard;                                       // this step's dQ piece (bf16)
    p.off_tmp = off; off += 4 * p.shard;                                   // this step's dK, dV pieces: [parity][dK|dV]
    p.off_rcv = off; off += 4 * p.shard;                                   // received dK, dV pieces:    [parity][dK|dV]
    p.off_acc = off; off += 3 * p.acc;                                     // fp32 running sums
    p.total = off;
    return p;
}
}  // namespace

static int ring_backward_impl(fa2_ring_ctx* c,
                              const void* Q_local, const void* K_local, const void* V_local,
                              const void* O_local, const float* L_local, const void* dO_local,
                              void* dQ_local, void* dK_local, void* dV_local,
                              int B, int H, int total_seq_len, int local_seq_len, int head_dim,
                              float softmax_scale, int dtype, int causal,
                              void* workspace, size_t workspace_bytes, void* stream)
{
    if (!c || !Q_local || !K_local || !V_local || !O_local || !L_local || !dO_local || !dQ_local || !dK_local || !dV_local)
        return FA2_ERR_NULL_POINTER;
    if (B <= 0 || H <= 0 || local_seq_len <= 0 || head_dim <= 0) return FA2_ERR_INVALID_SHAPE;
    const int P = c->nranks, rank = c->rank;
    if ((long long)local_seq_len * P != (long long)total_seq_len) return FA2_ERR_INVALID_SHAPE;
    if (dtype != FA2_DTYPE_BF16) return FA2_ERR_UNSUPPORTED_DTYPE;
    if (causal && (local_seq_len & 1)) return FA2_ERR_INVALID_SHAPE;
    const BwdPlan pl = make_bwd_plan(B, H, local_seq_len, head_dim, P);
    if (!workspace || workspace_bytes < pl.total) return FA2_ERR_WORKSPACE;

    const Dev dev{c};
    const fa2_ring_backend& be = c->be;
    char* ws = (char*)workspace;
    if (P == 1) {
        // one rank: no shard to fetch, no piece to send, nothing to sum -- the block kernels write the gradients themselves
        // (round 3 ran the fp32 sum and convert passes here too: 0.15 ms of 1.4 at B = 1, H = 16, N = 8192)
        return be.backward_block(be.user, Q_local, K_local, V_local, O_local, L_local, dO_local, dQ_local, dK_local, dV_local, B, H,
                                 local_seq_len, local_seq_len, head_dim, softmax_scale, dtype, 0, 0, 0, causal ? 1 : 0, 0,
                                 ws + pl.off_ws, pl.ws_bytes, stream, 7);
    }
    auto slotK = [&](int i) { return (void*)(ws + pl.off_kv + (size_t)i * 2 * pl.shard); };
    auto slotV = [&](int i) { return (void*)(ws + pl.off_kv + (size_t)i * 2 * pl.shard + pl.shard); };
    void* bws = ws + pl.off_ws;
    void* tq = ws + pl.off_tq;
    auto tk = [&](int par) { return (void*)(ws + pl.off_tmp + (size_t)par * 2 * pl.shard); };
    auto tv = [&](int par) { return (void*)(ws + pl.off_tmp + (size_t)par * 2 * pl.shard + pl.shard); };
    auto rk = [&](int par) { return (void*)(ws + pl.off_rcv + (size_t)par * 2 * pl.shard); };
    auto rv = [&](int par) { return (void*)(ws + pl.off_rcv + (size_t)par * 2 * pl.shard + pl.shard); };
    float* aq = (float*)(ws + pl.off_acc); float* ak = (float*)(ws + pl.off_acc + pl.acc); float* av = (float*)(ws + pl.off_acc + 2 * pl.acc);
    const size_t BH = (size_t)B * H;
    const size_t slab = (size_t)local_seq_len * head_dim;         // elements of one head
    const size_t elems = BH * slab;
    const size_t bytes = elems * 2;
    const int half = local_seq_len / 2;

    // K/V shards from their owners (two grouped exchanges, as in the mesh forward)
    if (P > 1) {
        RING_TRY(dev.record(c->ev_in, stream));
        RING_TRY(dev.wait(c->comm_stream, c->ev_in));
        for (int part = 0; part < 2; ++part) {
            const int lo = part == 0 ? 1 : 2, hi = part == 0 ? 1 : P - 1;
            if (lo > hi) continue;
            RING_TRY(dev.grouped([&] {
                int st = FA2_OK;
                for (int s = lo; s <= hi && !st; ++s)
                    st = dev.exchange_pair(K_local, V_local, slotK(s - 1), slotV(s - 1), bytes, (rank + s) % P, (rank - s + P) % P,
                                           c->comm_stream);
                return st;
            }));
            RING_TRY(dev.record(c->ev_recv[part], c->comm_stream));
        }
    }
    // D = rowsum(dO o O) and the row constants, once: they depend on local rows only
    RING_TRY(be.backward_block(be.user, Q_local, K_local, V_local, O_local, L_local, dO_local, tq, tk(0), tv(0), B, H, local_seq_len,
                               local_seq_len, head_dim, softmax_scale, dtype, 0, 0, 0, 0, 0, bws, pl.ws_bytes, stream, 1));

    // The two main kernels on the block (local rows) x (owner's keys); *rows_q / *rows_k: how many rows of every
    // head of the dQ / dK, dV pieces it defines, and where the dQ rows start.
    // with more than one rank the exchange of the previous step's pieces runs beside the block kernels: the single-kernel
    // form then leaves a few CUs to RCCL (fa2_mi355x.h: FA2_PHASE_LEAVE_ROOM)
    const int both = 6 | (P > 1 ? FA2_PHASE_LEAVE_CUS(c->reserve_cus) : 0);
    auto block = [&](const void* Kc, const void* Vc, int owner, int par, int* q0, int* nq, int* nk) -> int {
        if (!causal || owner == rank) {
            *q0 = 0; *nq = local_seq_len; *nk = local_seq_len;
            return be.backward_block(be.user, Q_local, Kc, Vc, O_local, L_local, dO_local, tq, tk(par), tv(par), B, H, local_seq_len,
                                     local_seq_len, head_dim, softmax_scale, dtype, 0, 0, 0, causal ? 1 : 0, 0, bws, pl.ws_bytes,
                                     stream, both);
        }
        if (owner < rank) {           // the owner's first chunk of keys, every local row
            *q0 = 0; *nq = local_seq_len; *nk = half;
            return be.backward_block(be.user, Q_local, Kc, Vc, O_local, L_local, dO_local, tq, tk(par), tv(par), B, H, local_seq_len,
                                     half, head_dim, softmax_scale, dtype, 0, local_seq_len, 0, 0, 0, bws, pl.ws_bytes, stream, both);
        }
        // all of the owner's keys, the rows of the second local chunk
        *q0 = half; *nq = half; *nk = local_seq_len;
        const size_t eo = (size_t)half * head_dim;
        return be.backward_block(be.user, (const char*)Q_local + eo * 2, Kc, Vc, (const char*)O_local + eo * 2, L_local + half,
                                 (const char*)dO_local + eo * 2, (char*)tq + eo * 2, tk(par), tv(par), B, H, half, local_seq_len,
                                 head_dim, softmax_scale, dtype, local_seq_len, 0, half, 0, 0, bws, pl.ws_bytes, stream, both);
    };
    // acc (+)= the first `nrows` rows of every head of a bf16 piece, starting at row r0
    auto add_rows = [&](float* acc, const void* src, int r0, int nrows, int init) -> int {
        const size_t off = (size_t)r0 * head_dim;
        if (nrows == local_seq_len) return be.accumulate_bf16_2d(be.user, acc, src, 1, elems, elems, init, stream);
        return be.accumulate_bf16_2d(be.user, acc + off, (const char*)src + off * 2, BH, (size_t)nrows * head_dim, slab, init, stream);
    };

    for (int s = 0; s < P; ++s) {
        const int par = s & 1;
        const int owner = (rank - s + P) % P;
        const void* Kc = s == 0 ? K_local : slotK(s - 1);
        const void* Vc = s == 0 ? V_local : slotV(s - 1);
        if (s == 1) RING_TRY(dev.wait(stream, c->ev_recv[0]));
        if (s == 2) RING_TRY(dev.wait(stream, c->ev_recv[1]));
        int q0 = 0, nq = 0, nk = 0;
        RING_TRY(block(Kc, Vc, owner, par, &q0, &nq, &nk));
        // dQ piece -> own running sum.  Step 0 defines every row (init); later pieces cover all rows or the second half.
        RING_TRY(s == 0 ? add_rows(aq, tq, 0, local_seq_len, 1) : add_rows(aq, tq, q0, nq, 0));
        if (s == 0) {                                // own keys: the pieces stay here
            RING_TRY(add_rows(ak, tk(par), 0, local_seq_len, 1));
            RING_TRY(add_rows(av, tv(par), 0, local_seq_len, 1));
        } else {
            RING_TRY(dev.record(c->ev_comp[s], stream));
        }
        // deferred: the pieces of step s-1 that came in while this step's kernels ran
        auto add_received = [&](int t) -> int {
            const int from = (rank + t) % P;          // it computed on MY keys; its case: me < from -> first chunk only
            const int rows = (causal && rank < from) ? half : local_seq_len;
            RING_TRY(dev.wait(stream, c->ev_x[t]));
            RING_TRY(add_rows(ak, rk(t & 1), 0, rows, 0));
            return add_rows(av, rv(t & 1), 0, rows, 0);
        };
        if (s >= 2) RING_TRY(add_received(s - 1));
        if (s >= 1) {
            // gradients of owner (rank - s)'s keys go to it; those of mine computed by rank + s come in
            RING_TRY(dev.wait(c->comm_stream, c->ev_comp[s]));
            RING_TRY(dev.grouped([&] {
                return dev.exchange_pair(tk(par), tv(par), rk(par), rv(par), bytes, (rank - s + P) % P, (rank + s) % P, c->comm_stream);
            }));
            RING_TRY(dev.record(c->ev_x[s], c->comm_stream));
        }
        if (s == P - 1 && s >= 1) RING_TRY(add_received(s));
    }
    RING_TRY(be.convert_f32_to_bf16(be.user, aq, dQ_local, elems, stream));
    RING_TRY(be.convert_f32_to_bf16(be.user, ak, dK_local, elems, stream));
    return be.convert_f32_to_bf16(be.user, av, dV_local, elems, stream);
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

size_t fa2_ring_backward_workspace_bytes(int B, int H, int local_seq_len, int head_dim, int dtype, int nranks)
{
    if (B <= 0 || H <= 0 || local_seq_len <= 0 || head_dim <= 0 || nranks <= 0 || dtype != FA2_DTYPE_BF16) return 0;
    return make_bwd_plan(B, H, local_seq_len, head_dim, nranks).total;
}

int fa2_ring_backward_block_workspace(int B, int H, int local_seq_len, int head_dim, int dtype, int nranks,
                                      size_t* offset, size_t* bytes)
{
    if (!offset || !bytes) return FA2_ERR_NULL_POINTER;
    if (B <= 0 || H <= 0 || local_seq_len <= 0 || head_dim <= 0 || nranks <= 0) return FA2_ERR_INVALID_SHAPE;
    if (dtype != FA2_DTYPE_BF16) return FA2_ERR_UNSUPPORTED_DTYPE;
    const BwdPlan pl = make_bwd_plan(B, H, local_seq_len, head_dim, nranks);
    *offset = pl.off_ws;
    *bytes = pl.ws_bytes;
    return FA2_OK;
}

int fa2_ring_attention_backward(fa2_ring_ctx* c,
                                const void* Q_local, const void* K_local, const void* V_local,
                                const void* O_local, const float* L_local, const void* dO_local,
                                void* dQ_local, void* dK_local, void* dV_local,
                                int B, int H, int total_seq_len, int local_seq_len, int head_dim,
                                float softmax_scale, int dtype,
                                void* workspace, size_t workspace_bytes, void* stream)
{
    return ring_backward_impl(c, Q_local, K_local, V_local, O_local, L_local, dO_local, dQ_local, dK_local, dV_local, B, H,
                              total_seq_len, local_seq_len, head_dim, softmax_scale, dtype, 0, workspace, workspace_bytes, stream);
}

int fa2_ring_attention_backward_causal(fa2_ring_ctx* c,
                                       const void* Q_local, const void* K_local, const void* V_local,
                                       const void* O_local, const float* L_local, const void* dO_local,
                                       void* dQ_local, void* dK_local, void* dV_local,
                                       int B, int H, int total_seq_len, int local_seq_len, int head_dim,
                                       float softmax_scale, int dtype,
                                       void* workspace, size_t workspace_bytes, void* stream)
{
    return ring_backward_impl(c, Q_local, K_local, V_local, O_local, L_local, dO_local, dQ_local, dK_local, dV_local, B, H,
                              total_seq_len, local_seq_len, head_dim, softmax_scale, dtype, 1, workspace, workspace_bytes, stream);
}

}  // extern "C"
