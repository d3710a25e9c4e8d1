// ring_loopback.cpp -- TEST transport for libfa2_ring_mi355x.so: P ranks as P host threads on ONE GPU.
//
// Everything except the transport is the product backend (fa2_ring_default_backend: real HIP streams and
// events, the real step / backward kernels); send/recv become hipMemcpyAsync between the ranks' buffers,
// ordered with events exactly where RCCL orders them: a transfer starts once BOTH the sender's and the
// receiver's streams have reached the grouped operation, and neither stream goes past it before the bytes
// have moved.  So fa2_ring_attention_forward / _causal / _backward run unmodified at P = 2, 4, 8 with their
// own comm streams, slots and fences -- the reference's 2-rank MPI run (04_ring_attention.cu:9-153) without MPI.
//
// Matching follows RCCL: per (source, destination) pair in issue order.  group_end blocks the calling host
// thread until its partners have posted (every rank must reach the matching group, as with RCCL).
#include "../../include/fa2_ring_mi355x.h"

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {

struct Op { bool is_send; const void* src; void* dst; size_t bytes; int peer; hipStream_t stream; };

struct Posted { const void* src = nullptr; size_t bytes = 0; hipEvent_t ready = nullptr; hipEvent_t done = nullptr; bool has_send = false, has_done = false; };

}  // namespace

struct lb_world {
    int nranks;
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::tuple<int, int, long>, Posted> box;          // (src, dst, seq)
    std::map<std::pair<int, int>, long> send_seq, recv_seq;
    std::vector<hipEvent_t> events;                            // destroyed with the world
    struct Rank { lb_world* w; int rank; bool in_group = false; std::vector<Op> ops; };
    std::vector<Rank> ranks;
    int failed = 0;
};

namespace {

int lb_flush(lb_world::Rank* r)
{
    lb_world* w = r->w;
    std::vector<long> seqs(r->ops.size());
    // 1. post every send (never blocks)
    for (size_t i = 0; i < r->ops.size(); ++i) {
        const Op& op = r->ops[i];
        if (!op.is_send) continue;
        hipEvent_t ready;
        if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) return FA2_ERR_HIP_BASE;
        if (hipEventRecord(ready, op.stream) != hipSuccess) return FA2_ERR_HIP_BASE;
        std::lock_guard<std::mutex> g(w->mu);
        const long seq = w->send_seq[{r->rank, op.peer}]++;
        seqs[i] = seq;
        Posted& p = w->box[{r->rank, op.peer, seq}];
        p.src = op.src; p.bytes = op.bytes; p.ready = ready; p.has_send = true;
        w->events.push_back(ready);
        w->cv.notify_all();
    }
    // 2. every receive: wait for the partner's post, copy on MY stream behind its "ready"
    for (size_t i = 0; i < r->ops.size(); ++i) {
        const Op& op = r->ops[i];
        if (op.is_send) continue;
        Posted p;
        long seq;
        {
            std::unique_lock<std::mutex> g(w->mu);
            seq = w->recv_seq[{op.peer, r->rank}]++;
            w->cv.wait(g, [&] { return w->failed || w->box[{op.peer, r->rank, seq}].has_send; });
            if (w->failed) return FA2_ERR_UNSUPPORTED;
            p = w->box[{op.peer, r->rank, seq}];
        }
        if (p.bytes != op.bytes) { std::lock_guard<std::mutex> g(w->mu); w->failed = 1; w->cv.notify_all(); return FA2_ERR_INVALID_SHAPE; }
        hipEvent_t done;
        if (hipStreamWaitEvent(op.stream, p.ready, 0) != hipSuccess) return FA2_ERR_HIP_BASE;
        if (hipMemcpyAsync(op.dst, p.src, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) return FA2_ERR_HIP_BASE;
        if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return FA2_ERR_HIP_BASE;
        if (hipEventRecord(done, op.stream) != hipSuccess) return FA2_ERR_HIP_BASE;
        std::lock_guard<std::mutex> g(w->mu);
        Posted& q = w->box[{op.peer, r->rank, seq}];
        q.done = done; q.has_done = true;
        w->events.push_back(done);
        w->cv.notify_all();
    }
    // 3. every send: my stream may not run past the group before the partner's copy has read the buffer
    for (size_t i = 0; i < r->ops.size(); ++i) {
        const Op& op = r->ops[i];
        if (!op.is_send) continue;
        hipEvent_t done;
        {
            std::unique_lock<std::mutex> g(w->mu);
            w->cv.wait(g, [&] { return w->failed || w->box[{r->rank, op.peer, seqs[i]}].has_done; });
            if (w->failed) return FA2_ERR_UNSUPPORTED;
            done = w->box[{r->rank, op.peer, seqs[i]}].done;
            w->box.erase({r->rank, op.peer, seqs[i]});
        }
        if (hipStreamWaitEvent(op.stream, done, 0) != hipSuccess) return FA2_ERR_HIP_BASE;
    }
    r->ops.clear();
    return FA2_OK;
}

int lb_group_start(void* user) { ((lb_world::Rank*)user)->in_group = true; return FA2_OK; }
int lb_group_end(void* user)
{
    lb_world::Rank* r = (lb_world::Rank*)user;
    r->in_group = false;
    return lb_flush(r);
}
int lb_send(void* user, const void* buf, size_t bytes, int peer, void* stream)
{
    lb_world::Rank* r = (lb_world::Rank*)user;
    if (peer < 0 || peer >= r->w->nranks || peer == r->rank) return FA2_ERR_INVALID_SHAPE;
    r->ops.push_back({true, buf, nullptr, bytes, peer, (hipStream_t)stream});
    return r->in_group ? FA2_OK : lb_flush(r);
}
int lb_recv(void* user, void* buf, size_t bytes, int peer, void* stream)
{
    lb_world::Rank* r = (lb_world::Rank*)user;
    if (peer < 0 || peer >= r->w->nranks || peer == r->rank) return FA2_ERR_INVALID_SHAPE;
    r->ops.push_back({false, nullptr, buf, bytes, peer, (hipStream_t)stream});
    return r->in_group ? FA2_OK : lb_flush(r);
}

}  // namespace

extern "C" {

lb_world* lb_world_create(int nranks)
{
    if (nranks < 1 || nranks > 64) return nullptr;
    lb_world* w = new lb_world();
    w->nranks = nranks;
    w->ranks.resize(nranks);
    for (int r = 0; r < nranks; ++r) { w->ranks[r].w = w; w->ranks[r].rank = r; }
    return w;
}

// Unblocks every thread waiting in a group (a rank that failed elsewhere calls this so its partners do not hang).
void lb_world_abort(lb_world* w)
{
    std::lock_guard<std::mutex> g(w->mu);
    w->failed = 1;
    w->cv.notify_all();
}

void lb_world_destroy(lb_world* w)
{
    if (!w) return;
    (void)hipDeviceSynchronize();
    for (hipEvent_t e : w->events) (void)hipEventDestroy(e);
    delete w;
}

// The product backend with the transport entries replaced by the loopback of `rank`.
int lb_backend(lb_world* w, int rank, fa2_ring_backend* out)
{
    if (!w || !out || rank < 0 || rank >= w->nranks) return FA2_ERR_INVALID_SHAPE;
    int st = fa2_ring_default_backend(out);
    if (st) return st;
    out->user = &w->ranks[rank];
    out->group_start = lb_group_start;
    out->send = lb_send;
    out->recv = lb_recv;
    out->group_end = lb_group_end;
    return FA2_OK;
}

}  // extern "C"
