// 02_overlap -- RCCL/xGMI bring-up diagnostics for the ring path:
//
//   1. comm/compute overlap (reference src/03_flash_attention_v2_ring/02_overlap.cu:63-92): per
//      ring step a K/V shard exchange with the ring neighbours and one attention step on the
//      resident shard.  The reference issues its exchange on the default stream (:68), so its two
//      "streams" never run concurrently; here the exchange really is enqueued on a comm stream of
//      its own (fa2_ring_exchange_kv) and the step kernel (fa2_forward, bf16) on a compute stream.
//      Reported per rank: exchange alone, compute alone, both together, and how much of the shorter
//      one was hidden.  Every step verifies the token that travelled with the shard, like the
//      reference's "Received block starting with ..." line (:86-90).
//   2. per-distance link probe: every rank sends `bytes` to rank + dist and receives from rank - dist
//      at the same time, for dist = 1 .. P-1 -- on the xGMI full mesh each distance uses a different
//      link of every GPU, so this is the per-link bandwidth the mesh schedule of the ring relies on.
//
// Single process, one host thread per GPU, ncclCommInitAll instead of MPI.
// Usage: 02_overlap [nranks [H Nlocal d [iters]]]   (defaults: all GPUs, H=16, Nlocal=2048, d=128, 5)
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <iostream>
#include <mutex>

#include "../../../include/fa2_ring_mi355x.h"
#include "../common/harness.h"

#define CHECK_RCCL(call)                                                                     \
    do {                                                                                     \
        ncclResult_t r_ = (call);                                                            \
        if (r_ != ncclSuccess) {                                                             \
            fprintf(stderr, "RCCL error at %s:%d - %s\n", __FILE__, __LINE__, ncclGetErrorString(r_)); \
            exit(1);                                                                         \
        }                                                                                    \
    } while (0)

struct Barrier {      // host-side rendezvous of the rank threads
    explicit Barrier(int n) : n_(n) {}
    void wait()
    {
        const int gen = gen_.load();
        if (count_.fetch_add(1) + 1 == n_) { count_.store(0); gen_.fetch_add(1); }
        else while (gen_.load() == gen) std::this_thread::yield();
    }
    int n_;
    std::atomic<int> count_{0}, gen_{0};
};

int main(int argc, char** argv)
{
    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    const int P = argc > 1 ? atoi(argv[1]) : ndev;
    const int H = argc > 2 ? atoi(argv[2]) : 16;
    const int Nl = argc > 3 ? atoi(argv[3]) : 2048;
    const int d = argc > 4 ? atoi(argv[4]) : 128;
    const int iters = argc > 5 ? atoi(argv[5]) : 5;
    if (P < 1 || P > ndev || (d != 64 && d != 128) || Nl < 1 || H < 1 || iters < 1) {
        fprintf(stderr, "usage: %s [nranks<=%d [H Nlocal d(64|128) [iters]]]\n", argv[0], ndev);
        return 2;
    }
    std::vector<ncclComm_t> comms(P);
    std::vector<int> devs(P);
    for (int i = 0; i < P; ++i) devs[i] = i;
    CHECK_RCCL(ncclCommInitAll(comms.data(), P, devs.data()));

    const size_t elems = (size_t)H * Nl * d;
    const size_t bytes = elems * 2;                              // one bf16 K (or V) shard
    const double step_flops = 4.0 * H * (double)Nl * Nl * d;     // one ring step of the forward
    printf("Ring overlap test: %d rank(s), shard H=%d Nlocal=%d d=%d bf16 = %.1f MiB K + %.1f MiB V per step\n", P, H, Nl, d,
           bytes / 1048576.0, bytes / 1048576.0);

    Barrier bar(P);
    std::mutex io;
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int rank = 0; rank < P; ++rank)
        th.emplace_back([&, rank] {
            CHECK_HIP(hipSetDevice(rank));
            fa2_ring_ctx* ctx = nullptr;
            CHECK_FA2(fa2_ring_ctx_create_from_comm(&ctx, comms[rank], rank, P));
            hipStream_t compute, comm;
            CHECK_HIP(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
            CHECK_HIP(hipStreamCreateWithFlags(&comm, hipStreamNonBlocking));

            // K/V shards: element 0 carries the owner's token (rank + 1), the rest small noise
            std::vector<float> f(elems);
            std::vector<uint16_t> hq, hk;
            harness::fill_uniform(f, elems, 11 + rank, 1.0f);
            harness::to_bf16(f, hq);
            f[0] = rank + 1.0f;
            harness::to_bf16(f, hk);
            harness::DevBuf<uint16_t> Q(elems), Kc(elems), Vc(elems), Kn(elems), Vn(elems), O(elems);
            harness::DevBuf<float> L((size_t)H * Nl);
            Q.up(hq.data()); Kc.up(hk.data()); Vc.up(hk.data());
            const float scale = 1.0f / std::sqrt((float)d);
            uint16_t* kc = Kc.p; uint16_t* vc = Vc.p; uint16_t* kn = Kn.p; uint16_t* vn = Vn.p;

            auto exchange = [&] { CHECK_FA2(fa2_ring_exchange_kv(ctx, kc, kn, vc, vn, bytes, comm)); };
            auto compute_step = [&] { CHECK_FA2(fa2_forward(Q.p, kc, vc, O.p, L.p, 1, H, Nl, d, scale, FA2_DTYPE_BF16, 0, compute)); };
            auto both_done = [&] { CHECK_HIP(hipStreamSynchronize(comm)); CHECK_HIP(hipStreamSynchronize(compute)); };
            auto wall = [&](auto&& fn) {
                bar.wait();
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < iters; ++i) { fn(); both_done(); }
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
                bar.wait();
                return ms;
            };
            compute_step(); exchange(); both_done();                       // warm-up (RCCL connects lazily)
            const double t_comm = wall([&] { exchange(); });
            const double t_comp = wall([&] { compute_step(); });
            const double t_both = wall([&] { exchange(); compute_step(); });
            const double hidden = (t_comm + t_comp - t_both) / std::min(t_comm, t_comp);

            // the ring itself: P steps, the token must come back home
            for (int step = 0; step < P; ++step) {
                exchange();
                compute_step();
                both_done();
                std::swap(kc, kn); std::swap(vc, vn);
                uint16_t tok = 0;
                CHECK_HIP(hipMemcpy(&tok, kc, 2, hipMemcpyDeviceToHost));
                const int from = ((rank - step - 1) % P + P) % P;
                std::lock_guard<std::mutex> g(io);
                printf("Rank %d, Step %d: Received block starting with %g (expected %d from rank %d)\n", rank, step,
                       harness::bf2f(tok), from + 1, from);
                if (harness::bf2f(tok) != from + 1.0f) ++bad;
            }
            {
                std::lock_guard<std::mutex> g(io);
                printf("Rank %d: exchange %.3f ms (%.1f GB/s out), step kernel %.3f ms (%.0f TFLOP/s), together %.3f ms -> %.0f%% of the shorter hidden\n",
                       rank, t_comm, 2.0 * bytes / t_comm / 1e6, t_comp, step_flops / t_comp / 1e9, t_both, 100.0 * hidden);
            }

            // per-distance link probe
            for (int dist = 1; dist < P; ++dist) {
                const int to = (rank + dist) % P, from = (rank - dist + P) % P;
                auto xfer = [&] {
                    CHECK_RCCL(ncclGroupStart());
                    CHECK_RCCL(ncclSend(kc, bytes, ncclChar, to, comms[rank], comm));
                    CHECK_RCCL(ncclRecv(kn, bytes, ncclChar, from, comms[rank], comm));
                    CHECK_RCCL(ncclGroupEnd());
                };
                xfer(); both_done();
                const double ms = wall(xfer);
                std::lock_guard<std::mutex> g(io);
                printf("Rank %d -> rank %d (distance %d): %.1f MiB in %.3f ms = %.1f GB/s per direction\n", rank, to, dist,
                       bytes / 1048576.0, ms, bytes / ms / 1e6);
            }
            CHECK_HIP(hipStreamDestroy(compute));
            CHECK_HIP(hipStreamDestroy(comm));
            CHECK_FA2(fa2_ring_ctx_destroy(ctx));
        });
    for (auto& t : th) t.join();
    for (auto c : comms) ncclCommDestroy(c);
    std::cout << (bad == 0 ? "Overlap test completed!" : "Overlap test FAILED") << " (" << P << " GPU(s))" << std::endl;
    return bad == 0 ? 0 : 1;
}
