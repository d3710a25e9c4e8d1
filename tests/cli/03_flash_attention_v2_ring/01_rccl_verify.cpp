// 01_rccl_verify -- RCCL ring smoke test (reference src/03_flash_attention_v2_ring/
// 01_nccl_verify.cu:37-59): every rank passes a 16-float token to rank+1 P times through
// fa2_ring_exchange_kv (the ring_exchange primitive, nccl_utils.h:115-121); after P steps
// each rank must hold its own value again.  Single process, one host thread per GPU,
// ncclCommInitAll instead of MPI (there is no MPI on this platform and none is needed
// inside one node).  Usage: 01_rccl_verify [nranks]   (default: every visible GPU).
#include <rccl/rccl.h>

#include <atomic>
#include <iostream>

#include "../../../include/fa2_ring_mi355x.h"
#include "../common/harness.h"

int main(int argc, char** argv)
{
    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    int P = argc > 1 ? atoi(argv[1]) : ndev;
    if (P < 1 || P > ndev) { fprintf(stderr, "need 1..%d ranks\n", ndev); return 2; }
    std::vector<ncclComm_t> comms(P);
    std::vector<int> devs(P);
    for (int i = 0; i < P; ++i) devs[i] = i;
    if (ncclCommInitAll(comms.data(), P, devs.data()) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }

    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int rank = 0; rank < P; ++rank)
        th.emplace_back([&, rank] {
            CHECK_HIP(hipSetDevice(rank));
            fa2_ring_ctx* ctx = nullptr;
            CHECK_FA2(fa2_ring_ctx_create_from_comm(&ctx, comms[rank], rank, P));
            const int size = 16;
            harness::DevBuf<float> sa(size), ra(size), sb(size), rb(size);
            std::vector<float> h(size, rank + 1.0f), g(size, -(rank + 1.0f)), got(size);
            sa.up(h.data()); sb.up(g.data());
            hipStream_t s;
            CHECK_HIP(hipStreamCreate(&s));
            for (int step = 0; step < P; ++step) {
                if (step > 0) {
                    CHECK_HIP(hipMemcpyAsync(sa.p, ra.p, size * 4, hipMemcpyDeviceToDevice, s));
                    CHECK_HIP(hipMemcpyAsync(sb.p, rb.p, size * 4, hipMemcpyDeviceToDevice, s));
                }
                CHECK_FA2(fa2_ring_exchange_kv(ctx, sa.p, ra.p, sb.p, rb.p, size * 4, s));
                CHECK_HIP(hipStreamSynchronize(s));
                ra.down(got.data());
                printf("Rank %d, Step %d: Received %g from rank %d\n", rank, step, got[0], (rank - 1 + P) % P);
            }
            ra.down(got.data());
            if (got[0] != rank + 1.0f) ++bad;
            rb.down(got.data());
            if (got[0] != -(rank + 1.0f)) ++bad;
            CHECK_HIP(hipStreamDestroy(s));
            CHECK_FA2(fa2_ring_ctx_destroy(ctx));
        });
    for (auto& t : th) t.join();
    for (auto c : comms) ncclCommDestroy(c);
    std::cout << (bad == 0 ? "Ring verify PASSED" : "Ring verify FAILED") << " on " << P << " GPU(s)" << std::endl;
    return bad == 0 ? 0 : 1;
}
