// 04_ring_attention -- test main of the sequence-sharded ring forward.
//   no arguments : the reference's case (src/03_flash_attention_v2_ring/04_ring_attention.cu):
//                  create_simple_test_data, N=5096, d=64, scale=1, each rank a contiguous shard
//                  of rows, ring_attention_forward (the fp32 reference-signature drop-in), gather
//                  in rank order, compare_outputs(rtol=5e-3, atol=1.0), "Test PASSED!" /
//                  "Test FAILED!".  With one visible GPU this is 03_attention_1GPU.cu's case.
//   B H N d [schedule [iters [nranks]]] : bf16 ring forward, N = total sequence length,
//                  schedule 0 = relay (reference), 1 = mesh; parity against the single-GPU
//                  forward on the same inputs (rel-L2 <= 1e-3) and timing.
// One process, one host thread per GPU, ncclCommInitAll (no MPI).
#include <rccl/rccl.h>

#include <iostream>

#include "../../../include/fa2_ring_mi355x.h"
#include "../common/harness.h"

using namespace harness;

static int pick_ranks(int want, int N)
{
    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    int P = want > 0 ? std::min(want, ndev) : ndev;
    while (P > 1 && N % P != 0) --P;      // "seq_len must be divisible by nranks!" (04_ring_attention.cu:55-63)
    return std::max(P, 1);
}

static int reference_case()
{
    const int N = 5096, d = 64;
    const float scale = 1.0f;
    const int P = pick_ranks(0, N);
    std::cout << "=== Test Case ===" << std::endl;
    std::vector<float> Q, K, V;
    create_simple_test_data(Q, K, V, N, d);
    std::vector<float> O_naive((size_t)N * d);
    std::cout << "\n=== Naive Attention Reference ===" << std::endl;
    oracle_naive_forward_pass(Q.data(), K.data(), V.data(), O_naive.data(), nullptr, N, d, scale);
    print_matrix("O_naive", O_naive.data(), N, d);
    std::cout << "\n=== Ring Attention (Distributed) ===\nRunning on " << P << " GPUs\nEach GPU processes " << N / P << " rows of Q" << std::endl;

    std::vector<ncclComm_t> comms(P);
    std::vector<int> devs(P);
    for (int i = 0; i < P; ++i) devs[i] = i;
    if (ncclCommInitAll(comms.data(), P, devs.data()) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
    const int nl = N / P;
    std::vector<float> gathered((size_t)N * d);
    std::vector<std::thread> th;
    for (int rank = 0; rank < P; ++rank)
        th.emplace_back([&, rank] {
            CHECK_HIP(hipSetDevice(rank));
            const size_t n = (size_t)nl * d, off = (size_t)rank * n;
            printf("Rank %d: Q_local rows %d-%d\n", rank, rank * nl, rank * nl + nl - 1);
            DevBuf<float> q(n), k(n), v(n), o(n), l(nl);
            q.up(Q.data() + off); k.up(K.data() + off); v.up(V.data() + off);
            CHECK_FA2(ring_attention_forward(q.p, k.p, v.p, o.p, l.p, N, nl, d, scale, comms[rank], rank, P));
            o.down(gathered.data() + off);      // the reference's MPI_Gather: rank order = row order
        });
    for (auto& t : th) t.join();
    for (auto c : comms) ncclCommDestroy(c);

    std::cout << "\n=== Ring Attention Output ===" << std::endl;
    print_matrix("O_ring", gathered.data(), N, d);
    std::cout << "\n=== Final Comparison: Naive vs Ring Attention ===" << std::endl;
    const bool match = compare_outputs(O_naive.data(), gathered.data(), (size_t)N * d, 5e-3f);
    std::cout << (match ? "Test PASSED!" : "Test FAILED!") << std::endl;
    return match ? 0 : 1;
}

static int run_shape(int B, int H, int N, int d, int schedule, int iters, int want)
{
    print_device();
    const int P = pick_ranks(want, N);
    const int nl = N / P, BH = B * H;
    const float scale = 1.0f / sqrtf((float)d);
    const size_t E = (size_t)BH * N * d;
    std::vector<float> Q, K, V;
    std::vector<uint16_t> q16, k16, v16;
    fill_uniform(Q, E, 1, 1.0f); fill_uniform(K, E, 2, 1.0f); fill_uniform(V, E, 3, 1.0f);
    to_bf16(Q, q16); to_bf16(K, k16); to_bf16(V, v16);

    // single-GPU result on device 0 as the comparison (ring vs single-GPU on the same inputs)
    std::vector<uint16_t> o_single(E);
    {
        CHECK_HIP(hipSetDevice(0));
        DevBuf<uint16_t> q(E), k(E), v(E), o(E);
        DevBuf<float> l((size_t)BH * N);
        q.up(q16.data()); k.up(k16.data()); v.up(v16.data());
        CHECK_FA2(fa2_forward(q.p, k.p, v.p, o.p, l.p, B, H, N, d, scale, FA2_DTYPE_BF16, 0, nullptr));
        CHECK_HIP(hipDeviceSynchronize());
        o.down(o_single.data());
    }

    std::vector<ncclComm_t> comms(P);
    std::vector<int> devs(P);
    for (int i = 0; i < P; ++i) devs[i] = i;
    if (ncclCommInitAll(comms.data(), P, devs.data()) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
    std::vector<uint16_t> o_ring(E);
    std::vector<float> ms(P, 0.0f);
    std::vector<std::thread> th;
    for (int rank = 0; rank < P; ++rank)
        th.emplace_back([&, rank] {
            CHECK_HIP(hipSetDevice(rank));
            fa2_ring_ctx* ctx = nullptr;
            CHECK_FA2(fa2_ring_ctx_create_from_comm(&ctx, comms[rank], rank, P));
            const size_t n = (size_t)BH * nl * d;
            // local shard = rows [rank*nl, (rank+1)*nl) of every head
            std::vector<uint16_t> ql(n), kl(n), vl(n), ol(n);
            for (int bh = 0; bh < BH; ++bh) {
                const size_t src = ((size_t)bh * N + (size_t)rank * nl) * d, dst = (size_t)bh * nl * d;
                memcpy(&ql[dst], &q16[src], (size_t)nl * d * 2);
                memcpy(&kl[dst], &k16[src], (size_t)nl * d * 2);
                memcpy(&vl[dst], &v16[src], (size_t)nl * d * 2);
            }
            DevBuf<uint16_t> q(n), k(n), v(n), o(n);
            DevBuf<float> l((size_t)BH * nl);
            const size_t wsb = fa2_ring_workspace_bytes(B, H, nl, d, FA2_DTYPE_BF16, P, schedule);
            DevBuf<char> ws(wsb ? wsb : 256);
            q.up(ql.data()); k.up(kl.data()); v.up(vl.data());
            hipStream_t s;
            CHECK_HIP(hipStreamCreate(&s));
            auto run = [&] {
                CHECK_FA2(fa2_ring_attention_forward(ctx, q.p, k.p, v.p, o.p, l.p, B, H, N, nl, d, scale, FA2_DTYPE_BF16,
                                                     schedule, ws.p, wsb, s));
            };
            run();
            CHECK_HIP(hipStreamSynchronize(s));
            o.down(ol.data());
            for (int bh = 0; bh < BH; ++bh)
                memcpy(&o_ring[((size_t)bh * N + (size_t)rank * nl) * d], &ol[(size_t)bh * nl * d], (size_t)nl * d * 2);
            GpuTimer t;
            t.start(s);
            for (int i = 0; i < iters; ++i) run();
            ms[rank] = t.stop(s) / iters;
            CHECK_HIP(hipStreamDestroy(s));
            CHECK_FA2(fa2_ring_ctx_destroy(ctx));
        });
    for (auto& t : th) t.join();
    for (auto c : comms) ncclCommDestroy(c);

    std::vector<float> a(E), b(E);
    for (size_t i = 0; i < E; ++i) { a[i] = bf2f(o_ring[i]); b[i] = bf2f(o_single[i]); }
    const double r = rel_l2(a.data(), b.data(), E);
    printf("ring (%d GPUs, schedule %s) vs single-GPU forward: rel-L2 = %.3e (gate 1e-3)\n", P, schedule ? "mesh" : "relay", r);
    std::cout << (r <= 1e-3 ? "Test PASSED!" : "Test FAILED!") << std::endl;
    const float worst = *std::max_element(ms.begin(), ms.end());
    const double tf = 4.0 * BH * (double)N * N * d / (worst * 1e-3) / 1e12;
    printf("ring FA2 forward bf16 (B=%d,H=%d,N=%d,d=%d) on %d GPUs: %.3f ms, %.1f TFLOP/s total, %.1f%% of %d x MFMA peak\n",
           B, H, N, d, P, worst, tf, 100.0 * tf / (P * kPeakBf16Tflops), P);
    return r <= 1e-3 ? 0 : 1;
}

int main(int argc, char** argv)
{
    if (argc == 1) return reference_case();
    if (argc < 5) { fprintf(stderr, "usage: %s [B H N d [schedule [iters [nranks]]]]\n", argv[0]); return 2; }
    return run_shape(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argc > 5 ? atoi(argv[5]) : 0,
                     argc > 6 ? atoi(argv[6]) : 5, argc > 7 ? atoi(argv[7]) : 0);
}
