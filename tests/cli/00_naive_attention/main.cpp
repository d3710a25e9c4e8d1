// 00_naive_attention -- the CPU path: self test of the naive attention on the reference's
// literal 2x2 case (reference src/00_naive_attention/main.cpp:40-85, same verdict strings and
// exit code), and with arguments `B H N d [causal [iters]]` the naive CPU forward timed on the
// host cores (one head per task, std::thread over heads, core count printed) -- the
// "00_naive beside it" number of the bench table.
#include <chrono>
#include <iostream>

#include "../common/harness.h"

static int self_test()
{
    const int N = 2, d = 2;
    std::vector<float> Q = {1.0f, 0.0f, 0.0f, 1.0f};
    std::vector<float> K = Q;
    std::vector<float> V = {1.0f, 2.0f, 3.0f, 4.0f};
    std::vector<float> O(N * d, 0.0f);
    oracle_naive_attention(Q.data(), K.data(), V.data(), O.data(), N, d);
    const std::vector<float> expected = {1.6604769f, 2.6604770f, 2.3395231f, 3.3395231f};
    bool passed = true;
    for (size_t i = 0; i < O.size(); ++i)
        if (std::abs(O[i] - expected[i]) > 1e-4f) {
            std::cerr << "Mismatch at index " << i << ": got " << O[i] << ", expected " << expected[i] << '\n';
            passed = false;
        }
    if (passed) {
        std::cout << "naive_attention test passed. Output:" << std::endl;
        for (int i = 0; i < N; ++i) {
            std::cout << "Row " << i << ": ";
            for (int j = 0; j < d; ++j) std::cout << O[i * d + j] << (j + 1 == d ? '\n' : ' ');
        }
        return 0;
    }
    std::cerr << "naive_attention test failed." << std::endl;
    return 1;
}

int main(int argc, char** argv)
{
    harness::Shape s = harness::parse_shape(argc, argv);
    if (!s.given) return self_test();

    // timed CPU naive forward: one head per task
    const int BH = s.B * s.H;
    unsigned cores = std::thread::hardware_concurrency();
    if (cores == 0) cores = 1;
    const int tasks = std::min<int>(BH, (int)cores);        // a stated subset: one head per core
    const size_t n = (size_t)s.N * s.d;
    std::vector<std::vector<float>> Q(tasks), K(tasks), V(tasks), O(tasks);
    for (int t = 0; t < tasks; ++t) {
        harness::fill_uniform(Q[t], n, 1 + 10 * t, 1.0f);
        harness::fill_uniform(K[t], n, 2 + 10 * t, 1.0f);
        harness::fill_uniform(V[t], n, 3 + 10 * t, 1.0f);
        O[t].resize(n);
    }
    const float scale = 1.0f / std::sqrt((float)s.d);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < tasks; ++t)
        th.emplace_back([&, t] { oracle_naive_forward_pass(Q[t].data(), K[t].data(), V[t].data(), O[t].data(), nullptr, s.N, s.d, scale); });
    for (auto& x : th) x.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double flops = 4.0 * s.N * (double)s.N * s.d * tasks;
    printf("00_naive CPU forward: %d of %d heads (N=%d, d=%d) on %d threads of %u cores: %.3f s, %.2f GFLOP/s\n",
           tasks, BH, s.N, s.d, tasks, cores, sec, flops / sec / 1e9);
    printf("extrapolated to all %d heads at this rate: %.1f s\n", BH, sec * BH / tasks);
    return 0;
}
