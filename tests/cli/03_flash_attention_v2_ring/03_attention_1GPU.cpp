// 03_attention_1GPU -- the reference's step 3 of the ring staircase (src/03_flash_attention_v2_ring/03_attention_1GPU.cu:
// 1-100, run.sh:21-23): before anything is distributed, the single-GPU flash_attention_2_forward is checked against the
// naive attention on the ring test's own data -- create_simple_test_data, N = 5096, d = 64, scale = 1
// (:17-19, :30) -- with compare_outputs (:69), and the shard sizes the distributed step will use are announced
// (:87-91).  Same flow and verdict strings here; the forward is the reference-signature drop-in of
// libfa2_mi355x.so (fp32, one head), the naive side is the CPU restatement the tests use as their checker.
//   no arguments : the reference's case;     N d : the same flow at another size.
#include <iostream>

#include "../../../include/fa2_mi355x.h"
#include "../common/harness.h"

using namespace harness;

int main(int argc, char** argv)
{
    int seq_len = 5096, head_dim = 64;
    if (argc == 3) { seq_len = atoi(argv[1]); head_dim = atoi(argv[2]); }
    const float softmax_scale = 1.0f;
    int nranks = 0;
    CHECK_HIP(hipGetDeviceCount(&nranks));

    std::cout << "=== Test Case ===" << std::endl;
    std::vector<float> h_Q, h_K, h_V;
    create_simple_test_data(h_Q, h_K, h_V, seq_len, head_dim);
    print_matrix("Q", h_Q.data(), seq_len, head_dim);
    print_matrix("K", h_K.data(), seq_len, head_dim);
    print_matrix("V", h_V.data(), seq_len, head_dim);

    std::cout << "\n=== Naive Attention ===" << std::endl;
    const size_t n = (size_t)seq_len * head_dim;
    std::vector<float> h_O_naive(n), h_O_flash(n);
    oracle_naive_forward_pass(h_Q.data(), h_K.data(), h_V.data(), h_O_naive.data(), nullptr, seq_len, head_dim, softmax_scale);
    print_matrix("O_naive", h_O_naive.data(), seq_len, head_dim);

    std::cout << "\n=== FlashAttention (Single GPU) ===" << std::endl;
    DevBuf<float> q(n), k(n), v(n), o(n), l((size_t)seq_len);
    q.up(h_Q.data()); k.up(h_K.data()); v.up(h_V.data());
    CHECK_FA2(flash_attention_2_forward(q.p, k.p, v.p, o.p, l.p, seq_len, head_dim, softmax_scale));
    CHECK_HIP(hipDeviceSynchronize());
    o.down(h_O_flash.data());
    print_matrix("O_flash", h_O_flash.data(), seq_len, head_dim);

    std::cout << "\n=== Comparison: Naive vs FlashAttention ===" << std::endl;
    const bool match = compare_outputs(h_O_naive.data(), h_O_flash.data(), n);
    if (!match) std::cout << "WARNING: FlashAttention output doesn't match naive!" << std::endl;

    std::cout << "\n=== Ring Attention (Distributed) ===" << std::endl;
    std::cout << "Running on " << nranks << " GPUs" << std::endl;
    std::cout << "Each GPU processes " << seq_len / (nranks > 0 ? nranks : 1) << " rows of Q" << std::endl;
    std::cout << (match ? "Test PASSED!" : "Test FAILED!") << std::endl;
    return match ? 0 : 1;
}
