// 02_flash_attention_v2_forward -- test main of the forward pass.
//   no arguments : the reference's two cases with its verdict strings
//                  (src/02_flash_attention_v2_forward/main.cu: "Simple test PASSED|FAILED" :247,
//                  "Test PASSED|FAILED" :89) through the fp32 reference-signature drop-in
//                  flash_attention_2_forward, then the same random case on the bf16 MFMA path;
//   B H N d [causal [iters [fp8]]] : bf16 (or, with the literal "fp8", e4m3) forward at that shape: parity on sampled query rows against
//                  the CPU oracle, then timing, TFLOP/s and % of the MFMA peak.
#include <iostream>

#include "../common/harness.h"

using namespace harness;

static bool test_simple_attention()
{
    std::cout << "\n=== Simple Test Case ===" << std::endl;
    const int N = 4, d = 4;
    const float scale = 1.0f;
    const std::vector<float> Q = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 1, 1, 0, 0};
    const std::vector<float> K = Q;
    std::vector<float> V(16);
    for (int i = 0; i < 16; ++i) V[i] = (float)(i + 1);
    std::vector<float> O_naive(16), O_flash(16), L(4);
    oracle_naive_forward_pass(Q.data(), K.data(), V.data(), O_naive.data(), nullptr, N, d, scale);

    DevBuf<float> dQ(16), dK(16), dV(16), dO(16), dL(4);
    dQ.up(Q.data()); dK.up(K.data()); dV.up(V.data());
    CHECK_FA2(flash_attention_2_forward(dQ.p, dK.p, dV.p, dO.p, dL.p, N, d, scale));
    CHECK_HIP(hipDeviceSynchronize());
    dO.down(O_flash.data()); dL.down(L.data());

    print_matrix("Naive Attention Output O", O_naive.data(), N, d);
    print_matrix("FlashAttention-2 Output O", O_flash.data(), N, d);
    std::cout << "\nLog-sum-exp values L:" << std::endl;
    for (int i = 0; i < N; ++i) std::cout << "L[" << i << "] = " << L[i] << std::endl;
    float max_diff = 0.0f;
    for (int i = 0; i < 16; ++i) max_diff = std::max(max_diff, std::fabs(O_naive[i] - O_flash[i]));
    std::cout << "\nMax difference: " << max_diff << std::endl;
    std::cout << "Simple test " << (max_diff < 1e-4 ? "PASSED" : "FAILED") << std::endl;
    return max_diff < 1e-4;
}

static bool test_flash_attention_2()
{
    const int N = 512, d = 64;
    const float scale = 1.0f / sqrtf((float)d);
    std::vector<float> Q, K, V;
    fwd_rand(N, d, Q, K, V);
    const size_t n = (size_t)N * d;
    std::vector<float> O_naive(n), O_flash(n), L(N);
    oracle_naive_forward_pass(Q.data(), K.data(), V.data(), O_naive.data(), nullptr, N, d, scale);

    DevBuf<float> dQ(n), dK(n), dV(n), dO(n), dL(N);
    dQ.up(Q.data()); dK.up(K.data()); dV.up(V.data());
    CHECK_FA2(flash_attention_2_forward(dQ.p, dK.p, dV.p, dO.p, dL.p, N, d, scale));
    CHECK_HIP(hipDeviceSynchronize());
    dO.down(O_flash.data());

    float max_diff = 0.0f;
    double avg = 0.0;
    int large = 0;
    for (size_t i = 0; i < n; ++i) {
        const float diff = std::fabs(O_naive[i] - O_flash[i]);
        max_diff = std::max(max_diff, diff);
        avg += diff;
        if (diff > 1e-3 || std::isnan(diff)) ++large;
    }
    std::cout << "\nTest Results:" << std::endl;
    std::cout << "Max difference: " << max_diff << std::endl;
    std::cout << "Avg difference: " << avg / n << std::endl;
    std::cout << "Number of large differences (>1e-3): " << large << " out of " << n << std::endl;
    std::cout << "Test " << (max_diff < 5e-3 ? "PASSED" : "FAILED") << std::endl;

    // the same case on the bf16 MFMA path (inputs rounded to bf16, oracle fed the rounded values)
    std::vector<uint16_t> q16, k16, v16, o16(n);
    std::vector<float> qr, kr, vr, Or(n), Ob(n);
    to_bf16(Q, q16, &qr); to_bf16(K, k16, &kr); to_bf16(V, v16, &vr);
    oracle_attention_forward_rows_f64(qr.data(), kr.data(), vr.data(), Or.data(), nullptr, 1, N, d, scale, 0, 0, 1, 0, 1);
    DevBuf<uint16_t> bQ(n), bK(n), bV(n), bO(n);
    bQ.up(q16.data()); bK.up(k16.data()); bV.up(v16.data());
    CHECK_FA2(fa2_forward(bQ.p, bK.p, bV.p, bO.p, dL.p, 1, 1, N, d, scale, FA2_DTYPE_BF16, 0, nullptr));
    CHECK_HIP(hipDeviceSynchronize());
    bO.down(o16.data());
    float mb = 0.0f;
    for (size_t i = 0; i < n; ++i) { Ob[i] = bf2f(o16[i]); mb = std::max(mb, std::fabs(Ob[i] - O_naive[i])); }
    const double r = rel_l2(Ob.data(), Or.data(), n);
    printf("bf16 path: max diff vs fp32 naive %.3g (gate 5e-3), rel-L2 vs oracle on rounded inputs %.3g (gate 5e-3)\n", mb, r);
    std::cout << "bf16 Test " << ((mb < 5e-3 && r < 5e-3) ? "PASSED" : "FAILED") << std::endl;
    return max_diff < 5e-3 && mb < 5e-3 && r < 5e-3;
}

// fp8 = true: BASELINE configs[4] -- e4m3 Q/K/V (d = 128), bf16 O; the oracle is fed the e4m3-rounded inputs
// and the gate is SURVEY 8c's 5e-2 (P is carried in e4m3 for the second product).
static int run_shape(const Shape& s, bool fp8)
{
    print_device();
    const size_t E = (size_t)s.B * s.H * s.N * s.d;
    const float scale = 1.0f / sqrtf((float)s.d);
    const int dtype = fp8 ? FA2_DTYPE_FP8_E4M3 : FA2_DTYPE_BF16;
    const double gate = fp8 ? 5e-2 : 5e-3;
    std::vector<float> Q, K, V, qr, kr, vr;
    std::vector<uint16_t> q16, k16, v16, o16(E);
    std::vector<uint8_t> q8, k8, v8;
    fill_uniform(Q, E, 1, 1.0f); fill_uniform(K, E, 2, 1.0f); fill_uniform(V, E, 3, 1.0f);
    DevBuf<uint16_t> dQ(E), dK(E), dV(E), dO(E);      // the fp8 inputs use the first half of these buffers
    DevBuf<float> dL((size_t)s.B * s.H * s.N);
    if (fp8) {
        to_e4m3(Q, q8, &qr); to_e4m3(K, k8, &kr); to_e4m3(V, v8, &vr);
        CHECK_HIP(hipMemcpy(dQ.p, q8.data(), E, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(dK.p, k8.data(), E, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(dV.p, v8.data(), E, hipMemcpyHostToDevice));
    } else {
        to_bf16(Q, q16, &qr); to_bf16(K, k16, &kr); to_bf16(V, v16, &vr);
        dQ.up(q16.data()); dK.up(k16.data()); dV.up(v16.data());
    }
    CHECK_FA2(fa2_forward(dQ.p, dK.p, dV.p, dO.p, dL.p, s.B, s.H, s.N, s.d, scale, dtype, s.causal, nullptr));
    CHECK_HIP(hipDeviceSynchronize());
    dO.down(o16.data());

    // parity on sampled rows: first and last head, >= 256 strided rows each
    const int BH = s.B * s.H;
    const int stride = std::max(1, s.N / 256);
    std::vector<float> Or(E, 0.0f), Lr((size_t)BH * s.N);
    bool ok = true;
    for (int bh : {0, BH - 1}) {
        oracle_attention_forward_rows_f64(qr.data(), kr.data(), vr.data(), Or.data(), Lr.data(), BH, s.N, s.d,
                                          scale, s.causal, bh, bh + 1, bh % stride, stride);
        double num = 0, den = 0;
        for (int i = bh % stride; i < s.N; i += stride)
            for (int c = 0; c < s.d; ++c) {
                const size_t at = ((size_t)bh * s.N + i) * s.d + c;
                const double e = (double)bf2f(o16[at]) - Or[at];
                num += e * e; den += (double)Or[at] * Or[at];
            }
        const double r = std::sqrt(num / std::max(den, 1e-300));
        printf("head %d: rel-L2(O) on %d sampled rows = %.3e (gate %.0e)\n", bh, (s.N + stride - 1) / stride, r, gate);
        ok = ok && r < gate;
        if (BH == 1) break;
    }
    std::cout << "Test " << (ok ? "PASSED" : "FAILED") << std::endl;

    GpuTimer t;
    for (int i = 0; i < 3; ++i)
        CHECK_FA2(fa2_forward(dQ.p, dK.p, dV.p, dO.p, dL.p, s.B, s.H, s.N, s.d, scale, dtype, s.causal, nullptr));
    t.start();
    for (int i = 0; i < s.iters; ++i)
        CHECK_FA2(fa2_forward(dQ.p, dK.p, dV.p, dO.p, dL.p, s.B, s.H, s.N, s.d, scale, dtype, s.causal, nullptr));
    const float ms = t.stop() / s.iters;
    const double flops = 4.0 * BH * (double)s.N * s.N * s.d * (s.causal ? 0.5 : 1.0);
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("FA2 forward %s (B=%d,H=%d,N=%d,d=%d,causal=%d): %.3f ms, %.1f TFLOP/s, %.1f%% of MFMA peak\n", fp8 ? "fp8-e4m3" : "bf16",
           s.B, s.H, s.N, s.d, s.causal, ms, tf, 100.0 * tf / (fp8 ? kPeakFp8Tflops : kPeakBf16Tflops));
    return ok ? 0 : 1;
}

int main(int argc, char** argv)
{
    const Shape s = parse_shape(argc, argv);
    if (s.given) return run_shape(s, argc >= 8 && !strcmp(argv[7], "fp8"));
    const bool a = test_simple_attention();
    const bool b = test_flash_attention_2();
    return (a && b) ? 0 : 1;       // unlike the reference (always 0), the exit status reflects the verdict
}
