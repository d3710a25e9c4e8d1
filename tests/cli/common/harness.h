// tests/cli/common/harness.h -- shared pieces of the C++ test mains (the counterpart of the
// reference's harness helpers): input recipes, pass criteria, bf16
// conversion, HIP error handling, timing.  Test infrastructure: these mains link the CPU
// oracle (oracle/naive_attention.c) as their checker, exactly as the reference's mains
// #include util/naive_attention.h.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/fa2_mi355x.h"

// ---- CPU oracle (oracle/naive_attention.c), C linkage
extern "C" {
void oracle_naive_attention(const float* Q, const float* K, const float* V, float* O, int N, int d);
void oracle_naive_forward_pass(const float* Q, const float* K, const float* V, float* O, float* L,
                               int N, int d, float scale);
void oracle_naive_attention_backward(const float* Q, const float* K, const float* V, const float* O,
                                     const float* L, const float* dO, float* dQ, float* dK, float* dV,
                                     int N, int d, float scale);
void oracle_attention_forward_rows_f64(const float* Q, const float* K, const float* V, float* O, float* L,
                                       int BH, int N, int d, float scale, int causal,
                                       int bh0, int bh1, int row0, int stride);
void oracle_attention_backward_f64(const float* Q, const float* K, const float* V, const float* dO,
                                   float* dQ, float* dK, float* dV, int BH, int N, int d, float scale, int causal);
void oracle_fwdbwd_rows_f32(const float* Q, const float* K, const float* V, const float* dO,
                            float* O_rows, float* dQ_rows, float* dK, float* dV,
                            int N, int d, float scale, int row0, int stride);
}

#define CHECK_HIP(call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "HIP error at %s:%d - %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                        \
        }                                                                                   \
    } while (0)

#define CHECK_FA2(call)                                                                     \
    do {                                                                                    \
        int s_ = (call);                                                                    \
        if (s_ != FA2_OK) {                                                                 \
            fprintf(stderr, "fa2 error at %s:%d - %d (%s)\n", __FILE__, __LINE__, s_, fa2_status_string(s_)); \
            exit(1);                                                                        \
        }                                                                                   \
    } while (0)

namespace harness {

// ---- bf16 <-> f32 on the host (round to nearest even)
inline uint16_t f2bf(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf2f(uint16_t h)
{
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline void to_bf16(const std::vector<float>& src, std::vector<uint16_t>& dst, std::vector<float>* rounded = nullptr)
{
    dst.resize(src.size());
    if (rounded) rounded->resize(src.size());
    for (size_t i = 0; i < src.size(); ++i) {
        dst[i] = f2bf(src[i]);
        if (rounded) (*rounded)[i] = bf2f(dst[i]);
    }
}

// ---- OCP e4m3 (fp8) <-> f32 on the host: round to nearest even, saturating at +-448 (0x7f / 0xff are NaN)
inline uint8_t f2e4m3(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 31) << 7);
    const float a = std::fabs(f);
    if (a != a) return sign | 0x7f;
    if (a >= 464.0f) return sign | 0x7e;                       // beyond 448 + half an ulp
    int e;
    const float m = std::frexp(a, &e);                         // a = m 2^e, m in [0.5, 1)
    int E = e - 1;                                             // a = (2m) 2^E
    if (a == 0.0f) return sign;
    if (E < -6) {                                              // subnormal: multiples of 2^-9
        const int r = (int)std::nearbyint(a * 512.0f);
        return sign | (uint8_t)r;                              // r == 8 is the smallest normal, 0x08
    }
    int r = (int)std::nearbyint((2.0f * m - 1.0f) * 8.0f);
    if (r == 8) { r = 0; ++E; }
    if (E > 8 || (E == 8 && r == 7)) return sign | 0x7e;
    return sign | (uint8_t)(((E + 7) << 3) | r);
}
inline float e4m32f(uint8_t b)
{
    const int E = (b >> 3) & 15, r = b & 7;
    float v;
    if (E == 15 && r == 7) v = NAN;
    else if (E == 0) v = std::ldexp((float)r, -9);
    else v = std::ldexp(1.0f + r / 8.0f, E - 7);
    return (b & 0x80) ? -v : v;
}
inline void to_e4m3(const std::vector<float>& src, std::vector<uint8_t>& dst, std::vector<float>* rounded = nullptr)
{
    dst.resize(src.size());
    if (rounded) rounded->resize(src.size());
    for (size_t i = 0; i < src.size(); ++i) {
        dst[i] = f2e4m3(src[i]);
        if (rounded) (*rounded)[i] = e4m32f(dst[i]);
    }
}

// ---- device buffers
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    explicit DevBuf(size_t count = 0) { if (count) alloc(count); }
    void alloc(size_t count) { n = count; CHECK_HIP(hipMalloc(&p, count * sizeof(T))); }
    void up(const T* h) { CHECK_HIP(hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice)); }
    void down(T* h) const { CHECK_HIP(hipMemcpy(h, p, n * sizeof(T), hipMemcpyDeviceToHost)); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// ---- recipes
// 02_flash_attention_v2_forward/main.cu:28-33
inline void fwd_rand(int N, int d, std::vector<float>& Q, std::vector<float>& K, std::vector<float>& V)
{
    const size_t n = (size_t)N * d;
    Q.resize(n); K.resize(n); V.resize(n);
    srand(42);
    for (size_t i = 0; i < n; ++i) {
        Q[i] = (rand() % 1000) / 1000.0f - 0.5f;
        K[i] = (rand() % 1000) / 1000.0f - 0.5f;
        V[i] = (rand() % 1000) / 1000.0f - 0.5f;
    }
}
// 02_flash_attention_v2_backward/main.cu:221-227
inline void bwd_rand(int N, int d, std::vector<float>& Q, std::vector<float>& K, std::vector<float>& V,
                     std::vector<float>& dO)
{
    const size_t n = (size_t)N * d;
    Q.resize(n); K.resize(n); V.resize(n); dO.resize(n);
    srand(42);
    for (size_t i = 0; i < n; ++i) {
        Q[i] = ((rand() % 2000) / 1000.0f - 1.0f) * 0.5f;
        K[i] = ((rand() % 2000) / 1000.0f - 1.0f) * 0.5f;
        V[i] = ((rand() % 2000) / 1000.0f - 1.0f) * 0.5f;
        dO[i] = ((rand() % 2000) / 1000.0f - 1.0f) * 0.2f;
    }
}
// counter-based generator for the (B,H,N,d) runs: uniform [-0.5, 0.5) * amp
inline float unit_rand(uint64_t idx, uint64_t seed)
{
    uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)((z >> 40) * (1.0 / 16777216.0)) - 0.5f;
}
inline void fill_uniform(std::vector<float>& x, size_t n, uint64_t seed, float amp)
{
    x.resize(n);
    for (size_t i = 0; i < n; ++i) x[i] = unit_rand(i, seed) * amp;
}
// util/attention_helper.h:151-173
inline void create_simple_test_data(std::vector<float>& Q, std::vector<float>& K, std::vector<float>& V,
                                    int seq_len, int head_dim)
{
    Q.assign((size_t)seq_len * head_dim, 0.0f);
    for (int i = 0; i < seq_len && i < head_dim; ++i) Q[(size_t)i * head_dim + i] = 1.0f;
    K = Q;
    V.resize((size_t)seq_len * head_dim);
    for (int i = 0; i < seq_len; ++i)
        for (int j = 0; j < head_dim; ++j) V[(size_t)i * head_dim + j] = i * 4.0f + j + 1.0f;
}

// ---- criteria
// util/attention_helper.h:174-208: wrong only if BOTH rel > rtol AND abs > atol
inline bool compare_outputs(const float* ref, const float* test, size_t size, float rtol = 1e-3f, float atol = 1.0f)
{
    size_t bad = 0;
    float worst = 0.0f;
    size_t worst_i = 0;
    for (size_t i = 0; i < size; ++i) {
        const float diff = std::fabs(ref[i] - test[i]);
        const float rel = diff / (std::fabs(ref[i]) + 1e-8f);
        if (rel > rtol && diff > atol) {
            if (bad < 10) printf("Diff at index %zu: ref=%.4f, test=%.4f, diff=%.4f (rel=%.6f)\n", i, ref[i], test[i], diff, rel);
            ++bad;
            if (diff > worst) { worst = diff; worst_i = i; }
        }
    }
    if (bad) {
        printf("Total significant differences: %zu out of %zu\n", bad, size);
        printf("Max difference: %.4f (rel=%.6f) at index %zu\n", worst, worst / (std::fabs(ref[worst_i]) + 1e-8f), worst_i);
    } else {
        printf("All outputs match within tolerance (rtol=%.1e, atol=%.1f)\n", rtol, atol);
    }
    return bad == 0;
}
// 02_flash_attention_v2_backward/main.cu:20-45
inline void compare_gradients(const char* name, const float* got, const float* want, size_t size,
                              float& max_diff, float& avg_diff)
{
    max_diff = 0.0f;
    double sum = 0.0;
    size_t large = 0;
    for (size_t i = 0; i < size; ++i) {
        const float diff = std::fabs(got[i] - want[i]);
        max_diff = std::max(max_diff, diff);
        sum += diff;
        if (diff > 1e-3f || std::isnan(diff)) {
            if (++large <= 5) printf("Large diff in %s at index %zu: flash=%g, naive=%g, diff=%g\n", name, i, got[i], want[i], diff);
        }
    }
    avg_diff = (float)(sum / (double)size);
    printf("%s - Max diff: %g, Avg diff: %g, Large diffs: %zu/%zu\n", name, max_diff, avg_diff, large, size);
}
inline double rel_l2(const float* got, const float* want, size_t n)
{
    double num = 0.0, den = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double e = (double)got[i] - want[i];
        num += e * e;
        den += (double)want[i] * want[i];
    }
    return std::sqrt(num / std::max(den, 1e-300));
}
inline void print_matrix(const char* name, const float* m, int rows, int cols, int max_rows = 4, int max_cols = 8)
{
    printf("\n%s (showing %dx%d):\n", name, std::min(rows, max_rows), std::min(cols, max_cols));
    for (int i = 0; i < std::min(rows, max_rows); ++i) {
        for (int j = 0; j < std::min(cols, max_cols); ++j) printf("%8.4f ", m[(size_t)i * cols + j]);
        printf("\n");
    }
}

// ---- timing
struct GpuTimer {
    hipEvent_t a, b;
    GpuTimer() { CHECK_HIP(hipEventCreate(&a)); CHECK_HIP(hipEventCreate(&b)); }
    ~GpuTimer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    void start(hipStream_t s = nullptr) { CHECK_HIP(hipEventRecord(a, s)); }
    float stop(hipStream_t s = nullptr)
    {
        CHECK_HIP(hipEventRecord(b, s));
        CHECK_HIP(hipEventSynchronize(b));
        float ms = 0.0f;
        CHECK_HIP(hipEventElapsedTime(&ms, a, b));
        return ms;
    }
};

constexpr double kPeakFp8Tflops = 5033.2;    // MI355X dense fp8 MFMA (twice the bf16 rate)
constexpr double kPeakBf16Tflops = 2516.6;   // MI355X dense bf16 MFMA (256 CU x 4096 flop/clk x 2.4 GHz)

inline void print_device()
{
    hipDeviceProp_t p;
    CHECK_HIP(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs, %.0f MHz, arch %s; bf16 dense MFMA peak used: %.1f TFLOP/s\n", p.name,
           p.multiProcessorCount, p.clockRate / 1000.0, p.gcnArchName, kPeakBf16Tflops);
}

struct Shape {
    int B = 0, H = 0, N = 0, d = 0, causal = 0, iters = 10;
    bool given = false;
};
// main [B H N d [causal [iters]]]
inline Shape parse_shape(int argc, char** argv)
{
    Shape s;
    if (argc >= 5) {
        s.B = atoi(argv[1]); s.H = atoi(argv[2]); s.N = atoi(argv[3]); s.d = atoi(argv[4]);
        if (argc >= 6) s.causal = atoi(argv[5]);
        if (argc >= 7) s.iters = atoi(argv[6]);
        s.given = true;
    } else if (argc != 1) {
        fprintf(stderr, "usage: %s [B H N d [causal [iters]]]   (no arguments: the reference's own test cases)\n", argv[0]);
        exit(2);
    }
    return s;
}

}  // namespace harness
