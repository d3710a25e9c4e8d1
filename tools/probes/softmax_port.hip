// Probe: what the VALU port sustains on the forward's softmax instruction stream, and what is left of it beside MFMAs.
// One "half-tile step" of a wave = 16 S elements per lane: 8 v_max3_f32 (A stage) + 8 x [2 fma, 2 exp, 2 add, 1 cvt_pk] (B stage)
// = 64 VALU instructions, in the kernel's own order and with its dependencies (fa2_fwd_bf16.hip: FA2_SOFTMAX_PAIR), plus
// M MFMAs (d = 64: 8 x 32x32x16; d = 128: 16; fp8 d = 128 per 32 keys: 4 x 32x32x64).  Variants: VALU only, MFMA only, both in
// the kernel's interleave, both with the VALU work spread evenly; 1 or 2 waves per SIMD.  Reports ticks per step per SIMD
// (two waves: per pair of steps / 2) and the shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;
#define PAIR(sa, sb, w)                                                                                                  \
    asm volatile("v_fma_f32 %[t0], %[a], %[c2], -%[mb]\n\tv_fma_f32 %[t1], %[b], %[c2], -%[mb]\n\tv_exp_f32 %[t0], %[t0]\n\t"   \
                 "v_exp_f32 %[t1], %[t1]\n\tv_add_f32 %[l], %[l], %[t0]\n\tv_cvt_pk_bf16_f32 %[ww], %[t0], %[t1]\n\t"         \
                 "v_add_f32 %[l], %[l], %[t1]"                                                                              \
                 : [l] "+v"(l), [t0] "=&v"(t0), [t1] "=&v"(t1), [ww] "=&v"(w) : [a] "v"(sa), [b] "v"(sb), [c2] "v"(c2), [mb] "v"(mb))
#define MAX3(x, a, b) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define M16(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define M64(c, a, b) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))

// MODE: 0 VALU only; 1 MFMA only; 2 kernel interleave (d = 64: MFMA, 2 max3 ... then MFMA, 2 pairs ...); 3 = d = 128 interleave
// (MFMA, 1 max3 / MFMA, 1 pair); 4 = fp8 (4 x 64-deep MFMA per 16 elements: MFMA, 2 pairs + 2 max3)
template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) k(const bf16x8* in, float* out, long long* cyc, int iters)
{
    bf16x8 a0 = in[threadIdx.x], b0 = in[threadIdx.x + 512];
    i32x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838 + threadIdx.x * 3 + i; b8[i] = 0x30303030 + threadIdx.x + i; }
    f32x16 c0 = {0}, c1 = {0};
    float s[16];
    for (int i = 0; i < 16; ++i) s[i] = 0.01f * ((threadIdx.x * 7 + i * 13) & 63);
    float l = 0.f, t0, t1, rm = -1e30f, c2 = 0.18f, mb = 0.3f;
    unsigned w[8];
    asm volatile("" :: "v"(a0), "v"(b0));
    long long t_0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            for (int i = 0; i < 8; ++i) MAX3(rm, s[2 * i], s[2 * i + 1]);
            for (int i = 0; i < 8; ++i) PAIR(s[2 * i], s[2 * i + 1], w[i]);
        }
        if constexpr (MODE == 1) { for (int i = 0; i < 4; ++i) { M16(c0, a0, b0); } for (int i = 0; i < 4; ++i) { M16(c1, a0, b0); } }
        if constexpr (MODE == 2) {
            for (int i = 0; i < 4; ++i) { M16(c0, a0, b0); MAX3(rm, s[4 * i], s[4 * i + 1]); MAX3(rm, s[4 * i + 2], s[4 * i + 3]); }
            for (int i = 0; i < 4; ++i) { M16(c1, a0, b0); PAIR(s[4 * i], s[4 * i + 1], w[2 * i]); PAIR(s[4 * i + 2], s[4 * i + 3], w[2 * i + 1]); }
        }
        if constexpr (MODE == 3) {
            for (int i = 0; i < 8; ++i) { M16(c0, a0, b0); MAX3(rm, s[2 * i], s[2 * i + 1]); }
            for (int i = 0; i < 8; ++i) { M16(c1, a0, b0); PAIR(s[2 * i], s[2 * i + 1], w[i]); }
        }
        if constexpr (MODE == 4) {
            for (int i = 0; i < 4; ++i) { M64(i & 1 ? c1 : c0, a8, b8); MAX3(rm, s[4 * i], s[4 * i + 1]); MAX3(rm, s[4 * i + 2], s[4 * i + 3]);
                                         PAIR(s[4 * i], s[4 * i + 1], w[2 * i]); PAIR(s[4 * i + 2], s[4 * i + 3], w[2 * i + 1]); }
        }
        if constexpr (MODE == 5) {     // MODE 2's work with the max3 folded between the pairs: one even stream
            for (int i = 0; i < 4; ++i) { M16(c0, a0, b0); PAIR(s[4 * i], s[4 * i + 1], w[2 * i]); MAX3(rm, s[4 * i], s[4 * i + 1]);
                                         M16(c1, a0, b0); PAIR(s[4 * i + 2], s[4 * i + 3], w[2 * i + 1]); MAX3(rm, s[4 * i + 2], s[4 * i + 3]); }
        }
        s[it & 15] += 0.001f;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    long long t_1 = __builtin_amdgcn_s_memtime();
    float r = l + rm;
    for (int q = 0; q < 16; ++q) r += c0[q] + c1[q] + s[q];
    for (int q = 0; q < 8; ++q) r += (float)w[q];
    out[blockIdx.x * THREADS + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = t_1 - t_0;
}
template <int MODE, int THREADS> void run(const char* name, const bf16x8* in, float* out, long long* cyc)
{
    const int iters = 200000, blocks = 256, waves = THREADS / 64;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, in, out, cyc, iters / 4);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, in, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * waves);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= h.size();
    printf("%-58s %d w/SIMD  %7.1f ticks per step per SIMD  (clock %4.0f MHz, %6.1f ms)\n", name, waves / 4, m / iters / (waves / 4), m / ms / 1e3, ms);
    fflush(stdout);
}
int main()
{
    bf16x8* in; float* out; long long* cyc;
    std::vector<unsigned short> h(4096 * 8);
    srand(1); for (auto& v : h) { float f = (rand() % 2000) / 1000.0f - 1.0f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    (void)hipMalloc(&in, 4096 * 16); (void)hipMemcpy(in, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    run<0, 256>("VALU only: 8 max3 + 8 softmax pairs (64 instr)", in, out, cyc);
    run<0, 512>("VALU only: 8 max3 + 8 softmax pairs (64 instr)", in, out, cyc);
    run<1, 256>("MFMA only: 8 x 32x32x16 (d = 64 step)", in, out, cyc);
    run<1, 512>("MFMA only: 8 x 32x32x16 (d = 64 step)", in, out, cyc);
    run<2, 256>("d = 64 step, kernel interleave (8 MFMA + 64 VALU)", in, out, cyc);
    run<2, 512>("d = 64 step, kernel interleave (8 MFMA + 64 VALU)", in, out, cyc);
    run<5, 256>("d = 64 step, even interleave (8 MFMA + 64 VALU)", in, out, cyc);
    run<5, 512>("d = 64 step, even interleave (8 MFMA + 64 VALU)", in, out, cyc);
    run<3, 256>("d = 128 step, kernel interleave (16 MFMA + 64 VALU)", in, out, cyc);
    run<3, 512>("d = 128 step, kernel interleave (16 MFMA + 64 VALU)", in, out, cyc);
    run<4, 256>("fp8 16 elements: 4 x 32x32x64 + 64 VALU", in, out, cyc);
    run<4, 512>("fp8 16 elements: 4 x 32x32x64 + 64 VALU", in, out, cyc);
    return 0;
}
