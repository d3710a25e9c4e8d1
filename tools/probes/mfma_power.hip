// Probe: sustained clock and MFMA rate under load.  Every SIMD of the chip runs back-to-back
// v_mfma_f32_32x32x16_bf16 on random operands for tens of milliseconds; the shader clock is
// s_memtime ticks / wall time, the rate is MFMAs / wall time.  Variants add the LDS fragment reads
// and the exponentials an attention loop carries, to see what they cost in clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MFMA(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define LDS(d, a) asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a))
template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) k(const bf16x8* in, float* out, long long* cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 4096; i += THREADS) ((bf16x8*)lds)[i] = in[(i * 7 + blockIdx.x) & 4095];
    __syncthreads();
    bf16x8 a0 = in[threadIdx.x], b0 = in[threadIdx.x + 512], a1 = in[threadIdx.x + 1024], b1 = in[threadIdx.x + 1536];
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float x0 = 0.1f * threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned la = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    bf16x8 l0 = a0, l1 = a1;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { MFMA(c0, a0, b0); MFMA(c1, a1, b1); MFMA(c2, a0, b1); MFMA(c3, a1, b0); }
        if (MODE == 1) {   // + one 1 KiB LDS fragment read per MFMA, used as the A operand
            LDS(l0, la); MFMA(c0, a0, b0); LDS(l1, la + 1024); MFMA(c1, a1, b1);
            asm volatile("s_waitcnt lgkmcnt(0)");
            MFMA(c2, l0, b1); MFMA(c3, l1, b0);
            la = (la + 2048) & 0xffff;
        }
        if (MODE == 2) {   // + one exp per MFMA
            MFMA(c0, a0, b0); EXP(x0); MFMA(c1, a1, b1); EXP(x1); MFMA(c2, a0, b1); EXP(x2); MFMA(c3, a1, b0); EXP(x3);
        }
        if (MODE == 3) {   // attention-like: LDS read + exp + 3 plain VALU per MFMA
            LDS(l0, la); MFMA(c0, a0, b0); EXP(x0); asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_add_f32 %1, %1, %0\n\tv_max_f32 %0, %0, %1" : "+v"(x1), "+v"(x2));
            LDS(l1, la + 1024); MFMA(c1, a1, b1); EXP(x3); asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_add_f32 %1, %1, %0\n\tv_max_f32 %0, %0, %1" : "+v"(x1), "+v"(x2));
            asm volatile("s_waitcnt lgkmcnt(0)");
            MFMA(c2, l0, b1); EXP(x0); asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_add_f32 %1, %1, %0\n\tv_max_f32 %0, %0, %1" : "+v"(x1), "+v"(x2));
            MFMA(c3, l1, b0); EXP(x3); asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_add_f32 %1, %1, %0\n\tv_max_f32 %0, %0, %1" : "+v"(x1), "+v"(x2));
            la = (la + 2048) & 0xffff;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = x0 + x1 + x2 + x3 + (float)l0[0] + (float)l1[0];
    for (int q = 0; q < 16; ++q) r += c0[q] + c1[q] + c2[q] + c3[q];
    out[blockIdx.x * THREADS + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE, int THREADS> void run(const char* name, const bf16x8* in, float* out, long long* cyc)
{
    const int iters = 400000, blocks = 256, waves = THREADS / 64;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, in, out, cyc, 20000);   // warm
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, in, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * waves);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= h.size();
    const double mfmas = 4.0 * iters * blocks * waves;
    printf("%-46s %d waves/SIMD: %6.1f ms, clock %.0f MHz, %5.1f ticks per MFMA per SIMD, %6.0f TFLOP/s\n", name, waves / 4, ms,
           m / ms / 1e3, m / (4.0 * iters * (waves / 4)), mfmas * 32768.0 / ms / 1e9);
}
int main()
{
    bf16x8* in; float* out; long long* cyc;
    std::vector<unsigned short> h(4096 * 8);
    srand(1); for (auto& v : h) { float f = (rand() % 2000) / 1000.0f - 1.0f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    (void)hipMalloc(&in, 4096 * 16); (void)hipMemcpy(in, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    run<0, 256>("mfma only", in, out, cyc);
    run<0, 512>("mfma only", in, out, cyc);
    run<1, 256>("mfma + 1 KiB LDS read each", in, out, cyc);
    run<1, 512>("mfma + 1 KiB LDS read each", in, out, cyc);
    run<2, 256>("mfma + exp each", in, out, cyc);
    run<2, 512>("mfma + exp each", in, out, cyc);
    run<3, 256>("mfma + LDS read + exp + 3 valu each", in, out, cyc);
    run<3, 512>("mfma + LDS read + exp + 3 valu each", in, out, cyc);
    return 0;
}
