// Probe: issue cost (cycles per wave instruction, one wave per SIMD) of the VALU ops the softmax
// uses, alone and in the shadow of v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define MFMA(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define FMA(x, y) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y))
#define ADD(x, y) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define MAX3(x, y) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y))
#define PKFMA(x, y) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y))
#define PKADD(x, y) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define PKMUL(x, y) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void __launch_bounds__(256, 1) k(const bf16x8* in, float* out, long long* cyc, int iters)
{
    bf16x8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 c0 = {0}, c1 = {0};
    float x[8]; f32x2 p[8]; unsigned d[8];
    for (int i = 0; i < 8; ++i) { x[i] = 0.001f * threadIdx.x + i; p[i] = f32x2{x[i], x[i] + 0.5f}; d[i] = 0; }
    float y = 0.999f; f32x2 py = {0.999f, 0.998f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { EXP(x[0]); EXP(x[1]); EXP(x[2]); EXP(x[3]); EXP(x[4]); EXP(x[5]); EXP(x[6]); EXP(x[7]); }
        if (MODE == 1) { FMA(x[0], y); FMA(x[1], y); FMA(x[2], y); FMA(x[3], y); FMA(x[4], y); FMA(x[5], y); FMA(x[6], y); FMA(x[7], y); }
        if (MODE == 2) { PKFMA(p[0], py); PKFMA(p[1], py); PKFMA(p[2], py); PKFMA(p[3], py); PKFMA(p[4], py); PKFMA(p[5], py); PKFMA(p[6], py); PKFMA(p[7], py); }
        if (MODE == 3) { PKADD(p[0], py); PKADD(p[1], py); PKADD(p[2], py); PKADD(p[3], py); PKADD(p[4], py); PKADD(p[5], py); PKADD(p[6], py); PKADD(p[7], py); }
        if (MODE == 4) { MAX3(x[0], y); MAX3(x[1], y); MAX3(x[2], y); MAX3(x[3], y); MAX3(x[4], y); MAX3(x[5], y); MAX3(x[6], y); MAX3(x[7], y); }
        if (MODE == 5) { CVT(d[0], x[0], x[1]); CVT(d[1], x[1], x[2]); CVT(d[2], x[2], x[3]); CVT(d[3], x[3], x[4]); CVT(d[4], x[4], x[5]); CVT(d[5], x[5], x[6]); CVT(d[6], x[6], x[7]); CVT(d[7], x[7], x[0]); }
        // exp interleaved with fma: does the transcendental overlap plain VALU?
        if (MODE == 6) { EXP(x[0]); FMA(x[1], y); EXP(x[2]); FMA(x[3], y); EXP(x[4]); FMA(x[5], y); EXP(x[6]); FMA(x[7], y); }
        if (MODE == 7) { EXP(x[0]); FMA(x[1], y); FMA(x[3], y); FMA(x[5], y); EXP(x[2]); FMA(x[7], y); FMA(x[4], y); FMA(x[6], y); }
        // MFMA + n VALU in its shadow (8 "slots" of 4 cycles per MFMA if plain VALU co-issues)
        if (MODE == 10) { MFMA(c0, a, b); MFMA(c1, a, b); }
        if (MODE == 11) { MFMA(c0, a, b); FMA(x[0], y); FMA(x[1], y); FMA(x[2], y); FMA(x[3], y); MFMA(c1, a, b); FMA(x[4], y); FMA(x[5], y); FMA(x[6], y); FMA(x[7], y); }
        if (MODE == 12) { MFMA(c0, a, b); REP8(FMA(x[0], y);) MFMA(c1, a, b); REP8(FMA(x[4], y);) }
        if (MODE == 13) { MFMA(c0, a, b); FMA(x[0], y); FMA(x[1], y); FMA(x[2], y); FMA(x[3], y); FMA(x[4], y); FMA(x[5], y); MFMA(c1, a, b); FMA(x[0], y); FMA(x[1], y); FMA(x[2], y); FMA(x[3], y); FMA(x[4], y); FMA(x[5], y); }
        if (MODE == 14) { MFMA(c0, a, b); EXP(x[0]); MFMA(c1, a, b); EXP(x[1]); }
        if (MODE == 15) { MFMA(c0, a, b); EXP(x[0]); EXP(x[2]); MFMA(c1, a, b); EXP(x[1]); EXP(x[3]); }
        if (MODE == 16) { MFMA(c0, a, b); EXP(x[0]); FMA(x[4], y); FMA(x[5], y); ADD(x[6], y); MFMA(c1, a, b); EXP(x[1]); FMA(x[4], y); FMA(x[5], y); ADD(x[7], y); }
        if (MODE == 17) { MFMA(c0, a, b); EXP(x[0]); EXP(x[2]); FMA(x[4], y); FMA(x[5], y); ADD(x[6], y); ADD(x[6], y); MFMA(c1, a, b); EXP(x[1]); EXP(x[3]); FMA(x[4], y); FMA(x[5], y); ADD(x[7], y); ADD(x[7], y); }
        if (MODE == 18) { MFMA(c0, a, b); EXP(x[0]); EXP(x[2]); PKFMA(p[4], py); PKADD(p[6], py); MFMA(c1, a, b); EXP(x[1]); EXP(x[3]); PKFMA(p[5], py); PKADD(p[7], py); }
        if (MODE == 19) { MFMA(c0, a, b); PKFMA(p[0], py); PKFMA(p[1], py); PKFMA(p[2], py); PKFMA(p[3], py); MFMA(c1, a, b); PKFMA(p[4], py); PKFMA(p[5], py); PKFMA(p[6], py); PKFMA(p[7], py); }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    for (int i = 0; i < 8; ++i) s += x[i] + p[i][0] + p[i][1] + d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, const bf16x8* in, float* out, long long* cyc)
{
    const int iters = 20000, blocks = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += h[i];
    printf("%-64s %.1f cycles per iteration\n", name, m / blocks / iters);
}
int main()
{
    bf16x8* in; float* out; long long* cyc;
    hipMalloc(&in, 512 * 16); hipMemset(in, 0x3c, 512 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("8 v_exp_f32", in, out, cyc);
    run<1>("8 v_fma_f32", in, out, cyc);
    run<2>("8 v_pk_fma_f32", in, out, cyc);
    run<3>("8 v_pk_add_f32", in, out, cyc);
    run<4>("8 v_max3_f32", in, out, cyc);
    run<5>("8 v_cvt_pk_bf16_f32", in, out, cyc);
    run<6>("4 exp + 4 fma alternating", in, out, cyc);
    run<7>("2 exp + 6 fma", in, out, cyc);
    run<10>("2 mfma", in, out, cyc);
    run<11>("2 x (mfma + 4 fma)", in, out, cyc);
    run<13>("2 x (mfma + 6 fma)", in, out, cyc);
    run<12>("2 x (mfma + 8 fma)", in, out, cyc);
    run<19>("2 x (mfma + 4 pk_fma)", in, out, cyc);
    run<14>("2 x (mfma + 1 exp)", in, out, cyc);
    run<15>("2 x (mfma + 2 exp)", in, out, cyc);
    run<16>("2 x (mfma + 1 exp + 2 fma + 1 add)", in, out, cyc);
    run<17>("2 x (mfma + 2 exp + 2 fma + 2 add)", in, out, cyc);
    run<18>("2 x (mfma + 2 exp + 1 pk_fma + 1 pk_add)", in, out, cyc);
    return 0;
}
