// Probe: what do the packed-fp32 and dot2 VALU forms cost the issue port on gfx950, one and two waves per SIMD,
// alone and in the shadow of MFMAs?  (The softmax of a 32 x 32 block of S is 16 elements per lane: fma, exp, add,
// half a max3, half a pack each = 64 instructions; with v_pk_fma_f32 / v_pk_add_f32 it would be 48.)
// Every line reports clocks (s_memtime ticks are 100 MHz: we use the shader cycle counter s_memtime on gfx950 = core
// clock) per loop iteration per wave, and the wall time per iteration from hipEvents.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define MFMA(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define FMA(x, y) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y))
#define ADD(x, y) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define MAX3(x, y, z) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define PKFMA(x, y) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y))
#define PKFMAB(x, c, m) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(c), "v"(m))
#define PKADD(x, y) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define PKMUL(x, y) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define DOT2(acc, p, o) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(p), "v"(o))
#define R4(X) X X X X
template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) k(const bf16x8* in, float* out, long long* cyc, int iters)
{
    bf16x8 a = in[threadIdx.x & 255], b = in[(threadIdx.x & 255) + 256];
    f32x16 c0 = {0}, c1 = {0};
    float x[16]; f32x2 p[8]; unsigned d[8];
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * threadIdx.x + i;
    for (int i = 0; i < 8; ++i) { p[i] = f32x2{x[i], x[i] + 0.5f}; d[i] = 0x3f803f80u; }
    float y = 0.999f, l0 = 0.f, l1 = 0.f, m0 = 0.f; f32x2 py = {0.999f, 0.998f}, pl = {0.f, 0.f}; f32x2 cm = {0.5f, 0.25f};
    unsigned one = 0x3f803f80u;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { R4(FMA(x[0], y); FMA(x[1], y); FMA(x[2], y); FMA(x[3], y);) }                    // 16 fma
        if (MODE == 1) { R4(PKFMA(p[0], py); PKFMA(p[1], py);) /* dependent at distance 2 */  }                                              // 8 pk_fma = 16 elements
        if (MODE == 2) { PKADD(p[0], py); PKADD(p[1], py); PKADD(p[2], py); PKADD(p[3], py); PKADD(p[4], py); PKADD(p[5], py); PKADD(p[6], py); PKADD(p[7], py); }
        if (MODE == 3) { PKMUL(p[0], py); PKMUL(p[1], py); PKMUL(p[2], py); PKMUL(p[3], py); PKMUL(p[4], py); PKMUL(p[5], py); PKMUL(p[6], py); PKMUL(p[7], py); }
        if (MODE == 4) { PKFMAB(p[0], py, cm); PKFMAB(p[1], py, cm); PKFMAB(p[2], py, cm); PKFMAB(p[3], py, cm); PKFMAB(p[4], py, cm); PKFMAB(p[5], py, cm); PKFMAB(p[6], py, cm); PKFMAB(p[7], py, cm); }                                      // broadcast scalars by op_sel
        if (MODE == 5) { R4(DOT2(l0, d[0], one); DOT2(l1, d[1], one);) }                                      // 8 dot2c = 16 elements summed
        if (MODE == 6) { R4(ADD(l0, x[0]); ADD(l1, x[1]); ADD(l0, x[2]); ADD(l1, x[3]);) }                     // 16 add
        if (MODE == 20) { MFMA(c0, a, b); MFMA(c1, a, b); }
        if (MODE == 21) { MFMA(c0, a, b); FMA(x[0], y); FMA(x[1], y); FMA(x[2], y); FMA(x[3], y); MFMA(c1, a, b); FMA(x[4], y); FMA(x[5], y); FMA(x[6], y); FMA(x[7], y); }
        if (MODE == 22) { MFMA(c0, a, b); PKFMA(p[0], py); PKFMA(p[1], py); MFMA(c1, a, b); PKFMA(p[2], py); PKFMA(p[3], py); }
        if (MODE == 23) { MFMA(c0, a, b); PKFMA(p[0], py); PKFMA(p[1], py); PKFMA(p[2], py); PKFMA(p[3], py); MFMA(c1, a, b); PKFMA(p[4], py); PKFMA(p[5], py); PKFMA(p[6], py); PKFMA(p[7], py); }
        if (MODE == 24) { MFMA(c0, a, b); PKADD(p[0], py); PKADD(p[1], py); PKADD(p[2], py); PKADD(p[3], py); MFMA(c1, a, b); PKADD(p[4], py); PKADD(p[5], py); PKADD(p[6], py); PKADD(p[7], py); }
        if (MODE == 25) { MFMA(c0, a, b); DOT2(x[0], d[0], one); DOT2(x[1], d[1], one); DOT2(x[2], d[2], one); DOT2(x[3], d[3], one); MFMA(c1, a, b); DOT2(x[4], d[0], one); DOT2(x[5], d[1], one); DOT2(x[6], d[2], one); DOT2(x[7], d[3], one); }
        if (MODE == 26) { MFMA(c0, a, b); ADD(x[0], y); ADD(x[1], y); ADD(x[2], y); ADD(x[3], y); MFMA(c1, a, b); ADD(x[4], y); ADD(x[5], y); ADD(x[6], y); ADD(x[7], y); }
        if (MODE == 27) { MFMA(c0, a, b); EXP(x[0]); EXP(x[1]); EXP(x[2]); EXP(x[3]); MFMA(c1, a, b); EXP(x[4]); EXP(x[5]); EXP(x[6]); EXP(x[7]); }
        if (MODE == 30) {       // a 32 x 32 block of S at d = 64: 8 MFMAs + the 64 VALU instructions of its softmax, one MFMA per pair
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {
                if (j & 1) { MFMA(c1, a, b); } else { MFMA(c0, a, b); }
                FMA(x[2 * j], y); FMA(x[2 * j + 1], y); EXP(x[2 * j]); EXP(x[2 * j + 1]); ADD(l0, x[2 * j]); ADD(l1, x[2 * j + 1]);
                CVT(d[j], x[2 * j], x[2 * j + 1]); MAX3(m0, x[2 * j], x[2 * j + 1]);
            }
        }
        // softmax of 16 elements per lane, plain: 16 fma, 16 exp, 16 add, 8 cvt, 8 max3 (=64)
        if (MODE == 10 || MODE == 12) {
            if (MODE == 12) { MFMA(c0, a, b); }
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {
                FMA(x[2 * j], y); FMA(x[2 * j + 1], y); EXP(x[2 * j]); EXP(x[2 * j + 1]); ADD(l0, x[2 * j]); ADD(l1, x[2 * j + 1]);
                CVT(d[j], x[2 * j], x[2 * j + 1]); MAX3(m0, x[2 * j], x[2 * j + 1]);
                if (MODE == 12 && (j == 1 || j == 3 || j == 5)) { MFMA(c1, a, b); }
            }
        }
        // packed: 8 pk_fma, 16 exp, 8 pk_add, 8 cvt, 8 max3 (=48)
        if (MODE == 11 || MODE == 13) {
            if (MODE == 13) { MFMA(c0, a, b); }
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {
                PKFMA(p[j], py);
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(p[j][0]), "+v"(p[j][1]));
                PKADD(pl, p[j]); CVT(d[j], p[j][0], p[j][1]); MAX3(m0, p[j][0], p[j][1]);
                if (MODE == 13 && (j == 1 || j == 3 || j == 5)) { MFMA(c1, a, b); }
            }
        }
        // dot2 row sums: 16 fma, 16 exp, 8 cvt, 8 dot2c, 8 max3 (=56)
        if (MODE == 14) {
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {
                FMA(x[2 * j], y); FMA(x[2 * j + 1], y); EXP(x[2 * j]); EXP(x[2 * j + 1]);
                CVT(d[j], x[2 * j], x[2 * j + 1]); DOT2(l0, d[j], one); MAX3(m0, x[2 * j], x[2 * j + 1]);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = l0 + l1 + m0 + pl[0] + pl[1] + cm[0];
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + x[r];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1] + d[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE> void run4(const char* name, const bf16x8* in, float* out, long long* cyc)
{
    const int iters = 40000, blocks = 256;
    static long long h[256 * 16];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-58s", name);
    for (int w = 1; w <= 4; ++w) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (w == 1) hipLaunchKernelGGL((k<MODE, 256>), dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
            if (w == 2) hipLaunchKernelGGL((k<MODE, 512>), dim3(blocks), dim3(512), 0, 0, in, out, cyc, iters);
            if (w == 3) hipLaunchKernelGGL((k<MODE, 768>), dim3(blocks), dim3(768), 0, 0, in, out, cyc, iters);
            if (w == 4) hipLaunchKernelGGL((k<MODE, 1024>), dim3(blocks), dim3(1024), 0, 0, in, out, cyc, iters);
            (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        (void)hipMemcpy(h, cyc, blocks * w * 4 * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < blocks * w * 4; ++i) m += h[i];
        printf(" | %d/SIMD: %6.1f ticks/iter, %6.1f ns per block-iteration", w, m / (blocks * w * 4) / iters, ms * 1e6 / iters / w);
    }
    printf("\n");
}
template <int MODE> void run(const char* name, const bf16x8* in, float* out, long long* cyc)
{
    const int iters = 40000, blocks = 256;
    static long long h[256 * 8];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double clk[2], ns[2];
    for (int w = 0; w < 2; ++w) {
        const int threads = w ? 512 : 256, waves = threads / 64;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (w) hipLaunchKernelGGL((k<MODE, 512>), dim3(blocks), dim3(512), 0, 0, in, out, cyc, iters);
            else hipLaunchKernelGGL((k<MODE, 256>), dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
            (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        (void)hipMemcpy(h, cyc, blocks * waves * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < blocks * waves; ++i) m += h[i];
        clk[w] = m / (blocks * waves) / iters; ns[w] = ms * 1e6 / iters;
    }
    printf("%-58s 1 wave/SIMD %7.1f ticks %7.1f ns | 2 waves/SIMD %7.1f ticks %7.1f ns (per iteration; ns are per SIMD-iteration of ALL its waves)\n",
           name, clk[0], ns[0], clk[1], ns[1]);
}
int main()
{
    bf16x8* in; float* out; long long* cyc;
    (void)hipMalloc(&in, 512 * 16); (void)hipMemset(in, 0x3c, 512 * 16); (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 16 * 8);
    run<0>("16 v_fma_f32 (4 independent chains)", in, out, cyc);
    run<1>("8 v_pk_fma_f32 (2 chains)", in, out, cyc);
    run<2>("8 v_pk_add_f32 (8 chains)", in, out, cyc);
    run<3>("8 v_pk_mul_f32 (8 chains)", in, out, cyc);
    run<4>("8 v_pk_fma_f32, scalars broadcast by op_sel_hi (8 chains)", in, out, cyc);
    run<5>("8 v_dot2c_f32_bf16", in, out, cyc);
    run<6>("16 v_add_f32 (2 chains)", in, out, cyc);
    run<10>("softmax 16 elements, plain (64 VALU)", in, out, cyc);
    run<11>("softmax 16 elements, pk_fma + pk_add (48 VALU)", in, out, cyc);
    run<14>("softmax 16 elements, dot2c row sums (56 VALU)", in, out, cyc);
    run<12>("4 MFMA + plain softmax", in, out, cyc);
    run<13>("4 MFMA + packed softmax", in, out, cyc);
    run<20>("2 MFMA", in, out, cyc);
    run<21>("2 x (MFMA + 4 fma)", in, out, cyc);
    run<26>("2 x (MFMA + 4 add)", in, out, cyc);
    run<27>("2 x (MFMA + 4 exp)", in, out, cyc);
    run<22>("2 x (MFMA + 2 pk_fma)", in, out, cyc);
    run<23>("2 x (MFMA + 4 pk_fma)", in, out, cyc);
    run<24>("2 x (MFMA + 4 pk_add)", in, out, cyc);
    run<25>("2 x (MFMA + 4 dot2c)", in, out, cyc);
    run4<0>("16 v_fma_f32", in, out, cyc);
    run4<10>("softmax of a 32 x 32 block (64 VALU)", in, out, cyc);
    run4<20>("2 MFMA", in, out, cyc);
    run4<12>("4 MFMA + softmax block (fp8 / d = 128-like ratio)", in, out, cyc);
    run4<30>("8 MFMA + softmax block (d = 64 ratio)", in, out, cyc);
    return 0;
}
