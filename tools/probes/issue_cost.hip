// Probe: per-instruction issue cost for a wave alone on its SIMD vs two waves per SIMD, for the
// instruction kinds the attention loops are made of (cycles per loop iteration, loop overhead ~4).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MFMA(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define FMA(x, y) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y))
#define NOP0 asm volatile("s_nop 0")
#define WAIT asm volatile("s_waitcnt lgkmcnt(0)")
#define SALU(s) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s))
#define DOT2(x, p, o) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x) : "v"(p), "v"(o))
#define ACCW(x) asm volatile("v_accvgpr_write_b32 a0, %0" :: "v"(x) : "a0")
#define ACCR(x) asm volatile("v_accvgpr_read_b32 %0, a1" : "=v"(x) :: "a1")
#define LDS128(d, a) asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a))
#define LDSTR(d, a) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(d) : "v"(a))
#define R8(X) X X X X X X X X
template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) k(const bf16x8* in, float* out, long long* cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) char lds[32768];
    for (int i = threadIdx.x; i < 8192; i += THREADS) ((float*)lds)[i] = 1.0f;
    __syncthreads();
    bf16x8 a = in[threadIdx.x & 255], b = in[(threadIdx.x & 255) + 256];
    f32x16 c0 = {0}, c1 = {0};
    float x[8]; for (int i = 0; i < 8; ++i) x[i] = 0.001f * threadIdx.x + i;
    float y = 0.999f; unsigned s = 0, pk = 0x3f803f80u, one = 0x3f803f80u;
    bf16x8 l0, l1; bf16x4 t0v, t1v;
    const unsigned la = (threadIdx.x & 63) * 16, lt = (threadIdx.x & 63) * 8;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { R8(FMA(x[0], y); ) }                                   // dependent chain
        if (MODE == 1) { FMA(x[0], y); FMA(x[1], y); FMA(x[2], y); FMA(x[3], y); FMA(x[4], y); FMA(x[5], y); FMA(x[6], y); FMA(x[7], y); }
        if (MODE == 2) { R8(NOP0; ) }
        if (MODE == 3) { R8(WAIT; ) }
        if (MODE == 4) { R8(SALU(s); ) }
        if (MODE == 5) { R8(DOT2(x[0], pk, one); ) }
        if (MODE == 6) { R8(ACCW(y); ) }
        if (MODE == 7) { R8(ACCR(x[0]); ) }
        if (MODE == 8) { LDS128(l0, la); LDS128(l1, la); LDS128(l0, la); LDS128(l1, la); LDS128(l0, la); LDS128(l1, la); LDS128(l0, la); LDS128(l1, la); WAIT; }
        if (MODE == 9) { LDSTR(t0v, lt); LDSTR(t1v, lt); LDSTR(t0v, lt); LDSTR(t1v, lt); LDSTR(t0v, lt); LDSTR(t1v, lt); LDSTR(t0v, lt); LDSTR(t1v, lt); WAIT; }
        if (MODE == 10) { FMA(x[0], y); NOP0; FMA(x[1], y); NOP0; FMA(x[2], y); NOP0; FMA(x[3], y); NOP0; }
        if (MODE == 11) { FMA(x[0], y); SALU(s); FMA(x[1], y); SALU(s); FMA(x[2], y); SALU(s); FMA(x[3], y); SALU(s); }
        if (MODE == 12) { MFMA(c0, a, b); MFMA(c1, a, b); }
        if (MODE == 13) { MFMA(c0, a, b); FMA(x[0], y); FMA(x[1], y); FMA(x[2], y); MFMA(c1, a, b); FMA(x[4], y); FMA(x[5], y); FMA(x[6], y); }
        if (MODE == 14) { MFMA(c0, a, b); FMA(x[0], y); FMA(x[1], y); FMA(x[2], y); FMA(x[3], y); FMA(x[7], y); MFMA(c1, a, b); FMA(x[4], y); FMA(x[5], y); FMA(x[6], y); FMA(x[3], y); FMA(x[7], y); }
        if (MODE == 15) { MFMA(c0, a, b); LDS128(l0, la); FMA(x[0], y); FMA(x[1], y); MFMA(c1, a, b); LDSTR(t0v, lt); FMA(x[4], y); FMA(x[5], y); }
        if (MODE == 16) { MFMA(c0, a, b); EXP(x[0]); EXP(x[1]); FMA(x[2], y); MFMA(c1, a, b); EXP(x[4]); EXP(x[5]); FMA(x[6], y); }
        if (MODE == 17) { MFMA(c0, a, b); EXP(x[0]); EXP(x[1]); EXP(x[2]); MFMA(c1, a, b); EXP(x[4]); EXP(x[5]); EXP(x[6]); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = s; for (int q = 0; q < 16; ++q) r += c0[q] + c1[q];
    for (int i = 0; i < 8; ++i) r += x[i] + (float)l0[i] + (float)l1[i];
    for (int i = 0; i < 4; ++i) r += (float)t0v[i] + (float)t1v[i];
    out[blockIdx.x * THREADS + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE> void run(const char* name, const bf16x8* in, float* out, long long* cyc)
{
    const int iters = 20000, blocks = 256;
    double res[2];
    long long h[256 * 8];
    hipLaunchKernelGGL((k<MODE, 256>), dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(h, cyc, 256 * 4 * 8, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks * 4; ++i) m += h[i]; res[0] = m / (blocks * 4) / iters;
    hipLaunchKernelGGL((k<MODE, 512>), dim3(blocks), dim3(512), 0, 0, in, out, cyc, iters);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(h, cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    m = 0; for (int i = 0; i < blocks * 8; ++i) m += h[i]; res[1] = m / (blocks * 8) / iters;
    printf("%-44s 1 wave/SIMD %6.1f   2 waves/SIMD %6.1f (per wave)\n", name, res[0], res[1]);
}
int main()
{
    bf16x8* in; float* out; long long* cyc;
    (void)hipMalloc(&in, 512 * 16); (void)hipMemset(in, 0x3c, 512 * 16); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    run<0>("8 fma dependent chain", in, out, cyc);
    run<1>("8 fma independent", in, out, cyc);
    run<2>("8 s_nop 0", in, out, cyc);
    run<3>("8 s_waitcnt lgkmcnt(0)", in, out, cyc);
    run<4>("8 s_add_u32", in, out, cyc);
    run<5>("8 v_dot2c_f32_bf16 (dependent)", in, out, cyc);
    run<6>("8 v_accvgpr_write", in, out, cyc);
    run<7>("8 v_accvgpr_read", in, out, cyc);
    run<8>("8 ds_read_b128 + wait", in, out, cyc);
    run<9>("8 ds_read_b64_tr_b16 + wait", in, out, cyc);
    run<10>("4 x (fma, s_nop 0)", in, out, cyc);
    run<11>("4 x (fma, s_add)", in, out, cyc);
    run<12>("2 mfma", in, out, cyc);
    run<13>("2 x (mfma + 3 fma)", in, out, cyc);
    run<14>("2 x (mfma + 5 fma)", in, out, cyc);
    run<15>("2 x (mfma + lds read + 2 fma)", in, out, cyc);
    run<16>("2 x (mfma + 2 exp + 1 fma)", in, out, cyc);
    run<17>("2 x (mfma + 3 exp)", in, out, cyc);
    return 0;
}
