// Probe: cycles per v_mfma_f32_32x32x16_bf16 for 1 / 2 / 4 interleaved accumulation chains, with
// VGPR or AGPR accumulators (one wave per SIMD, operands in registers).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MV(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define MA(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define MB(c, a) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[128:131], %0" : "+v"(c) : "v"(a) : "a128","a129","a130","a131")
#define MN(c, a, b) asm volatile("s_nop 0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define MA2(c, a) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[128:131], %1, %0" : "+v"(c) : "v"(a) : "a128","a129","a130","a131")
template <int MODE>
__global__ void __launch_bounds__(256, 1) k(const bf16x8* in, float* out, long long* cyc, int iters)
{
    bf16x8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { MV(c0, a, b); MV(c0, a, b); MV(c0, a, b); MV(c0, a, b); }
        if (MODE == 1) { MV(c0, a, b); MV(c1, a, b); MV(c0, a, b); MV(c1, a, b); }
        if (MODE == 2) { MV(c0, a, b); MV(c1, a, b); MV(c2, a, b); MV(c3, a, b); }
        if (MODE == 3) { MA(c0, a, b); MA(c0, a, b); MA(c0, a, b); MA(c0, a, b); }
        if (MODE == 4) { MA(c0, a, b); MA(c1, a, b); MA(c0, a, b); MA(c1, a, b); }
        if (MODE == 5) { MA(c0, a, b); MA(c1, a, b); MA(c2, a, b); MA(c3, a, b); }
        if (MODE == 6) { MB(c0, a); MB(c1, a); MB(c2, a); MB(c3, a); }
        if (MODE == 7) { MN(c0, a, b); MN(c1, a, b); MN(c2, a, b); MN(c3, a, b); }
        if (MODE == 8) { MA2(c0, a); MA2(c1, a); MA2(c2, a); MA2(c3, a); }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
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
    printf("%-28s %.1f cycles per MFMA\n", name, m / blocks / (4.0 * iters));
}
int main()
{
    bf16x8* in; float* out; long long* cyc;
    hipMalloc(&in, 512 * 16); hipMemset(in, 0x3c, 512 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("VGPR acc, 1 chain", in, out, cyc); run<1>("VGPR acc, 2 chains", in, out, cyc); run<2>("VGPR acc, 4 chains", in, out, cyc);
    run<6>("VGPR acc, B operand in AGPR", in, out, cyc); run<7>("AGPR acc, s_nop 0 before each", in, out, cyc);
    run<8>("VGPR acc, A operand in AGPR", in, out, cyc);
    run<3>("AGPR acc, 1 chain", in, out, cyc); run<4>("AGPR acc, 2 chains", in, out, cyc); run<5>("AGPR acc, 4 chains", in, out, cyc);
    return 0;
}
