// Probe: the forward kernel's two kinds of MFMA group, stand-alone (one wave per SIMD, 4 waves per CU,
// all CUs), to see what each ingredient costs: ticks per group of 2 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CL "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35"
// stage A pair: K fragment from LDS, Q fragments in AGPRs a[32:35], accumulators in VGPRs
#define A_LDS(k, addr) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(k) : "v"(addr), "i"(OFF))
#define A_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")")
#define A_MFMA(c0, c1, k) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, a[32:35], %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, a[32:35], %1" : "+v"(c0), "+v"(c1) : "v"(k) : CL)
#define MAX3(x, a, b) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
// stage XB group: V^T fragment by two transposed reads, accumulators a[0:15], a[16:31]
#define X_TR(v0, v1, addr) asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c3\n\tds_read_b64_tr_b16 %1, %2 offset:%c4" : "=&v"(v0), "=&v"(v1) : "v"(addr), "i"(OFF), "i"(OFF + 2048))
#define X_MFMA(v, p0, p1) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %0, %2, a[16:31]" :: "v"(v), "v"(p0), "v"(p1) : CL)
#define FMA(d, s, c, m) asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(d) : "v"(s), "v"(c), "v"(m))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define ADD(x, y) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define NOP asm volatile("s_nop 0")
constexpr int OFF = 4096;
template <int MODE>
__global__ void __launch_bounds__(256, 1) k(const bf16x8* in, float* out, long long* cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 16384; i += 256) ((float*)lds)[i] = 1.0f;
    __syncthreads();
    const unsigned la = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192, lt = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 8192;
    bf16x8 q = in[threadIdx.x], p0 = in[threadIdx.x + 256], p1 = p0;
    asm volatile("v_accvgpr_write_b32 a32, %0\n\tv_accvgpr_write_b32 a33, %0\n\tv_accvgpr_write_b32 a34, %0\n\tv_accvgpr_write_b32 a35, %0" :: "v"(q[0]) : CL);
    f32x16 c0 = {0}, c1 = {0}, s0, s1;
    for (int i = 0; i < 16; ++i) { s0[i] = 0.01f * i + threadIdx.x * 1e-4f; s1[i] = 0.02f * i; }
    float m0 = -1e30f, m1 = -1e30f, c2 = 0.1f, mb = 0.5f, l0 = 0, l1 = 0;
    float t0, t1, t2, t3; unsigned w0 = 0, w1 = 0;
    bf16x8 ka, kb; bf16x4 v0, v1, n0, n1;
    A_LDS(ka, la); X_TR(v0, v1, lt);
    long long tm0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE == 0) { A_LDS(kb, la); A_WAIT(1); A_MFMA(c0, c1, ka); MAX3(m0, s0[2 * g], s0[2 * g + 1]); MAX3(m1, s1[2 * g], s1[2 * g + 1]); ka = kb; }
            if (MODE == 1) { A_LDS(kb, la); A_WAIT(1); A_MFMA(c0, c1, ka); ka = kb; }
            if (MODE == 2) { A_MFMA(c0, c1, ka); MAX3(m0, s0[2 * g], s0[2 * g + 1]); MAX3(m1, s1[2 * g], s1[2 * g + 1]); }
            if (MODE == 3) { A_MFMA(c0, c1, ka); }
            bf16x8 vf; for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
            if (MODE == 10 || MODE == 11) {   // full XB group (11: with the two pad nops hipcc adds)
                X_TR(n0, n1, lt); A_WAIT(2); X_MFMA(vf, p0, p1);
                if (MODE == 11) NOP;
                FMA(t0, s0[2 * g], c2, mb); FMA(t1, s0[2 * g + 1], c2, mb); FMA(t2, s1[2 * g], c2, mb); FMA(t3, s1[2 * g + 1], c2, mb);
                EXP(t0); EXP(t1); EXP(t2); EXP(t3);
                CVT(w0, t0, t1); ADD(l0, t0); ADD(l0, t1); CVT(w1, t2, t3); ADD(l1, t2); ADD(l1, t3);
                if (MODE == 11) NOP;
                v0 = n0; v1 = n1;
            }
            if (MODE == 12) { X_TR(n0, n1, lt); A_WAIT(2); X_MFMA(vf, p0, p1); v0 = n0; v1 = n1; }      // reads + MFMAs only
            if (MODE == 13) {   // no LDS
                X_MFMA(vf, p0, p1);
                FMA(t0, s0[2 * g], c2, mb); FMA(t1, s0[2 * g + 1], c2, mb); FMA(t2, s1[2 * g], c2, mb); FMA(t3, s1[2 * g + 1], c2, mb);
                EXP(t0); EXP(t1); EXP(t2); EXP(t3);
                CVT(w0, t0, t1); ADD(l0, t0); ADD(l0, t1); CVT(w1, t2, t3); ADD(l1, t2); ADD(l1, t3);
            }
            if (MODE == 14) {   // no exps
                X_TR(n0, n1, lt); A_WAIT(2); X_MFMA(vf, p0, p1);
                FMA(t0, s0[2 * g], c2, mb); FMA(t1, s0[2 * g + 1], c2, mb); FMA(t2, s1[2 * g], c2, mb); FMA(t3, s1[2 * g + 1], c2, mb);
                CVT(w0, t0, t1); ADD(l0, t0); ADD(l0, t1); CVT(w1, t2, t3); ADD(l1, t2); ADD(l1, t3);
                v0 = n0; v1 = n1;
            }
            if (MODE == 15) { X_MFMA(vf, p0, p1); }
            if (MODE == 16) {   // interleaved order: exp right after its fma, sums through one add per pair
                X_TR(n0, n1, lt); A_WAIT(2); X_MFMA(vf, p0, p1);
                FMA(t0, s0[2 * g], c2, mb); FMA(t1, s0[2 * g + 1], c2, mb); EXP(t0); EXP(t1);
                FMA(t2, s1[2 * g], c2, mb); FMA(t3, s1[2 * g + 1], c2, mb); EXP(t2); EXP(t3);
                CVT(w0, t0, t1); ADD(t0, t1); ADD(l0, t0); CVT(w1, t2, t3); ADD(t2, t3); ADD(l1, t2);
                v0 = n0; v1 = n1;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    long long tm1 = __builtin_amdgcn_s_memtime();
    float r = m0 + m1 + l0 + l1 + w0 + w1; for (int j = 0; j < 16; ++j) r += c0[j] + c1[j];
    float a0v; asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(a0v) :: CL); r += a0v;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = tm1 - tm0;
}
template <int MODE> void run(const char* name, const bf16x8* in, float* out, long long* cyc)
{
    const int iters = 4000, blocks = 256;
    long long h[256 * 4];
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks * 4; ++i) m += h[i];
    printf("%-64s %6.1f ticks per group of 2 MFMAs\n", name, m / (blocks * 4) / iters / 8);
}
int main()
{
    bf16x8* in; float* out; long long* cyc;
    (void)hipMalloc(&in, 512 * 16); (void)hipMemset(in, 0x3c, 512 * 16); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 4 * 8);
    run<3>("A: 2 mfma (B in AGPR, VGPR acc)", in, out, cyc);
    run<2>("A: 2 mfma + 2 max3", in, out, cyc);
    run<1>("A: ds_read_b128 + wait + 2 mfma", in, out, cyc);
    run<0>("A: ds_read_b128 + wait + 2 mfma + 2 max3", in, out, cyc);
    run<15>("X: 2 mfma (AGPR acc)", in, out, cyc);
    run<12>("X: 2 tr + wait + 2 mfma", in, out, cyc);
    run<13>("X: 2 mfma + 4 fma + 4 exp + 2 cvt + 4 add", in, out, cyc);
    run<14>("X: 2 tr + wait + 2 mfma + 4 fma + 2 cvt + 4 add (no exp)", in, out, cyc);
    run<10>("X: full group", in, out, cyc);
    run<11>("X: full group + 2 s_nop", in, out, cyc);
    run<16>("X: full group, exp after its fma, pair sums", in, out, cyc);
    return 0;
}
