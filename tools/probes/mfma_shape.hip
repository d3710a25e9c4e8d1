// Probe: which bf16 MFMA shape sustains more FLOP/s under the chip's clock management --
// v_mfma_f32_32x32x16_bf16 (32 clocks, 32768 flop) or v_mfma_f32_16x16x32_bf16 (16 clocks, 16384 flop).
// MI355X_MICROARCH.md 'DVFS give-back' item 7 / cdna_hip_programming.md rule 28: decide by wall time on random
// operands, at the same output tile per wave.  A "unit" below is 32768 flop: one 32x32x16 or two 16x16x32 on the
// same 64 accumulator registers per wave; every variant carries the same fillers PER UNIT, so the two shapes do
// the same work and differ only in how the matrix pipe is fed.
//   mode 0  bare MFMAs
//   mode 1  + one 1 KiB ds_read_b128 per unit, used as the A operand of a later unit
//   mode 2  forward-like mix per unit: 1 ds_read_b128, 1 v_exp_f32, 3 plain VALU
//   modes 4-6  bare MFMAs with operand reuse between consecutive MFMAs: A stationary / nothing shared / both shared
//   mode 3  single-kernel-backward mix per 2 units (tools/isa_mix.py: 1.9 LDS, 1.2 VALU, 0.4 exp, 0.7 SALU,
//           0.4 waits per 32x32x16): 4 ds_read_b64, 1 v_exp_f32, 2 VALU, 1 SALU, 1 counted wait
// Prints, per variant: wall ms, shader clock (s_memtime ticks / wall), ticks per unit per SIMD, TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define M32(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define M16(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define LDS128(d, a, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(d) : "v"(a))
#define LDS64(d, a, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(d) : "v"(a))
#define VALU3(x, y) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_add_f32 %1, %1, %0\n\tv_max_f32 %0, %0, %1" : "+v"(x), "+v"(y))
#define VALU1(x, y) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x), "+v"(y))
#define SALU(s) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc")
#define WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")")
#define NONE do { } while (0)

// One unit of 32768 flop on accumulator slot i, with filler F1 behind its first half and F2 behind its second.
#define UNIT(i, a, b, F1, F2)                                                              \
    do {                                                                                   \
        if constexpr (SHAPE == 32) { M32(c32[i], a, b); F1; F2; }                          \
        else { M16(c16[2 * (i)], a, b); F1; M16(c16[2 * (i) + 1], b, a); F2; }             \
    } while (0)

template <int SHAPE, int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) k(const bf16x8* in, float* out, long long* cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 4096; i += THREADS) ((bf16x8*)lds)[i] = in[(i * 7 + blockIdx.x) & 4095];
    __syncthreads();
    bf16x8 a0 = in[threadIdx.x], b0 = in[threadIdx.x + 512], a1 = in[threadIdx.x + 1024], b1 = in[threadIdx.x + 1536];
    bf16x8 a2 = in[threadIdx.x + 2048], b2 = in[threadIdx.x + 2560], a3 = in[threadIdx.x + 3072], b3 = in[(threadIdx.x + 3584) & 4095];
    f32x16 c32[4] = {};
    f32x4 c16[8] = {};
    float x0 = 0.1f * threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned la = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096, sc = 0;
    bf16x8 l0 = a0, l1 = a1;
    f32x2 h0, h1, h2, h3;
    h0[0] = h0[1] = h1[0] = h1[1] = h2[0] = h2[1] = h3[0] = h3[1] = 0.f;
    asm volatile("" :: "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3));   // operands landed before the loop: no vmcnt wait inside it
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {
            UNIT(0, a0, b0, NONE, NONE); UNIT(1, a1, b1, NONE, NONE); UNIT(2, a0, b1, NONE, NONE); UNIT(3, a1, b0, NONE, NONE);
        }
        if constexpr (MODE == 4) {      // operand-stationary order: consecutive MFMAs share their A operand (4 in a row), then B changes
            UNIT(0, a0, b0, NONE, NONE); UNIT(1, a0, b1, NONE, NONE); UNIT(2, a0, b2, NONE, NONE); UNIT(3, a0, b3, NONE, NONE);
        }
        if constexpr (MODE == 5) {      // every MFMA with both operands different from the previous one's (8 fragments in rotation)
            UNIT(0, a0, b0, NONE, NONE); UNIT(1, a1, b1, NONE, NONE); UNIT(2, a2, b2, NONE, NONE); UNIT(3, a3, b3, NONE, NONE);
        }
        if constexpr (MODE == 6) {      // both operands the same for 4 MFMAs in a row (4 accumulators)
            UNIT(0, a0, b0, NONE, NONE); UNIT(1, a0, b0, NONE, NONE); UNIT(2, a0, b0, NONE, NONE); UNIT(3, a0, b0, NONE, NONE);
        }
        if constexpr (MODE == 1) {
            UNIT(0, a0, b0, LDS128(l0, la, 0), NONE); UNIT(1, a1, b1, NONE, LDS128(l1, la, 1024));
            WAIT(0);
            UNIT(2, l0, b1, NONE, NONE); UNIT(3, l1, b0, NONE, NONE);
            la = (la + 2048) & 0x7fff;
        }
        if constexpr (MODE == 2) {
            UNIT(0, a0, b0, LDS128(l0, la, 0); EXP(x0), VALU3(x1, x2));
            UNIT(1, a1, b1, LDS128(l1, la, 1024); EXP(x3), VALU3(x1, x2));
            WAIT(0);
            UNIT(2, l0, b1, EXP(x0), VALU3(x1, x2));
            UNIT(3, l1, b0, EXP(x3), VALU3(x1, x2));
            la = (la + 2048) & 0x7fff;
        }
        if constexpr (MODE == 3) {
            UNIT(0, a0, b0, LDS64(h0, la, 0); LDS64(h1, la, 512), EXP(x0));
            UNIT(1, a1, b1, LDS64(h2, la, 1024); LDS64(h3, la, 1536), VALU1(x1, x2); VALU1(x2, x3));
            WAIT(2);
            UNIT(2, a0, b1, LDS64(h0, la, 2048); LDS64(h1, la, 2560), EXP(x3); SALU(sc));
            UNIT(3, a1, b0, LDS64(h2, la, 3072); LDS64(h3, la, 3584), VALU1(x1, x2); VALU1(x2, x3); SALU(sc));
            WAIT(2);
            la = (la + 4096) & 0x7fff;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = x0 + x1 + x2 + x3 + (float)l0[0] + (float)l1[0] + h0[0] + h1[1] + h2[0] + h3[1] + (float)sc;
    for (int q = 0; q < 16; ++q) r += c32[0][q] + c32[1][q] + c32[2][q] + c32[3][q];
    for (int q = 0; q < 8; ++q) r += c16[q][0] + c16[q][1] + c16[q][2] + c16[q][3];
    out[blockIdx.x * THREADS + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int SHAPE, int MODE, int THREADS> void run(const char* name, const bf16x8* in, float* out, long long* cyc, int iters)
{
    const int blocks = 256, waves = THREADS / 64;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, in, out, cyc, iters / 4);   // ramp
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, in, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * waves);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= h.size();
    const double units = 4.0 * iters * blocks * waves;
    printf("%-10s %-44s %d w/SIMD %7.1f ms  clock %4.0f MHz  %5.1f ticks/unit/SIMD  %6.0f TFLOP/s\n",
           SHAPE == 32 ? "32x32x16" : "16x16x32", name, waves / 4, ms, m / ms / 1e3, m / (4.0 * iters * (waves / 4)),
           units * 32768.0 / ms / 1e9);
    fflush(stdout);
}

template <int MODE> void both(const char* name, const bf16x8* in, float* out, long long* cyc, int iters)
{
    run<32, MODE, 256>(name, in, out, cyc, iters);
    run<16, MODE, 256>(name, in, out, cyc, iters);
    run<32, MODE, 512>(name, in, out, cyc, iters / 2);
    run<16, MODE, 512>(name, in, out, cyc, iters / 2);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 600000;     // 600k x 4 units x 32 clocks ~ 40 ms per launch at 1.8 GHz
    bf16x8* in; float* out; long long* cyc;
    std::vector<unsigned short> h(4096 * 8);
    srand(1); for (auto& v : h) { float f = (rand() % 2000) / 1000.0f - 1.0f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    (void)hipMalloc(&in, 4096 * 16); (void)hipMemcpy(in, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {     // twice: the second table is the warmed-up one
        printf("--- pass %d\n", rep);
        both<0>("bare", in, out, cyc, iters);
        both<6>("bare, same A and B four times in a row", in, out, cyc, iters);
        both<4>("bare, A stationary over four MFMAs", in, out, cyc, iters);
        both<5>("bare, A and B change every MFMA", in, out, cyc, iters);
        both<1>("+1 KiB ds_read_b128 per unit", in, out, cyc, iters);
        both<2>("+b128, exp, 3 VALU per unit (forward mix)", in, out, cyc, iters);
        both<3>("+4 b64, exp, 2 VALU, SALU, wait per 2 units", in, out, cyc, iters);
    }
    return 0;
}
