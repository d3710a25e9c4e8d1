// fa2_util.hip -- small HBM-bound helpers around the attention kernels.
//
//   fill_f32      replaces init_array (reference src/util/cuda_helper.h:60-65): the -inf fill
//                 of the ring's running max that memset cannot do (memo.md:1).
//   f32 <-> bf16  conversions used by the fp32-in convenience paths of the C-ABI.
// All are grid-stride, 16 bytes per lane per access.
#include "fa2_common.h"
#include "fa2_launch.h"

namespace fa2 {

__global__ void fill_f32_kernel(float* p, size_t n, float value)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n / 4;
    f32x4 v = {value, value, value, value};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<f32x4*>(p)[i] = v;
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = value;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n8 = n / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i];
        const f32x4 b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = (__bf16)a[e]; o[4 + e] = (__bf16)b[e]; }
        reinterpret_cast<bf16x8*>(dst)[i] = o;
    }
    for (size_t i = n8 * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = (__bf16)src[i];
}

__global__ void bf16_to_f32_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n8 = n / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const bf16x8 v = reinterpret_cast<const bf16x8*>(src)[i];
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = (float)v[e]; b[e] = (float)v[4 + e]; }
        reinterpret_cast<f32x4*>(dst)[2 * i] = a;
        reinterpret_cast<f32x4*>(dst)[2 * i + 1] = b;
    }
    for (size_t i = n8 * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = (float)src[i];
}

static unsigned grid_for(size_t items)
{
    size_t g = (items + 255) / 256;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;   // 256 CUs x 8 blocks, grid-stride the rest
    return (unsigned)g;
}

hipError_t launch_fill_f32(float* p, size_t n, float value, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream, p, n, value);
    return hipGetLastError();
}

hipError_t launch_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, stream, src, (__bf16*)dst, n);
    return hipGetLastError();
}

hipError_t launch_bf16_to_f32(const void* src, float* dst, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, stream, (const __bf16*)src, dst, n);
    return hipGetLastError();
}

// Ring epilogue for the bf16 state layout (fa2_fwd1_bf16.hip, STATE kernels): the un-normalised accumulator,
// the running sum (in L) and the reference maximum become O = acc / l (bf16) and L = m + ln l -- what the
// last step's finalize switch does, as a pass of its own for schedules in which the last step does not
// touch every row (causal zig-zag ring).  One thread per 8 columns.
__global__ void __launch_bounds__(256) finalize_state_kernel(const float* Oacc, const float* M, float* L, __bf16* O,
                                                               size_t rows, int d)
{
    const int cpr = d / 8;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cpr) return;
    const size_t row = i / cpr;
    const int c = (int)(i % cpr);
    const float l = L[row];
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    const float* a = Oacc + row * d + 8 * c;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)(a[e] * inv);
    *reinterpret_cast<bf16x8*>(O + row * d + 8 * c) = o;
    __syncthreads();          // every thread of a row has read l before one of them replaces it
    if (c == 0) L[row] = M[row] + __builtin_logf(l);
}

hipError_t launch_finalize_state(const float* Oacc, const float* M, float* L, void* O, size_t rows, int d, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    const size_t n = rows * (size_t)(d / 8);
    hipLaunchKernelGGL(finalize_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, Oacc, M, L, (__bf16*)O, rows, d);
    return hipGetLastError();
}

// acc (fp32) = or += src (bf16): the ring backward keeps its running gradients in fp32 and adds each step's
// bf16 contribution to them.  blockIdx.y walks `rows` runs of `cols` elements, `pitch` elements apart.
__global__ void __launch_bounds__(256) accumulate_bf16_kernel(float* acc, const __bf16* src, size_t rows, size_t cols,
                                                                size_t pitch, int init)
{
    const size_t stride = (size_t)gridDim.x * 256 * 8;
    for (size_t row = blockIdx.y; row < rows; row += gridDim.y) {
        float* a = acc + row * pitch;
        const __bf16* b = src + row * pitch;
        for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8; i < cols; i += stride) {
            if (i + 8 <= cols) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(b + i);
                f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
                if (!init) { a0 = *reinterpret_cast<const f32x4*>(a + i); a1 = *reinterpret_cast<const f32x4*>(a + i + 4); }
#pragma unroll
                for (int e = 0; e < 4; ++e) { a0[e] += (float)v[e]; a1[e] += (float)v[4 + e]; }
                *reinterpret_cast<f32x4*>(a + i) = a0;
                *reinterpret_cast<f32x4*>(a + i + 4) = a1;
            } else {
                for (size_t j = i; j < cols; ++j) a[j] = (init ? 0.0f : a[j]) + (float)b[j];
            }
        }
    }
}

hipError_t launch_accumulate_bf16(float* acc, const void* src, size_t rows, size_t cols, size_t pitch, int init, hipStream_t stream)
{
    if (rows == 0 || cols == 0) return hipSuccess;
    const unsigned gy = (unsigned)(rows < 1024 ? rows : 1024);
    unsigned gx = grid_for(cols / 8 + 1);
    if (gy > 1 && gx > 2048 / gy + 1) gx = 2048 / gy + 1;
    hipLaunchKernelGGL(accumulate_bf16_kernel, dim3(gx, gy), dim3(256), 0, stream, acc, (const __bf16*)src, rows, cols, pitch, init);
    return hipGetLastError();
}

// Per XCC two 64-bit counters into out[2 x] , out[2 x + 1] (x = HW_REG_XCC_ID, 16 possible values): s_memtime (ticks of the
// shader clock) and s_memrealtime (the constant 100 MHz reference).  Two calls bracket a stretch of stream work;
// d(memtime) / d(memrealtime) x 100 MHz PER XCC is the mean shader clock the chip held over it (MI355X_MICROARCH.md, DVFS
// give-back item 6) -- what bench.py's `sustained` object reports.  Per XCC because s_memtime is an XCC's own counter: a
// one-workgroup kernel lands on whichever XCC the dispatcher's rotation points at, and two such samples taken on different
// XCCs differ by the counters' offset, not by elapsed clocks (round 3's rehearsal read "225 MHz" that way as soon as a
// second process shared the GPU and moved the rotation).  64 workgroups reach every XCC of the device; the workgroups of
// one XCC write the same slot with 16-byte stores a few hundred clocks apart: any of them will do.
__global__ void __launch_bounds__(64) read_clocks_kernel(unsigned long long* out)
{
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (threadIdx.x == 0) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 v = {__builtin_amdgcn_s_memtime(), __builtin_amdgcn_s_memrealtime()};
        *reinterpret_cast<u64x2*>(out + 2 * xcc) = v;
    }
}

hipError_t launch_read_clocks(unsigned long long* out, hipStream_t stream)
{
    hipLaunchKernelGGL(read_clocks_kernel, dim3(64), dim3(64), 0, stream, out);
    return hipGetLastError();
}

// Measurement aid: what THIS device sustains on nothing but v_mfma_f32_32x32x16_bf16 -- every SIMD, one wave each, four
// independent accumulators, operands from `in` (the caller fills it with random bf16: the clock a device holds depends on the
// operand data) -- so that a bench line can say how far the kernels are from the box they ran on, not only from the nominal
// peak: devices of one pool differ by several per cent on power-limited kernels.  tools/probes/mfma_power.hip, mode 0.
__global__ void __launch_bounds__(256) mfma_probe_kernel(const bf16x8* __restrict__ in, float* __restrict__ out, int iters)
{
    bf16x8 a0 = in[threadIdx.x], b0 = in[threadIdx.x + 256], a1 = in[threadIdx.x + 512], b1 = in[threadIdx.x + 768];
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
#pragma unroll 4
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a0), "v"(b0));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a1), "v"(b1));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a0), "v"(b1));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a1), "v"(b0));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float r = 0.0f;
    for (int q = 0; q < 16; ++q) r += c0[q] + c1[q] + c2[q] + c3[q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

hipError_t launch_mfma_probe(const void* in, float* out, int iters, int blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16x8*)in, out, iters);
    return hipGetLastError();
}

}  // namespace fa2
