// fa2_f32.hip -- the reference's own arithmetic type on gfx950: fp32 in, fp32 out, exact f32
// products on v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain, 1/16 of the bf16 MFMA rate).
//
// This family backs the reference-signature drop-ins flash_attention_2_forward /
// flash_attention_2_backward (include/fa2_mi355x.h): any seq_len, any head_dim <= 128, so the
// reference's literal 4x4 known-answer cases (02_forward/main.cu:134-155,
// 02_backward/main.cu:78-107) and its fp32 gates (1e-4 / 1e-3 / 5e-3) can be met as written.
// It is the parity path, not the fast path: plain single-buffered LDS tiles, no pipelining.
//
// Same algorithm and the same MFMA orientation as the bf16 kernels (fa2_fwd1_bf16.hip,
// fa2_bwd_bf16.hip), with the f32 operand map: one f32 per lane per operand,
// A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31], two k per instruction.
// An accumulator register r of lane-half h is row acc_row(r, h); feeding register r as the B
// operand of the next product therefore contracts over rows acc_row(r, 0) and acc_row(r, 1),
// and the A operand reads exactly those rows from LDS -- no packing, no transposed read.
#include "fa2_common.h"
#include "fa2_launch.h"

namespace fa2 {

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

constexpr int kF32Tile = 32;     // streamed rows per tile (keys or queries)
constexpr int kF32Waves = 4;     // waves per workgroup: 128 owned rows

// Stage `rows` x d floats (row-major, leading dimension d) from global row `g0` on into an LDS
// image [32][RS] with zero fill for rows >= limit and columns >= d.
template <int DP>
__device__ __forceinline__ void stage_tile_f32(float* dst, const float* src, int g0, int limit, int d)
{
    constexpr int RS = DP + 1;
    for (int i = threadIdx.x; i < kF32Tile * DP; i += 64 * kF32Waves) {
        const int row = i / DP, c = i % DP;
        const int g = g0 + row;
        dst[row * RS + c] = (g < limit && c < d) ? src[(size_t)g * d + c] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------ forward
template <int DT, bool CAUSAL>
__global__ void __launch_bounds__(256, 1) fa2_fwd_f32_kernel(F32Args p)
{
    constexpr int DP = 32 * DT;
    constexpr int RS = DP + 1;
    constexpr int KS = DP / 2;
    __shared__ float Ks[kF32Tile * RS];
    __shared__ float Vs[kF32Tile * RS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = lane & 31, h = lane >> 5;
    const int N = p.N, d = p.d;
    const int Nk = p.Nk > 0 ? p.Nk : N;
    const int nrb = (N + 32 * kF32Waves - 1) / (32 * kF32Waves);
    const int head = blockIdx.x / nrb, rb = blockIdx.x % nrb;
    const float* Qh = p.Q + (size_t)head * N * d;
    const float* Kh = p.K + (size_t)head * Nk * d;
    const float* Vh = p.V + (size_t)head * Nk * d;

    const int q0 = rb * 32 * kF32Waves + wave * 32;
    const int qrow = q0 + qi;
    const int qld = qrow < N ? qrow : N - 1;

    float qreg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 2 * s + h;
        qreg[s] = c < d ? Qh[(size_t)qld * d + c] : 0.0f;
    }
    f32x16 oacc[DT];
    float m_run = -INFINITY, l_run = 0.0f;
    if (p.resume) {
        const float* Oq = p.O + ((size_t)head * N + qld) * d;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * dt + acc_row(r, h);
                oacc[dt][r] = c < d ? Oq[c] : 0.0f;
            }
        m_run = p.M[(size_t)head * N + qld];
        l_run = h == 0 ? p.L[(size_t)head * N + qld] : 0.0f;
    } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.0f;
    }

    int ntiles = (Nk + kF32Tile - 1) / kF32Tile;
    if (CAUSAL) ntiles = min(ntiles, min(rb * 32 * kF32Waves + 32 * kF32Waves - 1, N - 1) / kF32Tile + 1);
    const float c2 = p.scale * kLog2e;

    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * kF32Tile;
        __syncthreads();
        stage_tile_f32<DP>(Ks, Kh, key0, Nk, d);
        stage_tile_f32<DP>(Vs, Vh, key0, Nk, d);
        __syncthreads();
        bool active = true;
        if (CAUSAL) active = key0 <= q0 + 31;
        if (!active) continue;

        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) sacc = mfma_f32(Ks[qi * RS + 2 * s + h], qreg[s], sacc);   // S^T[key][q]

        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + acc_row(r, h);
            bool dead = key >= Nk;
            if (CAUSAL) dead = dead || key > qrow;
            if (dead) sacc[r] = -INFINITY;
            mx = fmaxf(mx, sacc[r]);
        }
        mx = half_max(mx) * p.scale;
        const float m_new = fmaxf(m_run, mx);
        const float alpha = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        m_run = m_new;
        const float mb = m_run == -INFINITY ? 0.0f : m_run * kLog2e;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = __builtin_amdgcn_exp2f(sacc[r] * c2 - mb);
            l_run += sacc[r];
        }
        // O^T[dcol][q] += V^T[dcol][key] P^T[key][q]
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                oacc[dt] = mfma_f32(Vs[acc_row(r, h) * RS + 32 * dt + qi], sacc[r], oacc[dt]);
    }

    const float l_tot = half_sum(l_run);
    if (qrow < N) {
        const float inv = !p.finalize ? 1.0f : (l_tot > 0.0f ? 1.0f / l_tot : 0.0f);
        float* Oq = p.O + ((size_t)head * N + qrow) * d;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * dt + acc_row(r, h);
                if (c < d) Oq[c] = oacc[dt][r] * inv;
            }
        if (h == 0) {
            if (p.finalize) {
                p.L[(size_t)head * N + qrow] = m_run + logf(l_tot);
            } else {
                p.L[(size_t)head * N + qrow] = l_tot;
                p.M[(size_t)head * N + qrow] = m_run;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ delta
__global__ void fa2_delta_f32_kernel(const float* __restrict__ dO, const float* __restrict__ O,
                                     float* __restrict__ Dv, size_t rows, int d)
{
    const size_t row = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    float acc = 0.0f;
    if (row < rows)
        for (int c = sub; c < d; c += 16) acc += dO[row * d + c] * O[row * d + c];
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 16);
    if (row < rows && sub == 0) Dv[row] = acc;
}

// ------------------------------------------------------------------------------------ dQ
template <int DT, bool CAUSAL>
__global__ void __launch_bounds__(256, 1) fa2_dq_f32_kernel(F32Args p)
{
    constexpr int DP = 32 * DT;
    constexpr int RS = DP + 1;
    constexpr int KS = DP / 2;
    __shared__ float Ks[kF32Tile * RS];
    __shared__ float Vs[kF32Tile * RS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qi = lane & 31, h = lane >> 5;
    const int N = p.N, d = p.d;
    const int nrb = (N + 32 * kF32Waves - 1) / (32 * kF32Waves);
    const int head = blockIdx.x / nrb, rb = blockIdx.x % nrb;
    const size_t slab = (size_t)head * N * d;
    const float* Qh = p.Q + slab;
    const float* Kh = p.K + slab;
    const float* Vh = p.V + slab;
    const float* Gh = p.dO + slab;

    const int q0 = rb * 32 * kF32Waves + wave * 32;
    const int qrow = q0 + qi;
    const int qld = qrow < N ? qrow : N - 1;

    float qreg[KS], greg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 2 * s + h;
        qreg[s] = c < d ? Qh[(size_t)qld * d + c] : 0.0f;
        greg[s] = c < d ? Gh[(size_t)qld * d + c] : 0.0f;
    }
    const float Lq = p.L[(size_t)head * N + qld] * kLog2e;
    const float Dq = p.D[(size_t)head * N + qld];
    const float c2 = p.scale * kLog2e;

    f32x16 dqacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.0f;

    int ntiles = (N + kF32Tile - 1) / kF32Tile;
    if (CAUSAL) ntiles = min(ntiles, min(rb * 32 * kF32Waves + 32 * kF32Waves - 1, N - 1) / kF32Tile + 1);

    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * kF32Tile;
        __syncthreads();
        stage_tile_f32<DP>(Ks, Kh, key0, N, d);
        stage_tile_f32<DP>(Vs, Vh, key0, N, d);
        __syncthreads();
        bool active = true;
        if (CAUSAL) active = key0 <= q0 + 31;
        if (!active) continue;

        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.0f; dpacc[r] = 0.0f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            sacc = mfma_f32(Ks[qi * RS + 2 * s + h], qreg[s], sacc);     // S^T[key][q]
            dpacc = mfma_f32(Vs[qi * RS + 2 * s + h], greg[s], dpacc);   // dP^T[key][q]
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + acc_row(r, h);
            bool dead = key >= N;
            if (CAUSAL) dead = dead || key > qrow;
            const float pr = dead ? 0.0f : __builtin_amdgcn_exp2f(sacc[r] * c2 - Lq);
            sacc[r] = pr * (dpacc[r] - Dq);                                // dS^T[key][q]
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dqacc[dt] = mfma_f32(Ks[acc_row(r, h) * RS + 32 * dt + qi], sacc[r], dqacc[dt]);
    }

    if (qrow < N) {
        float* dQq = p.dQ + slab + (size_t)qrow * d;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * dt + acc_row(r, h);
                if (c < d) dQq[c] = dqacc[dt][r] * p.scale;
            }
    }
}

// ------------------------------------------------------------------------------------ dK, dV
template <int DT, bool CAUSAL>
__global__ void __launch_bounds__(256, 1) fa2_dkdv_f32_kernel(F32Args p)
{
    constexpr int DP = 32 * DT;
    constexpr int RS = DP + 1;
    constexpr int KS = DP / 2;
    __shared__ float Qs[kF32Tile * RS];
    __shared__ float Gs[kF32Tile * RS];
    __shared__ float rcs[2 * kF32Tile];      // -L/scale, -D of the tile's 32 query rows

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ki = lane & 31, h = lane >> 5;
    const int N = p.N, d = p.d;
    const int ncb = (N + 32 * kF32Waves - 1) / (32 * kF32Waves);
    const int head = blockIdx.x / ncb, cb = blockIdx.x % ncb;
    const size_t slab = (size_t)head * N * d;
    const float* Qh = p.Q + slab;
    const float* Kh = p.K + slab;
    const float* Vh = p.V + slab;
    const float* Gh = p.dO + slab;
    const float* Lh = p.L + (size_t)head * N;
    const float* Dh = p.D + (size_t)head * N;

    const int k0 = cb * 32 * kF32Waves + wave * 32;
    const int key = k0 + ki;
    const int kld = key < N ? key : N - 1;

    float kreg[KS], vreg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 2 * s + h;
        kreg[s] = c < d ? Kh[(size_t)kld * d + c] : 0.0f;
        vreg[s] = c < d ? Vh[(size_t)kld * d + c] : 0.0f;
    }
    f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[dt][r] = 0.0f; dvacc[dt][r] = 0.0f; }

    const int ntiles = (N + kF32Tile - 1) / kF32Tile;
    int t0 = 0;
    if (CAUSAL) t0 = (cb * 32 * kF32Waves) / kF32Tile;
    const float c2 = p.scale * kLog2e;
    const float inv_scale = 1.0f / p.scale;

    for (int t = t0; t < ntiles; ++t) {
        const int qb0 = t * kF32Tile;
        __syncthreads();
        stage_tile_f32<DP>(Qs, Qh, qb0, N, d);
        stage_tile_f32<DP>(Gs, Gh, qb0, N, d);
        if (threadIdx.x < 64) {
            int qr = qb0 + (threadIdx.x & 31);
            qr = qr < N ? qr : N - 1;
            rcs[threadIdx.x] = threadIdx.x < 32 ? -Lh[qr] * inv_scale : -Dh[qr];
        }
        __syncthreads();
        bool active = true;
        if (CAUSAL) active = qb0 + kF32Tile - 1 >= k0;
        if (!active) continue;

        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = rcs[acc_row(r, h)];
            dpacc[r] = rcs[32 + acc_row(r, h)];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            sacc = mfma_f32(Qs[ki * RS + 2 * s + h], kreg[s], sacc);     // S'[q][key]
            dpacc = mfma_f32(Gs[ki * RS + 2 * s + h], vreg[s], dpacc);   // dP'[q][key]
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = qb0 + acc_row(r, h);
            bool dead = q >= N;
            if (CAUSAL) dead = dead || key > q;
            const float pr = dead ? 0.0f : __builtin_amdgcn_exp2f(sacc[r] * c2);
            sacc[r] = pr;
            dpacc[r] = pr * dpacc[r];
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = acc_row(r, h) * RS + 32 * dt + ki;
                dvacc[dt] = mfma_f32(Gs[o], sacc[r], dvacc[dt]);     // dV^T[dcol][key] += dO^T P
                dkacc[dt] = mfma_f32(Qs[o], dpacc[r], dkacc[dt]);    // dK^T[dcol][key] += Q^T dS
            }
    }

    if (key < N) {
        float* dKk = p.dK + slab + (size_t)key * d;
        float* dVk = p.dV + slab + (size_t)key * d;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * dt + acc_row(r, h);
                if (c < d) { dKk[c] = dkacc[dt][r] * p.scale; dVk[c] = dvacc[dt][r]; }
            }
    }
}

// ------------------------------------------------------------------------------------ launch
template <int DT, bool CAUSAL>
static hipError_t fwd_one(const F32Args& a, hipStream_t stream)
{
    const int nrb = (a.N + 32 * kF32Waves - 1) / (32 * kF32Waves);
    hipLaunchKernelGGL((fa2_fwd_f32_kernel<DT, CAUSAL>), dim3((unsigned)(nrb * a.BH)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

template <int DT, bool CAUSAL>
static hipError_t bwd_one(const F32Args& a, hipStream_t stream)
{
    hipError_t e = hipSuccess;
    const size_t rows = (size_t)a.BH * a.N;
    if (a.phases & 1) {
        hipLaunchKernelGGL(fa2_delta_f32_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, stream,
                           a.dO, (const float*)a.O, a.D, rows, a.d);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const int nb = (a.N + 32 * kF32Waves - 1) / (32 * kF32Waves);
    if (a.phases & 2) {
        hipLaunchKernelGGL((fa2_dq_f32_kernel<DT, CAUSAL>), dim3((unsigned)(nb * a.BH)), dim3(256), 0, stream, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (a.phases & 4) {
        hipLaunchKernelGGL((fa2_dkdv_f32_kernel<DT, CAUSAL>), dim3((unsigned)(nb * a.BH)), dim3(256), 0, stream, a);
        e = hipGetLastError();
    }
    return e;
}

#define FA2_F32_DISPATCH(fn)                                                        \
    switch ((a.d + 31) / 32) {                                                      \
    case 1: return a.causal ? fn<1, true>(a, stream) : fn<1, false>(a, stream);     \
    case 2: return a.causal ? fn<2, true>(a, stream) : fn<2, false>(a, stream);     \
    case 3: return a.causal ? fn<3, true>(a, stream) : fn<3, false>(a, stream);     \
    case 4: return a.causal ? fn<4, true>(a, stream) : fn<4, false>(a, stream);     \
    default: return hipErrorInvalidValue;                                           \
    }

hipError_t launch_fwd_f32(const F32Args& a, hipStream_t stream) { FA2_F32_DISPATCH(fwd_one) }
hipError_t launch_bwd_f32(const F32Args& a, hipStream_t stream) { FA2_F32_DISPATCH(bwd_one) }

}  // namespace fa2
