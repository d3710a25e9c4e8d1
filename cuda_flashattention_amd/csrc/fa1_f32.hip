// fa1_f32.hip -- the FlashAttention-1 step of the reference's staircase, as a DIDACTIC baseline row.
//
// Restates flash_attention_kernel + its host wrapper (reference src/01_flash_attention_v1/flash_attention_kernel.cu:6-164,
// main.cu:7-70): fp32, one head, one thread per query row, K/V tiles of Bc rows staged in on-chip memory, and the FA1 update
// rule that keeps O NORMALISED after every tile --
//     m_new = max(m, m_tile),  l_new = e^{m - m_new} l + e^{m_tile - m_new} l_tile,
//     O     = (l / l_new) e^{m - m_new} O + (e^{m_tile - m_new} / l_new) P_tile V_tile          (:128-146)
// with the running (l, m) written back beside O (:147-153), scale = 1/sqrt(d) (main.cu:58).
// It is here to complete the staircase and to give the bench table a "what the teaching kernel's
// algorithm does on this chip" row; nothing in the hot path uses it and it is not tuned: scalar fp32 FMAs, no MFMA.
// What is MI355X about it: a workgroup is ONE 64-lane wave (64 query rows), so the tile loop needs no barrier beyond the
// staging one; the lane keeps its O row, m and l in registers across the whole key sweep instead of round-tripping them
// through HBM per tile as the reference does; K/V rows are read from LDS as broadcasts (every lane reads the same key).
#include <cfloat>

#include "fa2_common.h"
#include "fa2_launch.h"

namespace fa2 {

constexpr int kFa1Rows = 64;       // query rows per workgroup = one wave
constexpr int kFa1Bc = 64;         // most keys per staged tile (the caller's Bc, clamped to 1 .. 64: LDS is sized for this)

template <int DMAX>
__global__ void __launch_bounds__(kFa1Rows) fa1_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V, float* __restrict__ O,
                                                           float* __restrict__ l_out, float* __restrict__ m_out, int N, int d,
                                                           float scale, int Bc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Kt = reinterpret_cast<float*>(smem);            // [Bc][d]
    float* Vt = Kt + kFa1Bc * d;                            // [Bc][d]
    float* St = Vt + kFa1Bc * d;                            // [rows][Bc + 1]: the lane's score row, as the reference's S tile (:37)
    const int lane = threadIdx.x;
    const int row = blockIdx.x * kFa1Rows + lane;
    const int qr = row < N ? row : N - 1;                   // clamped: every lane stays in the loop (barriers)

    float q[DMAX], o[DMAX];
#pragma unroll
    for (int c = 0; c < DMAX; ++c) {
        q[c] = c < d ? Q[(size_t)qr * d + c] : 0.0f;
        o[c] = 0.0f;
    }
    float m = -FLT_MAX, l = 0.0f;                           // main.cu:22-31

    for (int j0 = 0; j0 < N; j0 += Bc) {                    // Tc = ceil(N / Bc) tiles of the caller's Bc keys (main.cu:36)
        const int nj = min(Bc, N - j0);
        __syncthreads();                                    // the previous tile has been consumed
        for (int i = lane; i < nj * d; i += kFa1Rows) {
            Kt[i] = K[(size_t)j0 * d + i];
            Vt[i] = V[(size_t)j0 * d + i];
        }
        __syncthreads();
        // pass 1: the tile's scores and their maximum (:101-117)
        float* srow = St + lane * (kFa1Bc + 1);             // +1: the lanes' rows start on different banks
        float mt = -FLT_MAX;
        for (int j = 0; j < nj; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < DMAX; ++c)
                if (c < d) acc += q[c] * Kt[j * d + c];
            acc *= scale;
            mt = fmaxf(mt, acc);
            srow[j] = acc;
        }
        // pass 2: P = exp(S - m_tile) (kept in the score row, as the reference keeps its P tile, :38) and l_tile (:118-127)
        const float m_new = fmaxf(m, mt);
        float lt = 0.0f;
        for (int j = 0; j < nj; ++j) {
            const float p = expf(srow[j] - mt);
            lt += p;
            srow[j] = p;
        }
        // the FA1 update: O stays normalised after every tile (:128-146)
        const float a_old = expf(m - m_new), a_new = expf(mt - m_new);
        const float l_new = a_old * l + a_new * lt;
        const float w_old = l / l_new * a_old, w_new = a_new / l_new;
#pragma unroll
        for (int c = 0; c < DMAX; ++c) o[c] *= w_old;
        for (int j = 0; j < nj; ++j) {
            const float p = w_new * srow[j];
#pragma unroll
            for (int c = 0; c < DMAX; ++c)
                if (c < d) o[c] += p * Vt[j * d + c];
        }
        m = m_new;
        l = l_new;
    }
    if (row < N) {
#pragma unroll
        for (int c = 0; c < DMAX; ++c)
            if (c < d) O[(size_t)row * d + c] = o[c];
        l_out[row] = l;
        m_out[row] = m;
    }
}

hipError_t launch_fa1_f32(const float* Q, const float* K, const float* V, float* O, float* l, float* m, int N, int d, int Bc,
                          hipStream_t stream)
{
    Bc = Bc < 1 ? 1 : Bc > kFa1Bc ? kFa1Bc : Bc;
    const dim3 grid((unsigned)((N + kFa1Rows - 1) / kFa1Rows));
    const size_t lds = ((size_t)2 * kFa1Bc * d + (size_t)kFa1Rows * (kFa1Bc + 1)) * sizeof(float);
    const float scale = 1.0f / sqrtf((float)d);
    if (d <= 16) hipLaunchKernelGGL((fa1_f32_kernel<16>), grid, dim3(kFa1Rows), lds, stream, Q, K, V, O, l, m, N, d, scale, Bc);
    else if (d <= 64) hipLaunchKernelGGL((fa1_f32_kernel<64>), grid, dim3(kFa1Rows), lds, stream, Q, K, V, O, l, m, N, d, scale, Bc);
    else hipLaunchKernelGGL((fa1_f32_kernel<128>), grid, dim3(kFa1Rows), lds, stream, Q, K, V, O, l, m, N, d, scale, Bc);
    return hipGetLastError();
}

}  // namespace fa2
