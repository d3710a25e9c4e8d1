// fa2_fwd_bf16.hip -- FlashAttention-2 forward for gfx950 (MI355X), bf16 in / fp32 accumulate.
//
// Replaces flash_attention_2_forward_kernel (reference
// src/02_flash_attention_v2_forward/flash_attention_kernel.cu:37-297, cleaned copy
// src/03_flash_attention_v2_ring/common/flash_attention_kernel.cu:13-130) and, through the
// resume/finalize switches, ring_attention_forward_kernel
// (src/03_flash_attention_v2_ring/common/ring_attention_kernel.cu:13-140).  Same maths --
// m' = max(m, rowmax), l = e^{m-m'} l + sum e^{s-m'}, O = e^{m-m'} O + P V, O/l, L = m + ln l
// (attention_helper.h:76-110) -- on a different machine mapping:
//
//   * workgroup = 8 waves = 256 query rows of one head; wave w owns rows 32w..32w+31;
//   * S^T = K Q^T on v_mfma_f32_32x32x16_bf16 ("swapped" product): the accumulator column
//     (= lane & 31) is the QUERY row, its 16 registers are keys, so a row's max and sum are
//     in-lane reductions plus one permlane32_swap -- the reference's (Bc + d) shuffle
//     butterflies per row per tile are gone;
//   * P stays in registers: the S^T accumulator, packed to bf16, IS the B operand of
//     O^T += V^T P^T (contraction over the accumulator's row index), V^T fragments come from
//     LDS through ds_read_b64_tr_b16;
//   * K/V tiles (64 keys) are register-staged into a double-buffered, XOR-swizzled LDS image
//     (fa2_common.h: lds_off) that is conflict-free for both the row and the transposed reads;
//     the next tile's global loads are issued before the current tile's MFMAs, its LDS
//     writes after them: one barrier per tile;
//   * exp2 domain (v_exp_f32), running max kept in natural units so L matches the
//     reference's natural-log LSE; O accumulators are rescaled only in tiles where some
//     row's max actually moved (exact, wave-uniform branch);
//   * work mapping is XCD-aware (fa2_common.h: map_block).
#include "fa2_common.h"
#include "fa2_launch.h"

namespace fa2 {

constexpr int kFwdRows = 256;   // query rows per workgroup
constexpr int kFwdKV = 64;      // keys per tile

template <int D, bool CAUSAL, bool STATE>
__global__ void __launch_bounds__(512, 2) fa2_fwd_bf16_kernel(FwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = D * 2;               // bytes per tile row
    constexpr int TILEB = kFwdKV * ROWB;      // bytes per K (or V) tile
    constexpr int CPR = D / 8;                // 16-byte chunks per row
    constexpr int CPT = kFwdKV * CPR / 512;   // chunks per thread per tensor
    constexpr int KS = D / 16;                // k-steps of QK^T
    constexpr int DT = D / 32;                // 32-column tiles of O

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31;
    const int h = lane >> 5;

    const int nrb = (p.Nq + kFwdRows - 1) / kFwdRows;
    int head, rb;
    map_block(blockIdx.x, p.BH, nrb, head, rb);
    if (CAUSAL) rb = nrb - 1 - rb;            // heaviest row-blocks first

    const int Nq = p.Nq, Nk = p.Nk;
    const char* Qh = (const char*)p.Q + (size_t)head * Nq * ROWB;
    const char* Kh = (const char*)p.K + (size_t)head * Nk * ROWB;
    const char* Vh = (const char*)p.V + (size_t)head * Nk * ROWB;

    const int q0 = rb * kFwdRows + wave * 32;         // first query row of this wave
    const int qrow = q0 + qi;                          // this lane's query row
    const int qld = qrow < Nq ? qrow : Nq - 1;         // clamped for loads (pad, don't mask)

    // Number of K/V tiles this workgroup walks.
    int ntiles = (Nk + kFwdKV - 1) / kFwdKV;
    if (CAUSAL) {
        const int last_q = min(rb * kFwdRows + kFwdRows - 1, Nq - 1);
        const int last_key = last_q + p.causal_shift;             // last visible key index
        const int lim = last_key < 0 ? 0 : last_key / kFwdKV + 1;
        ntiles = min(ntiles, lim);
    }

    // ---- Q fragments: B operand of S^T = K Q^T, lane holds Q[q][16s + 8h .. +7].
    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        qf[s] = *reinterpret_cast<const bf16x8*>(Qh + (size_t)qld * ROWB + 16 * (2 * s + h));

    // ---- running state
    f32x16 oacc[DT];
    float m_run, l_run;
    if (STATE && p.resume) {
        const float* Oa = p.Oacc + ((size_t)head * Nq + qld) * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(Oa + 32 * dt + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) oacc[dt][4 * g + e] = v[e];
            }
        m_run = p.M[(size_t)head * Nq + qld];
        l_run = h == 0 ? p.L[(size_t)head * Nq + qld] : 0.0f;
    } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.0f;
        m_run = -INFINITY;
        l_run = 0.0f;
    }

    // ---- tile staging: global -> registers -> swizzled LDS image
    u32x4 kreg[CPT], vreg[CPT];
    auto stage_load = [&](int t) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + 512 * i;
            const int row = c / CPR, ch = c % CPR;
            int krow = t * kFwdKV + row;
            krow = krow < Nk ? krow : Nk - 1;
            kreg[i] = *reinterpret_cast<const u32x4*>(Kh + (size_t)krow * ROWB + 16 * ch);
            vreg[i] = *reinterpret_cast<const u32x4*>(Vh + (size_t)krow * ROWB + 16 * ch);
        }
    };
    auto stage_write = [&](int buf) {
        char* kb = smem + buf * 2 * TILEB;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + 512 * i;
            const int row = c / CPR, ch = c % CPR;
            const int o = lds_off<D>(row, ch);
            *reinterpret_cast<u32x4*>(kb + o) = kreg[i];
            *reinterpret_cast<u32x4*>(kb + TILEB + o) = vreg[i];
        }
    };

    const float c2 = p.scale * kLog2e;   // exp(s * scale - m) = exp2(s * c2 - m * log2e)

    if (ntiles > 0) {
        stage_load(0);
        stage_write(0);
    }
    __syncthreads();

    // per-lane pieces of the transposed-read address (see lds_read_tr)
    const int trq = (lane & 15) >> 2;       // row inside the 4-row block
    const int trp = lane & 3;               // 4-column group inside the 16-column block
    const int trcb = (lane >> 4) & 1;       // which 16-column half of the 32-column tile

    for (int t = 0; t < ntiles; ++t) {
        const char* Kt = smem + (t & 1) * 2 * TILEB;
        const char* Vt = Kt + TILEB;
        const bool more = t + 1 < ntiles;
        if (more) stage_load(t + 1);

        // A wave whose rows all lie above this tile's keys (causal) has nothing to add.
        const int key0 = t * kFwdKV;
        bool active = true;
        if (CAUSAL) active = key0 <= q0 + 31 + p.causal_shift;

        if (active) {
            // ---- S^T tile: 64 keys x 32 queries = two 32x32 accumulators
            f32x16 sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.0f;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8 a = lds_read_frag(Kt, lds_off<D>(32 * kb + qi, 2 * s + h));
                    sacc[kb] = mfma32(a, qf[s], sacc[kb]);
                }
            }

            // ---- masks: key tail (last tile) and causal diagonal
            const bool tail = key0 + kFwdKV > Nk;
            bool diag = false;
            if (CAUSAL) diag = key0 + kFwdKV - 1 > q0 + p.causal_shift;
            if (tail || diag) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + 32 * kb + acc_row(r, h);
                        bool dead = key >= Nk;
                        if (CAUSAL) dead = dead || key > qrow + p.causal_shift;
                        if (dead) sacc[kb][r] = -INFINITY;
                    }
            }

            // ---- online softmax, one query row per lane (two lanes per row: h = 0, 1)
            float mx = sacc[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[1][r]);
            mx = half_max(mx) * p.scale;
            const float m_new = fmaxf(m_run, mx);
            if (__any(m_new != m_run)) {
                // first tile: m_run = -inf -> alpha = 0, accumulators are 0 anyway
                const float alpha = m_new == -INFINITY ? 1.0f
                                                       : __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                m_run = m_new;
            }
            // a row that has seen no visible key yet keeps p = 0 (avoid inf - inf)
            const float mb = m_run == -INFINITY ? 0.0f : m_run * kLog2e;
            float psum = 0.0f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(sacc[kb][r] * c2 - mb);
                    sacc[kb][r] = e;
                    psum += e;
                }
            l_run += psum;

            bf16x8 pf[2][2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) pf[kb][sp] = pack_acc(sacc[kb], sp);

            // ---- O^T += V^T P^T
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp) {
                        bf16x4 part[2];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int row = 32 * kb + 16 * sp + 8 * jj + 4 * h + trq;
                            const int ch = 4 * dt + 2 * trcb + (trp >> 1);
                            part[jj] = lds_read_tr(Vt, lds_off<D>(row, ch) + 8 * (trp & 1));
                        }
                        bf16x8 vf;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vf[e] = part[0][e]; vf[4 + e] = part[1][e]; }
                        oacc[dt] = mfma32(vf, pf[kb][sp], oacc[dt]);
                    }
        }

        if (more) stage_write((t + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue
    const float l_tot = half_sum(l_run);
    const size_t qoff = (size_t)head * Nq + qrow;
    if (!STATE || p.finalize) {
        const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
        if (qrow < Nq) {
            char* Oq = (char*)p.O + qoff * ROWB;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)(oacc[dt][4 * g + e] * inv);
                    *reinterpret_cast<bf16x4*>(Oq + 2 * (32 * dt + 8 * g + 4 * h)) = o;
                }
            if (h == 0) p.L[qoff] = m_run + __builtin_logf(l_tot);
        }
    } else {
        if (qrow < Nq) {
            float* Oa = p.Oacc + qoff * D;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = oacc[dt][4 * g + e];
                    *reinterpret_cast<f32x4*>(Oa + 32 * dt + 8 * g + 4 * h) = v;
                }
            if (h == 0) {
                p.L[qoff] = l_tot;
                p.M[qoff] = m_run;
            }
        }
    }
}

template <int D, bool CAUSAL, bool STATE>
static hipError_t launch_one(const FwdArgs& a, hipStream_t stream)
{
    constexpr int lds = 2 * 2 * kFwdKV * D * 2;
    auto kern = fa2_fwd_bf16_kernel<D, CAUSAL, STATE>;
    static bool attr_set[64] = {};
    hipError_t e = ensure_dynamic_lds(kern, lds, attr_set);
    if (e != hipSuccess) return e;
    const int nrb = (a.Nq + kFwdRows - 1) / kFwdRows;
    const dim3 grid((unsigned)(nrb * a.BH));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_fwd_bf16(const FwdArgs& a, hipStream_t stream)
{
    const bool state = a.resume || !a.finalize;
    if (a.d == 128) {
        if (state) return a.causal ? launch_one<128, true, true>(a, stream) : launch_one<128, false, true>(a, stream);
        return a.causal ? launch_one<128, true, false>(a, stream) : launch_one<128, false, false>(a, stream);
    }
    if (a.d == 64) {
        if (state) return a.causal ? launch_one<64, true, true>(a, stream) : launch_one<64, false, true>(a, stream);
        return a.causal ? launch_one<64, true, false>(a, stream) : launch_one<64, false, false>(a, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace fa2
