/*
 * fa2_ring_mi355x.h -- C ABI of the sequence-sharded ring forward (libfa2_ring_mi355x.so).
 *
 * Replaces, for the reference's 03_flash_attention_v2_ring:
 *   ring_attention_forward                     common/ring_attention_kernel.cu:143-239
 *   ring_exchange / _multi / ring_exchange_kv  util/nccl_utils.h:115-142
 *   init_nccl_comm / init_mpi_nccl / cleanup   util/nccl_utils.h:29-103
 *
 * One process (or one host thread) per GPU.  The sequence is split into P equal contiguous
 * shards (04_ring_attention.cu:55-84): Q, O, L and the running softmax state stay put; K/V
 * shards travel by RCCL ncclSend/ncclRecv grouped per step (as nccl_utils.h:123-131) over xGMI
 * on a dedicated communication stream, double-buffered against the local FA2 step kernel
 * (fa2_forward_step) and fenced with HIP events -- there is no device-wide synchronisation
 * per step (the reference has one, ring_attention_kernel.cu:220).
 *
 * Differences from the reference, all deliberate:
 *   - the bootstrap needs no MPI: rank 0 calls fa2_ring_get_unique_id() and ships the 128
 *     bytes to the other ranks by whatever the launcher offers (torch.distributed broadcast in
 *     bench.py; ncclCommInitAll + fa2_ring_ctx_create_from_comm in the single-process C++ CLI);
 *   - communicator, streams, events live in a context created once, not per call
 *     (the reference cudaMallocs, creates and leaks streams inside every call, :158-163, :192-194);
 *   - the caller's K_local / V_local are PRESERVED (the reference receives into them, :225-226);
 *   - scratch (receive buffers, fp32 accumulator, running max) is caller-provided workspace;
 *   - two schedules: FA2_RING_RELAY is the reference's neighbour relay (send the resident shard to
 *     rank+1, receive from rank-1, P-1 times); FA2_RING_MESH keeps the same step order but every
 *     rank fetches each shard DIRECTLY from its owner in one grouped exchange, which on the
 *     MI355X full xGMI mesh spreads the traffic over all 7 links instead of relaying over one.
 *   - errors are status codes (fa2_mi355x.h), RCCL failures are FA2_ERR_RCCL_BASE - ncclResult_t.
 */
#ifndef FA2_RING_MI355X_H
#define FA2_RING_MI355X_H

#include <stddef.h>
#include "fa2_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FA2_RING_UNIQUE_ID_BYTES 128

#define FA2_RING_RELAY 0   /* neighbour relay, the reference's schedule */
#define FA2_RING_MESH  1   /* owner-direct fetch over the full xGMI mesh */

typedef struct fa2_ring_ctx fa2_ring_ctx;

/* Rank 0: fills 128 bytes to be handed to every rank's fa2_ring_ctx_create
 * (ncclGetUniqueId; replaces the MPI_Bcast of nccl_utils.h:33-49). */
int fa2_ring_get_unique_id(void* id_out);

/* Collective over the nranks processes/threads: joins the RCCL communicator on the CURRENT
 * device and creates the communication stream and events. */
int fa2_ring_ctx_create(fa2_ring_ctx** out, const void* unique_id, int rank, int nranks);

/* Wraps an already-initialised communicator (an ncclComm_t passed as void*), e.g. one of
 * ncclCommInitAll's in a single-process multi-GPU program.  The context does not own it. */
int fa2_ring_ctx_create_from_comm(fa2_ring_ctx** out, void* nccl_comm, int rank, int nranks);

int fa2_ring_ctx_destroy(fa2_ring_ctx* ctx);

/* How many CUs the ring backward's single-kernel block launches leave free for the exchanges that run beside them on the
 * communication stream (fa2_mi355x.h: FA2_PHASE_LEAVE_CUS; only with more than one rank).  1 <= n <= 255; the default is 16
 * (6 % of the chip) -- a design choice no multi-GPU run has tuned yet, hence a run-time setting.
 * cuda_flashattention_amd/ring.py sets it from the environment variable FA2_RING_RESERVED_CUS when that is present. */
int fa2_ring_ctx_set_reserved_cus(fa2_ring_ctx* ctx, int n);

/* ---- backend injection ------------------------------------------------------------------------
 * Everything the ring schedules do to a device goes through this table: ordering (streams and
 * events), the grouped point-to-point transport and the per-step compute.  The product table
 * (fa2_ring_default_backend) is HIP streams/events + RCCL ncclSend/ncclRecv + the kernels of
 * libfa2_mi355x.so, and that is what fa2_ring_ctx_create / _from_comm install.  A caller may install
 * another table -- an MPI or in-process transport, or, in tests/, (a) a loopback transport that runs
 * P ranks as P host threads on ONE GPU (hipMemcpyAsync between the ranks' buffers) and (b) a CPU
 * simulator that executes the enqueued operations in adversarial legal orders with the oracle's ring
 * step: both run the SAME relay / mesh / causal / backward schedule code as an 8-GPU RCCL job.
 * Contract: every callback returns FA2_OK or a negative status, is asynchronous with respect to the
 * device (nothing in a schedule blocks the host), and `stream` / `event` are whatever stream_create /
 * event_create produced (the caller's compute stream is passed through untouched).  send/recv are
 * matched per (source, destination) pair in issue order and complete on `stream`, grouped between
 * group_start and group_end like ncclGroupStart/ncclGroupEnd (util/nccl_utils.h:117-131). */
typedef struct fa2_ring_backend {
    void* user;   /* first argument of every callback */
    int (*stream_create)(void* user, void** stream_out);
    int (*stream_destroy)(void* user, void* stream);
    int (*event_create)(void* user, void** event_out);
    int (*event_destroy)(void* user, void* event);
    int (*event_record)(void* user, void* event, void* stream);
    int (*stream_wait_event)(void* user, void* stream, void* event);
    int (*group_start)(void* user);
    int (*send)(void* user, const void* buf, size_t bytes, int peer, void* stream);
    int (*recv)(void* user, void* buf, size_t bytes, int peer, void* stream);
    int (*group_end)(void* user);
    /* fa2_forward_step_strided (bf16) / fa2_forward_step (fp32: strides and causal are 0) */
    int (*forward_step)(void* user, const void* Q, const void* K, const void* V, void* O, float* L, float* Oacc, float* M,
                        int B, int H, int q_len, int kv_len, int head_dim, float softmax_scale, int dtype,
                        int first, int last, int q_head_stride, int kv_head_stride, int causal, int causal_shift,
                        void* stream);
    /* fa2_forward_state_finalize */
    int (*state_finalize)(void* user, void* O, float* L, const float* Oacc, const float* M, size_t rows, int head_dim,
                          int dtype, void* stream);
    /* fa2_backward_block */
    int (*backward_block)(void* user, const void* Q, const void* K, const void* V, const void* O, const float* L,
                          const void* dO, void* dQ, void* dK, void* dV, int B, int H, int q_len, int kv_len, int head_dim,
                          float softmax_scale, int dtype, int q_head_stride, int kv_head_stride, int q_row0, int causal,
                          int causal_shift, void* workspace, size_t workspace_bytes, void* stream, int phases);
    /* fa2_accumulate_bf16_2d, fa2_convert_f32_to_bf16 */
    int (*accumulate_bf16_2d)(void* user, float* acc, const void* src, size_t rows, size_t cols, size_t pitch, int init,
                              void* stream);
    int (*convert_f32_to_bf16)(void* user, const float* src, void* dst, size_t n, void* stream);
} fa2_ring_backend;

/* Fills *out with the product table.  Its transport entries need a context with an RCCL communicator
 * as `user` (fa2_ring_ctx_create sets that); the ordering and compute entries ignore `user`. */
int fa2_ring_default_backend(fa2_ring_backend* out);

/* A context on the CURRENT device that runs the schedules over *backend (copied).  No RCCL
 * communicator is created; rank / nranks are the caller's. */
int fa2_ring_ctx_create_with_backend(fa2_ring_ctx** out, const fa2_ring_backend* backend, int rank, int nranks);

/* Bytes of workspace fa2_ring_attention_forward needs per rank. */
size_t fa2_ring_workspace_bytes(int B, int H, int local_seq_len, int head_dim, int dtype,
                                int nranks, int schedule);

/* O_local, L_local = rows [rank*local, (rank+1)*local) of softmax(scale Q K^T) V over the whole
 * sequence of total_seq_len = nranks * local_seq_len.  Tensors [B][H][local_seq_len][d];
 * dtype FA2_DTYPE_BF16 (d in {64,128}) or FA2_DTYPE_F32 (d <= 128).  Non-causal, like the
 * reference.  Compute runs on `stream`, exchanges on the context's own stream. */
int fa2_ring_attention_forward(fa2_ring_ctx* ctx,
                               const void* Q_local, const void* K_local, const void* V_local,
                               void* O_local, float* L_local,
                               int B, int H, int total_seq_len, int local_seq_len, int head_dim,
                               float softmax_scale, int dtype, int schedule,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Causal ring forward, load-balanced ("zig-zag") sharding -- past the reference, whose ring is non-causal
 * (SURVEY 8f rank 2).  The sequence is cut into 2 * nranks chunks of local_seq_len / 2 rows; rank r holds
 * chunk r in its local rows [0, local/2) and chunk 2 * nranks - 1 - r in rows [local/2, local): every step
 * after the first then costs half a dense block on every rank.  bf16 only; local_seq_len must be even.  Same
 * workspace, schedules and stream rules as fa2_ring_attention_forward. */
int fa2_ring_attention_forward_causal(fa2_ring_ctx* ctx,
                                      const void* Q_local, const void* K_local, const void* V_local,
                                      void* O_local, float* L_local,
                                      int B, int H, int total_seq_len, int local_seq_len, int head_dim,
                                      float softmax_scale, int dtype, int schedule,
                                      void* workspace, size_t workspace_bytes, void* stream);

/* Ring backward -- past the reference, whose ring is forward-only.  All tensors
 * [B][H][local_seq_len][d] bf16 (L fp32); O_local / L_local are the ring forward's outputs, L being the
 * log-sum-exp over the WHOLE sequence.  Each rank runs the ordinary backward kernels on its rows against
 * every shard of keys: dQ adds up locally, the dK/dV pieces are sent to the shard's owner, all sums in fp32.
 * The gradient pieces of a step travel while the next step's kernels run (two sets of buffers, event-fenced).
 * bf16 only.  Workspace: fa2_ring_backward_workspace_bytes.  fa2_ring_attention_backward_causal is the
 * backward of fa2_ring_attention_forward_causal (same zig-zag layout of the local rows). */
size_t fa2_ring_backward_workspace_bytes(int B, int H, int local_seq_len, int head_dim, int dtype, int nranks);
/* Where, inside that workspace, the block kernels' own scratch lives (fa2_backward_workspace_bytes(B, H, local_seq_len, ...)
 * bytes at *offset): after a ring backward has finished, fa2_backward_status(workspace + *offset, *bytes, B, H, local_seq_len,
 * head_dim, dtype, stream) says whether a hand-off of the single-kernel blocks timed out on this rank (dQ then NaN). */
int fa2_ring_backward_block_workspace(int B, int H, int local_seq_len, int head_dim, int dtype, int nranks,
                                      size_t* offset, size_t* bytes);
int fa2_ring_attention_backward(fa2_ring_ctx* ctx,
                                const void* Q_local, const void* K_local, const void* V_local,
                                const void* O_local, const float* L_local, const void* dO_local,
                                void* dQ_local, void* dK_local, void* dV_local,
                                int B, int H, int total_seq_len, int local_seq_len, int head_dim,
                                float softmax_scale, int dtype,
                                void* workspace, size_t workspace_bytes, void* stream);
int fa2_ring_attention_backward_causal(fa2_ring_ctx* ctx,
                                       const void* Q_local, const void* K_local, const void* V_local,
                                       const void* O_local, const float* L_local, const void* dO_local,
                                       void* dQ_local, void* dK_local, void* dV_local,
                                       int B, int H, int total_seq_len, int local_seq_len, int head_dim,
                                       float softmax_scale, int dtype,
                                       void* workspace, size_t workspace_bytes, void* stream);

/* Reference-signature drop-in (ring_attention_kernel.cu:143-156): single head, fp32, `comm` is the
 * caller's ncclComm_t.  Allocates its scratch per call as the reference does, synchronises the
 * device before returning, and -- unlike the reference -- leaves K_local / V_local intact. */
int ring_attention_forward(const float* Q_local, float* K_local, float* V_local,
                           float* O_local, float* L_local,
                           int total_seq_len, int local_seq_len, int head_dim, float softmax_scale,
                           void* comm, int rank, int nranks);

/* The bare exchange primitive (ring_exchange_kv, nccl_utils.h:133-142): one grouped
 * send-to-next / receive-from-previous of a K and a V buffer of `bytes` bytes each on `stream`. */
int fa2_ring_exchange_kv(fa2_ring_ctx* ctx, const void* send_k, void* recv_k,
                         const void* send_v, void* recv_v, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FA2_RING_MI355X_H */
