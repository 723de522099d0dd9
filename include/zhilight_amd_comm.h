/*
 * zhilight_amd_comm.h -- C ABI of the tensor-parallel exchange step of the decode hot path (SURVEY.md 8a row a20, 8e,
 * 8f rank 2).  A separate shared library (libzhilight_amd_comm.so, links librccl) so that single-GPU users of
 * libzhilight_amd.so carry no communication dependency.
 *
 * Two transports behind one "sum the (M, dim_model) partial outputs over the ranks of a node" operation:
 *   zl_comm_*   a communicator = ncclCommInitRank on the calling thread's current device (one per GPU, as the reference's
 *               engine creates them, 3rd/bmengine/bmengine/core/engine.cpp:56-59,140-157), collectives enqueued on the
 *               caller's stream like c10d::NCCL* (3rd/bmengine/bmengine/c10d/c10d.cpp:24-146).  RCCL over xGMI: the
 *               bandwidth-bound prompt-chunk messages (4-64 MB).
 *   zl_ar_*     a one-shot peer-read all-reduce for the latency-bound decode messages (8 KB .. 512 KB): every rank
 *               publishes its partial rows in a buffer its peers map (hipIpc, or plain pointers inside one process),
 *               pushes a per-chunk flag into every peer's flag table, waits for its peers' flags, reads their rows over
 *               its direct xGMI links and sums them in RANK ORDER in fp32 -- every rank gets bit-identical sums -- with the
 *               residual add of the layer fused (ModelContext::reduce_sum + element_add_scale, src/model/
 *               model_context.cpp:203-242, src/nn/block/block.cpp:123-140).  One launch, no host round trip, epoch
 *               counters live on the device: capturable in a hipGraph and replayable.  Every wait is bounded (2^22 polls, about
 *               0.6 s on MI355X: a peer later than that -- a stalled host thread, a dead rank -- makes the launch set the
 *               error word of its state, zl_ar_status, instead of hanging; the step's results are then invalid).
 * Status codes as in zhilight_amd.h (0 ok, < 0 ZL_E*, > 0 hipError_t); ncclResult_t errors are returned as 1000 + code.
 */
#ifndef ZHILIGHT_AMD_COMM_H
#define ZHILIGHT_AMD_COMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* zl_comm_stream_t;   /* hipStream_t */
typedef struct zl_comm zl_comm_t; /* opaque: an ncclComm_t + its rank / size */

#define ZL_COMM_UNIQUE_ID_BYTES 128

/* ---- RCCL communicator (c10d.cpp:24-146) -------------------------------------------------------------------------- */
int zl_comm_unique_id(void* id /* ZL_COMM_UNIQUE_ID_BYTES, filled on rank 0 and shipped to the others by the host */);
int zl_comm_create(zl_comm_t** out, int world_size, int rank, const void* id);   /* collective: every rank calls it */
int zl_comm_destroy(zl_comm_t* comm);
int zl_comm_rank(const zl_comm_t* comm);
int zl_comm_size(const zl_comm_t* comm);
/* dtype: ZL_F16 = 0, ZL_BF16 = 1, 2 = float32, 3 = int32, 4 = int8.  In place when send == recv. */
int zl_comm_all_reduce_sum(zl_comm_t* comm, const void* send, void* recv, int64_t count, int dtype, zl_comm_stream_t s);
int zl_comm_all_gather(zl_comm_t* comm, const void* send, void* recv /* world * count */, int64_t count, int dtype,
                       zl_comm_stream_t s);
int zl_comm_reduce_scatter_sum(zl_comm_t* comm, const void* send /* world * count */, void* recv, int64_t count, int dtype,
                               zl_comm_stream_t s);
int zl_comm_broadcast(zl_comm_t* comm, void* buf, int64_t count, int dtype, int root, zl_comm_stream_t s);
int zl_comm_send(zl_comm_t* comm, const void* buf, int64_t count, int dtype, int peer, zl_comm_stream_t s);
int zl_comm_recv(zl_comm_t* comm, void* buf, int64_t count, int dtype, int peer, zl_comm_stream_t s);
int zl_comm_group_start(void);
int zl_comm_group_end(void);

/* ---- one-shot peer-read all-reduce ----------------------------------------------------------------------------------
 * Setup (host, once; none of it inside the timed path):
 *   1. every rank allocates ONE device buffer of zl_ar_buffer_bytes(max_message_bytes) bytes, zeroed, from memory its peers
 *      can map (hipExtMallocWithFlags(hipDeviceMallocFinegrained) across processes; any hipMalloc inside one process);
 *   2. cross-process: zl_ar_export(buffer, handle) -> ship the 64-byte handles to every rank -> zl_ar_open(handle) gives the
 *      local mapping of a peer's buffer;
 *   3. zl_ar_init(state, ...) writes the table of the `world` buffer addresses (own buffer at index `rank`) into the
 *      rank's device-resident state block (zl_ar_state_bytes() bytes, zeroed by the caller first).
 * Per message: zl_ar_all_reduce(state, x, residual, out, n, ...):  out[i] = T(T(sum_r x_r[i]) + residual[i])  (sum in fp32 in
 * rank order, rounded to T once; residual == NULL: out = T(sum)); n * 2 bytes <= max_message_bytes; x, residual, out are
 * ordinary device pointers of the calling rank.  Every rank must issue the same sequence of calls. */
#define ZL_AR_MAX_RANKS 8
#define ZL_AR_IPC_HANDLE_BYTES 64
int64_t zl_ar_buffer_bytes(int64_t max_message_bytes);
int64_t zl_ar_state_bytes(void);
int zl_ar_alloc(int64_t bytes, void** out);     /* setup: zeroed fine-grained device memory peers can map (hipExtMallocWithFlags) */
int zl_ar_free(void* buffer);
int zl_ar_export(void* buffer, void* handle /* ZL_AR_IPC_HANDLE_BYTES */);
int zl_ar_open(const void* handle, void** mapped);
int zl_ar_close(void* mapped);
int zl_ar_init(void* state, int world_size, int rank, void* const* buffers /* host array [world_size] of device addresses */,
               int64_t max_message_bytes, zl_comm_stream_t s);
int zl_ar_all_reduce(void* state, const uint16_t* x, const uint16_t* residual, uint16_t* out, int64_t n, int dtype,
                     zl_comm_stream_t s);
/* The same message with the rows travelling as group-32 int8 codes + T scales -- ModelContext::reduce_tp_int8
 * (src/model/model_context.cpp:244-326; the reference switches to it above REDUCE_TP_INT8_THRES rows) in ONE launch: quantise,
 * peers read the codes of their own slice, sum with the unquantised own slice, re-quantise, everyone reads the re-quantised slices:
 * 2 x 1.06 (W - 1) / W bytes per value over the links instead of 2 (W - 1).  Bit for bit the reference's five-step composition
 * (quant_group_32 / dequant_sum_quant_g32 / dequant_group_32, quant_reduce_kernel.cu:13-330), then the residual add of
 * zl_ar_all_reduce.  n % (64 world_size) == 0; n * 2 bytes <= max_message_bytes; world_size = the state's. */
int zl_ar_all_reduce_int8(void* state, const uint16_t* x, const uint16_t* residual, uint16_t* out, int64_t n, int world_size, int dtype,
                          zl_comm_stream_t s);
/* error word of the state: 0 ok, else the number of bounded waits that expired (synchronises the stream it reads on) */
int zl_ar_status(void* state, zl_comm_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* ZHILIGHT_AMD_COMM_H */
