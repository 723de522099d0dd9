// bm_c10d.h -- bmengine::c10d (3rd/bmengine/bmengine/include/bmengine/c10d/c10d.h:15-42): the collective wrappers layer code
// calls by name (FeedForward::route broadcasts the routing result, feedforward.cpp:472-478; the expert-parallel paths gather and
// reduce).  Same names and argument orders.  This layer owns no communicator: with one rank every call is the local identity /
// copy; with more the owner of the RCCL communicator (include/zhilight_amd_comm.h: zl_comm_broadcast, zl_comm_all_gather, ...)
// installs the operations with set_collectives, and a call without them throws instead of returning unreduced data.
#pragma once
#include <rccl/rccl.h>

#include <functional>

#include "bm_hip.h"

namespace bmengine {
namespace c10d {

// what the communicator owner provides (count in elements of the tensor's dtype)
struct Collectives {
    std::function<void(const core::Tensor& send, core::Tensor& recv, hipStream_t)> all_gather;
    std::function<void(const core::Tensor& send, core::Tensor& recv, ncclRedOp_t, hipStream_t)> all_reduce;
    std::function<void(const core::Tensor& send, core::Tensor& recv, int root, hipStream_t)> broadcast;
    std::function<void(const core::Tensor& send, core::Tensor& recv, ncclRedOp_t, int root, hipStream_t)> reduce;
    std::function<void(const core::Tensor& send, core::Tensor& recv, ncclRedOp_t, hipStream_t)> reduce_scatter;
    std::function<void(const core::Tensor& buf, int peer, hipStream_t)> send;
    std::function<void(core::Tensor& buf, int peer, hipStream_t)> recv;
    std::function<void()> group_start, group_end;
    int comm_count = 1, user_rank = 0;
};
void set_collectives(const core::Context& ctx, const Collectives& c);    // per rank (one context and one communicator per GPU thread)

void NCCLAllGather(const core::Context& ctx, const core::Tensor& sendbuff, core::Tensor& recvbuff);
void NCCLAllReduce(const core::Context& ctx, const core::Tensor& sendbuff, core::Tensor& recvbuff, ncclRedOp_t op);
void NCCLBroadcast(const core::Context& ctx, const core::Tensor& sendbuff, core::Tensor& recvbuff, int root);
void NCCLReduce(const core::Context& ctx, const core::Tensor& sendbuff, core::Tensor& recvbuff, ncclRedOp_t op, int root);
void NCCLReduceScatter(const core::Context& ctx, const core::Tensor& sendbuff, core::Tensor& recvbuff, ncclRedOp_t op);
void NCCLSend(const core::Context& ctx, const core::Tensor& sendbuff, int peer);
void NCCLRecv(const core::Context& ctx, core::Tensor& recvbuff, int peer);
void NCCLGroupStart();
void NCCLGroupEnd();
void NCCLGroupEndCheck(ncclComm_t comm);
int NCCLCommCount(const core::Context& ctx);
int NCCLCommUserRank(const core::Context& ctx);

}  // namespace c10d
}  // namespace bmengine
