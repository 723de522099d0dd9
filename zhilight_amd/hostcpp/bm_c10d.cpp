// bm_c10d.cpp -- bmengine::c10d over operations installed by the communicator's owner (bm_c10d.h).  Host code only.
#include "bm_c10d.h"

#include <functional>
#include <map>
#include <mutex>
#include <vector>

namespace bmengine {
namespace c10d {

namespace {
std::mutex g_mu;
std::map<int, Collectives> g_ops;                 // by rank: one communicator per GPU thread (engine.cpp:56-59)

// a COPY of the rank's operations, taken under the lock: a pointer into the map would race with set_collectives, and calling an
// operation with the lock held would let one rank's blocking send / group end stall every other rank's lookup
bool ops_of(const core::Context& ctx, Collectives& out) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ops.find(ctx.rank());
    if (it == g_ops.end()) return false;
    out = it->second;
    return true;
}
// one rank: the collective is a copy (or nothing when it works in place)
void local_copy(const core::Context& ctx, const core::Tensor& send, core::Tensor& recv) {
    BM_ASSERT_EQ(send.nbytes(), recv.nbytes(), "c10d: send / recv sizes differ on a single rank");
    if (send.data() != recv.data())
        BM_HIPRT_ASSERT(hipMemcpyAsync(recv.data(), send.data(), send.nbytes(), hipMemcpyDeviceToDevice, ctx.current_cuda_stream()));
}
}  // namespace

#define ZL_NEED(member, what)                                                                                              \
    Collectives cc_;                                                                                                       \
    const Collectives* c_ = ops_of(ctx, cc_) ? &cc_ : nullptr;                                                             \
    BM_ASSERT(c_ && c_->member, what ": no communicator operations installed for this device (c10d::set_collectives)");

void set_collectives(const core::Context& ctx, const Collectives& c) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_ops[ctx.rank()] = c;
}

void NCCLAllGather(const core::Context& ctx, const core::Tensor& sendbuff, core::Tensor& recvbuff) {
    if (ctx.world_size() == 1) return local_copy(ctx, sendbuff, recvbuff);
    ZL_NEED(all_gather, "NCCLAllGather")
    c_->all_gather(sendbuff, recvbuff, ctx.current_cuda_stream());
}
void NCCLAllReduce(const core::Context& ctx, const core::Tensor& sendbuff, core::Tensor& recvbuff, ncclRedOp_t op) {
    if (ctx.world_size() == 1) return local_copy(ctx, sendbuff, recvbuff);
    ZL_NEED(all_reduce, "NCCLAllReduce")
    c_->all_reduce(sendbuff, recvbuff, op, ctx.current_cuda_stream());
}
void NCCLBroadcast(const core::Context& ctx, const core::Tensor& sendbuff, core::Tensor& recvbuff, int root) {
    if (ctx.world_size() == 1) return local_copy(ctx, sendbuff, recvbuff);
    ZL_NEED(broadcast, "NCCLBroadcast")
    c_->broadcast(sendbuff, recvbuff, root, ctx.current_cuda_stream());
}
void NCCLReduce(const core::Context& ctx, const core::Tensor& sendbuff, core::Tensor& recvbuff, ncclRedOp_t op, int root) {
    if (ctx.world_size() == 1) return local_copy(ctx, sendbuff, recvbuff);
    ZL_NEED(reduce, "NCCLReduce")
    c_->reduce(sendbuff, recvbuff, op, root, ctx.current_cuda_stream());
}
void NCCLReduceScatter(const core::Context& ctx, const core::Tensor& sendbuff, core::Tensor& recvbuff, ncclRedOp_t op) {
    if (ctx.world_size() == 1) return local_copy(ctx, sendbuff, recvbuff);
    ZL_NEED(reduce_scatter, "NCCLReduceScatter")
    c_->reduce_scatter(sendbuff, recvbuff, op, ctx.current_cuda_stream());
}
void NCCLSend(const core::Context& ctx, const core::Tensor& sendbuff, int peer) {
    ZL_NEED(send, "NCCLSend")
    c_->send(sendbuff, peer, ctx.current_cuda_stream());
}
void NCCLRecv(const core::Context& ctx, core::Tensor& recvbuff, int peer) {
    ZL_NEED(recv, "NCCLRecv")
    c_->recv(recvbuff, peer, ctx.current_cuda_stream());
}
// group calls carry no context in the reference: they reach every installed communicator owner (normally one per thread)
// (the callbacks are copied out under the lock and run WITHOUT it: a group end may block on a peer's handshake, and that peer's
//  thread must still be able to look its own operations up)
void NCCLGroupStart() {
    std::vector<std::function<void()>> calls;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& kv : g_ops)
            if (kv.second.group_start) calls.push_back(kv.second.group_start);
    }
    for (auto& f : calls) f();
}
void NCCLGroupEnd() {
    std::vector<std::function<void()>> calls;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& kv : g_ops)
            if (kv.second.group_end) calls.push_back(kv.second.group_end);
    }
    for (auto& f : calls) f();
}
void NCCLGroupEndCheck(ncclComm_t) { NCCLGroupEnd(); }
int NCCLCommCount(const core::Context& ctx) {
    Collectives c;
    return ops_of(ctx, c) ? c.comm_count : ctx.world_size();
}
int NCCLCommUserRank(const core::Context& ctx) {
    Collectives c;
    return ops_of(ctx, c) ? c.user_rank : ctx.rank();
}

}  // namespace c10d
}  // namespace bmengine
