// bm_engine.h -- bmengine::core::Engine (3rd/bmengine/bmengine/include/bmengine/core/engine.h:17-50, engine_config.h:8-29) for
// MI355X: the owner of the tensor-parallel ranks of ONE node inside ONE process.  Same names, argument orders and meanings as
// the reference for what ModelContext::create (src/model/model_context.cpp:89-122) and the model loaders touch:
//   Engine(devices, dist)        one rank per DeviceConfiguration entry (engine.cpp:140-157: the reference's EngineImpl starts a
//                                TaskThreadPool thread per device and calls ncclCommInitRank in it, :56-59);
//   create_context_rank(rank)    the rank's Context: its device, its rank / world size, and -- new here, because a Context of
//                                this shim owns no engine pointer -- the rank's collectives installed behind c10d::NCCL*
//                                (bm_c10d.h) so that ModelContext::reduce_sum / reduce_sum2 / reduce_tp_int8, RawEmbedding's
//                                vocab-parallel gather and Context::all_gather reach a transport;
//   device_foreach(fn)           fn(rank) on every rank's thread, waited for.
// Two transports, both behind include/zhilight_amd_comm.h (libzhilight_amd_comm.so):
//   * the one-shot peer-read all-reduce (zl_ar_*) for fp16 / bf16 messages up to ZL_ENGINE_ONESHOT_BYTES (default 8 MB): inside
//     one process the peers' exchange buffers are plain pointers (peer access enabled between distinct devices);
//   * an RCCL communicator per rank (zl_comm_*: ncclCommInitRank on the rank's thread) for everything else -- created only when
//     the ranks sit on DISTINCT devices, RCCL refuses duplicates.  Ranks that share a device (the one-GPU test box) run every
//     collective on the one-shot transport: larger sums in pieces, gathers / broadcasts / reduce-scatters as sums of zero-padded
//     slices (adding zeros is exact; payloads other than fp16 / bf16 travel byte by byte as exact fp16 integers), send / recv not
//     at all.  Their contexts get streams of DIFFERENT PRIORITIES: the exchange kernels wait for each other inside a launch, and two
//     streams of one priority may share a hardware queue, where the waiting kernel would block its peer's (bm_engine.cpp).
// A Context stays bound to the thread that created it (bm_hip.h); create the rank's Context ON the rank's thread
// (device_foreach, or Engine::run).
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "bm_hip.h"
#include "bm_layer.h"

namespace bmengine {
namespace core {

struct DeviceConfiguration {
    int device_id;
    size_t memory_limit;
    DeviceConfiguration(int device_id, size_t memory_limit) : device_id(device_id), memory_limit(memory_limit) {}
};
struct DistConfiguration {
    int tp{-1};
    std::string dist_init_addr;
    int nnodes{1};
    int node_rank{0};
};
struct GPUInfo {
    int real_device_idx;
    int compute_capability;
    size_t total_memory;
    size_t free_memory;
    size_t alloc_memory;
};

class EngineImpl;
// Engine can be accessed from multiple threads.
class Engine {
    std::unique_ptr<EngineImpl> pimpl;

public:
    Engine(const std::vector<DeviceConfiguration>& dev_cfg, const DistConfiguration& dist_cfg);
    Engine(const std::vector<DeviceConfiguration>& dev_cfg) : Engine(dev_cfg, DistConfiguration()) {}
    ~Engine();
    Engine(const Engine&) = delete;
    Engine(Engine&&) = delete;

    Context create_context(const std::vector<int>& devices) const;   // devices[0] names the rank (engine.cpp:320-324)
    Context create_context() const;                                  // rank 0
    Context create_context_rank(int rank) const;
    int num_gpus() const;
    int world_size() const;
    int local_ranks() const;
    int nnodes() const { return 1; }
    int node_rank() const { return 0; }
    template <typename T> void broadcast_data(T&, int = 0) {}         // (host communicator between nodes: one node here)
    GPUInfo get_gpu_info(int device_idx) const;

    void device_foreach(std::function<void(int)> fn);                 // fn(rank) on every rank's thread; rethrows the first failure
    void run(int rank, std::function<void()> fn);                     // one rank's thread, waited for
    void print_memory_summary();
    void freeze_model_memory() {}
    MemoryAllocator* get_allocator(int dev_id);

    // what the transports are (tests and logs): true when the ranks own an RCCL communicator
    bool has_rccl() const;
    // number of bounded waits of the one-shot exchange that expired on `rank` since the start (0 = every sum is valid);
    // synchronises the rank's setup stream -- call between steps, on the rank's thread
    int exchange_errors(int rank) const;
};

}  // namespace core
}  // namespace bmengine
