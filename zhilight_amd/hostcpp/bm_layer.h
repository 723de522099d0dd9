// bm_layer.h -- the rest of the bmengine::core surface the reference's layer translation units (src/nn/linear/linear.cpp,
// src/nn/feedforward/feedforward.cpp, ...) include besides Tensor / Context: the Layer base class with its parameter
// registry and state_dict loading, the BM_PIMPL / BM_LAYER_DEF macros every layer header is written with, the device
// guards, the small integer helpers, a task thread pool and an opaque Engine.  What it mirrors (names and signatures only):
//   core::Layer            3rd/bmengine/bmengine/include/bmengine/core/layer.h:17-66, behaviour of core/layer.cpp
//   BM_PIMPL, BM_LAYER_DEF, BM_LAYER_DEF_PUBLIC, round_up, ceil_div, vector_equal   .../core/utils.h:7-78
//   WithDevice, ScopeDevice, WithDebug                                               .../core/guard.h:10-49
//   TaskThreadPool                                                                   .../core/thread_pool.h:12-32
//   Engine                                                                           .../core/engine.h:17-49 (opaque here)
// One process drives one MI355X, so the guards are no-ops that assert the device is the context's own and an Engine is
// never constructed: Context(device, rank, world_size) is the entry point (bm_hip.h).
#pragma once
#include <pthread.h>

#include <condition_variable>
#include <exception>
#include <future>
#include <iostream>
#include <mutex>
#include <queue>
#include <thread>
#include <type_traits>
#include <utility>

#include "bm_hip.h"

#define BM_PIMPL                                                                                        \
private:                                                                                                \
    class impl;                                                                                         \
    std::unique_ptr<impl> pimpl;

#define BM_LAYER_DEF_PUBLIC(name)                                                                       \
public:                                                                                                 \
    name(const name&) = delete;                                                                         \
    name(name&&) = delete;                                                                              \
    template <typename... Params> inline auto operator()(Params&&... params) {                          \
        pthread_testcancel();                                                                           \
        return forward(std::forward<Params>(params)...);                                                \
    }                                                                                                   \
    const char* layer_type() const override { return #name; }

#define BM_LAYER_DEF(name)                                                                              \
    BM_PIMPL                                                                                            \
public:                                                                                                 \
    ~name();                                                                                            \
    BM_LAYER_DEF_PUBLIC(name)

#define BM_KERNEL(name) BMEngine_KERNEL_##name
#define MAX_NUM_THREADS 1024

template <typename T, typename Tb> inline T round_up(T m, Tb d) { return ((m + T(d) - 1) / T(d)) * T(d); }
template <typename T, typename Tb> inline T ceil_div(T m, Tb d) { return (m + T(d) - 1) / T(d); }
template <typename T> inline T round_up_thread(T m) {
    T x = m > T(MAX_NUM_THREADS) ? T(MAX_NUM_THREADS) : m;
    return ((x + 31) / 32) * 32;
}
template <typename T> inline bool vector_equal(const std::vector<T>& a, const std::vector<T>& b) { return a == b; }
template <typename Ta, typename Tb> inline bool vector_equal_2(const std::vector<Ta>& a, const std::vector<Tb>& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
        if (!(a[i] == b[i])) return false;
    return true;
}

typedef struct zl_rand_generator_st* curandGenerator_t;   // Layer::init_parameters' generator: an opaque handle here

namespace bmengine {
namespace core {

class Layer {
public:
    std::map<std::string, Layer*> modules;
    std::map<std::string, Tensor*> parameters;
    std::vector<std::string> module_names;       // registration order
    std::vector<std::string> param_names;

    std::string prefix;
    std::string output_name;
    std::string name;
    int dev { 0 };
    int output_dev { 0 };

    Layer() = default;
    virtual ~Layer() = default;
    Layer(const Layer&) = delete;
    Layer(Layer&&) = delete;

    void add_submodule(const std::string& name, Layer& module) { add_submodule(name, &module); }
    void add_submodule(const std::string& name, Layer* module);
    void add_parameter(const std::string& name, Tensor& t);
    std::map<const std::string, Tensor*> named_parameters(const std::string& prefix, bool recursive = true);
    virtual const char* layer_type() const = 0;
    friend std::ostream& operator<<(std::ostream& os, const Layer& layer);
    virtual void init_parameters(const Context& ctx, curandGenerator_t& gen, const std::string& prefix = "");

    // this layer's own parameters (prefix + "." + name, registration order), then the children recursively
    virtual void load_state_dict(const Context& ctx, const std::map<std::string, const Tensor>& state_dict,
                                 const std::string& prefix, bool allow_missing = false);
    static void load_param_from_state_dict(const Context& ctx, const std::map<std::string, const Tensor>& state_dict,
                                           const std::string& name, Tensor* param, bool allow_missing = false);
    static void load_param_cast(const Context& ctx, const std::map<std::string, const Tensor>& state_dict,
                                const std::string& name, Tensor* param, DataType cast_src_dtype);
};

class WithDevice {
public:
    WithDevice(const Context& ctx, int dev);
    ~WithDevice() {}
    WithDevice(const WithDevice&) = delete;
    WithDevice& operator=(const WithDevice&) = delete;
    WithDevice(WithDevice&&) {}
    WithDevice& operator=(WithDevice&&) = delete;
};
class ScopeDevice {
public:
    ScopeDevice(const Context& ctx, int dev);
    ~ScopeDevice() {}
    ScopeDevice(const ScopeDevice&) = delete;
    ScopeDevice& operator=(const ScopeDevice&) = delete;
    ScopeDevice(ScopeDevice&&) {}
    ScopeDevice& operator=(ScopeDevice&&) = delete;
};
class WithDebug {
public:
    WithDebug(const Context&, int) {}
    ~WithDebug() {}
    WithDebug(const WithDebug&) = delete;
    WithDebug& operator=(const WithDebug&) = delete;
    WithDebug(WithDebug&&) {}
    WithDebug& operator=(WithDebug&&) = delete;
};

class TaskThreadPool {
public:
    explicit TaskThreadPool(size_t num_threads = 1, int cpu_offset = 0);
    ~TaskThreadPool();
    void run(std::function<void()> fn);
    void runSync(std::function<void()> fn);
    void wait();

protected:
    std::queue<std::function<void()>> tasks_;
    volatile long task_count_ = 0;
    std::vector<std::thread> threads_;
    std::mutex mutex_;
    std::condition_variable task_notifier_;
    std::condition_variable stop_notifier_;
    std::exception_ptr e_ptr;
    bool finished_ = false;
    void execution_loop();
};

class Engine;            // the multi-rank owner of one node: bm_engine.h
class MemoryAllocator;

// Create and record an event pair around a scope (context.h:176-187); recordEvent is a no-op unless tracing is on
struct EventScope {
    const Context& ctx;
    int debug_level;
    std::string end_name;
    EventScope(const Context& ctx, const std::string& name, int debug_level = 2, float flops = 0)
        : ctx(ctx), debug_level(debug_level), end_name("End>" + name) {
        ctx.recordEvent("Start>" + name, debug_level, flops);
    }
    ~EventScope() { ctx.recordEvent(end_name, debug_level); }
};

}  // namespace core
}  // namespace bmengine
