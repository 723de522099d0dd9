// bm_layer.cpp -- Layer's parameter registry and state_dict loading, Context's parameter functions, the device guards and
// the task thread pool behind bm_layer.h / bm_hip.h.  Behaviour follows the reference's 3rd/bmengine/bmengine/core/
// layer.cpp:12-137 (registration order; own parameters first, then the children under prefix + "." + name) and
// context.cpp:466-492, 664-789 (assign_or_copy, parameter, load_parameter, load_parameter_part).
#include <cstdlib>
#include <cstring>
#include <iomanip>

#include "bm_functions.h"
#include "bm_layer.h"

namespace bmengine {
namespace core {

DistLayout transpose_layout(DistLayout l) {
    return l == DistLayout::COLUMNAR ? DistLayout::ROW : l == DistLayout::ROW ? DistLayout::COLUMNAR : l;
}
const char* get_dist_layout_name(DistLayout l) {
    return l == DistLayout::COLUMNAR ? "COLUMNAR" : l == DistLayout::ROW ? "ROW" : "REPLICATED";
}

// ---- Layer ---------------------------------------------------------------------------------------------------------
void Layer::add_submodule(const std::string& name, Layer* module) {
    module->name = name;
    modules.emplace(name, module);
    module_names.push_back(name);
}
void Layer::add_parameter(const std::string& name, Tensor& t) {
    parameters.emplace(name, &t);
    param_names.push_back(name);
}
static void print_layer(std::ostream& os, const Layer* layer, int depth) {
    os << ": (" << layer->layer_type() << ")";
    for (auto& p : layer->parameters) {
        os << "\n" << std::setw((depth + 1) * 4) << "" << p.first << " [";
        const auto& sz = p.second->size();
        for (size_t i = 0; i < sz.size(); ++i) os << (i ? ", " : "") << sz[i];
        os << "] dtype=" << get_data_type_name(p.second->dtype()) << " device=" << p.second->device();
    }
    for (auto& m : layer->modules) {
        os << "\n" << std::setw((depth + 1) * 4) << "" << m.first;
        print_layer(os, m.second, depth + 1);
    }
}
std::ostream& operator<<(std::ostream& os, const Layer& layer) {
    print_layer(os, &layer, 0);
    return os;
}
void Layer::init_parameters(const Context& ctx, curandGenerator_t& gen, const std::string& prefix) {
    // random initialisation is a training / smoke-test facility of bmengine; here parameters are allocated and zeroed
    for (auto& p : parameters) ctx.init_parameter(prefix + "." + p.first, p.second);
    for (auto& m : modules) m.second->init_parameters(ctx, gen, prefix + "." + m.first);
}
std::map<const std::string, Tensor*> Layer::named_parameters(const std::string& prefix, bool recursive) {
    std::map<const std::string, Tensor*> out;
    const std::string lp = (!prefix.empty() && prefix.back() != '.') ? prefix + "." : prefix;
    for (auto& n : param_names) out.emplace(lp + n, parameters.at(n));
    if (recursive)
        for (auto& m : module_names)
            for (auto& p : modules[m]->named_parameters(m + ".", recursive)) out.emplace(lp + p.first, p.second);
    return out;
}
void Layer::load_state_dict(const Context& ctx, const std::map<std::string, const Tensor>& state_dict, const std::string& prefix,
                            bool allow_missing) {
    this->prefix = prefix;
    for (auto& n : param_names) load_param_from_state_dict(ctx, state_dict, prefix + "." + n, parameters[n], allow_missing);
    for (auto& m : module_names) modules[m]->load_state_dict(ctx, state_dict, prefix + "." + m, allow_missing);
}
void Layer::load_param_from_state_dict(const Context& ctx, const std::map<std::string, const Tensor>& state_dict,
                                       const std::string& name, Tensor* param, bool allow_missing) {
    auto it = state_dict.find(name);
    if (it == state_dict.end()) {
        BM_ASSERT(allow_missing, "param " + name + " not found in state_dict");
        return;
    }
    ctx.assign_or_copy(param, &it->second);
}
void Layer::load_param_cast(const Context& ctx, const std::map<std::string, const Tensor>& state_dict, const std::string& name,
                            Tensor* param, DataType cast_src_dtype) {
    auto it = state_dict.find(name);
    BM_ASSERT(it != state_dict.end(), "param " + name + " not found in state_dict");
    if (it->second.dtype() == cast_src_dtype) {
        Tensor tmp = ctx.parameter(param->shape(), cast_src_dtype);
        ctx.assign_or_copy(&tmp, &it->second);
        *param = functions::typecast(ctx, tmp, param->dtype());
    } else {
        ctx.assign_or_copy(param, &it->second);
    }
}

// ---- Context: parameters ---------------------------------------------------------------------------------------------
static DataType source_dtype(const Tensor& t) { return t.dtype() != DataType::kInt16 ? t.dtype() : DataType::kBFloat16; }

void Context::init_parameter(const std::string& name, Tensor* t) const {
    Tensor fresh = tensor(t->shape(), t->dtype(), name.empty() ? t->name() : name);
    BM_HIPRT_ASSERT(hipMemsetAsync(fresh.data(), 0, fresh.nbytes(), current_cuda_stream()));
    fresh.quant_scale = t->quant_scale;
    *t = fresh;
}
void Context::assign_or_copy(Tensor* dst, const Tensor* src) const {
    if (src->device() >= 0) {   // already on the device: share it
        *dst = *src;
        return;
    }
    BM_ASSERT(src->shape() == dst->shape(), "src and dst have different shape: " + src->name());
    static const int auto_cast = std::getenv("LOAD_AUTO_CAST") ? std::atoi(std::getenv("LOAD_AUTO_CAST")) : 0;
    const DataType sdt = source_dtype(*src);
    if (auto_cast > 0 && sdt != dst->dtype()) {
        Tensor buf = tensor(src->shape(), sdt);
        buf.from_buffer(src->data(), false, current_cuda_stream());
        *dst = functions::typecast(*this, buf, dst->dtype());
        return;
    }
    BM_ASSERT_EQ(sdt, dst->dtype(), "Assign to different dtype. src: " + src->name());
    if (dst->nullable_data() == nullptr) init_parameter(dst->name(), dst);
    dst->from_buffer(src->data(), false, current_cuda_stream());
}
Tensor Context::distribute_parameter(const Tensor& param, DistLayout layout) const {
    if (world_size() == 1 || layout == DistLayout::REPLICATED) return param;
    std::map<std::string, const Tensor> sd;
    sd.emplace("p", param);
    Tensor w = parameter(param.shape(), source_dtype(param));
    load_parameter_part(&w, "p", sd, layout, rank(), world_size());
    return w;
}
void Context::load_parameter(Tensor* weight, const std::string& name, const std::map<std::string, const Tensor>& state_dict,
                             bool parallel, DistLayout layout) const {
    auto it = state_dict.find(name);
    BM_ASSERT(it != state_dict.end(), "param " + name + " not found in state_dict");
    const Tensor& param = it->second;
    BM_ASSERT(weight->shape() == param.shape(), name + " shape mismatch");
    if (!parallel || world_size() == 1 || layout == DistLayout::REPLICATED) {
        assign_or_copy(weight, &param);
        weight->set_name(name);
        return;
    }
    BM_ASSERT_EQ(source_dtype(param), weight->dtype(), name + " dtype mismatch");
    load_parameter_part(weight, name, state_dict, layout, rank(), world_size());
}
void Context::load_parameter_part(Tensor* weight, const std::string& name, const std::map<std::string, const Tensor>& state_dict,
                                  DistLayout layout, size_t part, size_t total) const {
    auto it = state_dict.find(name);
    BM_ASSERT(it != state_dict.end(), "param " + name + " not found in state_dict");
    const Tensor& param = it->second;
    std::vector<size_t> shape = weight->shape();
    BM_ASSERT(!shape.empty() && part < total, "load_parameter_part: bad partition");
    const size_t shard_dim = shape.size() - (layout == DistLayout::ROW && shape.size() >= 2 ? 2 : 1);
    BM_ASSERT(shape[shard_dim] % total == 0, "size can't be divided by world_size");
    shape[shard_dim] /= total;
    const size_t shard_len = shape[shard_dim];
    const DataType dt = weight->dtype();
    *weight = tensor(shape, dt, name);
    if (shard_dim == 0) {     // contiguous rows
        Tensor rows = param.slice_dim0_len(part * shard_len, shard_len);
        if (rows.device() >= 0)
            BM_HIPRT_ASSERT(hipMemcpyAsync(weight->data(), rows.data(), weight->nbytes(), hipMemcpyDeviceToDevice, current_cuda_stream()));
        else
            weight->from_buffer(rows.data(), false, current_cuda_stream());
        return;
    }
    BM_ASSERT_EQ(weight->ndim(), 2, "Unsupported ndim");
    // columns [part * shard_len, (part + 1) * shard_len) of every row: one pitched copy, host or device source
    const size_t shard_bytes = shard_len * get_elem_size(dt), row_bytes = shard_bytes * total;
    const char* src = param.data<char>() + part * shard_bytes;
    BM_HIPRT_ASSERT(hipMemcpy2DAsync(weight->data(), shard_bytes, src, row_bytes, shard_bytes, weight->size(0),
                                     param.device() >= 0 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, current_cuda_stream()));
    BM_HIPRT_ASSERT(hipStreamSynchronize(current_cuda_stream()));
}

// ---- guards ----------------------------------------------------------------------------------------------------------
bool Context::switch_to_device(int idx) const {
    BM_ASSERT_EQ(idx, 0, "one process drives one device: the context has a single device index");
    return true;
}
WithDevice::WithDevice(const Context& ctx, int dev) { ctx.switch_to_device(dev); }
ScopeDevice::ScopeDevice(const Context& ctx, int dev) { ctx.switch_to_device(dev); }
WithDevice Context::with_device(int dev_id) const { return WithDevice(*this, dev_id); }
ScopeDevice Context::scope_device(int dev_id) const { return ScopeDevice(*this, dev_id); }
WithDebug Context::with_debug(int debug_level) const { return WithDebug(*this, debug_level); }

// ---- TaskThreadPool ----------------------------------------------------------------------------------------------------
TaskThreadPool::TaskThreadPool(size_t num_threads, int) {
    for (size_t i = 0; i < num_threads; ++i) threads_.emplace_back([this] { execution_loop(); });
}
TaskThreadPool::~TaskThreadPool() {
    {
        std::lock_guard<std::mutex> lk(mutex_);
        finished_ = true;
    }
    task_notifier_.notify_all();
    for (auto& t : threads_) t.join();
}
void TaskThreadPool::execution_loop() {
    for (;;) {
        std::function<void()> fn;
        {
            std::unique_lock<std::mutex> lk(mutex_);
            task_notifier_.wait(lk, [this] { return finished_ || !tasks_.empty(); });
            if (tasks_.empty()) return;
            fn = std::move(tasks_.front());
            tasks_.pop();
        }
        try {
            fn();
        } catch (...) {
            std::lock_guard<std::mutex> lk(mutex_);
            if (!e_ptr) e_ptr = std::current_exception();
        }
        {
            std::lock_guard<std::mutex> lk(mutex_);
            --task_count_;
        }
        stop_notifier_.notify_all();
    }
}
void TaskThreadPool::run(std::function<void()> fn) {
    {
        std::lock_guard<std::mutex> lk(mutex_);
        tasks_.push(std::move(fn));
        ++task_count_;
    }
    task_notifier_.notify_one();
}
void TaskThreadPool::wait() {
    std::unique_lock<std::mutex> lk(mutex_);
    stop_notifier_.wait(lk, [this] { return task_count_ == 0; });
    if (e_ptr) {
        std::exception_ptr e = e_ptr;
        e_ptr = nullptr;
        std::rethrow_exception(e);
    }
}
void TaskThreadPool::runSync(std::function<void()> fn) {
    run(std::move(fn));
    wait();
}

}  // namespace core
}  // namespace bmengine
