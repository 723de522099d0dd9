#include <vector>
// bm_hip.cpp -- storage, tensors and the per-device context behind bm_hip.h (HIP runtime only; no torch, no BLAS).
#include "bm_hip.h"

#include <mutex>
#include <set>
#include "bm_c10d.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <sstream>

BMEngineException::BMEngineException(const std::string& msg, const char* file, int line, const char* func, const std::string& info)
    : std::runtime_error(msg) {
    std::ostringstream os;
    os << "File: " << file << ":" << line << " " << func << "\n" << msg << "\n" << info << "\n";
    text_ = os.str();
}

namespace bmengine {
namespace core {

namespace {
thread_local int tl_shared_device_rank = -1;
thread_local hipStream_t tl_secondary_stream = nullptr;
std::mutex g_secondary_mu;
std::set<hipStream_t> g_secondary_streams;       // engine-owned: never destroyed through the shim
}
void set_shared_device_rank(int share_index, hipStream_t secondary) {
    tl_shared_device_rank = share_index;
    tl_secondary_stream = secondary;
    if (secondary) {
        std::lock_guard<std::mutex> lk(g_secondary_mu);
        g_secondary_streams.insert(secondary);
    }
}
void forget_secondary_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_secondary_mu);
    g_secondary_streams.erase(s);
}
hipStream_t shim_secondary_stream() { return tl_shared_device_rank >= 0 ? tl_secondary_stream : nullptr; }
bool shim_is_secondary_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_secondary_mu);
    return g_secondary_streams.count(s) != 0;
}
int shared_device_rank() { return tl_shared_device_rank; }

static const char* kTypeNames[] = {"double", "float", "half", "int8", "int16", "int32", "bfloat", "fp8_e4m3", "fp8_e5m2"};
static const size_t kTypeSizes[] = {8, 4, 2, 1, 2, 4, 2, 1, 1};

const char* get_data_type_name(DataType dtype) { return kTypeNames[(int)dtype]; }
DataType name_to_data_type(const std::string& name) {
    for (int i = 0; i < 9; ++i)
        if (name == kTypeNames[i]) return (DataType)i;
    if (name == "bfloat16" || name == "bf16") return DataType::kBFloat16;
    if (name == "float16" || name == "fp16") return DataType::kHalf;
    if (name == "int") return DataType::kInt32;
    throw std::runtime_error("unknown data type name " + name);
}
size_t get_elem_size(DataType dtype) { return kTypeSizes[(int)dtype]; }
size_t get_numel(const std::vector<size_t>& size) {
    size_t n = 1;
    for (size_t s : size) n *= s;
    return n;
}

// ---- storage -------------------------------------------------------------------------------------------------------
struct Storage {
    void* ptr = nullptr;
    size_t bytes = 0;
    int device = -1;
    Stream home_keep;                                // the owning context's stream when the block was handed out (Context::tensor), kept
    hipStream_t home = nullptr;                      // alive with the block: the binding creates a context per call and drops it
    std::function<void(void*, size_t)> release;      // back to the pool / hipFree / free / nothing (borrowed)
    ~Storage() {
        if (release) release(ptr, bytes);
    }
};

// Size-class pool over hipMalloc: blocks are rounded up to a power of two between 1 KB and 64 MB (multiples of 64 MB
// above) and recycled.  Blocks never move, so a raw pointer taken from a live Tensor stays valid (the reference's arena
// may defragment unless a GCStopper is alive, allocator.cpp:74-109).
class Pool {
public:
    explicit Pool(int device) : device_(device) {}
    ~Pool() { trim(); }
    static size_t round(size_t n) {
        size_t c = 1024;
        while (c < n && c < ((size_t)64 << 20)) c <<= 1;
        if (c >= n) return c;
        const size_t unit = (size_t)64 << 20;
        return (n + unit - 1) / unit * unit;
    }
    void* get(size_t cls) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = free_.find(cls);
        void* p = nullptr;
        if (it != free_.end() && !it->second.empty()) {
            p = it->second.back();
            it->second.pop_back();
        } else {
            hipError_t e = hipMalloc(&p, cls);
            if (e != hipSuccess) {
                trim_locked();
                e = hipMalloc(&p, cls);
            }
            if (e != hipSuccess) throw std::runtime_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
            reserved_ += cls;
        }
        used_ += cls;
        peak_ = std::max(peak_, used_);
        return p;
    }
    void put(void* p, size_t cls) {
        std::lock_guard<std::mutex> lk(mu_);
        free_[cls].push_back(p);
        used_ -= cls;
    }
    void trim() {
        std::lock_guard<std::mutex> lk(mu_);
        trim_locked();
    }
    size_t used() const { return used_; }
    size_t peak() const { return peak_; }

private:
    void trim_locked() {
        for (auto& kv : free_) {
            for (void* p : kv.second) {
                (void)hipFree(p);
                reserved_ -= kv.first;
            }
            kv.second.clear();
        }
    }
    int device_;
    std::mutex mu_;
    std::map<size_t, std::vector<void*>> free_;
    size_t used_ = 0, peak_ = 0, reserved_ = 0;
};

// ---- Tensor --------------------------------------------------------------------------------------------------------
Tensor::Tensor() = default;
Tensor::~Tensor() = default;
Tensor::Tensor(const Tensor&) = default;
Tensor::Tensor(Tensor&&) noexcept = default;
Tensor& Tensor::operator=(const Tensor&) = default;
Tensor& Tensor::operator=(Tensor&&) noexcept = default;

void Tensor::set_shape(const std::vector<size_t>& s) {
    shape_ = s;
    strides_.assign(s.size(), 1);
    for (int i = (int)s.size() - 2; i >= 0; --i) strides_[i] = strides_[i + 1] * s[i + 1];
}
size_t Tensor::numel() const { return (mem_ || param_) ? get_numel(shape_) : 0; }
int Tensor::normalize_dim(int dim) const {
    const int n = ndim();
    BM_ASSERT(dim >= -n && dim < n, "dim out of range");
    return dim < 0 ? dim + n : dim;
}
// ---- deferred launches (bm_hip.h) ------------------------------------------------------------------------------------------------
namespace {
thread_local std::vector<DeferredOp> tl_deferred;
thread_local bool tl_flushing = false;
struct Poisoned {
    const void* y;
    size_t bytes;
    std::weak_ptr<void> alive;
};
thread_local std::vector<Poisoned> tl_poisoned;
bool overlaps(const void* a, size_t an, const void* b, size_t bn) {
    return (const char*)a < (const char*)b + bn && (const char*)b < (const char*)a + an;
}
void prune_dead() {
    for (size_t i = 0; i < tl_deferred.size();)
        if (tl_deferred[i].y_alive.expired()) tl_deferred.erase(tl_deferred.begin() + (long)i);
        else ++i;
}
}  // namespace
bool boundary_fusion_enabled() {
    static const bool on = [] { const char* e = getenv("ZL_BOUNDARY_FUSE"); return !(e && e[0] == '0'); }();
    return on;
}
void defer_op(DeferredOp&& op) {
    prune_dead();
    while (tl_deferred.size() >= 6) {                 // a handful at most: the oldest goes out as an ordinary launch
        DeferredOp d = std::move(tl_deferred.front());
        tl_deferred.erase(tl_deferred.begin());
        tl_flushing = true;
        d.launch();
        tl_flushing = false;
    }
    tl_deferred.push_back(std::move(op));
}
DeferredOp* find_deferred(const void* y, int kind) {
    if (tl_deferred.empty() || !y) return nullptr;
    prune_dead();
    for (auto& d : tl_deferred)
        if (d.y == y && d.kind == kind) return &d;
    return nullptr;
}
void drop_deferred(const void* y) {
    for (size_t i = 0; i < tl_deferred.size(); ++i)
        if (tl_deferred[i].y == y) {
            tl_deferred.erase(tl_deferred.begin() + (long)i);
            return;
        }
}
void flush_deferred_touching(const void* p, size_t bytes) {
    if (tl_flushing) return;                          // the launch closure itself reads its operands
    prune_dead();
    for (size_t i = 0; i < tl_deferred.size();) {
        DeferredOp& d = tl_deferred[i];
        if (overlaps(p, bytes, d.y, d.y_bytes) || overlaps(p, bytes, d.x, d.x_bytes)) {
            DeferredOp run = std::move(d);
            tl_deferred.erase(tl_deferred.begin() + (long)i);
            tl_flushing = true;
            run.launch();
            tl_flushing = false;
            i = 0;                                    // the launch may have touched the list
        } else {
            ++i;
        }
    }
}
void flush_deferred_producing(const void* p, size_t bytes) {
    if (tl_flushing || tl_deferred.empty() || !p) return;
    prune_dead();
    for (size_t i = 0; i < tl_deferred.size();) {
        if (overlaps(p, bytes, tl_deferred[i].y, tl_deferred[i].y_bytes)) {
            DeferredOp run = std::move(tl_deferred[i]);
            tl_deferred.erase(tl_deferred.begin() + (long)i);
            tl_flushing = true;
            run.launch();
            tl_flushing = false;
            i = 0;
        } else {
            ++i;
        }
    }
}
void retire_deferred_inputs(const void* p, size_t bytes) {
    if (tl_flushing) return;
    prune_dead();
    for (size_t i = 0; i < tl_deferred.size();) {
        DeferredOp& d = tl_deferred[i];
        if (!overlaps(p, bytes, d.x, d.x_bytes) && !overlaps(p, bytes, d.y, d.y_bytes)) {
            ++i;
            continue;
        }
        DeferredOp run = std::move(d);
        tl_deferred.erase(tl_deferred.begin() + (long)i);
        if (run.consumed && !overlaps(p, bytes, run.y, run.y_bytes)) {
            tl_poisoned.push_back({run.y, run.y_bytes, run.y_alive});
        } else {
            tl_flushing = true;
            run.launch();
            tl_flushing = false;
        }
        i = 0;
    }
}
void flush_all_deferred() {
    if (tl_flushing) return;
    prune_dead();
    while (!tl_deferred.empty()) {
        DeferredOp run = std::move(tl_deferred.front());
        tl_deferred.erase(tl_deferred.begin());
        tl_flushing = true;
        run.launch();
        tl_flushing = false;
    }
}

void* Tensor::data() const {
    if (param_ && !mem_) return nullptr;             // Context::parameter(): declared, not loaded yet
    BM_ASSERT(mem_ && mem_->ptr, "Tensor is empty");
    if (!tl_deferred.empty()) flush_deferred_touching((char*)mem_->ptr + offset_, mem_->bytes - offset_);
    if (!tl_poisoned.empty()) {
        for (size_t i = 0; i < tl_poisoned.size();) {
            if (tl_poisoned[i].alive.expired()) {
                tl_poisoned.erase(tl_poisoned.begin() + (long)i);
                continue;
            }
            BM_ASSERT(!overlaps((char*)mem_->ptr + offset_, mem_->bytes - offset_, tl_poisoned[i].y, tl_poisoned[i].bytes),
                      "boundary fusion: a normalised tensor is read after its fused consumer ran AND its input was overwritten in place -- "
                      "set ZL_BOUNDARY_FUSE=0 for this call sequence");
            ++i;
        }
    }
    return (char*)mem_->ptr + offset_;
}
std::weak_ptr<void> Tensor::storage_token() const { return std::weak_ptr<void>(std::static_pointer_cast<void>(mem_)); }
void* Tensor::nullable_data() const { return mem_ && mem_->ptr ? (char*)mem_->ptr + offset_ : nullptr; }
size_t Tensor::mem_bytes() const { return mem_ ? mem_->bytes - offset_ : 0; }
bool Tensor::is_continuous() const {
    size_t expect = 1;
    for (int i = ndim() - 1; i >= 0; --i) {
        if (shape_[i] != 1 && strides_[i] != expect) return false;
        expect *= shape_[i];
    }
    return true;
}
Tensor Tensor::view_unchecked(const std::vector<size_t>& size, DataType dtype) const {
    Tensor t(*this);
    t.quant_scale = quant_scale;
    t.dtype_ = dtype;
    t.set_shape(size);
    return t;
}
Tensor Tensor::view_type(const std::vector<size_t>& size, DataType dtype) const {
    BM_ASSERT(is_continuous(), "view of a non-continuous tensor");
    BM_ASSERT_EQ(get_numel(size) * get_elem_size(dtype), nbytes(), "view: size mismatch");
    return view_unchecked(size, dtype);
}
Tensor Tensor::view(const std::vector<size_t>& size) const { return view_type(size, dtype_); }
Tensor Tensor::slice_dim0(size_t from, size_t to) const {
    BM_ASSERT(ndim() >= 1 && from <= to && to <= shape_[0], "slice_dim0 out of range");
    Tensor t(*this);
    t.offset_ = offset_ + from * strides_[0] * get_elem_size(dtype_);
    t.shape_[0] = to - from;
    return t;
}
Tensor Tensor::index_dim0(size_t i) const {
    BM_ASSERT(ndim() >= 1 && i < shape_[0], "index_dim0 out of range");
    Tensor t(*this);
    t.offset_ = offset_ + i * strides_[0] * get_elem_size(dtype_);
    t.shape_.erase(t.shape_.begin());
    t.strides_.erase(t.strides_.begin());
    return t;
}
Tensor Tensor::virtual_slice(size_t from, size_t len, int dim) const {
    const int d = normalize_dim(dim);
    BM_ASSERT(from + len <= shape_[d], "virtual_slice out of range");
    Tensor t(*this);
    t.offset_ = offset_ + from * strides_[d] * get_elem_size(dtype_);
    t.shape_[d] = len;
    return t;
}
Tensor Tensor::virtual_transpose(int dim0, int dim1) const {
    const int a = normalize_dim(dim0), b = normalize_dim(dim1);
    Tensor t(*this);
    std::swap(t.shape_[a], t.shape_[b]);
    std::swap(t.strides_[a], t.strides_[b]);
    return t;
}
// Re-shape a strided tensor without moving data: every new dimension must either split one old dimension or merge old
// dimensions that are contiguous with respect to each other (stride[i] == stride[i+1] * shape[i+1]).
Tensor Tensor::view_uncontinuous(const std::vector<size_t>& size) const {
    BM_ASSERT_EQ(get_numel(size), get_numel(shape_), "view_uncontinuous: size mismatch");
    // merge the old dimensions into maximal contiguous chunks (extent, stride of the innermost element)
    std::vector<std::pair<size_t, size_t>> chunks;
    for (int i = 0; i < ndim(); ++i) {
        if (shape_[i] == 1) continue;
        if (!chunks.empty() && chunks.back().second == strides_[i] * shape_[i]) {
            chunks.back().first *= shape_[i];
            chunks.back().second = strides_[i];
        } else {
            chunks.emplace_back(shape_[i], strides_[i]);
        }
    }
    Tensor t(*this);
    t.shape_ = size;
    t.strides_.assign(size.size(), 1);
    size_t c = 0, left = chunks.empty() ? 1 : chunks[0].first;   // elements of chunk c not yet covered by new dimensions
    for (size_t i = 0; i < size.size(); ++i) {
        if (size[i] == 1) {
            t.strides_[i] = 1;
            continue;
        }
        BM_ASSERT(c < chunks.size() && left % size[i] == 0, "view_uncontinuous: the new shape crosses a stride boundary");
        left /= size[i];
        t.strides_[i] = chunks[c].second * left;
        if (left == 1 && ++c < chunks.size()) left = chunks[c].first;
    }
    return t;
}
Tensor Tensor::to_device(int dev_id) const {
    BM_ASSERT(dev_id < 0 || device_ < 0 || dev_id == device_, "to_device: one process drives one device");
    if (device_ >= 0 || !mem_) return *this;
    int dev = dev_id;
    if (dev < 0) BM_HIPRT_ASSERT(hipGetDevice(&dev));
    void* p = nullptr;
    BM_HIPRT_ASSERT(hipMalloc(&p, nbytes()));
    BM_HIPRT_ASSERT(hipMemcpy(p, data(), nbytes(), hipMemcpyHostToDevice));
    Tensor t = from_external(shape_, dtype_, p, nbytes(), dev, true);
    t.name_ = name_;
    return t;
}
std::vector<Tensor> Tensor::chunk() const {
    std::vector<Tensor> out;
    for (size_t i = 0; i < shape_[0]; ++i) out.push_back(index_dim0(i));
    return out;
}
Tensor Tensor::squeeze() const {
    std::vector<size_t> s;
    for (size_t v : shape_)
        if (v != 1) s.push_back(v);
    return view(s);
}
void Tensor::from_buffer(const void* host, bool async, hipStream_t stream) {
    BM_ASSERT(is_continuous(), "from_buffer needs a continuous tensor");
    if (device_ < 0) {
        std::memcpy(data(), host, nbytes());
        return;
    }
    // No stream given: the copy would ride the null stream, which does NOT order against the context's non-blocking stream -- and the
    // pool hands out blocks whose last kernel may still be queued there (a copy racing that kernel is a corrupted tensor, timing
    // dependent).  The copy therefore rides the stream of the context that handed the block out; a tensor without one (external
    // memory) waits for the device first.  (Not a device-wide wait where it can be avoided: with several rank threads on one device
    // it would wait for a peer's exchange kernel that is itself waiting for THIS rank -- bm_engine.cpp.)
    const bool defaulted = !stream;
    if (defaulted) stream = mem_ ? mem_->home : nullptr;
    if (!stream) BM_HIPRT_ASSERT(hipDeviceSynchronize());
    hipError_t e = hipMemcpyAsync(data(), host, nbytes(), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess && defaulted && stream) {          // (the owning context's stream is gone: the old, device-wide route)
        (void)hipGetLastError();
        stream = nullptr;
        BM_HIPRT_ASSERT(hipDeviceSynchronize());
        e = hipMemcpyAsync(data(), host, nbytes(), hipMemcpyHostToDevice, stream);
    }
    BM_HIPRT_ASSERT(e);
    if (!async || defaulted) BM_HIPRT_ASSERT(hipStreamSynchronize(stream));
}
void Tensor::to_buffer(void* host, hipStream_t stream) const {
    BM_ASSERT(is_continuous(), "to_buffer needs a continuous tensor");
    if (device_ < 0) {
        std::memcpy(host, data(), nbytes());
        return;
    }
    const bool defaulted = !stream;
    if (defaulted) stream = mem_ ? mem_->home : nullptr;     // (as above: the producer is queued on the owning context's stream)
    if (!stream) BM_HIPRT_ASSERT(hipDeviceSynchronize());
    hipError_t e = hipMemcpyAsync(host, data(), nbytes(), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess && defaulted && stream) {
        (void)hipGetLastError();
        stream = nullptr;
        BM_HIPRT_ASSERT(hipDeviceSynchronize());
        e = hipMemcpyAsync(host, data(), nbytes(), hipMemcpyDeviceToHost, stream);
    }
    BM_HIPRT_ASSERT(e);
    BM_HIPRT_ASSERT(hipStreamSynchronize(stream));
}
Tensor Tensor::from_external(const std::vector<size_t>& shape, DataType dtype, void* ptr, size_t nbytes, int device, bool own_ptr) {
    BM_ASSERT(get_numel(shape) * get_elem_size(dtype) <= nbytes, "from_external: buffer too small");
    Tensor t;
    t.mem_ = std::make_shared<Storage>();
    t.mem_->ptr = ptr;
    t.mem_->bytes = nbytes;
    t.mem_->device = device;
    if (own_ptr) t.mem_->release = device >= 0 ? std::function<void(void*, size_t)>([](void* p, size_t) { (void)hipFree(p); })
                                               : std::function<void(void*, size_t)>([](void* p, size_t) { std::free(p); });
    t.dtype_ = dtype;
    t.device_ = device;
    t.set_shape(shape);
    return t;
}
std::string Tensor::info(int) const {
    std::ostringstream os;
    os << "Tensor(" << (name_.empty() ? "?" : name_) << ", " << get_data_type_name(dtype_) << ", [";
    for (size_t i = 0; i < shape_.size(); ++i) os << (i ? "," : "") << shape_[i];
    os << "], device " << device_ << ")";
    return os.str();
}

static float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t u;
    if (e == 0) {
        if (m == 0) {
            u = sign;
        } else {   // subnormal: m * 2^-24
            float f = (float)m * 5.9604644775390625e-8f;
            std::memcpy(&u, &f, 4);
            u |= sign;
        }
    } else if (e == 31) {
        u = sign | 0x7f800000u | (m << 13);
    } else {
        u = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
std::ostream& operator<<(std::ostream& os, const Tensor& t) {
    os << t.info();
    if (t.numel() == 0 || !t.mem_ || !t.is_continuous()) return os;
    // the first elements, read back synchronously (a debugging aid, as in the reference)
    const size_t n = std::min<size_t>(t.numel(), 16), es = get_elem_size(t.dtype_);
    std::vector<char> host(n * es);
    if (t.device_ >= 0) {
        if (hipDeviceSynchronize() != hipSuccess) return os;      // (the producer may still be queued on a non-blocking stream)
        if (hipMemcpy(host.data(), t.data(), n * es, hipMemcpyDeviceToHost) != hipSuccess) return os;
    } else {
        std::memcpy(host.data(), t.data(), n * es);
    }
    os << " [";
    for (size_t i = 0; i < n; ++i) {
        const char* p = host.data() + i * es;
        os << (i ? ", " : "");
        switch (t.dtype_) {
        case DataType::kDouble: os << *(const double*)p; break;
        case DataType::kFloat: os << *(const float*)p; break;
        case DataType::kHalf: os << half_bits_to_float(*(const uint16_t*)p); break;
        case DataType::kBFloat16: { uint32_t u = (uint32_t)*(const uint16_t*)p << 16; float f; std::memcpy(&f, &u, 4); os << f; } break;
        case DataType::kInt8: os << (int)*(const int8_t*)p; break;
        case DataType::kInt16: os << *(const int16_t*)p; break;
        case DataType::kInt32: os << *(const int32_t*)p; break;
        default: os << (unsigned)*(const uint8_t*)p; break;
        }
    }
    return os << (n < t.numel() ? ", ...]" : "]");
}

// ---- Context -------------------------------------------------------------------------------------------------------
class ContextImpl {
public:
    int device, rank, world;
    hipDeviceProp_t prop;
    std::shared_ptr<Pool> pool;                      // shared with the tensors it handed out
    Stream stream;
    Context::ReduceHook reduce;
    MemoryAllocator cache_arena;
    std::shared_ptr<Pool> side_pool;                 // see Context::use_cache_alloc
    bool use_side_pool = false;
    long next_id = 0;
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    std::vector<void*> retired;                      // outgrown scratch blocks, alive as long as the context
    ~ContextImpl() {
        if (scratch) (void)hipFree(scratch);
        for (void* p : retired) (void)hipFree(p);
    }
};
const std::string Context::EMPTY_STR;

Context::Context(int device, int rank, int world_size) : pimpl(new ContextImpl) {
    pimpl->device = device;
    pimpl->rank = rank;
    pimpl->world = world_size;
    BM_HIPRT_ASSERT(hipSetDevice(device));
    BM_HIPRT_ASSERT(hipGetDeviceProperties(&pimpl->prop, device));
    pimpl->pool = std::make_shared<Pool>(device);
    pimpl->stream = get_stream();
}
Context::~Context() = default;
Context::Context(Context&&) noexcept = default;
Tensor Context::parameter(const std::vector<size_t>& size, DataType dtype) const {
    for (size_t v : size) BM_ASSERT(v > 0, "parameter: zero-sized dimension");
    Tensor t;
    t.dtype_ = dtype;
    t.device_ = pimpl->device;
    t.param_ = true;
    t.set_shape(size);
    return t;
}
Tensor Context::tensor_s(const std::vector<long>& size, DataType dtype) const {
    return tensor(std::vector<size_t>(size.begin(), size.end()), dtype);
}
void Context::print_memory_summary() const {
    std::cerr << "device " << pimpl->device << ": used " << (pimpl->pool->used() >> 20) << " MB, peak " << (pimpl->pool->peak() >> 20)
              << " MB" << std::endl;
}
int Context::active_device() const { return pimpl->device; }
int Context::rank() const { return pimpl->rank; }
int Context::world_size() const { return pimpl->world; }
int Context::get_compute_capability() const { return 90; }
int Context::get_mp_count() const { return pimpl->prop.multiProcessorCount; }
int Context::get_max_shared_memory() const { return (int)pimpl->prop.sharedMemPerBlock; }
int Context::get_L2_cache_size() const { return pimpl->prop.l2CacheSize; }
Stream Context::current_stream() const { return pimpl->stream; }
void Context::set_current_stream(Stream s) { pimpl->stream = std::move(s); }
hipStream_t Context::current_cuda_stream() const { return pimpl->stream->ptr; }
Stream Context::get_stream() const {
    hipStream_t s;
    BM_HIPRT_ASSERT(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return std::make_shared<Stream_>(s, [](hipStream_t p) { (void)hipStreamDestroy(p); });
}
Tensor Context::tensor(const std::vector<size_t>& size, DataType dtype, const std::string& name, size_t round_up_bytes) const {
    const size_t nbytes = get_numel(size) * get_elem_size(dtype);
    if (nbytes == 0) return Tensor();
    const size_t want = (nbytes + round_up_bytes - 1) / round_up_bytes * round_up_bytes;
    const size_t cls = Pool::round(want);
    // (while use_cache_alloc(true) is in force -- the reduce-stream phases of dual_stream_encode -- blocks come from, and go back to,
    //  a second pool: nothing freed under one stream is recycled under the other)
    std::shared_ptr<Pool> pool = pimpl->use_side_pool ? pimpl->side_pool : pimpl->pool;
    Tensor t;
    t.mem_ = std::make_shared<Storage>();
    t.mem_->ptr = pool->get(cls);
    t.mem_->bytes = cls;
    t.mem_->device = pimpl->device;
    t.mem_->home_keep = pimpl->stream;
    t.mem_->home = pimpl->stream ? pimpl->stream->ptr : nullptr;
    t.mem_->release = [pool, cls](void* p, size_t) { pool->put(p, cls); };
    t.dtype_ = dtype;
    t.device_ = pimpl->device;
    t.set_shape(size);
    t.id_ = pimpl->next_id++;
    t.name_ = name;
    return t;
}
Tensor Context::cuda(const Tensor& cpu_tensor) const {
    if (cpu_tensor.device() >= 0 || cpu_tensor.empty()) return cpu_tensor;
    Tensor t = tensor(cpu_tensor.shape(), cpu_tensor.dtype(), cpu_tensor.name());
    t.from_buffer(cpu_tensor.data(), false, current_cuda_stream());
    return t;
}
const Tensor Context::copy(const Tensor& src) const {
    if (src.empty()) return Tensor();
    BM_ASSERT(src.is_continuous(), "copy of a non-continuous tensor");
    Tensor t = tensor(src.shape(), src.dtype(), src.name());
    BM_HIPRT_ASSERT(hipMemcpyAsync(t.data(), src.data(), src.nbytes(), src.device() >= 0 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                   current_cuda_stream()));
    return t;
}
void* Context::scratch(size_t bytes) const {
    if (pimpl->scratch_bytes < bytes) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(pimpl->stream->ptr, &st);
        if (st != hipStreamCaptureStatusNone) return pimpl->scratch;     // keep what exists: the launchers then do not split
        const size_t want = std::max(bytes, (size_t)64 << 20);
        void* p = nullptr;
        BM_HIPRT_ASSERT(hipMalloc(&p, want));
        BM_HIPRT_ASSERT(hipMemset(p, 0, want));
        // an outgrown block is retired, not freed: a hipGraph captured earlier has its address in the K-split launches
        if (pimpl->scratch) pimpl->retired.push_back(pimpl->scratch);
        pimpl->scratch = p;
        pimpl->scratch_bytes = want;
    }
    return pimpl->scratch;
}
size_t Context::scratch_bytes() const { return pimpl->scratch_bytes; }
size_t Context::used_memory() const { return pimpl->pool->used(); }
size_t Context::peak_memory() const { return pimpl->pool->peak(); }
void Context::mem_gc() { pimpl->pool->trim(); }
void Context::recordEvent(const std::string&, int, float) const {}
void Context::set_reduce_hook(ReduceHook h) { pimpl->reduce = std::move(h); }
Tensor Context::reduce_sum(Tensor& data, DataType out_type) const {
    BM_ASSERT(out_type == data.dtype(), "reduce_sum: the output type is the input type on this path");
    if (pimpl->world > 1) {
        // the communicator owner's hook (in place), else the rank's c10d operations (bm_engine.cpp installs both)
        if (pimpl->reduce) pimpl->reduce(data, current_cuda_stream());
        else c10d::NCCLAllReduce(*this, data, data, ncclSum);
    }
    return data;
}

// context.cpp:862-876: a new tensor with dimension 0 divided / multiplied by the world size, filled by the rank's collectives
Tensor Context::reduce_scatter(const Tensor& data) const {
    if (pimpl->world == 1) return data;
    std::vector<size_t> shape = data.shape();
    BM_ASSERT(!shape.empty() && shape[0] % (size_t)pimpl->world == 0, "reduce_scatter: dimension 0 must divide by the world size");
    shape[0] /= (size_t)pimpl->world;
    Tensor out = tensor(shape, data.dtype());
    c10d::NCCLReduceScatter(*this, data, out, ncclSum);
    return out;
}
Tensor Context::all_gather(const Tensor& data) const {
    if (pimpl->world == 1) return data;
    std::vector<size_t> shape = data.shape();
    BM_ASSERT(!shape.empty(), "all_gather: a tensor with at least one dimension");
    shape[0] *= (size_t)pimpl->world;
    Tensor out = tensor(shape, data.dtype());
    c10d::NCCLAllGather(*this, data, out);
    return out;
}
void Context::reserve_cache_alloc(size_t) {
    if (!pimpl->side_pool) pimpl->side_pool = std::make_shared<Pool>(pimpl->device);     // grows on demand: nothing to reserve
}
void Context::use_cache_alloc(bool b) {
    if (b && !pimpl->side_pool) pimpl->side_pool = std::make_shared<Pool>(pimpl->device);
    pimpl->use_side_pool = b;
}
void Context::free_cache_alloc() {
    pimpl->use_side_pool = false;
    if (pimpl->side_pool) pimpl->side_pool->trim();                                      // blocks still held by tensors return later
}
MemoryAllocator* Context::get_cache_allocator() const { return &pimpl->cache_arena; }
MemoryAllocator* Context::get_allocator() const { return &pimpl->cache_arena; }
size_t MemoryAllocator::get_memory_limit() const {
    size_t free_b = 0, total_b = 0;
    BM_HIPRT_ASSERT(hipMemGetInfo(&free_b, &total_b));
    return total_b;
}
size_t MemoryAllocator::used_memory() const {
    size_t free_b = 0, total_b = 0;
    BM_HIPRT_ASSERT(hipMemGetInfo(&free_b, &total_b));
    return total_b - free_b;
}
void Context::set_cache_arena(void* base) { pimpl->cache_arena.set_base_ptr(base); }

}  // namespace core
}  // namespace bmengine


// the reference's cudaStreamCreateWithPriority / cudaStreamDestroy (refshim/cuda_runtime.h): see bm_hip.h set_shared_device_rank
extern "C" hipError_t zl_shim_stream_create_with_priority(hipStream_t* stream, unsigned int flags, int priority) {
    if (hipStream_t s = bmengine::core::shim_secondary_stream()) {
        *stream = s;
        return hipSuccess;
    }
    return hipStreamCreateWithPriority(stream, flags, priority);
}
extern "C" hipError_t zl_shim_stream_destroy(hipStream_t stream) {
    if (bmengine::core::shim_is_secondary_stream(stream)) return hipStreamSynchronize(stream);
    return hipStreamDestroy(stream);
}
// the reference's cudaEventRecord (refshim/cuda_runtime.h): what this thread still holds back goes out first
extern "C" hipError_t zl_shim_event_record(hipEvent_t event, hipStream_t stream) {
    bmengine::core::flush_all_deferred();
    return hipEventRecord(event, stream);
}
