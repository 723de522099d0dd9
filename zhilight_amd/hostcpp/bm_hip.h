// bm_hip.h -- the slice of bmengine's tensor / context surface that ZhiLight's hot-path host code touches, re-built on
// the HIP runtime for MI355X.  Source-compatible with the reference for the members it declares (same namespace, class
// and member names, argument orders and meanings), so the reference's callers -- Int4GPTQ::forward (src/nn/linear/
// linear.cpp:934-1004), NormalImpl::dynamic_batch_forward (src/nn/attention/attention.cpp:846-964), FeedForward
// (src/nn/feedforward/feedforward.cpp:113-187), EncoderLayer (src/nn/block/block.cpp:86-143) -- compile against it with
// cudaStream_t spelled hipStream_t.  What it mirrors:
//   core::DataType      3rd/bmengine/bmengine/include/bmengine/core/dtype.h:12-22 (same enumerator order)
//   core::Tensor        .../core/tensor.h:30-138  (shape / stride / dtype / data<T>(), view, slice_dim0, index_dim0,
//                       virtual_slice, chunk, from_buffer / to_buffer, from_external, the public quant_scale side tensor)
//   core::Stream        .../core/stream.h:12-27
//   core::Context       .../core/context.h:24-173 (tensor(), null_tensor(), current_stream(), set_current_stream(),
//                       rank / world_size, get_mp_count / get_compute_capability / get_max_shared_memory, recordEvent,
//                       is_BSHD, high_precision, reduce_sum)
//   BMEngineException   .../core/exception.h:25-130 (+ the BM_ASSERT* / BM_CUDART_ASSERT macros)
// Nothing below is taken from those files beyond the names: storage is a ref-counted block from a per-context size-class
// pool over hipMalloc (the reference carves a movable arena, allocator.cpp:31-226; tensors here never move, so raw
// pointers stay valid while their Tensor lives), a Context is bound to one device and one thread.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

class BMEngineException : public std::runtime_error {
public:
    BMEngineException(const std::string& msg, const char* file, int line, const char* func, const std::string& info = "");
    const char* what() const noexcept override { return text_.c_str(); }

private:
    std::string text_;
};

#define BM_EXCEPTION(msg) throw BMEngineException("Exception:\n", __FILE__, __LINE__, __PRETTY_FUNCTION__, msg)
// statement macros of the `if (...) { throw }` form: the reference's callers also write them without a trailing semicolon
#define BM_ASSERT(cond, msg)                                                                            \
    if (__builtin_expect(!(cond), 0)) {                                                                 \
        throw BMEngineException("Assertion failed: " #cond, __FILE__, __LINE__, __PRETTY_FUNCTION__, msg); \
    }
#define BM_ASSERT_CMP_(x, y, op, opname, msg)                                                           \
    if (__builtin_expect(!((x)op(y)), 0)) {                                                             \
        throw BMEngineException(std::string("Assertion failed: " #x " " opname " " #y " i.e. ") + std::to_string(x) + \
                                    " " opname " " + std::to_string(y), __FILE__, __LINE__, __PRETTY_FUNCTION__, msg); \
    }
#define BM_ASSERT_EQ(x, y, msg) BM_ASSERT_CMP_(x, y, ==, "!=", msg)
#define BM_ASSERT_LT(x, y, msg) BM_ASSERT_CMP_(x, y, <, "<", msg)
#define BM_ASSERT_LE(x, y, msg) BM_ASSERT_CMP_(x, y, <=, "<=", msg)
// the reference spells the runtime check BM_CUDART_ASSERT; both names are accepted
#define BM_HIPRT_ASSERT(expr)                                                                           \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (__builtin_expect(e_ != hipSuccess, 0))                                                      \
            throw BMEngineException(std::string("HIP runtime error: ") + hipGetErrorString(e_), __FILE__, __LINE__, \
                                    __PRETTY_FUNCTION__, #expr);                                        \
    } while (0)
#define BM_CUDART_ASSERT(expr) BM_HIPRT_ASSERT(expr)

namespace bmengine {
namespace core {

enum class DataType { kDouble, kFloat, kHalf, kInt8, kInt16, kInt32, kBFloat16, kFP8_E4M3, kFP8_E5M2 };
const char* get_data_type_name(DataType dtype);
DataType name_to_data_type(const std::string& name);
size_t get_elem_size(DataType dtype);
size_t get_numel(const std::vector<size_t>& size);

// element type of a host vector handed to Context::tensor_of (dtype.h:27-49)
template <typename T> struct DTypeDeducer {
    static DataType data_type() { throw std::runtime_error("data_type must be overwrite"); }
};
template <> struct DTypeDeducer<int> { static DataType data_type() { return DataType::kInt32; } };
template <> struct DTypeDeducer<int8_t> { static DataType data_type() { return DataType::kInt8; } };
template <> struct DTypeDeducer<float> { static DataType data_type() { return DataType::kFloat; } };
template <> struct DTypeDeducer<void*> { static DataType data_type() { return DataType::kDouble; } };   // 8-byte addresses

// how a Linear weight (dim_out, dim_in) is split over the tensor-parallel ranks (tensor.h:18-21): COLUMNAR = the last
// dimension, ROW = the one before it
enum class DistLayout { COLUMNAR, ROW, REPLICATED };
DistLayout transpose_layout(DistLayout dist_layout);
const char* get_dist_layout_name(DistLayout dist_layout);

// Engine ranks that share ONE device (the one-GPU test mode of hostcpp/bm_engine.cpp) exchange through kernels that wait for
// each other inside a launch, so no two of their streams may sit on one hardware queue.  The engine creates every rank's
// SECONDARY stream up front (consecutively, in the runtime's lowest-priority queue pool, which nothing else uses) and marks its
// rank threads with it; zl_shim_stream_create_with_priority / zl_shim_stream_destroy -- what the reference's
// cudaStreamCreateWithPriority / cudaStreamDestroy become (refshim/cuda_runtime.h) -- then hand that stream out and leave it
// alive.  Threads that are not so marked (every rank of a distinct-device engine, every other caller) get the plain HIP calls.
void set_shared_device_rank(int share_index, hipStream_t secondary = nullptr);     // -1: not a shared-device rank thread (the default)
int shared_device_rank();
void forget_secondary_stream(hipStream_t s);     // the engine is about to destroy it

struct Stream_ {
    hipStream_t ptr;
    std::function<void(hipStream_t)> deleter;
    Stream_(hipStream_t p, std::function<void(hipStream_t)> d) : ptr(p), deleter(std::move(d)) {}
    Stream_(const Stream_&) = delete;
    Stream_& operator=(const Stream_&) = delete;
    ~Stream_() { if (deleter) deleter(ptr); }
};
typedef std::shared_ptr<Stream_> Stream;

// ---- deferred launches: what lets the boundary fuse ACROSS two calls of the reference's unchanged call sequence (round 6) ----------
// The reference's layer code calls LayerNorm::forward and hands the result to Linear::forward, calls Linear::forward and hands the
// result to element_add_scale_out (block.cpp:86-143): two launches each where the kernels of this repository fuse the norm into the
// GEMV's prologue and the residual add into its epilogue.  So a producer may hand back its result tensor WITHOUT launching and leave a
// DeferredOp behind; a consumer that recognises the address (find_deferred, before it touches any operand) launches the fused form.
// Everything else stays exactly as it was:
//   * Tensor::data() on memory that overlaps a deferred op's result OR its input launches the op first, on its stream -- whoever reads
//     the result finds it, whoever is about to overwrite the input finds the op already queued in front of him (stream order);
//   * a result nobody ever reads dies with its tensor: the op holds only a weak token of the result's block and is dropped, unlaunched,
//     once that expires (checked whenever the list is consulted);
//   * per thread (a Context is bound to its thread), a handful of entries; ZL_BOUNDARY_FUSE=0 turns deferral off.
class Context;
struct DeferredOp {
    int kind = 0;                                     // 1: RMSNorm rows -> y      2: W4A16 linear -> y      3: decode attention split merge -> y
    const void* y = nullptr;                          // where the result belongs
    size_t y_bytes = 0;
    const void* x = nullptr;                          // the input a late launch reads
    size_t x_bytes = 0;
    std::weak_ptr<void> y_alive;
    hipStream_t stream = nullptr;
    std::function<void()> launch;                     // the unfused launch (keeps the input tensors alive, NOT the result)
    bool consumed = false;                            // a fused consumer already used it (the result was never materialised)
    // kind 1: what a GEMV with a fused norm prologue needs
    const uint16_t* norm_w = nullptr;
    float eps = 0.f;
    int64_t rows = 0, dim = 0;
    // kind 2: the same linear with a residual operand / another output (residual may be null)
    std::function<void(const uint16_t* residual, uint16_t* out)> launch_into;
    // ... or as the fused qkv projection + neox rotation, q / k / v rows scattered through per-task tables (zl_w4a16_qkv_rope_scatter_ex);
    // returns false when the kernels do not cover the shape
    std::function<bool(const float* cosv, const float* sinv, const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                       uint16_t* const* v_bufs, uint16_t* q_out, int64_t h, int64_t hkv, int64_t d)> launch_rope;
    int64_t m = 0, n = 0;
    // ... or as the fused gate|up projection with the activation in its epilogue: out (m, n / 2) = silu(y[:, :n/2]) * y[:, n/2:] from a
    // row-interleaved packing of the same weight (made once, cached by the weight's identity) -- what functions::gate_fuse computes
    // from y, bit for bit (feedforward.cpp:107-170 under CPM_FUSE_FF_IN); returns false when the kernels do not cover the case
    std::function<bool(const Context& ctx, uint16_t* out)> launch_gated;
    // kind 3: the half-precision split records of a decode attention launch that has ALREADY run (zl_decode_attn_splits_h_mask); what is
    // held back is only their merge -- the attention output projection takes the records over in its prologue (zl_w4a16_gemm_attn_merge_h),
    // `launch` is the stand-alone merge (zl_decode_attn_combine_h, the same arithmetic).  rows = tasks, dim = heads x 128; x = the records.
    const int32_t* attn_buf_lens = nullptr;
    int64_t split_len = 0, max_splits = 0;
    std::shared_ptr<void> keep;                       // the records' and the lengths' blocks, for whoever takes the op over
};
bool boundary_fusion_enabled();
void defer_op(DeferredOp&& op);
DeferredOp* find_deferred(const void* y, int kind);  // by result address; never launches
void drop_deferred(const void* y);                    // a consumer launched the fused form: the plain one is not needed
void flush_deferred_touching(const void* p, size_t bytes);
void flush_deferred_producing(const void* p, size_t bytes);   // only ops whose RESULT overlaps: for a wrapper's read-only operands
// [p, p + bytes) is about to be OVERWRITTEN by a fused consumer (which does not go through Tensor::data()): ops that read it as their
// input are launched first -- except ops a fused consumer has already used: their result is not materialised (that would be the
// launch the fusion saves) but POISONED -- a later Tensor::data() on it throws instead of returning stale rows.
void retire_deferred_inputs(const void* p, size_t bytes);
void flush_all_deferred();

class Context;
struct Storage;   // ref-counted device (or host) block
// private/allocator.h:13-66 as far as layer code sees it: the base address of the KV-cache arena (multi_head_latent_attention.cpp:
// 924-925) and how much device memory is left (attention.cpp:159-170 sizes its KV-head splits of a long prompt by it).  Storage
// here comes from a size-class pool over hipMalloc, so the numbers are the device's own (hipMemGetInfo).
class MemoryAllocator {
    void* base_ptr_ = nullptr;

public:
    explicit MemoryAllocator(void* base = nullptr) : base_ptr_(base) {}
    char* get_base_ptr() { return (char*)base_ptr_; }
    void set_base_ptr(void* base) { base_ptr_ = base; }
    size_t get_memory_limit() const;
    size_t used_memory() const;
    size_t get_free_memory() const { return get_memory_limit() - used_memory(); }
    void freeze_model_memory() {}                   // (allocator.h: the reference pins what was allocated so far; nothing moves here)
};

class Tensor {
public:
    std::shared_ptr<Tensor> quant_scale;           // int8 activations carry their per-row scale (linear.cpp:557-635)
    void set_quant_scale(const Tensor& scale) { quant_scale = std::make_shared<Tensor>(scale); }

    Tensor();
    ~Tensor();
    Tensor(const Tensor&);
    Tensor(Tensor&&) noexcept;
    Tensor& operator=(const Tensor&);
    Tensor& operator=(Tensor&&) noexcept;

    long id() const { return id_; }
    void set_id(long i) const { id_ = i; }
    const std::string& name() const { return name_; }
    void set_name(const std::string& n) const { name_ = n; }

    DataType dtype() const { return dtype_; }
    int ndim() const { return (int)shape_.size(); }
    size_t numel() const;
    size_t nbytes() const { return numel() * get_elem_size(dtype_); }
    bool empty() const { return numel() == 0; }
    const std::vector<size_t>& size() const { return shape_; }
    const std::vector<size_t>& shape() const { return shape_; }
    int normalize_dim(int dim) const;
    size_t size(int dim) const { return shape_[normalize_dim(dim)]; }
    size_t stride(int dim) const { return strides_[normalize_dim(dim)]; }
    size_t stride_bytes(int dim) const { return stride(dim) * get_elem_size(dtype_); }

    void* data() const;                              // throws on an empty tensor; launches a deferred op whose operands it touches (below)
    void* nullable_data() const;                     // nullptr for an empty tensor; never launches anything
    std::weak_ptr<void> storage_token() const;       // expires when the last tensor over this block dies (deferred ops watch their result's)
    void* mutable_data() { return data(); }
    template <typename T> T* data() const { return reinterpret_cast<T*>(data()); }
    template <typename T> void* nullable_data() const { return nullable_data(); }
    template <typename T> T* mutable_data() { return reinterpret_cast<T*>(data()); }
    size_t mem_bytes() const;
    int device() const { return device_; }           // -1: host

    Tensor view(const std::vector<size_t>& size) const;
    Tensor view_type(const std::vector<size_t>& size, DataType dtype) const;
    Tensor view_unchecked(const std::vector<size_t>& size, DataType dtype) const;
    Tensor index_dim0(size_t i) const;
    Tensor slice_dim0(size_t from, size_t to) const;
    Tensor slice_dim0_len(size_t from, size_t len) const { return slice_dim0(from, from + len); }
    Tensor virtual_slice(size_t from, size_t len, int dim = -1) const;   // strided: no longer continuous
    Tensor virtual_transpose(int dim0, int dim1) const;                  // swaps two strides: no longer continuous
    Tensor view_uncontinuous(const std::vector<size_t>& size) const;     // split / merge dimensions of a strided tensor
    bool is_continuous() const;
    std::vector<Tensor> chunk() const;
    Tensor squeeze() const;

    void from_buffer(const void* host, bool async = false, hipStream_t stream = nullptr);
    void to_buffer(void* host, hipStream_t stream = nullptr) const;
    Tensor to_device(int dev_id = -1) const;         // one device per process: the tensor itself (or a copy of a host tensor)
    template <typename T> std::vector<T> to_vector(hipStream_t stream = nullptr) const {
        std::vector<T> v(nbytes() / sizeof(T));
        to_buffer(v.data(), stream);
        return v;
    }
    // wrap memory owned elsewhere (device = -1: host); own_ptr: free it (hipFree / free) with the last reference
    static Tensor from_external(const std::vector<size_t>& shape, DataType dtype, void* ptr, size_t nbytes, int device = -1,
                                bool own_ptr = false);
    std::string info(int level = 0) const;
    friend std::ostream& operator<<(std::ostream& os, const Tensor& tensor);

private:
    friend class Context;
    std::shared_ptr<Storage> mem_;
    size_t offset_ = 0;                              // bytes into mem_
    std::vector<size_t> shape_, strides_;            // strides in elements
    DataType dtype_ = DataType::kHalf;
    int device_ = -1;
    bool param_ = false;                             // Context::parameter(): shape and dtype, no memory yet
    mutable long id_ = -1;
    mutable std::string name_;
    void set_shape(const std::vector<size_t>& s);
};

class ContextImpl;
class WithDevice;
class ScopeDevice;
class WithDebug;
// One Context per device per thread (context.cpp:275-280).  All launches of the nn:: wrappers go to current_stream().
class Context {
public:
    static const std::string EMPTY_STR;
    explicit Context(int device, int rank = 0, int world_size = 1);
    virtual ~Context();                              // model::ModelContext derives from it (src/model/model_context.h:77)
    Context(const Context&) = delete;
    Context(Context&&) noexcept;

    int active_device() const;
    int active_device_idx() const { return 0; }      // index in this context's device list: always the one device
    const std::vector<int> devices() const { return {active_device()}; }
    virtual bool switch_to_device(int idx) const;    // true when idx is the context's device; anything else throws
    int rank() const;
    int world_size() const;
    int get_compute_capability() const;              // 90: the reference gates dual-stream / wmma paths on > 80
    int get_mp_count() const;
    int get_max_shared_memory() const;
    int get_L2_cache_size() const;

    Stream current_stream() const;
    void set_current_stream(Stream s);
    hipStream_t current_cuda_stream() const;         // (name kept from the reference)
    Stream get_stream() const;                       // a fresh non-blocking stream
    // Int8Linear (linear.cpp:600-616) hands this to cublasLtMatmul; here the product is int8_op::int8_gemm_nt and the
    // handle is only a token passed through (the refshim's cublasLtMatmul ignores it)
    void* current_cublas_handle() const { return nullptr; }

    Tensor null_tensor() const { return Tensor(); }
    Tensor tensor(const std::vector<size_t>& size, DataType dtype, const std::string& name = EMPTY_STR,
                  size_t round_up_bytes = 1024) const;
    Tensor tensor_s(const std::vector<long>& size, DataType dtype) const;
    Tensor cuda(const Tensor& cpu_tensor) const;     // host -> device copy on the current stream (synchronous)
    template <typename T> Tensor tensor_of(const std::vector<T>& data, const std::vector<size_t>& shape, DataType dt) const {
        Tensor t = tensor(shape, dt);
        BM_ASSERT_EQ(data.size() * sizeof(T), t.nbytes(), "data not fit for tensor");
        t.from_buffer(data.data(), false, current_cuda_stream());     // ordered behind the block's previous user on this stream
        return t;
    }
    template <typename T, typename DTD = DTypeDeducer<T>>
    Tensor tensor_of(const std::vector<T>& data, const std::vector<size_t>& shape) const {
        if (data.empty()) return Tensor();
        Tensor t = tensor(shape, DTD::data_type());
        if (data.size() != t.numel()) throw std::runtime_error("data not fit for tensor");
        t.from_buffer(data.data(), false, current_cuda_stream());
        return t;
    }
    template <typename T> Tensor tensor_of(const std::vector<T>& data) const { return tensor_of(data, {data.size()}); }
    const Tensor copy(const Tensor& t) const;

    // ---- parameters (context.cpp:640-789) ---------------------------------------------------------------------------
    // parameter(): shape and dtype without memory; load_parameter(): fill it from state_dict[name] (host or device
    // tensors) -- whole when !parallel / world_size 1 / REPLICATED, else this rank's shard along the layout's dimension
    // (ROW: dim -2, contiguous rows; COLUMNAR: dim -1, gathered row by row on the host).  int16 sources are bf16 bits.
    Tensor parameter(const std::vector<size_t>& size, DataType dtype) const;
    Tensor distribute_parameter(const Tensor& param, DistLayout layout) const;
    void load_parameter(Tensor* weight, const std::string& name, const std::map<std::string, const Tensor>& state_dict,
                        bool parallel, DistLayout layout) const;
    void load_parameter_part(Tensor* weight, const std::string& name, const std::map<std::string, const Tensor>& state_dict,
                             DistLayout layout, size_t part, size_t total) const;
    void assign_or_copy(Tensor* dst, const Tensor* src) const;     // host source: upload into dst (allocating it); device: alias
    const Tensor* identity(const Tensor* tensor, const std::string& name) const { (void)name; return tensor; }   // one device
    void clear_identity_cache() {}
    void init_parameter(const std::string& name, Tensor* tensor) const;   // allocate + zero

    WithDevice with_device(int dev_id) const;
    ScopeDevice scope_device(int dev_id) const;
    WithDebug with_debug(int debug_level) const;

    // a persistent zero-initialised device buffer of at least `bytes` (the caller-provided scratch of the C ABI's K-split
    // launchers, zl_w4_opts_t::scratch): grown on demand, never while a stream capture is open
    void* scratch(size_t bytes) const;
    size_t scratch_bytes() const;

    size_t used_memory() const;
    size_t peak_memory() const;
    void mem_gc();                                   // return cached blocks to the driver

    void recordEvent(const std::string& name, int ev_level = 2, float flops = 0) const;
    int debug() const { return 0; }
    void enable_debug(int) const {}
    bool checking_numerics() const { return false; }
    int event_level() const { return -1; }
    void set_event_level(int) const {}
    void print_events() {}
    void print_memory_summary() const;
    int current_layer() const { return cur_layer_; }
    virtual void set_current_layer(int i) { cur_layer_ = i; }
    bool is_layer(int layer, int rank = 0) const { return cur_layer_ == layer && this->rank() == rank; }
    int high_precision() const { return high_precision_; }
    void set_high_precision(int level) { high_precision_ = level; }
    virtual bool is_BSHD() const { return bshd_; }
    void set_BSHD(bool b) { bshd_ = b; }

    // tensor parallelism: sum over the ranks of the node (ModelContext::reduce_sum, model_context.cpp:203-242).  The hook
    // is installed by the communicator owner (RCCL / the one-shot xGMI all-reduce); with world_size 1 it is the identity.
    typedef std::function<void(Tensor& data, hipStream_t stream)> ReduceHook;
    void set_reduce_hook(ReduceHook h);
    virtual Tensor reduce_sum(Tensor& data, DataType out_type) const;
    // the other two collectives of context.h:157-159: with one rank the identity; with more they need a communicator this
    // context does not own (the all-reduce hook above is the only exchange the decode path has) and say so
    virtual Tensor reduce_scatter(const Tensor& data) const;
    virtual Tensor all_gather(const Tensor& data) const;
    // the KV-cache arena's allocator (context.h:121; layer code asks it for the arena's base address only)
    MemoryAllocator* get_cache_allocator() const;
    MemoryAllocator* get_allocator() const;         // (context.h:120) the same object: one pool serves tensors and caches here
    void set_cache_arena(void* base);
    // context.h:169-171: the reference carves a second arena for the reduce stream of dual_stream_encode (block.cpp:252-290, 429) so
    // that nothing freed under one stream is recycled under the other; here: a second size-class pool, selected while
    // use_cache_alloc(true) is in force (tensors return to the pool they came from)
    void reserve_cache_alloc(size_t bytes);
    void free_cache_alloc();
    void use_cache_alloc(bool b);

private:
    std::unique_ptr<ContextImpl> pimpl;
    int cur_layer_ = -1, high_precision_ = 0;
    bool bshd_ = true;                               // Python default flash_attention=True (zhilight/dynamic_batch.py:36)
};

// context.h:189-194: the reference pauses its compacting allocator while raw buffer addresses are in use; tensors here never
// move (see the header comment), so holding one is free
class GCStopper {
public:
    explicit GCStopper(const Context&) {}
    ~GCStopper() {}
};

}  // namespace core
}  // namespace bmengine

namespace std {
// BM_ASSERT_EQ prints both sides with std::to_string; the reference extends it to dtypes (dtype.h:125-129)
static inline std::string to_string(bmengine::core::DataType dt) { return bmengine::core::get_data_type_name(dt); }
}  // namespace std
