// ref_glue.cpp -- the pybind11 TEST module zl_reflinear: the reference's own host translation units (compiled unmodified from
// /root/reference into libzhilight_amd_host.so, zhilight_amd/build.py: build_host) driven from numpy.  This file: nn::Linear (the
// reference's class, the reference's code: construct -> load_state_dict -> forward) for the GPTQ (Int4GPTQ), INT8 (Int8Linear),
// FP8-block and unquantised flavours; ref_attention_glue.cpp / ref_block_glue.cpp / ref_model_glue.cpp add the layers above it.
// tests/test_gpu_refcompile.py compares the results with the oracle.  Test infrastructure: nothing in the product links this file.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "model/model_context.h"
#include "nn/linear/linear.h"
#include "nn/quant/fp8/fp8.h"
#include "nn/quant/gptq/gptq.h"
#include "nn/quant/marlin/marlin.h"
#include "3rd/deep_gemm/deep_gemm_api.h"
#include "zhilight_amd.h"

namespace nn {
namespace gptq {   // from hostcpp/nn_amd.h (not included: it re-declares the reference's gptq.h with its default arguments)
void amd_weight_cache_clear();
size_t amd_weight_cache_size();
}  // namespace gptq
}  // namespace nn

// ---- the test module ----------------------------------------------------------------------------------------------
namespace py = pybind11;
using bmengine::core::Context;
using bmengine::core::DataType;
using bmengine::core::Tensor;

namespace {

DataType np_dtype(const py::array& a) {
    const char k = a.dtype().kind();
    const size_t sz = (size_t)a.dtype().itemsize();
    if (k == 'f' && sz == 2) return DataType::kHalf;
    if (k == 'f' && sz == 4) return DataType::kFloat;
    if (k == 'i' && sz == 4) return DataType::kInt32;
    if (k == 'i' && sz == 1) return DataType::kInt8;
    if ((k == 'i' || k == 'u') && sz == 2) return DataType::kInt16;
    if (k == 'u' && sz == 1) return DataType::kInt8;
    throw std::runtime_error("unsupported numpy dtype");
}
// a HOST tensor aliasing a C-contiguous numpy array (what the reference's python binding hands load_state_dict)
Tensor host_tensor(const py::array& a, const std::string& name) {
    if (!(a.flags() & py::array::c_style)) throw std::runtime_error(name + ": C-contiguous array expected");
    std::vector<size_t> shape(a.shape(), a.shape() + a.ndim());
    Tensor t = Tensor::from_external(shape, np_dtype(a), const_cast<void*>(a.data()), (size_t)a.nbytes(), -1, false);
    t.set_name(name);
    return t;
}
py::array to_numpy(const Context& ctx, const Tensor& t) {
    std::vector<py::ssize_t> shape(t.shape().begin(), t.shape().end());
    py::dtype dt = t.dtype() == DataType::kHalf ? py::dtype("float16") : t.dtype() == DataType::kFloat ? py::dtype("float32")
                 : t.dtype() == DataType::kInt32 ? py::dtype("int32") : t.dtype() == DataType::kBFloat16 ? py::dtype("uint16") /* raw bits */
                 : py::dtype("int8");
    py::array out(dt, shape);
    t.to_buffer(out.mutable_data(), ctx.current_cuda_stream());
    return out;
}

// One reference nn::Linear, loaded from numpy arrays, run on numpy activations
class RefLinear {
public:
    RefLinear(int dim_in, int dim_out, int quant_type, int group_size, bool sym, bool act_order, const std::string& act_fn, int device,
              bool weight_transposed, bool bf16)
        : ctx_(device), bf16_(bf16) {
        model::QuantConfig qc(quant_type);
        qc.group_size = group_size;
        qc.sym = sym;
        qc.act_order = act_order;
        linear_.reset(new nn::Linear(ctx_, dim_in, dim_out, act_fn, qc, false, weight_transposed, false, bmengine::core::DistLayout::COLUMNAR, bf16 ? DataType::kBFloat16 : DataType::kHalf));
    }
    void load(const std::map<std::string, py::array>& arrays, const std::string& prefix) {
        std::map<std::string, const Tensor> sd;
        for (auto& kv : arrays) sd.emplace(kv.first, host_tensor(kv.second, kv.first));
        linear_->load_state_dict(ctx_, sd, prefix, false);
    }
    py::array forward(const py::array& x) {
        Tensor hx = host_tensor(x, "x");
        Tensor dx = ctx_.tensor(hx.shape(), hx.dtype());
        dx.from_buffer(hx.data(), false, ctx_.current_cuda_stream());
        if (bf16_ && dx.dtype() == DataType::kInt16) dx = dx.view_type(dx.shape(), DataType::kBFloat16);   // numpy has no bfloat16: int16 bits
        Tensor y = linear_->forward(ctx_, dx);
        return to_numpy(ctx_, y);
    }
    py::array dequant_weight() { return to_numpy(ctx_, linear_->get_dequant_weight(ctx_)); }
    std::string layer_type() const { return linear_->layer_type(); }

private:
    Context ctx_;
    bool bf16_;
    std::unique_ptr<nn::Linear> linear_;
};

}  // namespace

void bind_ref_attention(py::module_& m);      // hostcpp/ref_attention_glue.cpp: the reference's nn::Attention decode / encode paths
void bind_ref_block(py::module_& m);          // hostcpp/ref_block_glue.cpp: the reference's nn::EncoderLayer, nn::FeedForward
void bind_ref_model(py::module_& m);          // hostcpp/ref_model_glue.cpp: the reference's model::LLaMA

PYBIND11_MODULE(zl_reflinear, m) {
    m.doc() = "the reference's nn::Linear (src/nn/linear/linear.cpp, compiled unmodified) on the MI355X boundary";
    py::class_<RefLinear>(m, "RefLinear")
        .def(py::init<int, int, int, int, bool, bool, const std::string&, int, bool, bool>(), py::arg("dim_in"), py::arg("dim_out"), py::arg("quant_type"),
             py::arg("group_size") = 128, py::arg("sym") = false, py::arg("act_order") = false, py::arg("act_fn") = "", py::arg("device") = 0,
             py::arg("weight_transposed") = false, py::arg("bf16") = false)
        .def("load", &RefLinear::load)
        .def("forward", &RefLinear::forward)
        .def("dequant_weight", &RefLinear::dequant_weight)
        .def("layer_type", &RefLinear::layer_type);
    bind_ref_attention(m);
    bind_ref_block(m);
    bind_ref_model(m);
    m.def("weight_cache_size", &nn::gptq::amd_weight_cache_size);
    m.def("weight_cache_clear", &nn::gptq::amd_weight_cache_clear);
    py::register_exception<BMEngineException>(m, "BMEngineException");
}
