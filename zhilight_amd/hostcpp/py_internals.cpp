// py_internals.cpp -- pybind11 test module over the C++ operator layer (the role of the reference's
// `zhilight.internals_`, tests/py_export_internal/*.cpp): numpy in, numpy out, every call goes
//     Python -> this file -> nn:: / int8_op:: wrapper (nn_amd.cpp, reference signatures) -> C ABI -> HIP kernel
// so the GPU tests can check the C++ boundary itself against the CPU oracle.  No torch types anywhere.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "bm_functions.h"
#include "nn_amd.h"

namespace py = pybind11;
using namespace bmengine;
using core::DataType;
using core::Tensor;

namespace {

DataType dt_of(const py::array& a) {
    const char k = a.dtype().kind();
    const auto sz = a.itemsize();
    if (k == 'f' && sz == 2) return DataType::kHalf;
    if (k == 'f' && sz == 4) return DataType::kFloat;
    if (k == 'f' && sz == 8) return DataType::kDouble;
    if ((k == 'i' || k == 'u') && sz == 1) return DataType::kInt8;
    if ((k == 'i' || k == 'u') && sz == 2) return DataType::kInt16;     // bf16 travels as int16 views (zhilight/llama.py:190-199)
    if ((k == 'i' || k == 'u') && sz == 4) return DataType::kInt32;
    if ((k == 'i' || k == 'u') && sz == 8) return DataType::kDouble;    // addresses
    throw std::runtime_error("unsupported numpy dtype");
}

struct PyCtx {
    std::unique_ptr<core::Context> ctx;
    explicit PyCtx(int device) : ctx(new core::Context(device)) {}

    Tensor up(const py::array& a, int as_dtype = -1) const {
        py::array c = py::array::ensure(a, py::array::c_style);
        std::vector<size_t> shape(c.shape(), c.shape() + c.ndim());
        if (c.size() == 0) return Tensor();
        Tensor t = ctx->tensor(shape, as_dtype >= 0 ? (DataType)as_dtype : dt_of(c));
        t.from_buffer(c.data(), false, ctx->current_cuda_stream());
        return t;
    }
    Tensor up_opt(const py::object& o, int as_dtype = -1) const { return o.is_none() ? Tensor() : up(py::cast<py::array>(o), as_dtype); }
    py::array down(const Tensor& t, const char* np_dtype) const {
        std::vector<py::ssize_t> shape(t.shape().begin(), t.shape().end());
        py::array out(py::dtype(np_dtype), shape);
        if (t.numel()) t.to_buffer(out.mutable_data(), ctx->current_cuda_stream());
        return out;
    }
    // device array of raw pointers to per-task buffers (RagBufferContext::buf_k_addr)
    Tensor addr_table(const std::vector<Tensor>& bufs) const {
        std::vector<void*> p;
        for (auto& b : bufs) p.push_back(b.data());
        Tensor t = ctx->tensor({p.size()}, DataType::kDouble);
        t.from_buffer(p.data(), false, ctx->current_cuda_stream());
        return t;
    }
    std::vector<Tensor> up_list(const py::list& l) const {
        std::vector<Tensor> v;
        for (auto h : l) v.push_back(up(py::cast<py::array>(h)));
        return v;
    }
};

const int kBF = (int)DataType::kBFloat16;
const char* fdt(bool bf16) { return bf16 ? "uint16" : "float16"; }

}  // namespace

PYBIND11_MODULE(zl_internals, m) {
    m.doc() = "C++ operator layer of zhilight_amd (bmengine-on-HIP shim + nn:: wrappers) exposed for tests";
    py::register_exception<BMEngineException>(m, "BMEngineException", PyExc_RuntimeError);

    py::class_<PyCtx>(m, "Context")
        .def(py::init<int>(), py::arg("device") = 0)
        .def("used_memory", [](PyCtx& c) { return c.ctx->used_memory(); })
        .def("peak_memory", [](PyCtx& c) { return c.ctx->peak_memory(); })
        .def("set_bshd", [](PyCtx& c, bool b) { c.ctx->set_BSHD(b); })
        // ---- MoE router + dispatch / combine through the C++ names (ff_kernel.h)
        .def("moe_route", [](PyCtx& c, py::array logits, py::object bias, int num_group, int topk_group, int k, int k_ext, bool renorm, float scale,
                             std::string scoring, bool bf16) {
            Tensor lg = c.up(logits, bf16 ? kBF : -1), none;
            auto r = topk_group > 1 ? nn::group_topk_softmax(*c.ctx, lg, c.up_opt(bias), none, none, num_group, topk_group, k, k_ext, renorm, scale, scoring)
                                    : nn::top_k_softmax(*c.ctx, lg, none, none, k, k_ext, renorm, scale, scoring);
            return py::make_tuple(c.down(std::get<0>(r), "float32"), c.down(std::get<1>(r), "int32"));
        }, py::arg("logits"), py::arg("bias"), py::arg("num_group"), py::arg("topk_group"), py::arg("k"), py::arg("k_ext"), py::arg("renorm"),
           py::arg("scale"), py::arg("scoring"), py::arg("bf16") = false)
        .def("moe_dispatch", [](PyCtx& c, py::array exp_ids, py::array order, std::vector<int> all_loads, int num_experts, int block_m) {
            Tensor ids = c.up(exp_ids), ord = c.up(order);
            Tensor keys = nn::plus_for_sort(*c.ctx, ids, num_experts);
            Tensor rev = nn::calc_reverse_idx(*c.ctx, ids, ord, all_loads, num_experts, false);
            auto f = nn::fill_m_indices_padded_indices(*c.ctx, all_loads, block_m, num_experts, false);
            return py::make_tuple(c.down(keys, "int32"), c.down(rev, "int32"), c.down(std::get<0>(f), "int32"), c.down(std::get<1>(f), "int32"),
                                  std::get<2>(f));
        })
        .def("moe_combine", [](PyCtx& c, py::list parts, py::array experts, py::array index, py::array weights) {
            std::vector<Tensor> inputs;
            for (auto h : parts) inputs.push_back(h.is_none() ? Tensor() : c.up(py::cast<py::array>(h)));
            Tensor out = nn::sum_experts(*c.ctx, inputs, Tensor(), c.up(experts), c.up(index), c.up(weights), false);
            return c.down(out, "float16");
        })
        .def("mha_fwd_kvcache_mla", [](PyCtx& c, py::array q, py::array kcache, py::array seqlens_k, py::array block_table, float scale, bool bf16) {
            Tensor tq = c.up(q, bf16 ? kBF : -1), tk = c.up(kcache, bf16 ? kBF : -1), ts = c.up(seqlens_k), tb = c.up(block_table);
            auto meta = ds::get_mla_metadata(*c.ctx, ts, tq.size(1) * tq.size(2));
            Tensor out = c.ctx->tensor({tq.size(0), tq.size(1), tq.size(2), 512}, tq.dtype());
            auto r = ds::mha_fwd_kvcache_mla(*c.ctx, tq, tk, 512, ts, tb, scale, false, std::get<0>(meta), std::get<1>(meta), out);
            return py::make_tuple(c.down(out, bf16 ? "int16" : "float16"), c.down(std::get<1>(r), "float32"));
        }, py::arg("q"), py::arg("kcache"), py::arg("seqlens_k"), py::arg("block_table"), py::arg("scale"), py::arg("bf16") = false)
        // ---- tensor surface (views, slices) exercised directly
        .def("tensor_roundtrip", [](PyCtx& c, py::array a, size_t from, size_t to) {
            Tensor t = c.up(a);
            Tensor s = t.slice_dim0(from, to);
            Tensor v = s.view({s.numel()});
            BM_ASSERT(v.is_continuous(), "slice of dim 0 stays continuous");
            return c.down(v, py::str(a.dtype()).cast<std::string>().c_str());
        })
        // ---- GPTQ
        .def("gptq_gemm_k_major", [](PyCtx& c, py::array x, py::array qw, py::array qz, py::array sc, py::object bias, bool sym,
                                      bool prepack) {
            Tensor tx = c.up(x), tq = c.up(qw), tz = c.up(qz), ts = c.up(sc), tb = c.up_opt(bias);
            if (prepack) {   // what Int4GPTQ::load_state_dict does once per weight
                auto p = nn::gptq::amd_pack_k_major(*c.ctx, tq, tz, ts);
                tq = p.q_weight; tz = p.qzeros; ts = p.scales;
            }
            Tensor y = nn::gptq::gptq_gemm_k_major(*c.ctx, tx, tq, tz, ts, Tensor(), Tensor(), tb.numel() ? &tb : nullptr, sym);
            return c.down(y, "float16");
        }, py::arg("x"), py::arg("qweight"), py::arg("qzeros"), py::arg("scales"), py::arg("bias") = py::none(), py::arg("sym") = false,
           py::arg("prepack") = true)
        .def("gptq_gemm_legacy", [](PyCtx& c, py::array x, py::array qweight, py::array qzeros, py::array scales, py::object g_idx,
                                    bool use_exllama, int group_size) {
            // nn::gptq::gptq_gemm (GPTQ_KERNEL_ALGO=0, q_gemm.cu:874-918): operands as Int4GPTQ::preprocess_weight(trans = false) leaves them
            Tensor gi = c.up_opt(g_idx);
            return c.down(nn::gptq::gptq_gemm(*c.ctx, c.up(x), c.up(qweight), c.up(qzeros), c.up(scales), gi, use_exllama, group_size,
                                              (int)qweight.shape(1), (int)qweight.shape(1)), "float16");
        })
        .def("gptq_reconstruct", [](PyCtx& c, py::array qweight, py::array qzeros, py::array scales, py::object g_idx) {
            Tensor qw = c.up(qweight), qz = c.up(qzeros), sc = c.up(scales), gi = c.up_opt(g_idx);
            Tensor out = c.ctx->tensor({qw.size(0) * 8, qw.size(1)}, DataType::kHalf);
            nn::gptq::reconstruct_gptq(qw.data<uint32_t>(), qz.data<uint32_t>(), reinterpret_cast<const __half*>(sc.data()),
                                       gi.numel() ? gi.data<int>() : nullptr, reinterpret_cast<__half*>(out.mutable_data()), (int)qw.size(0) * 8,
                                       (int)qw.size(1), (int)qz.size(0), c.ctx->current_cuda_stream());
            return c.down(out, "float16");
        })
        .def("gptq_dequant_k_major", [](PyCtx& c, py::array qw, py::array qz, py::array sc, bool prepack) {
            Tensor tq = c.up(qw), tz = c.up(qz), ts = c.up(sc);
            if (prepack) {
                auto p = nn::gptq::amd_pack_k_major(*c.ctx, tq, tz, ts);
                tq = p.q_weight; tz = p.qzeros; ts = p.scales;
            }
            return c.down(nn::gptq::dequant_k_major(*c.ctx, tq, tz, ts), "float16");
        }, py::arg("qweight"), py::arg("qzeros"), py::arg("scales"), py::arg("prepack") = false)
        .def("gemm_fuse_gate_in", [](PyCtx& c, py::array x, py::array q1, py::array z1, py::array s1, py::array q2, py::array z2,
                                      py::array s2) {
            Tensor y = nn::gptq::gemm_fuse_gate_in(*c.ctx, c.up(x), c.up(q1), c.up(z1), c.up(s1), Tensor(), c.up(q2), c.up(z2), c.up(s2),
                                                   Tensor(), false);
            return c.down(y, "float16");
        })
        .def("gemm_moe_steps", [](PyCtx& c, py::array x, py::array q1, py::array z1, py::array s1, py::array q2, py::array z2,
                                   py::array s2, py::array a_down, py::array qd, py::array zd, py::array sd, py::array ids, py::array wts,
                                   int n_shared, bool prepack) {
            // the two fused MoE GEMVs of FeedForward under FUSE_GPTQ_MOE: up = silu(x . gate_e) * (x . up_e) per (token, expert),
            // down = weighted sum over the token's experts of a_down[m, t] . W_e; returns (up (M, T, n_ff), down (M, dim))
            Tensor tx = c.up(x), ta = c.up(a_down), ti = c.up(ids), tw = c.up(wts);
            Tensor tq1 = c.up(q1), tz1 = c.up(z1), ts1 = c.up(s1), tq2 = c.up(q2), tz2 = c.up(z2), ts2 = c.up(s2);
            Tensor tqd = c.up(qd), tzd = c.up(zd), tsd = c.up(sd);
            Tensor up, down;
            if (prepack) {     // the load path: pack once, then the *_packed entry points
                auto wu = nn::gptq::amd_pack_moe(*c.ctx, tq1, tz1, ts1, &tq2, &tz2, &ts2);
                auto wd = nn::gptq::amd_pack_moe(*c.ctx, tqd, tzd, tsd);
                up = nn::gptq::gemm_moe_up_packed(*c.ctx, tx, wu, ti, n_shared, false);
                down = nn::gptq::gemm_moe_down_packed(*c.ctx, ta, wd, ti, tw, n_shared, false);
            } else {           // the reference's own signatures
                up = nn::gptq::gemm_moe_up(*c.ctx, tx, tq1, tz1, ts1, Tensor(), tq2, tz2, ts2, Tensor(), false, ti, n_shared, false);
                down = nn::gptq::gemm_moe_down(*c.ctx, ta, tqd, tzd, tsd, ti, tw, false, n_shared, false);
            }
            return py::make_tuple(c.down(up, "float16"), c.down(down, "float16"));
        })
        .def("gptq_load_transforms", [](PyCtx& c, py::array qweight_hf, py::array qzeros_hf) {
            // Int4GPTQ::preprocess_weight (linear.cpp:1139-1160): shuffle the words, +1 the zeros, one byte per zero
            Tensor q = c.up(qweight_hf), z = c.up(qzeros_hf);
            nn::gptq::gptq_shuffle(*c.ctx, q, Tensor());
            nn::gptq::increase_zero(*c.ctx, z);
            Tensor z8 = nn::gptq::q4_to_q8(*c.ctx, z);
            return py::make_tuple(c.down(q, "uint32"), c.down(z8, "uint8"));
        })
        .def("awq_gemm", [](PyCtx& c, py::array x, py::array qweight, py::array qzeros, py::array scales, size_t split_k_iters) {
            return c.down(nn::awq::awq_gemm(*c.ctx, c.up(x), c.up(qweight), c.up(scales), c.up(qzeros), split_k_iters), "float16");
        })
        .def("awq_dequantize", [](PyCtx& c, py::array qweight, py::array qzeros, py::array scales) {
            return c.down(nn::awq::awq_dequantize(*c.ctx, c.up(qweight), c.up(scales), c.up(qzeros), 0, 0, 0), "float16");
        })
        .def("gptq_w4a8", [](PyCtx& c, py::array x, py::array qw, py::array qz, py::array sc) {
            // W4_INT8_ALGO: scale + int8 codes at load (calc_w4a8_scale, dequant_k_major(out_type 1)), then the M > 40 branch
            Tensor tq = c.up(qw), tz = c.up(qz), ts = c.up(sc);
            nn::gptq::calc_w4a8_scale(*c.ctx, tq, tz, ts);
            Tensor w8 = nn::gptq::dequant_k_major(*c.ctx, tq, tz, ts, 1);
            Tensor y = nn::gptq::gptq_gemm_k_major(*c.ctx, c.up(x), tq, tz, ts, Tensor(), Tensor(), nullptr, false, false, nullptr, &w8);
            return py::make_tuple(c.down(y, "float16"), c.down(w8, "int8"), c.down(*w8.quant_scale, "float32"));
        })
        // ---- attention / rope / scatter
        .def("multi_query_attention_rag_buffer", [](PyCtx& c, py::array q, py::array buf_lens, py::list kbufs, py::list vbufs, py::array mask,
                                                     float scale, int max_len_buf, int m_query, bool bf16) {
            Tensor tq = c.up(q, bf16 ? kBF : -1), tl = c.up(buf_lens), tm = c.up(mask);
            auto ks = c.up_list(kbufs), vs = c.up_list(vbufs);
            Tensor ka = c.addr_table(ks), va = c.addr_table(vs);
            Tensor out = c.ctx->tensor(tq.shape(), tq.dtype());
            nn::multi_query_attention_rag_buffer(*c.ctx, tq, tl, ka, va, tm, scale, max_len_buf, out, m_query);
            return c.down(out, fdt(bf16));
        }, py::arg("q"), py::arg("buf_lens"), py::arg("k_bufs"), py::arg("v_bufs"), py::arg("mask"), py::arg("scale"), py::arg("max_len_buf"),
           py::arg("m_query"), py::arg("bf16") = false)
        .def("attention_qkv_rag_buffer", [](PyCtx& c, py::array q, py::array buf_lens, py::list kbufs, py::list vbufs, py::array mask, float scale,
                                             int max_len_buf) {
            Tensor tq = c.up(q), tl = c.up(buf_lens), tm = c.up(mask);
            auto ks = c.up_list(kbufs), vs = c.up_list(vbufs);
            Tensor ka = c.addr_table(ks), va = c.addr_table(vs);
            Tensor out = c.ctx->tensor(tq.shape(), tq.dtype());
            nn::attention_qkv_rag_buffer(*c.ctx, tq, tl, ka, va, tm, Tensor(), scale, max_len_buf, out);
            return c.down(out, "float16");
        })
        .def("rope_qk_cache", [](PyCtx& c, py::array cosv, py::array sinv, py::array in, size_t h, size_t hkv, size_t d, bool neox) {
            Tensor ti = c.up(in);
            const size_t s = ti.size(0);
            Tensor q = c.ctx->tensor({s, h * d}, DataType::kHalf), k = c.ctx->tensor({s, hkv * d}, DataType::kHalf),
                   v = c.ctx->tensor({s, hkv * d}, DataType::kHalf);
            nn::rope_qk_cache(*c.ctx, c.up(cosv), c.up(sinv), ti, q, k, v, h, hkv, d, DataType::kHalf, neox);
            return py::make_tuple(c.down(q, "float16"), c.down(k, "float16"), c.down(v, "float16"));
        })
        .def("copy_to_rag_buffer2", [](PyCtx& c, py::array placement, py::array buf_lens, py::array k_src, py::array v_src, py::list kbufs,
                                        py::list vbufs) {
            auto ks = c.up_list(kbufs), vs = c.up_list(vbufs);
            Tensor ka = c.addr_table(ks), va = c.addr_table(vs);
            nn::copy_to_rag_buffer2(*c.ctx, c.up(placement), c.up(buf_lens), c.up(k_src), c.up(v_src), &ka, &va);
            py::list ko, vo;
            for (auto& t : ks) ko.append(c.down(t, "float16"));
            for (auto& t : vs) vo.append(c.down(t, "float16"));
            return py::make_tuple(ko, vo);
        })
        // ---- norm / element-wise
        .def("layernorm", [](PyCtx& c, py::array x, py::array w, float eps, float scale) {
            nn::LayerNorm ln(*c.ctx, (int)w.size(), false, eps, scale);
            std::map<std::string, const Tensor> sd;
            sd.emplace("ln.weight", c.up(w));
            ln.load_state_dict(*c.ctx, sd, "ln");
            return c.down(ln.forward(*c.ctx, c.up(x)), "float16");
        })
        .def("layernorm_fuse_add", [](PyCtx& c, py::array a, py::array b, py::array w, float eps) {
            nn::LayerNorm ln(*c.ctx, (int)w.size(), false, eps);
            std::map<std::string, const Tensor> sd;
            sd.emplace("ln.weight", c.up(w));
            ln.load_state_dict(*c.ctx, sd, "ln");
            Tensor ta = c.up(a), sum = c.ctx->tensor(ta.shape(), ta.dtype());
            Tensor out = ln.fuse_add(*c.ctx, ta, c.up(b), sum);
            return py::make_tuple(c.down(out, "float16"), c.down(sum, "float16"));
        })
        .def("element_add_scale", [](PyCtx& c, py::array a, py::array b, float scale, bool scale_residual) {
            return c.down(nn::element_add_scale(*c.ctx, c.up(a), c.up(b), scale, scale_residual), "float16");
        })
        .def("gate_mul", [](PyCtx& c, py::array a, py::array b, std::string act) {
            Tensor ta = c.up(a);
            nn::gate_mul_inplace(*c.ctx, ta, c.up(b), act);
            return c.down(ta, "float16");
        })
        .def("gate_fuse", [](PyCtx& c, py::array a, std::string act) { return c.down(nn::gate_fuse(*c.ctx, c.up(a), act), "float16"); })
        // ---- bmengine::functions glue the attention layers call (index_select.h:32, tensor_ops.h:13)
        .def("copy_last_dim", [](PyCtx& c, py::array a, size_t width, int from, bool padding_zero) {
            Tensor in = c.up(a);
            std::vector<size_t> shape = in.shape();
            shape.back() = width;
            Tensor out = c.ctx->tensor(shape, in.dtype());
            bmengine::functions::copy_last_dim(c.ctx->current_cuda_stream(), in, out, from, -1, padding_zero);
            return c.down(out, "float16");
        })
        .def("concat_broadcast_b", [](PyCtx& c, py::array a, py::array b) {
            return c.down(bmengine::functions::concat_broadcast_b(*c.ctx, c.up(a), c.up(b)), "float16");
        })
        // ---- int8
        .def("quant_calc_scale", [](PyCtx& c, py::array x) {
            Tensor q = int8_op::quant_calc_scale(*c.ctx, c.up(x));
            BM_ASSERT(q.quant_scale, "quant_scale side tensor missing");
            return py::make_tuple(c.down(q, "int8"), c.down(*q.quant_scale, "float32"));
        })
        .def("layernorm_quant", [](PyCtx& c, py::array x, py::array w, float eps) {
            Tensor tx = c.up(x);
            Tensor out = c.ctx->tensor(tx.shape(), tx.dtype()), q = c.ctx->tensor(tx.shape(), DataType::kInt8),
                   sc = c.ctx->tensor({tx.numel() / tx.size(-1)}, DataType::kFloat);
            int8_op::layernorm_quant(*c.ctx, tx, c.up(w), &out, &q, &sc, eps, 1.0f);
            return py::make_tuple(c.down(out, "float16"), c.down(q, "int8"), c.down(sc, "float32"));
        })
        .def("int8_linear", [](PyCtx& c, py::array x, py::array w_q, py::array w_scale) {
            // Int8Linear::forward (linear.cpp:557-635): quantise the rows, int8 x int8^T -> int32, scale back
            Tensor xq = int8_op::quant_calc_scale(*c.ctx, c.up(x));
            Tensor acc = int8_op::int8_gemm_nt(*c.ctx, xq, c.up(w_q));
            Tensor ws = c.up(w_scale);
            Tensor y = int8_op::quant_scale_back(*c.ctx, acc, xq.quant_scale.get(), &ws);
            return py::make_tuple(c.down(y, "float16"), c.down(acc, "int32"));
        })
        .def("quant_back_act_mul", [](PyCtx& c, py::array a, py::array asx, py::array asy, py::array b, py::array bsx, py::array bsy,
                                       std::string act) {
            Tensor tasx = c.up(asx), tasy = c.up(asy), tbsx = c.up(bsx), tbsy = c.up(bsy);
            return c.down(int8_op::quant_back_act_mul(*c.ctx, c.up(a), &tasx, &tasy, c.up(b), &tbsx, &tbsy, act), "float16");
        })
        .def("raises_on_bad_shape", [](PyCtx& c) {
            Tensor a = c.ctx->tensor({1, 64}, DataType::kHalf), q = c.ctx->tensor({16, 16}, DataType::kInt32),
                   z = c.ctx->tensor({16, 1}, DataType::kInt8), s = c.ctx->tensor({16, 1}, DataType::kHalf);
            nn::gptq::gptq_gemm_k_major(*c.ctx, a, q, z, s, Tensor(), Tensor(), nullptr, false);   // K = 128 vs 64: throws
        });
}
