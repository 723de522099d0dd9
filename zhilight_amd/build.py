"""Build the gfx950 shared library (hipcc cross-compiles without a GPU).

    python -m zhilight_amd.build [--force]

Produces zhilight_amd/libzhilight_amd.so in-tree (git-ignored, but it travels with gpurun snapshots).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libzhilight_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-fast-math", "-ffp-contract=off"]
# per-source extras.  w4_slab.hip: its item loops (up to 4 groups x 8 tiles, the next group's normalisation in slices between the
# items) must unroll completely -- ring slots, accumulators and wait counts are static only then -- and the NORM instantiations
# exceed the unroller's default budget for `#pragma unroll` (16 K): the ring went to scratch behind dynamic indices
EXTRA_FLAGS = {"w4_slab.hip": ["-mllvm", "-pragma-unroll-threshold=200000"]}


# ZL_BUILD_EXPERIMENTAL=1: also build what was measured and left off the product path: the loader /
# consumer engine (tools/experimental/w4_engine.hip -- it is not in csrc/) and the digit-plane entry points for 5..32 rows (the
# #ifdef ZL_EXPERIMENTAL blocks of w4_phase.hip / w4_mfma.hip); their tests live next to the engine, tools/experimental/test_gpu_*.py
# (python -m pytest tools/experimental -m gpu, after such a build).  The default library and the default test run carry neither.
EXPERIMENTAL = os.environ.get("ZL_BUILD_EXPERIMENTAL", "0") == "1"
EXPERIMENTAL_DIR = os.path.join(HERE, "..", "tools", "experimental")


def sources():
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    if EXPERIMENTAL:
        srcs.append(os.path.join(EXPERIMENTAL_DIR, "w4_engine.hip"))
    return srcs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    srcs = sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "zhilight_amd.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = FLAGS + (["-DZL_EXPERIMENTAL", "-I" + CSRC] if EXPERIMENTAL else [])
    stamp = os.path.join(objdir, ".flags")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True                                   # another flavour's objects: rebuild all of them
    objs = []
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + flags + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(OUT, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    with open(stamp, "w") as fh:
        fh.write(" ".join(flags))
    build_hostcpp(force, verbose)
    build_comm(force, verbose)
    build_refcompile(force, verbose)
    build_binding(force, verbose)
    build_refcheck(force, verbose)
    return OUT


COMM_OUT = os.path.join(HERE, "libzhilight_amd_comm.so")


def build_comm(force=False, verbose=False):
    """libzhilight_amd_comm.so: the tensor-parallel exchange step (direct RCCL communicator + one-shot peer-read all-reduce,
    include/zhilight_amd_comm.h).  Separate from the main library so that single-GPU users carry no RCCL dependency."""
    src = os.path.join(HERE, "csrc_comm", "comm.hip")
    deps = [src, os.path.join(HERE, "..", "include", "zhilight_amd_comm.h"), os.path.join(HERE, "..", "include", "zhilight_amd.h")]
    if not (force or _stale(COMM_OUT, deps)):
        return COMM_OUT
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-shared", "-o", COMM_OUT, src,
           "-L" + os.path.join(rocm, "lib"), "-lrccl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return COMM_OUT


HOSTCPP = os.path.join(HERE, "hostcpp")


HOSTCPP_SOURCES = ("bm_hip.cpp", "bm_layer.cpp", "bm_functions.cpp", "bm_c10d.cpp", "nn_amd.cpp")
HOSTCPP_HEADERS = ("bm_hip.h", "bm_layer.h", "bm_functions.h", "bm_c10d.h", "nn_amd.h")


def hostcpp_target():
    import sysconfig
    return os.path.join(HERE, "zl_internals" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_hostcpp(force=False, verbose=False):
    """The C++ host layer (bmengine-on-HIP shim + the reference's nn:: / int8_op:: operator names over the C ABI) and its
    pybind11 test module zhilight_amd/zl_internals*.so.  Host code only: g++ against the HIP runtime headers."""
    import pybind11
    import sysconfig
    srcs = [os.path.join(HOSTCPP, f) for f in HOSTCPP_SOURCES + ("py_internals.cpp",)]
    deps = srcs + [os.path.join(HOSTCPP, f) for f in HOSTCPP_HEADERS] + [os.path.join(HERE, "..", "include", "zhilight_amd.h")]
    target = hostcpp_target()
    if not (force or _stale(target, deps)):
        return target
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"),
           "-I" + os.path.join(HERE, "..", "include"), "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"]] + srcs + [
           "-o", target, "-L" + HERE, "-lzhilight_amd", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return target


REFERENCE = os.environ.get("ZL_REFERENCE_ROOT", "/root/reference")
REFDIR = os.path.join(HERE, "_ref")
# reference host translation units compiled UNMODIFIED, from where they lie, against hostcpp/refshim + the bmengine-on-HIP
# headers ("prove the boundary compiles the reference")
REF_TUS = ("src/nn/linear/linear.cpp", "src/nn/attention/attention.cpp", "src/nn/attention/multi_head_latent_attention.cpp",
           "src/nn/feedforward/feedforward.cpp", "src/nn/block/block.cpp", "src/model/llama.cpp", "src/model/model_context.cpp",
           "src/model/buffer_context.cpp", "src/kvcache/block_allocator.cpp", "src/model/host_all_reducer.cpp")
# reference translation units that are compiled unmodified and LINK-CHECKED only (build_refcheck): every name they reference in the
# namespaces the boundary stands in for must be defined by the boundary under the same mangled name -- i.e. with the reference's
# exact signature; names of layers that are not on the path (their device code lives in the reference's .cu files) are listed
REF_CHECK_TUS = ("src/nn/attention/attention.cpp", "src/nn/attention/multi_head_latent_attention.cpp", "src/nn/feedforward/feedforward.cpp",
                 "src/nn/block/block.cpp", "src/model/llama.cpp", "src/model/model_context.cpp", "src/model/buffer_context.cpp")
REF_CHECK_NAMESPACES = ("ds::", "bmengine::", "nn::fp8::", "nn::gptq::", "int8_op::", "nn::top_k_softmax(", "nn::group_topk_softmax(",
                        "nn::sum_experts(", "nn::route_shared_lb(", "nn::plus_for_sort(", "nn::calc_reverse_idx(",
                        "nn::fill_m_indices_padded_indices(", "nn::gate_mul_inplace(", "nn::gate_fuse(", "nn::gelu_inplace(", "nn::silu_inplace(",
                        "nn::attention_qkv_rag_buffer(", "nn::multi_query_attention_rag_buffer(", "nn::get_mqa_workspace(", "nn::rope_qk_cache(",
                        "nn::rotary_embedding_qk(", "nn::copy_to_rag_buffer2(")
# attempted and REPORTED only (never fail the build): what a full drop-in of zhilight.C would still need
REF_REPORT_TUS = ("src/py_export/bind.cpp", "src/py_export/py_batch_generator.cpp", "src/py_export/py_llama.cpp", "src/py_export/py_model_base.cpp",
                  "src/py_export/py_model_config.cpp", "src/py_export/py_utils.cpp", "src/generator/batch_generator.cpp")
# declared by the shim so that the units compile, NOT provided by the boundary (smooth-quant calibration helpers): reported as "pending"
# (round 4: the MoE dispatch route's arange / sort_pair_1d / divide / scatter_update_dim0 left this list -- bm_functions.cpp)
REF_CHECK_PENDING = ("bmengine::functions::pow(", "bmengine::functions::clamp(")


# the host library: the bmengine-on-HIP layer, the engine, the classes around the operators (host_*.cpp) and the reference's units
HOST_SOURCES = HOSTCPP_SOURCES + ("bm_engine.cpp", "host_kvcache.cpp", "host_position.cpp", "host_embedding.cpp", "host_layernorm.cpp",
                                  "host_attention_ext.cpp", "host_offpath.cpp", "host_generator_ext.cpp")
HOST_HEADERS = HOSTCPP_HEADERS + ("bm_engine.h", "host_common.h")
# the pybind11 harness around it (test infrastructure)
HARNESS_SOURCES = (os.path.join("refshim", "ref_glue.cpp"), "ref_attention_glue.cpp", "ref_block_glue.cpp", "ref_model_glue.cpp")


def host_target():
    return os.path.join(REFDIR, "libzhilight_amd_host.so")


def refcompile_target():
    import sysconfig
    return os.path.join(REFDIR, "zl_reflinear" + sysconfig.get_config_var("EXT_SUFFIX"))


def _ref_compile_env():
    import pybind11
    import sysconfig
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    shim = os.path.join(HOSTCPP, "refshim")
    inc = ["-I" + shim, "-I" + HOSTCPP, "-I" + os.path.join(rocm, "include"), "-I" + os.path.join(HERE, "..", "include"),
           "-I" + os.path.join(REFERENCE, "src"), "-I" + REFERENCE, "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"]]
    cxx = os.environ.get("CXX", "g++")
    return rocm, shim, cxx, [cxx, "-O1", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-DENABLE_DS_DEEP_GEMM", "-DZL_REF_LAYERNORM_EXTERNAL", "-w"] + inc


def _undefined_outside(target, libs):
    """names `target` needs that none of `libs` defines (the CPython API and the C / C++ runtime aside)"""
    have = set()
    for lib in libs:
        for line in subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True).splitlines():
            have.add(line.split()[-1])
    missing = []
    for line in subprocess.check_output(["nm", "-D", "-u", target], text=True).splitlines():
        kind, sym = line.split()[-2:]
        if kind in "wv" or "@" in sym or sym.startswith(("Py", "_Py", "__")) or sym in have:
            continue
        missing.append(sym)
    return missing


def build_host(force=False, verbose=False):
    """zhilight_amd/_ref/libzhilight_amd_host.so = the host library a drop-in of the reference links instead of its .cu files and
    bmengine: the bmengine-on-HIP layer (Context / Tensor / Layer / functions / c10d), core::Engine (one thread + one exchange state +
    one RCCL communicator per tensor-parallel rank, bm_engine.cpp), the operator wrappers under the reference's names (nn_amd.cpp), the
    classes around them (host_*.cpp: TransformerBuffer, RotaryEmbedding, RopePreparer, RawEmbedding incl. its vocab-parallel form,
    the reference-layout LayerNorm, FlashDecoding::mha_fwd) -- and, compiled UNMODIFIED from where they lie in /root/reference,
    the reference's own host translation units REF_TUS (linear, attention, MLA, feed-forward, block, llama, model_context,
    buffer_context).  Links libzhilight_amd.so and libzhilight_amd_comm.so.  Only where the reference tree exists (this
    container); the GPU box uses the prebuilt file.  _ref/ is git-ignored (it holds compiled reference code) but travels with
    gpurun snapshots.  Returns the path, or None without a reference."""
    target = host_target()
    if target in _HOST_BUILT:                           # once per process: build_refcompile and build_binding both ask (a
        return _HOST_BUILT[target]                      # forced build compiled every reference unit twice)
    _HOST_BUILT[target] = _build_host(force, verbose)
    return _HOST_BUILT[target]


_HOST_BUILT = {}


def _build_host(force, verbose):
    target = host_target()
    tus = [os.path.join(REFERENCE, t) for t in REF_TUS]
    if not all(os.path.exists(t) for t in tus):
        return target if os.path.exists(target) else None
    rocm, shim, cxx, common = _ref_compile_env()
    own = [os.path.join(HOSTCPP, f) for f in HOST_SOURCES]
    deps = tus + own + [os.path.join(HOSTCPP, f) for f in HOST_HEADERS] + [os.path.join(HERE, "..", "include", "zhilight_amd.h"),
                                                                            os.path.join(HERE, "..", "include", "zhilight_amd_comm.h"), OUT, COMM_OUT]
    for root, _, files in os.walk(shim):
        deps += [os.path.join(root, f) for f in files if f != "ref_glue.cpp"]
    if not (force or _stale(target, deps)):
        return target
    os.makedirs(REFDIR, exist_ok=True)
    objs = []
    jobs = []
    for src in tus + own:
        o = os.path.join(REFDIR, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        objs.append(o)
        jobs.append(common + ["-c", src, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(16, len(jobs))) as ex:
        list(ex.map(run, jobs))
    run([cxx, "-shared", "-o", target] + objs + ["-L" + HERE, "-lzhilight_amd", "-lzhilight_amd_comm", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lpthread",
                                                   "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + os.path.join(rocm, "lib")])
    for o in objs:
        os.remove(o)
    # every name the reference's translation units need must be DEFINED by the boundary: a leftover undefined symbol would only
    # surface as an ImportError on the GPU box
    missing = _undefined_outside(target, [OUT, COMM_OUT, os.path.join(rocm, "lib", "libamdhip64.so")])
    if missing:
        os.remove(target)
        demangled = subprocess.run(["c++filt"], input="\n".join(missing), text=True, capture_output=True).stdout
        raise RuntimeError("the reference translation units need names the boundary does not define:\n" + demangled)
    return target


def build_refcompile(force=False, verbose=False):
    """zhilight_amd/_ref/zl_reflinear*.so = the pybind11 TEST module around libzhilight_amd_host.so (HARNESS_SOURCES: RefLinear,
    RefAttention, RefEncoderLayer, RefFeedForward, RefLLaMA, RefEngineLLaMA).  Returns the path, or None without a reference."""
    target = refcompile_target()
    host = build_host(force, verbose)
    if host is None or not os.path.exists(REFERENCE):
        return target if os.path.exists(target) and host is not None else None
    rocm, shim, cxx, common = _ref_compile_env()
    srcs = [os.path.join(HOSTCPP, f) for f in HARNESS_SOURCES]
    if not (force or _stale(target, srcs + [host])):
        return target
    objs = []
    jobs = []
    for src in srcs:
        o = os.path.join(REFDIR, "harness_" + os.path.basename(src).rsplit(".", 1)[0] + ".o")
        objs.append(o)
        jobs.append(common + ["-c", src, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        list(ex.map(run, jobs))
    run([cxx, "-shared", "-o", target] + objs + ["-L" + REFDIR, "-lzhilight_amd_host", "-L" + HERE, "-lzhilight_amd", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
                                                   "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + os.path.join(rocm, "lib")])
    for o in objs:
        os.remove(o)
    missing = _undefined_outside(target, [host, OUT, os.path.join(rocm, "lib", "libamdhip64.so")])
    if missing:
        os.remove(target)
        demangled = subprocess.run(["c++filt"], input="\n".join(missing), text=True, capture_output=True).stdout
        raise RuntimeError("the test module needs names the host library does not define:\n" + demangled)
    return target


# the reference's Python binding and dynamic-batch scheduler, compiled UNMODIFIED: together with libzhilight_amd_host.so they ARE zhilight.C
REF_BINDING_TUS = ("src/py_export/bind.cpp", "src/py_export/py_batch_generator.cpp", "src/py_export/py_llama.cpp", "src/py_export/py_model_base.cpp",
                   "src/py_export/py_model_config.cpp", "src/py_export/py_utils.cpp", "src/generator/batch_generator.cpp")


def binding_target():
    import sysconfig
    return os.path.join(REFDIR, "C" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_binding(force=False, verbose=False):
    """zhilight_amd/_ref/C*.so = the reference's own `zhilight.C` extension module: src/py_export/*.cpp (the pybind11 surface zhilight.llama /
    zhilight.dynamic_batch import) and src/generator/batch_generator.cpp (the dynamic-batch scheduler: task queue, chunked prefill, beam /
    greedy / random search), every unit compiled unmodified from where it lies, linked against libzhilight_amd_host.so -- no unit of this
    repository is compiled into it.  Fails on a leftover undefined symbol.  Returns the path, or None without a reference."""
    target = binding_target()
    host = build_host(force, verbose)
    tus = [os.path.join(REFERENCE, t) for t in REF_BINDING_TUS]
    if host is None or not all(os.path.exists(t) for t in tus):
        return target if os.path.exists(target) and host is not None else None
    rocm, shim, cxx, common = _ref_compile_env()
    deps = tus + [host]
    for root, _, files in os.walk(shim):
        deps += [os.path.join(root, f) for f in files if f != "ref_glue.cpp"]
    if not (force or _stale(target, deps)):
        return target
    objs, jobs = [], []
    for src in tus:
        o = os.path.join(REFDIR, "binding_" + os.path.basename(src).rsplit(".", 1)[0] + ".o")
        objs.append(o)
        jobs.append(common + ["-c", src, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        list(ex.map(run, jobs))
    run([cxx, "-shared", "-o", target] + objs + ["-L" + REFDIR, "-lzhilight_amd_host", "-L" + HERE, "-lzhilight_amd", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
                                                   "-lpthread", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + os.path.join(rocm, "lib")])
    for o in objs:
        os.remove(o)
    missing = _undefined_outside(target, [host, OUT, os.path.join(rocm, "lib", "libamdhip64.so")])
    if missing:
        os.remove(target)
        demangled = subprocess.run(["c++filt"], input="\n".join(missing), text=True, capture_output=True).stdout
        raise RuntimeError("zhilight.C needs names the host library does not define:\n" + demangled)
    return target


def refcheck_report():
    return os.path.join(REFDIR, "linkcheck.json")


def build_refcheck(force=False, verbose=False):
    """Compile REF_CHECK_TUS in place against hostcpp/refshim and compare what they reference with what the boundary defines
    (libzhilight_amd.so + libzhilight_amd_host.so).  Writes zhilight_amd/_ref/linkcheck.json:
    {tu: {"resolved": [...], "outside": [...], "pending": [...], "reference": [... defined by another checked unit]}} (demangled); raises when a name in REF_CHECK_NAMESPACES is not
    defined -- a signature that drifted from the reference's -- unless it is one of REF_CHECK_PENDING.  Only where the reference tree exists; returns the report path or None."""
    import json
    import pybind11
    import sysconfig
    report = refcheck_report()
    tus = [os.path.join(REFERENCE, t) for t in REF_CHECK_TUS]
    module = build_refcompile(verbose=verbose)
    if module is None or not all(os.path.exists(t) for t in tus):
        return report if os.path.exists(report) else None
    shim = os.path.join(HOSTCPP, "refshim")
    deps = tus + [module, os.path.abspath(__file__)]
    for root, _, files in os.walk(shim):
        deps += [os.path.join(root, f) for f in files]
    if not (force or _stale(report, deps)):
        return report
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = ["-I" + shim, "-I" + HOSTCPP, "-I" + os.path.join(rocm, "include"), "-I" + os.path.join(HERE, "..", "include"),
           "-I" + os.path.join(REFERENCE, "src"), "-I" + REFERENCE, "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"]]
    common = [os.environ.get("CXX", "g++"), "-O1", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-DENABLE_DS_DEEP_GEMM", "-w"] + inc
    have = set()
    for lib in (OUT, host_target(), os.path.join(rocm, "lib", "libamdhip64.so")):
        for line in subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True).splitlines():
            have.add(line.split()[-1])
    out, drifted = {}, []
    objs = [os.path.join(REFDIR, os.path.basename(src).rsplit(".", 1)[0] + ".check.o") for src in tus]

    def compile_one(job):
        src, obj = job
        cmd = common + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=len(tus)) as ex:
        list(ex.map(compile_one, zip(tus, objs)))
    undefined, other_units = {}, set()               # names one checked unit needs and another one defines are the reference's own
    for obj in objs:
        undefined[obj] = [line.split()[-1] for line in subprocess.check_output(["nm", "-u", obj], text=True).splitlines()]
        for line in subprocess.check_output(["nm", "--defined-only", obj], text=True).splitlines():
            other_units.add(line.split()[-1])
        os.remove(obj)
    for rel, obj in zip(REF_CHECK_TUS, objs):
        syms = undefined[obj]
        names = subprocess.run(["c++filt"], input="\n".join(syms), text=True, capture_output=True).stdout.splitlines()
        resolved, outside, pending, reference = [], [], [], []
        for sym, name in zip(syms, names):
            if sym == name:                                       # C symbols: libc / libm / the HIP runtime (versioned there)
                continue
            if name.startswith(("std::", "operator ", "vtable for", "typeinfo for", "VTT for", "__")):
                continue
            if name.startswith(REF_CHECK_PENDING):                # (the executed module defines them as stubs that throw: still pending)
                pending.append(name)
                continue
            if sym not in have and sym in other_units:
                reference.append(name)
                continue
            (resolved if sym in have else outside).append(name)
        drifted += [n for n in outside if n.startswith(REF_CHECK_NAMESPACES)]
        out[rel] = {"resolved": sorted(resolved), "outside": sorted(outside), "pending": sorted(pending), "reference": sorted(reference)}
    if drifted:
        raise RuntimeError("reference call sites name boundary functions the boundary does not define with that signature:\n" + "\n".join(drifted))
    # REPORT ONLY (VERDICT r04 missing 5 / item 9): the units above the model -- the Python binding, the batch generator, the model
    # context with its engine -- attempted against the same shim.  Nothing here fails the build: the report says how far each gets.
    report_objs, report_defined = {}, set()
    for rel in REF_REPORT_TUS:
        src = os.path.join(REFERENCE, rel)
        if not os.path.exists(src):
            continue
        obj = os.path.join(REFDIR, os.path.basename(src).rsplit(".", 1)[0] + ".report.o")
        r = subprocess.run(common + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            errs = [ln for ln in r.stderr.splitlines() if " error" in ln or "fatal" in ln]
            out[rel] = {"report_only": True, "compiles": False, "errors": len(errs),
                        "first_errors": [e.replace(REFERENCE + "/", "").replace(HERE + "/", "") for e in errs[:6]]}
            continue
        report_objs[rel] = obj
        for line in subprocess.check_output(["nm", "--defined-only", obj], text=True).splitlines():
            report_defined.add(line.split()[-1])
    for rel, obj in report_objs.items():
        syms = [line.split()[-1] for line in subprocess.check_output(["nm", "-u", obj], text=True).splitlines()]
        os.remove(obj)
        names = subprocess.run(["c++filt"], input="\n".join(syms), text=True, capture_output=True).stdout.splitlines()
        res = [n for sy, n in zip(syms, names) if sy != n and sy in have]
        own = [n for sy, n in zip(syms, names) if sy != n and sy not in have and sy in report_defined]     # defined by another unit of the binding
        outside = [n for sy, n in zip(syms, names) if sy != n and sy not in have and sy not in report_defined
                   and not n.startswith(("std::", "operator ", "vtable for", "typeinfo for", "VTT for", "__", "pybind11::", "Py"))]
        out[rel] = {"report_only": True, "compiles": True, "resolved": sorted(res), "reference": sorted(own), "outside": sorted(outside)}
    with open(report, "w") as f:
        json.dump(out, f, indent=1)
    return report


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
