// host_common.h -- shared by the host_*.cpp units of the host library (libzhilight_amd_host.so, zhilight_amd/build.py: build_host):
// the classes AROUND the operators that the reference keeps in its .cu / scheduler files, re-built on the C ABI so that the
// reference's own host translation units (linear.cpp, attention.cpp, multi_head_latent_attention.cpp, feedforward.cpp, block.cpp,
// llama.cpp, model_context.cpp, buffer_context.cpp -- compiled unmodified) link and run on MI355X.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "bm_hip.h"
#include "zhilight_amd.h"

#define ZL_OFF_PATH(what) \
    throw BMEngineException(std::string(what) + " is not on the decode path this library runs (SURVEY.md section 8)", __FILE__, __LINE__, __func__)
#define ZL_CK(call, what)                                                                                        \
    do {                                                                                                         \
        const int st_ = (call);                                                                                  \
        if (st_ != 0) throw BMEngineException(std::string(what) + ": " + zl_status_string(st_), __FILE__, __LINE__, __func__); \
    } while (0)

using bmengine::core::Context;
using bmengine::core::DataType;
using bmengine::core::Tensor;
