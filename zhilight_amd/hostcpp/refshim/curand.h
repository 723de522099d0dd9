// refshim: curand.h -- the generator's headers name curandGenerator_t and the sampling set-up calls (src/generator/generator.h:9,
// batch_generator.cpp:189-190); the sampling kernels behind them are off the MI355X hot-path boundary (beam
// search / sampling stay the scheduler's).  The handle type is bm_layer.h's opaque one; the functions are DECLARED so that the host
// translation units compile for the report-only link check (zhilight_amd/build.py: REF_REPORT_TUS) and show up there as names the
// boundary does not provide.
#pragma once
#include "bm_layer.h"
typedef enum { CURAND_STATUS_SUCCESS = 0 } curandStatus_t;
typedef enum { CURAND_ORDERING_PSEUDO_BEST = 100, CURAND_ORDERING_PSEUDO_DEFAULT = 101 } curandOrdering_t;
typedef enum { CURAND_RNG_PSEUDO_DEFAULT = 100, CURAND_RNG_PSEUDO_MRG32K3A = 121 } curandRngType_t;
extern "C" {
curandStatus_t curandCreateGenerator(curandGenerator_t* generator, curandRngType_t rng_type);
curandStatus_t curandDestroyGenerator(curandGenerator_t generator);
curandStatus_t curandSetStream(curandGenerator_t generator, hipStream_t stream);
curandStatus_t curandSetPseudoRandomGeneratorSeed(curandGenerator_t generator, unsigned long long seed);
curandStatus_t curandGenerateUniform(curandGenerator_t generator, float* out, size_t n);
curandStatus_t curandSetGeneratorOrdering(curandGenerator_t generator, curandOrdering_t order);
curandStatus_t curandSetGeneratorOffset(curandGenerator_t generator, unsigned long long offset);
}
#define CURAND_CHECK(err)                                                                                                  \
    do {                                                                                                                   \
        curandStatus_t err_ = (err);                                                                                       \
        if (err_ != CURAND_STATUS_SUCCESS) throw BMEngineException("Exception:\n", __FILE__, __LINE__, __PRETTY_FUNCTION__, "curand error"); \
    } while (0)
