// refshim: the CUDA toolkit names the reference's HOST translation units spell, as aliases of the HIP runtime's.  This directory is
// on the include path ONLY when a unit of /root/reference is compiled (zhilight_amd/build.py: the host library's reference units,
// the zhilight.C binding, the link check); no source file of this repository includes it.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
typedef hipStream_t cudaStream_t;
typedef hipEvent_t cudaEvent_t;
typedef hipError_t cudaError_t;
typedef hipDeviceProp_t cudaDeviceProp;
typedef __half half;
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaMemcpyAsync hipMemcpyAsync
#define cudaMemcpy hipMemcpy
#define cudaMemsetAsync hipMemsetAsync
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaStreamSynchronize hipStreamSynchronize
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaEventCreate hipEventCreate
// an event another stream will wait on must see everything the host library is still holding back on this thread (bm_hip.h DeferredOp:
// results handed back unlaunched): the deferred launches go out, on their streams, before the event is recorded
extern "C" hipError_t zl_shim_event_record(hipEvent_t event, hipStream_t stream);
#define cudaEventRecord zl_shim_event_record
#define cudaEventDestroy hipEventDestroy
#define cudaEventSynchronize hipEventSynchronize
#define cudaEventElapsedTime hipEventElapsedTime
#define cudaStreamWaitEvent hipStreamWaitEvent
#define cudaHostAlloc hipHostAlloc
#define cudaFreeHost hipFreeHost
#define cudaHostAllocDefault hipHostMallocDefault
#define cudaHostAllocPortable hipHostMallocPortable
// block.cpp's dual_stream_encode creates its reduce stream with this call (block.cpp:221-224).  A passthrough to the HIP runtime --
// except on a rank thread of an Engine whose ranks SHARE a device (the one-GPU test mode, hostcpp/bm_engine.cpp), where the stream
// must not share a hardware queue with any other rank's stream: see zl_shim_stream_create_with_priority in hostcpp/bm_hip.cpp.
extern "C" hipError_t zl_shim_stream_create_with_priority(hipStream_t* stream, unsigned int flags, int priority);
extern "C" hipError_t zl_shim_stream_destroy(hipStream_t stream);
#define cudaStreamCreateWithPriority zl_shim_stream_create_with_priority
#define cudaStreamNonBlocking hipStreamNonBlocking
#define cudaStreamDestroy zl_shim_stream_destroy
#define cudaEventCreateWithFlags hipEventCreateWithFlags
#define cudaEventDefault hipEventDefault
#define cudaEventQuery hipEventQuery
#define cudaErrorNotReady hipErrorNotReady
#define cudaMemcpyPeer hipMemcpyPeer
#define cudaGetDevice hipGetDevice
#define cudaSetDevice hipSetDevice
#define cudaEventDisableTiming hipEventDisableTiming
#define cudaMallocHost hipHostMalloc
#define cudaGetDeviceCount hipGetDeviceCount
#define cudaMemGetInfo hipMemGetInfo
