// refshim: the CUDA toolkit names the reference's HOST translation units spell, as aliases of the HIP runtime's.  Used only
// by the compile-the-reference check (hostcpp/refcompile.py); nothing in the product includes this directory.
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
#define cudaEventRecord hipEventRecord
#define cudaEventDestroy hipEventDestroy
#define cudaEventSynchronize hipEventSynchronize
#define cudaEventElapsedTime hipEventElapsedTime
#define cudaStreamWaitEvent hipStreamWaitEvent
#define cudaHostAlloc hipHostAlloc
#define cudaFreeHost hipFreeHost
#define cudaHostAllocDefault hipHostMallocDefault
#define cudaHostAllocPortable hipHostMallocPortable
#define cudaStreamCreateWithPriority hipStreamCreateWithPriority
#define cudaStreamNonBlocking hipStreamNonBlocking
#define cudaStreamDestroy hipStreamDestroy
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
