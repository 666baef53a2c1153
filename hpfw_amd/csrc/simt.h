// simt.h -- the few macros that let the bodies of the in-LDS transform kernels be written once.
//
// In the product (hipcc, gfx950) HPFW_FOR_THREADS runs its body once for threadIdx.x and
// HPFW_BARRIER is __syncthreads().  tests/emu/ compiles the same bodies with g++ and
// -DHPFW_SIMT_EMU, where HPFW_FOR_THREADS loops over all threads of the workgroup and the LDS
// accessor is bounds-checked: a CPU-side check of the index arithmetic (no GPU in the build
// container) that is test infrastructure only -- libhpfw_gpu.so contains no host evaluation of
// any kernel and has no CPU fallback.
#pragma once

#if defined(HPFW_SIMT_EMU)
#include <cassert>
#include <cmath>
#include <cstdint>
#define HPFW_DEVICE static inline
#define HPFW_DEVICE_STATIC static inline
#define HPFW_DEVICE_MEMBER inline
#define HPFW_FOR_THREADS(tid, nt) for (int tid = 0; tid < (nt); ++tid)
#define HPFW_BARRIER() ((void)0)
// registers a thread keeps across a barrier: one array per emulated thread
#include <vector>
#define HPFW_CARRY(type, name, count, nt) std::vector<type> name##_store((size_t)(count) * (size_t)(nt))
#define HPFW_CARRY_AT(name, count, tid) (name##_store.data() + (size_t)(count) * (size_t)(tid))
#else
#include <hip/hip_runtime.h>
#define HPFW_DEVICE __device__ __forceinline__
#define HPFW_DEVICE_STATIC static __device__ __forceinline__
#define HPFW_DEVICE_MEMBER __device__ __forceinline__
#define HPFW_FOR_THREADS(tid, nt) for (int tid = threadIdx.x, hpfw_once_ = 1; hpfw_once_; hpfw_once_ = 0)
#define HPFW_BARRIER() __syncthreads()
#define HPFW_CARRY(type, name, count, nt) type name##_store[count]
#define HPFW_CARRY_AT(name, count, tid) (name##_store)
#endif

#define HPFW_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
