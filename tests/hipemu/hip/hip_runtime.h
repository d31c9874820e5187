// TEST INFRASTRUCTURE ONLY -- a single-threaded CPU interpreter for the HIP kernel sources.
//
// The build container has no GPU, so `tests/hipemu` compiles owshen_amd/csrc/*.hip with g++
// against this header: every kernel launch runs its grid block by block, every lane of a
// block is a ucontext fiber and `__syncthreads()` yields to a round-robin scheduler.  It
// exists so kernel LOGIC (indexing, digit decomposition, reduction trees, prover glue) can
// be checked against the oracle in the `-m "not gpu"` suite before GPU minutes are spent.
// owshen_amd never loads the resulting library; the product is the gfx950 binary only and
// the `-m gpu` parity tests run that binary.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <functional>
#include <cstdio>

#define OG_HIPEMU 1
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return {x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return {x, y}; }

namespace hipemu {
struct Idx { unsigned x, y, z; };
extern Idx threadIdx_, blockIdx_;
extern dim3 blockDim_, gridDim_;
extern void* dyn_shared;
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body, const char* name = nullptr);
void sync_threads();
}  // namespace hipemu
#define threadIdx hipemu::threadIdx_
#define blockIdx hipemu::blockIdx_
#define blockDim hipemu::blockDim_
#define gridDim hipemu::gridDim_
#define __syncthreads() hipemu::sync_threads()
// dynamic LDS: a heap block per launch (owshen_amd/csrc/ctx.h lets the runtime header supply this)
#define OG_DYN_LDS(name) uint8_t* name = (uint8_t*)hipemu::dyn_shared
#define OG_FILLER_PRIO() ((void)0)  // wave priority: nothing to interpret
#define OG_CLAIM_VGPR(n) ((void)0)  // register allocation: nothing to interpret
#define OG_PAIR_SWAP32(x) ((uint32_t)__shfl_xor((int)(x), 1))  // the DPP lane-pair swap (ctx.h), as a rendezvous
#define OG_SHADER_CYCLES() 0ull  // no shader clock to read
#define OG_LDS_ATOMIC_INC_AGG(arr, key) atomicAdd(&(arr)[key], 1u)  // wave-aggregated LDS increment (ctx.h): lanes run one after the other here
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  hipemu::launch(dim3(grid), dim3(block), (shmem), [&]() { kern(__VA_ARGS__); }, #kern)

// ---- device intrinsics ---------------------------------------------------------
static inline unsigned long long __brevll(unsigned long long v) {
  unsigned long long r = 0;
  for (int i = 0; i < 64; i++) r |= ((v >> i) & 1ull) << (63 - i);
  return r;
}
static inline unsigned __brev(unsigned v) { return (unsigned)(__brevll(v) >> 32); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
static inline void __threadfence() {}
// lane exchange: every lane of the block parks its value, yields, and reads its partner's (the fibers of a block run one after
// the other, so this is a block-wide rendezvous: all live lanes must call it the same number of times -- as on the GPU)
namespace hipemu { extern int shfl_buf[1024]; }
static inline int __shfl_xor(int v, int lane_mask) {
  const unsigned me = hipemu::threadIdx_.x;
  hipemu::shfl_buf[me] = v;
  hipemu::sync_threads();
  const int r = hipemu::shfl_buf[(me ^ (unsigned)lane_mask) & 1023];
  hipemu::sync_threads();
  return r;
}

// the cross-lane reads of the wave-wide field form (csrc/field_w9.hip.h), as rendezvous over the 64-lane wave of the caller:
// v_readlane_b32 / v_readfirstlane_b32, the DPP row shifts (a lane without a source in its row of 16 reads 0), the DPP row
// broadcast of a row's first lane, and rows 0 | 1 (2 | 3) both reading row 1 (3)
static inline uint32_t hipemu_w9_read(uint32_t v, int how, int lane) {
  const unsigned me = hipemu::threadIdx_.x;
  hipemu::shfl_buf[me] = (int)v;
  hipemu::sync_threads();
  uint32_t r = 0;
  if (how == 0) r = (uint32_t)hipemu::shfl_buf[(me & ~63u) + (unsigned)lane];
  else if (how == 1) r = (me & 15u) == 15u ? 0u : (uint32_t)hipemu::shfl_buf[me + 1];
  else if (how == 2) r = (me & 15u) == 0u ? 0u : (uint32_t)hipemu::shfl_buf[me - 1];
  else if (how == 3) r = (uint32_t)hipemu::shfl_buf[me & ~15u];                           // row_newbcast:0
  else if (how == 4) r = (uint32_t)hipemu::shfl_buf[(me & ~31u) | 16u | (me & 15u)];      // v_permlane16_swap_b32 x, copy(x): the copy
  else r = (uint32_t)hipemu::shfl_buf[(me & ~31u) | (me & 15u)];                          // ... and x itself: the even row of the pair
  hipemu::sync_threads();
  return r;
}
#define OG_W9_READLANE(x, i) hipemu_w9_read((x), 0, (i))
#define OG_W9_FIRST(x) hipemu_w9_read((x), 0, 0)
#define OG_W9_FROM_NEXT(x) hipemu_w9_read((x), 1, 0)
#define OG_W9_FROM_PREV(x) hipemu_w9_read((x), 2, 0)
#define OG_W9_ROWFIRST(x) hipemu_w9_read((x), 3, 0)
#define OG_W9_FROM_ROW1(x) hipemu_w9_read((x), 4, 0)
#define OG_W9_ROWS01(x, r0, r1) do { (r0) = hipemu_w9_read((x), 5, 0); (r1) = hipemu_w9_read((x), 4, 0); } while (0)

// ---- runtime API ------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600, hipErrorUnknown = 999 };
typedef void* hipStream_t;
struct hipemu_event { std::chrono::steady_clock::time_point t; };
typedef hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum { hipStreamNonBlocking = 1 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; };

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess(emu)" : "hipError(emu)"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
// OG_EMU_DEVICES: how many "devices" the interpreter pretends to have (multi-GPU host logic, tests/test_emu_multi.py)
static inline hipError_t hipGetDeviceCount(int* n) { const char* e = getenv("OG_EMU_DEVICES"); *n = e && atoi(e) > 0 ? atoi(e) : 1; return hipSuccess; }
// the "current device" of the (single) interpreter thread, and where the last allocation landed: lets the multi-device tests
// see that a ctx-taking entry point binds its own device before it allocates scratch
inline int hipemu_current_device = 0;
inline int hipemu_last_malloc_device = -1;
static inline hipError_t hipSetDevice(int d) { hipemu_current_device = d; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = hipemu_current_device; return hipSuccess; }
static inline hipError_t hipDeviceGetPCIBusId(char* out, int len, int d) { snprintf(out, (size_t)len, "emu0:%02x:00.0", d); return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p)); strcpy(p->name, "hipemu"); p->multiProcessorCount = 8; return hipSuccess;  // few CUs: grids sized per CU stay small
}
static inline hipError_t hipMalloc(void** p, size_t n) {
  hipemu_last_malloc_device = hipemu_current_device;
  *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256);
  return *p ? hipSuccess : hipErrorUnknown;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = *total_b = (size_t)256 << 30; return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t = nullptr) {
  for (size_t r = 0; r < height; r++) memmove((uint8_t*)d + r * dpitch, (const uint8_t*)s + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)0x1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event(); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemu_event(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }  // (launches run to completion when issued)
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess;
}
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
