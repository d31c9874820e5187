// Instruction-rate micro-benchmarks that calibrate the integer roofline for the big-integer
// kernels (SURVEY.md 8d caveat: the path is bound by 32-bit integer-multiply VALU rate, not
// HBM).  Each lane runs `iters` iterations of 16 independent instructions of one kind.
#include "ctx.h"

namespace og {

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ void __launch_bounds__(256) k_ubench(uint32_t* out, int iters, uint32_t seed, unsigned long long* cycles) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 977u;
  uint64_t acc[16];
  uint32_t lo[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    acc[i] = ((uint64_t)a << 17) + i;
    lo[i] = a + i;
  }
  // shader-clock cycles of this wave's loop (s_memtime ticks at the shader clock on gfx950, MI355X_MICROARCH.md): the
  // rate in CYCLES needs no clock assumption, and cycles / milliseconds gives the effective clock of the run
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int k = 0; k < iters; k++) {
    if (KIND == 10) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(acc[(i + 5) & 15]), "v"(acc[(i + 9) & 15]));
      REP16(X)
#undef X
    } else if (KIND == 11) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(acc[(i + 5) & 15]));
      REP16(X)
#undef X
    } else if (KIND == 0) {
#define X(i) asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s2", "s3");
      REP16(X)
#undef X
    } else if (KIND == 1) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (KIND == 2) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (KIND == 3) {
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(lo[i]) : "v"(b) : "vcc");
      REP16(X)
#undef X
    } else if (KIND == 4) {
#define X(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo[i]) : "v"(b) : "vcc");
      REP16(X)
#undef X
    } else if (KIND == 5) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 15]));
      REP16(X)
#undef X
    } else if (KIND == 6) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (KIND == 7) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo[i]) : "v"(b), "v"(a));
      REP16(X)
#undef X
    } else if (KIND == 8) {
#define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (KIND == 9) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(lo[i]) : "v"(b));
      REP16(X)
#undef X
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (cycles && (threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) r ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32) ^ lo[i];
  if (r == 0x12345678u) out[0] = r;  // keep the chain live
}

int ubench(og_ctx* ctx, int kind, int iters, int blocks, float* ms, uint64_t* wave_cycles) {
  uint32_t* out = nullptr;
  OG_HIP(hipMalloc((void**)&out, 64));
  unsigned long long* cyc = (unsigned long long*)(out + 8);
  dim3 g(blocks), b(256);
  for (int rep = 0; rep < 2; rep++) {
    OG_HIP(hipMemsetAsync(out, 0, 64, ctx->stream));
    OG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(k_ubench<K>, g, b, 0, ctx->stream, out, iters, 1u, cyc); break;
      C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11)
#undef C
      default: set_error("ubench: unknown kind"); (void)hipFree(out); return OG_ERR_INVALID;
    }
    OG_HIP(hipGetLastError());
    OG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    OG_HIP(hipEventSynchronize(ctx->ev1));
    OG_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  }
  if (wave_cycles) {
    unsigned long long c = 0;
    OG_HIP(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    *wave_cycles = c;
  }
  OG_HIP(hipFree(out));
  return OG_OK;
}

}  // namespace og
