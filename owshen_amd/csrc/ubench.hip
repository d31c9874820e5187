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
    } else if (KIND == 12) {  // ONE dependent chain: what a column-serial Montgomery product issues (field.cuh)
#define X(i) asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "s2", "s3");
      REP16(X)
#undef X
    } else if (KIND == 13) {
#define X(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(acc[i]));
      REP16(X)
#undef X
    } else if (KIND == 14) {  // the dependent chain with the s_nop the hazard recognizer puts after an asm statement
#define X(i) asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0\n\ts_nop 0" : "+v"(acc[0]) : "v"(a), "v"(b) : "s2", "s3");
      REP16(X)
#undef X
    } else if (KIND == 15) {  // two interleaved dependent chains
#define X(i) asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(acc[i & 1]) : "v"(a), "v"(b) : "s2", "s3");
      REP16(X)
#undef X
    } else if (KIND == 16) {  // four interleaved chains
#define X(i) asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b) : "s2", "s3");
      REP16(X)
#undef X
    } else if (KIND >= 20 && KIND < 40) {  // KIND - 20 interleaved chains (dependency distance)
#define X(i) asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(acc[i % (KIND - 20)]) : "v"(a), "v"(b) : "s2", "s3");
      REP16(X)
#undef X
    } else if (KIND >= 40 && KIND < 60) {  // the same, not in place: the chain ping-pongs between two register pairs
#define X(i) asm volatile("v_mad_u64_u32 %0, s[2:3], %2, %3, %1" : "=v"(acc[(i % (KIND - 40)) + ((i / (KIND - 40)) & 1 ? 8 : 0)]) : "v"(acc[(i % (KIND - 40)) + ((i / (KIND - 40)) & 1 ? 0 : 8)]), "v"(a), "v"(b) : "s2", "s3");
      REP16(X)
#undef X
    } else if (KIND == 17) {  // one chain, the carry-out in vcc as the generated field code writes it
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
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

// Register-bank probe: the accumulator pair is v[40:41] (banks 0, 1 of the four VGPR banks = register index mod 4), the
// two 32-bit factors sit in v(44 + BA) and v(48 + BB) -- or the second one in an SGPR (BB == 4).  Values are garbage: only
// the issue rate matters.
template <int BA, int BB>
__global__ void __launch_bounds__(256) k_ubench_bank(uint32_t* out, int iters, uint32_t seed, unsigned long long* cycles) {
  unsigned long long t0, t1;
  uint32_t sink = seed;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int k = 0; k < iters; k++) {
#define X(i)                                                                                                            \
  if (BB < 4)                                                                                                           \
    asm volatile("v_mad_u64_u32 v[40:41], vcc, v%c0, v%c1, v[40:41]" ::"n"(44 + BA), "n"(48 + BB)                         \
                 : "v40", "v41", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "vcc");                        \
  else                                                                                                                  \
    asm volatile("v_mad_u64_u32 v[40:41], vcc, v%c0, s6, v[40:41]" ::"n"(44 + BA)                                        \
                 : "v40", "v41", "v44", "v45", "v46", "v47", "s6", "vcc");
    REP16(X)
#undef X
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (cycles && (threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);
  if (sink == 0x12345678u && t1 == 1) out[0] = sink;
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
      C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(21) C(23) C(25) C(26) C(28) C(32) C(41) C(42) C(44)
#undef C
#define B(A, Bk) case 100 + 5 * A + Bk: hipLaunchKernelGGL((k_ubench_bank<A, Bk>), g, b, 0, ctx->stream, out, iters, 1u, cyc); break;
      B(0, 0) B(0, 1) B(0, 2) B(0, 3) B(0, 4) B(1, 0) B(1, 1) B(1, 2) B(1, 3) B(1, 4) B(2, 0) B(2, 1) B(2, 2) B(2, 3) B(2, 4)
      B(3, 0) B(3, 1) B(3, 2) B(3, 3) B(3, 4)
#undef B
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
