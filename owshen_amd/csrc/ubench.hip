// Instruction-rate micro-benchmarks that calibrate the integer roofline for the big-integer
// kernels (SURVEY.md 8d caveat: the path is bound by 32-bit integer-multiply VALU rate, not
// HBM).  Each lane runs `iters` iterations of 16 independent instructions of one kind.
#include "ctx.h"
#include "mimc7.hip.h"
#include <time.h>

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
    } else if (KIND == 12) {  // ONE dependent chain: what a column-serial Montgomery product issues (field.hip.h)
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
    } else if (KIND == 60) {
#define X(i) asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(lo[i]) : "v"(b));
      REP16(X)
#undef X
    } else if (KIND == 61) {
#define X(i) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(lo[i]));
      REP16(X)
#undef X
    } else if (KIND == 62) {
#define X(i) asm volatile("v_lshrrev_b32 %0, 29, %0" : "+v"(lo[i]));
      REP16(X)
#undef X
    } else if (KIND == 63) {
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(lo[i]));
      REP16(X)
#undef X
    } else if (KIND == 64) {  // a column of the Montgomery product as generated: 9 multiply-adds, mask, 64-bit shift (x 16 / 11 per iteration)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\t" \
                          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\t" \
                          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_and_b32 %3, 0x1fffffff, %1\n\tv_lshrrev_b64 %0, 29, %0" : "+v"(acc[0]), "+v"(a), "+v"(b), "=v"(lo[i]) : : "vcc");
      REP16(X)
#undef X
    } else if (KIND == 65) {  // the same column without the 64-bit shift (what a shift-free form would issue)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\t" \
                          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\t" \
                          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_and_b32 %3, 0x1fffffff, %1" : "+v"(acc[0]), "+v"(a), "+v"(b), "=v"(lo[i]) : : "vcc");
      REP16(X)
#undef X
    } else if (KIND == 66) {  // 16 independent accumulators, carry-out in vcc
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      REP16(X)
#undef X
    } else if (KIND == 67) {  // one chain, vcc, the two factors alternating (as kind 64 without mask / shift)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
      REP16(X)
#undef X
    } else if (KIND == 68) {  // one chain, vcc, second factor in an SGPR (the reduction half's multiply-adds)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, s6, %0" : "+v"(acc[0]) : "v"(a) : "vcc", "s6");
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

// Latency of ONE wave-wide MiMC7 round on a lone wave (kinds 200 + FORM: mimc7.hip.h w9_mimc7_round<FORM>; values are garbage
// after the first round, only the chain matters): `iters` rounds, one 64-lane workgroup per block.
template <int FORM>
__global__ void __launch_bounds__(64) k_ubench_w9_round(uint32_t* out, int iters, uint32_t seed, unsigned long long* cycles) {
  const int tid = threadIdx.x, lane = FORM ? w9_row_limb(tid) : tid;
  const bool row1 = (tid & 16) != 0;
  const uint32_t nj = w9_modulus_limb<FrParams>(lane);
  uint32_t x = w9_const_limb(FrParams::ONE, lane), k = lane < 9 ? (seed >> 4) + (uint32_t)lane : 0u;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
#pragma unroll 1
  for (int i = 0; i < iters; i++) {
    uint32_t t2, t4, t6, t6r0;
    w9_mimc7_round<FORM>(x + k + (lane < 9 ? (uint32_t)i : 0u), nj, row1, t2, t4, t6, t6r0, x);
    x = w9_carry(x, lane) & 0x0fffffffu;  // (keeps the garbage inside the products' bounds: not part of a real round, ~3 instructions)
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (cycles && tid == 0) atomicMax(cycles, t1 - t0);
  if (x == 0x12345678u) out[0] = x;
}

#ifdef OG_AB_HOOKS
// Where a wave-wide product's cycles go (kinds 210 + 4 D + 2 S + G; hooks build; garbage values): three products per iteration like
// a two-row round, the digit D = 0 on the scalar unit (v_readfirstlane_b32 + s_mul_i32) | 1 per row (v_mul_lo_u32 + DPP row_newbcast)
// | 2 none (a constant: the digit's path removed); the lane shift S = 0 as built | 1 without its DPP move; G = 1 with the nine
// v_readlane_b32 of the gather per product | 0 without.
template <int D, int S, int G>
__global__ void __launch_bounds__(64) k_ubench_w9_parts(uint32_t* out, int iters, uint32_t seed, unsigned long long* cycles) {
  const int tid = threadIdx.x, lane = w9_row_limb(tid);
  const uint32_t nj = w9_modulus_limb<FrParams>(lane);
  uint32_t x = w9_const_limb(FrParams::ONE, lane) + seed;
  U9 a = w9_gather(x);
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int p = 0; p < 3; p++) {
      if (G) a = w9_gather(x);
      uint64_t acc = 0;
#pragma unroll
      for (int i = 0; i < 9; i++) {
        acc += (uint64_t)a.l[i] * x;
        const uint32_t mi = (uint32_t)acc * FrParams::INV;
        const uint32_t m = D == 0 ? OG_W9_FIRST(mi) : D == 1 ? OG_W9_ROWFIRST(mi) : 0x12345u + (uint32_t)i;
        acc += (uint64_t)m * nj;
        acc = (acc >> 29) + (S == 0 ? OG_W9_NEXT_LOW29((uint32_t)acc) : ((uint32_t)acc & MASK29));
      }
      x = (((uint32_t)acc & MASK29) + OG_W9_FROM_PREV((uint32_t)(acc >> 29))) & 0x0fffffffu;
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (cycles && tid == 0) atomicMax(cycles, t1 - t0);
  if (x == 0x12345678u) out[0] = x;
}
#endif

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
      C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(21) C(23) C(25) C(26) C(28) C(32) C(41) C(42) C(44) C(60) C(61) C(62) C(63) C(64) C(65) C(66) C(67) C(68)
#undef C
#define B(A, Bk) case 100 + 5 * A + Bk: hipLaunchKernelGGL((k_ubench_bank<A, Bk>), g, b, 0, ctx->stream, out, iters, 1u, cyc); break;
      B(0, 0) B(0, 1) B(0, 2) B(0, 3) B(0, 4) B(1, 0) B(1, 1) B(1, 2) B(1, 3) B(1, 4) B(2, 0) B(2, 1) B(2, 2) B(2, 3) B(2, 4)
      B(3, 0) B(3, 1) B(3, 2) B(3, 3) B(3, 4)
#undef B
#ifdef OG_AB_HOOKS
      case 200: hipLaunchKernelGGL(k_ubench_w9_round<0>, g, dim3(64), 0, ctx->stream, out, iters, 1u, cyc); break;
      case 201: hipLaunchKernelGGL(k_ubench_w9_round<1>, g, dim3(64), 0, ctx->stream, out, iters, 1u, cyc); break;
#define P(D, S, G) case 210 + 4 * D + 2 * S + G: hipLaunchKernelGGL((k_ubench_w9_parts<D, S, G>), g, dim3(64), 0, ctx->stream, out, iters, 1u, cyc); break;
      P(0, 0, 0) P(0, 0, 1) P(0, 1, 0) P(0, 1, 1) P(1, 0, 0) P(1, 0, 1) P(1, 1, 0) P(1, 1, 1) P(2, 0, 0) P(2, 0, 1) P(2, 1, 0) P(2, 1, 1)
#undef P
#endif
      case 202: hipLaunchKernelGGL(k_ubench_w9_round<2>, g, dim3(64), 0, ctx->stream, out, iters, 1u, cyc); break;
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

// ---- co-residency probe -----------------------------------------------------------------------------------------------------
// Round 3 question: when a persistent kernel holds `w` one-wave workgroups on every CU of the chip (the bucket accumulation),
// how fast does a kernel queued on ANOTHER stream run beside it?  k_resident is that persistent kernel in miniature -- 128
// registers per lane like k_accumulate_p<Fq>, either a dependent v_mad_u64_u32 chain (VALU busy, KIND 0) or s_sleep (slots
// held, VALU idle, KIND 1) -- and k_filler a short kernel of configurable workgroup size / LDS footprint / wave priority.
template <int KIND>
__global__ void __launch_bounds__(64) k_resident(uint32_t* out, int iters) {
  uint64_t acc = threadIdx.x + 1;
  uint32_t a = threadIdx.x * 2654435761u + 17u, b = blockIdx.x * 40503u + 977u;
  asm volatile("v_mov_b32 v120, 0" ::: "v120");  // 121 registers -> a 128-register allocation, 4 waves per SIMD at most
  for (int k = 0; k < iters; k++) {
    if (KIND == 0) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
    } else {
      asm volatile("s_sleep 64");
    }
  }
  if (acc == 0x1234567812345678ull) out[0] = (uint32_t)acc;
}

template <int PRIO>
__global__ void k_filler(const uint32_t* in, uint32_t* out, int work, int lds_bytes) {
  OG_DYN_LDS(smem);
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = in[i & 0xffff];
  for (int k = 0; k < work; k++) v = v * 1664525u + 1013904223u;
  if (lds_bytes > 0 && threadIdx.x == 0) ((volatile uint32_t*)smem)[0] = v;  // keep the allocation alive
  out[i & 0xffff] = v;
}

// out[0] = resident kernel ms, out[1] = filler ms while the resident kernel runs (queued `delay_us` after it), out[2] = the
// same filler launch alone
int ubench_coresidency(og_ctx* ctx, int wgs_per_cu, int kind, int iters, int filler_blocks, int filler_threads, int filler_lds,
                       int filler_prio, int filler_work, int delay_us, float out[3]) {
  OG_REQUIRE(delay_us >= 0 && delay_us < 1000000, "og_ubench_coresidency: delay_us must be 0 .. 999999");
  OG_REQUIRE(filler_work >= 0 && (kind == 0 || kind == 1), "og_ubench_coresidency: kind must be 0 or 1, filler_work >= 0");
  struct Res {  // every early return (OG_HIP) releases the buffer and the events
    uint32_t* buf = nullptr;
    hipEvent_t e[6] = {};
    ~Res() {
      for (auto& x : e)
        if (x) (void)hipEventDestroy(x);
      if (buf) (void)hipFree(buf);
    }
  } res;
  OG_HIP(hipMalloc((void**)&res.buf, (size_t)(1 << 16) * 4 * 2 + 64));
  uint32_t* buf = res.buf;
  OG_HIP(hipMemset(buf, 1, (size_t)(1 << 16) * 4 * 2 + 64));
  hipEvent_t (&e)[6] = res.e;
  for (auto& x : e) OG_HIP(hipEventCreate(&x));
  hipStream_t sa = ctx->lanes[0], sb = ctx->lanes[1];
  auto filler = [&](hipStream_t st) {
    if (filler_prio)
      hipLaunchKernelGGL(k_filler<1>, dim3(filler_blocks), dim3(filler_threads), (size_t)filler_lds, st, buf, buf + (1 << 16), filler_work, filler_lds);
    else
      hipLaunchKernelGGL(k_filler<0>, dim3(filler_blocks), dim3(filler_threads), (size_t)filler_lds, st, buf, buf + (1 << 16), filler_work, filler_lds);
  };
  filler(sb);  // warm the code objects
  OG_HIP(hipStreamSynchronize(sb));
  OG_HIP(hipEventRecord(e[0], sa));
  if (kind == 0)
    hipLaunchKernelGGL(k_resident<0>, dim3(wgs_per_cu * ctx->n_cu), dim3(64), 0, sa, buf + (1 << 17), iters);
  else
    hipLaunchKernelGGL(k_resident<1>, dim3(wgs_per_cu * ctx->n_cu), dim3(64), 0, sa, buf + (1 << 17), iters);
  OG_HIP(hipGetLastError());
  OG_HIP(hipEventRecord(e[1], sa));
  if (delay_us > 0) {
    struct timespec ts = {0, (long)delay_us * 1000};  // < 1 s: checked above
    nanosleep(&ts, nullptr);
  }
  OG_HIP(hipEventRecord(e[2], sb));
  filler(sb);
  OG_HIP(hipGetLastError());
  OG_HIP(hipEventRecord(e[3], sb));
  OG_HIP(hipStreamSynchronize(sa));
  OG_HIP(hipStreamSynchronize(sb));
  OG_HIP(hipEventRecord(e[4], sb));
  filler(sb);
  OG_HIP(hipEventRecord(e[5], sb));
  OG_HIP(hipStreamSynchronize(sb));
  OG_HIP(hipEventElapsedTime(&out[0], e[0], e[1]));
  OG_HIP(hipEventElapsedTime(&out[1], e[2], e[3]));
  OG_HIP(hipEventElapsedTime(&out[2], e[4], e[5]));
  return OG_OK;
}

}  // namespace og
