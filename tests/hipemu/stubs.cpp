// TEST INFRASTRUCTURE ONLY -- ubench.hip is inline gfx950 assembly; the interpreter build stubs it.
#include "ctx.h"
namespace og {
int ubench(og_ctx*, int, int, int, float* ms, uint64_t*) { *ms = 0.f; set_error("og_ubench: not available in the hipemu interpreter"); return OG_ERR_INVALID; }
int ubench_coresidency(og_ctx*, int, int, int, int, int, int, int, int, int, float*) { set_error("og_ubench_coresidency: not available in the hipemu interpreter"); return OG_ERR_INVALID; }
}

// raw-limb entry points so the 9 x 29-bit field layer can be driven with adversarial operands
// (values up to the documented bounds, not just canonical inputs)
#include "field.hip.h"
namespace {
template <class M> void fe_op_raw(int op, const uint32_t* a9, const uint32_t* b9, uint32_t* out9) {
  og::Fe<M> a, b, r;
  for (int i = 0; i < 9; i++) { a.l[i] = a9[i]; b.l[i] = b9[i]; }
  switch (op) {
    case 0: r = og::fe_add(a, b); break;
    case 1: r = og::fe_sub(a, b); break;
    case 2: r = og::fe_mul(a, b); break;
    case 3: r = og::fe_sqr(a); break;
    case 4: r = og::fe_from_mont(a); break;
    case 5: r = og::fe_neg(a); break;
    case 6: r = og::fe_inv(a); break;
    case 9: r = og::fe_mul_add(a, b, og::fe_neg_lazy(b), a); break;              /* a b + (4N - b) a */
    case 10: r = og::fe_mul_add4(a, b, og::fe_neg_lazy(a), a, og::fe_neg_lazy(b), b, a, b); break; /* 2ab - a^2 - b^2 */
    case 11: r = og::fe_mul(og::fe_dbl_lazy(a), b); break;
    case 12: r = og::fe_sub_weak(a, b); break;
    case 13: r = og::fe_add2_weak(a, b); break;
    case 14: r = og::Fe<M>::zero(); r.l[0] = og::fe_weak_diff_is_zero(og::fe_sub_weak(a, b)) ? 1u : 0u; break;
    case 15: r = og::fe_mul_add(a, a, og::fe_neg_lazy4(b), a); break;  /* a^2 + (4N - b) a */
    case 16: r = og::fe_mul_plus(a, a, og::fe_neg_lazy4(b)); break;      /* a^2 / R + 4N - b, one reduction */
    case 17: r = og::fe_mul_plus(og::fe_neg_lazy4(a), b, og::fe_neg_lazy4(a)); break;  /* (4N - a) b / R + 4N - a */
    case 7: { uint32_t w[8]; og::fe_to_words(w, a); r = og::fe_from_words<M>(w); break; }
    default: r = og::Fe<M>::zero(); r.l[0] = (a == b) ? 1u : 0u; r.l[1] = a.is_zero() ? 1u : 0u; break;
  }
  for (int i = 0; i < 9; i++) out9[i] = r.l[i];
}
}  // namespace
extern "C" void emu_fe_op(int field, int op, const uint32_t* a9, const uint32_t* b9, uint32_t* out9) {
  if (field == 0) fe_op_raw<og::FrParams>(op, a9, b9, out9); else fe_op_raw<og::FqParams>(op, a9, b9, out9);
}

#include "ec.hip.h"
// raw-limb Fq2 product / square / fused forms (operands may be weak: normalized limbs, value < 6N)
extern "C" void emu_fq2_op(int op, const uint32_t* a18, const uint32_t* b18, const uint32_t* c18, const uint32_t* d18, uint32_t* out18) {
  auto ld = [](const uint32_t* p) { og::Fq2 r; for (int i = 0; i < 9; i++) { r.c0.l[i] = p[i]; r.c1.l[i] = p[9 + i]; } return r; };
  og::Fq2 a = ld(a18), b = ld(b18), c = ld(c18), d = ld(d18), r;
  switch (op) {
    case 0: r = og::f_mul(a, b); break;
    case 1: r = og::f_sqr(a); break;
    case 2: r = og::f_mul_sub(a, b, c, d); break;
    default: r = og::f_sqr_sub(a, c, d); break;
  }
  for (int i = 0; i < 9; i++) { out18[i] = r.c0.l[i]; out18[9 + i] = r.c1.l[i]; }
}

// which pretend device is current / received the last hipMalloc (tests/test_emu_multi.py)
extern "C" int emu_current_device() { return hipemu_current_device; }
extern "C" int emu_last_malloc_device() { return hipemu_last_malloc_device; }
extern "C" void emu_set_device(int d) { hipemu_current_device = d; }

// the wave-wide ("w9") product on raw limbs: one 64-lane block, a uniform, b spread (csrc/field_w9.hip.h).  out9[9] = 1 if a
// lane above the ninth ended up non-zero (they must carry zeros: the lane above the top limb feeds the shift).
#include "field_w9.hip.h"
namespace {
template <class M> void w9_mul_raw(const uint32_t* a9, const uint32_t* b9, uint32_t* out10) {
  out10[9] = 0;
  hipemu::launch(dim3(1), dim3(64), 0, [&]() {
    const int lane = threadIdx.x;
    const og::U9 a = og::w9_uniform(a9);
    const uint32_t r = og::w9_mul<M>(a, og::w9_const_limb(b9, lane), og::w9_modulus_limb<M>(lane));
    if (lane < 9) out10[lane] = r; else if (r != 0) out10[9] = 1;
  }, "w9_mul_raw");
}
}  // namespace
// the two-row product: a uniform, row 0 carries b0, row 1 carries b1 (per-row Montgomery digit); lazy = the 32-bit digit.
// out[0..8] row 0's limbs, out[9..17] row 1's, out[18] = 1 if any other lane ended up non-zero
template <class M, bool LAZY> void w9_mul_rows_raw(const uint32_t* a9, const uint32_t* b0, const uint32_t* b1, uint32_t* out19) {
  out19[18] = 0;
  hipemu::launch(dim3(1), dim3(64), 0, [&]() {
    const int tid = threadIdx.x, lane = og::w9_row_limb(tid);
    const og::U9 a = og::w9_uniform(a9);
    const uint32_t r = og::w9_mul<M, true, LAZY>(a, og::w9_const_limb((tid & 16) ? b1 : b0, lane), og::w9_modulus_limb<M>(lane));
    if (lane < 9) out19[((tid & 16) ? 9 : 0) + lane] = r; else if (r != 0) out19[18] = 1;
  }, "w9_mul_rows_raw");
}
extern "C" void emu_w9_mul_rows(int field, int lazy, const uint32_t* a9, const uint32_t* b0, const uint32_t* b1, uint32_t* out19) {
  if (field == 0) { if (lazy) w9_mul_rows_raw<og::FrParams, true>(a9, b0, b1, out19); else w9_mul_rows_raw<og::FrParams, false>(a9, b0, b1, out19); }
  else { if (lazy) w9_mul_rows_raw<og::FqParams, true>(a9, b0, b1, out19); else w9_mul_rows_raw<og::FqParams, false>(a9, b0, b1, out19); }
}
extern "C" void emu_w9_mul(int field, const uint32_t* a9, const uint32_t* b9, uint32_t* out10) {
  if (field == 0) w9_mul_raw<og::FrParams>(a9, b9, out10); else w9_mul_raw<og::FqParams>(a9, b9, out10);
}

// the HOST's Fr arithmetic (csrc/host_fr4.h: four 64-bit limbs, R' = 2^256, constants derived from FrParams::N) on raw values:
// op 0 a + b, 1 a b / R', 2 a R' (to_mont; a may be any 256-bit value), 3 a / R' (from_mont); 4: out = [N | -N^-1 mod 2^64 | R' mod N | R'^2 mod N]
#include "host_fr4.h"
extern "C" void emu_h4_op(int op, const uint64_t* a4, const uint64_t* b4, uint64_t* out) {
  const og::H4Field& f = og::h4_field();
  og::H4 a, b, r = {{0, 0, 0, 0}};
  memcpy(a.v, a4, 32);
  memcpy(b.v, b4, 32);
  switch (op) {
    case 0: r = og::h4_add(f, a, b); break;
    case 1: r = og::h4_mul(f, a, b); break;
    case 2: r = og::h4_to_mont(f, a); break;
    case 3: r = og::h4_from_mont(f, a); break;
    default:
      memcpy(out, f.n.v, 32); out[4] = f.ninv; memcpy(out + 5, f.one.v, 32); memcpy(out + 9, f.r2.v, 32);
      return;
  }
  memcpy(out, r.v, 32);
}
