// ln_forms.h -- investigation builds only (-DENF_LN_FORM=n [-DENF_LN_PAD=1], scripts/k3_race/README.md "Packed-operand forms").
// The LayerNorm apply x <- (x - mu) * rstd of the pair kernels written with ONE explicit packed-fp32 operand form per build, as
// inline asm, so that scripts/k3_race/store_probe.py can run each form in the context the faulty one was caught in (K3's STORE
// instantiation, younger wave of a SIMD, behind a foreign kernel).  The unselected half of every broadcast pair holds 1000.0f:
// a half that picks the wrong register, or zero, changes the stored LayerNorm output.
// ENF_LN_PAD=1 puts `s_nop 1` in front of and behind every packed instruction (the hazard recognizer cannot see into asm; with
// the pads no producer / consumer is closer than two wait states: a fault that survives them is not a missing wait state).
#pragma once
#ifndef ENF_LN_PAD
#define ENF_LN_PAD 0
#endif
#if ENF_LN_PAD
#define LNF_PRE "s_nop 1\n\t"
#define LNF_POST "\n\ts_nop 1"
#else
#define LNF_PRE ""
#define LNF_POST ""
#endif

template <int NT> __device__ __forceinline__ void ln_apply_form(f32x4 (&X)[NT], float mu, float rstd) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const float nmr = -mu * rstd, junk = 1000.0f;
  asm volatile("s_nop 0" : "+v"(rstd), "+v"(mu));          // rstd is a transcendental's result: one wait state, once
  auto vmul = [](float a, float b) { float y; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b)); return y; };
  auto vsub = [](float a, float b) { float y; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b)); return y; };
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      f32x2_ x = {X[t][2 * e], X[t][2 * e + 1]}, y;
#if ENF_LN_FORM == 1      // the form caught failing: add, second operand = HIGH register broadcast by op_sel, negated
      const f32x2_ m = {junk, mu};
      asm volatile(LNF_PRE "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" LNF_POST : "=v"(y) : "v"(x), "v"(m));
      y = f32x2_{vmul(y[0], rstd), vmul(y[1], rstd)};
#elif ENF_LN_FORM == 2    // its mirror: LOW register broadcast by op_sel_hi, negated
      const f32x2_ m = {mu, junk};
      asm volatile(LNF_PRE "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" LNF_POST : "=v"(y) : "v"(x), "v"(m));
      y = f32x2_{vmul(y[0], rstd), vmul(y[1], rstd)};
#elif ENF_LN_FORM == 3    // mul, HIGH register broadcast
      const f32x2_ r = {junk, rstd}, d = {vsub(x[0], mu), vsub(x[1], mu)};
      asm volatile(LNF_PRE "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" LNF_POST : "=v"(y) : "v"(d), "v"(r));
#elif ENF_LN_FORM == 4    // mul, LOW register broadcast (what the compiler emits for the second half of the apply)
      const f32x2_ r = {rstd, junk}, d = {vsub(x[0], mu), vsub(x[1], mu)};
      asm volatile(LNF_PRE "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" LNF_POST : "=v"(y) : "v"(d), "v"(r));
#elif ENF_LN_FORM == 5    // fma, both broadcasts from HIGH registers
      const f32x2_ r = {junk, rstd}, n = {junk, nmr};
      asm volatile(LNF_PRE "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1]" LNF_POST : "=v"(y) : "v"(x), "v"(r), "v"(n));
#elif ENF_LN_FORM == 6    // fma, both broadcasts from LOW registers
      const f32x2_ r = {rstd, junk}, n = {nmr, junk};
      asm volatile(LNF_PRE "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" LNF_POST : "=v"(y) : "v"(x), "v"(r), "v"(n));
#elif ENF_LN_FORM == 7    // first operand's halves swapped (op_sel:[1,..], op_sel_hi:[0,..]): the tail kernel's most frequent form
      const f32x2_ m = {mu, mu};
      f32x2_ z;
      asm volatile(LNF_PRE "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" LNF_POST : "=v"(z) : "v"(x), "v"(m));
      y = f32x2_{vmul(z[1], rstd), vmul(z[0], rstd)};
#else
#error "ENF_LN_FORM: 1..7"
#endif
      X[t][2 * e] = y[0];
      X[t][2 * e + 1] = y[1];
    }
  if constexpr (NT == 8)
    asm volatile("s_nop 1" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]));
  else if constexpr (NT == 4)
    asm volatile("s_nop 1" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]));
}
