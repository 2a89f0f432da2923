// ucf_math.h -- fp64 complex arithmetic and complex elementary functions for gfx950.
//
// The reference's CPU build (flang + compiler-rt + glibc) fixes the semantics that
// the parity gate is measured against; this header re-implements those semantics
// as wave-resident device functions:
//   * complex*complex : plain 4-multiply form (flang inlines it);
//   * complex/complex, real/complex and complex/real : the scaled division of
//     compiler-rt's __divdc3 (lib/builtins/divdc3.c): divisor scaled by
//     2^-ilogb(max(|c|,|d|)), textbook quotient, C99 Annex G recovery cases;
//   * csqrt / cexp / ccosh / csinh : the case analysis of glibc's
//     s_csqrt/s_cexp/s_ccosh/s_csinh templates (same thresholds, same overflow
//     staging with t = 709, same Inf/NaN results) on top of real exp, sincos, cosh,
//     sinh, hypot.  The in-band rules of the reference (NaN scrub, series
//     truncation) key on exactly where Inf/NaN appear, so these cases matter.
//
// Two build flavours of every kernel include this header:
//   UCF_FAST == 0  "faithful": compiled with -ffp-contract=off, reference operation
//                   order everywhere; real functions from the device libm.
//   UCF_FAST == 1  "fast": FMA contraction allowed, exp-sharing cosh/sinh pairs and
//                   reciprocal-based division; same case analysis.
#include <hip/hip_runtime.h>
#include <math.h>

#ifndef UCF_FAST
#define UCF_FAST 0
#endif

#define UCF_DEV __device__ __forceinline__

#ifndef UCF_NS
#error "define UCF_NS (ucf_faithful / ucf_fast) before including ucf_math.h"
#endif
namespace UCF_NS {
namespace ucfm {

struct cplx {
    double re, im;
};

UCF_DEV cplx cmake(double re, double im) { cplx z; z.re = re; z.im = im; return z; }
UCF_DEV cplx cadd(cplx a, cplx b) { return cmake(a.re + b.re, a.im + b.im); }
UCF_DEV cplx csub(cplx a, cplx b) { return cmake(a.re - b.re, a.im - b.im); }
UCF_DEV cplx cneg(cplx a) { return cmake(-a.re, -a.im); }
#if UCF_FAST
// explicit FMA form: the same bits wherever it is inlined (a free-form expression is contracted differently
// in different contexts under -ffp-contract=fast)
UCF_DEV cplx cmul(cplx a, cplx b)
{
    return cmake(__builtin_fma(a.re, b.re, -(a.im * b.im)), __builtin_fma(a.re, b.im, a.im * b.re));
}
#else
UCF_DEV cplx cmul(cplx a, cplx b) { return cmake(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
#endif
UCF_DEV cplx cscale(cplx a, double s) { return cmake(a.re * s, a.im * s); }    // complex*real
UCF_DEV cplx rscale(double s, cplx a) { return cmake(s * a.re, s * a.im); }    // real*complex
UCF_DEV cplx caddr(cplx a, double s) { return cmake(a.re + s, a.im); }         // complex+real
UCF_DEV cplx radd(double s, cplx a) { return cmake(s + a.re, a.im); }          // real+complex
UCF_DEV cplx rsub(double s, cplx a) { return cmake(s - a.re, -a.im); }         // real-complex
UCF_DEV cplx csubr(cplx a, double s) { return cmake(a.re - s, a.im); }         // complex-real

UCF_DEV bool d_isnan(double x) { return x != x; }
UCF_DEV bool d_isinf(double x) { return fabs(x) == __builtin_huge_val(); }
UCF_DEV bool d_isfinite(double x) { return fabs(x) < __builtin_huge_val(); }

// rare Annex-G recovery of __divdc3, kept out of line of the hot code
__device__ __noinline__ static cplx cdiv_recover(double a, double b, double c, double d, double denom, bool scaled_inf,
                                                 cplx z)
{
    const double INF = __builtin_huge_val();
    if ((denom == 0.0) && (!d_isnan(a) || !d_isnan(b))) {
        z.re = copysign(INF, c) * a;
        z.im = copysign(INF, c) * b;
    } else if ((d_isinf(a) || d_isinf(b)) && d_isfinite(c) && d_isfinite(d)) {
        a = copysign(d_isinf(a) ? 1.0 : 0.0, a);
        b = copysign(d_isinf(b) ? 1.0 : 0.0, b);
        z.re = INF * (a * c + b * d);
        z.im = INF * (b * c - a * d);
    } else if (scaled_inf && d_isfinite(a) && d_isfinite(b)) {
        c = copysign(d_isinf(c) ? 1.0 : 0.0, c);
        d = copysign(d_isinf(d) ? 1.0 : 0.0, d);
        z.re = 0.0 * (a * c + b * d);
        z.im = 0.0 * (b * c - a * d);
    }
    return z;
}

#if UCF_FAST
// 1/x to < 1 ulp (fast flavour only): hardware estimate (v_rcp_f64, measured 2^-24.4 on gfx950,
// tools/ubench/acc.hip) and ONE third-order step r (1 + e + e^2), e = 1 - x r: residual e^3 = 2^-73, one
// instruction less than two Newton steps
UCF_DEV double fast_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}

// a*b + C, a*C and a + C with the 64-bit constant C held in an SGPR pair.  Left to itself the compiler
// selects v_fmac_f64 for a Horner step and first copies the coefficient into the destination VGPR pair
// (two v_mov_b32 per step: on CDNA every VALU instruction, a 32-bit move included, costs a full fp64 issue
// slot, so the moves doubled the cost of every polynomial); s_mov_b32 runs on the scalar unit instead.
UCF_DEV double fmak(double a, double b, double c)
{
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "s"(c));
    return o;
}
UCF_DEV double mulk(double a, double c)
{
    double o;
    asm("v_mul_f64 %0, %1, %2" : "=v"(o) : "v"(a), "s"(c));
    return o;
}
UCF_DEV double addk(double a, double c)
{
    double o;
    asm("v_add_f64 %0, %1, %2" : "=v"(o) : "v"(a), "s"(c));
    return o;
}
// A 64-bit constant materialised (two s_mov_b32) where it is used.  The compiler hoists loop-invariant
// constants into SGPRs for the whole abscissa loop; ~45 coefficient pairs do not fit beside the loop's own
// invariants, and what does not fit is parked in VGPR lanes and fetched back with v_readlane_b32 -- VALU work
// again.  `salt` is any wave-uniform value that changes per iteration: it only pins the s_mov to its place.
template <unsigned long long BITS>
UCF_DEV double sgpr_const_here(int salt)
{
    unsigned lo, hi;
    asm("s_mov_b32 %0, %1 ; constant kept local (%2)" : "=s"(lo) : "n"((unsigned)(BITS & 0xffffffffull)), "s"(salt));
    asm("s_mov_b32 %0, %1 ; constant kept local (%2)" : "=s"(hi) : "n"((unsigned)(BITS >> 32)), "s"(salt));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
#define UCF_KHERE(c, salt) sgpr_const_here<__builtin_bit_cast(unsigned long long, (double)(c))>(salt)
// g = sqrt(x) and h = 1/(2 sqrt(x)) together, 1e-290 < x < 1e290 (no scaling): v_rsq_f64 estimate and two
// coupled Newton steps on (g, h), then one correction of g.  g < 1 ulp, h ~ 1 ulp.
UCF_DEV void sqrt_hrsqrt(double x, double* g_out, double* h_out)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    *g_out = __builtin_fma(d, h, g);
    *h_out = h;
}
// sqrt(x) alone, same range: one coupled step takes the 2^-23 estimate to 2^-45 in g AND h (their errors are the same
// to first order, which is what makes the coupled step quadratic), the correction squares that again for g.  < 1 ulp.
UCF_DEV double sqrt_only(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
#endif

// compiler-rt __divdc3
UCF_DEV cplx cdiv(cplx x, cplx y)
{
    double a = x.re, b = x.im, c = y.re, d = y.im;
    const double m = fmax(fabs(c), fabs(d));          // fmax ignores a NaN operand, like compiler-rt's
    int il = 0;
    const bool fin = (m != 0.0) && d_isfinite(m);     // logb(m) finite
    if (fin) {
        il = __builtin_amdgcn_frexp_exp(m) - 1;       // == (int)logb(m), subnormals included
        c = ldexp(c, -il);
        d = ldexp(d, -il);
    }
    const double denom = c * c + d * d;
    cplx z;
#if UCF_FAST
    const double rden = fast_rcp(denom);
    z.re = ldexp((a * c + b * d) * rden, -il);
    z.im = ldexp((b * c - a * d) * rden, -il);
#else
    z.re = ldexp((a * c + b * d) / denom, -il);
    z.im = ldexp((b * c - a * d) / denom, -il);
#endif
    if (__builtin_expect(d_isnan(z.re) && d_isnan(z.im), 0))
        z = cdiv_recover(a, b, c, d, denom, d_isinf(m), z);
    return z;
}
UCF_DEV cplx rdiv(double s, cplx y) { return cdiv(cmake(s, 0.0), y); }      // real/complex
UCF_DEV cplx cdivr(cplx x, double s) { return cdiv(x, cmake(s, 0.0)); }     // complex/real (flang -> __divdc3)

UCF_DEV double cabs_(cplx z) { return hypot(z.re, z.im); }
// utility.f90:59-64: abs(z) is neither NaN nor > huge.  Decided WITHOUT forming hypot where that is certain (same answers:
// hypot is NaN or Inf as soon as a component is; two finite components below huge / sqrt(2) cannot overflow it; the rare
// rest asks hypot itself) -- the series acceleration calls this once per term, ~30 instructions each through hypot.
UCF_DEV bool c_is_finite(cplx z)
{
    if (d_isnan(z.re) || d_isnan(z.im)) return false;             // hypot: NaN, or Inf if the other component is infinite
    const double m = fmax(fabs(z.re), fabs(z.im));
    if (__builtin_expect(m < 1.2e308, 1)) return true;            // (an infinite component lands below)
    const double h = cabs_(z);
    return !(h > 1.7976931348623157e308);
}
// abs(z) > 0 as hypot decides it (driver.f90:209): an infinite component, or no NaN and a nonzero component (hypot of
// finite arguments is at least the larger modulus: it never rounds a nonzero value to zero)
UCF_DEV bool c_abs_positive(cplx z)
{
    const bool inf = d_isinf(z.re) || d_isinf(z.im);
    const bool nan = d_isnan(z.re) || d_isnan(z.im);
    return inf || (!nan && (z.re != 0.0 || z.im != 0.0));
}

// ---------------------------------------------------------------- real helpers
#define UCF_T709 709             /* (int)((DBL_MAX_EXP-1)*ln2) */
#define UCF_DBL_MIN 2.2250738585072014e-308
#define UCF_DBL_MAX 1.7976931348623157e308

#if UCF_FAST
// sin and cos together, |y| < 2^20*pi/2: first stage of fdlibm's __ieee754_rem_pio2 (pi/2 = pio2_1 + pio2_1t,
// 86 bits: the reduced argument is good to ~1e-20 ABSOLUTE for |y| < 1e6, i.e. sin and cos to < 1 ulp except
// within ~1e-4 of their zeros, where the error stays below 1e-20 absolute -- all a cosh/sinh of a complex
// argument can use) + the fdlibm/msun kernels __kernel_sin / __kernel_cos on (head, tail).
// Larger arguments go to libm.
__device__ __noinline__ static double2 sincos_huge_(double y) { double s, c; sincos(y, &s, &c); return make_double2(s, c); }

// |x| < 1e6 (the caller's business)
UCF_DEV void sincos_medium_(double x, double* sn, double* cs)
{
    const int salt = __builtin_amdgcn_readfirstlane(__double2hiint(x));
#define K(c) UCF_KHERE(c, salt)
    const double fn = __builtin_rint(mulk(x, K(6.36619772367581382433e-01)));
    const double r = __builtin_fma(-fn, 1.57079632673412561417e+00, x);    // pio2_1: exact (33-bit constant, |fn| < 2^20)
    const double w = mulk(fn, K(6.07710050650619224932e-11));              // pio2_1t = pi/2 - pio2_1
    const double y0 = r - w;
    const double y1 = (r - y0) - w;                                        // (y0, y1) = x - fn pi/2 to ~1e-20 absolute
    const double z = y0 * y0;
    // __kernel_sin(y0, y1, 1)
    const double v = z * y0;
    const double rs = fmak(z, fmak(z, fmak(z, addk(mulk(z, K(1.58969099521155010221e-10)), K(-2.50507602534068634195e-08)),
                                           K(2.75573137070700676789e-06)), K(-1.98412698298579493134e-04)), K(8.33333333332248946124e-03));
    const double ksin = y0 - ((z * (0.5 * y1 - v * rs) - y1) - mulk(v, K(-1.66666666666666324348e-01)));
    // __kernel_cos(y0, y1)  (msun form, branch free)
    const double ww = z * z;
    const double rc = __builtin_fma(ww * ww, fmak(z, addk(mulk(z, K(-1.13596475577881948265e-11)), K(2.08757232129817482790e-09)), K(-2.75573143513906633035e-07)),
                                    z * fmak(z, addk(mulk(z, K(2.48015872894767294178e-05)), K(-1.38888888888741095749e-03)), K(4.16666666666666019037e-02)));
#undef K
    const double hz = 0.5 * z;
    const double w1 = 1.0 - hz;
    const double kcos = w1 + (((1.0 - w1) - hz) + (z * rc - y0 * y1));
    const int q = (int)fn & 3;
    const double s_sel = (q & 1) ? kcos : ksin;
    const double c_sel = (q & 1) ? ksin : kcos;
    *sn = (q & 2) ? -s_sel : s_sel;
    *cs = ((q + 1) & 2) ? -c_sel : c_sel;
}
// sin and cos together from a table in LDS, |x| < 1e6 (the caller's business): x = k h + y, h = 2 pi / UCF_SC_N,
// |y| <= h/2 = 0.0123, (sin, cos)(k h) from the table (one ds_read_b128 per lane: LDS work, not VALU work), short
// polynomials in y and the angle-addition formulas in the form S + (C sin y + S (cos y - 1)).  ~20 VALU instructions
// against ~55 of sincos_medium_ (no quadrant selects, degree 7 / 6 instead of 13 / 14).
//   * k = rint(x / h) by the magic-number addition (the integer is left in the low mantissa bits: no conversion);
//   * h = H1 + H2 with H1 = pio2_1 / 64 (33 significant bits): x - k H1 is EXACT in one fma for |x| < 2^21 (the
//     difference has < 53 significant bits), the second fma rounds once; what is left of h is 1e-28 k: nothing;
//   * the table holds the correctly rounded values with exact symmetry (sin(k h) = 0 and 1 where it is), so that the
//     results keep their relative accuracy next to the zeros of sin and cos: 0 + (1 * sin y + 0).
// Error < 1.5 ulp (table entry 0.5, correction 0.5 ulp of a term <= 0.0123, final addition 0.5).
#define UCF_SC_N 256
// What a fast evaluator needs for its sin/cos: the table, a wave-uniform value that changes per loop iteration (`salt`,
// see sgpr_const_here) and three constants held in VGPRs.  A VALU instruction of gfx9 reads at most ONE scalar operand:
// x * (128/pi) + MAGIC and the leading Horner steps c7 z + c5 have two constants each, and the compiler's way out was to
// copy one of them into a VGPR pair in place (two v_mov_b32 each: 5-6 of the 26 VALU instructions of every sin/cos).
// The three constants below live in VGPRs for the whole kernel instead (opaque to the compiler, so that it does not
// rematerialise them with the same moves).
struct sc_ctx {
    const double2* tab;
    int salt;
    int kv;                     // how many of the constants below are kept in VGPRs: 0, 1 (magic) or 4 (all); compile-time
    double magic, s7, c6, e5;   // after inlining (the kernel sets it from its template arguments)
};
#define UCF_MAGIC 6755399441055744.0                                           /* 1.5 * 2^52 */
UCF_DEV void sc_ctx_init(sc_ctx& C, const double2* tab, int kv)
{
    C.tab = tab;
    C.salt = 0;
    C.kv = kv;
    C.magic = UCF_MAGIC;
    C.s7 = -1.98412698412698412698e-04;
    C.c6 = -1.38888888888888888889e-03;
    C.e5 = 1.0 / 120.0;
    if (kv >= 1) asm volatile("" : "+v"(C.magic));
    if (kv >= 4) asm volatile("" : "+v"(C.s7), "+v"(C.c6), "+v"(C.e5));
}
// a*C + b and a*b + C: C in an SGPR pair, the other two in VGPRs
UCF_DEV double fma_vsv(double a, double c, double b)
{
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(a), "s"(c), "v"(b));
    return o;
}
// t = x * c + MAGIC (c a constant in an SGPR pair) and fn = t - MAGIC = rint(x c); the integer is in the low mantissa bits
// of t.  Without the VGPR copy of MAGIC: product and sum on their own (the double rounding can only move a tie).
#define UCF_SC_MAGIC_ADD(t, fn, x, c, C, salt)                                                           \
    double t, fn;                                                                                        \
    if ((C).kv >= 1) { t = fma_vsv(x, c, (C).magic); fn = t - (C).magic; }                               \
    else { t = addk(mulk(x, c), UCF_KHERE(UCF_MAGIC, salt)); fn = addk(t, UCF_KHERE(-UCF_MAGIC, salt)); }
// first Horner step z * a + b of two constants
#define UCF_SC_LEAD(z, av, a, b, C, salt) (((C).kv >= 4) ? fmak(z, av, UCF_KHERE(b, salt)) : addk(mulk(z, UCF_KHERE(a, salt)), UCF_KHERE(b, salt)))

UCF_DEV void sincos_tab_(double x, const sc_ctx& C, double* sn, double* cs)
{
    const int salt = C.salt;
#define K(c) UCF_KHERE(c, salt)
    UCF_SC_MAGIC_ADD(t, fn, x, K(4.07436654315252059568e+01), C, salt)         // x * (128/pi)
    const int k = __double2loint(t) & (UCF_SC_N - 1);
    const double r = __builtin_fma(-fn, 2.45436926052207127214e-02, x);        // pio2_1 / 64: exact
    const double y = __builtin_fma(-fn, 9.49546954141592538956e-13, r);        // pio2_1t / 64
    const double2 T = C.tab[k];                                                // (sin, cos)(k pi / 128)
    const double z = y * y;
    // sin y - y = y z (s3 + z (s5 + z s7));  cos y - 1 = z (c2 + z (c4 + z c6))      (Taylor: |y| <= 0.0123 leaves
    // relative 2e-21 in sin, absolute 1e-20 in cos)
    const double ps = fmak(z, UCF_SC_LEAD(z, C.s7, -1.98412698412698412698e-04, 8.33333333333333333333e-03, C, salt), K(-1.66666666666666666667e-01));
    const double pc = fmak(z, UCF_SC_LEAD(z, C.c6, -1.38888888888888888889e-03, 4.16666666666666666667e-02, C, salt), K(-0.5));
#undef K
    const double sy = __builtin_fma(y * z, ps, y);
    const double cm1 = z * pc;
    *sn = T.x + __builtin_fma(T.x, cm1, T.y * sy);
    *cs = T.y + __builtin_fma(T.y, cm1, -(T.x * sy));
}

// exp(x), |x| <= 709, from a table in LDS behind the sin/cos table: x = (128 e + j) ln2/128 + r, |r| <= ln2/256 = 0.0027,
// exp(x) = 2^e * T[j] * (1 + r + r^2/2 + ... + r^5/120), T[j] = 2^(j/128) as (hi, lo) (remainder r^6/720 = 5e-19).
// k = 128 e + j by the magic-number addition as in sincos_tab_ (two's complement in the low mantissa bits: the arithmetic
// shift and the mask split a negative k correctly); ln2/128 = L1 + L2 with a 32-bit L1: k L1 is exact for |k| < 2^21.
// 15-17 VALU instructions against 18 of the table-free form, and 3 coefficients instead of 12.  The result is
// hi + (hi p + lo): one rounding of a term <= 0.003 hi and the final addition, < 0.6 ulp (a one-word table, with its own
// half ulp per entry, doubled the noise that the ill-conditioned fixtures amplify: tools/dbg_truth.py c4_malama_partpen).
#define UCF_EX_N 128
UCF_DEV double exp_tab_(double x, const sc_ctx& C)
{
    const int salt = C.salt;
#define K(c) UCF_KHERE(c, salt)
    UCF_SC_MAGIC_ADD(t, fn, x, K(184.6649652337873161420704), C, salt)         // x * (128/ln2)
    const int k = __double2loint(t);
    double r = __builtin_fma(-fn, 0x1.62e42feep-8, x);                         // exact
    r = __builtin_fma(-fn, 1.4907929134926466e-12, r);
    const double2 T = C.tab[UCF_SC_N + (k & (UCF_EX_N - 1))];
    double q = UCF_SC_LEAD(r, C.e5, 1.0 / 120.0, 1.0 / 24.0, C, salt);
    q = fmak(q, r, K(1.0 / 6.0));
#undef K
    q = __builtin_fma(q, r, 0.5);
    const double p = __builtin_fma(r * r, q, r);
    return ldexp(T.x + __builtin_fma(T.x, p, T.y), k >> 7);
}

UCF_DEV void sincos_(double x, double* sn, double* cs)
{
    if (__builtin_expect(!(fabs(x) < 1.0e6), 0)) {     // also NaN/Inf
        const double2 sc = sincos_huge_(x);
        *sn = sc.x;
        *cs = sc.y;
        return;
    }
    sincos_medium_(x, sn, cs);
}
#else
// the faithful flavour calls the device libm; out of line to keep the kernels' code small
__device__ __noinline__ static double2 sincos_libm_(double y) { double s, c; sincos(y, &s, &c); return make_double2(s, c); }
UCF_DEV void sincos_(double y, double* s, double* c)
{
    if (fabs(y) > UCF_DBL_MIN) {
        const double2 sc = sincos_libm_(y);
        *s = sc.x;
        *c = sc.y;
    } else {
        *s = y;
        *c = 1.0;
    }
}
__device__ __noinline__ static double2 coshsinh_libm_(double x) { return make_double2(cosh(x), sinh(x)); }
#endif

// cosh(x) and sinh(x) for |x| <= 709
UCF_DEV void coshsinh_(double x, double* ch, double* sh)
{
#if UCF_FAST
    // one exponential for both; below 0.35 the difference form loses bits in sinh -> series
    const double ax = fabs(x);
    const double e = exp(ax);
    const double ei = fast_rcp(e);
    *ch = 0.5 * (e + ei);
    double s;
    if (ax < 0.35) {
        const double x2 = ax * ax;
        // sinh x = x (1 + x^2/6 + x^4/120 + ... ) up to x^17 (|x|<0.35: rel err < 1e-18)
        double pl = 1.0 / 355687428096000.0;
        pl = pl * x2 + 1.0 / 1307674368000.0;
        pl = pl * x2 + 1.0 / 6227020800.0;
        pl = pl * x2 + 1.0 / 39916800.0;
        pl = pl * x2 + 1.0 / 362880.0;
        pl = pl * x2 + 1.0 / 5040.0;
        pl = pl * x2 + 1.0 / 120.0;
        pl = pl * x2 + 1.0 / 6.0;
        s = ax + ax * (x2 * pl);
    } else {
        s = 0.5 * (e - ei);
    }
    *sh = copysign(s, x);
#else
    const double2 cs_ = coshsinh_libm_(x);
    *ch = cs_.x;
    *sh = cs_.y;
#endif
}

// staged overflow handling shared by ccosh/csinh (glibc s_ccosh_template.c / s_csinh_template.c)
UCF_DEV cplx ch_sh_big_(double absx, double cosix, double sinix)
{
    const double exp_t = exp((double)UCF_T709);
    double rx = absx - UCF_T709;
    sinix *= exp_t / 2;
    cosix *= exp_t / 2;
    if (rx > UCF_T709) {
        rx -= UCF_T709;
        sinix *= exp_t;
        cosix *= exp_t;
    }
    if (rx > UCF_T709) return cmake(UCF_DBL_MAX * cosix, UCF_DBL_MAX * sinix);
    const double ev = exp(rx);
    return cmake(ev * cosix, ev * sinix);
}

// ------------------------------------------------------------ complex functions
// glibc s_ccosh_template.c
UCF_DEV cplx ccosh_(cplx x)
{
    if (__builtin_expect(d_isfinite(x.re) && d_isfinite(x.im), 1)) {
        double sinix, cosix;
        sincos_(x.im, &sinix, &cosix);
        if (__builtin_expect(fabs(x.re) > UCF_T709, 0)) {
            if (signbit(x.re)) sinix = -sinix;
            return ch_sh_big_(fabs(x.re), cosix, sinix);
        }
        double ch, sh;
        coshsinh_(x.re, &ch, &sh);
        return cmake(ch * cosix, sh * sinix);
    }
    const double NaN = __builtin_nan("");
    const double INF = __builtin_huge_val();
    if (d_isfinite(x.re)) {                     // imaginary part Inf/NaN
        return cmake(x.im - x.im, (x.re == 0.0) ? 0.0 : NaN);
    }
    if (d_isinf(x.re)) {
        if (d_isfinite(x.im) && x.im != 0.0) {
            double sinix, cosix;
            sincos_(x.im, &sinix, &cosix);
            return cmake(copysign(INF, cosix), copysign(INF, sinix) * copysign(1.0, x.re));
        }
        if (x.im == 0.0) return cmake(INF, x.im * copysign(1.0, x.re));
        return cmake(INF, x.im - x.im);
    }
    return cmake(NaN, (x.im == 0.0) ? x.im : NaN);
}

// glibc s_csinh_template.c
UCF_DEV cplx csinh_(cplx x)
{
    const bool negate = signbit(x.re);
    const double ax = fabs(x.re);
    if (__builtin_expect(d_isfinite(ax) && d_isfinite(x.im), 1)) {
        double sinix, cosix;
        sincos_(x.im, &sinix, &cosix);
        if (negate) cosix = -cosix;
        if (__builtin_expect(ax > UCF_T709, 0)) return ch_sh_big_(ax, cosix, sinix);
        double ch, sh;
        coshsinh_(ax, &ch, &sh);
        return cmake(sh * cosix, ch * sinix);
    }
    const double NaN = __builtin_nan("");
    const double INF = __builtin_huge_val();
    if (d_isfinite(ax)) {                       // imaginary part Inf/NaN
        if (ax == 0.0) return cmake(copysign(0.0, negate ? -1.0 : 1.0), x.im - x.im);
        return cmake(NaN, NaN);
    }
    if (d_isinf(ax)) {
        if (d_isfinite(x.im) && x.im != 0.0) {
            double sinix, cosix;
            sincos_(x.im, &sinix, &cosix);
            double re = copysign(INF, cosix);
            if (negate) re = -re;
            return cmake(re, copysign(INF, sinix));
        }
        if (x.im == 0.0) return cmake(negate ? -INF : INF, x.im);
        return cmake(INF, x.im - x.im);
    }
    return cmake(NaN, (x.im == 0.0) ? x.im : NaN);
}

// glibc s_cexp_template.c
UCF_DEV cplx cexp_(cplx x)
{
    if (__builtin_expect(d_isfinite(x.re) && d_isfinite(x.im), 1)) {
        double sinix, cosix;
        sincos_(x.im, &sinix, &cosix);
        double re = x.re;
        if (__builtin_expect(re > UCF_T709, 0)) {
            const double exp_t = exp((double)UCF_T709);
            re -= UCF_T709; sinix *= exp_t; cosix *= exp_t;
            if (re > UCF_T709) { re -= UCF_T709; sinix *= exp_t; cosix *= exp_t; }
            if (re > UCF_T709) return cmake(UCF_DBL_MAX * cosix, UCF_DBL_MAX * sinix);
        }
        const double ev = exp(re);
        return cmake(ev * cosix, ev * sinix);
    }
    const double NaN = __builtin_nan("");
    const double INF = __builtin_huge_val();
    if (d_isfinite(x.re)) return cmake(NaN, NaN);
    if (d_isinf(x.re)) {
        if (d_isfinite(x.im)) {
            const double value = signbit(x.re) ? 0.0 : INF;
            if (x.im == 0.0) return cmake(value, x.im);
            double sinix, cosix;
            sincos_(x.im, &sinix, &cosix);
            return cmake(copysign(value, cosix), copysign(value, sinix));
        }
        if (!signbit(x.re)) return cmake(INF, x.im - x.im);
        return cmake(0.0, copysign(0.0, x.im));
    }
    return cmake(NaN, (x.im == 0.0) ? x.im : NaN);
}

// glibc s_csqrt_template.c
UCF_DEV cplx csqrt_(cplx x)
{
    double re = x.re, im = x.im;
    if (__builtin_expect(d_isfinite(re) && d_isfinite(im) && re != 0.0 && im != 0.0, 1)) {
        int scale = 0;
        if (__builtin_expect(fabs(re) > UCF_DBL_MAX / 4, 0)) {
            scale = 1; re = ldexp(re, -2); im = ldexp(im, -2);
        } else if (__builtin_expect(fabs(im) > UCF_DBL_MAX / 4, 0)) {
            scale = 1;
            re = (fabs(re) >= 4 * UCF_DBL_MIN) ? ldexp(re, -2) : 0.0;
            im = ldexp(im, -2);
        } else if (__builtin_expect(fabs(re) < 2 * UCF_DBL_MIN && fabs(im) < 2 * UCF_DBL_MIN, 0)) {
            scale = -((53 + 1) / 2);
            re = ldexp(re, -2 * scale); im = ldexp(im, -2 * scale);
        }
#if UCF_FAST
        const double ar_ = fabs(re), ai_ = fabs(im);
        const double mx_ = fmax(ar_, ai_);
        const double d = (mx_ < 1.0e150 && mx_ > 1.0e-150) ? sqrt(re * re + im * im) : hypot(re, im);
#else
        const double d = hypot(re, im);
#endif
        double r, s;
        if (re > 0) {
            r = sqrt(0.5 * (d + re));
            if (scale == 1 && fabs(im) < 1) { s = im / r; r = ldexp(r, scale); scale = 0; }
            else s = 0.5 * (im / r);
        } else {
            s = sqrt(0.5 * (d - re));
            if (scale == 1 && fabs(im) < 1) { r = fabs(im / s); s = ldexp(s, scale); scale = 0; }
            else r = fabs(0.5 * (im / s));
        }
        if (scale) { r = ldexp(r, scale); s = ldexp(s, scale); }
        return cmake(r, copysign(s, x.im));
    }
    const double NaN = __builtin_nan("");
    const double INF = __builtin_huge_val();
    if (!(d_isfinite(re) && d_isfinite(im))) {
        if (d_isinf(im)) return cmake(INF, im);
        if (d_isinf(re)) {
            if (re < 0) return cmake(d_isnan(im) ? NaN : 0.0, copysign(INF, im));
            return cmake(re, d_isnan(im) ? NaN : copysign(0.0, im));
        }
        return cmake(NaN, NaN);
    }
    if (im == 0.0) {
        if (re < 0) return cmake(0.0, copysign(sqrt(-re), im));
        return cmake(fabs(sqrt(re)), copysign(0.0, im));
    }
    // re == 0
    double r;
    if (fabs(im) >= 2 * UCF_DBL_MIN) r = sqrt(0.5 * fabs(im));
    else r = 0.5 * sqrt(2 * fabs(im));
    return cmake(r, copysign(r, im));
}

}  // namespace ucfm
}  // namespace UCF_NS
