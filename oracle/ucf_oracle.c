/*
 * TEST INFRASTRUCTURE -- see ucf_oracle.h.  CPU restatement of the reference's
 * Laplace-Hankel drawdown path in plain C.  Each function cites the reference
 * file:line it follows and keeps the reference's operation order, so that with
 * glibc libm and IEEE arithmetic without contraction it reproduces the
 * reference's flang -O2 build bit for bit wherever that is achievable.
 *
 * Arithmetic conventions of the reference build that are restated here:
 *   - complex*complex is the plain 4-multiply form (flang inlines it, no
 *     __muldc3 recovery);
 *   - complex * real and complex (+,-) real act component-wise (flang's
 *     mixed-mode folding);
 *   - complex/complex, real/complex AND complex/real go through compiler-rt's __divdc3
 *     (LLVM compiler-rt lib/builtins/divdc3.c, the version in ROCm 7.2's
 *     clang 22): scale the divisor by 2^-ilogb(max|c|,|d|), then the textbook
 *     formula, then the C99 Annex G recovery cases -- restated in cdiv();
 *   - csqrt/cexp/ccosh/csinh/cabs/j0/j1 are glibc's (binary64 build).
 */
#define _GNU_SOURCE
#include "ucf_oracle.h"
#include <complex.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef UCF_ORACLE_QUAD
#include <quadmath.h>
typedef __float128 R;
#define RC(x) x##Q
#define R_EPS FLT128_EPSILON
#define R_TINY FLT128_MIN
#define R_HUGE FLT128_MAX
#define r_fabs fabsq
#define r_sqrt sqrtq
#define r_exp expq
#define r_log logq
#define r_log10 log10q
#define r_cosh coshq
#define r_sinh sinhq
#define r_tanh tanhq
#define r_cos cosq
#define r_sin sinq
#define r_atan2 atan2q
#define r_atan atanq
#define r_floor floorq
#define r_ceil ceilq
#define r_pow powq
#define r_j0 j0q
#define r_j1 j1q
#define r_logb logbq
#define r_scalbn scalbnq
#define r_fmax fmaxq
#define r_copysign copysignq
#define r_isnan isnanq
#define r_isinf isinfq
#define r_isfinite finiteq
#define r_hypot hypotq
#define r_nextafter nextafterq
#else
typedef double R;
#define RC(x) x
#define R_EPS DBL_EPSILON
#define R_TINY DBL_MIN
#define R_HUGE DBL_MAX
#define r_fabs fabs
#define r_sqrt sqrt
#define r_exp exp
#define r_log log
#define r_log10 log10
#define r_cosh cosh
#define r_sinh sinh
#define r_tanh tanh
#define r_cos cos
#define r_sin sin
#define r_atan2 atan2
#define r_atan atan
#define r_floor floor
#define r_ceil ceil
#define r_pow pow
#define r_j0 j0
#define r_j1 j1
#define r_logb logb
#define r_scalbn scalbn
#define r_fmax fmax
#define r_copysign copysign
#define r_isnan isnan
#define r_isinf isinf
#define r_isfinite isfinite
#define r_hypot hypot
#define r_nextafter nextafter
#endif

typedef struct { R re, im; } C;

/* ------------------------------------------------------------------ complex */
static inline C c_make(R re, R im) { C z = { re, im }; return z; }
static inline C c_add(C a, C b) { return c_make(a.re + b.re, a.im + b.im); }
static inline C c_sub(C a, C b) { return c_make(a.re - b.re, a.im - b.im); }
static inline C c_neg(C a) { return c_make(-a.re, -a.im); }
static inline C c_mul(C a, C b) { return c_make(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline C c_scale(C a, R s) { return c_make(a.re * s, a.im * s); }     /* complex*real */
static inline C c_rscale(R s, C a) { return c_make(s * a.re, s * a.im); }     /* real*complex */
static inline C c_addr(C a, R s) { return c_make(a.re + s, a.im); }           /* complex+real */
static inline C c_radd(R s, C a) { return c_make(s + a.re, a.im); }           /* real+complex */
static inline C c_rsub(R s, C a) { return c_make(s - a.re, -a.im); }          /* real-complex */
static inline C c_subr(C a, R s) { return c_make(a.re - s, a.im); }           /* complex-real */

/* compiler-rt __divdc3 (lib/builtins/divdc3.c) restated */
static C c_div(C x, C y)
{
    R a = x.re, b = x.im, c = y.re, d = y.im;
    int ilogbw = 0;
    R logbw = r_logb(r_fmax(r_fabs(c), r_fabs(d)));
    if (r_isfinite(logbw)) {
        ilogbw = (int)logbw;
        c = r_scalbn(c, -ilogbw);
        d = r_scalbn(d, -ilogbw);
    }
    R denom = c * c + d * d;
    C z;
    z.re = r_scalbn((a * c + b * d) / denom, -ilogbw);
    z.im = r_scalbn((b * c - a * d) / denom, -ilogbw);
    if (r_isnan(z.re) && r_isnan(z.im)) {
        if ((denom == RC(0.0)) && (!r_isnan(a) || !r_isnan(b))) {
            z.re = r_copysign((R)INFINITY, c) * a;
            z.im = r_copysign((R)INFINITY, c) * b;
        } else if ((r_isinf(a) || r_isinf(b)) && r_isfinite(c) && r_isfinite(d)) {
            a = r_copysign(r_isinf(a) ? RC(1.0) : RC(0.0), a);
            b = r_copysign(r_isinf(b) ? RC(1.0) : RC(0.0), b);
            z.re = (R)INFINITY * (a * c + b * d);
            z.im = (R)INFINITY * (b * c - a * d);
        } else if (r_isinf(logbw) && logbw > RC(0.0) && r_isfinite(a) && r_isfinite(b)) {
            c = r_copysign(r_isinf(c) ? RC(1.0) : RC(0.0), c);
            d = r_copysign(r_isinf(d) ? RC(1.0) : RC(0.0), d);
            z.re = RC(0.0) * (a * c + b * d);
            z.im = RC(0.0) * (b * c - a * d);
        }
    }
    return z;
}
static inline C c_rdiv(R s, C y) { return c_div(c_make(s, RC(0.0)), y); }     /* real/complex */
/* complex/real: flang converts the real to (s,0) and calls __divdc3 -- NOT component-wise
   (verified bit-for-bit against the reference build, see tests/test_oracle_golden.py) */
static inline C c_divr(C a, R s) { return c_div(a, c_make(s, RC(0.0))); }

#ifdef UCF_ORACLE_QUAD
static inline __complex128 toq(C z) { __complex128 w; __real__ w = z.re; __imag__ w = z.im; return w; }
static inline C fromq(__complex128 w) { return c_make(__real__ w, __imag__ w); }
static inline C c_sqrt(C z) { return fromq(csqrtq(toq(z))); }
static inline C c_exp(C z) { return fromq(cexpq(toq(z))); }
static inline C c_cosh(C z) { return fromq(ccoshq(toq(z))); }
static inline C c_sinh(C z) { return fromq(csinhq(toq(z))); }
static inline R c_abs(C z) { return cabsq(toq(z)); }
static inline C c_log(C z) { return fromq(clogq(toq(z))); }
#else
static inline double complex tod(C z) { return CMPLX(z.re, z.im); }
static inline C fromd(double complex w) { return c_make(creal(w), cimag(w)); }
static inline C c_sqrt(C z) { return fromd(csqrt(tod(z))); }
static inline C c_exp(C z) { return fromd(cexp(tod(z))); }
static inline C c_cosh(C z) { return fromd(ccosh(tod(z))); }
static inline C c_sinh(C z) { return fromd(csinh(tod(z))); }
static inline R c_abs(C z) { return cabs(tod(z)); }
static inline C c_log(C z) { return fromd(clog(tod(z))); }
#endif

static inline int c_isnan_part(C z) { return r_isnan(z.re) || r_isnan(z.im); }
/* utility.f90:59-64 */
static inline int c_is_finite(C z)
{
    R m = c_abs(z);
    return !(r_isnan(m) || m > R_HUGE);
}

static const R PI_R = RC(4.0) * RC(0.78539816339744830961566084581987572);   /* 4*atan(1) */
static const R PIOV2_R = RC(2.0) * RC(0.78539816339744830961566084581987572);

static inline C ld(const double* v, int i) { return c_make((R)v[2 * i], (R)v[2 * i + 1]); }
static inline void st(double* v, int i, C z) { v[2 * i] = (double)z.re; v[2 * i + 1] = (double)z.im; }

/* constants.f90:66 -- the MAXEXP of the *reference build* (EP = DP), also in the quad build,
 * because the branch it selects (laplace_hankel_solutions.f90:84-92) is part of the contract */
double ucfo_maxexp(void) { return -log(DBL_EPSILON) / 3.0; }

/* ------------------------------------------------------------ host-side prep */
/* utility.f90:34-49 */
void ucfo_linspace(double lo, double hi, int n, double* v)
{
    if (n == 1) {
        v[0] = (lo + hi) / 2.0;
    } else {
        double dx = (hi - lo) / (n - 1);
        for (int i = 1; i <= n; i++) v[i - 1] = lo + (i - 1) * dx;
    }
}
/* utility.f90:51-57 */
void ucfo_logspace(int lo, int hi, int n, double* v)
{
    ucfo_linspace((double)lo, (double)hi, n, v);
    for (int i = 0; i < n; i++) v[i] = pow(10.0, v[i]);
}

/* driver_io.f90:159-186 and 531-567 */
int ucfo_nondim(const ucf_params* P, ucf_derived* D)
{
    const double PI = 4.0 * atan(1.0);
    memset(D, 0, sizeof(*D));
    double l = P->l, d = P->d, ac = P->ac;
    if (P->MNtype == 1) {                       /* driver_io.f90:159-186 */
        if (fabs(P->ac - P->ak) > FLT_EPSILON) ac = P->ak;   /* epsilon(1.0) is single precision */
        if (fabs(l - P->b) > FLT_EPSILON) l = P->b;
        if (d > FLT_EPSILON) d = 0.0;
    }
    D->l_eff = l; D->d_eff = d; D->ac_eff = ac;
    D->Lc = P->b;
    D->Tc = D->Lc * D->Lc / (P->Kr / P->Ss);
    D->Hc = P->Q / (4 * PI * P->Kr * P->b);
    D->sigma = P->Sy / (P->Ss * P->b);
    D->alphaD = P->kappa / D->sigma;
    D->betaD = P->beta / D->Lc;
    D->lD = l / D->Lc;
    D->dD = d / D->Lc;
    D->bD = D->lD - D->dD;
    D->rDw = P->rw / D->Lc;
    D->rDwobs = P->rwobs / D->Lc;
    for (int m = 0; m < P->MoenchM && m < UCF_MAX_MOENCH; m++)
        D->MoenchGamma[m] = P->MoenchAlpha[m] * D->Lc * P->Sy / (P->kappa * P->Kr);
    D->acD = ac * D->Lc;
    D->akD = P->ak * D->Lc;
    D->lambdaD = (P->ak - ac) * D->Lc;
    D->psiaD = P->psia / D->Lc;
    D->psikD = P->psik / D->Lc;
    D->usLD = P->usL / D->Lc;
    D->b1 = P->psia - P->psik;
    D->PsiD = D->b1 / D->Lc;
    D->np = 2 * P->M + 1;
    D->N = (1 << P->k) - 1;
    D->nj0z = (P->j0s[0] > P->j0s[1] ? P->j0s[0] : P->j0s[1]) + P->nacc + 1;
    D->nabs = D->N + P->nacc * (P->ord - 2);
    return 0;
}

/* driver_io.f90:575-586 */
void ucfo_zlay(const ucf_derived* D, int nz, const double* zD, int* zLay)
{
    for (int i = 0; i < nz; i++) {
        if (zD[i] <= 0.0 || zD[i] < (1.0 - D->lD)) zLay[i] = 1;
        else if ((zD[i] - 1.0) >= 0.0 || zD[i] < (1.0 - D->dD)) zLay[i] = 2;
        else zLay[i] = 3;
    }
}

/* driver_io.f90:628-647 */
void ucfo_j0_zeros(int n, double* j0z)
{
    const double PI = 4.0 * atan(1.0);
    for (int i = 0; i < n; i++) {
        double x = (i + 0.75) * PI;
        for (;;) {
            double dx = j0(x) / j1(x);
            x = x + dx;
            /* spacing(x): distance to the next representable number of larger magnitude */
            if (fabs(dx) < nextafter(fabs(x), INFINITY) - fabs(x)) break;
        }
        j0z[i] = x;
    }
}

/* driver_io.f90:654-664 */
void ucfo_split_vector(const int j0s[2], int nt, const double* tD, int* sv)
{
    int mx = j0s[0] > j0s[1] ? j0s[0] : j0s[1];
    int mn = j0s[0] < j0s[1] ? j0s[0] : j0s[1];
    int zrange = mx - mn;
    double lmin = INFINITY, lmax = -INFINITY;
    for (int i = 0; i < nt; i++) {
        double lg = log10(tD[i]);
        if (lg < lmin) lmin = lg;
        if (lg > lmax) lmax = lg;
    }
    int minlsp = (int)floor(lmin);
    int maxlsp = (int)ceil(lmax);
    int sprange = maxlsp - minlsp + 1;
    for (int i = 0; i < nt; i++)
        sv[i] = mn + (int)(zrange * ((maxlsp - log10(tD[i])) / sprange));
}

/* ------------------------------------------------------------------- de Hoog */
/* invlap.f90:154-172 */
static void pvalues_R(R tee, int M, R alpha, R tol, C* p)
{
    R sigma = alpha - r_log(tol) / (RC(2.0) * tee);
    for (int i = 0; i <= 2 * M; i++) p[i] = c_make(sigma, PI_R * i / tee);
}
void ucfo_pvalues(double tee, int M, double alpha, double tol, double* p_re_im)
{
    C* p = (C*)malloc(sizeof(C) * (2 * M + 1));
    pvalues_R((R)tee, M, (R)alpha, (R)tol, p);
    for (int i = 0; i <= 2 * M; i++) st(p_re_im, i, p[i]);
    free(p);
}

/* counts of the reference's in-band rules taken since ucfo_stats_reset (same six counters as ucf_stats, include/ucf.h):
 * 0 nan_scrubbed (invlap.f90:71-74, samples), 1 zero_vectors (invlap.f90:69,139), 2 wynn_truncated, 3 wynn_sentinel,
 * 4 wynn_early_exit (integration.f90:140-177), 5 wynn_all_zero (driver.f90:209).  Only point_R counts (g_count_on). */
static long long g_cnt[6];
static int g_count_on = 0;
#pragma omp threadprivate(g_count_on)
static void cnt_add(int which, long long n)
{
    if (!g_count_on || n == 0) return;
#pragma omp atomic
    g_cnt[which] += n;
}
void ucfo_stats_reset(void) { for (int i = 0; i < 6; i++) g_cnt[i] = 0; }
void ucfo_stats_get(long long* out) { for (int i = 0; i < 6; i++) out[i] = g_cnt[i]; }

/* invlap.f90:46-141 with nt = 1 (scalar wrapper :143-152) */
static R dehoog_R(int M, R alpha, R tol, R t, R tee, const C* fp)
{
    const int n2 = 2 * M;
    R mx = RC(0.0);
    int anynan = 0;
    /* maxval(abs(fp)): flang's inlined MAXVAL ignores NaN operands unless all are NaN */
    for (int i = 0; i <= n2; i++) {
        R m = c_abs(fp[i]);
        if (r_isnan(m)) { anynan = 1; continue; }
        if (m > mx) mx = m;
    }
    (void)anynan;
    if (!(mx > R_TINY)) { cnt_add(1, 1); return RC(0.0); }                /* :69,139 */
    { long long nn = 0; for (int i = 0; i <= n2; i++) nn += c_isnan_part(fp[i]) ? 1 : 0; cnt_add(0, nn); }

    C* ff = (C*)malloc(sizeof(C) * (n2 + 1));
    C* e = (C*)calloc((size_t)(n2 + 1) * (M + 1), sizeof(C));             /* e(0:2M,0:M) */
    C* q = (C*)calloc((size_t)(n2 + 1) * (M + 1), sizeof(C));             /* q(0:2M,1:M) */
    C* d = (C*)malloc(sizeof(C) * (n2 + 1));
#define E(i, r) e[(size_t)(r) * (n2 + 1) + (i)]
#define Q(i, r) q[(size_t)(r) * (n2 + 1) + (i)]
    for (int i = 0; i <= n2; i++)                                         /* :71-74 */
        ff[i] = c_isnan_part(fp[i]) ? c_make(RC(0.0), RC(0.0)) : fp[i];

    R gamma = alpha - r_log(tol) / (RC(2.0) * tee);                       /* :77 */

    for (int i = 0; i <= n2; i++) E(i, 0) = c_make(RC(0.0), RC(0.0));     /* :80 */
    Q(0, 1) = c_div(ff[1], c_divr(ff[0], RC(2.0)));                       /* :81 */
    for (int i = 1; i <= n2 - 1; i++) Q(i, 1) = c_div(ff[i + 1], ff[i]);  /* :82 */

    for (int r = 1; r <= M; r++) {                                        /* :85-95 */
        int max = 2 * (M - r);
        for (int i = 0; i <= max; i++)
            E(i, r) = c_add(c_sub(Q(i + 1, r), Q(i, r)), E(i + 1, r - 1));
        if (r != M) {
            int rq = r + 1;
            max = 2 * (M - rq) + 1;
            for (int i = 0; i <= max; i++)
                Q(i, rq) = c_div(c_mul(Q(i + 1, rq - 1), E(i + 1, rq - 1)), E(i, rq - 1));
        }
    }

    d[0] = c_divr(ff[0], RC(2.0));                                        /* :98-102 */
    for (int r = 1; r <= M; r++) {
        d[2 * r - 1] = c_neg(Q(0, r));
        d[2 * r] = c_neg(E(0, r));
    }

    /* :105-117, A(-1)=0, A(0)=d(0), B(-1)=B(0)=1 */
    C Am2 = c_make(RC(0.0), RC(0.0)), Am1 = d[0];
    C Bm2 = c_make(RC(1.0), RC(0.0)), Bm1 = c_make(RC(1.0), RC(0.0));
    /* z = exp(cmplx(0,1)*PI*t/tee)  (:110) */
    C z = c_exp(c_divr(c_scale(c_scale(c_make(RC(0.0), RC(1.0)), PI_R), t), tee));
    for (int n = 1; n <= n2 - 1; n++) {
        C An = c_add(Am1, c_mul(c_mul(d[n], Am2), z));
        C Bn = c_add(Bm1, c_mul(c_mul(d[n], Bm2), z));
        Am2 = Am1; Am1 = An;
        Bm2 = Bm1; Bm1 = Bn;
    }
    /* :120-121 */
    C brem = c_divr(c_radd(RC(1.0), c_mul(c_sub(d[n2 - 1], d[n2]), z)), RC(2.0));
    C inner = c_sqrt(c_radd(RC(1.0), c_div(c_mul(d[n2], z), c_mul(brem, brem))));
    C rem = c_neg(c_mul(brem, c_rsub(RC(1.0), inner)));
    /* :124-125 */
    C A2M = c_add(Am1, c_mul(rem, Am2));
    C B2M = c_add(Bm1, c_mul(rem, Bm2));
    /* :129 */
    R ft = r_exp(gamma * t) / tee * c_div(A2M, B2M).re;
#undef E
#undef Q
    free(ff); free(e); free(q); free(d);
    return ft;
}
double ucfo_dehoog(int M, double alpha, double tol, double t, double tee, const double* fp_re_im)
{
    C* fp = (C*)malloc(sizeof(C) * (2 * M + 1));
    for (int i = 0; i <= 2 * M; i++) fp[i] = ld(fp_re_im, i);
    R ft = dehoog_R(M, (R)alpha, (R)tol, (R)t, (R)tee, fp);
    free(fp);
    return (double)ft;
}

/* -------------------------------------------------------------- quadrature */
/* integration.f90:31-67 */
static void tanh_sinh_R(int k, R s, R* w, R* a)
{
    int N = (1 << k) - 1;
    int r = (N - 1) / 2;
    R h = RC(4.0) / (R)(1 << k);
    R sum = RC(0.0);
    R* u2 = (R*)malloc(sizeof(R) * N);
    for (int i = -r; i <= r; i++) {
        R u1 = PIOV2_R * r_cosh(h * i);
        u2[i + r] = PIOV2_R * r_sinh(h * i);
        R c = r_cosh(u2[i + r]);
        w[i + r] = u1 / (c * c);
    }
    for (int i = 0; i < N; i++) sum = sum + w[i];
    for (int i = 0; i < N; i++) w[i] = RC(2.0) * w[i] / sum;
    if (a)
        for (int i = 0; i < N; i++) a[i] = (r_tanh(u2[i]) + RC(1.0)) * s / RC(2.0);
    free(u2);
}
void ucfo_tanh_sinh(int k, double s, double* w, double* a)
{
    int N = (1 << k) - 1;
    R* wr = (R*)malloc(sizeof(R) * N);
    R* ar = (R*)malloc(sizeof(R) * N);
    tanh_sinh_R(k, (R)s, wr, a ? ar : NULL);
    for (int i = 0; i < N; i++) { w[i] = (double)wr[i]; if (a) a[i] = (double)ar[i]; }
    free(wr); free(ar);
}

/* integration.f90:70-120 */
static void gauss_lobatto_R(int ord, R* xo, R* wo)
{
    int N = ord - 1, N1 = N + 1;
    R* P = (R*)calloc((size_t)ord * ord, sizeof(R));   /* P(i,k): i node, k degree+1 */
    R* x = (R*)malloc(sizeof(R) * ord);
    R* xold = (R*)malloc(sizeof(R) * ord);
#define PP(i, k) P[(size_t)((k) - 1) * ord + (i)]
    for (int i = 0; i <= N; i++) x[i] = r_cos(PI_R * i / N);
    for (int i = 0; i < ord; i++) xold[i] = RC(2.0);
    for (;;) {
        R mx = RC(0.0);
        for (int i = 0; i < ord; i++) { R dd = r_fabs(x[i] - xold[i]); if (dd > mx) mx = dd; }
        if (!(mx > R_EPS)) break;                 /* spacing(1.0) = epsilon */
        for (int i = 0; i < ord; i++) {
            xold[i] = x[i];
            PP(i, 1) = RC(1.0);
            PP(i, 2) = x[i];
        }
        for (int k = 2; k <= N; k++)
            for (int i = 0; i < ord; i++)
                PP(i, k + 1) = ((2 * k - 1) * x[i] * PP(i, k) - (k - 1) * PP(i, k - 1)) / k;
        for (int i = 0; i < ord; i++)
            x[i] = xold[i] - (x[i] * PP(i, N1) - PP(i, N)) / (N1 * PP(i, N1));
    }
    for (int i = 1; i <= ord - 2; i++) {
        xo[i - 1] = x[i];
        wo[i - 1] = RC(2.0) / ((N * N1) * (PP(i, N1) * PP(i, N1)));
    }
#undef PP
    free(P); free(x); free(xold);
}
void ucfo_gauss_lobatto(int ord, double* x, double* w)
{
    R* xr = (R*)malloc(sizeof(R) * ord);
    R* wr = (R*)malloc(sizeof(R) * ord);
    gauss_lobatto_R(ord, xr, wr);
    for (int i = 0; i < ord - 2; i++) { x[i] = (double)xr[i]; w[i] = (double)wr[i]; }
    free(xr); free(wr);
}

/* integration.f90:125-189 */
static C wynn_R(int nin, const C* series, int* status)
{
    enum { MINTERMS = 4 };
    int ns = nin;
    /* eps(1:ns, -1:ns-1) */
    C* eps = (C*)calloc((size_t)(nin + 1) * (nin + 2), sizeof(C));
#define EPS(m, j) eps[(size_t)((j) + 1) * (nin + 1) + (m)]
    C acc;
    if (status) *status = 0;
    C run = c_make(RC(0.0), RC(0.0));
    for (int i = 1; i <= nin; i++) {
        if (!c_is_finite(series[i - 1])) {
            ns = i - 1;
            if (ns < MINTERMS) {
                acc = c_make((R)(-999999.9f), RC(0.0));    /* default-real literal, :148 */
                if (status) *status = 2;
                goto done;
            }
            if (status) *status = 1;
            break;
        }
        run = (i == 1) ? series[0] : c_add(run, series[i - 1]);    /* sum(series(1:i)) */
        EPS(i, 0) = run;
    }
    for (int m = 1; m <= nin; m++) EPS(m, -1) = c_make(RC(0.0), RC(0.0));
    for (int j = 0; j <= ns - 2; j++) {
        for (int m = 1; m <= ns - (j + 1); m++) {
            C denom = c_sub(EPS(m + 1, j), EPS(m, j));
            if (c_abs(denom) > R_EPS) {
                EPS(m, j + 1) = c_add(EPS(m + 1, j - 1), c_rdiv(RC(1.0), denom));
            } else {
                acc = EPS(m + 1, j);
                if (status) *status = 3;
                goto done;
            }
        }
    }
    if (ns % 2 == 0) acc = EPS(2, ns - 2);
    else acc = EPS(2, ns - 3);
done:
#undef EPS
    free(eps);
    return acc;
}
void ucfo_wynn_epsilon(int n, const double* series_re_im, double* acc_re_im, int* status)
{
    C* s = (C*)malloc(sizeof(C) * n);
    for (int i = 0; i < n; i++) s[i] = ld(series_re_im, i);
    C acc = wynn_R(n, s, status);
    st(acc_re_im, 0, acc);
    free(s);
}

/* integration.f90:192-237 */
static C extrap_R(int n, const R* xin, const C* yin)
{
    C c[64], d[64], den[64];
    int ns = 1;
    for (int i = 2; i <= n; i++) if (xin[i - 1] < xin[ns - 1]) ns = i;    /* minloc, first minimum */
    for (int i = 0; i < n; i++) { c[i] = yin[i]; d[i] = yin[i]; }
    C y = yin[ns - 1];
    ns = ns - 1;
    for (int m = 1; m <= n - 1; m++) {
        for (int i = 1; i <= n - m; i++) {
            /* den is a COMPLEX array in the reference: real difference stored as (dx,0),
               then a complex/complex division (:222-228) */
            R dx = xin[i - 1] - xin[i + m - 1];
            den[i - 1] = c_div(c_sub(c[i], d[i - 1]), c_make(dx, RC(0.0)));
        }
        for (int i = 1; i <= n - m; i++) {
            d[i - 1] = c_rscale(xin[i + m - 1], den[i - 1]);
            c[i - 1] = c_rscale(xin[i - 1], den[i - 1]);
        }
        C dy;
        if (2 * ns < n - m) {
            dy = c[ns];
        } else {
            dy = d[ns - 1];
            ns = ns - 1;
        }
        y = c_add(y, dy);
    }
    return y;
}
void ucfo_extraptozero(int n, const double* x, const double* y_re_im, double* out_re_im)
{
    R xr[64]; C yr[64];
    for (int i = 0; i < n; i++) { xr[i] = (R)x[i]; yr[i] = ld(y_re_im, i); }
    st(out_re_im, 0, extrap_R(n, xr, yr));
}


/* ---------------------------------------------------------- K0, K1 (Amos) */
/* cbesk(z, fnu=0, kode=1, n=2) for Re z >= 0: cbessel.f90:877-1146 -> cbknu :5036-5495.
 * Only the branches that fnu = 0, n = 2, kode = 1 can reach are restated:
 *   |z| <= 2 : power series (:5098-5201);  |z| > 2 : Miller backward recurrence (:5209-5327).
 * Returns ierr as cbesk does (0 ok; 1 bad input; 2 overflow; 4/5 range/convergence). */
static int cbesk01_R(C z, C* k0, C* k1)
{
    const R tol = (R)DBL_EPSILON;            /* MAX(EPSILON, 1e-18) of the binary64 reference build */
    const R elim = RC(2.303) * ((R)1021 * r_log10(RC(2.0)) - RC(3.0));
    const R aa0 = r_log10(RC(2.0)) * (R)52;
    const R alim = elim + r_fmax(-(aa0 * RC(2.303)), RC(-41.45));
    /* cbessel.f90:20-25: derived from atan(1) in working precision, not literals */
    const R at1 = RC(0.78539816339744830961566084581987572);
    const R pi = RC(4.0) * at1, hpi = RC(2.0) * at1, spi = RC(3.0) / (RC(2.0) * at1);
    const R rthpi = r_sqrt(RC(8.0) * at1) / RC(2.0);
    const R fpi = RC(1.89769999331517738), tth = RC(6.66666666666666666e-01);
    const R gamma_e = RC(5.77215664901532861e-01);
    const R xx = z.re, yy = z.im;
    *k0 = c_make(RC(0.0), RC(0.0));
    *k1 = *k0;
    if (r_fabs(yy) == RC(0.0) && r_fabs(xx) == RC(0.0)) return 1;
    const R az = c_abs(z);
    {   /* range tests of cbesk :1061-1078 */
        R a = RC(0.5) / tol, b = (R)2147483647 * RC(0.5);
        if (b < a) a = b;
        if (az > a) return 4;
        if (az < (R)DBL_MIN * RC(1.0e3)) return 2;
        /* fn = 1 <= 2: only the az <= tol overflow test applies (:1082-1087) */
        if (!(az > tol)) { R aln = -RC(1.0) * r_log(RC(0.5) * az); if (aln > elim) return 2; }
    }
    const R caz = az;
    const C rz = c_div(c_make(RC(2.0), RC(0.0)), z);                      /* :5085 */
    C s1, s2;
    if (caz <= RC(2.0)) {                                                 /* :5098 */
        const C smu = c_log(rz);
        C f = c_add(c_make(-gamma_e, RC(0.0)), smu);                      /* g1*cch + smu*g2, g1=-gamma, cch=1, g2=1 */
        /* exactly as the reference orders it: f = g1*cch + smu*g2 with cch = (1,0): g1*cch = (g1*1, g1*0) */
        f = c_add(c_rscale(-gamma_e, c_make(RC(1.0), RC(0.0))), c_scale(smu, RC(1.0)));
        C p = c_make(RC(0.5), RC(0.0)), q = c_make(RC(0.5), RC(0.0));
        s1 = f;
        s2 = p;
        R ak = RC(1.0), a1 = RC(1.0), bk = RC(1.0);
        C ck = c_make(RC(1.0), RC(0.0));
        if (caz >= tol) {
            const C cz = c_scale(c_mul(z, z), RC(0.25));
            const R t1 = RC(0.25) * caz * caz;
            do {                                                          /* :5172-5182 */
                f = c_divr(c_add(c_add(c_scale(f, ak), p), q), bk);
                p = c_divr(p, ak);            /* ak - dnu, dnu = 0 */
                q = c_divr(q, ak);
                const R rk = RC(1.0) / ak;
                ck = c_scale(c_mul(ck, cz), rk);
                s1 = c_add(s1, c_mul(ck, f));
                s2 = c_add(s2, c_mul(ck, c_sub(p, c_scale(f, ak))));
                a1 = a1 * t1 * rk;
                bk = bk + ak + ak + RC(1.0);
                ak = ak + RC(1.0);
            } while (a1 > tol);
        }
        /* kflag = 2 unless (fnu+1)*|Re smu| > alim (never for |z| > 1e-288) */
        s2 = c_mul(s2, rz);                                               /* p2 = s2*css(2); s2 = p2*rz */
        *k0 = s1;
        *k1 = s2;
        return 0;
    }
    /* |z| > 2: :5209-5327 */
    if (xx > alim) return 2;      /* the scaled-exponential continuation (:5476-5479) is not restated: |K| < 1e-288 */
    C coef = c_div(c_make(rthpi, RC(0.0)), c_sqrt(z));
    {
        const R a1 = r_exp(-xx);
        const C pt = c_rscale(a1, c_make(r_cos(yy), -r_sin(yy)));
        coef = c_mul(coef, pt);
    }
    R ak = r_fabs(r_cos(pi * RC(0.0)));
    R fhs = RC(0.25);
    R t1 = (R)52 * r_log10(RC(2.0)) * RC(3.321928094);
    if (t1 < RC(12.0)) t1 = RC(12.0);
    if (t1 > RC(60.0)) t1 = RC(60.0);
    R t2 = tth * t1 - RC(6.0);
    if (xx == RC(0.0)) t1 = hpi;
    else t1 = r_fabs(r_atan(yy / xx));
    R fk;
    if (t2 <= caz) {
        const R etest = ak / (pi * caz * tol);
        fk = RC(1.0);
        if (!(etest < RC(1.0))) {
            R fks = RC(2.0), rk = caz + caz + RC(2.0), a1 = RC(0.0), a2 = RC(1.0);
            int found = 0;
            for (int i = 1; i <= 30; i++) {
                ak = fhs / fks;
                const R bk = rk / (fk + RC(1.0));
                R tm = a2;
                a2 = bk * a2 - ak * a1;
                a1 = tm;
                rk = rk + RC(2.0);
                fks = fks + fk + fk + RC(2.0);
                fhs = fhs + fk + fk;
                fk = fk + RC(1.0);
                tm = r_fabs(a2) * fk;
                if (etest < tm) { found = 1; break; }
            }
            if (!found) return 5;
            fk = fk + spi * t1 * r_sqrt(t2 / caz);
            fhs = RC(0.25);
        }
    } else {
        const R a2 = r_sqrt(caz);
        ak = fpi * ak / (tol * r_sqrt(a2));
        const R aa = RC(3.0) * t1 / (RC(1.0) + caz);
        const R bb = RC(14.7) * t1 / (RC(28.0) + caz);
        ak = (r_log(ak) + caz * r_cos(aa) / (RC(1.0) + RC(0.008) * caz)) / r_cos(bb);
        fk = RC(0.12125) * ak * ak / caz + RC(1.5);
    }
    const int k = (int)fk;
    fk = (R)k;
    R fks = fk * fk;
    C p1 = c_make(RC(0.0), RC(0.0)), p2 = c_make(tol, RC(0.0)), cs = p2;
    for (int i = 1; i <= k; i++) {                                        /* :5279-5291 */
        const R a1 = fks - fk;
        const R a2 = (fks + fk) / (a1 + fhs);
        const R rk = RC(2.0) / (fk + RC(1.0));
        const R tt1 = (fk + xx) * rk, tt2 = yy * rk;
        const C pt = p2;
        p2 = c_scale(c_sub(c_mul(p2, c_make(tt1, tt2)), p1), a2);
        p1 = pt;
        cs = c_add(cs, p2);
        fks = a1 - fk + RC(1.0);
        fk = fk - RC(1.0);
    }
    R tm = c_abs(cs);
    C pt = c_make(RC(1.0) / tm, RC(0.0));
    s1 = c_mul(pt, p2);
    cs = c_mul(c_make(cs.re, -cs.im), pt);
    s1 = c_mul(c_mul(coef, s1), cs);
    tm = c_abs(p2);
    pt = c_make(RC(1.0) / tm, RC(0.0));
    p1 = c_mul(pt, p1);
    p2 = c_mul(c_make(p2.re, -p2.im), pt);
    pt = c_mul(p1, p2);
    s2 = c_mul(s1, c_add(c_make(RC(1.0), RC(0.0)), c_div(c_sub(c_make(RC(0.5), RC(0.0)), pt), z)));
    *k0 = s1;
    *k1 = s2;
    return 0;
}

int ucfo_cbesk01(double zr, double zi, double* k_re_im /* K0re,K0im,K1re,K1im */)
{
    C k0, k1;
    int ierr = cbesk01_R(c_make((R)zr, (R)zi), &k0, &k1);
    st(k_re_im, 0, k0);
    st(k_re_im, 1, k1);
    return ierr;
}

/* ------------------------------------------------ Laplace-Hankel evaluators */
typedef struct {
    int model, MNtype, order, timeType, MoenchM;
    R timePar[2];
    R sched_t[UCF_MAX_SCHEDULE], sched_dq[UCF_MAX_SCHEDULE], sched_tf, sched_sum;   /* time.f90:81-95 */
    R kappa, alphaD, beta, lD, dD, bD;
    R MoenchGamma[UCF_MAX_MOENCH];
    R Sy, Ss, b, ak, ac, psia, psik, b1, akD, lambdaD, usLD;
    R rDw, rDwobs, sF, screen;   /* model 2: screen = l - d (dimensional) */
    R maxexp;
} mdl;

static void mdl_fill(mdl* m, const ucf_params* P, const ucf_derived* D)
{
    m->model = P->model; m->MNtype = P->MNtype; m->order = P->order;
    m->timeType = P->timeType; m->MoenchM = P->MoenchM;
    m->timePar[0] = (R)P->timePar[0]; m->timePar[1] = (R)P->timePar[1];
    if (P->timeType < 0 && P->timeType >= -(100 + UCF_MAX_SCHEDULE)) {
        /* time.f90:83-89 piecewise constant: rate increments; time.f90:99-113 piecewise linear (timeType <= -101):
         * slope increments W_k - W_{k-1}, W_k = (y_{k+1} - y_k)/(t_{k+1} - t_k).  The reference unpacks the n rates into
         * y(1:n) and then reads y(n+1), one past the array (:101,113); its transform uses rate differences only, i.e. a
         * rate that starts from 0 at t_1 ("no jumps", :98).  Restated in bounds: y(t_1) = 0 and the n parameters are the
         * rates at t_2..t_n, t_f (every parameter used, nothing read out of bounds). */
        const int linear = P->timeType <= -101;
        const int n = linear ? -P->timeType - 100 : -P->timeType;
        double prev = 0.0, sum = 0.0, yprev = 0.0;
        for (int k = 0; k < n; k++) {
            double cur = P->timeParExt[n + 1 + k];
            if (linear) {
                const double denom = P->timeParExt[k + 1] - P->timeParExt[k];
                const double w = (cur - yprev) / denom;
                yprev = cur;
                cur = w;
            }
            const double dq = cur - prev;
            prev = cur;
            m->sched_t[k] = (R)P->timeParExt[k];
            m->sched_dq[k] = (R)dq;
            sum = (k == 0) ? dq : sum + dq;
        }
        m->sched_tf = (R)P->timeParExt[n];
        m->sched_sum = (R)sum;
    }
    m->kappa = (R)P->kappa; m->alphaD = (R)D->alphaD; m->beta = (R)P->beta;
    m->lD = (R)D->lD; m->dD = (R)D->dD; m->bD = (R)D->bD;
    for (int i = 0; i < UCF_MAX_MOENCH; i++) m->MoenchGamma[i] = (R)D->MoenchGamma[i];
    m->Sy = (R)P->Sy; m->Ss = (R)P->Ss; m->b = (R)P->b; m->ak = (R)P->ak; m->ac = (R)D->ac_eff;
    m->psia = (R)P->psia; m->psik = (R)P->psik; m->b1 = (R)D->b1; m->akD = (R)D->akD;
    m->lambdaD = (R)D->lambdaD; m->usLD = (R)D->usLD;
    m->rDw = (R)D->rDw; m->rDwobs = (R)D->rDwobs; m->sF = (R)P->sF; m->screen = (R)D->l_eff - (R)D->d_eff;
    m->maxexp = (R)ucfo_maxexp();
}

/* time.f90:34-124 -- cases 1-6, 8 (case 7 is identically zero in the reference, quirk Q4) */
static C lap_time(const mdl* m, C p)
{
    const R t1 = m->timePar[0], t2 = m->timePar[1];
    if (m->timeType < 0 && m->timeType >= -(100 + UCF_MAX_SCHEDULE)) {   /* :81-95 piecewise-constant, :97-122 piecewise-linear rate */
        const int linear = m->timeType <= -101;
        const int n = linear ? -m->timeType - 100 : -m->timeType;
        C sum = c_make(RC(0.0), RC(0.0));
        for (int k = 0; k < n; k++) {
            C term = c_rscale(m->sched_dq[k], c_exp(c_rscale(-m->sched_t[k], p)));
            sum = (k == 0) ? term : c_add(sum, term);
        }
        C num = c_sub(sum, c_rscale(m->sched_sum, c_exp(c_rscale(-m->sched_tf, p))));
        return c_div(num, linear ? c_mul(p, p) : p);                     /* :94 / :119 (p**2 is p*p in flang) */
    }
    switch (m->timeType) {
    case 1: return c_div(c_exp(c_scale(p, -t1)), p);                                       /* :47-49 */
    case 2: return c_sub(c_div(c_exp(c_scale(p, -t1)), p), c_div(c_exp(c_scale(p, -t2)), p)); /* :50-52 */
    case 3: return c_exp(c_scale(p, -t1));                                                 /* :53-55 */
    case 4: {                                                                              /* :56-60 */
        C a = c_rdiv(RC(1.0), c_sub(p, c_mul(p, c_exp(c_scale(p, -t1)))));
        C b = c_rsub(RC(1.0), c_exp(c_scale(p, -t2)));
        return c_div(c_mul(a, b), p);
    }
    case 5: return c_div(c_exp(c_scale(p, -t2)), c_add(p, c_mul(p, c_exp(c_scale(p, -t1)))));  /* :61-64 */
    case 6: return c_div(c_mul(c_exp(c_scale(p, -t2)), p), c_addr(c_mul(p, p), t1 * t1));      /* :65-68 */
    case 7: return c_make(RC(0.0), RC(0.0));                                               /* :69-74 */
    case 8: {                                                                              /* :75-80 */
        C ex = c_exp(c_divr(c_scale(p, -t1), RC(2.0)));
        C num = c_mul(c_exp(c_scale(p, -t2)), c_rsub(RC(1.0), ex));
        return c_div(num, c_mul(c_radd(RC(1.0), ex), p));
    }
    default: return c_make((R)NAN, (R)NAN);
    }
}

/* laplace_hankel_solutions.f90:122-131 */
static inline C theis_s(R a, C p) { return c_rdiv(RC(2.0), c_addr(p, a * a)); }

/* laplace_hankel_solutions.f90:133-202, one (p, z) element; lay = layer of this z */
static C hantush_s(const mdl* m, R a, C p, R zD, int lay)
{
    R dD1 = RC(1.0) - m->dD;
    R lD1 = RC(1.0) - m->lD;
    C eta = c_sqrt(c_divr(c_addr(p, a * a), m->kappa));
    C g1 = c_cosh(c_scale(eta, dD1 - zD));
    C ff1 = c_sinh(c_scale(eta, m->dD));
    C ff2 = c_sinh(c_scale(eta, lD1));
    C sh = c_sinh(eta);
    C g2 = c_div(c_add(c_mul(ff1, c_cosh(c_scale(eta, zD))),
                       c_mul(ff2, c_cosh(c_scale(eta, RC(1.0) - zD)))), sh);
    C udp;
    if (lay == 1) {
        C g3 = c_sub(c_exp(c_neg(c_scale(eta, lD1))),
                     c_div(c_add(ff1, c_mul(c_exp(c_neg(eta)), ff2)), sh));
        udp = c_mul(g3, c_cosh(c_scale(eta, zD)));
    } else if (lay == 2) {
        udp = c_rsub(RC(1.0), g2);
    } else {
        udp = c_sub(g1, g2);
    }
    return c_divr(c_mul(udp, theis_s(a, p)), m->bD);
}

/* laplace_hankel_solutions.f90:204-301, one p, all z.  The reference leaves ff(1:2) unassigned for
 * depths above the screen (zLay = 3, its WHERE masks at :273-276) and then uses them at :278-282 --
 * undefined there; here they are always computed (the Hantush expressions), so parity with the
 * reference is claimed for layers 1 and 2 only. */
static void hantush_storage_p(const mdl* m, R a, C p, int nz, const R* zD, const int* zLay, C* u)
{
    const R dD1 = RC(1.0) - m->dD, lD1 = RC(1.0) - m->lD;
    const R CDw = m->rDw * m->rDw / (RC(2.0) * m->screen * m->Ss);        /* :250 */
    const R tDb = PI_R * (m->rDwobs * m->rDwobs) / (m->sF * m->Ss);       /* :253 */
    const C xi = c_rscale(m->rDw, c_sqrt(p));                             /* :255 */
    const C eta = c_sqrt(c_divr(c_addr(p, a * a), m->kappa));             /* :256 */
    C K0, K1;
    cbesk01_R(xi, &K0, &K1);                                              /* :258-265 */
    const C A0 = c_rdiv(RC(2.0), c_add(c_mul(c_scale(p, CDw), K0), c_mul(xi, K1)));   /* :267 */
    const C uDf = c_div(A0, c_mul(c_addr(p, a * a), c_addr(c_scale(p, tDb), RC(1.0))));   /* :268 */
    const C ff1 = c_sinh(c_scale(eta, m->dD)), ff2 = c_sinh(c_scale(eta, lD1)), sh = c_sinh(eta);
    const C pre = c_divr(uDf, m->bD);
    for (int k = 0; k < nz; k++) {
        C uDp;
        if (zLay[k] == 1) {
            C ff3 = c_sub(c_exp(c_neg(c_scale(eta, lD1))), c_div(c_add(ff1, c_mul(c_exp(c_neg(eta)), ff2)), sh));
            uDp = c_mul(ff3, c_cosh(c_scale(eta, zD[k])));
        } else {
            C g2 = c_div(c_add(c_mul(ff1, c_cosh(c_scale(eta, zD[k]))),
                               c_mul(ff2, c_cosh(c_scale(eta, RC(1.0) - zD[k])))), sh);
            if (zLay[k] == 2) uDp = c_rsub(RC(1.0), g2);
            else uDp = c_sub(c_cosh(c_scale(eta, dD1 - zD[k])), g2);
        }
        u[k] = c_mul(pre, uDp);                                           /* :299 */
    }
}

/* complex Thomas algorithm, utility.f90:96-135, for one p */
static void tridiag_s(int n, const C* a, const C* b, const C* c, const C* v, C* x, C* bp, C* vp)
{
    bp[0] = b[0];
    vp[0] = v[0];
    for (int i = 1; i < n; i++) {
        C mm = c_div(a[i], bp[i - 1]);
        bp[i] = c_sub(b[i], c_mul(mm, c[i - 1]));
        vp[i] = c_sub(v[i], c_mul(mm, vp[i - 1]));
    }
    x[n - 1] = c_div(vp[n - 1], bp[n - 1]);
    for (int i = n - 2; i >= 0; i--)
        x[i] = c_div(c_sub(vp[i], c_mul(c[i], x[i + 1])), bp[i]);
}

/* one p, all z: laplace_hankel_solutions.f90:30-120 */
static int soln_p(const mdl* m, R a, R rD, C p, int nz, const R* zD, const int* zLay, C* fp /*[nz]*/)
{
    const C zero = c_make(RC(0.0), RC(0.0));
    switch (m->model) {
    case 0:
        for (int k = 0; k < nz; k++) fp[k] = theis_s(a, p);
        break;
    case 1:
        for (int k = 0; k < nz; k++) fp[k] = hantush_s(m, a, p, zD[k], zLay[k]);
        break;
    case 2:
        hantush_storage_p(m, a, p, nz, zD, zLay, fp);
        break;
    case 3: case 4: case 5: {                                              /* :64-93 */
        C eta = c_sqrt(c_divr(c_addr(p, a * a), m->kappa));
        C xi = c_div(c_scale(eta, m->alphaD), p);
        if (m->model == 3) {                                               /* :72-75 */
            C sum = zero;
            for (int j = 0; j < m->MoenchM; j++) {
                C t = c_rdiv(RC(1.0), c_radd(RC(1.0), c_scale(p, RC(1.0) / m->MoenchGamma[j])));
                sum = (j == 0) ? t : c_add(sum, t);
            }
            xi = c_div(c_scale(xi, (R)m->MoenchM), sum);
        }
        C top = (m->model == 4) ? theis_s(a, p) : hantush_s(m, a, p, RC(1.0), 3);
        C ce = c_cosh(eta), se = c_sinh(eta);
        for (int k = 0; k < nz; k++) {
            C u = (m->model == 4) ? theis_s(a, p) : hantush_s(m, a, p, zD[k], zLay[k]);
            if (eta.re < m->maxexp) {                                      /* :84-87 */
                C den = c_add(c_mul(c_radd(RC(1.0), c_mul(c_rscale(m->beta, eta), xi)), ce), c_mul(xi, se));
                fp[k] = c_sub(u, c_div(c_mul(top, c_cosh(c_scale(eta, zD[k]))), den));
            } else {                                                       /* :88-91 */
                C den = c_add(c_radd(RC(1.0), c_mul(c_rscale(m->beta, eta), xi)), xi);
                fp[k] = c_sub(u, c_div(c_mul(top, c_exp(c_scale(eta, zD[k] - RC(1.0)))), den));
            }
        }
        break;
    }
    case 6:
        if (m->MNtype == 1) {                                              /* :404-442 */
            R beta0 = m->ak * m->b;
            R phiDa = m->psia / m->b;
            R phiDk = m->psik / m->b;
            R vartheta = beta0 * m->Sy / (m->Ss * m->b) * r_exp(-(beta0 * (phiDa - phiDk)));
            C eta1 = c_sqrt(c_divr(c_addr(c_scale(p, vartheta), a * a), m->kappa));
            R u0 = beta0 / RC(2.0);
            C e1 = c_divr(eta1, u0);
            C v = c_sqrt(c_radd(RC(1.0), c_mul(e1, e1)));
            C u = c_rscale(u0, c_rsub(RC(1.0), v));
            C etasq = c_divr(c_addr(p, a * a), m->kappa);
            C eta = c_sqrt(etasq);
            C Delta0 = c_sub(c_mul(eta, c_sinh(eta)), c_mul(u, c_cosh(eta)));
            C pre = c_rdiv(RC(2.0), c_rscale(m->kappa, etasq));
            C uod = c_div(u, Delta0);
            for (int k = 0; k < nz; k++)
                fp[k] = c_mul(pre, c_radd(RC(1.0), c_mul(uod, c_cosh(c_scale(eta, zD[k])))));
        } else if (m->MNtype == 2) {                                       /* :444-544 */
            int n = m->order;
            R h = m->usLD / (R)(n - 1);
            R invhsq = RC(1.0) / (h * h);
            R beta0 = m->ac * m->Sy / m->Ss;
            R beta1 = -m->lambdaD;
            R beta2 = m->ak * m->b1;
            R beta3 = m->akD;
            C eta = c_sqrt(c_divr(c_radd(a * a, p), m->kappa));
            C sHtop = hantush_s(m, a, p, RC(1.0), 3);
            C B1 = c_divr(c_scale(c_scale(p, beta0), r_exp(-beta2)), m->kappa);
            R B2 = (a * a) / m->kappa;
            C* aa = (C*)calloc((size_t)7 * n, sizeof(C));
            C *bb = aa + n, *cc_ = aa + 2 * n, *vv = aa + 3 * n, *sig = aa + 4 * n, *bp = aa + 5 * n, *vp = aa + 6 * n;
            C omega1 = zero;
            for (int j = 1; j <= n; j++) {
                C om = c_addr(c_scale(B1, r_exp(-(beta1 * (R)(j - 1) * h))), B2);
                if (j == 1) omega1 = om;
                if (j >= 2) bb[j - 1] = c_rsub(beta3 / h - RC(2.0) * invhsq, om);
            }
            C cc = c_rsub(beta3 / h - invhsq, omega1);
            bb[0] = c_rscale(RC(0.5), c_add(c_mul(c_exp(eta), c_sub(cc, c_divr(eta, h))),
                                            c_mul(c_exp(c_neg(eta)), c_add(cc, c_divr(eta, h)))));
            bb[n - 1] = c_subr(c_addr(bb[n - 1], invhsq), beta3 / h);
            for (int j = 1; j <= n - 1; j++) cc_[j - 1] = c_make(invhsq - beta3 / h, RC(0.0));
            for (int j = 2; j <= n; j++) aa[j - 1] = c_make(invhsq, RC(0.0));
            aa[1] = c_mul(aa[1], c_cosh(eta));
            for (int j = 3; j <= n; j++) vv[j - 1] = zero;
            vv[1] = c_rscale(-invhsq, sHtop);
            vv[0] = c_mul(c_neg(cc), sHtop);
            tridiag_s(n, aa, bb, cc_, vv, sig, bp, vp);
            C s1 = sig[0];
            for (int k = 0; k < nz; k++) {
                C sH = hantush_s(m, a, p, zD[k], zLay[k]);
                if (c_abs(s1) > R_TINY) fp[k] = c_add(sH, c_mul(s1, c_cosh(c_scale(eta, zD[k]))));
                else fp[k] = sH;
            }
            free(aa);
        } else {
            return -1;
        }
        break;
    default:
        return -1;
    }
    /* :118  fp = a*bessel_j0(a*rD)*fp*lapTime */
    R aj = a * r_j0(a * rD);
    C lt = lap_time(m, p);
    for (int k = 0; k < nz; k++) fp[k] = c_mul(c_rscale(aj, fp[k]), lt);
    return 0;
}

int ucfo_lap_hank_soln(const ucf_params* P, const ucf_derived* D, double a, double rD,
                       int np, const double* p_re_im, int nz, const double* zD, const int* zLay,
                       double* fp_re_im)
{
    mdl m; mdl_fill(&m, P, D);
    R zr[UCF_MAX_NZ]; C fp[UCF_MAX_NZ];
    if (nz > UCF_MAX_NZ) return -1;
    for (int k = 0; k < nz; k++) zr[k] = (R)zD[k];
    for (int i = 0; i < np; i++) {
        int rc = soln_p(&m, (R)a, (R)rD, ld(p_re_im, i), nz, zr, zLay, fp);
        if (rc) return rc;
        for (int k = 0; k < nz; k++) st(fp_re_im, k * np + i, fp[k]);
    }
    return 0;
}

/* ---------------------------------------------------------- one (t,r) point */
/* body of driver.f90:100-232 (single-point-run semantics) */
static int point_R(const ucf_params* P, const mdl* m, const double* j0z,
                   R tD, R rD, int sv, int nz, const R* zD, const int* zLay,
                   R* hout, R* dhout, ucfo_stage* stage)
{
    const int M = P->M, np = 2 * M + 1;
    const int k = P->k, Rl = P->R, N = (1 << k) - 1;
    const int nacc = P->nacc, ngl = P->ord - 2;
    int rc = 0;

    C* p = (C*)malloc(sizeof(C) * np);
    pvalues_R(RC(2.0) * tD, M, (R)P->alpha, (R)P->tol, p);                /* driver.f90:106 */

    R arg = (R)j0z[sv - 1] / rD;                                          /* :120 */

    /* tanh-sinh tables (driver.f90:121-126,138-151) */
    R* wl = (R*)malloc(sizeof(R) * (size_t)N * Rl);     /* weights per level, level j at wl + j*N */
    R* ab = (R*)malloc(sizeof(R) * N);
    R hv[32];
    int Nv[32];
    for (int j = 1; j <= Rl; j++) {
        int kv = k - Rl + j;
        Nv[j - 1] = (1 << kv) - 1;
        hv[j - 1] = RC(4.0) / (R)(1 << kv);
        tanh_sinh_R(kv, arg, wl + (size_t)(j - 1) * N, (j == Rl) ? ab : NULL);
    }

    C* fa = (C*)malloc(sizeof(C) * (size_t)N * nz * np);
    C* fz = (C*)malloc(sizeof(C) * nz);
    for (int n = 0; n < N && !rc; n++)
        for (int i = 0; i < np; i++) {
            rc = soln_p(m, ab[n], rD, p[i], nz, zD, zLay, fz);
            if (rc) break;
            for (int z = 0; z < nz; z++) fa[((size_t)n * nz + z) * np + i] = fz[z];
        }

    C* tmp = (C*)malloc(sizeof(C) * (size_t)Rl * nz * np);
    C* finint = (C*)malloc(sizeof(C) * (size_t)nz * np);
    C* glarea = (C*)malloc(sizeof(C) * (size_t)nacc * nz * np);
    C* infint = (C*)malloc(sizeof(C) * (size_t)nz * np);
    C* totlap = (C*)malloc(sizeof(C) * (size_t)nz * np);
    R* gx = (R*)malloc(sizeof(R) * (ngl + 2));
    R* gw = (R*)malloc(sizeof(R) * (ngl + 2));
    if (rc) goto cleanup;

    /* driver.f90:135-157 */
    for (int j = 1; j <= Rl; j++) {
        int stride = 1 << (Rl - j);
        const R* w = wl + (size_t)(j - 1) * N;
        for (int z = 0; z < nz; z++)
            for (int i = 0; i < np; i++) {
                C s = c_make(RC(0.0), RC(0.0));
                for (int mm = 1; mm <= Nv[j - 1]; mm++) {
                    int n = mm * stride - 1;
                    C term = c_rscale(w[mm - 1], fa[((size_t)n * nz + z) * np + i]);
                    s = (mm == 1) ? term : c_add(s, term);
                }
                tmp[((size_t)(j - 1) * nz + z) * np + i] = c_rscale(arg / RC(2.0), s);
            }
    }
    /* driver.f90:159-171 */
    for (int z = 0; z < nz; z++)
        for (int i = 0; i < np; i++) {
            if (Rl > 1) {
                C y[32];
                for (int j = 0; j < Rl; j++) y[j] = tmp[((size_t)j * nz + z) * np + i];
                finint[(size_t)z * np + i] = extrap_R(Rl, hv, y);
            } else {
                finint[(size_t)z * np + i] = tmp[(size_t)z * np + i];
            }
        }

    /* driver.f90:179-203 */
    gauss_lobatto_R(P->ord, gx, gw);
    for (int j = sv + 1; j <= sv + nacc; j++) {
        R lob = (R)j0z[j - 2] / rD;
        R hib = (R)j0z[j - 1] / rD;
        R width = hib - lob;
        for (int z = 0; z < nz; z++)
            for (int i = 0; i < np; i++) glarea[((size_t)(j - sv - 1) * nz + z) * np + i] = c_make(RC(0.0), RC(0.0));
        for (int mm = 0; mm < ngl; mm++) {
            R y = (width * gx[mm] + (hib + lob)) / RC(2.0);
            for (int i = 0; i < np; i++) {
                rc = soln_p(m, y, rD, p[i], nz, zD, zLay, fz);
                for (int z = 0; z < nz; z++) {
                    C* acc = &glarea[((size_t)(j - sv - 1) * nz + z) * np + i];
                    C term = c_scale(fz[z], gw[mm]);
                    *acc = (mm == 0) ? term : c_add(*acc, term);
                }
            }
        }
        for (int z = 0; z < nz; z++)
            for (int i = 0; i < np; i++) {
                C* acc = &glarea[((size_t)(j - sv - 1) * nz + z) * np + i];
                *acc = c_rscale(width / RC(2.0), *acc);
            }
    }
    /* driver.f90:205-216 */
    for (int z = 0; z < nz; z++)
        for (int i = 0; i < np; i++) {
            C ser[64];
            int any = 0;
            for (int j = 0; j < nacc; j++) {
                ser[j] = glarea[((size_t)j * nz + z) * np + i];
                if (c_abs(ser[j]) > RC(0.0)) any = 1;
            }
            int wst = 0;
            infint[(size_t)z * np + i] = any ? wynn_R(nacc, ser, &wst) : c_make(RC(0.0), RC(0.0));
            if (!any) cnt_add(5, 1);
            else if (wst >= 1 && wst <= 3) cnt_add(1 + wst, 1);
            totlap[(size_t)z * np + i] = c_add(finint[(size_t)z * np + i], infint[(size_t)z * np + i]);
        }
    /* driver.f90:217-230 */
    {
        R tee = tD * RC(2.0);
        C* tl = (C*)malloc(sizeof(C) * np);
        for (int z = 0; z < nz; z++) {
            hout[z] = dehoog_R(M, (R)P->alpha, (R)P->tol, tD, tee, totlap + (size_t)z * np);
            for (int i = 0; i < np; i++) tl[i] = c_mul(totlap[(size_t)z * np + i], p[i]);
            dhout[z] = dehoog_R(M, (R)P->alpha, (R)P->tol, tD, tee, tl) * tD;
        }
        free(tl);
    }
    if (stage) {
#define DUMP(dst, src, cnt) if (stage->dst) for (size_t q_ = 0; q_ < (size_t)(cnt); q_++) st(stage->dst, (int)q_, (src)[q_])
        DUMP(p, p, np);
        DUMP(fa, fa, (size_t)N * nz * np);
        DUMP(tmp, tmp, (size_t)Rl * nz * np);
        DUMP(finint, finint, (size_t)nz * np);
        DUMP(glarea, glarea, (size_t)nacc * nz * np);
        DUMP(infint, infint, (size_t)nz * np);
        DUMP(totlap, totlap, (size_t)nz * np);
#undef DUMP
    }
cleanup:
    free(p); free(wl); free(ab); free(fa); free(fz); free(tmp); free(finint);
    free(glarea); free(infint); free(totlap); free(gx); free(gw);
    return rc;
}

int ucfo_point(const ucf_params* P, const ucf_derived* D, const double* j0z,
               double tD, double rD, int sv, int nz, const double* zD, const int* zLay,
               double* h, double* dh, ucfo_stage* stage)
{
    mdl m; mdl_fill(&m, P, D);
    R zr[UCF_MAX_NZ], hr[UCF_MAX_NZ], dr[UCF_MAX_NZ];
    if (nz > UCF_MAX_NZ) return -1;
    for (int k = 0; k < nz; k++) zr[k] = (R)zD[k];
    g_count_on = 1;
    int rc = point_R(P, &m, j0z, (R)tD, (R)rD, sv, nz, zr, zLay, hr, dr, stage);
    g_count_on = 0;
    for (int k = 0; k < nz; k++) { h[k] = (double)hr[k]; dh[k] = (double)dr[k]; }
    return rc;
}

int ucfo_batch(const ucf_params* P, int npts, const double* tD, const double* rD, const int* sv,
               int nz, const double* zD, const int* zLay, double* h, double* dh, int threads)
{
    ucf_derived D;
    ucfo_nondim(P, &D);
    double* j0z = (double*)malloc(sizeof(double) * D.nj0z);
    ucfo_j0_zeros(D.nj0z, j0z);
    int rc = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int i = 0; i < npts; i++) {
        int r = ucfo_point(P, &D, j0z, tD[i], rD[i], sv[i], nz, zD, zLay, h + (size_t)i * nz, dh + (size_t)i * nz, NULL);
        if (r) {
#pragma omp atomic write
            rc = r;
        }
    }
    free(j0z);
    return rc;
}
