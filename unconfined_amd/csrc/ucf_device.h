// ucf_device.h -- the hot path on the device.  Included twice by ucf_kernels.hip
// (UCF_FAST = 0 -> namespace ucf_faithful, UCF_FAST = 1 -> namespace ucf_fast).
//
// Mapping: a 64-lane wavefront walks the Hankel abscissae of its work item serially, so the quadrature sums keep
// the reference's summation order and need no cross-lane traffic.  What the 64 lanes are depends on the lane
// layout (decode_item below): 64 consecutive TIMES of one radius and one Laplace index (grids, LAYOUT 1), 64 POINTS
// of a list and one Laplace index (LAYOUT 3), or the Laplace samples p_m of one point (LAYOUT 0; chunks of 64 of
// them for 2M+1 > 64, LAYOUT 2).  The abscissa loop has its own kernel (integrate_kernel with the fast evaluators of
// ucf_fastpath.h -- every model family has one; integrate_generic_kernel with the reference-order evaluators in the faithful
// flavour): level sums in LDS ([slot][lane] complex, 16 B per lane -> conflict-free b128 accesses); when the loop is done the
// level sums and the finished J0-interval areas go to the work item's STATE in HBM (1 KB coalesced per slot, ~15 KB
// per item on the C2 settings -- deliberate traffic that buys the loop its occupancy).  What depends on
// the time and the Laplace index only -- the pumping-schedule multiplier lapTime(p) and the constants the fast evaluators
// leave out of their samples -- comes from laptime_kernel, one value per (time, p) instead of one evaluation per work item.
// finish_kernel reads the state back (Richardson/Neville and Wynn-epsilon per lane), writes the accelerated
// transform totlap(r, z, m, t) (16 B each) and dehoog_*_kernel inverts it with lane = Laplace index: only de Hoog's
// quotient-difference table needs neighbours (lane i+1: one DPP move per dword).  The abscissa tables (a, a J0(a rD)
// per radius, written by abscissa_kernel) and the quadrature tables (< 10 KB) are read through the scalar cache.
//
// Every device function cites the reference lines it restates.

#include "ucf_math.h"
#include "ucf_plan.h"
#include <cstdlib>

namespace UCF_NS {

using namespace ucfm;

#define UCF_STR_(x) #x
#define UCF_STR(x) UCF_STR_(x)
#define UCF_PI 3.14159265358979323846   /* 4*atan(1) in binary64 */
#define UCF_EPS 2.220446049250313e-16

// ------------------------------------------------------------------ wave helpers
UCF_DEV double shfl_down1(double v) { return __shfl_down(v, 1, 64); }
// lane i <- lane i+1 (lane 63 keeps its own value, like __shfl_down): one DPP move per dword
// (wave_shl:1) instead of a ds_bpermute round trip through the LDS crossbar
UCF_DEV double shift_down1(double x)
{
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int l2 = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    const int h2 = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(h2, l2);
}
UCF_DEV cplx shfl_down1(cplx v) { return cmake(shift_down1(v.re), shift_down1(v.im)); }
UCF_DEV cplx bcast0(cplx v) { return cmake(__shfl(v.re, 0, 64), __shfl(v.im, 0, 64)); }
UCF_DEV double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
UCF_DEV void stat_add(long long* ctr, bool pred)
{
    if (ctr == nullptr) return;
    unsigned long long bal = __ballot(pred);
    if (bal != 0ull && (threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)ctr, (unsigned long long)__popcll(bal));
}

// Translation-unit split of the fast flavour (build time): UCF_TU = 0 / 1 / 2 compiles the launchers and the plain
// kernels of one lane layout only; undefined (-1): everything.
#ifndef UCF_TU
#define UCF_TU -1
#endif
#define UCF_TU_HAS(L) (UCF_TU < 0 || UCF_TU == (L))

typedef double2 lds_c;   // one complex per lane per slot
UCF_DEV cplx lds_ld(const lds_c* base, int slot, int lane) { lds_c v = base[slot * UCF_WAVE + lane]; return cmake(v.x, v.y); }
UCF_DEV void lds_st(lds_c* base, int slot, int lane, cplx z) { base[slot * UCF_WAVE + lane] = make_double2(z.re, z.im); }
// scratch columns hold UCF_PART lanes per slot: the per-lane tails (Neville, Wynn) run on one
// quarter-wave at a time, which quarters their LDS footprint at ~2 % of the point's time
#define UCF_PART 16
// (finish_kernel picks wider parts when its LDS budget allows: PART lanes work at a time, 64/PART turns)
template <int PART = UCF_PART>
UCF_DEV cplx scr_ld(const lds_c* base, int slot, int lane) { lds_c v = base[slot * PART + (lane & (PART - 1))]; return cmake(v.x, v.y); }
template <int PART = UCF_PART>
UCF_DEV void scr_st(lds_c* base, int slot, int lane, cplx z) { base[slot * PART + (lane & (PART - 1))] = make_double2(z.re, z.im); }

#if UCF_FAST
// 1/z without exponent scaling (|z| well inside [1e-150, 1e150])
UCF_DEV cplx cinv_plain_(cplx z)
{
    const double r = fast_rcp(z.re * z.re + z.im * z.im);
    return cmake(z.re * r, -(z.im * r));
}
#endif

// ------------------------------------------------------------------ time.f90:34-80
UCF_DEV cplx lap_time(const ucf_dev_params& P, cplx p)
{
    if (P.timeType < 0) {                                                                        // :81-95, :97-122
        const bool linear = P.timeType <= -101;          // piecewise linear: slope increments, and p^2 below
        const int n = linear ? -P.timeType - 100 : -P.timeType;
        cplx sum = cmake(0.0, 0.0);
        for (int k = 0; k < n; k++) {
            const cplx term = rscale(P.sched[n + k], cexp_(rscale(-P.sched[k], p)));
            sum = (k == 0) ? term : cadd(sum, term);
        }
        const cplx num = csub(sum, rscale(P.sched[2 * n + 1], cexp_(rscale(-P.sched[2 * n], p))));
        return cdiv(num, linear ? cmul(p, p) : p);                                               // :92-94 / :117-119
    }
    const double t1 = P.timePar[0], t2 = P.timePar[1];
    switch (P.timeType) {
    case 1: return cdiv(cexp_(cscale(p, -t1)), p);                                               // :47-49
    case 2: return csub(cdiv(cexp_(cscale(p, -t1)), p), cdiv(cexp_(cscale(p, -t2)), p));         // :50-52
    case 3: return cexp_(cscale(p, -t1));                                                        // :53-55
    case 4: {                                                                                    // :56-60
        cplx a = rdiv(1.0, csub(p, cmul(p, cexp_(cscale(p, -t1)))));
        cplx b = rsub(1.0, cexp_(cscale(p, -t2)));
        return cdiv(cmul(a, b), p);
    }
    case 5: return cdiv(cexp_(cscale(p, -t2)), cadd(p, cmul(p, cexp_(cscale(p, -t1)))));         // :61-64
    case 6: return cdiv(cmul(cexp_(cscale(p, -t2)), p), caddr(cmul(p, p), t1 * t1));             // :65-68
    case 7: return cmake(0.0, 0.0);                                                              // :69-74 (quirk Q4)
    case 8: {                                                                                    // :75-80
        cplx ex = cexp_(cdivr(cscale(p, -t1), 2.0));
        cplx num = cmul(cexp_(cscale(p, -t2)), rsub(1.0, ex));
        return cdiv(num, cmul(radd(1.0, ex), p));
    }
    default: return cmake(__builtin_nan(""), __builtin_nan(""));
    }
}

// ------------------------------------------------------------------ cbessel.f90:877,5036
// K0(z), K1(z) for Re z >= 0 by the Amos algorithm (cbesk -> cbknu with fnu = 0, n = 2, kode = 1):
// power series for |z| <= 2 (:5098-5201), Miller backward recurrence beyond (:5209-5327).
// Evaluated once per lane and point (z = rDw*sqrt(p) does not depend on the abscissa).
struct k01 { cplx k0, k1; int ierr; };
UCF_DEV cplx clog_(cplx z) { return cmake(log(hypot(z.re, z.im)), atan2(z.im, z.re)); }

__device__ __noinline__ static k01 cbesk01_(cplx z)
{
    k01 out;
    out.k0 = cmake(0.0, 0.0); out.k1 = out.k0; out.ierr = 0;
    const double tol = UCF_EPS;
    const double elim = 2.303 * (1021.0 * log10(2.0) - 3.0);
    const double alim = elim + fmax(-(log10(2.0) * 52.0 * 2.303), -41.45);
    const double at1 = 0.78539816339744830962;
    const double pi = 4.0 * at1, hpi = 2.0 * at1, spi = 3.0 / (2.0 * at1);
    const double rthpi = sqrt(8.0 * at1) / 2.0, fpi = 1.89769999331517738, tth = 6.66666666666666666e-01;
    const double gamma_e = 5.77215664901532861e-01;
    const double xx = z.re, yy = z.im;
    if (xx == 0.0 && yy == 0.0) { out.ierr = 1; return out; }
    const double caz = hypot(xx, yy);
    if (caz > 1073741823.5) { out.ierr = 4; return out; }
    if (caz < UCF_DBL_MIN * 1.0e3) { out.ierr = 2; return out; }
    if (!(caz > tol) && -log(0.5 * caz) > elim) { out.ierr = 2; return out; }
    const cplx rz = cdiv(cmake(2.0, 0.0), z);                                                   // :5085
    cplx s1, s2;
    if (caz <= 2.0) {
        const cplx smu = clog_(rz);
        cplx f = cadd(rscale(-gamma_e, cmake(1.0, 0.0)), cscale(smu, 1.0));                     // :5137
        cplx p = cmake(0.5, 0.0), q = cmake(0.5, 0.0);
        s1 = f; s2 = p;
        double ak = 1.0, a1 = 1.0, bk = 1.0;
        cplx ck = cmake(1.0, 0.0);
        if (caz >= tol) {
            const cplx cz = cscale(cmul(z, z), 0.25);
            const double t1 = 0.25 * caz * caz;
            do {                                                                                // :5172-5182
                f = cdivr(cadd(cadd(cscale(f, ak), p), q), bk);
                p = cdivr(p, ak);
                q = cdivr(q, ak);
                const double rk = 1.0 / ak;
                ck = cscale(cmul(ck, cz), rk);
                s1 = cadd(s1, cmul(ck, f));
                s2 = cadd(s2, cmul(ck, csub(p, cscale(f, ak))));
                a1 = a1 * t1 * rk;
                bk = bk + ak + ak + 1.0;
                ak = ak + 1.0;
            } while (a1 > tol);
        }
        out.k0 = s1;
        out.k1 = cmul(s2, rz);
        return out;
    }
    if (xx > alim) { out.ierr = 2; return out; }
    cplx coef = cdiv(cmake(rthpi, 0.0), csqrt_(z));
    {
        double sn, cs;
        sincos_(yy, &sn, &cs);
        coef = cmul(coef, rscale(exp(-xx), cmake(cs, -sn)));
    }
    double ak = fabs(cos(pi * 0.0));
    double fhs = 0.25;
    double t1 = 52.0 * log10(2.0) * 3.321928094;
    t1 = fmin(fmax(t1, 12.0), 60.0);
    const double t2 = tth * t1 - 6.0;
    t1 = (xx == 0.0) ? hpi : fabs(atan(yy / xx));
    double fk;
    if (t2 <= caz) {
        const double etest = ak / (pi * caz * tol);
        fk = 1.0;
        if (!(etest < 1.0)) {
            double fks = 2.0, rk = caz + caz + 2.0, a1 = 0.0, a2 = 1.0;
            bool found = false;
            for (int i = 1; i <= 30 && !found; i++) {
                ak = fhs / fks;
                const double bk = rk / (fk + 1.0);
                double tm = a2;
                a2 = bk * a2 - ak * a1;
                a1 = tm;
                rk = rk + 2.0;
                fks = fks + fk + fk + 2.0;
                fhs = fhs + fk + fk;
                fk = fk + 1.0;
                tm = fabs(a2) * fk;
                if (etest < tm) found = true;
            }
            if (!found) { out.ierr = 5; return out; }
            fk = fk + spi * t1 * sqrt(t2 / caz);
            fhs = 0.25;
        }
    } else {
        const double a2 = sqrt(caz);
        ak = fpi * ak / (tol * sqrt(a2));
        const double aa = 3.0 * t1 / (1.0 + caz);
        const double bb = 14.7 * t1 / (28.0 + caz);
        ak = (log(ak) + caz * cos(aa) / (1.0 + 0.008 * caz)) / cos(bb);
        fk = 0.12125 * ak * ak / caz + 1.5;
    }
    const int k = (int)fk;
    fk = (double)k;
    double fks = fk * fk;
    cplx p1 = cmake(0.0, 0.0), p2 = cmake(tol, 0.0), cs = p2;
    for (int i = 1; i <= k; i++) {                                                              // :5279-5291
        const double a1 = fks - fk;
        const double a2 = (fks + fk) / (a1 + fhs);
        const double rk = 2.0 / (fk + 1.0);
        const cplx pt = p2;
        p2 = cscale(csub(cmul(p2, cmake((fk + xx) * rk, yy * rk)), p1), a2);
        p1 = pt;
        cs = cadd(cs, p2);
        fks = a1 - fk + 1.0;
        fk = fk - 1.0;
    }
    cplx pt = cmake(1.0 / hypot(cs.re, cs.im), 0.0);
    s1 = cmul(pt, p2);
    cs = cmul(cmake(cs.re, -cs.im), pt);
    s1 = cmul(cmul(coef, s1), cs);
    pt = cmake(1.0 / hypot(p2.re, p2.im), 0.0);
    p1 = cmul(pt, p1);
    p2 = cmul(cmake(p2.re, -p2.im), pt);
    pt = cmul(p1, p2);
    s2 = cmul(s1, cadd(cmake(1.0, 0.0), cdiv(csub(cmake(0.5, 0.0), pt), z)));
    out.k0 = s1;
    out.k1 = s2;
    return out;
}

// A0(p) of the Hantush-with-storage solution (laplace_hankel_solutions.f90:255-267), per lane
UCF_DEV cplx hstorage_A0(const ucf_dev_params& P, cplx p)
{
    const cplx xi = rscale(P.hs_rDw, csqrt_(p));                                                // :255
    const k01 K = cbesk01_(xi);                                                                 // :258-265
    return rdiv(2.0, cadd(cmul(cscale(p, P.hs_CDw), K.k0), cmul(xi, K.k1)));                    // :267
}

// ------------------------------------------- laplace_hankel_solutions.f90:30-120
// z-independent part of one (a,p) sample
struct sample_common {
    cplx th;        // theis: 2/(p+a^2)                                   (:122-131)
    cplx eta;       // sqrt((p+a^2)/kappa)                                (:69,172)
    cplx ff1, ff2;  // sinh(eta*dD), sinh(eta*lD1)                        (:176-177)
    cplx she, che;  // sinh(eta), cosh(eta)
    cplx g3;        // below-screen factor                                (:183-184)
    cplx top;       // Hantush value at the water table zD=1 (layer 3)    (:81,162-170)
    cplx den;       // water-table closure denominator                    (:86-87 or :90-91)
    cplx mn_pre, mn_uod;   // Mishra/Neuman-Malama: 2/(kappa*etasq), u/Delta0  (:437-439)
    cplx fd_s1;     // Mishra/Neuman FD: sigma(1) = A1                     (:517-523)
    cplx hs_pre;    // Hantush+storage: uDf/bD                             (:268,299)
    bool small_eta; // Re(eta) < MAXEXP                                   (:84)
    bool fd_use;    // |sigma1| > tiny                                    (:521)
    bool clean3;    // fast flavour, resumed items: depths above the screen top take the cancellation-free form (lay3_udp_scaled)
};

// laplace_hankel_solutions.f90:133-202 for one depth; `lay` = layer of this depth
UCF_DEV cplx hantush_z(const ucf_dev_params& P, const sample_common& S, double zD, int lay, cplx chz)
{
    cplx udp;
    if (lay == 1) {
        udp = cmul(S.g3, chz);                                                                  // :188
    } else {
        cplx ch1z = ccosh_(cscale(S.eta, 1.0 - zD));
        cplx g2 = cdiv(cadd(cmul(S.ff1, chz), cmul(S.ff2, ch1z)), S.she);                       // :179-180
        if (lay == 2) {
            udp = rsub(1.0, g2);                                                                // :192
        } else {
            cplx g1 = ccosh_(cscale(S.eta, P.dD1 - zD));                                        // :175
            udp = csub(g1, g2);                                                                 // :196
        }
    }
    return cdivr(cmul(udp, S.th), P.bD);                                                        // :200
}

#if UCF_FAST
// 1 / B_2 of the finite-difference elimination (laplace_hankel_solutions.f90:501-509 with utility.f90:96-135 run from the
// bottom node up; only x(1) is wanted, :521-523).  B_i = b_i - K / B_{i+1} is a Moebius map per node; with
// B_{i+1} = N_{i+1} / N_{i+2} it becomes the three-term recurrence N_i = b_i N_{i+1} - K N_{i+2} (N_{n+1} = 1, N_n = b_n):
// products only and ONE complex reciprocal at the end instead of one per node (8 instead of 14 instructions per node).
// Scaled by c >= |b_i| per lane so that nothing over- or underflows (beta_i = b_i / c has modulus <= 1, K / c^2 <= 1/4);
// the recurrence runs in the direction in which its dominant solution grows, i.e. stably.
// b_i = bmid - B1 e_i (i < n), b_n = b_i + invhsq - b3h  (:501-502).
// With K > 0 the substitution N_i = K^((n+1-i)/2) M_i makes the second coefficient -1: M_i = (b_i / sqrt K) M_{i+1} - M_{i+2},
// 6 instructions per node (the K-term folds into the inner fma), 1 / B_2 = M_3 / (sqrt(K) M_2).  |M_i| <= (g + 1)^n with
// g >= |b_i| / sqrt K: taken when (g + 1)^n < 2^500 in every lane of the wave (|M|^2 is formed at the end); the per-lane
// scaled form otherwise.
template <bool UNIT>
UCF_DEV cplx fd_chain(const ucf_dev_params& P, cplx B1c, double bmc, double bn_extra, double k2, cplx* N2_out)
{
    const int n = P.order;
    cplx N2 = cmake(1.0, 0.0);                                                                       // N_{n+1}
    cplx N1 = cmake(__builtin_fma(B1c.re, P.fd_e[n - 1], bmc + bn_extra), B1c.im * P.fd_e[n - 1]);   // N_n = b_n (scaled)
    // one node: N_i = beta_i N_{i+1} + k2 N_{i+2}, beta_i = scaled b_i = (bmc + B1c.re e_i, B1c.im e_i)            (:501)
    auto node = [&](double e, cplx Na, cplx Nb) {
        const double br = __builtin_fma(B1c.re, e, bmc), bi = B1c.im * e;
        if (UNIT) return cmake(__builtin_fma(br, Na.re, __builtin_fma(-bi, Na.im, -Nb.re)),
                               __builtin_fma(br, Na.im, __builtin_fma(bi, Na.re, -Nb.im)));
        return cmake(__builtin_fma(br, Na.re, __builtin_fma(-bi, Na.im, k2 * Nb.re)),
                     __builtin_fma(br, Na.im, __builtin_fma(bi, Na.re, k2 * Nb.im)));
    };
    // four nodes per trip: their four table entries are adjacent (one scalar load, one wait for it instead of four) and
    // the three-term recurrence rotates through its registers without copies
    int i = n - 1;
    for (; i >= 5; i -= 4) {
        const double e0 = P.fd_e[i - 1], e1 = P.fd_e[i - 2], e2 = P.fd_e[i - 3], e3 = P.fd_e[i - 4];
        const cplx Na = node(e0, N1, N2);
        const cplx Nb = node(e1, Na, N1);
        N2 = node(e2, Nb, Na);
        N1 = node(e3, N2, Nb);
    }
    for (; i >= 2; i--) {
        const cplx N0 = node(P.fd_e[i - 1], N1, N2);
        N2 = N1;
        N1 = N0;
    }
    *N2_out = N2;
    return N1;
}
UCF_DEV cplx fd_inverse_B2(const ucf_dev_params& P, cplx B1, double bmid, double invhsq, double b3h, double K)
{
    const int n = P.order;
    const double emax = fmax(P.fd_e[0], P.fd_e[n - 1]);
    const double bsum = fabs(bmid) + fabs(invhsq - b3h) + (fabs(B1.re) + fabs(B1.im)) * emax;        // >= |b_i|
    cplx N1, N2;
    double r;
    if (P.fd_isk > 0.0 && __builtin_amdgcn_ballot_w64(!(bsum * P.fd_isk < P.fd_gmax)) == 0) {
        const double s = P.fd_isk;
        N1 = fd_chain<true>(P, cmake(-(B1.re * s), -(B1.im * s)), bmid * s, (invhsq - b3h) * s, -1.0, &N2);
        // 1 / B_2 = M_3 / (sqrt(K) M_2) = M_3 conj(M_2) / (sqrt(K) |M_2|^2)
        r = fast_rcp(N1.re * N1.re + N1.im * N1.im) * s;
    } else {
        const double ic = fast_rcp(bsum);                                                             // 1 / c
        N1 = fd_chain<false>(P, cmake(-(B1.re * ic), -(B1.im * ic)), bmid * ic, (invhsq - b3h) * ic, -(K * ic) * ic, &N2);
        // B_2 = c N_2 / N_3  ->  1 / B_2 = N_3 / (c N_2) = N_3 conj(N_2) / (c |N_2|^2)
        r = fast_rcp(N1.re * N1.re + N1.im * N1.im) * ic;
    }
    return cmake((N2.re * N1.re + N2.im * N1.im) * r, (N2.im * N1.re - N2.re * N1.im) * r);
}
#endif

#if UCF_FAST
// g1 - g2 of the Hantush factor at a depth ABOVE THE SCREEN TOP (layer 3, laplace_hankel_solutions.f90:175-180,196) for
// any eta, in decaying exponentials.  The reference subtracts two terms of size e^{eta c}/2, c = zD - dD1, that agree to
// e^{-2 eta c}: its value carries rounding noise u e^{+eta c} which overtakes the value itself at eta c = 18 and keeps
// growing.  With cosh A sinh B = [sinh(A+B) - sinh(A-B)]/2 the large parts cancel analytically,
//   (g1 - g2) sinh(eta) = [sinh(eta A1) + sinh(eta A2) - sinh(eta A3) - sinh(eta A4)] / 2,
//   A1 = 2 - zD - dD,  A2 = zD - dD,  A3 = lD1 + 1 - zD,  A4 = lD1 - 1 + zD      (all in [-1, 1]),
// and sinh(eta A) / sinh(eta) = e^{-eta (1 - A)} (1 - e^{-2 eta A}) / (1 - e^{-2 eta}) neither overflows nor cancels.
// The fast evaluators use the cosh/sinh form of the same identity inside their range (ucf_fastpath.h); this one serves the
// items that point_kernel resumes beyond it, so that a series never changes from clean values to that noise in mid-course
// (tools/fuzz_hunt.py: errors of 3e-2 ... 5 in h where the reference itself is good to 1e-9).
UCF_DEV cplx sinh_ratio_(cplx eta, double A, cplx one_minus_E2)
{
    const double a = fabs(A);
    const cplx e1 = cexp_(cscale(eta, -(1.0 - a)));
    const cplx e2 = cexp_(cscale(eta, -2.0 * a));
    const cplx r = cmul(e1, cdiv(rsub(1.0, e2), one_minus_E2));
    return (A < 0.0) ? cneg(r) : r;
}
UCF_DEV cplx lay3_udp_scaled(const ucf_dev_params& P, cplx eta, double zD)
{
    const cplx omE2 = rsub(1.0, cexp_(cscale(eta, -2.0)));
    const cplx t1 = sinh_ratio_(eta, 2.0 - zD - P.dD, omE2), t2 = sinh_ratio_(eta, zD - P.dD, omE2);
    const cplx t3 = sinh_ratio_(eta, P.lD1 + 1.0 - zD, omE2), t4 = sinh_ratio_(eta, P.lD1 - 1.0 + zD, omE2);
    return rscale(0.5, csub(cadd(t1, t2), cadd(t3, t4)));
}
// the reference's value where it is not finite (its in-band rules key on that), else the clean one
UCF_DEV cplx lay3_pick(cplx reference_value, cplx clean_value) { return c_is_finite(reference_value) ? clean_value : reference_value; }
#endif

template <int FAMILY>   // 0 Theis, 1 Hantush, 2 water-table (models 3,4,5), 3 MN-Malama, 4 MN-FD, 5 Hantush+storage
UCF_DEV void sample_prepare(const ucf_dev_params& P, double a, cplx p, bool need_lay1, sample_common& S,
                            lds_c* fdbuf, int lane, cplx lane_aux)
{
    const double a2 = a * a;
    const cplx q = caddr(p, a2);
    if (FAMILY == 5) {                                                                          // :204-301
        // lane_aux = A0(p);  uDf = A0/((p+a^2)(p*tDb+1)) (:268);  u = (uDf/bD)*uDp (:299)
        const cplx uDf = cdiv(lane_aux, cmul(q, caddr(cscale(p, P.hs_tDb), 1.0)));
        S.hs_pre = cdivr(uDf, P.bD);
        S.eta = csqrt_(cdivr(q, P.kappa));
        S.ff1 = csinh_(cscale(S.eta, P.dD));
        S.ff2 = csinh_(cscale(S.eta, P.lD1));
        S.she = csinh_(S.eta);
        if (need_lay1)
            S.g3 = csub(cexp_(cneg(cscale(S.eta, P.lD1))),
                        cdiv(cadd(S.ff1, cmul(cexp_(cneg(S.eta)), S.ff2)), S.she));             // :284-287
        return;
    }
    S.th = rdiv(2.0, q);
    if (FAMILY == 0) return;
    if (FAMILY == 3) {                                                                          // :404-442
        cplx eta1 = csqrt_(cdivr(caddr(cscale(p, P.mn_vartheta), a2), P.kappa));
        cplx e1 = cdivr(eta1, P.mn_u0);
        cplx v = csqrt_(radd(1.0, cmul(e1, e1)));
        cplx u = rscale(P.mn_u0, rsub(1.0, v));
        cplx etasq = cdivr(q, P.kappa);
        S.eta = csqrt_(etasq);
        cplx Delta0 = csub(cmul(S.eta, csinh_(S.eta)), cmul(u, ccosh_(S.eta)));
        S.mn_pre = rdiv(2.0, rscale(P.kappa, etasq));
        S.mn_uod = cdiv(u, Delta0);
        return;
    }
    S.eta = csqrt_(cdivr(q, P.kappa));
    const bool hantush_needed = !(FAMILY == 2 && P.model == 4);
    if (hantush_needed) {
        S.ff1 = csinh_(cscale(S.eta, P.dD));
        S.ff2 = csinh_(cscale(S.eta, P.lD1));
        S.she = csinh_(S.eta);
        if (need_lay1)
            S.g3 = csub(cexp_(cneg(cscale(S.eta, P.lD1))),
                        cdiv(cadd(S.ff1, cmul(cexp_(cneg(S.eta)), S.ff2)), S.she));             // :183-184
    }
    if (FAMILY == 1) return;
    S.che = ccosh_(S.eta);
    // water-table value of the confined solution: theis (model 4) or hantush at zD = 1, layer 3
    if (FAMILY == 2 && P.model == 4) {
        S.top = S.th;                                                                           // :78-79
    } else {
        // cosh(eta*1) == cosh(eta);  cosh(eta*(1-1)) = cosh((+0, +-0)) = (1, (+0)*(+-0))
        const cplx ch0 = cmake(1.0, 0.0 * (S.eta.im * 0.0));
        cplx g2 = cdiv(cadd(cmul(S.ff1, S.che), cmul(S.ff2, ch0)), S.she);
        cplx g1 = ccosh_(cscale(S.eta, P.dD1 - 1.0));
        S.top = cdivr(cmul(csub(g1, g2), S.th), P.bD);
    }
    if (FAMILY == 2) {
        cplx xi = cdiv(cscale(S.eta, P.alphaD), p);                                             // :70
        if (P.model == 3) {                                                                     // :72-75
            cplx sum = cmake(0.0, 0.0);
            for (int j = 0; j < P.MoenchM; j++) {
                cplx t = rdiv(1.0, radd(1.0, cscale(p, P.MoenchInvGamma[j])));
                sum = (j == 0) ? t : cadd(sum, t);
            }
            xi = cdiv(cscale(xi, (double)P.MoenchM), sum);
        }
        S.small_eta = S.eta.re < P.maxexp;                                                      // :84
        cplx one_bex = radd(1.0, cmul(rscale(P.beta, S.eta), xi));
        if (S.small_eta) {
            S.she = hantush_needed ? S.she : csinh_(S.eta);
            S.den = cadd(cmul(one_bex, S.che), cmul(xi, S.she));                                // :86-87
        } else {
            S.den = cadd(one_bex, xi);                                                          // :90-91
        }
        return;
    }
    if (FAMILY == 4) {                                                                          // :444-544
        // FD system in the vadose zone; complex Thomas (utility.f90:96-135) per lane.
        // bp/vp of the forward sweep are kept in this lane's LDS column fdbuf[2*order].
        const int n = P.order;
        const double h = P.fd_h, invhsq = P.fd_invhsq, b3h = P.fd_beta3 / h;
        const cplx B1 = cdivr(cscale(cscale(p, P.fd_beta0), P.fd_expmb2), P.kappa);            // :492
        const double B2 = a2 / P.kappa;                                                         // :493
        const cplx om1 = caddr(cscale(B1, P.fd_e[0]), B2);                                      // :495
        const cplx cc = rsub(b3h - invhsq, om1);                                                // :498
        const cplx eoh = cdivr(S.eta, h);
        const cplx b1 = rscale(0.5, cadd(cmul(cexp_(S.eta), csub(cc, eoh)),
                                         cmul(cexp_(cneg(S.eta)), cadd(cc, eoh))));             // :499-500
        const double csup = invhsq - b3h;                                                       // :505
        const cplx a2v = cmul(cmake(invhsq, 0.0), S.che);                                       // :508-509
        const cplx v1 = cmul(cneg(cc), S.top);                                                  // :514
        const cplx v2 = rscale(-invhsq, S.top);                                                 // :513
#if UCF_FAST
        // fast flavour: only x(1) is needed (:521-523) and v(3:n) = 0, so eliminate from the bottom up --
        // a continued fraction B_i = b_i - c_i a_{i+1} / B_{i+1} with nothing to store (same system,
        // exact arithmetic gives the same x(1) as the forward Thomas sweep of utility.f90:96-135)
        (void)fdbuf; (void)lane;
        const double K = csup * invhsq;
        const cplx iB2 = fd_inverse_B2(P, B1, b3h - 2.0 * invhsq - B2, invhsq, b3h, K);     // 1/B_2
        const cplx B1p = csub(b1, cmul(rscale(csup, a2v), iB2));        // B_1 = b_1 - c_1 a_2 / B_2
        const cplx v1p = csub(v1, cmul(rscale(csup, iB2), v2));         // v'_1 = v_1 - (c_1/B_2) v_2
        cplx x = cdiv(v1p, B1p);
#else
        // forward sweep
        cplx bp = b1, vp = v1;
        lds_st(fdbuf, 0, lane, bp);
        lds_st(fdbuf, n, lane, vp);
        for (int i = 2; i <= n; i++) {
            cplx bi = rsub(b3h - 2.0 * invhsq, caddr(cscale(B1, P.fd_e[i - 1]), B2));           // :501
            if (i == n) bi = csubr(caddr(bi, invhsq), b3h);                                     // :502
            cplx ai = (i == 2) ? a2v : cmake(invhsq, 0.0);
            cplx vi = (i == 2) ? v2 : cmake(0.0, 0.0);
            cplx mm = cdiv(ai, bp);
            bp = csub(bi, cmul(mm, cmake(csup, 0.0)));
            vp = csub(vi, cmul(mm, vp));
            lds_st(fdbuf, i - 1, lane, bp);
            lds_st(fdbuf, n + i - 1, lane, vp);
        }
        // back substitution down to x(1)
        cplx x = cdiv(vp, bp);
        for (int i = n - 1; i >= 1; i--) {
            cplx bpi = lds_ld(fdbuf, i - 1, lane);
            cplx vpi = lds_ld(fdbuf, n + i - 1, lane);
            x = cdiv(csub(vpi, cmul(cmake(csup, 0.0), x)), bpi);
        }
#endif
        S.fd_s1 = x;
        S.fd_use = cabs_(x) > UCF_DBL_MIN;                                                      // :521
    }
}

template <int FAMILY>
UCF_DEV cplx sample_z(const ucf_dev_params& P, const sample_common& S, int iz)
{
    if (FAMILY == 0) return S.th;
    const double zD = P.zD[iz];
    const int lay = P.zLay[iz];
    if (FAMILY == 3) {
        return cmul(S.mn_pre, radd(1.0, cmul(S.mn_uod, ccosh_(cscale(S.eta, zD)))));            // :437-439
    }
    const cplx chz = ccosh_(cscale(S.eta, zD));
    if (FAMILY == 5) {
        cplx uDp;
        if (lay == 1) {
            uDp = cmul(S.g3, chz);                                                              // :293
        } else {
            const cplx g2 = cdiv(cadd(cmul(S.ff1, chz), cmul(S.ff2, ccosh_(cscale(S.eta, 1.0 - zD)))), S.she);   // :278-282
            uDp = (lay == 2) ? rsub(1.0, g2) : csub(ccosh_(cscale(S.eta, P.dD1 - zD)), g2);    // :296 / :290
#if UCF_FAST
            if (S.clean3 && lay == 3) uDp = lay3_pick(uDp, lay3_udp_scaled(P, S.eta, zD));       // (resumed items: see lay3_udp_scaled)
#endif
        }
        return cmul(S.hs_pre, uDp);                                                             // :299
    }
#if UCF_FAST
    // (resumed items of the fast flavour: a depth above the screen top takes the cancellation-free forms of its Hantush
    //  factor and of the water-table value the closures are built on; see lay3_udp_scaled)
    const bool clean3 = S.clean3 && lay == 3 && !(FAMILY == 2 && P.model == 4);
#else
    const bool clean3 = false;
#endif
    cplx uH = cmake(0.0, 0.0), top = S.top, fd_s1 = S.fd_s1;
    if (!(FAMILY == 2 && P.model == 4)) uH = hantush_z(P, S, zD, lay, chz);
#if UCF_FAST
    if (clean3) {
        uH = lay3_pick(uH, cdivr(cmul(lay3_udp_scaled(P, S.eta, zD), S.th), P.bD));
        if (FAMILY != 1) {
            const cplx topc = lay3_pick(S.top, cdivr(cmul(lay3_udp_scaled(P, S.eta, 1.0), S.th), P.bD));
            if (FAMILY == 4 && S.fd_use && c_is_finite(S.top) && cabs_(S.top) > 0.0) fd_s1 = cmul(S.fd_s1, cdiv(topc, S.top));   // sigma(1) is linear in top (:513-523)
            top = topc;
        }
    }
#endif
    if (FAMILY == 1) return uH;
    if (FAMILY == 2) {
        const cplx u = (P.model == 4) ? S.th : uH;
        if (S.small_eta) return csub(u, cdiv(cmul(top, chz), S.den));                           // :85-87
        return csub(u, cdiv(cmul(top, cexp_(cscale(S.eta, zD - 1.0))), S.den));                 // :89-91
    }
    // FAMILY 4
    if (S.fd_use) return cadd(uH, cmul(fd_s1, chz));                                            // :522-523
    return uH;                                                                                  // :525
}

}  // namespace UCF_NS
#if UCF_FAST
#include "ucf_fastpath.h"
#endif
namespace UCF_NS {

// ------------------------------------------------------------------ invlap.f90:46-141
// Wave-cooperative de Hoog: lane i holds f(p_i), i = 0..2M.  Returns f(t) (uniform).
UCF_DEV double dehoog_wave(cplx f, int M, double alpha, double logtol, double t, double tee, int lane,
                           ucf_stats* st)
{
    const int n2 = 2 * M;
    const bool act = lane <= n2;
    double mag = act ? cabs_(f) : 0.0;
    if (d_isnan(mag)) mag = 0.0;                         // MAXVAL skips NaN operands
    const double mx = wave_max(mag);
    if (!(mx > UCF_DBL_MIN)) {                           // :69,139
        if (st) stat_add(&st->zero_vectors, lane == 0);
        return 0.0;
    }
    const bool nanp = act && (d_isnan(f.re) || d_isnan(f.im));
    if (st) stat_add(&st->nan_scrubbed, nanp);
    cplx ff = (nanp || !act) ? cmake(act ? 0.0 : 1.0, 0.0) : f;                                 // :71-74
    const double gamma = alpha - logtol / (2.0 * tee);                                          // :77

    const cplx ff0 = bcast0(ff);
    const cplx d0 = cdivr(ff0, 2.0);                                                            // :98
    cplx fnext = shfl_down1(ff);
    cplx q = (lane == 0) ? cdiv(fnext, d0) : cdiv(fnext, ff);                                   // :81-82  q(i,1)
    if (lane > n2 - 1) q = cmake(1.0, 0.0);
    cplx e = cmake(0.0, 0.0);                                                                   // :80     e(i,0)

    cplx Am2 = cmake(0.0, 0.0), Am1 = d0, Bm2 = cmake(1.0, 0.0), Bm1 = cmake(1.0, 0.0);         // :105-107
    const cplx z = cexp_(cdivr(cscale(cscale(cmake(0.0, 1.0), UCF_PI), t), tee));               // :110
    cplx dlast_q = cmake(0.0, 0.0), dlast_e = cmake(0.0, 0.0);
    for (int r = 1; r <= M; r++) {                                                              // :85-95
        const cplx qn = shfl_down1(q);
        const cplx en = shfl_down1(e);
        cplx enew = cadd(csub(qn, q), en);
        if (lane > 2 * (M - r)) enew = cmake(1.0, 0.0);      // outside the rhombus: keep lanes benign
        const cplx dq = cneg(bcast0(q));                     // d(2r-1) = -q(0,r)                :100
        const cplx de = cneg(bcast0(enew));                  // d(2r)   = -e(0,r)                :101
        {                                                    // n = 2r-1                         :114-117
            cplx An = cadd(Am1, cmul(cmul(dq, Am2), z));
            cplx Bn = cadd(Bm1, cmul(cmul(dq, Bm2), z));
            Am2 = Am1; Am1 = An; Bm2 = Bm1; Bm1 = Bn;
        }
        if (r < M) {
            cplx An = cadd(Am1, cmul(cmul(de, Am2), z));     // n = 2r
            cplx Bn = cadd(Bm1, cmul(cmul(de, Bm2), z));
            Am2 = Am1; Am1 = An; Bm2 = Bm1; Bm1 = Bn;
            const cplx enn = shfl_down1(enew);
            q = cdiv(cmul(qn, enn), enew);                                                      // :93
            if (lane > 2 * (M - r - 1) + 1) q = cmake(1.0, 0.0);
            e = enew;
        } else {
            dlast_q = dq;
            dlast_e = de;
        }
    }
    // :120-125 improved remainder
    const cplx brem = cdivr(radd(1.0, cmul(csub(dlast_q, dlast_e), z)), 2.0);
    const cplx inner = csqrt_(radd(1.0, cdiv(cmul(dlast_e, z), cmul(brem, brem))));
    const cplx rem = cneg(cmul(brem, rsub(1.0, inner)));
    const cplx A2M = cadd(Am1, cmul(rem, Am2));
    const cplx B2M = cadd(Bm1, cmul(rem, Bm2));
    return exp(gamma * t) / tee * cdiv(A2M, B2M).re;                                            // :129
}

// dehoog_wave in two halves, for kernels that invert many vectors per wave (dehoog_tiles_kernel):
//   dehoog_qd_wave   the quotient-difference rhombus (:80-101), cooperative as above; the continued-fraction
//                    coefficients d(k) go to an LDS column instead of every lane running the recurrence on broadcasts;
//   dehoog_cf_lane   the A/B recurrence, improved remainder and scaling (:105-129) of ONE vector by ONE lane,
//                    coefficients read from an LDS column -- 2T lanes finish 2T vectors at once.
// Operation for operation the arithmetic of dehoog_wave: same bits (faithful flavour; the fast one takes the unscaled
// quotient below where it can).
// q(i,r+1) = q(i+1,r) e(i+1,r) / e(i,r) (:93).  Fast flavour: product times conj(e)/|e|^2 without the exponent scaling of
// __divdc3 when every lane's |e| is far from over- and underflow of its square (the lanes outside the rhombus hold 1).
// `valid`: this lane's entry belongs to the rhombus (the others carry whatever the shifts brought: they never feed a valid
// entry -- q(i,r+1) needs e(i+1,r), e(i,r+1) needs q(i+1,r+1), both inside -- and must not decide the wave's path)
UCF_DEV cplx qd_quotient(cplx qn, cplx enn, cplx enew, bool valid)
{
#if UCF_FAST
    const double m = fmax(fabs(enew.re), fabs(enew.im));
    if (__builtin_amdgcn_ballot_w64(valid && !(m < 1.0e150 && m > 1.0e-150)) == 0) {
        const double r = fast_rcp(__builtin_fma(enew.re, enew.re, enew.im * enew.im));
        const cplx num = cmul(qn, enn);
        return cmake((num.re * enew.re + num.im * enew.im) * r, (num.im * enew.re - num.re * enew.im) * r);
    }
#endif
    return cdiv(cmul(qn, enn), enew);
}
// The continued-fraction coefficients go straight to column `dcol` of an LDS tile (d(k) in row k, `pitch` apart): lane 0,
// which holds q(0,r) and e(0,r), stores d(2r-1) and d(2r) as they appear -- no broadcast, no per-lane selects.  e(i+1,r)
// of one step is the shifted e of the next: one shift less per step.
UCF_DEV void dehoog_qd_wave(cplx f, int M, int lane, ucf_stats* st, bool* zero, lds_c* dcol, int pitch)
{
    const int n2 = 2 * M;
    const bool act = lane <= n2;
    // maxval(abs(f)) > tiny (:69,139; MAXVAL skips NaN operands): the larger component decides wherever it is clear of the
    // threshold by the factor sqrt 2 that hypot can add; only a vector at the very edge of the denormals asks hypot itself
    double mag = act ? fmax(fabs(f.re), fabs(f.im)) : 0.0;            // (fmax drops a NaN operand; NaN, NaN -> NaN -> 0 below)
    if (d_isnan(mag)) mag = 0.0;
    double mx = wave_max(mag);
    if (__builtin_expect(!(mx > UCF_DBL_MIN) && mx > 0.7 * UCF_DBL_MIN, 0)) {
        mag = act ? cabs_(f) : 0.0;
        if (d_isnan(mag)) mag = 0.0;
        mx = wave_max(mag);
    }
    *zero = !(mx > UCF_DBL_MIN);
    if (*zero) {
        if (st) stat_add(&st->zero_vectors, lane == 0);
        return;
    }
    const bool nanp = act && (d_isnan(f.re) || d_isnan(f.im));
    if (st) stat_add(&st->nan_scrubbed, nanp);
    cplx ff = (nanp || !act) ? cmake(act ? 0.0 : 1.0, 0.0) : f;                                 // :71-74
    const cplx ff0 = bcast0(ff);
    const cplx d0 = cdivr(ff0, 2.0);                                                            // :98
    if (lane == 0) dcol[0] = make_double2(d0.re, d0.im);
    cplx fnext = shfl_down1(ff);
    cplx q = (lane == 0) ? cdiv(fnext, d0) : cdiv(fnext, ff);                                   // :81-82  q(i,1)
    cplx en = cmake(0.0, 0.0);                                                                  // :80     e(i+1,0)
    // Lanes outside the rhombus -- i > 2(M - r) for e(.,r), i > 2(M - r) - 1 for q(.,r+1) -- hold whatever the shifts bring
    // and no entry inside ever reads them.  The fast flavour leaves them alone (8 selects per step less); the faithful one
    // keeps them at 1 so that its scaled division never takes the out-of-line Annex-G recovery for a lane nobody reads.
#if !UCF_FAST
    if (lane > n2 - 1) q = cmake(1.0, 0.0);
#endif
    for (int r = 1; r <= M; r++) {                                                              // :85-95
        const cplx qn = shfl_down1(q);
        cplx enew = cadd(csub(qn, q), en);
#if !UCF_FAST
        if (lane > 2 * (M - r)) enew = cmake(1.0, 0.0);
#endif
        if (lane == 0) {
            dcol[(size_t)(2 * r - 1) * pitch] = make_double2(-q.re, -q.im);                     // d(2r-1) = -q(0,r)   :100
            dcol[(size_t)(2 * r) * pitch] = make_double2(-enew.re, -enew.im);                   // d(2r)   = -e(0,r)   :101
        }
        if (r < M) {
            const cplx enn = shfl_down1(enew);
            q = qd_quotient(qn, enn, enew, lane <= 2 * (M - r - 1) + 1);                        // :93
#if !UCF_FAST
            if (lane > 2 * (M - r - 1) + 1) q = cmake(1.0, 0.0);
#endif
            en = enn;
        }
    }
}

UCF_DEV double dehoog_cf_lane(const lds_c* dcol, int pitch, int M, double alpha, double logtol, double t, double tee)
{
    const double gamma = alpha - logtol / (2.0 * tee);                                          // :77
    const lds_c v0 = dcol[0];
    const cplx d0 = cmake(v0.x, v0.y);
    cplx Am2 = cmake(0.0, 0.0), Am1 = d0, Bm2 = cmake(1.0, 0.0), Bm1 = cmake(1.0, 0.0);         // :105-107
    const cplx z = cexp_(cdivr(cscale(cscale(cmake(0.0, 1.0), UCF_PI), t), tee));               // :110
    for (int n = 1; n <= 2 * M - 1; n++) {                                                      // :114-117
        const lds_c v = dcol[(size_t)n * pitch];
        const cplx d = cmake(v.x, v.y);
        const cplx An = cadd(Am1, cmul(cmul(d, Am2), z));
        const cplx Bn = cadd(Bm1, cmul(cmul(d, Bm2), z));
        Am2 = Am1; Am1 = An; Bm2 = Bm1; Bm1 = Bn;
    }
    const lds_c vq = dcol[(size_t)(2 * M - 1) * pitch], ve = dcol[(size_t)(2 * M) * pitch];
    const cplx dlast_q = cmake(vq.x, vq.y), dlast_e = cmake(ve.x, ve.y);
    // :120-125 improved remainder
    const cplx brem = cdivr(radd(1.0, cmul(csub(dlast_q, dlast_e), z)), 2.0);
    const cplx inner = csqrt_(radd(1.0, cdiv(cmul(dlast_e, z), cmul(brem, brem))));
    const cplx rem = cneg(cmul(brem, rsub(1.0, inner)));
    const cplx A2M = cadd(Am1, cmul(rem, Am2));
    const cplx B2M = cadd(Bm1, cmul(rem, Bm2));
    return exp(gamma * t) / tee * cdiv(A2M, B2M).re;                                            // :129
}

// Same algorithm for 2M+1 > 64: element i = lane + 64 g lives in register set g of its lane, G = 2 sets for
// 2M+1 <= 128, G = 4 for 2M+1 <= 256 (M up to 127).  Used by dehoog_points_kernel and the big-M branch of
// dehoog_tiles_kernel; elementwise identical to dehoog_wave.
template <int G>
UCF_DEV void shift_down_sets(const cplx (&in)[G], cplx (&out)[G], int lane)
{
#pragma unroll
    for (int g = 0; g < G; g++) {
        const cplx a = shfl_down1(in[g]);
        // lane 63 of set g takes element 64 (g + 1) = lane 0 of the next set; the last set keeps its own
        const cplx nx = (g + 1 < G) ? bcast0(in[g + 1 < G ? g + 1 : g]) : a;
        out[g] = (lane == 63 && g + 1 < G) ? nx : a;
    }
}

template <int G>
UCF_DEV double dehoog_sets(const cplx (&f)[G], int M, double alpha, double logtol, double t, double tee, int lane,
                           ucf_stats* st)
{
    const int n2 = 2 * M;
    double mag = 0.0;
    cplx ff[G], q[G], e[G];
    bool nans[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int i = lane + 64 * g;
        const bool act = i <= n2;
        double m1 = act ? cabs_(f[g]) : 0.0;
        if (d_isnan(m1)) m1 = 0.0;
        mag = fmax(mag, m1);
        const bool nanp = act && (d_isnan(f[g].re) || d_isnan(f[g].im));
        nans[g] = nanp;
        ff[g] = (nanp || !act) ? cmake(act ? 0.0 : 1.0, 0.0) : f[g];
    }
    const double mx = wave_max(mag);
    if (!(mx > UCF_DBL_MIN)) {
        if (st) stat_add(&st->zero_vectors, lane == 0);
        return 0.0;
    }
    if (st) {
#pragma unroll
        for (int g = 0; g < G; g++) stat_add(&st->nan_scrubbed, nans[g]);      // per sample
    }
    const double gamma = alpha - logtol / (2.0 * tee);
    const cplx ff0 = bcast0(ff[0]);
    const cplx d0 = cdivr(ff0, 2.0);
    cplx fn[G];
    shift_down_sets<G>(ff, fn, lane);
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int i = lane + 64 * g;
        q[g] = (i == 0) ? cdiv(fn[g], d0) : cdiv(fn[g], ff[g]);
        if (i > n2 - 1) q[g] = cmake(1.0, 0.0);
        e[g] = cmake(0.0, 0.0);
    }
    cplx Am2 = cmake(0.0, 0.0), Am1 = d0, Bm2 = cmake(1.0, 0.0), Bm1 = cmake(1.0, 0.0);
    const cplx z = cexp_(cdivr(cscale(cscale(cmake(0.0, 1.0), UCF_PI), t), tee));
    cplx dlast_q = cmake(0.0, 0.0), dlast_e = cmake(0.0, 0.0);
    for (int r = 1; r <= M; r++) {
        cplx qn[G], en[G], enew[G];
        shift_down_sets<G>(q, qn, lane);
        shift_down_sets<G>(e, en, lane);
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int i = lane + 64 * g;
            enew[g] = cadd(csub(qn[g], q[g]), en[g]);
            if (i > 2 * (M - r)) enew[g] = cmake(1.0, 0.0);
        }
        const cplx dq = cneg(bcast0(q[0]));
        const cplx de = cneg(bcast0(enew[0]));
        {
            cplx An = cadd(Am1, cmul(cmul(dq, Am2), z));
            cplx Bn = cadd(Bm1, cmul(cmul(dq, Bm2), z));
            Am2 = Am1; Am1 = An; Bm2 = Bm1; Bm1 = Bn;
        }
        if (r < M) {
            cplx An = cadd(Am1, cmul(cmul(de, Am2), z));
            cplx Bn = cadd(Bm1, cmul(cmul(de, Bm2), z));
            Am2 = Am1; Am1 = An; Bm2 = Bm1; Bm1 = Bn;
            cplx enn[G];
            shift_down_sets<G>(enew, enn, lane);
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int i = lane + 64 * g;
                q[g] = cdiv(cmul(qn[g], enn[g]), enew[g]);
                if (i > 2 * (M - r - 1) + 1) q[g] = cmake(1.0, 0.0);
                e[g] = enew[g];
            }
        } else {
            dlast_q = dq;
            dlast_e = de;
        }
    }
    const cplx brem = cdivr(radd(1.0, cmul(csub(dlast_q, dlast_e), z)), 2.0);
    const cplx inner = csqrt_(radd(1.0, cdiv(cmul(dlast_e, z), cmul(brem, brem))));
    const cplx rem = cneg(cmul(brem, rsub(1.0, inner)));
    const cplx A2M = cadd(Am1, cmul(rem, Am2));
    const cplx B2M = cadd(Bm1, cmul(rem, Bm2));
    return exp(gamma * t) / tee * cdiv(A2M, B2M).re;
}

// h and dh of one vector of 2M+1 > 64 transform samples read by `load(i)`, p_i = sigma + i pi/tee
template <int G, class LOAD>
UCF_DEV void dehoog_big(LOAD load, int np, int M, double alpha, double logtol, double sigma, double tD, double tee, int lane,
                        ucf_stats* st, double* hval, double* dval)
{
    cplx tl[G], tp[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int i = lane + 64 * g;
        tl[g] = cmake(0.0, 0.0);
        if (i < np) tl[g] = load(i);
        tp[g] = cmul(tl[g], cmake(sigma, UCF_PI * i / tee));
    }
    *hval = dehoog_sets<G>(tl, M, alpha, logtol, tD, tee, lane, st);
    *dval = dehoog_sets<G>(tp, M, alpha, logtol, tD, tee, lane, st) * tD;
}

// ------------------------------------------------------------- integration.f90:125-189
// Per-lane Wynn-epsilon on two quarter-wave LDS scratch columns: colA[i] holds series(i+1) on entry
// (overwritten by the even epsilon columns), colB the odd columns.
// status: 0 ok, 1 truncated, 2 sentinel, 3 early exit.
template <bool CUR_IS_A, int PART = UCF_PART>
UCF_DEV bool wynn_column(lds_c* colA, lds_c* colB, int count, int lane, cplx* acc)
{
    // new(m) = prev(m+1) + 1/(cur(m+1) - cur(m)), m = 1..count; new column overwrites prev storage
    lds_c* cur = CUR_IS_A ? colA : colB;
    lds_c* prv = CUR_IS_A ? colB : colA;
    for (int m = 1; m <= count; m++) {
        const cplx hi = scr_ld<PART>(cur, m, lane);
        const cplx lo = scr_ld<PART>(cur, m - 1, lane);
        const cplx denom = csub(hi, lo);
#if UCF_FAST
        // |denom| > eps  <=>  |denom|^2 > eps^2 (no hypot), and 1/denom = conj(denom)/|denom|^2 reuses it
        const double d2 = __builtin_fma(denom.re, denom.re, denom.im * denom.im);
        if (d2 > UCF_EPS * UCF_EPS && d2 < 1.0e300) {
            const double r = fast_rcp(d2);
            const cplx pv = scr_ld<PART>(prv, m, lane);
            scr_st<PART>(prv, m - 1, lane, cmake(__builtin_fma(denom.re, r, pv.re), __builtin_fma(-denom.im, r, pv.im)));
        } else if (cabs_(denom) > UCF_EPS) {                 // huge or non-finite difference: exact reference path
            scr_st<PART>(prv, m - 1, lane, cadd(scr_ld<PART>(prv, m, lane), rdiv(1.0, denom)));
        } else {
#else
        if (cabs_(denom) > UCF_EPS) {                                                           // :172
            scr_st<PART>(prv, m - 1, lane, cadd(scr_ld<PART>(prv, m, lane), rdiv(1.0, denom)));             // :173
        } else {
#endif
            *acc = hi;                                                                          // :175
            return true;
        }
    }
    return false;
}

template <int PART = UCF_PART>
UCF_DEV cplx wynn_lane(lds_c* colA, lds_c* colB, int nin, int lane, int* status)
{
    int ns = nin;
    int stat = 0;
    cplx run = cmake(0.0, 0.0);
    for (int i = 1; i <= nin; i++) {                                                            // :140-163
        cplx s = scr_ld<PART>(colA, i - 1, lane);
        if (!c_is_finite(s)) {
            ns = i - 1;
            stat = (ns < 4) ? 2 : 1;
            break;
        }
        run = (i == 1) ? s : cadd(run, s);
        scr_st<PART>(colA, i - 1, lane, run);
    }
    if (stat == 2) {
        *status = 2;
        return cmake((double)(-999999.9f), 0.0);                                                // :148
    }
    for (int m = 0; m < ns; m++) scr_st<PART>(colB, m, lane, cmake(0.0, 0.0));                        // :166
    cplx acc = cmake(0.0, 0.0);
    bool done = false;
    for (int j = 0; j <= ns - 2 && !done; j++) {                                                // :169-181
        const int count = ns - (j + 1);
        done = (j & 1) ? wynn_column<false, PART>(colA, colB, count, lane, &acc) : wynn_column<true, PART>(colA, colB, count, lane, &acc);
        if (done) stat = 3;
    }
    if (!done) acc = scr_ld<PART>(colA, 1, lane);   // eps(2,ns-2) [ns even] or eps(2,ns-3) [ns odd]: both even columns
    *status = stat;
    return acc;
}

// The same algorithm with both epsilon columns in registers (fully unrolled over NMAX >= nin terms, every step
// guarded per lane by the reference's own loop bounds): no LDS round trips, which is what limits finish_kernel.
// series[i] holds term i+1 on entry.
template <int NMAX>
UCF_DEV cplx wynn_regs(const cplx (&series)[NMAX], int nin, int* status)
{
    cplx A[NMAX], B[NMAX];
    int ns = nin;
    int stat = 0;
    bool cut = false;
    cplx run = cmake(0.0, 0.0);
#pragma unroll
    for (int i = 1; i <= NMAX; i++) {                                                           // :140-163
        A[i - 1] = cmake(0.0, 0.0);
        B[i - 1] = cmake(0.0, 0.0);                                                             // :166
        if (i <= nin && !cut) {
            const cplx sv = series[i - 1];
            if (!c_is_finite(sv)) {
                ns = i - 1;
                stat = (ns < 4) ? 2 : 1;
                cut = true;
            } else {
                run = (i == 1) ? sv : cadd(run, sv);
                A[i - 1] = run;
            }
        }
    }
    if (stat == 2) {
        *status = 2;
        return cmake((double)(-999999.9f), 0.0);                                                // :148
    }
    cplx acc = cmake(0.0, 0.0);
    bool done = false;
#pragma unroll
    for (int j = 0; j <= NMAX - 2; j++) {                                                       // :169-181
        const int count = ns - (j + 1);
#pragma unroll
        for (int m = 1; m <= NMAX - 1 - j; m++) {
            if (!done && m <= count) {
                // new(m) = prev(m+1) + 1/(cur(m+1) - cur(m)); the new column overwrites prev's storage
                const cplx hi = (j & 1) ? B[m] : A[m];
                const cplx lo = (j & 1) ? B[m - 1] : A[m - 1];
                const cplx pv = (j & 1) ? A[m] : B[m];
                const cplx denom = csub(hi, lo);
                cplx nv = pv;
                bool ok;
#if UCF_FAST
                const double d2 = __builtin_fma(denom.re, denom.re, denom.im * denom.im);
                if (d2 > UCF_EPS * UCF_EPS && d2 < 1.0e300) {
                    const double r = fast_rcp(d2);
                    nv = cmake(__builtin_fma(denom.re, r, pv.re), __builtin_fma(-denom.im, r, pv.im));
                    ok = true;
                } else if (cabs_(denom) > UCF_EPS) {             // huge or non-finite difference: exact reference path
                    nv = cadd(pv, rdiv(1.0, denom));
                    ok = true;
                } else {
                    ok = false;
                }
#else
                ok = cabs_(denom) > UCF_EPS;                                                    // :172
                if (ok) nv = cadd(pv, rdiv(1.0, denom));                                        // :173
#endif
                if (ok) {
                    if (j & 1) A[m - 1] = nv; else B[m - 1] = nv;
                } else {
                    acc = hi;                                                                   // :175
                    done = true;
                    stat = 3;
                }
            }
        }
    }
    if (!done) acc = A[1];        // eps(2,ns-2) [ns even] or eps(2,ns-3) [ns odd]: both even columns
    *status = stat;
    return acc;
}

#if UCF_FAST
// The usual case of wynn_regs on its own: every lane of the wave has all its nin terms finite and no step of the table
// comes near the absolute-epsilon exit (:172).  Then the bounds of the table are wave-uniform (scalar branches, no per-lane
// predicates or selects) and the steps run unguarded, each lane only noting whether a guard WOULD have fired; if any did,
// in any lane, the result is discarded and the caller runs the guarded table.  Same arithmetic, same bits as wynn_regs.
template <int NMAX>
UCF_DEV bool wynn_regs_clean(const cplx (&series)[NMAX], int nin, cplx* acc_out)
{
    cplx A[NMAX], B[NMAX];
    bool bad = false;
    cplx run = cmake(0.0, 0.0);
#pragma unroll
    for (int i = 1; i <= NMAX; i++) {                                                           // :140-163
        A[i - 1] = cmake(0.0, 0.0);
        B[i - 1] = cmake(0.0, 0.0);                                                             // :166
        if (i <= nin) {
            const cplx sv = series[i - 1];
            bad |= !c_is_finite(sv);
            run = (i == 1) ? sv : cadd(run, sv);
            A[i - 1] = run;
        }
    }
    if (nin < 4 || __builtin_amdgcn_ballot_w64(bad) != 0) return false;
#pragma unroll
    for (int j = 0; j <= NMAX - 2; j++) {                                                       // :169-181
        const int count = nin - (j + 1);
#pragma unroll
        for (int m = 1; m <= NMAX - 1 - j; m++) {
            if (m <= count) {
                const cplx hi = (j & 1) ? B[m] : A[m];
                const cplx lo = (j & 1) ? B[m - 1] : A[m - 1];
                const cplx pv = (j & 1) ? A[m] : B[m];
                const cplx denom = csub(hi, lo);
                const double d2 = __builtin_fma(denom.re, denom.re, denom.im * denom.im);
                bad |= !(d2 > UCF_EPS * UCF_EPS && d2 < 1.0e300);
                const double r = fast_rcp(d2);
                const cplx nv = cmake(__builtin_fma(denom.re, r, pv.re), __builtin_fma(-denom.im, r, pv.im));
                if (j & 1) A[m - 1] = nv; else B[m - 1] = nv;
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0) return false;
    *acc_out = A[1];
    return true;
}
#endif

// ------------------------------------------------------------- integration.f90:192-237
// Per-lane Neville extrapolation to x = 0.  colC[(i*strideC)][lane] holds y(i+1) (destroyed),
// colD is half-wave scratch.  x is wave-uniform.
template <int PART = UCF_PART>
UCF_DEV cplx extrap_lane(lds_c* colC, int strideC, lds_c* colD, const double* x, int n, int lane)
{
    int ns = 1;
    for (int i = 2; i <= n; i++) if (x[i - 1] < x[ns - 1]) ns = i;                              // minloc
    for (int i = 0; i < n; i++) scr_st<PART>(colD, i, lane, lds_ld(colC, i * strideC, lane));
    cplx y = lds_ld(colC, (ns - 1) * strideC, lane);
    ns = ns - 1;
    for (int m = 1; m <= n - 1; m++) {
        for (int i = 1; i <= n - m; i++) {
            const double dx = x[i - 1] - x[i + m - 1];
            cplx ci = lds_ld(colC, i * strideC, lane);
            cplx di = scr_ld<PART>(colD, i - 1, lane);
            cplx den = cdiv(csub(ci, di), cmake(dx, 0.0));                                      // :227 (complex/complex)
            scr_st<PART>(colD, i - 1, lane, rscale(x[i + m - 1], den));
            lds_st(colC, (i - 1) * strideC, lane, rscale(x[i - 1], den));
        }
        cplx dy;
        if (2 * ns < n - m) {
            dy = lds_ld(colC, ns * strideC, lane);
        } else {
            dy = scr_ld<PART>(colD, ns - 1, lane);
            ns = ns - 1;
        }
        y = cadd(y, dy);
    }
    return y;
}

// ------------------------------------------------------------------ abscissa tables
// Everything about an abscissa that does not depend on the Laplace sample: a_n and
// a_n*J0(a_n*rD) (laplace_hankel_solutions.f90:118).  It depends only on (rD, sv), so a sweep
// computes it once per radius (and split index) instead of once per lane, time and abscissa:
// row-major table tab[row][n] = (a, a*J0(a*rD)), n < N tanh-sinh (integration.f90:62 with this
// row's own arg), n >= N Gauss-Lobatto nodes between J0 zeros (driver.f90:189-193).
#if !UCF_FAST
__global__ void __launch_bounds__(256)
abscissa_kernel(const ucf_dev_params P, int nrows, int per_point, int nsv, int svmin,
                const double* __restrict__ rDv, const int* __restrict__ svv, double2* __restrict__ tab)
{
    // P.tab_premul (fast flavour): the Gauss-Lobatto entries carry their quadrature weight, a J0(a rD) w_m, so that the
    // abscissa loop accumulates a sample with one FMA per component (the faithful flavour keeps the reference's
    // ((a J0 f) lapTime) w order and multiplies by w_m itself)
    const int nabs = P.N + P.nacc * P.ngl;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)nrows * nabs) return;
    const int row = (int)(gid / nabs), n = (int)(gid % nabs);
    double rD;
    int sv;
    if (per_point) { rD = rDv[row]; sv = svv[row]; }
    else { rD = rDv[row / nsv]; sv = svmin + row % nsv; }
    double a, w = 1.0;
    if (n < P.N) {
        const double arg = P.j0z[sv - 1] / rD;                                                  // driver.f90:120
        a = (P.ts_x[n] * arg) / 2.0;                                                            // integration.f90:62
    } else {
        const int jj = (n - P.N) / P.ngl, m = (n - P.N) % P.ngl;
        const double lob = P.j0z[sv + jj - 1] / rD;                                             // driver.f90:188-190
        const double hib = P.j0z[sv + jj] / rD;
        const double width = hib - lob;
        a = (width * P.gl_x[m] + (hib + lob)) / 2.0;                                            // :193
        if (P.tab_premul) w = P.gl_w[m];
    }
    const double aj = a * j0(a * rD);
    tab[gid] = make_double2(a, P.tab_premul ? aj * w : aj);
}
#endif

// ------------------------------------------------------------------ the point kernel
// driver.f90:100-232 for one (t,r) point per wavefront (single-point-run semantics,
// SURVEY.md quirks Q1 and Q5).
// LDS per wave (slots of 64 complex):  [R*nz] level sums | [nacc*nz] interval areas |
// [max(nacc,R)] x 32 lanes scratch (Richardson / Wynn run on one half-wave at a time) |
// FAMILY 4: [2*order] Thomas sweep.
//
// Two lane layouts of the same body:
//   LAYOUT 0  lane = Laplace sample p_m of ONE point (2M+1 <= 64 live lanes); de Hoog follows in the same
//             wave.  Used for arbitrary point lists and short time vectors.
//   LAYOUT 1  lane = one of 64 consecutive TIMES at one radius, the wave owns one Laplace index m for all of
//             them (all 64 lanes live whatever M is; abscissa row still wave-uniform).  The accelerated
//             transform totlap(t, r, z, m) goes to an HBM workspace (16 B per sample point) and
//             dehoog_points_kernel inverts it with lane = m.  Per-lane arithmetic is identical to LAYOUT 0,
//             so both give the same bits.
// decoding of a work-item index into (time, radius, Laplace index), shared by the kernels below
struct work_item {
    int it, ir, mlap, pidx, plan;
    bool live;
};
template <int LAYOUT>
UCF_DEV work_item decode_item(const ucf_dev_params& P, int pt, int lane, int per_point, int nr, int nt, int ir0, int npts)
{
    work_item W;
    W.plan = 0;
    if (LAYOUT == 0) {
        W.it = per_point ? pt : pt / nr;
        W.ir = per_point ? pt : pt % nr;
        W.mlap = lane;
        W.live = lane < P.np;
        W.pidx = pt;
    } else if (LAYOUT == 2) {
        // 2M+1 > 64: work item = (point, chunk of 64 Laplace samples); inversion in dehoog_points_kernel
        const int nchunk = (P.np + UCF_WAVE - 1) / UCF_WAVE;
        const int q = pt / nchunk;
        W.it = per_point ? q : q / nr;
        W.ir = per_point ? q : q % nr;
        W.mlap = (pt % nchunk) * UCF_WAVE + lane;
        W.live = W.mlap < P.np;
        W.pidx = q;
    } else if (LAYOUT == 3) {
        // lane = point of an arbitrary point list: work item = (tile of 64 points, Laplace index), all lanes live
        // whatever M is; times, radii, split indices and abscissa rows are per lane.  nr = points per plan (a
        // parameter batch cuts its tiles per plan so that a wave never straddles two parameter blocks), nt = points
        // of the launch
        const int tpp = (nr + UCF_WAVE - 1) / UCF_WAVE;
        const int tile = pt / P.np;
        W.mlap = pt % P.np;
        W.plan = tile / tpp;
        const int qin = (tile % tpp) * UCF_WAVE + lane;
        W.live = qin < nr;
        W.pidx = W.plan * nr + (W.live ? qin : nr - 1);
        W.it = W.ir = W.pidx;
    } else {
        // work item = (radius of this chunk, tile of 64 times, Laplace index); m fastest so that
        // neighbouring waves share the abscissa row and the times
        // The LAST radius of the chunk first: with radii in ascending order (the usual sweep) the far ones stay longest on
        // the cosh/sinh form of the closure and cost up to 1.5 x the near ones -- longest items first leaves the cheap ones
        // to fill the tail of the launch (matters for small launches: the shards of a strong-scaling run).
        const int ntiles = (nt + UCF_WAVE - 1) / UCF_WAVE;
        W.mlap = pt % P.np;
        const int tile = (pt / P.np) % ntiles;
        const int per_r = P.np * ntiles;
        // (water-table closures only, models 3-5: measured -3 % on a 1/8 shard of C2; the finite-difference closure, whose
        //  items cost the same at every radius, LOSES 3 % to the reversed order)
        const bool far_first = (P.model >= 3 && P.model <= 5);
        W.ir = ir0 + (far_first ? npts / per_r - 1 - pt / per_r : pt / per_r);
        W.it = tile * UCF_WAVE + lane;
        W.live = W.it < nt;
        if (!W.live) W.it = nt - 1;
        W.pidx = pt;
    }
    return W;
}

// ------------------------------------------------------------------ the tail of a work item
// driver.f90:159-230 per depth: scale the level sums, Richardson/Neville to h -> 0, Wynn-epsilon over the
// J0-interval areas, then the transform goes to the workspace (LAYOUT 1, 2) or straight through de Hoog
// (LAYOUT 0).  accTS: [R][nz] level sums in LDS; the finished areas come from LDS (accGL) or global memory.
// PART lanes of the wave work at a time on the scratch columns scr ([max(2 nacc, R)][PART]).
#define UCF_WYNN_REGS 12      /* terms the register-resident Wynn-epsilon of finish_kernel holds */
// MODE (finish_kernel, WREG, fast flavour): 0 = everything here; 1 = the unguarded epsilon table only -- a (item, depth) in
// which some lane needs one of the reference's guards (:297-311: a zero difference, the early exit, an all-zero series) is
// appended to the list `defer` = [count | pt * nz + z ...] and left alone; 2 = the depths [z0, z1) of a listed item with the
// guarded table.  Split so that the kernel every item runs is not allocated for the table almost none needs (168 -> 125
// VGPRs, 3 -> 4 waves per SIMD; C2: 1.20 -> 0.9 ms)
template <int LAYOUT, int PART, bool WREG = false, int MODE = 0>
UCF_DEV void finish_item(const ucf_dev_params& P, lds_c* accTS, lds_c* scr, const lds_c* accGL,
                         const double2* __restrict__ areas, double arg, const work_item& W, int pt, double tD, double tee,
                         cplx p, ucf_stats* st, int nt, int ir0, double2* __restrict__ totlap, double* __restrict__ hout,
                         double* __restrict__ dhout, int z0 = 0, int z1 = -1, int* __restrict__ defer = nullptr)
{
    const int lane = threadIdx.x;
    const int nz = P.nz, R = P.R, nacc = P.nacc;
    if (z1 < 0) z1 = nz;
    for (int z = z0; z < z1; z++) {
        for (int j = 0; j < R; j++) {
            const int slot = j * nz + z;
            lds_st(accTS, slot, lane, rscale(arg / 2.0, lds_ld(accTS, slot, lane)));            // :135,154
        }
        bool any = false;
        cplx finint = lds_ld(accTS, z, lane);
        cplx infint = cmake(0.0, 0.0);
        int wst = 0;
        if (WREG) {
            // finish_kernel, nacc <= UCF_WYNN_REGS: the areas go from the state straight into registers
            cplx ser[UCF_WYNN_REGS];
#pragma unroll
            for (int jj = 0; jj < UCF_WYNN_REGS; jj++) {
                ser[jj] = cmake(0.0, 0.0);
                if (jj < nacc) {
                    const double2 v = areas[(size_t)(jj * nz + z) * UCF_WAVE + lane];
                    ser[jj] = cmake(v.x, v.y);
                    any |= c_abs_positive(ser[jj]);                                              // :209
                }
            }
            for (int part = 0; part < UCF_WAVE / PART; part++)
                if ((lane / PART) == part && R > 1) finint = extrap_lane<PART>(accTS + (size_t)z * UCF_WAVE, nz, scr, P.hv, R, lane);
#if UCF_FAST
            // (the whole wave on the unguarded table when no lane needs a guard -- all areas zero in a lane counts as a guard)
            if (MODE == 2) {
                if (any) infint = wynn_regs<UCF_WYNN_REGS>(ser, nacc, &wst);
            } else if (__builtin_amdgcn_ballot_w64(!any) != 0 || !wynn_regs_clean<UCF_WYNN_REGS>(ser, nacc, &infint)) {
                if (MODE == 1) {
                    if (lane == 0) defer[1 + atomicAdd(&defer[0], 1)] = pt * nz + z;
                    continue;                              // (nothing of this depth has left the wave yet)
                }
                infint = cmake(0.0, 0.0);
                if (any) infint = wynn_regs<UCF_WYNN_REGS>(ser, nacc, &wst);
            }
#else
            if (any) infint = wynn_regs<UCF_WYNN_REGS>(ser, nacc, &wst);
#endif
        } else
        for (int part = 0; part < UCF_WAVE / PART; part++) {
            if ((lane / PART) == part) {
                if (R > 1) finint = extrap_lane<PART>(accTS + (size_t)z * UCF_WAVE, nz, scr, P.hv, R, lane);
                lds_c* colA = scr;
                lds_c* colB = scr + (size_t)nacc * PART;
                for (int jj = 0; jj < nacc; jj++) {
                    cplx ar;
                    if (accGL) {
                        ar = lds_ld(accGL, jj * nz + z, lane);
                    } else {
                        const double2 v = areas[(size_t)(jj * nz + z) * UCF_WAVE + lane];
                        ar = cmake(v.x, v.y);
                    }
                    any |= c_abs_positive(ar);                                                   // :209
                    scr_st<PART>(colA, jj, lane, ar);
                }
                if (any) infint = wynn_lane<PART>(colA, colB, nacc, lane, &wst);
            }
        }
        if (st) {
            stat_add(&st->wynn_all_zero, W.live && !any);
            stat_add(&st->wynn_truncated, W.live && wst == 1);
            stat_add(&st->wynn_sentinel, W.live && wst == 2);
            stat_add(&st->wynn_early_exit, W.live && wst == 3);
        }
        const cplx tl = cadd(finint, infint);                                                   // :216
        if (LAYOUT == 1) {
            // [radius of the chunk][z][m][time]: the 64 lanes (consecutive times) store 1 KB contiguously
            if (W.live) totlap[(((size_t)(W.ir - ir0) * nz + z) * P.np + W.mlap) * nt + W.it] = make_double2(tl.re, tl.im);
        } else if (LAYOUT == 3) {
            // [z][m][point]: the same, the points of the launch in the place of the times of one radius
            if (W.live) totlap[((size_t)z * P.np + W.mlap) * nt + W.it] = make_double2(tl.re, tl.im);
        } else if (LAYOUT == 2) {
            if (W.live) totlap[((size_t)W.pidx * nz + z) * P.np + W.mlap] = make_double2(tl.re, tl.im);
        } else {
            if (totlap && W.live) totlap[((size_t)pt * nz + z) * P.np + W.mlap] = make_double2(tl.re, tl.im);   // (ucf_debug_stages only)
            const double hval = dehoog_wave(tl, P.M, P.alpha, P.logtol, tD, tee, lane, st);      // :219-223
            const double dval = dehoog_wave(cmul(tl, p), P.M, P.alpha, P.logtol, tD, tee, lane, st) * tD;   // :225-230
            if (lane == 0) {
                hout[(size_t)pt * P.nz_out + P.z_off + z] = hval;
                dhout[(size_t)pt * P.nz_out + P.z_off + z] = dval;
            }
        }
    }
}

// Parameter block of a work item: the launch's own (kernel argument), or -- parameter-batched launches, MULTI --
// entry pidx / ppp of a device array: every plan of the batch owns ppp consecutive points.  The plans of such a
// launch share everything that shapes the work (model family, M, k, R, nacc, ord, alpha, tol, depths count), so
// the kernel-argument block P0 (plan 0) still sizes LDS, loops and tables; only the evaluators read Pv.
template <bool MULTI>
UCF_DEV const ucf_dev_params& item_params(const ucf_dev_params& P0, const ucf_dev_params* __restrict__ Pv, int pidx, int ppp)
{
    if constexpr (MULTI) return Pv[pidx / ppp];
    else return P0;
}

// State handed from integrate_kernel to point_kernel (fast flavour): per work item
// [(R + 1 + nacc) * nz][64] complex = level sums | area of the interval in progress | finished areas,
// plus the number of abscissae already integrated.
UCF_DEV size_t state_slots(const ucf_dev_params& P) { return (size_t)(P.R + 1 + P.nacc) * P.nz; }

template <int FAMILY, int LAYOUT, bool MULTI>
__global__ void __launch_bounds__(UCF_WAVE, 2)
point_kernel(const ucf_dev_params P0, int npts, int per_point, int nr, int nsv, int svmin,
             const double* __restrict__ tDv, const double* __restrict__ rDv, const int* __restrict__ svv,
             const double2* __restrict__ tab, double* __restrict__ hout, double* __restrict__ dhout, ucf_stats* st,
             int nt, int ir0, int nrc, double2* __restrict__ totlap, double2* __restrict__ glscr,
             double2* __restrict__ state, const int* __restrict__ ndone, const int* __restrict__ todo,
             const ucf_dev_params* __restrict__ Pv, int ppp, int pbase)
{
    extern __shared__ lds_c lds[];
    const int lane = threadIdx.x;
    const int nz = P0.nz, R = P0.R, nacc = P0.nacc, N = P0.N, ngl = P0.ngl;
    const int nabs = N + nacc * ngl;
    lds_c* accTS = lds;                                     // [R][nz]  level sums
    lds_c* accCur = lds + (size_t)R * nz * UCF_WAVE;        // [nz]     area of the J0 interval being integrated
    lds_c* scr = accCur + (size_t)nz * UCF_WAVE;            // [max(2 nacc, R)] quarter-wave slots: Wynn columns / Neville
    lds_c* accGL = scr + (size_t)(2 * nacc > R ? 2 * nacc : R) * UCF_PART;   // [nacc][nz] finished areas (if kept in LDS)
    // finished interval areas: in LDS while that does not cost occupancy (nz = 1), else in an L2-resident
    // global scratch slot of this (then persistent, grid-strided) workgroup: [nacc][nz][64] complex;
    // when resuming after integrate_kernel they stay where that kernel put them (the item's state), and the
    // items are the ones that kernel listed as unfinished: todo = [count | item, item, ...]
    const bool resume = (state != nullptr);
    const int nloop = resume ? todo[0] : npts;
    const bool areas_lds = (glscr == nullptr) && !resume;
    lds_c* fdbuf = accGL + (areas_lds ? (size_t)nacc * nz * UCF_WAVE : 0);
    double2* __restrict__ areas = areas_lds ? nullptr : glscr + (size_t)blockIdx.x * nacc * nz * UCF_WAVE;

    for (int wi = blockIdx.x; wi < nloop; wi += gridDim.x) {
        const int pt = resume ? todo[1 + wi] : wi;
        const work_item W = decode_item<LAYOUT>(P0, pt, lane, per_point, nr, nt, ir0, npts);
        const int it = W.it, ir = W.ir, mlap = W.mlap, pidx = W.pidx;
        const ucf_dev_params& P = item_params<MULTI>(P0, Pv, (LAYOUT == 3 ? W.plan * ppp : pidx) + pbase, ppp);
        bool need_lay1 = false;
        for (int z = 0; z < nz; z++) need_lay1 |= (P.zLay[z] == 1);
        const double tD = tDv[it], rD = rDv[ir];
        const int sv = (LAYOUT == 1) ? svmin : svv[it];
        const double2* __restrict__ row = tab + (size_t)(per_point ? pidx : (ir * nsv + (sv - svmin))) * nabs;
        const double tee = 2.0 * tD;                                                            // driver.f90:106,217
        const double sigma = P.alpha - P.logtol / (2.0 * tee);                                  // invlap.f90:165
        const cplx p = cmake(sigma, UCF_PI * mlap / tee);                                       // invlap.f90:168
        const cplx lt = lap_time(P, p);
        const double arg = P.j0z[sv - 1] / rD;                                                  // driver.f90:120
        const cplx lane_aux = (FAMILY == 5) ? hstorage_A0(P, p) : cmake(0.0, 0.0);

        int n = 0;
        if (resume) {
            double2* __restrict__ sti = state + (size_t)pt * state_slots(P0) * UCF_WAVE;
            for (int s = 0; s < (R + 1) * nz; s++) lds[s * UCF_WAVE + lane] = sti[(size_t)s * UCF_WAVE + lane];
            areas = sti + (size_t)(R + 1) * nz * UCF_WAVE;
            n = ndone[pt];
        } else {
            for (int s = 0; s < (R + 1) * nz; s++) lds_st(lds, s, lane, cmake(0.0, 0.0));
        }

        // one pass over all abscissae: n < N tanh-sinh on [0,arg] feeding every Richardson level
        // (driver.f90:129-157); n >= N Gauss-Lobatto between successive J0 zeros (:187-203)
        // val = a*J0(a rD) * f(a,p,z) * lapTime(p)  (lhs.f90:118), accumulated into the level sums /
        // the current interval's area
        auto accumulate = [&](int n, double aj, int z, cplx f) {
            const cplx val = cmul(rscale(aj, f), lt);
            if (n < N) {
                const int n1 = n + 1;
                for (int j = 1; j <= R; j++) {
                    const int sh = R - j;
                    if ((n1 & ((1 << sh) - 1)) == 0) {                                          // driver.f90:150
                        const double w = P.ts_w[(size_t)(j - 1) * N + ((n1 >> sh) - 1)];
                        const int slot = (j - 1) * nz + z;
                        lds_st(accTS, slot, lane, cadd(lds_ld(accTS, slot, lane), rscale(w, val)));
                    }
                }
            } else {
                const int g = n - N;
                const int jj = g / ngl, m = g - jj * ngl;
                cplx acc = cadd(lds_ld(accCur, z, lane), cscale(val, P0.tab_premul ? 1.0 : P.gl_w[m]));   // :201-202
                if (m == ngl - 1) {
                    const double lob = P.j0z[sv + jj - 1] / rD;
                    const double hib = P.j0z[sv + jj] / rD;
                    acc = rscale((hib - lob) / 2.0, acc);
                    if (areas_lds) lds_st(accGL, jj * nz + z, lane, acc);
                    else areas[(size_t)(jj * nz + z) * UCF_WAVE + lane] = make_double2(acc.re, acc.im);
                    acc = cmake(0.0, 0.0);
                }
                lds_st(accCur, z, lane, acc);
            }
        };
        for (; n < nabs; n++) {
            const double2 aa = row[n];
            sample_common S;
            S.clean3 = resume;
            sample_prepare<FAMILY>(P, aa.x, p, need_lay1, S, fdbuf, lane, lane_aux);
            for (int z = 0; z < nz; z++) {
                const cplx f = sample_z<FAMILY>(P, S, z);
                accumulate(n, aa.y, z, f);
            }
        }
        finish_item<LAYOUT, UCF_PART>(P, accTS, scr, areas_lds ? accGL : nullptr, areas, arg, W, pt, tD, tee, p, st, nt, ir0, totlap,
                                      hout, dhout);
    }
}

#if UCF_FAST
// ------------------------------------------------------------------ lapTime(p) x the evaluators' constants, once per (time, p)
// The pumping-schedule multiplier (time.f90:34-122) depends on the time and the Laplace index only -- not on the radius, not
// on the abscissa: one small launch forms it for every (row, m) of the call, row = index into the call's tD array (the time
// of a grid, the point of a list), times everything the fast evaluators leave out of their samples (fast_scale; model 2:
// A0(p) / (p tDb + 1), laplace_hankel_solutions.f90:267-268).  integrate_kernel loads the value instead of carrying the
// code of eight schedule types, their complex exponentials and the Amos K0 / K1 through its set-up: ltab[m][nrows].
template <int FAMILY, bool MULTI>
__global__ void __launch_bounds__(256)
laptime_kernel(const ucf_dev_params P0, int nrows, const double* __restrict__ tDv, double2* __restrict__ ltab,
               const ucf_dev_params* __restrict__ Pv, int ppp, int pbase)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)nrows * P0.np) return;
    const int m = (int)(gid / nrows), row = (int)(gid % nrows);
    const ucf_dev_params& P = item_params<MULTI>(P0, Pv, row + pbase, ppp);
    const double tee = 2.0 * tDv[row];                                                          // driver.f90:106,217
    const double sigma = P.alpha - P.logtol / (2.0 * tee);                                      // invlap.f90:165
    const cplx p = cmake(sigma, UCF_PI * m / tee);                                              // invlap.f90:168
    cplx lt = cscale(lap_time(P, p), fast_scale<FAMILY>(P));
    if (FAMILY == 5) lt = cmul(lt, cdiv(hstorage_A0(P, p), caddr(cscale(p, P.hs_tDb), 1.0)));
    ltab[gid] = make_double2(lt.re, lt.im);
}

// ------------------------------------------------------------------ the integration kernel (fast flavour)
// The abscissa loop of point_kernel on its own (driver.f90:129-157,187-203 with the fast evaluators of
// ucf_fastpath.h): same work items, same lanes, same accumulation order.  It owns nothing but the loop, so
// that it needs half the registers and a quarter of the LDS of point_kernel and runs at twice the
// occupancy: a wave can issue one instruction per 4-cycle slot at best and a dependent fp64 result takes
// ~2.2 slots, so the VALU is only kept busy by waves, not by one wave's instruction stream.
// The level sums, the running interval area and the finished interval areas go to the item's state
// (HBM, written once, coalesced 1 KB per slot); point_kernel resumes from there: remaining abscissae
// with the generic evaluator (the overflow regime the fast one leaves alone), Richardson, Wynn, de Hoog.
#ifndef UCF_FOLD_WAVES
#define UCF_FOLD_WAVES 5
#endif
#ifndef UCF_UNFOLD_WAVES
#define UCF_UNFOLD_WAVES 4
#endif
// parts (2^k) of the work items of the last round of a launch (launch_transform_)
#ifndef UCF_TAIL_LSPLIT_DEFAULT
#define UCF_TAIL_LSPLIT_DEFAULT 3
#endif
// constants of sincos_tab_ / exp_tab_ kept in VGPRs (sc_ctx::kv)
#ifndef UCF_KV
#define UCF_KV(FAMILY, FOLD) ((FAMILY) == 4 ? 0 : 4)
#endif
// WAVES per SIMD the register budget is cut for: as many as the LDS footprint admits (6 for nz = 1 at R = 4,
// fully penetrating; 5; else 4)
// UCF_IWPB waves per workgroup: they share nothing but the sin/cos table of sincos_tab_ in LDS (4 KB, copied once from
// the plan's tables; one barrier, before the work loop); every wave keeps walking its own work items.
template <int FAMILY, int LAYOUT, int WAVES, bool MULTI, bool FOLD, bool LAY3, int NZC = 0, bool LAY1 = true, bool NOFOLD = false>      // NZC = 1: one depth per launch, known at compile time
__global__ void __launch_bounds__(UCF_WAVE * UCF_IWPB, WAVES)
integrate_kernel(const ucf_dev_params P0, int npts, int per_point, int nr, int nsv, int svmin,
                 const double* __restrict__ tDv, const double* __restrict__ rDv, const int* __restrict__ svv,
                 const double2* __restrict__ tab, int nt, int ir0, double2* __restrict__ state, int* __restrict__ ndone,
                 int* __restrict__ todo, const ucf_dev_params* __restrict__ Pv, int ppp, int pbase, int lsplit,
                 const double2* __restrict__ ltab, int nrows, int nheadw, int nworkw, int* __restrict__ wcount)
{
    extern __shared__ lds_c lds[];
#ifdef UCF_K1_ASSUME
    const ucf_dev_params& P = P0;
    UCF_K1_ASSUME      // tools/: specialise a probe build to one plan shape to read its inner loop
#endif
    // (the wave's index within the workgroup is wave-uniform, but only readfirstlane tells the compiler: without it the
    //  work item, its abscissa row and the row's entries live in VGPRs and are fetched by vector loads)
    const int lane = threadIdx.x & (UCF_WAVE - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / UCF_WAVE);
    constexpr int EF = (FAMILY == 5) ? 1 : FAMILY;      // the evaluator: model 2 runs family 1's with its own lane constant
    const int nz = NZC ? NZC : P0.nz, R = P0.R, nacc = P0.nacc, N = P0.N, ngl = P0.ngl;
    __builtin_assume(nz >= 1 && R >= 1 && nacc >= 1 && ngl >= 1);     // (the launcher's business: no loop guards in the kernel)
    const int nabs = N + nacc * ngl;
    // LDS: [256] sin/cos table | per wave: [R][nz] level sums, [nz] area of the J0 interval being integrated
    {
        const double2* __restrict__ gt = (const double2*)P0.sc_tab;
        for (int k = threadIdx.x; k < UCF_SC_ENTRIES; k += UCF_WAVE * UCF_IWPB) lds[k] = gt[k];
    }
    __syncthreads();
    const lds_c* sct = lds;
    // (NZC = 1, 2: the running areas live in registers and the wave's LDS holds the level sums only -- for two depths at
    //  R = 4 that is 8 instead of 10 KB per wave, which is what lets a fourth wave per SIMD in: 4 workgroups of 38 KB per CU)
    const int lslots = (NZC ? R : R + 1) * nz;
    lds_c* const wlds = lds + UCF_SC_ENTRIES + (size_t)wv * lslots * UCF_WAVE;
    lds_c* accTS = wlds;                                    // [R][nz]  level sums
    lds_c* accCur = wlds + (size_t)R * nz * UCF_WAVE;       // [nz]     area of the J0 interval being integrated (NZC = 0 only)
    // A work item may be cut into parts of whole quadrature units (the tanh-sinh part, then the J0 intervals: every
    // level sum and every interval area is formed by ONE part, in the reference's order -- same bits whatever the cut):
    // small launches (a shard of a strong-scaling run) then still fill the chip and end with a short tail.
    // The first nhead items run in 2^lsplit parts, the LAST npts - nhead items -- the ones the dispatcher hands out when
    // nothing is left to refill a SIMD, and a wave does not run faster on an emptier SIMD -- in 2^ltail: the drain of the
    // launch lasts as long as a part, not as an item, and only those items pay a part's set-up more than once.
    // (lsplit = lsplit | ltail << 8; nheadw = nhead << lsplit work units of the leading items, nworkw = all of them)
    // The grid is PERSISTENT -- about as many workgroups as the chip holds -- and every wave draws its work units from one
    // counter (lane 0's atomic, broadcast): a workgroup's launch, its copy of the tables and the barrier are paid once per
    // resident wave instead of once per work unit, units are handed out in order (the finer parts of the last items last)
    // and a wave that ends early simply takes the next one.  Exit: the counter has passed nworkw -- every wave gets there.
    // (wcount = NULL, diagnostic UCF_PERSIST=0: the grid covers the units, workgroup b takes units 4b .. 4b + 3 and ends)
    int ustat = blockIdx.x * UCF_IWPB + wv - gridDim.x * UCF_IWPB;
    auto next_unit = [&]() {
        if (!wcount) return ustat += gridDim.x * UCF_IWPB;
        int u = 0;
        if (lane == 0) u = __hip_atomic_fetch_add(wcount, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __builtin_amdgcn_readfirstlane(u);
    };
    for (int wi = next_unit(); wi < nworkw; wi = next_unit()) {
        const bool tail = wi >= nheadw;
        const int ls = tail ? (lsplit >> 8) : (lsplit & 255), nsplit = 1 << ls;
        const int wj = tail ? wi - nheadw : wi;
        const int pt = (tail ? (nheadw >> (lsplit & 255)) : 0) + (wj >> ls), sub = wj & (nsplit - 1);
        // abscissae [n0, n1) of part `sub`: boundaries at N + j ngl, j the nearest to an even cut of the nabs abscissae
        auto part_bound = [&](int k) {
            if (k <= 0) return 0;
            if (k >= nsplit) return nabs;
            int j = (k * nabs / nsplit - N + ngl / 2) / ngl;
            j = j < 0 ? 0 : (j > nacc ? nacc : j);
            return N + j * ngl;
        };
        const int n0 = part_bound(sub), n1 = part_bound(sub + 1);
        if (n1 <= n0 && sub != 0) continue;
#ifdef UCF_TIMELINE      /* diagnostic build (tools/timeline.py): when and where every work item ran */
        const unsigned long long tl_t0 = __builtin_amdgcn_s_memtime();
#endif
        const work_item W = decode_item<LAYOUT>(P0, pt, lane, per_point, nr, nt, ir0, npts);
        const ucf_dev_params& P = item_params<MULTI>(P0, Pv, (LAYOUT == 3 ? W.plan * ppp : W.pidx) + pbase, ppp);
        bool need_lay1 = false, need_lay3 = false, need_lay12 = false;
        for (int z = 0; z < nz; z++) { need_lay1 |= (P.zLay[z] == 1); need_lay3 |= (P.zLay[z] == 3); need_lay12 |= (P.zLay[z] != 3); }
        const double tD = tDv[W.it], rD = rDv[W.ir];
        const int sv = (LAYOUT == 1) ? svmin : svv[W.it];
        const double2* __restrict__ row = tab + (size_t)(per_point ? W.pidx : (W.ir * nsv + (sv - svmin))) * nabs;
        const double tee = 2.0 * tD;                                                            // driver.f90:106,217
        const double sigma = P.alpha - P.logtol / (2.0 * tee);                                  // invlap.f90:165
        const cplx p = cmake(sigma, UCF_PI * W.mlap / tee);                                     // invlap.f90:168
        // (the pumping-schedule multiplier lapTime(p) is constant over the abscissae: the sums are formed without it and
        //  scaled when they leave the kernel -- one complex product per sample less)
        // times the constants the evaluators leave out: laptime_kernel formed it for every (row of tD, m)
        // (lanes past the last Laplace sample -- LAYOUT 0 / 2, 2M+1 not a multiple of 64 -- compute on the last sample's entry:
        //  the table has np rows)
        const double2 ltv = ltab[(size_t)(W.mlap < P0.np ? W.mlap : P0.np - 1) * nrows + W.it];
        const cplx lt = cmake(ltv.x, ltv.y);
        const lane_consts LC = make_lane_consts(P, p, lt);
        double2* __restrict__ sti = state + (size_t)pt * state_slots(P0) * UCF_WAVE;
        double2* __restrict__ areas = sti + (size_t)(R + 1) * nz * UCF_WAVE;
        for (int s = 0; s < lslots; s++) lds_st(wlds, s, lane, cmake(0.0, 0.0));
        fast_common F;
        sc_ctx_init(F.sc, sct, UCF_KV(FAMILY, FOLD));
        // the closure's unscaled reciprocal (fast_common_terms) wants |xi| = |eta| |xifac| far from overflow: an item with a
        // lane beyond that goes to point_kernel whole
        const bool lane_ok = FAMILY != 2 || (fabs(LC.xifac.re) + fabs(LC.xifac.im) < 1.0e90);
        const int nlim = (__builtin_amdgcn_ballot_w64(!lane_ok) == 0) ? n1 : n0;

        // abscissae increase monotonically (tanh-sinh on [0,arg], then the J0 intervals), so does Re(eta):
        // the fast evaluation is valid for a leading run of abscissae; point_kernel finishes the rest.
        // The row entry of the next abscissa is requested one iteration ahead (scalar loads).
        int n = n0;
        double2 aa = row[n0 < nabs ? n0 : 0];
        int m = 0, jj = n0 > N ? (n0 - N) / ngl : 0;   // Gauss-Lobatto node and J0 interval of abscissa n >= N
        cplx acc0 = cmake(0.0, 0.0), acc1 = cmake(0.0, 0.0);      // running area of the interval: registers when nz = 1 (or
        for (; n < nlim; n++) {                   // NZC = 2: two depths known at compile time), else accCur[z] in LDS
            const double2 nxt = row[n + 1 < nabs ? n + 1 : n];
            const bool ts = n < N;
            F.sc.salt = n;
            if (!fast_eta_wave<EF>(P, LC, aa.x, F)) break;                                       // (every lane is live here)
            fast_common_terms<EF, FOLD, LAY3, LAY1, NOFOLD>(P, LC, aa.x, need_lay1, F, need_lay3, need_lay12);
            const int n1 = n + 1;
            int tz = __builtin_ctz(n1);
            if (tz > R - 1) tz = R - 1;
            for (int z = 0; z < nz; z++) {
                // val = a*J0(a rD) * f(a,p,z) [* lapTime(p): at the end]                         (lhs.f90:118)
                // (Gauss-Lobatto part: aa.y carries the node's weight, abscissa_kernel)
                const cplx fz = fast_sample_z<EF, FOLD, LAY3, LAY1, NOFOLD>(P, F, z);
                if (ts) {
                    const cplx val = rscale(aa.y, fz);
                    // tanh-sinh on [0, arg]: abscissa n+1 belongs to level j when 2^(R-j) divides it  (driver.f90:129-157)
                    for (int sh = 0; sh <= tz; sh++) {
                        const int j = R - sh;
                        const double wl = P.ts_w[(size_t)(j - 1) * N + ((n1 >> sh) - 1)];
                        const int slot = (j - 1) * nz + z;
                        lds_st(accTS, slot, lane, cadd(lds_ld(accTS, slot, lane), rscale(wl, val)));
                    }
                } else if (nz == 1 || (NZC == 2 && z == 0)) {
                    acc0 = cmake(__builtin_fma(fz.re, aa.y, acc0.re), __builtin_fma(fz.im, aa.y, acc0.im));       // :201-202
                } else if (NZC == 2) {
                    acc1 = cmake(__builtin_fma(fz.re, aa.y, acc1.re), __builtin_fma(fz.im, aa.y, acc1.im));
                } else {
                    const cplx a1 = lds_ld(accCur, z, lane);
                    lds_st(accCur, z, lane, cmake(__builtin_fma(fz.re, aa.y, a1.re), __builtin_fma(fz.im, aa.y, a1.im)));
                }
            }
            if (!ts && ++m == ngl) {
                // Gauss-Lobatto between successive J0 zeros: the interval is complete           (driver.f90:187-203)
                const double lob = P.j0z[sv + jj - 1] / rD;
                const double hib = P.j0z[sv + jj] / rD;
                const double hw = (hib - lob) / 2.0;
                for (int z = 0; z < nz; z++) {
                    const cplx run = (nz == 1 || (NZC == 2 && z == 0)) ? acc0 : (NZC == 2 ? acc1 : lds_ld(accCur, z, lane));
                    const cplx ar = cmul(rscale(hw, run), lt);
                    areas[(size_t)(jj * nz + z) * UCF_WAVE + lane] = make_double2(ar.re, ar.im);
                    if (nz != 1 && NZC != 2) lds_st(accCur, z, lane, cmake(0.0, 0.0));
                }
                acc0 = acc1 = cmake(0.0, 0.0);
                m = 0;
                jj++;
            }
            aa = nxt;
        }
        // What leaves the kernel: the level sums (the part that owns the tanh-sinh abscissae) and the finished areas (above).
        // A part that had to stop hands over at the START of the J0 interval it was in (inside the tanh-sinh part: at the
        // abscissa itself, the level sums are additive): point_kernel redoes that interval whole, so that no running area
        // travels and the parts of an item need not agree on one.  ndone[pt] was set to nabs by the launcher; the smallest
        // hand-over point of the item's parts counts, the part that lowers it first lists the item.
        if (sub == 0) {
            for (int s = 0; s < R * nz; s++) {
                const cplx v = cmul(lds_ld(wlds, s, lane), lt);
                sti[(size_t)s * UCF_WAVE + lane] = make_double2(v.re, v.im);
            }
            // (the slots of the running areas: nothing travels in them, point_kernel starts its interval afresh)
            for (int s = R * nz; s < (R + 1) * nz; s++) sti[(size_t)s * UCF_WAVE + lane] = make_double2(0.0, 0.0);
        }
#ifdef UCF_TIMELINE      /* into the (unread) running-area slot: lane 0 = (start, end) in shader clocks, lane 1 = (HW_ID, part) */
        if (sub == 0) {
            const unsigned long long tl_t1 = __builtin_amdgcn_s_memtime();
            const unsigned hwid = __builtin_amdgcn_s_getreg(63492);      // HW_REG_HW_ID: wave, SIMD, CU, SH, SE ...
            if (lane == 0) sti[(size_t)(R * nz) * UCF_WAVE + 0] = make_double2(__longlong_as_double((long long)tl_t0), __longlong_as_double((long long)tl_t1));
            if (lane == 1) sti[(size_t)(R * nz) * UCF_WAVE + 1] = make_double2((double)hwid, (double)__builtin_amdgcn_s_getreg(63508));   // XCC_ID (reg 20)
        }
#endif
        if (n < n1 && lane == 0) {
            const int nrep = (n < N) ? n : n - m;
            if (atomicMin(&ndone[pt], nrep) == nabs) todo[1 + atomicAdd(&todo[0], 1)] = pt;      // point_kernel takes it from here
        }
    }
}

#endif   // UCF_FAST

// ------------------------------------------------------------------ the integration kernel, generic evaluators
// Same split for the reference-order evaluators (faithful flavour; models without a fast evaluator): the abscissa
// loop alone, every abscissa (these evaluators reproduce the reference's Inf/NaN behaviour themselves, nothing is
// left for point_kernel), state to HBM, finish_kernel does the tail.  The out-of-line libm calls keep it at
// 3 waves/SIMD instead of the 2 of the monolithic point_kernel.
template <int FAMILY, int LAYOUT>
__global__ void __launch_bounds__(UCF_WAVE, 3)
integrate_generic_kernel(const ucf_dev_params P, int npts, int per_point, int nr, int nsv, int svmin,
                         const double* __restrict__ tDv, const double* __restrict__ rDv, const int* __restrict__ svv,
                         const double2* __restrict__ tab, int nt, int ir0, double2* __restrict__ state, int* __restrict__ ndone)
{
    extern __shared__ lds_c lds[];
    const int lane = threadIdx.x;
    const int nz = P.nz, R = P.R, nacc = P.nacc, N = P.N, ngl = P.ngl;
    const int nabs = N + nacc * ngl;
    lds_c* accTS = lds;                                     // [R][nz]  level sums
    lds_c* accCur = lds + (size_t)R * nz * UCF_WAVE;        // [nz]     area of the J0 interval being integrated
    lds_c* fdbuf = accCur + (size_t)nz * UCF_WAVE;          // FAMILY 4, faithful: [2*order] Thomas sweep
    bool need_lay1 = false;
    for (int z = 0; z < nz; z++) need_lay1 |= (P.zLay[z] == 1);
    for (int pt = blockIdx.x; pt < npts; pt += gridDim.x) {
        const work_item W = decode_item<LAYOUT>(P, pt, lane, per_point, nr, nt, ir0, npts);
        const double tD = tDv[W.it], rD = rDv[W.ir];
        const int sv = (LAYOUT == 1) ? svmin : svv[W.it];
        const double2* __restrict__ row = tab + (size_t)(per_point ? W.pidx : (W.ir * nsv + (sv - svmin))) * nabs;
        const double tee = 2.0 * tD;                                                            // driver.f90:106,217
        const double sigma = P.alpha - P.logtol / (2.0 * tee);                                  // invlap.f90:165
        const cplx p = cmake(sigma, UCF_PI * W.mlap / tee);                                     // invlap.f90:168
        const cplx lt = lap_time(P, p);
        const cplx lane_aux = (FAMILY == 5) ? hstorage_A0(P, p) : cmake(0.0, 0.0);
        double2* __restrict__ sti = state + (size_t)pt * state_slots(P) * UCF_WAVE;
        double2* __restrict__ areas = sti + (size_t)(R + 1) * nz * UCF_WAVE;
        for (int s = 0; s < (R + 1) * nz; s++) lds_st(lds, s, lane, cmake(0.0, 0.0));
        int m = 0, jj = 0;
        for (int n = 0; n < nabs; n++) {
            const double2 aa = row[n];
            sample_common S;
            S.clean3 = false;
            sample_prepare<FAMILY>(P, aa.x, p, need_lay1, S, fdbuf, lane, lane_aux);
            const bool ts = n < N;
            for (int z = 0; z < nz; z++) {
                const cplx val = cmul(rscale(aa.y, sample_z<FAMILY>(P, S, z)), lt);              // lhs.f90:118
                if (ts) {
                    const int n1 = n + 1;                                                       // driver.f90:150
                    for (int j = 1; j <= R; j++) {
                        const int sh = R - j;
                        if ((n1 & ((1 << sh) - 1)) == 0) {
                            const double w = P.ts_w[(size_t)(j - 1) * N + ((n1 >> sh) - 1)];
                            const int slot = (j - 1) * nz + z;
                            lds_st(accTS, slot, lane, cadd(lds_ld(accTS, slot, lane), rscale(w, val)));
                        }
                    }
                } else {
                    cplx acc = cadd(lds_ld(accCur, z, lane), cscale(val, P.tab_premul ? 1.0 : P.gl_w[m]));   // :201-202
                    if (m == ngl - 1) {
                        const double lob = P.j0z[sv + jj - 1] / rD;
                        const double hib = P.j0z[sv + jj] / rD;
                        acc = rscale((hib - lob) / 2.0, acc);
                        areas[(size_t)(jj * nz + z) * UCF_WAVE + lane] = make_double2(acc.re, acc.im);
                        acc = cmake(0.0, 0.0);
                    }
                    lds_st(accCur, z, lane, acc);
                }
            }
            if (!ts && ++m == ngl) { m = 0; jj++; }
        }
        for (int s = 0; s < (R + 1) * nz; s++) sti[(size_t)s * UCF_WAVE + lane] = lds[s * UCF_WAVE + lane];
        if (lane == 0) ndone[pt] = nabs;
    }
}

// ------------------------------------------------------------------ the finishing kernel
// Tail of every work item integrate_kernel completed (all of them outside the overflow regime): state ->
// finish_item.  Model independent and small, so it runs at full occupancy; PART is chosen at launch from the
// LDS footprint ((R+1) nz slots of level sums + max(2 nacc, R) scratch columns of PART lanes).
// MODE 1 / 2 (finish_item): the pass over all items with the unguarded epsilon table, then the pass over the (item, depth)
// pairs it listed in `defer` with the guarded one.
template <int LAYOUT, int PART, bool WREG, int MODE = 0>
__global__ void __launch_bounds__(UCF_WAVE, (LAYOUT == 0 || (WREG && MODE != 1)) ? 3 : 4)      // LAYOUT 0 carries de Hoog, WREG the guarded epsilon table
finish_kernel(const ucf_dev_params P, int npts, int per_point, int nr, int nsv, int svmin,
              const double* __restrict__ tDv, const double* __restrict__ rDv, const int* __restrict__ svv,
              double* __restrict__ hout, double* __restrict__ dhout, ucf_stats* st, int nt, int ir0,
              double2* __restrict__ totlap, const double2* __restrict__ state, const int* __restrict__ ndone, int* __restrict__ defer)
{
    extern __shared__ lds_c lds[];
    const int lane = threadIdx.x;
    const int nz = P.nz, R = P.R, nacc = P.nacc;
    const int nabs = P.N + nacc * P.ngl;
    lds_c* accTS = lds;
    lds_c* scr = lds + (size_t)R * nz * UCF_WAVE;
    const int nloop = (MODE == 2) ? defer[0] : npts;
    for (int wi = blockIdx.x; wi < nloop; wi += gridDim.x) {
        const int pt = (MODE == 2) ? defer[1 + wi] / nz : wi;
        const int zsel = (MODE == 2) ? defer[1 + wi] % nz : 0;
        if (MODE != 2 && ndone[pt] < nabs) continue;
        const work_item W = decode_item<LAYOUT>(P, pt, lane, per_point, nr, nt, ir0, npts);
        const double tD = tDv[W.it], rD = rDv[W.ir];
        const int sv = (LAYOUT == 1) ? svmin : svv[W.it];
        const double tee = 2.0 * tD;
        const double sigma = P.alpha - P.logtol / (2.0 * tee);
        const cplx p = cmake(sigma, UCF_PI * W.mlap / tee);
        const double arg = P.j0z[sv - 1] / rD;                                                  // driver.f90:120
        const double2* __restrict__ sti = state + (size_t)pt * state_slots(P) * UCF_WAVE;
        for (int s = 0; s < R * nz; s++) lds[s * UCF_WAVE + lane] = sti[(size_t)s * UCF_WAVE + lane];
        finish_item<LAYOUT, PART, WREG, MODE>(P, accTS, scr, nullptr, sti + (size_t)(R + 1) * nz * UCF_WAVE, arg, W, pt, tD, tee, p, st, nt, ir0,
                                              totlap, hout, dhout, zsel, (MODE == 2) ? zsel + 1 : nz, (MODE == 1) ? defer : nullptr);
    }
}

// second half of LAYOUT 1: the transform arrives as [radius][z][m][time] (written coalesced by 64 consecutive
// times); a wave takes a tile of UCF_DH_TILE consecutive times of one radius, transposes it through LDS
// ([m][tile], rows padded to dodge bank conflicts) and inverts one time after the other with lane = m
// (driver.f90:217-230).
// For 2M+1 <= 64 the rhombus runs per vector (h and dh of each time) and leaves the continued-fraction
// coefficients in LDS (h's in the column the input came from, dh's in a second tile); then 2 x UCF_DH_TILE lanes
// each finish one vector.  The recurrence used to be uniform work repeated by all 64 lanes for every vector.
// tile of 4 times and a register budget of 4 waves per SIMD (measured on C2: 2.44 ms at 8 / 2, 2.06 ms at 4 / 4: the
// rhombus is a dependent chain per vector, resident waves are what hides it)
#ifndef UCF_DH_TILE
#define UCF_DH_TILE 4
#endif
#if UCF_TU_HAS(1) || UCF_TU_HAS(3)
#ifndef UCF_DH_WAVES
#define UCF_DH_WAVES 4
#endif
// BIG: more Laplace samples than lanes (M > 31) -- the chunked rhombus of dehoog_big and its registers get their own
// instantiation, so that the usual one is not allocated for them
template <int TU, bool BIG>          // (TU: a template parameter only so that two translation units may hold the kernel)
__global__ void __launch_bounds__(UCF_WAVE, BIG ? 4 : UCF_DH_WAVES)
dehoog_tiles_kernel(const ucf_dev_params P, int nt, int nr, int ir0, int nrc, const double* __restrict__ tDv,
                    const double2* __restrict__ totlap, double* __restrict__ hout, double* __restrict__ dhout,
                    ucf_stats* st)
{
    extern __shared__ lds_c lds[];            // [np][UCF_DH_TILE + 1] x 2, then 2 x UCF_DH_TILE flags
    const int lane = threadIdx.x;
    const int nz = P.nz, np = P.np;
    const int ntile = (nt + UCF_DH_TILE - 1) / UCF_DH_TILE;
    const long long nwork = (long long)nrc * ntile;
    const int pitch = UCF_DH_TILE + 1;
    lds_c* tileB = lds + (size_t)np * pitch;
    int* zflag = (int*)(tileB + (size_t)np * pitch);
    for (long long w = blockIdx.x; w < nwork; w += gridDim.x) {
        const int irl = (int)(w / ntile), it0 = (int)(w % ntile) * UCF_DH_TILE;
        const int ncur = (nt - it0 < UCF_DH_TILE) ? nt - it0 : UCF_DH_TILE;
        for (int z = 0; z < nz; z++) {
            const double2* src = totlap + ((size_t)irl * nz + z) * np * nt;
            for (int e = lane; e < np * UCF_DH_TILE; e += UCF_WAVE) {
                const int m = e / UCF_DH_TILE, tt = e % UCF_DH_TILE;
                if (tt < ncur) lds[m * pitch + tt] = src[(size_t)m * nt + it0 + tt];
            }
            __syncthreads();
            if constexpr (!BIG) {
                for (int tt = 0; tt < ncur; tt++) {
                    const double tD = tDv[it0 + tt];
                    const double tee = 2.0 * tD;
                    const double sigma = P.alpha - P.logtol / (2.0 * tee);
                    cplx tl = cmake(0.0, 0.0);
                    if (lane < np) { const lds_c v = lds[lane * pitch + tt]; tl = cmake(v.x, v.y); }
                    const cplx p = cmake(sigma, UCF_PI * lane / tee);
                    bool z0, z1;
                    dehoog_qd_wave(tl, P.M, lane, st, &z0, lds + tt, pitch);          // (column tt of the tile is in `tl` by now)
                    dehoog_qd_wave(cmul(tl, p), P.M, lane, st, &z1, tileB + tt, pitch);
                    if (lane == 0) { zflag[2 * tt] = z0; zflag[2 * tt + 1] = z1; }
                }
                __syncthreads();
                if (lane < 2 * ncur) {
                    const int tt = lane >> 1, which = lane & 1;
                    const int it = it0 + tt;
                    const double tD = tDv[it];
                    double val = 0.0;
                    if (!zflag[lane]) val = dehoog_cf_lane((which ? tileB : lds) + tt, pitch, P.M, P.alpha, P.logtol, tD, 2.0 * tD);
                    const size_t o = ((size_t)it * nr + ir0 + irl) * P.nz_out + P.z_off + z;
                    if (which) dhout[o] = val * tD;
                    else hout[o] = val;
                }
            } else {
                for (int tt = 0; tt < ncur; tt++) {
                    const int it = it0 + tt;
                    const double tD = tDv[it];
                    const double tee = 2.0 * tD;
                    const double sigma = P.alpha - P.logtol / (2.0 * tee);
                    double hval, dval;
                    auto ld = [&](int i) { const lds_c v = lds[i * pitch + tt]; return cmake(v.x, v.y); };
                    if (np <= 2 * UCF_WAVE) dehoog_big<2>(ld, np, P.M, P.alpha, P.logtol, sigma, tD, tee, lane, st, &hval, &dval);
                    else dehoog_big<4>(ld, np, P.M, P.alpha, P.logtol, sigma, tD, tee, lane, st, &hval, &dval);
                    if (lane == 0) {
                        const size_t o = ((size_t)it * nr + ir0 + irl) * P.nz_out + P.z_off + z;
                        hout[o] = hval;
                        dhout[o] = dval;
                    }
                }
            }
            __syncthreads();
        }
    }
}

#endif   // UCF_TU_HAS(1) || UCF_TU_HAS(3)

// second half of LAYOUTs 1 and 2: one wave per point of the chunk, lane = Laplace index (driver.f90:217-230).
// flat = 0: chunk-local point lp = it*nrc + irl, output at (it*nr + ir0 + irl);  flat = 1: lp is the point itself
// (tD per point when per_point, else tD[lp / nr]).
#if UCF_TU_HAS(2)
__global__ void __launch_bounds__(UCF_WAVE)
dehoog_points_kernel(const ucf_dev_params P, long long npc, int flat, int per_point, int nr, int ir0, int nrc,
                     const double* __restrict__ tDv, const double2* __restrict__ totlap, double* __restrict__ hout,
                     double* __restrict__ dhout, ucf_stats* st)
{
    const int lane = threadIdx.x;
    const int nz = P.nz;
    for (long long lp = blockIdx.x; lp < npc; lp += gridDim.x) {
        int it;
        size_t obase;
        if (flat) {
            it = per_point ? (int)lp : (int)(lp / nr);
            obase = (size_t)lp;
        } else {
            it = (int)(lp / nrc);
            obase = (size_t)it * nr + ir0 + (int)(lp % nrc);
        }
        const double tD = tDv[it];
        const double tee = 2.0 * tD;
        const double sigma = P.alpha - P.logtol / (2.0 * tee);
        for (int z = 0; z < nz; z++) {
            const double2* src = totlap + ((size_t)lp * nz + z) * P.np;
            double hval, dval;
            if (P.np <= UCF_WAVE) {
                const cplx p = cmake(sigma, UCF_PI * lane / tee);
                cplx tl = cmake(0.0, 0.0);
                if (lane < P.np) { const double2 v = src[lane]; tl = cmake(v.x, v.y); }
                hval = dehoog_wave(tl, P.M, P.alpha, P.logtol, tD, tee, lane, st);
                dval = dehoog_wave(cmul(tl, p), P.M, P.alpha, P.logtol, tD, tee, lane, st) * tD;
            } else {
                auto ld = [&](int i) { const double2 v = src[i]; return cmake(v.x, v.y); };
                if (P.np <= 2 * UCF_WAVE) dehoog_big<2>(ld, P.np, P.M, P.alpha, P.logtol, sigma, tD, tee, lane, st, &hval, &dval);
                else dehoog_big<4>(ld, P.np, P.M, P.alpha, P.logtol, sigma, tD, tee, lane, st, &hval, &dval);
            }
            if (lane == 0) {
                hout[obase * P.nz_out + P.z_off + z] = hval;
                dhout[obase * P.nz_out + P.z_off + z] = dval;
            }
        }
    }
}

#endif   // UCF_TU_HAS(2)

// ------------------------------------------------------------------ stage-hook kernels
// lap_hank_soln for a list of abscissae: block = abscissa, lane = p index; fp[n_a][nz][np]
template <int FAMILY>
__global__ void __launch_bounds__(UCF_WAVE)
samples_kernel(const ucf_dev_params P, int n_a, const double* __restrict__ av, double rD,
               const double* __restrict__ pv, double* __restrict__ fp)
{
    extern __shared__ lds_c lds[];
    const int lane = threadIdx.x;
    const int ia = blockIdx.x;
    if (ia >= n_a) return;
    const bool live = lane < P.np;
    const cplx p = live ? cmake(pv[2 * lane], pv[2 * lane + 1]) : cmake(1.0, 0.0);
    const double a = av[ia];
    bool need_lay1 = false;
    for (int z = 0; z < P.nz; z++) need_lay1 |= (P.zLay[z] == 1);
    const cplx lt = lap_time(P, p);
    const double aj = a * j0(a * rD);
    sample_common S;
    S.clean3 = false;
    bool fast = false;
#if UCF_FAST
    fast_common F;
    {   // sin/cos table of the fast evaluators (the kernel's dynamic LDS holds it)
        const double2* __restrict__ gt = (const double2*)P.sc_tab;
        for (int k = threadIdx.x; k < UCF_SC_ENTRIES; k += UCF_WAVE) lds[k] = gt[k];
        __syncthreads();
        sc_ctx_init(F.sc, lds, 0);
        F.sc.salt = ia;
    }
    constexpr int EF = (FAMILY == 5) ? 1 : FAMILY;
    cplx c5 = cmake(1.0, 0.0);
    {
        const lane_consts LC = make_lane_consts(P, p, lt);
        const bool lane_ok = FAMILY != 2 || (fabs(LC.xifac.re) + fabs(LC.xifac.im) < 1.0e90);      // as integrate_kernel
        fast = __all(lane_ok && fast_prepare<EF>(P, LC, a, need_lay1, F));
        if (FAMILY == 5) c5 = cdiv(hstorage_A0(P, p), caddr(cscale(p, P.hs_tDb), 1.0));
    }
#endif
    const cplx lane_aux = (FAMILY == 5) ? hstorage_A0(P, p) : cmake(0.0, 0.0);
    if (!fast) sample_prepare<FAMILY>(P, a, p, need_lay1, S, lds, lane, lane_aux);
    for (int z = 0; z < P.nz; z++) {
        cplx f;
#if UCF_FAST
        if (fast) f = cmul(cscale(fast_sample_z<EF>(P, F, z), fast_scale<FAMILY>(P)), c5);
        else
#endif
            f = sample_z<FAMILY>(P, S, z);
        const cplx val = cmul(rscale(aj, f), lt);
        if (live) {
            const size_t o = (((size_t)ia * P.nz + z) * P.np + lane) * 2;
            fp[o] = val.re;
            fp[o + 1] = val.im;
        }
    }
}

// ucf_debug_stages: the state that the transform kernel of the PRODUCTION launch sequence left in the workspace, brought
// from its work-item order (decode_item: the layout the launcher chose) into [point][Laplace index][slot]
#if !UCF_FAST
template <int LAYOUT>
__global__ void __launch_bounds__(UCF_WAVE)
debug_gather_kernel(const ucf_dev_params P, int nwork, int per_point, int nr, int nt, int ir0, const double2* __restrict__ state,
                    const int* __restrict__ ndone, double2* __restrict__ out_state, int* __restrict__ out_ndone)
{
    const int lane = threadIdx.x;
    const int slots = (int)state_slots(P);
    for (int pt = blockIdx.x; pt < nwork; pt += gridDim.x) {
        const work_item W = decode_item<LAYOUT>(P, pt, lane, per_point, nr, nt, ir0, nwork);
        if (!W.live) continue;
        const size_t q = (LAYOUT == 1) ? (size_t)W.it * nr + W.ir : (size_t)W.pidx;
        for (int sl = 0; sl < slots; sl++)
            out_state[(q * P.np + W.mlap) * slots + sl] = state[((size_t)pt * slots + sl) * UCF_WAVE + lane];
        out_ndone[q * P.np + W.mlap] = ndone[pt];
    }
}
int launch_debug_gather(const ucf_dev_params& dp, int layout, int nwork, int per_point, int nr, int nt, int ir0, const double* d_state,
                        const int* d_ndone, double* d_out_state, int* d_out_ndone, void* stream)
{
    const dim3 grid((unsigned)(nwork < 65536 ? nwork : 65536)), block(UCF_WAVE);
    hipStream_t s = (hipStream_t)stream;
    if (layout == 0) hipLaunchKernelGGL(debug_gather_kernel<0>, grid, block, 0, s, dp, nwork, per_point, nr, nt, ir0, (const double2*)d_state, d_ndone, (double2*)d_out_state, d_out_ndone);
    else if (layout == 1) hipLaunchKernelGGL(debug_gather_kernel<1>, grid, block, 0, s, dp, nwork, per_point, nr, nt, ir0, (const double2*)d_state, d_ndone, (double2*)d_out_state, d_out_ndone);
    else if (layout == 3) hipLaunchKernelGGL(debug_gather_kernel<3>, grid, block, 0, s, dp, nwork, per_point, nr, nt, ir0, (const double2*)d_state, d_ndone, (double2*)d_out_state, d_out_ndone);
    else return UCF_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}
#endif

#if UCF_TU_HAS(1)
// ucf_debug_wynn: the register-resident epsilon table that finish_kernel runs (wynn_regs<UCF_WYNN_REGS>), lane = problem
__global__ void __launch_bounds__(UCF_WAVE, 3)
wynn_regs_kernel(int n, int nterms, const double* __restrict__ series, double* __restrict__ acc, int* __restrict__ status)
{
    const int i = blockIdx.x * UCF_WAVE + threadIdx.x;
    const int ii = i < n ? i : n - 1;
    cplx ser[UCF_WYNN_REGS];
#pragma unroll
    for (int k = 0; k < UCF_WYNN_REGS; k++) {
        ser[k] = cmake(0.0, 0.0);
        if (k < nterms) ser[k] = cmake(series[((size_t)ii * nterms + k) * 2], series[((size_t)ii * nterms + k) * 2 + 1]);
    }
    int stt = 0;
    const cplx r = wynn_regs<UCF_WYNN_REGS>(ser, nterms, &stt);
    if (i < n) {
        acc[2 * i] = r.re;
        acc[2 * i + 1] = r.im;
        status[i] = stt;
    }
}
int launch_wynn_regs(int n, int nterms, const double* d_series, double* d_acc, int* d_status, void* stream)
{
    if (nterms > UCF_WYNN_REGS) return UCF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(wynn_regs_kernel, dim3((n + UCF_WAVE - 1) / UCF_WAVE), dim3(UCF_WAVE), 0, (hipStream_t)stream, n, nterms, d_series, d_acc, d_status);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}
// ucf_debug_dehoog_tiles: dehoog_tiles_kernel itself -- the kernel every grid and every long point list ends with -- on a
// transform given as [m][n] (one radius, one depth, n times)
int launch_dehoog_tiles_hook(const ucf_dev_params& dp, int n, const double* d_tD, const double* d_totlap, double* d_h, double* d_dh, void* stream)
{
    const long long ntl = (n + UCF_DH_TILE - 1) / UCF_DH_TILE;
    const size_t dlds = 2 * (size_t)dp.np * (UCF_DH_TILE + 1) * sizeof(lds_c) + 2 * UCF_DH_TILE * sizeof(int);
    if (dp.np <= UCF_WAVE)
        hipLaunchKernelGGL((dehoog_tiles_kernel<1, false>), dim3((unsigned)ntl), dim3(UCF_WAVE), dlds, (hipStream_t)stream, dp, n, 1, 0, 1, d_tD,
                           (const double2*)d_totlap, d_h, d_dh, (ucf_stats*)nullptr);
    else
        hipLaunchKernelGGL((dehoog_tiles_kernel<1, true>), dim3((unsigned)ntl), dim3(UCF_WAVE), dlds, (hipStream_t)stream, dp, n, 1, 0, 1, d_tD,
                           (const double2*)d_totlap, d_h, d_dh, (ucf_stats*)nullptr);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}
#endif

#if !UCF_FAST
__global__ void __launch_bounds__(UCF_WAVE)
bessel_kernel(int n, const double* __restrict__ z, double* __restrict__ k, int* __restrict__ ierr)
{
    const int i = blockIdx.x * UCF_WAVE + threadIdx.x;
    if (i >= n) return;
    const k01 r = cbesk01_(cmake(z[2 * i], z[2 * i + 1]));
    k[4 * i] = r.k0.re; k[4 * i + 1] = r.k0.im; k[4 * i + 2] = r.k1.re; k[4 * i + 3] = r.k1.im;
    ierr[i] = r.ierr;
}

__global__ void __launch_bounds__(UCF_WAVE)
dehoog_kernel(int n, int M, double alpha, double logtol, const double* __restrict__ t, const double* __restrict__ tee,
              const double* __restrict__ fp, double* __restrict__ ft)
{
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    if (i >= n) return;
    const int np = 2 * M + 1;
    auto ld = [&](int k) { return (k < np) ? cmake(fp[((size_t)i * np + k) * 2], fp[((size_t)i * np + k) * 2 + 1]) : cmake(0.0, 0.0); };
    double v;
    if (np <= UCF_WAVE) {
        v = dehoog_wave(ld(lane), M, alpha, logtol, t[i], tee[i], lane, nullptr);
    } else if (np <= 2 * UCF_WAVE) {
        const cplx f2[2] = {ld(lane), ld(lane + 64)};
        v = dehoog_sets<2>(f2, M, alpha, logtol, t[i], tee[i], lane, nullptr);
    } else {
        const cplx f4[4] = {ld(lane), ld(lane + 64), ld(lane + 128), ld(lane + 192)};
        v = dehoog_sets<4>(f4, M, alpha, logtol, t[i], tee[i], lane, nullptr);
    }
    if (lane == 0) ft[i] = v;
}

// lane = problem index within the block
__global__ void __launch_bounds__(UCF_WAVE)
wynn_kernel(int n, int nterms, const double* __restrict__ series, double* __restrict__ acc, int* __restrict__ status)
{
    extern __shared__ lds_c lds[];
    const int lane = threadIdx.x;
    const int i = blockIdx.x * UCF_WAVE + lane;
    lds_c* colA = lds;
    lds_c* colB = lds + (size_t)nterms * UCF_PART;
    const int ii = i < n ? i : n - 1;
    int stt = 0;
    cplx r = cmake(0.0, 0.0);
    for (int part = 0; part < UCF_WAVE / UCF_PART; part++)
        if ((lane / UCF_PART) == part) {
            for (int k = 0; k < nterms; k++)
                scr_st(colA, k, lane, cmake(series[((size_t)ii * nterms + k) * 2], series[((size_t)ii * nterms + k) * 2 + 1]));
            r = wynn_lane(colA, colB, nterms, lane, &stt);
        }
    if (i < n) {
        acc[2 * i] = r.re;
        acc[2 * i + 1] = r.im;
        status[i] = stt;
    }
}

__global__ void __launch_bounds__(UCF_WAVE)
extrap_kernel(int n, int R, const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ out)
{
    extern __shared__ lds_c lds[];
    const int lane = threadIdx.x;
    const int i = blockIdx.x * UCF_WAVE + lane;
    lds_c* colC = lds;
    lds_c* colD = lds + (size_t)R * UCF_WAVE;
    const int ii = i < n ? i : n - 1;
    for (int k = 0; k < R; k++)
        lds_st(colC, k, lane, cmake(y[((size_t)ii * R + k) * 2], y[((size_t)ii * R + k) * 2 + 1]));
    cplx r = cmake(0.0, 0.0);
    for (int part = 0; part < UCF_WAVE / UCF_PART; part++)
        if ((lane / UCF_PART) == part) r = extrap_lane(colC, 1, colD, x, R, lane);
    if (i < n) {
        out[2 * i] = r.re;
        out[2 * i + 1] = r.im;
    }
}
#endif

// ------------------------------------------------------------------ launchers
#ifdef UCF_PROBE      /* tools/probe_kernel.sh: one kernel instantiation on its own (seconds instead of minutes), no launchers */
const void* ucf_probe_kernel_address = (const void*)&UCF_PROBE;
#else
static inline int family_of(const ucf_dev_params& dp)
{
    switch (dp.model) {
    case 0: return 0;
    case 1: return 1;
    case 2: return 5;
    case 3: case 4: case 5: return 2;
    case 6: return dp.MNtype == 1 ? 3 : (dp.MNtype == 2 ? 4 : -1);
    default: return -1;
    }
}

// finished interval areas stay in LDS while the footprint still admits 8 single-wave workgroups per CU
static inline bool areas_in_lds(const ucf_dev_params& dp)
{
    const size_t with_areas = ((size_t)(dp.R + 1 + dp.nacc) * dp.nz * UCF_WAVE + (size_t)(2 * dp.nacc > dp.R ? 2 * dp.nacc : dp.R) * UCF_PART) * sizeof(lds_c);
    return with_areas <= 20 * 1024;
}

static inline size_t point_lds_bytes(const ucf_dev_params& dp, bool resume = false)
{
    size_t bytes = ((size_t)(dp.R + 1 + ((areas_in_lds(dp) && !resume) ? dp.nacc : 0)) * dp.nz * UCF_WAVE + (size_t)(2 * dp.nacc > dp.R ? 2 * dp.nacc : dp.R) * UCF_PART) * sizeof(lds_c);
#if !UCF_FAST
    if (family_of(dp) == 4) bytes += 2 * (size_t)dp.order * UCF_WAVE * sizeof(lds_c);
#endif
    return bytes;
}

#if !UCF_FAST
// (time, radius) grid -> the point list it stands for, point = it * nr + ir (the grid's own output order)
__global__ void __launch_bounds__(256)
expand_grid_kernel(int nt, int nr, const double* __restrict__ tDv, const int* __restrict__ svv, const double* __restrict__ rDv,
                   double* __restrict__ tDp, double* __restrict__ rDp, int* __restrict__ svp)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long long)nt * nr) return;
    const int it = (int)(q / nr), ir = (int)(q % nr);
    tDp[q] = tDv[it];
    svp[q] = svv[it];
    rDp[q] = rDv[ir];
}
int launch_expand_grid(int nt, int nr, const double* d_tD, const int* d_sv, const double* d_rD, double* d_tDp, double* d_rDp,
                       int* d_svp, void* stream)
{
    const long long n = (long long)nt * nr;
    hipLaunchKernelGGL(expand_grid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nt, nr, d_tD, d_sv, d_rD,
                       d_tDp, d_rDp, d_svp);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}

int launch_abscissae(const ucf_dev_params& dp, int nrows, int per_point, int nsv, int svmin, const double* d_rD,
                     const int* d_sv, double* d_tab, void* stream)
{
    const long long total = (long long)nrows * (dp.N + dp.nacc * dp.ngl);
    const int threads = 256;
    const long long blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(abscissa_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, dp, nrows, per_point,
                       nsv, svmin, d_rD, d_sv, (double2*)d_tab);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}
#endif

// How the abscissa loop is run: 0 inside point_kernel; 1 integrate_kernel with the fast evaluators (fast flavour,
// Hantush-based models), point_kernel resumes the items it leaves unfinished; 2 integrate_generic_kernel with the
// reference-order evaluators (everything else, unless the finite-difference Thomas buffer makes the footprint huge)
static inline int split_kind(const ucf_dev_params& dp)
{
    const int fam = family_of(dp);
#if UCF_FAST
    if (fam >= 0 && fam <= 5) return 1;
#else
    if (fam == 4 && 2 * (size_t)dp.order * UCF_WAVE * sizeof(lds_c) > 16 * 1024) return 0;
#endif
    return 2;
}
#if UCF_TU_HAS(1)
// bytes of state per work item (0: no state needed)
size_t state_bytes_per_item(const ucf_dev_params& dp)
{
    return split_kind(dp) ? (size_t)(dp.R + 1 + dp.nacc) * dp.nz * UCF_WAVE * sizeof(lds_c) : 0;
}
#if UCF_FAST
// bytes of the lapTime table of a launch over `rows` rows of tD (behind the state in the same buffer; laptime_kernel)
size_t lt_table_bytes(const ucf_dev_params& dp, size_t rows)
{
    return split_kind(dp) == 1 ? rows * dp.np * sizeof(lds_c) : 0;
}
#endif
#endif

// The transform stage for `nwork` work items of lane layout LAYOUT: [integrate kernel -> finish_kernel ->] point_kernel.
// tm (optional): every kernel of the stage is bracketed by HIP events on the launch stream (ucf_timers).
template <int LAYOUT, bool MULTI>
static int launch_transform_(const ucf_dev_params& dp, int nwork, int per_point, int nr, int nsv, int svmin, const double* d_tD,
                            const double* d_rD, const int* d_sv, const double* d_tab, double* d_h, double* d_dh,
                            ucf_stats* d_stats, int nt, int ir0, int nrc, double* d_totlap, double* d_glscr,
                            double* d_state, int* d_ndone, void* stream, ucf_timers* tm,
                            const ucf_dev_params* d_params, int ppp, int pbase)
{
    char kname[96];
    int* d_todo = d_ndone ? d_ndone + nwork : nullptr;     // [count | items]: the caller sizes d_ndone for 2 nwork + 1 ints
    const int fam = family_of(dp);
    if (fam < 0) return UCF_ERR_UNSUPPORTED;
    const int kind = split_kind(dp);
    const bool split = kind != 0;
    if (split && (!d_state || !d_ndone)) return UCF_ERR_BAD_ARGUMENT;
    const size_t lds = point_lds_bytes(dp, split);
    if (lds > 160 * 1024) return UCF_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const bool al = areas_in_lds(dp) || split;
    if (al) d_glscr = nullptr;
    if (!split) { d_state = nullptr; d_ndone = nullptr; d_todo = nullptr; }
    dim3 grid((unsigned)((al || nwork < UCF_GRID_SLOTS) ? nwork : UCF_GRID_SLOTS)), block(UCF_WAVE);
    if (split) {
        (void)hipMemsetAsync(d_todo, 0, sizeof(int), s);
    }
#if UCF_FAST
    if (kind == 1) {
        // every item starts as "all abscissae done"; the parts that must stop lower it (integrate_kernel)
        (void)hipMemsetD32Async((hipDeviceptr_t)d_ndone, dp.N + dp.nacc * dp.ngl, (size_t)nwork, s);
        // parts per item: launches of fewer than ~8 rounds of resident waves (256 CUs x 4 SIMDs x <= 6 waves) run two parts
        // per item -- measured on the 1/8 shard of C2 (27 136 items, tools/gpu_shard.sh): 5.37 / 5.26 / 5.35 / 5.58 ms with
        // 1 / 2 / 4 / 8 parts (every part pays the item's set-up again).  UCF_NSPLIT (diagnostic): force 1, 2, 4 or 8 parts.
        static const int force_split = [] { const char* e = std::getenv("UCF_NSPLIT"); return e ? std::atoi(e) : 0; }();
        int lsplit = 0;
        while (lsplit < 1 && ((long long)nwork << lsplit) < 8LL * 256 * 4 * 6) lsplit++;
        if (force_split > 0) { lsplit = 0; while ((1 << (lsplit + 1)) <= force_split && lsplit < 3) lsplit++; }
        if ((1 << lsplit) > dp.nacc + 1) lsplit = 0;
        // ... and the last items of EVERY launch run in finer parts: ntail = one round of resident waves, 2^ltail parts each
        // (UCF_TAIL_ITEMS / UCF_TAIL_LSPLIT: diagnostic overrides; UCF_TAIL_LSPLIT=0 turns the finer tail off).  Measured
        // (tools/gpu_tail_parts.sh): C2 34.78 -> 34.64 ms, its 1/8 shard 4.835 -> 4.79 ms with 8 parts for the last 5 120
        // items (2 or 4 parts, or 10 240 items: the same within 0.2 %) -- a small gain: a wave on an emptying SIMD does
        // speed up enough to hide most of the quantisation of a launch into rounds.  Bit-neutral like every cut
        // (test_results_do_not_depend_on_how_work_items_are_cut).
        static const int tail_ls_env = [] { const char* e = std::getenv("UCF_TAIL_LSPLIT"); return e ? std::atoi(e) : -1; }();
        static const int tail_items_env = [] { const char* e = std::getenv("UCF_TAIL_ITEMS"); return e ? std::atoi(e) : -1; }();
        int ltail = tail_ls_env >= 0 ? tail_ls_env : UCF_TAIL_LSPLIT_DEFAULT;
        if (ltail > 3) ltail = 3;
        while (ltail > 0 && (1 << ltail) > dp.nacc + 1) ltail--;
        if (ltail < lsplit) ltail = lsplit;
        int ntail = tail_items_env >= 0 ? tail_items_env : 256 * 4 * 5;
        if (ntail > nwork) ntail = nwork;
        const int nhead = (ltail == lsplit) ? nwork : nwork - ntail;
        const long long nworkw = ((long long)nhead << lsplit) + ((long long)(nwork - nhead) << ltail);
        if (nworkw > 0x7fffffffLL) return UCF_ERR_UNSUPPORTED;
        // per workgroup: the sin/cos table + UCF_IWPB waves' accumulators; wlds = the footprint one wave accounts for
        const size_t wlds = (size_t)(dp.R + 1) * dp.nz * UCF_WAVE * sizeof(lds_c) + UCF_SC_ENTRIES * sizeof(lds_c) / UCF_IWPB;
        const size_t ilds = wlds * UCF_IWPB;
        // persistent grid: at most 8 workgroups per CU (more than any register / LDS budget admits; the ones that do not fit
        // start when others have finished and find the counter exhausted).  UCF_PERSIST=0 (diagnostic): one workgroup per
        // UCF_IWPB work units, as before round 3's last pass
        static const bool persist = [] { const char* e = std::getenv("UCF_PERSIST"); return !e || *e != '0'; }();
        const long long nwg = (nworkw + UCF_IWPB - 1) / UCF_IWPB;
        const dim3 igrid((unsigned)((persist && nwg > 256 * 8) ? 256 * 8 : nwg)), iblock(UCF_WAVE * UCF_IWPB);
        int* const d_wcount = d_ndone + 2 * (size_t)nwork + 2 + (size_t)nwork * dp.nz;      // behind the deferred list (finish_kernel)
        if (persist) (void)hipMemsetAsync(d_wcount, 0, sizeof(int), s);
        // lapTime(p) x constants for every (row of the call's tD, m): rows = the times of a grid / the points of a list.
        // The table lives behind the state of this launch's work items (the caller sized the buffer for it: lt_table_bytes)
        // (LAYOUT 1: the nt times; 3: the nt points of the launch; 0 / 2: the points, or -- a small grid walked point by point,
        //  per_point = 0 -- the time rows those points stand on: tD has no more entries than that)
        const int npts_l = (LAYOUT == 2) ? nwork / ((dp.np + UCF_WAVE - 1) / UCF_WAVE) : nwork;
        const int nrows = (LAYOUT == 1 || LAYOUT == 3) ? nt : (per_point ? npts_l : (npts_l + nr - 1) / nr);
        double* const d_ltab = d_state + (size_t)nwork * (dp.R + 1 + dp.nacc) * dp.nz * UCF_WAVE * 2;
        {
            const long long nlt = (long long)nrows * dp.np;
            const dim3 lgrid((unsigned)((nlt + 255) / 256)), lblock(256);
#define UCF_LAUNCH_LT(F) hipLaunchKernelGGL((laptime_kernel<F, MULTI>), lgrid, lblock, 0, s, dp, nrows, d_tD, (double2*)d_ltab, d_params, ppp, pbase)
            ucf_tm_mark(tm, UCF_STR(UCF_NS) "::laptime_kernel", s);
            switch (fam) {
            case 0: UCF_LAUNCH_LT(0); break;
            case 1: UCF_LAUNCH_LT(1); break;
            case 2: UCF_LAUNCH_LT(2); break;
            case 3: UCF_LAUNCH_LT(3); break;
            case 4: UCF_LAUNCH_LT(4); break;
            case 5: UCF_LAUNCH_LT(5); break;
            }
#undef UCF_LAUNCH_LT
        }
        static const bool nzc2_on = [] { const char* e = std::getenv("UCF_NZC2"); return !e || *e != '0'; }();      // diagnostic: 0 = off
#define UCF_LAUNCH_I4(F, W, FO, L3, NZC, L1, NF)                                                                     \
    do {                                                                                                       \
        const size_t ilds = ((size_t)((NZC) ? dp.R : dp.R + 1) * dp.nz * UCF_WAVE * sizeof(lds_c)) * UCF_IWPB + UCF_SC_ENTRIES * sizeof(lds_c); \
        if (ilds > 64 * 1024)                                                                                  \
            (void)hipFuncSetAttribute((const void*)integrate_kernel<F, LAYOUT, W, MULTI, FO, L3, NZC, L1, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ilds); \
        std::snprintf(kname, sizeof(kname), UCF_STR(UCF_NS) "::integrate_kernel<%d, %d, %d, %s, %s, %s, %d, %s, %s>", F, LAYOUT, W, MULTI ? "true" : "false", FO ? "true" : "false", L3 ? "true" : "false", NZC, L1 ? "true" : "false", NF ? "true" : "false"); \
        ucf_tm_mark(tm, kname, s);                                                                             \
        hipLaunchKernelGGL((integrate_kernel<F, LAYOUT, W, MULTI, FO, L3, NZC, L1, NF>), igrid, iblock, ilds, s, dp, nwork, per_point, nr, nsv, svmin, \
                           d_tD, d_rD, d_sv, (const double2*)d_tab, nt, ir0, (double2*)d_state, d_ndone, d_todo, d_params, ppp, pbase, lsplit | (ltail << 8), \
                           (const double2*)d_ltab, nrows, nhead << lsplit, (int)nworkw, persist ? d_wcount : (int*)nullptr);                                                     \
    } while (0)
    // launches of ONE depth of the fully penetrating water-table family in the lane = time layout (the headline sweep) run
    // an instantiation that knows nz = 1 at compile time: no depth loop, no running area in LDS (measured on C2: -2.4 %).
    // Only there: the unfolded and the finite-difference kernels LOSE 12 ... 46 % to it (C2pp 88 -> 100 ms, C4 237 -> 266,
    // C5 204 -> 297: the compiler hoists the depth's constants into registers those kernels do not have)
    // Launches of TWO depths in that layout (a screened observation well, C3; every pair of depths of a contour-style call,
    // which the host walks two at a time) run NZC = 2 in every family: the two running areas in registers, so that the wave's
    // LDS holds the level sums alone and a fourth workgroup fits the CU (C3 130.5 -> 115.2 ms per launch; 21-depth calls on
    // 128 x 64 points: +5 ... +27 %, Theis +52 %; tools/gpu_depths.sh).  UCF_NZC2=0 (diagnostic) turns it off.
#ifndef UCF_NZC
#define UCF_NZC(F, FO) (LAYOUT == 1 && (F) == 2 && (FO) ? 1 : 0)
#endif
#ifndef UCF_NZC2
#define UCF_NZC2(F, FO) ((LAYOUT == 1 || LAYOUT == 3) ? 2 : 0)
#endif
#define UCF_LAUNCH_I3(F, W, FO, L3, L1, NF)                                                                    \
    do {                                                                                                       \
        if (UCF_NZC(F, FO) && dp.nz == 1) UCF_LAUNCH_I4(F, W, FO, L3, UCF_NZC(F, FO), L1, NF);                 \
        else if (UCF_NZC2(F, FO) && dp.nz == 2 && nzc2_on) UCF_LAUNCH_I4(F, W, FO, L3, UCF_NZC2(F, FO), L1, NF); \
        else UCF_LAUNCH_I4(F, W, FO, L3, 0, L1, NF);                                                           \
    } while (0)
        // a depth above the screen top anywhere in the call (in any plan of a parameter batch)?
        const bool lay3 = dp.any_lay3 != 0;
        // ... below the screen bottom?  Three instantiations of an unfolded kernel: every layer / beside and below the screen /
        // beside the screen only (the usual piezometer or observation well)
        const bool lay1 = dp.any_lay1 != 0;
#define UCF_LAUNCH_FOLD(F, W) UCF_LAUNCH_I3(F, W, true, false, true, false)
        // neither screen term folds (d > 0 and l < b: the usual partially penetrating well) -- known at compile time in an
        // instantiation of its own (NOFOLD, ucf_fastpath.h); a plan that folds exactly one term, and a parameter batch with
        // such a plan or a fully penetrating one in it, run the general one.  UCF_NOFOLD=0 (diagnostic): always the general one
        static const bool nofold_on = [] { const char* e = std::getenv("UCF_NOFOLD"); return !e || *e != '0'; }();
        const bool nofold = nofold_on && !dp.any_fold;
#define UCF_LAUNCH_UNF_(F, W, NF)                                                                              \
    do {                                                                                                       \
        if (lay3) UCF_LAUNCH_I3(F, W, false, true, true, NF);                                                  \
        else if (lay1) UCF_LAUNCH_I3(F, W, false, false, true, NF);                                            \
        else UCF_LAUNCH_I3(F, W, false, false, false, NF);                                                     \
    } while (0)
#define UCF_LAUNCH_UNF(F, W)                                                                                   \
    do {                                                                                                       \
        if (nofold) { UCF_LAUNCH_UNF_(F, W, true); break; }                                                    \
        UCF_LAUNCH_UNF_(F, W, false);                                                                          \
    } while (0)
        // fully penetrating pumping well (every plan of a parameter batch must be): the screen terms are compiled out
        const bool fold = dp.fold_dD && dp.fold_lD1 && !MULTI;
        // two depths of the water-table family in the lane = time layout: running areas in registers (NZC = 2), level sums
        // alone in LDS -- at R = 4 a workgroup then needs 38 instead of 46 KB and FOUR of them fit a CU (measured on C3:
        // 130.5 -> 115.2 ms per launch with the 4-waves register budget; 127.3 ms with 3)
        const bool nzc2 = UCF_NZC2(2, false) != 0 && fam == 2 && nzc2_on && dp.nz == 2;
        const size_t wlds_eff = (nzc2 || (UCF_NZC(2, true) != 0 && fam == 2 && fold && dp.nz == 1))
                                    ? (size_t)dp.R * dp.nz * UCF_WAVE * sizeof(lds_c) + UCF_SC_ENTRIES * sizeof(lds_c) / UCF_IWPB : wlds;
        const bool w5 = wlds_eff * 20 <= 160 * 1024;
        // (parameter batches: the water-table and Hantush families only, ucf_drawdown_multi)
        if (MULTI && (fam == 0 || fam == 3 || fam == 5)) return UCF_ERR_UNSUPPORTED;
        switch (fam) {
        case 0: if constexpr (!MULTI) { if (wlds * 24 <= 160 * 1024) UCF_LAUNCH_FOLD(0, 6); else UCF_LAUNCH_FOLD(0, 4); } break;
        case 3: if constexpr (!MULTI) UCF_LAUNCH_FOLD(3, 4); break;       // (MNtype 1 is fully penetrating by construction, driver_io.f90:159-186)
        case 5: if constexpr (!MULTI) { if (fold) UCF_LAUNCH_FOLD(5, 4); else UCF_LAUNCH_UNF(5, 4); } break;
        case 1: if (fold) UCF_LAUNCH_FOLD(1, 4); else UCF_LAUNCH_UNF(1, 4); break;
        case 2:
            if (fold) {
                // register budget: 5 waves per SIMD (96 VGPRs, 8 of them spilled around the abscissa loop) where the LDS
                // footprint admits them.  Round 3, C2: 35.8 / 34.8 / 34.8 ms at 4 / 5 / 6 waves -- the sixth wave buys nothing
                // any more and costs 16 more spilled registers per item (2 GB of scratch traffic per sweep); the 1/8 shard
                // runs 5.29 against 5.33 ms.  UCF_FOLD_WAVES_RT (diagnostic): force 4, 5 or 6.
                static const int force_w = [] { const char* e = std::getenv("UCF_FOLD_WAVES_RT"); return e ? std::atoi(e) : 0; }();
                if (force_w == 4) UCF_LAUNCH_FOLD(2, 4);
                else if (force_w == 6 && wlds_eff * 24 <= 160 * 1024) UCF_LAUNCH_FOLD(2, 6);
                else if (w5) UCF_LAUNCH_FOLD(2, UCF_FOLD_WAVES);
                else UCF_LAUNCH_FOLD(2, 4);
            }
            // the screen terms need the registers: 4 waves/SIMD (128 VGPRs, ~60 spilled; 5 waves: -31 %); with two or more
            // depths per launch 3 waves/SIMD and no spills are 3 % faster (C3), with one depth 5 % slower (C2pp)
            else {
                static const int unf_w = [] { const char* e = std::getenv("UCF_UNFOLD_WAVES_RT"); return e ? std::atoi(e) : 0; }();   // diagnostic: 3 or 4
                if (unf_w == 3 || (unf_w != 4 && dp.nz >= 2 && !nzc2)) UCF_LAUNCH_UNF(2, 3);
                else UCF_LAUNCH_UNF(2, UCF_UNFOLD_WAVES);
            }
            break;
        case 4: if (fold) UCF_LAUNCH_FOLD(4, 4); else UCF_LAUNCH_UNF(4, 4); break;
        }
#undef UCF_LAUNCH_FOLD
#undef UCF_LAUNCH_UNF
#undef UCF_LAUNCH_UNF_
#undef UCF_LAUNCH_I3
#undef UCF_LAUNCH_I4
    }
#endif
    if (kind == 2) {
        if (MULTI) return UCF_ERR_UNSUPPORTED;
        size_t ilds = (size_t)(dp.R + 1) * dp.nz * UCF_WAVE * sizeof(lds_c);
#if !UCF_FAST
        if (fam == 4) ilds += 2 * (size_t)dp.order * UCF_WAVE * sizeof(lds_c);
#endif
#define UCF_LAUNCH_G(F)                                                                                        \
    do {                                                                                                       \
        if (ilds > 64 * 1024)                                                                                  \
            (void)hipFuncSetAttribute((const void*)integrate_generic_kernel<F, LAYOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ilds); \
        std::snprintf(kname, sizeof(kname), UCF_STR(UCF_NS) "::integrate_generic_kernel<%d, %d>", F, LAYOUT);  \
        ucf_tm_mark(tm, kname, s);                                                                             \
        hipLaunchKernelGGL((integrate_generic_kernel<F, LAYOUT>), dim3((unsigned)nwork), block, ilds, s, dp, nwork, per_point, nr, nsv, svmin, \
                           d_tD, d_rD, d_sv, (const double2*)d_tab, nt, ir0, (double2*)d_state, d_ndone);      \
    } while (0)
        switch (fam) {
#if !UCF_FAST                   /* the fast flavour has integrate_kernel for every family */
        case 0: UCF_LAUNCH_G(0); break;
        case 3: UCF_LAUNCH_G(3); break;
        case 5: UCF_LAUNCH_G(5); break;
        case 1: UCF_LAUNCH_G(1); break;
        case 2: UCF_LAUNCH_G(2); break;
        case 4: UCF_LAUNCH_G(4); break;
#endif
        }
#undef UCF_LAUNCH_G
    }
    if (split) {
        if (hipGetLastError() != hipSuccess) return UCF_ERR_HIP;
        // tails of the completed items.  nacc <= UCF_WYNN_REGS: epsilon table in registers, LDS only for the level sums
        // and the Neville column; else the widest scratch part that still leaves 4 waves per CU (measured on C2:
        // 4.8 / 3.6 / 3.1 ms for parts of 16 / 32 / 64 lanes)
        const bool wreg = dp.nacc <= UCF_WYNN_REGS && !ucf_finish_part;
        const size_t scols = wreg ? (size_t)dp.R : (size_t)(2 * dp.nacc > dp.R ? 2 * dp.nacc : dp.R);
        auto flds = [&](int part) { return ((size_t)dp.R * dp.nz * UCF_WAVE + scols * part) * sizeof(lds_c); };
        int part = ucf_finish_part;
        if (part != 16 && part != 32 && part != 64) part = (flds(64) <= 40 * 1024) ? 64 : (flds(32) <= 40 * 1024) ? 32 : 16;
        const size_t fl = flds(part);
        if (fl > 160 * 1024) return UCF_ERR_UNSUPPORTED;
#define UCF_LAUNCH_FM(PART, WR, MODE, GRID)                                                                     \
    do {                                                                                                       \
        if (fl > 64 * 1024)                                                                                    \
            (void)hipFuncSetAttribute((const void*)finish_kernel<LAYOUT, PART, WR, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl); \
        std::snprintf(kname, sizeof(kname), UCF_STR(UCF_NS) "::finish_kernel<%d, %d, %s, %d>", LAYOUT, PART, WR ? "true" : "false", MODE); \
        ucf_tm_mark(tm, kname, s);                                                                             \
        hipLaunchKernelGGL((finish_kernel<LAYOUT, PART, WR, MODE>), dim3((unsigned)(GRID)), block, fl, s, dp, nwork, per_point, nr, nsv, svmin, \
                           d_tD, d_rD, d_sv, d_h, d_dh, d_stats, nt, ir0, (double2*)d_totlap, (const double2*)d_state, \
                           (const int*)d_ndone, d_defer);                                                      \
    } while (0)
        // fast flavour, epsilon table in registers: the pass with the unguarded table over all items, then the guarded one over
        // what that pass listed (d_defer: [count | pt * nz + z ...] behind the two lists of integrate_kernel)
        int* const d_defer = d_ndone + 2 * (size_t)nwork + 1;
        const bool two_pass = UCF_FAST && wreg;
        if (two_pass) (void)hipMemsetAsync(d_defer, 0, sizeof(int), s);
        // (a grid-stride pass with 4 096 ... 65 536 workgroups instead of one per item: 0.92 ms on C2 either way -- the pass is
        //  bound by the 3.3 GB of state it reads, not by workgroup launches)
#define UCF_LAUNCH_F(PART, WR)                                                                                 \
    do {                                                                                                       \
        if constexpr (UCF_FAST && (WR)) {                                                                      \
            UCF_LAUNCH_FM(PART, WR, 1, nwork);                                                                 \
            UCF_LAUNCH_FM(PART, WR, 2, (nwork < 12288 ? nwork : 12288));      /* (4 rounds of resident waves; an empty list costs ~6 us) */ \
        } else UCF_LAUNCH_FM(PART, WR, 0, nwork);                                                              \
    } while (0)
        if (wreg) { if (part == 64) UCF_LAUNCH_F(64, true); else if (part == 32) UCF_LAUNCH_F(32, true); else UCF_LAUNCH_F(16, true); }
        else if (part == 64) UCF_LAUNCH_F(64, false);
        else if (part == 32) UCF_LAUNCH_F(32, false);
        else UCF_LAUNCH_F(16, false);
#undef UCF_LAUNCH_F
#undef UCF_LAUNCH_FM
        if (hipGetLastError() != hipSuccess) return UCF_ERR_HIP;
        if (kind == 2) { ucf_tm_close(tm, s); return UCF_OK; }      // the generic evaluators leave nothing unfinished
        // the unfinished ones (overflow regime): point_kernel over the list integrate_kernel left
        grid = dim3((unsigned)(nwork < 2048 ? nwork : 2048));
    }
#define UCF_LAUNCH(F)                                                                                          \
    do {                                                                                                       \
        if (lds > 64 * 1024)                                                                                   \
            (void)hipFuncSetAttribute((const void*)point_kernel<F, LAYOUT, MULTI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        std::snprintf(kname, sizeof(kname), UCF_STR(UCF_NS) "::point_kernel<%d, %d, %s>", F, LAYOUT, MULTI ? "true" : "false"); \
        ucf_tm_mark(tm, kname, s);                                                                             \
        hipLaunchKernelGGL((point_kernel<F, LAYOUT, MULTI>), grid, block, lds, s, dp, nwork, per_point, nr, nsv, svmin, d_tD, d_rD, d_sv, \
                           (const double2*)d_tab, d_h, d_dh, d_stats, nt, ir0, nrc, (double2*)d_totlap, (double2*)d_glscr,          \
                           (double2*)d_state, (const int*)d_ndone, (const int*)d_todo, d_params, ppp, pbase);  \
    } while (0)
    switch (fam) {
    case 0: UCF_LAUNCH(0); break;
    case 1: UCF_LAUNCH(1); break;
    case 2: UCF_LAUNCH(2); break;
    case 3: UCF_LAUNCH(3); break;
    case 4: UCF_LAUNCH(4); break;
    case 5: UCF_LAUNCH(5); break;
    }
#undef UCF_LAUNCH
    ucf_tm_close(tm, s);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}

// d_params != NULL: parameter-batched launch (per-point layouts of the fast flavour only), plan k owns points
// [k ppp, (k+1) ppp) and reads d_params[k]; dp is plan 0's block
template <int LAYOUT>
static int launch_transform(const ucf_dev_params& dp, int nwork, int per_point, int nr, int nsv, int svmin, const double* d_tD,
                            const double* d_rD, const int* d_sv, const double* d_tab, double* d_h, double* d_dh,
                            ucf_stats* d_stats, int nt, int ir0, int nrc, double* d_totlap, double* d_glscr,
                            double* d_state, int* d_ndone, void* stream, ucf_timers* tm,
                            const ucf_dev_params* d_params = nullptr, int ppp = 1, int pbase = 0)
{
#if UCF_FAST
    if (d_params) {
        // parameter batches run in the per-point layouts 0, 2, 3 (the lane = time translation unit instantiates none of it)
        if constexpr (LAYOUT == 1) return UCF_ERR_BAD_ARGUMENT;
        else {
            if (!per_point) return UCF_ERR_BAD_ARGUMENT;
            return launch_transform_<LAYOUT, true>(dp, nwork, per_point, nr, nsv, svmin, d_tD, d_rD, d_sv, d_tab, d_h, d_dh, d_stats, nt,
                                                   ir0, nrc, d_totlap, d_glscr, d_state, d_ndone, stream, tm, d_params, ppp, pbase);
        }
    }
#else
    if (d_params) return UCF_ERR_UNSUPPORTED;
#endif
    return launch_transform_<LAYOUT, false>(dp, nwork, per_point, nr, nsv, svmin, d_tD, d_rD, d_sv, d_tab, d_h, d_dh, d_stats, nt, ir0, nrc,
                                            d_totlap, d_glscr, d_state, d_ndone, stream, tm, nullptr, 1, 0);
}

#if UCF_TU_HAS(0)
// LAYOUT 0 (lane = Laplace sample, de Hoog in the same wave)
int launch_points(const ucf_dev_params& dp, int npts, int per_point, int nr, int nsv, int svmin, const double* d_tD,
                  const double* d_rD, const int* d_sv, const double* d_tab, double* d_h, double* d_dh,
                  ucf_stats* d_stats, void* stream, double* d_glscr, double* d_state, int* d_ndone,
                  const ucf_dev_params* d_params, int ppp, int pbase, double* d_dbg_totlap)
{
    return launch_transform<0>(dp, npts, per_point, nr, nsv, svmin, d_tD, d_rD, d_sv, d_tab, d_h, d_dh, d_stats, 0, 0, 0,
                               d_dbg_totlap, d_glscr, d_state, d_ndone, stream, nullptr, d_params, ppp, pbase);
}

#endif

#if UCF_TU_HAS(1)
// LAYOUT 1 (lane = time): transform kernel(s) over (radius chunk x time tiles x Laplace index), then de Hoog
int launch_grid_transposed(const ucf_dev_params& dp, int nt, int nr, int ir0, int nrc, int svmin, const double* d_tD,
                           const double* d_rD, const double* d_tab, double* d_totlap, double* d_h, double* d_dh,
                           ucf_stats* d_stats, void* stream, ucf_timers* tm, double* d_glscr, double* d_state,
                           int* d_ndone)
{
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (nt + UCF_WAVE - 1) / UCF_WAVE;
    const long long nwork = (long long)nrc * ntiles * dp.np;
    if (nwork > 0x7fffffffLL) return UCF_ERR_BAD_ARGUMENT;
    int rc = launch_transform<1>(dp, (int)nwork, 0, nr, 1, svmin, d_tD, d_rD, nullptr, d_tab, d_h, d_dh, d_stats, nt, ir0, nrc,
                                 d_totlap, d_glscr, d_state, d_ndone, stream, tm);
    if (rc) return rc;
    const long long ntl = (long long)nrc * ((nt + UCF_DH_TILE - 1) / UCF_DH_TILE);
    const size_t dlds = 2 * (size_t)dp.np * (UCF_DH_TILE + 1) * sizeof(lds_c) + 2 * UCF_DH_TILE * sizeof(int);
    // (one workgroup per tile: a capped grid walking the tiles with a stride is SLOWER -- C2 1.34 ms against 1.99 / 1.55 / 1.44 /
    //  1.37 ms with 2 048 / 4 096 / 8 192 / 16 384 workgroups, measured: a static stride cannot rebalance what the dispatcher does)
    const dim3 dgrid((unsigned)(ntl > 0x7fffffffLL ? 0x7fffffff : ntl));
    if (dp.np <= UCF_WAVE) {
        ucf_tm_mark(tm, UCF_STR(UCF_NS) "::dehoog_tiles_kernel<1, false>", s);
        hipLaunchKernelGGL((dehoog_tiles_kernel<1, false>), dgrid, dim3(UCF_WAVE), dlds, s, dp, nt, nr, ir0, nrc, d_tD, (const double2*)d_totlap, d_h, d_dh, d_stats);
    } else {
        ucf_tm_mark(tm, UCF_STR(UCF_NS) "::dehoog_tiles_kernel<1, true>", s);
        hipLaunchKernelGGL((dehoog_tiles_kernel<1, true>), dgrid, dim3(UCF_WAVE), dlds, s, dp, nt, nr, ir0, nrc, d_tD, (const double2*)d_totlap, d_h, d_dh, d_stats);
    }
    ucf_tm_close(tm, s);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}

#endif

#if UCF_TU_HAS(3)
// LAYOUT 3 (lane = point of an arbitrary list, 2M+1 <= 64): npts points, ppp of them per plan (npts for one plan);
// transform over (64-point tiles x Laplace index), then the tiled de Hoog with the points in the place of the times
int launch_points_lanes(const ucf_dev_params& dp, int npts, int ppp, const double* d_tD, const double* d_rD, const int* d_sv,
                        const double* d_tab, double* d_totlap, double* d_h, double* d_dh, ucf_stats* d_stats, void* stream,
                        double* d_state, int* d_ndone, const ucf_dev_params* d_params, int pbase)
{
    hipStream_t s = (hipStream_t)stream;
    if (ppp < 1 || npts % ppp != 0 || dp.np > UCF_WAVE) return UCF_ERR_BAD_ARGUMENT;
    const long long nwork = (long long)(npts / ppp) * ((ppp + UCF_WAVE - 1) / UCF_WAVE) * dp.np;
    if (nwork > 0x7fffffffLL) return UCF_ERR_BAD_ARGUMENT;
    int rc = launch_transform<3>(dp, (int)nwork, 1, ppp, 1, 0, d_tD, d_rD, d_sv, d_tab, d_h, d_dh, d_stats, npts, 0, 0, d_totlap, nullptr,
                                 d_state, d_ndone, stream, nullptr, d_params, ppp, pbase);
    if (rc) return rc;
    const long long ntl = (npts + UCF_DH_TILE - 1) / UCF_DH_TILE;
    const size_t dlds = 2 * (size_t)dp.np * (UCF_DH_TILE + 1) * sizeof(lds_c) + 2 * UCF_DH_TILE * sizeof(int);
    if (dp.np <= UCF_WAVE)
        hipLaunchKernelGGL((dehoog_tiles_kernel<3, false>), dim3((unsigned)ntl), dim3(UCF_WAVE), dlds, s, dp, npts, 1, 0, 1, d_tD,
                           (const double2*)d_totlap, d_h, d_dh, d_stats);
    else
        hipLaunchKernelGGL((dehoog_tiles_kernel<3, true>), dim3((unsigned)ntl), dim3(UCF_WAVE), dlds, s, dp, npts, 1, 0, 1, d_tD,
                           (const double2*)d_totlap, d_h, d_dh, d_stats);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}
#endif

#if UCF_TU_HAS(2)
// LAYOUT 2 (2M+1 > 64): (point, 64-sample chunk) work items write the transform, dehoog_points_kernel inverts.
// Same addressing as launch_points; d_h/d_dh/d_totlap point at this chunk of points.
int launch_points_chunked(const ucf_dev_params& dp, int npts, int per_point, int nr, int nsv, int svmin, const double* d_tD,
                          const double* d_rD, const int* d_sv, const double* d_tab, double* d_totlap, double* d_h,
                          double* d_dh, ucf_stats* d_stats, void* stream, double* d_glscr, double* d_state, int* d_ndone,
                          const ucf_dev_params* d_params, int ppp, int pbase)
{
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = (dp.np + UCF_WAVE - 1) / UCF_WAVE;
    const long long nwork = (long long)npts * nchunk;
    if (nwork > 0x7fffffffLL) return UCF_ERR_BAD_ARGUMENT;
    int rc = launch_transform<2>(dp, (int)nwork, per_point, nr, nsv, svmin, d_tD, d_rD, d_sv, d_tab, d_h, d_dh, d_stats, 0, 0, 0,
                                 d_totlap, d_glscr, d_state, d_ndone, stream, nullptr, d_params, ppp, pbase);
    if (rc) return rc;
    hipLaunchKernelGGL(dehoog_points_kernel, dim3((unsigned)npts), dim3(UCF_WAVE), 0, s, dp, (long long)npts, 1, per_point, nr, 0, 0, d_tD,
                       (const double2*)d_totlap, d_h, d_dh, d_stats);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}

#endif

#if UCF_TU_HAS(1)
int launch_samples(const ucf_dev_params& dp, int n_a, const double* d_a, double rD, const double* d_p, double* d_fp,
                   void* stream)
{
    const int fam = family_of(dp);
    if (fam < 0) return UCF_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    size_t lds = (fam == 4 && !UCF_FAST) ? 2 * (size_t)dp.order * UCF_WAVE * sizeof(lds_c) : 16;
    if (UCF_FAST) lds = UCF_SC_ENTRIES * sizeof(lds_c);
    dim3 grid(n_a), block(UCF_WAVE);
#define UCF_LAUNCH(F)                                                                                          \
    do {                                                                                                       \
        if (lds > 64 * 1024)                                                                                   \
            (void)hipFuncSetAttribute((const void*)samples_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(samples_kernel<F>, grid, block, lds, s, dp, n_a, d_a, rD, d_p, d_fp);               \
    } while (0)
    switch (fam) {
    case 0: UCF_LAUNCH(0); break;
    case 1: UCF_LAUNCH(1); break;
    case 2: UCF_LAUNCH(2); break;
    case 3: UCF_LAUNCH(3); break;
    case 4: UCF_LAUNCH(4); break;
    case 5: UCF_LAUNCH(5); break;
    }
#undef UCF_LAUNCH
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}

#endif

#if !UCF_FAST
int launch_bessel(int n, const double* d_z, double* d_k, int* d_ierr, void* stream)
{
    hipLaunchKernelGGL(bessel_kernel, dim3((n + UCF_WAVE - 1) / UCF_WAVE), dim3(UCF_WAVE), 0, (hipStream_t)stream, n, d_z, d_k, d_ierr);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}
int launch_dehoog(int n, int M, double alpha, double logtol, const double* d_t, const double* d_tee,
                  const double* d_fp, double* d_ft, void* stream)
{
    hipLaunchKernelGGL(dehoog_kernel, dim3(n), dim3(UCF_WAVE), 0, (hipStream_t)stream, n, M, alpha, logtol, d_t, d_tee,
                       d_fp, d_ft);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}
int launch_wynn(int n, int nterms, const double* d_series, double* d_acc, int* d_status, void* stream)
{
    const size_t lds = 2 * (size_t)nterms * UCF_PART * sizeof(lds_c);
    hipLaunchKernelGGL(wynn_kernel, dim3((n + UCF_WAVE - 1) / UCF_WAVE), dim3(UCF_WAVE), lds, (hipStream_t)stream, n,
                       nterms, d_series, d_acc, d_status);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}
int launch_extrap(int n, int R, const double* d_x, const double* d_y, double* d_out, void* stream)
{
    const size_t lds = ((size_t)R * UCF_WAVE + (size_t)R * UCF_PART) * sizeof(lds_c);
    hipLaunchKernelGGL(extrap_kernel, dim3((n + UCF_WAVE - 1) / UCF_WAVE), dim3(UCF_WAVE), lds, (hipStream_t)stream, n,
                       R, d_x, d_y, d_out);
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}
#endif
#endif   // UCF_PROBE

}  // namespace UCF_NS
