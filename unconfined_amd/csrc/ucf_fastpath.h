// ucf_fastpath.h -- "fast" flavour of the sample evaluators: families 1, 2, 4 (Hantush, Moench / Malama / Neuman
// water-table closure, Mishra-Neuman FD), 0 (Theis: the reciprocal alone), 5 (Hantush with wellbore storage: family 1
// times a per-lane constant) and 3 (Mishra-Neuman in Malama's closed form).
//
// Same formulas as laplace_hankel_solutions.f90:64-93,133-202 (cited per line below), evaluated
// with the minimum of work a wave needs:
//   * every cosh/sinh/exp of eta*c comes from ONE primitive per distinct real factor c:
//     E = exp(|Re|) and sincos(Im) from tables in LDS (exp_tab_, sincos_tab_: ucf_math.h), 1/E by a
//     Newton reciprocal; cosh, sinh and exp(-.) of that argument are 2-4 multiplies away;
//   * factors that differ by a rounding error share a primitive (cosh(eta (dD1 - 1)) next to sinh(eta dD):
//     first-order correction in the exactly known difference, fast_common_terms);
//   * divisions by per-lane constants (p, kappa, bD, Moench sum) are reciprocals hoisted out of
//     the abscissa loop; sinh(eta) and the closure denominator are inverted once per sample;
//   * exact zeros are folded at plan level: a fully penetrating well (d = 0, l = b) has
//     sinh(eta*dD) = sinh(eta*lD1) = 0 and cosh(eta*(dD1-1)) = 1 identically.
// The fast evaluation is only used while Re(eta)*cmax <= 700, i.e. while no cosh/sinh or product
// of two of them can overflow; beyond that the generic evaluator (which reproduces the
// reference's Inf/NaN behaviour, needed by the in-band rules) takes over for the whole wave.
#pragma once

namespace UCF_NS {

// Marks a block that depends on a WAVE-UNIFORM condition: the block stays a branch.  Without it the compiler turns
// "uniform ? a : b" into selects on the vector ALU -- one issue slot per dword for a decision the scalar unit has already taken
// (16 of the 382 VALU instructions per abscissa of the partially penetrating water-table kernel were such selects).
#define UCF_UNIFORM_BLOCK() asm volatile("" ::: "memory")
// ... except in the finite-difference kernel, whose node recurrence has no register to spare for the longer live ranges
// (0 -> 4 spilled VGPRs, +0.7 % on C5 with the marks)
#define UCF_UNIFORM_BLOCK_F(FAMILY) do { if constexpr ((FAMILY) != 4) UCF_UNIFORM_BLOCK(); } while (0)

struct fprim {
    double ch, sh, ei, sn, cs;    // cosh(x), sinh(x), exp(-|x|), sin(y), cos(y)
};

// POS: the caller knows x >= 0 (Re eta > 0 times a non-negative factor): no sign to put back on sinh
template <bool POS = false>
UCF_DEV fprim prim(double x, double y, const sc_ctx& sc)
{
    const int salt = sc.salt;
    fprim f;
    const double ax = POS ? x : fabs(x);          // (POS: the absolute value would be two VALU instructions in front of the inline asm)
    const double e = exp_tab_(ax, sc);
    const double ei = fast_rcp(e);
    f.ei = ei;
    const double he = 0.5 * e;
    f.ch = __builtin_fma(0.5, ei, he);
    double s;
    if (ax < 0.35) {
        const double x2 = ax * ax;
        double pl = addk(mulk(x2, UCF_KHERE(1.0 / 6227020800.0, salt)), UCF_KHERE(1.0 / 39916800.0, salt));
        pl = fmak(pl, x2, UCF_KHERE(1.0 / 362880.0, salt));
        pl = fmak(pl, x2, UCF_KHERE(1.0 / 5040.0, salt));
        pl = fmak(pl, x2, UCF_KHERE(1.0 / 120.0, salt));
        pl = fmak(pl, x2, UCF_KHERE(1.0 / 6.0, salt));
        s = __builtin_fma(ax * x2, pl, ax);
    } else {
        s = __builtin_fma(-0.5, ei, he);
    }
    f.sh = POS ? s : copysign(s, x);
    sincos_tab_(y, sc, &f.sn, &f.cs);     // |y| < 1e6: fast_eta() vouches for it
    return f;
}
// primitive of x1 - x2, y1 - y2 from the primitives of (x1, y1) and (x2, y2), 0 <= x2 <= x1: the real
// exponentials divide, the angles subtract (no cancellation in either; sinh of a small difference loses
// relative, not absolute, accuracy, which is all cosh/sinh products of complex arguments need)
UCF_DEV fprim prim_difference(const fprim& a, const fprim& b)
{
    fprim f;
    const double e = (a.ch + a.sh) * b.ei;            // e^(x1 - x2)
    const double ei = a.ei * (b.ch + b.sh);
    f.ei = ei;
    f.ch = 0.5 * (e + ei);
    f.sh = 0.5 * (e - ei);
    f.sn = __builtin_fma(a.sn, b.cs, -(a.cs * b.sn));
    f.cs = __builtin_fma(a.cs, b.cs, a.sn * b.sn);
    return f;
}
UCF_DEV cplx pcosh(const fprim& f) { return cmake(f.ch * f.cs, f.sh * f.sn); }
UCF_DEV cplx psinh(const fprim& f) { return cmake(f.sh * f.cs, f.ch * f.sn); }
// exp(-(x+iy)) for x >= 0
UCF_DEV cplx pexpneg(const fprim& f) { return cmake(f.ei * f.cs, -(f.ei * f.sn)); }

// exp(-(x + iy)) for x >= 0 on its own: no reciprocal, no cosh/sinh
UCF_DEV cplx expneg_direct(double x, double y, const sc_ctx& sc)
{
    const double ei = exp_tab_(-x, sc);
    double sn, cs;
    sincos_tab_(y, sc, &sn, &cs);
    return cmake(ei * cs, -(ei * sn));
}

// a b + c in four FMAs (one rounding less per component than product, then sum)
UCF_DEV cplx cfma(cplx a, cplx b, cplx c)
{
    return cmake(__builtin_fma(a.re, b.re, __builtin_fma(-a.im, b.im, c.re)), __builtin_fma(a.re, b.im, __builtin_fma(a.im, b.re, c.im)));
}
// c - a b
UCF_DEV cplx cfnma(cplx a, cplx b, cplx c)
{
    return cmake(__builtin_fma(-a.re, b.re, __builtin_fma(a.im, b.im, c.re)), __builtin_fma(-a.re, b.im, __builtin_fma(-a.im, b.re, c.im)));
}
// 1/z without scaling: |z| in [1e-150, 1e150]
UCF_DEV cplx cinv_plain(cplx z)
{
    const double r = fast_rcp(z.re * z.re + z.im * z.im);
    return cmake(z.re * r, -(z.im * r));
}
// 1/z with the exponent scaling of __divdc3 (|z| may be up to e^700)
UCF_DEV cplx cinv_scaled(cplx z)
{
    const double m = fmax(fabs(z.re), fabs(z.im));
    const int il = __builtin_amdgcn_frexp_exp(m) - 1;
    const double c = ldexp(z.re, -il), d = ldexp(z.im, -il);
    const double r = fast_rcp(c * c + d * d);
    return cmake(ldexp(c * r, -il), ldexp(-(d * r), -il));
}

// 1/z: unscaled when every lane of the wave is far from overflow/underflow of |z|^2 (same bits either way:
// the scaling is by powers of two)
UCF_DEV cplx cinv_auto(cplx z)
{
    const double m = fmax(fabs(z.re), fabs(z.im));
    if (__builtin_amdgcn_ballot_w64(!(m < 1.0e150 && m > 1.0e-150)) == 0) return cinv_plain(z);
    return cinv_scaled(z);
}

struct lane_consts {     // per lane, constant over the abscissa loop of one point
    cplx p, lt, xifac;   // xifac = alphaD/p [* MoenchM / sum_m 1/(1+p/gamma_m)]   (:70,72-75)
    cplx fdB1;           // FD: p*beta0*exp(-beta2)/kappa                          (:492)
    cplx mnpv;           // MN-Malama: 1 + p vartheta / (kappa u0^2): (eta1/u0)^2 + 1 without its a^2 term   (:428-432)
};

UCF_DEV lane_consts make_lane_consts(const ucf_dev_params& P, cplx p, cplx lt)
{
    lane_consts L;
    L.p = p;
    L.lt = lt;
    cplx xf = rscale(P.alphaD, cinv_plain(p));
    if (P.model == 3) {
        cplx sum = cmake(0.0, 0.0);
        for (int j = 0; j < P.MoenchM; j++) sum = cadd(sum, cinv_plain(radd(1.0, cscale(p, P.MoenchInvGamma[j]))));
        xf = cmul(cscale(xf, (double)P.MoenchM), cinv_plain(sum));
    }
    L.xifac = xf;
    L.fdB1 = cscale(cscale(cscale(p, P.fd_beta0), P.fd_expmb2), P.inv_kappa);
    L.mnpv = radd(1.0, cscale(p, P.mn_vartheta * P.mn_c3));
    return L;
}

struct fast_common {
    sc_ctx sc;           // sin/cos table in LDS + constants (sincos_tab_), set once per kernel; sc.salt = the loop counter
    cplx th, eta, ff1, ff2, inv_she, she, che, top, g3, inv_den, ex1;   // ex1 = exp(-eta)
    fprim p1;            // primitive of eta itself (valid when have_p1)
    int have_p1;         // (wave-uniform flags are ints: a uniform bool that crosses a join is rebuilt through VALU selects)
    cplx mn_uod;         // MN-Malama: u / Delta_0                                                     (:437-439)
#ifdef UCF_SINGLE_RCP    /* investigation build only (tools/dbg_single_rcp.py): NOT the product */
    cplx q, den;
#endif
    cplx fd_s1;          // FD: sigma(1)
    cplx top3, fd_s13;   // the same two for depths above the screen top (cancellation-free water-table value)
    bool small_eta, fd_use, fd_use3;
    int any_small, any_large;    // wave-uniform: some lane on the cosh/sinh form, some lane on the exponential form
};

// Every fast sample is linear in theis(a, p) = 2/(p + a^2) and, for the Hantush-based models, carries the factor
// 1/bD (:200): both constants are left out of the evaluators and applied once to whatever the samples were summed
// into (integrate_kernel: with lapTime(p), when the sums leave the kernel).  fast_sample_z() * fast_scale() is the
// reference's sample.
template <int FAMILY>
UCF_DEV double fast_scale(const ucf_dev_params& P)
{
    // (family 0 and 3: theis and 2 / (kappa eta^2) = 2 / q; family 5: uDf / bD carries no 2, its A0 / (p tDb + 1) is the
    //  caller's lane constant)
    if (FAMILY == 0 || FAMILY == 3) return 2.0;
    if (FAMILY == 5) return P.inv_bD;
    return (FAMILY == 2 && P.model == 4) ? 2.0 : 2.0 * P.inv_bD;
}

// theta and eta of this abscissa.  Returns false (for this lane) if the fast evaluation is not applicable:
// a cosh/sinh could overflow, or the argument of a sin/cos (|Im eta| times a factor <= 1) leaves the range of
// the two-stage Cody-Waite reduction.
template <int FAMILY>
UCF_DEV bool fast_eta(const ucf_dev_params& P, const lane_consts& L, double a, fast_common& S)
{
    const double a2 = a * a;
    const cplx q = caddr(L.p, a2);
#ifdef UCF_SINGLE_RCP
    S.q = q;
#endif
    {
        const double r = fast_rcp(q.re * q.re + q.im * q.im);
        S.th = cmake(q.re * r, -(q.im * r));                      // HALF of theis = 2/q (:122-131): see fast_scale()
    }
    if (FAMILY == 0) return q.re > 0.0;                           // Theis is that reciprocal and nothing else
    {   // eta = sqrt(q/kappa), Re q > 0                                                         (:69,172)
        const double qr = q.re * P.inv_kappa, qi = q.im * P.inv_kappa;
        double r, hr;
        const double d = sqrt_only(__builtin_fma(qr, qr, qi * qi));
        sqrt_hrsqrt(0.5 * (d + qr), &r, &hr);
        S.eta = cmake(r, qi * hr);
    }
    // (three compares and two scalar ANDs; the short-circuit form rebuilt the flag through a select)
    return (S.eta.re <= P.fast_eta_max) & (q.re > 0.0) & (fabs(S.eta.im) < P.fast_im_max);
}
// the same for a whole wave: true if EVERY lane may take the fast evaluation.  One ballot per compare and the scalar unit's OR
// (the ballot of the combined per-lane flag came back as flag -> 0 / 1 in a VGPR -> compare again: two VALU instructions per
// abscissa for nothing)
template <int FAMILY>
UCF_DEV bool fast_eta_wave(const ucf_dev_params& P, const lane_consts& L, double a, fast_common& S)
{
    (void)fast_eta<FAMILY>(P, L, a, S);
    const cplx q = caddr(L.p, a * a);
    unsigned long long bad = __builtin_amdgcn_ballot_w64(!(q.re > 0.0));
    if (FAMILY != 0) {
        bad |= __builtin_amdgcn_ballot_w64(!(S.eta.re <= P.fast_eta_max));
        bad |= __builtin_amdgcn_ballot_w64(!(fabs(S.eta.im) < P.fast_im_max));
    }
    return bad == 0;
}

// z-independent part (after fast_eta said yes for every lane of the wave)
// FOLD: the plan is known to be fully penetrating (fold_dD and fold_lD1), so that none of the screen terms is even
// compiled in (the launcher picks the instantiation; FOLD = false handles every plan).  NOFOLD: the plan is known to fold
// NEITHER screen term (d > 0 and l < b, the usual partially penetrating well): the run-time tests of the two flags, the
// values a folded term would have taken and the copies where the two cases join are not compiled in (round 3, last pass:
// C2pp 81.8 -> 77.3 ms, C3 106.7 -> 99.9, C4 219.8 -> 208.6, Hantush with storage 66.8 -> 60.8; a plan that folds exactly
// one of the two terms, and a parameter batch with such a plan in it, runs the general instantiation).  A NOFOLD launch also knows that its
// plans carry the Hantush factor (no model 4 -- Neuman 1972 has no screen -- among them: dp.any_fold covers both): the run-time
// choice between "Hantush factor" and "Theis factor" and the copies where the two join are not compiled in either
// (C2pp 77.2 -> 75.3 ms, C3 100.6 -> 96.2, C4 209.0 -> 203.7)
// LAY3 = false: the launcher knows that no depth of the call lies above the screen top; LAY1 = false: none below its bottom
// (the usual piezometer beside the screen: the terms of the layer below -- exp(-eta), g3 -- are then not even allocated:
//  18 -> 14 spilled VGPRs in the partially penetrating water-table kernel, C2pp 87.9 -> 85.7 ms, C4 236.2 -> 230.3)
template <int FAMILY, bool FOLD = false, bool LAY3 = true, bool LAY1 = true, bool NOFOLD = false>
UCF_DEV void fast_common_terms(const ucf_dev_params& P, const lane_consts& L, double a, bool need_lay1_in, fast_common& S,
                               bool need_lay3_in = false, bool need_lay12 = true)
{
    const double a2 = a * a;
    if (FAMILY == 0) return;
    if (FAMILY == 3) {
        // Mishra-Neuman, Malama's closed form (:404-442): v = sqrt(1 + (eta1 / u0)^2), eta1^2 = (p vartheta + a^2) / kappa;
        // u = u0 (1 - v); Delta0 = eta sinh(eta) - u cosh(eta); the sample is (2 / q) (1 + (u / Delta0) cosh(eta zD))
        const fprim p1 = prim<true>(S.eta.re, S.eta.im, S.sc);
        S.che = pcosh(p1);
        S.she = psinh(p1);
        const double wr = __builtin_fma(a2, P.mn_c3, L.mnpv.re), wi = L.mnpv.im;          // 1 + (eta1 / u0)^2, Re > 0
        double vr, hr;
        sqrt_hrsqrt(0.5 * (sqrt_only(__builtin_fma(wr, wr, wi * wi)) + wr), &vr, &hr);
        const cplx u = cmake(P.mn_u0 * (1.0 - vr), -(P.mn_u0 * (wi * hr)));
        const cplx Delta0 = cfnma(u, S.che, cmul(S.eta, S.she));
        S.mn_uod = cmul(u, cinv_auto(Delta0));
        return;
    }
    const bool hantush = NOFOLD || !(FAMILY == 2 && P.model == 4);      // (a NOFOLD launch holds no model-4 plan: launch_transform_)
    const bool z1 = FOLD || (!NOFOLD && P.fold_dD != 0), z2 = FOLD || (!NOFOLD && P.fold_lD1 != 0);
    const bool need_lay1 = (FOLD || !LAY1) ? false : need_lay1_in;   // a fully penetrating screen has no layer below it
    const bool need_lay3 = (FOLD || !LAY3) ? false : need_lay3_in;   // ... nor above it
    // the water-table closure switches to its exponential form at Re(eta) >= maxexp (:84) and then needs neither
    // cosh(eta) nor sinh(eta): decide per wave what has to be evaluated at all
    S.small_eta = (FAMILY != 2) || (S.eta.re < P.maxexp);                                       // :84
    S.any_small = ((FAMILY != 2) || (__builtin_amdgcn_ballot_w64(S.small_eta) != 0)) ? 1 : 0;
    S.any_large = ((FAMILY == 2) && (__builtin_amdgcn_ballot_w64(!S.small_eta) != 0)) ? 1 : 0;
    const bool need_p1 = (hantush && (!(z1 && z2) || need_lay1)) || FAMILY == 4 || (FAMILY == 2 && S.any_small);
    S.have_p1 = (need_p1 && !z2) ? 1 : 0;      // kept only where fast_hantush_z derives the primitive of eta (1 - zD) from it
    fprim p1;
    if (need_p1) {
        p1 = prim<true>(S.eta.re, S.eta.im, S.sc);
        if (!z2) S.p1 = p1;
        S.che = pcosh(p1);
        S.she = psinh(p1);
        if (need_lay1) S.ex1 = pexpneg(p1);                    // (its only use: g3, below)
        else S.ex1 = cmake(0.0, 0.0);
    } else {
        S.che = S.she = S.ex1 = cmake(0.0, 0.0);     // (read by lanes that do not use them: must be benign numbers)
    }
    if (hantush) {
        fprim pd;
        if (!z1) { pd = prim<true>(S.eta.re * P.dD, S.eta.im * P.dD, S.sc); S.ff1 = psinh(pd); }           // :176
        else S.ff1 = cmake(0.0, 0.0);
        fprim pl;
        if (!z2) { pl = prim<true>(S.eta.re * P.lD1, S.eta.im * P.lD1, S.sc); S.ff2 = psinh(pl); }         // :177
        else S.ff2 = cmake(0.0, 0.0);
        if (!(z1 && z2) || need_lay1) S.inv_she = cinv_auto(S.she);
        if (need_lay1) {                                                                        // :183-184
            const cplx exl = z2 ? cmake(1.0, 0.0) : pexpneg(pl);
            S.g3 = csub(exl, cmul(cadd(S.ff1, cmul(S.ex1, S.ff2)), S.inv_she));
        }
        if (FAMILY != 1) {
            // water-table value: hantush at zD = 1 (layer 3): g1 - g2                          (:81,162-170,196)
            S.top = S.top3 = S.th;
            if (!(z1 && z2)) {
                if (need_lay12) {
                    cplx g1;
                    if (z1) g1 = cmake(1.0, 0.0);
                    else if (P.share_g1top == 1) g1 = pcosh(pd);
                    else if (P.share_g1top == 2) g1 = csub(pcosh(pd), cmul(rscale(P.g1_delta, S.eta), S.ff1));
                    else { const double c = P.dD1 - 1.0; g1 = pcosh(prim(S.eta.re * c, S.eta.im * c, S.sc)); }
                    const cplx udp = csub(g1, cmul(cadd(cmul(S.ff1, S.che), S.ff2), S.inv_she));
                    S.top = cmul(udp, S.th);                                                    // :200 (x fast_scale)
                }
                if (need_lay3) {
                    // For a depth above the screen top the rounding noise of that form matters (two terms of size
                    // e^{eta dD} leave one of size e^{-eta dD}, and the depth's own factor e^{eta (zD - 1)} no
                    // longer damps it).  Equivalent without the cancellation, used for those depths only:
                    //   g1 - g2 = cosh(eta dD) - [sinh(eta dD) cosh(eta) + sinh(eta lD1)] / sinh(eta)
                    //           = [sinh(eta (1 - dD)) - sinh(eta lD1)] / sinh(eta)
                    cplx sd1 = S.she;
                    if (!z1) {
                        // sinh(eta (1 - dD)) from the primitives of eta and eta dD (exponentials divide, angles
                        // subtract): accurate in the absolute sense only, good for a sinh of modulus >~ 1; the few
                        // abscissae with a small eta (1 - dD) get their own primitive
                        const double x1 = S.eta.re * P.dD1;
                        if (__builtin_amdgcn_ballot_w64(!(x1 >= 1.0)) == 0) sd1 = psinh(prim_difference(p1, pd));
                        else sd1 = psinh(prim(x1, S.eta.im * P.dD1, S.sc));
                    }
                    const cplx udp = cmul(z2 ? sd1 : csub(sd1, S.ff2), S.inv_she);
                    S.top3 = cmul(udp, S.th);
                }
            }
        }
    } else {
        S.top = S.top3 = S.th;                                                                  // :78-79
    }
    if (FAMILY == 2) {
        // |den| <= e^{maxexp} (1 + |xi|)(1 + beta |eta|) on the cosh/sinh form and <= that without the exponential on the
        // other: with |xi| < 1e100 (make_lane_consts vouches for it, else the item is not taken) the reciprocal needs no
        // exponent scaling
        const cplx xi = cmul(S.eta, L.xifac);                                                   // :70-75
        if (P.beta != 0.0) {
            const cplx one_bex = radd(1.0, cmul(rscale(P.beta, S.eta), xi));
            if (S.small_eta) S.inv_den = cinv_plain(cfma(one_bex, S.che, cmul(xi, S.she)));     // :86-87
            else S.inv_den = cinv_plain(cadd(one_bex, xi));                                     // :90-91
        } else {                                                 // beta = 0 (wave-uniform): no product with (1, 0)
            if (S.small_eta) S.inv_den = cinv_plain(cfma(xi, S.she, S.che));
            else S.inv_den = cinv_plain(radd(1.0, xi));
        }
#ifdef UCF_SINGLE_RCP
        // the rejected form: where the Hantush factor and the water-table value are both 1/q the sample
        // (1/q)(1 - f_z / den) is taken as (den - f_z) / (q den) -- one reciprocal, no 1/q of its own
        if (FOLD || !hantush) {
            cplx den;
            if (P.beta != 0.0) {
                const cplx one_bex = radd(1.0, cmul(rscale(P.beta, S.eta), xi));
                den = S.small_eta ? cfma(one_bex, S.che, cmul(xi, S.she)) : cadd(one_bex, xi);
            } else {
                den = S.small_eta ? cfma(xi, S.she, S.che) : radd(1.0, xi);
            }
            S.den = den;
            S.inv_den = cinv_auto(cmul(S.q, den));
        }
#endif
    }
    if (FAMILY == 4) {
        // Mishra/Neuman finite-difference vadose zone (:444-544): sigma(1) of the tridiagonal system by
        // elimination from the bottom up (continued fraction; v(3:n) = 0 and only x(1) is used, :521-523)
        const int n = P.order;
        const double h = P.fd_h, invhsq = P.fd_invhsq, b3h = P.fd_beta3 / h;
        const double B2 = a2 * P.inv_kappa;                                                     // :493
        const double csup = invhsq - b3h, K = csup * invhsq, bmid = b3h - 2.0 * invhsq - B2;
        const cplx B1 = L.fdB1;
        const cplx cc = cmake((b3h - invhsq - B2) - B1.re * P.fd_e[0], -(B1.im * P.fd_e[0]));   // :495,498
        const cplx eoh = cscale(S.eta, 1.0 / h);
        const cplx b1 = csub(cmul(cc, S.che), cmul(eoh, S.she));                                // :499-500
        (void)n;
        const cplx iB2 = fd_inverse_B2(P, B1, bmid, invhsq, b3h, K);                             // :501-502 (ucf_device.h)
        const cplx a2v = cscale(S.che, invhsq);                                                 // :508-509
        const cplx B1p = csub(b1, cmul(rscale(csup, a2v), iB2));
        const cplx iB1p = cinv_scaled(B1p);
        const cplx v1 = cneg(cmul(cc, S.top)), v2 = rscale(-invhsq, S.top);                     // :513-514
        const cplx v1p = csub(v1, cmul(rscale(csup, iB2), v2));
        S.fd_s1 = cmul(v1p, iB1p);
        S.fd_use = (fabs(S.fd_s1.re) + fabs(S.fd_s1.im)) > 2.3e-308;                            // :521
        S.fd_s13 = S.fd_s1;
        S.fd_use3 = S.fd_use;
        if (need_lay3) {
            const cplx w1 = cneg(cmul(cc, S.top3)), w2 = rscale(-invhsq, S.top3);
            S.fd_s13 = cmul(csub(w1, cmul(rscale(csup, iB2), w2)), iB1p);
            S.fd_use3 = (fabs(S.fd_s13.re) + fabs(S.fd_s13.im)) > 2.3e-308;
        }
    }
}

template <int FAMILY>
UCF_DEV bool fast_prepare(const ucf_dev_params& P, const lane_consts& L, double a, bool need_lay1, fast_common& S)
{
    if (!fast_eta<FAMILY>(P, L, a, S)) return false;
    bool need_lay3 = false, need_lay12 = false;
    for (int z = 0; z < P.nz; z++) { need_lay3 |= (P.zLay[z] == 3); need_lay12 |= (P.zLay[z] != 3); }
    fast_common_terms<FAMILY>(P, L, a, need_lay1, S, need_lay3, need_lay12);
    return true;
}

// Hantush factor at depth zD (:133-202); chz = cosh(eta*zD) is returned for the closure
template <int FAMILY, bool FOLD = false, bool LAY3 = true, bool LAY1 = true, bool NOFOLD = false>
UCF_DEV cplx fast_hantush_z(const ucf_dev_params& P, const fast_common& S, double zD, int lay_in, cplx* chz_out,
                            cplx* exz_out)
{
    const bool z1 = FOLD || (!NOFOLD && P.fold_dD != 0), z2 = FOLD || (!NOFOLD && P.fold_lD1 != 0);
    const int lay = (FOLD || (!LAY1 && lay_in == 1)) ? 2 : lay_in;      // (FOLD: everything is beside the screen)
    const bool need_chz = (lay == 1) || !z1 || (FAMILY == 2 && S.any_small) || FAMILY == 4;
    cplx chz = cmake(1.0, 0.0);
    fprim pz;
    if (need_chz) { pz = prim<true>(S.eta.re * zD, S.eta.im * zD, S.sc); chz = pcosh(pz); }      // (0 <= zD <= 1 on this path)
    *chz_out = chz;
    cplx udp;
    if (lay == 1) {
        udp = cmul(S.g3, chz);                                                                  // :188
    } else {
        cplx g2 = cmake(0.0, 0.0);
        const bool need_1z = !z2;
        fprim p1z;
        if (z2 && FAMILY == 2 && S.any_large) {                  // only exp(eta (zD - 1)) is wanted
            const double c = 1.0 - zD;
            *exz_out = expneg_direct(S.eta.re * c, S.eta.im * c, S.sc);
            UCF_UNIFORM_BLOCK_F(FAMILY);
        }
        if (need_1z) {
            if (need_chz && S.have_p1 && zD >= 0.0 && zD <= 1.0) p1z = prim_difference(S.p1, pz);     // eta (1 - zD)
            else
            { const double c = 1.0 - zD; p1z = prim(S.eta.re * c, S.eta.im * c, S.sc); }
        }
        if (!z2 && FAMILY == 2 && S.any_large) { *exz_out = pexpneg(p1z); UCF_UNIFORM_BLOCK_F(FAMILY); } // exp(eta*(zD-1))
        cplx f2c = cmake(0.0, 0.0);
        if (!z2) f2c = cmul(S.ff2, pcosh(p1z));                  // sinh(eta lD1) cosh(eta (1 - zD))
        if (lay == 2 || !LAY3) {
            if (z1 && z2) { UCF_UNIFORM_BLOCK_F(FAMILY); return S.th; }                                 // g2 = 0: udp = 1
            cplx num = f2c;
            if (!z1) { num = cadd(cmul(S.ff1, chz), f2c); UCF_UNIFORM_BLOCK_F(FAMILY); }
            g2 = cmul(num, S.inv_she);                                                          // :179-180
            udp = rsub(1.0, g2);                                                                // :192
            UCF_UNIFORM_BLOCK_F(FAMILY);
        } else {
            // above the screen top: g1 - g2 = cosh(eta (dD1 - zD)) - g2 (:175,196) subtracts two terms of size
            // e^{eta c}/2, c = zD - dD1 >= 0, that agree to e^{-2 eta c}.  With cosh A sinh B = [sinh(A+B) - sinh(A-B)]/2
            // the large parts cancel analytically:
            //   (g1 - g2) sinh(eta) = sinh(eta (2 - zD - dD))/2 + sinh(eta (zD - dD))/2 - sinh(eta lD1) cosh(eta (1 - zD))
            const double ca = 2.0 - zD - P.dD, cb = zD - P.dD;
            const cplx sa = psinh(prim(S.eta.re * ca, S.eta.im * ca, S.sc));
            const cplx sb = psinh(prim(S.eta.re * cb, S.eta.im * cb, S.sc));
            // (explicit FMAs: the same bits in every instantiation of the kernel)
            const cplx num = cmake(__builtin_fma(0.5, sa.re + sb.re, -f2c.re), __builtin_fma(0.5, sa.im + sb.im, -f2c.im));
            udp = cmul(num, S.inv_she);
        }
    }
    const cplx res = cmul(udp, S.th);                                                           // :200 (x fast_scale)
    UCF_UNIFORM_BLOCK_F(FAMILY);         // (the product stays on this path: merged with the early return above it became a select)
    return res;
}

template <int FAMILY, bool FOLD = false, bool LAY3 = true, bool LAY1 = true, bool NOFOLD = false>
UCF_DEV cplx fast_sample_z(const ucf_dev_params& P, const fast_common& S, int iz)
{
    const double zD = P.zD[iz];
    const int lay = P.zLay[iz];
    cplx chz, exz = cmake(0.0, 0.0);
    if (FAMILY == 0) return S.th;                                                                // :122-131
    if (FAMILY == 3) {
        const cplx cz = pcosh(prim<true>(S.eta.re * zD, S.eta.im * zD, S.sc));
        return cfma(S.th, cmul(S.mn_uod, cz), S.th);                                            // :437-439 (x fast_scale)
    }
    if (FAMILY == 1) return fast_hantush_z<1, FOLD, LAY3, LAY1, NOFOLD>(P, S, zD, lay, &chz, &exz);
    if (FAMILY == 4) {
        const cplx sH = fast_hantush_z<4, FOLD, LAY3, LAY1, NOFOLD>(P, S, zD, lay, &chz, &exz);
        if (!FOLD && LAY3 && lay == 3) return S.fd_use3 ? cadd(sH, cmul(S.fd_s13, chz)) : sH;
        return S.fd_use ? cadd(sH, cmul(S.fd_s1, chz)) : sH;                                    // :522-525
    }
    cplx u;
    if (!NOFOLD && P.model == 4) {
        u = S.th;
        if (S.any_small) chz = pcosh(prim(S.eta.re * zD, S.eta.im * zD, S.sc));
        if (S.any_large) { const double c = 1.0 - zD; exz = expneg_direct(S.eta.re * c, S.eta.im * c, S.sc); }
    } else {
        u = fast_hantush_z<2, FOLD, LAY3, LAY1, NOFOLD>(P, S, zD, lay, &chz, &exz);
        if (LAY1 && S.any_large && lay == 1) { const double c = 1.0 - zD; exz = expneg_direct(S.eta.re * c, S.eta.im * c, S.sc); }
    }
#ifdef UCF_SINGLE_RCP
    if (FOLD || (!NOFOLD && P.model == 4)) return cmul(csub(S.den, S.small_eta ? chz : exz), S.inv_den);
#endif
    const cplx top = (!FOLD && LAY3 && (NOFOLD || P.model != 4) && lay == 3) ? S.top3 : S.top;
    // u - top (cosh(eta zD) | exp(eta (zD - 1))) / den  (:85-87 | :89-91); a wave that is on one form only (the usual case)
    // does not select per lane
    // (one select per lane where a wave has lanes on the exponential form -- small_eta is true in every lane of a wave that has
    //  none; the three-way form "all small / all large / mixed" came back from the compiler as three selects with uniform masks)
    cplx g = chz;
    if (S.any_large) {
        g = S.small_eta ? chz : exz;
        asm volatile("" ::: "memory");      // (keeps the block a branch: no select at all in a wave on the cosh form)
    }
    return cfnma(cmul(top, g), S.inv_den, u);
}

}  // namespace UCF_NS
