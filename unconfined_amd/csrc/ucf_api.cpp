// ucf_api.cpp -- host side of the C ABI declared in include/ucf.h.
//
// Plan creation restates the numerical half of the reference's read_input and the
// driver's `first`-time setup (reference driver_io.f90:159-186,531-567,628-647;
// driver.f90:79-91,121-126,138-151,179-183; integration.f90:31-120) on the host:
// these run once, their results (J0 zeros, tanh-sinh weights, Gauss-Lobatto nodes)
// are uploaded once and stay resident.  This file is product code: it does not
// include, link or call anything under oracle/.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>   // (after <cstring>: its headers use memset on the host)

#include "ucf_plan.h"

// resident-workgroup budget: 256 CUs x 32 single-wave workgroups (only used when the interval areas live in global scratch, nz > 1); UCF_GRID_SLOTS in the environment overrides (tuning)
int ucf_finish_part = [] { const char* e = std::getenv("UCF_FINISH_PART"); return e ? std::atoi(e) : 0; }();
int ucf_grid_slots = [] { const char* e = std::getenv("UCF_GRID_SLOTS"); int v = e ? std::atoi(e) : 8192; return v > 0 ? v : 8192; }();

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(UCF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(UCF_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return UCF_OK;
}

// ---- driver_io.f90:88-333: the checks read_input performs before stopping
int validate(const ucf_params& P)
{
    if (P.model < 0 || P.model > 6) return fail(UCF_ERR_INVALID_MODEL, "invalid model choice %d (valid: 0..6)", P.model);
    if (P.model == 3 && P.MoenchM < 1) return fail(UCF_ERR_MOENCH, "number of Moench alphas must be >= 1 for model 3");
    if (P.MoenchM > UCF_MAX_MOENCH) return fail(UCF_ERR_MOENCH, "more than %d Moench alphas", UCF_MAX_MOENCH);
    if (P.model > 0 && (P.gammaSkin < 0.0 || P.d < 0.0 || P.l < 0.0))
        return fail(UCF_ERR_GEOMETRY, "negative geometry parameters (gamma, d, l)");
    if (P.b <= 0.0 || P.Kr <= 0.0 || P.Ss <= 0.0) return fail(UCF_ERR_AQUIFER, "zero or negative aquifer parameters (b, Kr, Ss)");
    if (P.model > 2 && (P.kappa <= 0.0 || P.Sy <= 0.0))
        return fail(UCF_ERR_AQUIFER, "zero or negative unconfined aquifer parameters (kappa, Sy)");
    double l = P.l, d = P.d;
    if (P.MNtype == 1) {                     // overrides happen before the check in the reference (:159-186)
        if (std::fabs(l - P.b) > FLT_EPSILON) l = P.b;
        if (d > FLT_EPSILON) d = 0.0;
    }
    if (P.model > 0 && d >= l) return fail(UCF_ERR_GEOMETRY, "screen top/bottom: l must be > d (l=%g d=%g)", l, d);
    if (P.model == 6) {
        if (P.ac < 0.0 || P.ak < 0.0 || P.usL < 0.0 || P.psia < 0.0 || P.psik < 0.0)
            return fail(UCF_ERR_MISHRA_NEUMAN, "invalid Mishra/Neuman parameters (a_c, a_k, L, psi_a, psi_k)");
        if (P.MNtype == 2 && P.order < 3) return fail(UCF_ERR_MISHRA_NEUMAN, "Mishra/Neuman finite difference order must be >= 3");
        if (P.MNtype == 0)
            return fail(UCF_ERR_UNSUPPORTED, "Mishra/Neuman type 0 is the quad-precision ARB path, excluded from this build");
        if (P.MNtype < 0 || P.MNtype > 2) return fail(UCF_ERR_MISHRA_NEUMAN, "invalid Mishra/Neuman solution type %d", P.MNtype);
    }
    if ((P.model == 4 || P.model == 5) && P.beta < 0.0) return fail(UCF_ERR_MALAMA_BETA, "Malama beta cannot be negative");
    if (P.model == 3)
        for (int i = 0; i < P.MoenchM; i++)
            if (P.MoenchAlpha[i] < 0.0) return fail(UCF_ERR_MOENCH, "Moench alphas cannot be negative");
    if (P.M < 2) return fail(UCF_ERR_DEHOOG, "de Hoog M must be >= 2 (M=%d)", P.M);
    if (P.M > UCF_MAX_LAP_M) return fail(UCF_ERR_UNSUPPORTED, "de Hoog M=%d: the wave-cooperative inversion holds at most two Laplace samples per lane (M <= %d)", P.M, UCF_MAX_LAP_M);
    if (P.k - P.R < 2) return fail(UCF_ERR_TANH_SINH, "tanh-sinh k (%d) too low for %d Richardson levels", P.k, P.R);
    if (P.R < 1) return fail(UCF_ERR_TANH_SINH, "Richardson extrapolation level must be >= 1");
    if (P.R > UCF_MAX_R || P.k > 20) return fail(UCF_ERR_UNSUPPORTED, "tanh-sinh k=%d R=%d beyond build limits", P.k, P.R);
    if (P.j0s[0] < 1 || P.j0s[1] < 1 || P.nacc < 1 || P.k < 1)
        return fail(UCF_ERR_GAUSS_LOBATTO, "min/max split, # accelerated terms and k must be >= 1");
    if (P.ord < 3) return fail(UCF_ERR_GAUSS_LOBATTO, "Gauss-Lobatto order must be >= 3");
    if (P.model == 2 && (P.rwobs <= 0.0 || P.sF <= 0.0))
        return fail(UCF_ERR_OBSERVATION, "model 2 needs a positive observation-well radius and shape factor");   // driver_io.f90:374-383
    if (P.timeType <= -101) return fail(UCF_ERR_UNSUPPORTED, "piecewise-linear pumping schedules (time behaviour <= -101) index out of bounds in the reference (time.f90:101,115) and are not built");
    if (P.timeType < -UCF_MAX_SCHEDULE || P.timeType == 0 || P.timeType > 8) return fail(UCF_ERR_UNSUPPORTED, "time behaviour %d does not exist (1..8, -1..-99 do)", P.timeType);
    if (P.timeType == -100) return fail(UCF_ERR_UNSUPPORTED, "time behaviour -100: the reference allocates a single parameter for it (driver_io.f90:124)");
    return UCF_OK;
}

// ---- driver_io.f90:159-186, 531-567
void nondimensionalise(const ucf_params& P, ucf_derived& D)
{
    const double PI = 4.0 * std::atan(1.0);
    std::memset(&D, 0, sizeof(D));
    double l = P.l, d = P.d, ac = P.ac;
    if (P.MNtype == 1) {
        if (std::fabs(P.ac - P.ak) > FLT_EPSILON) ac = P.ak;
        if (std::fabs(l - P.b) > FLT_EPSILON) l = P.b;
        if (d > FLT_EPSILON) d = 0.0;
    }
    D.l_eff = l; D.d_eff = d; D.ac_eff = ac;
    D.Lc = P.b;
    D.Tc = D.Lc * D.Lc / (P.Kr / P.Ss);
    D.Hc = P.Q / (4 * PI * P.Kr * P.b);
    D.sigma = P.Sy / (P.Ss * P.b);
    D.alphaD = P.kappa / D.sigma;
    D.betaD = P.beta / D.Lc;
    D.lD = l / D.Lc;
    D.dD = d / D.Lc;
    D.bD = D.lD - D.dD;
    D.rDw = P.rw / D.Lc;
    D.rDwobs = P.rwobs / D.Lc;
    for (int m = 0; m < P.MoenchM && m < UCF_MAX_MOENCH; m++)
        D.MoenchGamma[m] = P.MoenchAlpha[m] * D.Lc * P.Sy / (P.kappa * P.Kr);
    D.acD = ac * D.Lc;
    D.akD = P.ak * D.Lc;
    D.lambdaD = (P.ak - ac) * D.Lc;
    D.psiaD = P.psia / D.Lc;
    D.psikD = P.psik / D.Lc;
    D.usLD = P.usL / D.Lc;
    D.b1 = P.psia - P.psik;
    D.PsiD = D.b1 / D.Lc;
    D.np = 2 * P.M + 1;
    D.N = (1 << P.k) - 1;
    D.nj0z = (P.j0s[0] > P.j0s[1] ? P.j0s[0] : P.j0s[1]) + P.nacc + 1;
    D.nabs = D.N + P.nacc * (P.ord - 2);
}

// ---- driver_io.f90:628-647: Newton on J0 from the asymptotic guess (i+3/4)*pi
void j0_zeros(int n, double* z)
{
    const double PI = 4.0 * std::atan(1.0);
    for (int i = 0; i < n; i++) {
        double x = (i + 0.75) * PI;
        for (int it = 0; it < 100; it++) {
            const double dx = j0(x) / j1(x);
            x = x + dx;
            if (std::fabs(dx) < std::nextafter(std::fabs(x), INFINITY) - std::fabs(x)) break;   // spacing(x)
        }
        z[i] = x;
    }
}

// ---- integration.f90:31-67: weights of a 2^k-1 point rule (sum normalised to 2) and, on
// request, tanh(u2)+1 (the abscissa on [0,s] is (that)*s/2, applied per point on the device)
void tanh_sinh_level(int k, double* w, double* x_unit)
{
    const double PIOV2 = 2.0 * std::atan(1.0);
    const int N = (1 << k) - 1, r = (N - 1) / 2;
    const double h = 4.0 / (double)(1 << k);
    std::vector<double> u2(N);
    for (int i = -r; i <= r; i++) {
        const double u1 = PIOV2 * std::cosh(h * i);
        u2[i + r] = PIOV2 * std::sinh(h * i);
        const double c = std::cosh(u2[i + r]);
        w[i + r] = u1 / (c * c);
    }
    double sum = 0.0;
    for (int i = 0; i < N; i++) sum = sum + w[i];
    for (int i = 0; i < N; i++) w[i] = 2.0 * w[i] / sum;
    if (x_unit)
        for (int i = 0; i < N; i++) x_unit[i] = std::tanh(u2[i]) + 1.0;
}

// ---- integration.f90:70-120: Newton on the Legendre recurrence; interior nodes only
void gauss_lobatto(int ord, double* xo, double* wo)
{
    const int N = ord - 1, N1 = N + 1;
    const double PI = 4.0 * std::atan(1.0);
    std::vector<double> Pm((size_t)ord * ord, 0.0), x(ord), xold(ord, 2.0);
    auto PP = [&](int i, int k) -> double& { return Pm[(size_t)(k - 1) * ord + i]; };
    for (int i = 0; i <= N; i++) x[i] = std::cos(PI * i / N);
    for (int it = 0; it < 1000; it++) {
        double mx = 0.0;
        for (int i = 0; i < ord; i++) mx = std::fmax(mx, std::fabs(x[i] - xold[i]));
        if (!(mx > DBL_EPSILON)) break;
        for (int i = 0; i < ord; i++) { xold[i] = x[i]; PP(i, 1) = 1.0; PP(i, 2) = x[i]; }
        for (int k = 2; k <= N; k++)
            for (int i = 0; i < ord; i++) PP(i, k + 1) = ((2 * k - 1) * x[i] * PP(i, k) - (k - 1) * PP(i, k - 1)) / k;
        for (int i = 0; i < ord; i++) x[i] = xold[i] - (x[i] * PP(i, N1) - PP(i, N)) / (N1 * PP(i, N1));
    }
    for (int i = 1; i <= ord - 2; i++) {
        xo[i - 1] = x[i];
        wo[i - 1] = 2.0 / ((N * N1) * (PP(i, N1) * PP(i, N1)));
    }
}

// depths per launch.  The integrate kernels keep (R+1) KB of accumulators per depth in LDS and share the
// z-independent half of every sample among the depths of a launch: ~12 KB (2 depths at R = 4) balances occupancy
// against that sharing (measured: 21 depths of the C2 settings take 54 / 44 / 49 / 51 / 66 ms at 1 / 2 / 3 / 4 / 7
// depths per launch).  The monolithic point_kernel (faithful finite-difference closure) takes as many as keep its
// footprint <= 40 KB.  At most UCF_MAX_NZ; UCF_Z_CHUNK overrides (diagnostic).
int z_chunk(const ucf_plan* plan)
{
    const int R = plan->P.R, nacc = plan->P.nacc;
    ucf_dev_params one = plan->dev;
    one.nz = 1;
    const bool split = (plan->mode == 1 ? ucf_fast::state_bytes_per_item(one) : ucf_faithful::state_bytes_per_item(one)) != 0;
    int n;
    if (split) {
        n = (int)(((size_t)12 * 1024) / ((size_t)(R + 1) * UCF_WAVE * 16));
    } else {
        const size_t scr = (size_t)(2 * nacc > R ? 2 * nacc : R) * 16 * 16;
        n = (int)(((size_t)40 * 1024 - scr) / ((size_t)(R + 1) * UCF_WAVE * 16));
    }
    static const int forced = [] { const char* e = std::getenv("UCF_Z_CHUNK"); return e ? std::atoi(e) : 0; }();
    if (forced > 0) n = forced;
    if (n < 1) n = 1;
    if (n > UCF_MAX_NZ) n = UCF_MAX_NZ;
    return n;
}

int fill_call_params(const ucf_plan* plan, int nz, const double* zD, const int* zLay, ucf_dev_params& dp,
                     int nz_out = 0, int z_off = 0)
{
    if (nz < 1 || nz > UCF_MAX_NZ) return fail(UCF_ERR_BAD_ARGUMENT, "nz=%d out of range 1..%d", nz, UCF_MAX_NZ);
    if (!zD || !zLay) return fail(UCF_ERR_BAD_ARGUMENT, "zD / zLay must not be NULL");
    dp = plan->dev;
    dp.nz = nz;
    dp.nz_out = nz_out > 0 ? nz_out : nz;
    dp.z_off = z_off;
    for (int i = 0; i < nz; i++) {
        if (zLay[i] < 1 || zLay[i] > 3) return fail(UCF_ERR_BAD_ARGUMENT, "zLay[%d]=%d not in 1..3", i, zLay[i]);
        dp.zD[i] = zD[i];
        dp.zLay[i] = zLay[i];
    }
    dp.any_lay3 = 0;
    for (int i = 0; i < nz; i++) dp.any_lay3 |= (zLay[i] == 3);
    // the fast evaluators take sin/cos of Im(eta)*c for c in {1, dD, 1-lD, dD1-1, zD, 1-zD, dD1-zD}:
    // the largest |c| bounds the argument (two-stage Cody-Waite reduction is good below 1e6)
    double cmax = 1.0;
    const double cs[] = {dp.dD, dp.lD1, dp.dD1 - 1.0};
    for (double c : cs) cmax = std::fmax(cmax, std::fabs(c));
    for (int i = 0; i < nz; i++) {
        cmax = std::fmax(cmax, std::fabs(zD[i]));
        cmax = std::fmax(cmax, std::fabs(1.0 - zD[i]));
        cmax = std::fmax(cmax, std::fabs(dp.dD1 - zD[i]));
    }
    dp.fast_im_max = 1.0e6 / cmax;
    if (!(dp.fold_dD && dp.fold_lD1)) {
        // products the reference forms (laplace_hankel_solutions.f90:179-180 and the zD = 1 evaluation of :81):
        // sinh(eta dD) cosh(eta zD), sinh(eta lD1) cosh(eta (1-zD)), sinh(eta dD) cosh(eta): exponents eta*(c1 + c2)
        double cprod = 1.0 + std::fabs(dp.dD);
        for (int i = 0; i < nz; i++) {
            if (zLay[i] == 1) continue;                       // below the screen: g(3) cosh(eta zD), no such product
            cprod = std::fmax(cprod, std::fabs(dp.dD) + std::fabs(zD[i]));
            cprod = std::fmax(cprod, std::fabs(dp.lD1) + std::fabs(1.0 - zD[i]));
        }
        dp.fast_eta_max = 700.0 / cprod;
        if (dp.model == 6 && dp.MNtype == 2 && dp.fast_eta_max > 350.0) dp.fast_eta_max = 350.0;   // FD closure: cosh(eta)/h^2 terms
    }
    // depths outside the aquifer (the reference evaluates them all the same: growing exponentials): the fast
    // evaluators assume 0 <= zD <= 1, so the generic evaluator takes every abscissa of such a call
    for (int i = 0; i < nz; i++)
        if (!(zD[i] >= 0.0 && zD[i] <= 1.0)) dp.fast_eta_max = -1.0;
    static const double eta_cap = [] { const char* e = std::getenv("UCF_FAST_ETA_MAX"); return e ? std::atof(e) : 0.0; }();
    if (eta_cap > 0.0 && dp.fast_eta_max > eta_cap) dp.fast_eta_max = eta_cap;      // diagnostic: hand more of the range to the generic evaluator
    return UCF_OK;
}

struct dev_buf {
    void* p = nullptr;
    ~dev_buf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess ? 0 : 1; }
};

}  // namespace

extern "C" {

int ucf_version(void) { return UCF_VERSION; }
const char* ucf_last_error(void) { return g_err; }

const char* ucf_status_string(int status)
{
    switch (status) {
    case UCF_OK: return "ok";
    case UCF_ERR_INVALID_MODEL: return "invalid model";
    case UCF_ERR_GEOMETRY: return "invalid well geometry";
    case UCF_ERR_AQUIFER: return "invalid aquifer parameters";
    case UCF_ERR_MISHRA_NEUMAN: return "invalid Mishra/Neuman parameters";
    case UCF_ERR_MALAMA_BETA: return "invalid Malama beta";
    case UCF_ERR_MOENCH: return "invalid Moench parameters";
    case UCF_ERR_DEHOOG: return "invalid de Hoog parameters";
    case UCF_ERR_TANH_SINH: return "invalid tanh-sinh parameters";
    case UCF_ERR_GAUSS_LOBATTO: return "invalid Gauss-Lobatto / split parameters";
    case UCF_ERR_UNSUPPORTED: return "valid in the reference but not built";
    case UCF_ERR_BAD_ARGUMENT: return "bad argument";
    case UCF_ERR_NO_DEVICE: return "no HIP device";
    case UCF_ERR_HIP: return "HIP runtime error";
    case UCF_ERR_NOMEM: return "out of memory";
    case UCF_ERR_OBSERVATION: return "invalid observation well";
    default: return "unknown status";
    }
}

namespace {
// Everything of a plan that depends on the parameters.  create: also the quadrature tables (which depend only on the
// numerical settings k, R, ord, nacc, the J0 split) and the device allocation.  !create (ucf_plan_update): the new set
// must leave those settings and the model alone; only the parameter-dependent table segments (finite-difference
// exponentials, pumping schedule) are uploaded again.
int plan_set_params(ucf_plan* pl, const ucf_params& Pin, bool create)
{
    ucf_params Pn = Pin;
    if (Pn.tol < DBL_EPSILON) Pn.tol = DBL_EPSILON;                  // driver_io.f90:311-314
    if (!create) {
        const ucf_params& O = pl->P;
        const bool same = O.model == Pn.model && O.MNtype == Pn.MNtype && O.order == Pn.order && O.timeType == Pn.timeType &&
                          O.MoenchM == Pn.MoenchM && O.M == Pn.M && O.k == Pn.k && O.R == Pn.R && O.nacc == Pn.nacc && O.ord == Pn.ord &&
                          O.j0s[0] == Pn.j0s[0] && O.j0s[1] == Pn.j0s[1];
        if (!same)
            return fail(UCF_ERR_BAD_ARGUMENT, "ucf_plan_update: the model and the numerical settings (M, k, R, nacc, ord, J0 split, "
                                              "schedule length, FD order, number of Moench terms) must stay as they are; create a new plan");
    }
    pl->P = Pn;
    const ucf_params& P = pl->P;
    nondimensionalise(P, pl->D);
    const ucf_derived& D = pl->D;
    const int N = D.N, R = P.R, ngl = P.ord - 2;
    ucf_dev_params& dp = pl->dev;
    if (create) {
        (void)hipGetDevice(&pl->device);
        pl->h_j0z = (double*)std::malloc(sizeof(double) * D.nj0z);
        pl->h_ts_x = (double*)std::malloc(sizeof(double) * N);
        pl->h_ts_w = (double*)std::calloc((size_t)R * N, sizeof(double));
        pl->h_gl_x = (double*)std::malloc(sizeof(double) * ngl);
        pl->h_gl_w = (double*)std::malloc(sizeof(double) * ngl);
        j0_zeros(D.nj0z, pl->h_j0z);
        for (int j = 1; j <= R; j++) {                                    // driver.f90:86-91
            const int kv = P.k - R + j;
            pl->Nv[j - 1] = (1 << kv) - 1;
            dp.hv[j - 1] = 4.0 / (double)(1 << kv);
            tanh_sinh_level(kv, pl->h_ts_w + (size_t)(j - 1) * N, (j == R) ? pl->h_ts_x : nullptr);
        }
        gauss_lobatto(P.ord, pl->h_gl_x, pl->h_gl_w);
    }

    // FD table exp(-beta1*(j-1)*h)  (laplace_hankel_solutions.f90:494)
    std::vector<double> fd_e;
    if (P.model == 6 && P.MNtype == 2) {
        const double h = D.usLD / (double)(P.order - 1);
        const double beta1 = -D.lambdaD;
        fd_e.resize(P.order);
        for (int j = 1; j <= P.order; j++) fd_e[j - 1] = std::exp(-(beta1 * (double)(j - 1) * h));
        dp.fd_h = h;
        dp.fd_invhsq = 1.0 / (h * h);
        dp.fd_beta0 = D.ac_eff * P.Sy / P.Ss;
        dp.fd_beta3 = D.akD;
        dp.fd_expmb2 = std::exp(-(P.ak * D.b1));
    }
    if (P.model == 2) {                                               // laplace_hankel_solutions.f90:250-253
        const double PI = 4.0 * std::atan(1.0);
        dp.hs_rDw = D.rDw;
        dp.hs_CDw = D.rDw * D.rDw / (2.0 * (D.l_eff - D.d_eff) * P.Ss);
        dp.hs_tDb = PI * (D.rDwobs * D.rDwobs) / (P.sF * P.Ss);
    }
    if (P.model == 6 && P.MNtype == 1) {                              // :420-427
        const double beta0 = P.ak * P.b;
        const double phiDa = P.psia / P.b, phiDk = P.psik / P.b;
        dp.mn_vartheta = beta0 * P.Sy / (P.Ss * P.b) * std::exp(-(beta0 * (phiDa - phiDk)));
        dp.mn_u0 = beta0 / 2.0;
    }

    // piecewise-constant pumping schedule (time.f90:81-95): increments and their sum, once
    std::vector<double> sched;
    if (P.timeType < 0) {
        const int n = -P.timeType;
        sched.resize(2 * n + 2);
        double qprev = 0.0, sum = 0.0;
        for (int k = 0; k < n; k++) {
            const double dq = P.timeParExt[n + 1 + k] - qprev;
            qprev = P.timeParExt[n + 1 + k];
            sched[k] = P.timeParExt[k];
            sched[n + k] = dq;
            sum = (k == 0) ? dq : sum + dq;
        }
        sched[2 * n] = P.timeParExt[n];
        sched[2 * n + 1] = sum;
    }
    if (create) {
        // one device allocation for all tables
        const size_t n_tab = (size_t)N + (size_t)R * N + 2 * (size_t)ngl + D.nj0z + fd_e.size() + sched.size();
        std::vector<double> host(n_tab);
        size_t o = 0;
        pl->o_tsx = o; std::memcpy(&host[o], pl->h_ts_x, sizeof(double) * N); o += N;
        pl->o_tsw = o; std::memcpy(&host[o], pl->h_ts_w, sizeof(double) * (size_t)R * N); o += (size_t)R * N;
        pl->o_glx = o; std::memcpy(&host[o], pl->h_gl_x, sizeof(double) * ngl); o += ngl;
        pl->o_glw = o; std::memcpy(&host[o], pl->h_gl_w, sizeof(double) * ngl); o += ngl;
        pl->o_j0z = o; std::memcpy(&host[o], pl->h_j0z, sizeof(double) * D.nj0z); o += D.nj0z;
        pl->o_fde = o; if (!fd_e.empty()) std::memcpy(&host[o], fd_e.data(), sizeof(double) * fd_e.size());
        o += fd_e.size();
        pl->o_sched = o; if (!sched.empty()) std::memcpy(&host[o], sched.data(), sizeof(double) * sched.size());
        pl->tables_bytes = n_tab * sizeof(double);
        if (hipMalloc((void**)&pl->d_tables, pl->tables_bytes) != hipSuccess)
            return fail(UCF_ERR_NOMEM, "hipMalloc of %zu table bytes failed", pl->tables_bytes);
        if (hipMemcpy(pl->d_tables, host.data(), pl->tables_bytes, hipMemcpyHostToDevice) != hipSuccess)
            return fail(UCF_ERR_HIP, "table upload failed");
    } else if (!fd_e.empty() || !sched.empty()) {
        (void)hipDeviceSynchronize();                                  // launches of the old parameter set may still read them
        if (!fd_e.empty() && hipMemcpy(pl->d_tables + pl->o_fde, fd_e.data(), sizeof(double) * fd_e.size(), hipMemcpyHostToDevice) != hipSuccess)
            return fail(UCF_ERR_HIP, "table upload failed");
        if (!sched.empty() && hipMemcpy(pl->d_tables + pl->o_sched, sched.data(), sizeof(double) * sched.size(), hipMemcpyHostToDevice) != hipSuccess)
            return fail(UCF_ERR_HIP, "table upload failed");
    }

    dp.model = P.model; dp.MNtype = P.MNtype; dp.order = P.order; dp.timeType = P.timeType; dp.MoenchM = P.MoenchM;
    dp.M = P.M; dp.np = D.np; dp.k = P.k; dp.N = N; dp.R = R; dp.nacc = P.nacc; dp.ngl = ngl; dp.nz = 0;
    dp.nj0z = D.nj0z;
    dp.timePar[0] = P.timePar[0]; dp.timePar[1] = P.timePar[1];
    dp.kappa = P.kappa; dp.alphaD = D.alphaD; dp.beta = P.beta;
    dp.lD = D.lD; dp.dD = D.dD; dp.bD = D.bD; dp.dD1 = 1.0 - D.dD; dp.lD1 = 1.0 - D.lD;
    for (int m = 0; m < P.MoenchM; m++) dp.MoenchInvGamma[m] = 1.0 / D.MoenchGamma[m];
    dp.alpha = P.alpha; dp.logtol = std::log(P.tol); dp.maxexp = -std::log(DBL_EPSILON) / 3.0;   // constants.f90:66
    dp.inv_kappa = 1.0 / P.kappa;
    dp.inv_bD = 1.0 / D.bD;
    dp.fold_dD = (D.dD == 0.0);                       // sinh(eta*0) == 0 exactly
    dp.fold_lD1 = (dp.lD1 == 0.0);
    dp.share_g1top = ((dp.dD1 - 1.0) == -D.dD);       // cosh(eta*(dD1-1)) == cosh(eta*dD) bit for bit
    // the fast path is only used where the REFERENCE's own intermediates stay finite (beyond that its results are
    // shaped by Inf/NaN and the in-band rules, which the generic evaluator reproduces): no cosh/sinh (<= e^{Re eta})
    // and none of the products of two of them that the reference forms may overflow.  Refined per call in
    // fill_call_params (the products depend on the depths).
    dp.fast_eta_max = (dp.fold_dD && dp.fold_lD1) ? 700.0 : 350.0;
    dp.ts_x = pl->d_tables + pl->o_tsx;
    dp.ts_w = pl->d_tables + pl->o_tsw;
    dp.gl_x = pl->d_tables + pl->o_glx;
    dp.gl_w = pl->d_tables + pl->o_glw;
    dp.j0z = pl->d_tables + pl->o_j0z;
    dp.fd_e = pl->d_tables + pl->o_fde;
    dp.sched = pl->d_tables + pl->o_sched;
    return UCF_OK;
}
}  // namespace

int ucf_plan_create(const ucf_params* Pin, ucf_plan** out)
{
    if (!Pin || !out) return fail(UCF_ERR_BAD_ARGUMENT, "NULL argument");
    *out = nullptr;
    int rc = validate(*Pin);
    if (rc) return rc;
    rc = require_device();
    if (rc) return rc;
    ucf_plan* pl = new (std::nothrow) ucf_plan();
    if (!pl) return fail(UCF_ERR_NOMEM, "host allocation failed");
    std::memset(pl, 0, sizeof(*pl));
    rc = plan_set_params(pl, *Pin, true);
    if (rc) { ucf_plan_destroy(pl); return rc; }
    pl->mode = 0;
    *out = pl;
    return UCF_OK;
}

int ucf_plan_update(ucf_plan* pl, const ucf_params* Pin)
{
    if (!pl || !Pin) return fail(UCF_ERR_BAD_ARGUMENT, "NULL argument");
    int rc = validate(*Pin);
    if (rc) return rc;
    return plan_set_params(pl, *Pin, false);
}

void ucf_plan_destroy(ucf_plan* pl)
{
    if (!pl) return;
    if (pl->d_tables) (void)hipFree(pl->d_tables);
    if (pl->d_work) (void)hipFree(pl->d_work);
    if (pl->d_totlap) (void)hipFree(pl->d_totlap);
    if (pl->d_glscr) (void)hipFree(pl->d_glscr);
    if (pl->d_expand) (void)hipFree(pl->d_expand);
    if (pl->d_sort) (void)hipFree(pl->d_sort);
    if (pl->d_state) (void)hipFree(pl->d_state);
    if (pl->d_ndone) (void)hipFree(pl->d_ndone);
    if (pl->ev0) (void)hipEventDestroy((hipEvent_t)pl->ev0);
    if (pl->ev1) (void)hipEventDestroy((hipEvent_t)pl->ev1);
    std::free(pl->h_j0z); std::free(pl->h_ts_x); std::free(pl->h_ts_w); std::free(pl->h_gl_x); std::free(pl->h_gl_w);
    delete pl;
}

int ucf_plan_derived(const ucf_plan* pl, ucf_derived* out)
{
    if (!pl || !out) return fail(UCF_ERR_BAD_ARGUMENT, "NULL argument");
    *out = pl->D;
    return UCF_OK;
}

int ucf_plan_j0z(const ucf_plan* pl, int n, double* j0z)
{
    if (!pl || !j0z || n > pl->D.nj0z) return fail(UCF_ERR_BAD_ARGUMENT, "bad j0z request");
    std::memcpy(j0z, pl->h_j0z, sizeof(double) * n);
    return UCF_OK;
}

int ucf_plan_tanh_sinh(const ucf_plan* pl, int level, int n, double* w, double* x_unit)
{
    if (!pl || !w || level < 1 || level > pl->P.R) return fail(UCF_ERR_BAD_ARGUMENT, "bad tanh-sinh level");
    if (n != pl->Nv[level - 1]) return fail(UCF_ERR_BAD_ARGUMENT, "level %d has %d abscissae", level, pl->Nv[level - 1]);
    std::memcpy(w, pl->h_ts_w + (size_t)(level - 1) * pl->D.N, sizeof(double) * n);
    if (x_unit) {
        if (level != pl->P.R) return fail(UCF_ERR_BAD_ARGUMENT, "abscissae exist for the densest level only");
        std::memcpy(x_unit, pl->h_ts_x, sizeof(double) * n);
    }
    return UCF_OK;
}

int ucf_plan_gauss_lobatto(const ucf_plan* pl, int n, double* x, double* w)
{
    if (!pl || !x || !w || n != pl->P.ord - 2) return fail(UCF_ERR_BAD_ARGUMENT, "bad Gauss-Lobatto request");
    std::memcpy(x, pl->h_gl_x, sizeof(double) * n);
    std::memcpy(w, pl->h_gl_w, sizeof(double) * n);
    return UCF_OK;
}

int ucf_plan_set_timing(ucf_plan* pl, int enable)
{
    if (!pl) return fail(UCF_ERR_BAD_ARGUMENT, "NULL plan");
    pl->timing = enable ? 1 : 0;
    pl->ev_valid = 0;
    return UCF_OK;
}

int ucf_plan_kernel_ms(ucf_plan* pl, double* ms, const char** kernel_name)
{
    if (!pl || !ms) return fail(UCF_ERR_BAD_ARGUMENT, "NULL argument");
    if (!pl->ev_valid) return fail(UCF_ERR_BAD_ARGUMENT, "no timed launch: enable timing and run a single-chunk grid call first");
    HIP_TRY(hipEventSynchronize((hipEvent_t)pl->ev1));
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, (hipEvent_t)pl->ev0, (hipEvent_t)pl->ev1));
    *ms = (double)f;
    if (kernel_name) *kernel_name = pl->last_kernel;
    return UCF_OK;
}

int ucf_plan_set_mode(ucf_plan* pl, int mode)
{
    // bit 0: 0 faithful / 1 fast;  bit 1 (diagnostic): force the lane = Laplace-sample layout for grids
    if (!pl || mode < 0 || mode > 3) return fail(UCF_ERR_BAD_ARGUMENT, "mode must be 0 (faithful) or 1 (fast) [+2: lane=sample layout]");
    pl->mode = mode & 1;
    pl->force_layout0 = (mode >> 1) & 1;
    return UCF_OK;
}

// ---- utility.f90:34-57
int ucf_linspace(double lo, double hi, int n, double* v)
{
    if (n < 1 || !v) return fail(UCF_ERR_BAD_ARGUMENT, "bad linspace request");
    if (n == 1) {
        v[0] = (lo + hi) / 2.0;
    } else {
        const double dx = (hi - lo) / (n - 1);
        for (int i = 1; i <= n; i++) v[i - 1] = lo + (i - 1) * dx;
    }
    return UCF_OK;
}
int ucf_logspace(int lo, int hi, int n, double* v)
{
    int rc = ucf_linspace((double)lo, (double)hi, n, v);
    if (rc) return rc;
    for (int i = 0; i < n; i++) v[i] = std::pow(10.0, v[i]);
    return UCF_OK;
}

// ---- driver_io.f90:575-586
int ucf_zlay(const ucf_plan* pl, int nz, const double* zD, int* zLay)
{
    if (!pl || !zD || !zLay) return fail(UCF_ERR_BAD_ARGUMENT, "NULL argument");
    for (int i = 0; i < nz; i++) {
        if (zD[i] <= 0.0 || zD[i] < (1.0 - pl->D.lD)) zLay[i] = 1;
        else if ((zD[i] - 1.0) >= 0.0 || zD[i] < (1.0 - pl->D.dD)) zLay[i] = 2;
        else zLay[i] = 3;
    }
    return UCF_OK;
}

// ---- driver_io.f90:654-664
int ucf_split_vector(const ucf_plan* pl, int nt, const double* tD, int* sv)
{
    if (!pl || !tD || !sv || nt < 1) return fail(UCF_ERR_BAD_ARGUMENT, "bad split-vector request");
    const int* j0s = pl->P.j0s;
    const int mx = j0s[0] > j0s[1] ? j0s[0] : j0s[1], mn = j0s[0] < j0s[1] ? j0s[0] : j0s[1];
    const int zrange = mx - mn;
    double lmin = INFINITY, lmax = -INFINITY;
    for (int i = 0; i < nt; i++) {
        const double lg = std::log10(tD[i]);
        lmin = std::fmin(lmin, lg);
        lmax = std::fmax(lmax, lg);
    }
    const int minlsp = (int)std::floor(lmin), maxlsp = (int)std::ceil(lmax);
    const int sprange = maxlsp - minlsp + 1;
    for (int i = 0; i < nt; i++) sv[i] = mn + (int)(zrange * ((maxlsp - std::log10(tD[i])) / sprange));
    return UCF_OK;
}

// ---- driver.f90:234-243
int ucf_screen_average(int npts, int zOrd, const double* h, double* havg)
{
    if (!h || !havg || zOrd < 1) return fail(UCF_ERR_BAD_ARGUMENT, "bad screen-average request");
    for (int i = 0; i < npts; i++) {
        const double* v = h + (size_t)i * zOrd;
        if (zOrd == 1) { havg[i] = v[0]; continue; }
        double s = v[1];
        for (int j = 2; j < zOrd; j++) s = s + v[j];
        havg[i] = ((v[0] + 2.0 * s) + v[zOrd - 1]) / (2 * zOrd);
    }
    return UCF_OK;
}

// ---- the hot path
namespace {

// abscissa-table workspace: nrows x nabs x (a, a*J0) doubles
int ensure_work(ucf_plan* pl, size_t bytes)
{
    if (pl->work_bytes >= bytes) return UCF_OK;
    if (pl->d_work) {
        (void)hipDeviceSynchronize();      // a previous launch may still read the old table
        (void)hipFree(pl->d_work);
        pl->d_work = nullptr;
        pl->work_bytes = 0;
    }
    if (hipMalloc((void**)&pl->d_work, bytes) != hipSuccess) return fail(UCF_ERR_NOMEM, "hipMalloc of %zu workspace bytes failed", bytes);
    pl->work_bytes = bytes;
    return UCF_OK;
}

// scratch for the finished interval areas of every resident workgroup
int ensure_glscr(ucf_plan* pl, int nz)
{
    const size_t need = (size_t)UCF_GRID_SLOTS * pl->P.nacc * nz * UCF_WAVE * 2 * sizeof(double);
    if (pl->glscr_bytes >= need) return UCF_OK;
    if (pl->d_glscr) { (void)hipDeviceSynchronize(); (void)hipFree(pl->d_glscr); pl->d_glscr = nullptr; pl->glscr_bytes = 0; }
    if (hipMalloc((void**)&pl->d_glscr, need) != hipSuccess) return fail(UCF_ERR_NOMEM, "hipMalloc of %zu area-scratch bytes failed", need);
    pl->glscr_bytes = need;
    return UCF_OK;
}

// state of `items` work items between integrate_kernel and point_kernel (fast flavour, Hantush-based models)
size_t state_item_bytes(const ucf_plan* pl, const ucf_dev_params& dp)
{
    return (pl->mode == 1) ? ucf_fast::state_bytes_per_item(dp) : ucf_faithful::state_bytes_per_item(dp);
}
int ensure_state(ucf_plan* pl, const ucf_dev_params& dp, size_t items)
{
    const size_t need = state_item_bytes(pl, dp) * items;
    if (need == 0) return UCF_OK;
    if (pl->state_bytes < need) {
        if (pl->d_state) { (void)hipDeviceSynchronize(); (void)hipFree(pl->d_state); pl->d_state = nullptr; pl->state_bytes = 0; }
        if (hipMalloc((void**)&pl->d_state, need) != hipSuccess) return fail(UCF_ERR_NOMEM, "hipMalloc of %zu integration-state bytes failed", need);
        pl->state_bytes = need;
    }
    if (pl->ndone_items < items) {      // [done per item | count of unfinished | unfinished items]
        if (pl->d_ndone) { (void)hipDeviceSynchronize(); (void)hipFree(pl->d_ndone); pl->d_ndone = nullptr; pl->ndone_items = 0; }
        if (hipMalloc((void**)&pl->d_ndone, (2 * items + 1) * sizeof(int)) != hipSuccess) return fail(UCF_ERR_NOMEM, "hipMalloc of %zu counters failed", items);
        pl->ndone_items = items;
    }
    return UCF_OK;
}
// abscissa-table bytes per chunk of an arbitrary point list (UCF_TABLE_BYTES, default 256 MiB)
size_t table_budget()
{
    static const size_t b = [] { const char* e = std::getenv("UCF_TABLE_BYTES"); long long v = e ? std::atoll(e) : 0; return v > 0 ? (size_t)v : ((size_t)256 << 20); }();
    return b;
}
// work items per launch such that their state stays within UCF_STATE_BYTES (default 8 GiB of the 288 GB)
size_t state_budget()
{
    static const size_t b = [] { const char* e = std::getenv("UCF_STATE_BYTES"); long long v = e ? std::atoll(e) : 0; return v > 0 ? (size_t)v : ((size_t)8 << 30); }();
    return b;
}

// lane layout of arbitrary point lists: 3 = lane is a point (all 64 lanes live whatever M is), 0 = lane is a Laplace
// sample (2M+1 of 64 lanes live).  UCF_BATCH_LAYOUT=0 forces the latter (diagnostic).
int batch_layout()
{
    static const int v = [] { const char* e = std::getenv("UCF_BATCH_LAYOUT"); return e ? std::atoi(e) : 3; }();
    return v;
}

int launch_points_any(ucf_plan* pl, const ucf_dev_params& dp, int npts, int per_point, int nr, int nsv, int svmin,
                      const double* d_tD, const double* d_rD, const int* d_sv, double* d_h, double* d_dh,
                      ucf_stats* d_stats, void* stream, const ucf_dev_params* d_params = nullptr, int ppp = 1, size_t p0 = 0,
                      int npts_call = -1)
{
    // (d_params: parameter batch, plan of point q of this call = (p0 + q) / ppp; p0 must be a multiple of ppp)
    int rc = ensure_glscr(pl, dp.nz);
    if (rc) return rc;
    const bool chunked = pl->D.np > UCF_WAVE;     // more Laplace samples than lanes: (point, 64-sample chunk) work items
    const size_t per_item = state_item_bytes(pl, dp);
    // lane = point when that fills the waves better and the abscissa loop has its own kernel
    // (decided on the size of the whole call, npts_call, so that the chunking of a long list cannot change a bit)
    const bool lanes = batch_layout() == 3 && per_point && !chunked && per_item != 0 && !pl->force_layout0 &&
                       (npts_call >= 0 ? npts_call : npts) >= 4 * UCF_WAVE && (!d_params || ppp >= UCF_WAVE / 2);
    const int items_per_pt = chunked ? (pl->D.np + UCF_WAVE - 1) / UCF_WAVE : 1;
    // points per launch: bounded by the integration-state budget; a grid (per_point = 0) is cut at whole time rows,
    // a parameter batch in the lane = point layout at whole plans
    size_t step = (size_t)npts;
    size_t items = 0;
    if (lanes) {
        const size_t unit = d_params ? (size_t)ppp : UCF_WAVE;                                   // points that go together
        const size_t unit_items = (size_t)((unit + UCF_WAVE - 1) / UCF_WAVE) * pl->D.np;          // their work items
        size_t nunits = state_budget() / (per_item * unit_items);
        if (nunits < 1) nunits = 1;
        step = nunits * unit;
        if (step > (size_t)npts) step = npts;
        items = ((step + unit - 1) / unit) * unit_items;
    } else if (per_item) {
        step = state_budget() / (per_item * items_per_pt);
        if (!per_point) step = (step / nr) * nr;
        if (step < (size_t)(per_point ? 1 : nr)) step = per_point ? 1 : nr;
        if (step > (size_t)npts) step = npts;
        items = step * items_per_pt;
    }
    if (per_item) {
        rc = ensure_state(pl, dp, items);
        if (rc) return rc;
    }
    if (chunked || lanes) {
        const size_t need = step * dp.nz * pl->D.np * 2 * sizeof(double);
        if (pl->totlap_bytes < need) {
            if (pl->d_totlap) { (void)hipDeviceSynchronize(); (void)hipFree(pl->d_totlap); pl->d_totlap = nullptr; pl->totlap_bytes = 0; }
            if (hipMalloc((void**)&pl->d_totlap, need) != hipSuccess) return fail(UCF_ERR_NOMEM, "hipMalloc of %zu transform-workspace bytes failed", need);
            pl->totlap_bytes = need;
        }
    }
    const size_t nabs = (size_t)pl->D.nabs;
    for (size_t base = 0; base < (size_t)npts; base += step) {
        const int n = (int)(((size_t)npts - base < step) ? (size_t)npts - base : step);
        // per_point: everything is indexed by the point; grid: times (and their split indices) by the row
        const size_t tb = per_point ? base : base / nr;
        const double* tD = d_tD + tb;
        const int* sv = d_sv + tb;
        const double* rD = per_point ? d_rD + base : d_rD;
        const double* tab = per_point ? pl->d_work + base * nabs * 2 : pl->d_work;
        double* h = d_h + base * dp.nz_out;
        double* dh = d_dh + base * dp.nz_out;
        if (lanes) {
            const int pp = d_params ? ppp : n;
            rc = (pl->mode == 1)
                     ? ucf_fast::launch_points_lanes(dp, n, pp, tD, rD, sv, tab, pl->d_totlap, h, dh, d_stats, stream, pl->d_state, pl->d_ndone, d_params, (int)(p0 + base))
                     : ucf_faithful::launch_points_lanes(dp, n, pp, tD, rD, sv, tab, pl->d_totlap, h, dh, d_stats, stream, pl->d_state, pl->d_ndone, nullptr, 0);
        } else if (chunked) {
            rc = (pl->mode == 1)
                     ? ucf_fast::launch_points_chunked(dp, n, per_point, nr, nsv, svmin, tD, rD, sv, tab, pl->d_totlap, h, dh, d_stats, stream, pl->d_glscr, pl->d_state, pl->d_ndone, d_params, ppp, (int)(p0 + base))
                     : ucf_faithful::launch_points_chunked(dp, n, per_point, nr, nsv, svmin, tD, rD, sv, tab, pl->d_totlap, h, dh, d_stats, stream, pl->d_glscr, pl->d_state, pl->d_ndone);
        } else {
            rc = (pl->mode == 1)
                     ? ucf_fast::launch_points(dp, n, per_point, nr, nsv, svmin, tD, rD, sv, tab, h, dh, d_stats, stream, pl->d_glscr, pl->d_state, pl->d_ndone, d_params, ppp, (int)(p0 + base))
                     : ucf_faithful::launch_points(dp, n, per_point, nr, nsv, svmin, tD, rD, sv, tab, h, dh, d_stats, stream, pl->d_glscr, pl->d_state, pl->d_ndone);
        }
        if (rc == UCF_ERR_UNSUPPORTED) return fail(rc, "this model / size combination has no kernel (model %d, LDS limit)", dp.model);
        if (rc) return fail(rc, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    return UCF_OK;
}

}  // namespace

namespace {
// arbitrary points with the parameter block dp: one abscissa-table row per point, in chunks that keep the table within
// UCF_TABLE_BYTES
int batch_points(ucf_plan* pl, const ucf_dev_params& dp, int npts, const double* d_tD, const double* d_rD, const int* d_sv,
                 double* d_h, double* d_dh, ucf_stats* d_stats, void* stream)
{
    const size_t row_bytes = (size_t)pl->D.nabs * 2 * sizeof(double);
    int chunk = (int)(table_budget() / row_bytes);
    if (chunk < 1) chunk = 1;
    if (chunk > npts) chunk = npts;
    int rc = ensure_work(pl, (size_t)chunk * row_bytes);
    if (rc) return rc;
    for (int base = 0; base < npts; base += chunk) {
        const int n = (npts - base < chunk) ? npts - base : chunk;
        rc = ucf_faithful::launch_abscissae(dp, n, 1, 1, 0, d_rD + base, d_sv + base, pl->d_work, stream);
        if (rc) return fail(rc, "abscissa kernel launch failed");
        rc = launch_points_any(pl, dp, n, 1, 1, 1, 0, d_tD + base, d_rD + base, d_sv + base, d_h + (size_t)base * dp.nz_out,
                               d_dh + (size_t)base * dp.nz_out, d_stats, stream, nullptr, 1, 0, npts);
        if (rc) return rc;
    }
    return UCF_OK;
}
}  // namespace

namespace {
int grid_device_chunk(ucf_plan* pl, int nt, const double* d_tD, const int* d_sv, int nr, const double* d_rD,
                      int nz, const double* zD, const int* zLay, int nz_out, int z_off, double* d_h, double* d_dh,
                      ucf_stats* d_stats, void* stream);
int batch_device_chunk(ucf_plan* pl, int npts, const double* d_tD, const double* d_rD, const int* d_sv,
                       int nz, const double* zD, const int* zLay, int nz_out, int z_off, double* d_h, double* d_dh,
                       ucf_stats* d_stats, void* stream);
int batch_device_impl(ucf_plan* pl, int npts, const double* d_tD, const double* d_rD, const int* d_sv,
                      int nz, const double* zD, const int* zLay, double* d_h, double* d_dh, ucf_stats* d_stats, void* stream,
                      bool presorted);
}

int ucf_drawdown_grid_device(ucf_plan* pl, int nt, const double* d_tD, const int* d_sv, int nr, const double* d_rD,
                             int nz, const double* zD, const int* zLay, double* d_h, double* d_dh,
                             ucf_stats* d_stats, void* stream)
{
    if (!pl) return fail(UCF_ERR_BAD_ARGUMENT, "NULL plan");
    if (nt < 0 || nr < 0) return fail(UCF_ERR_BAD_ARGUMENT, "negative grid size");
    if (nz < 1) return fail(UCF_ERR_BAD_ARGUMENT, "nz < 1");
    if (nt == 0 || nr == 0) return UCF_OK;
    if ((long long)nt * nr > 0x7fffffffLL) return fail(UCF_ERR_BAD_ARGUMENT, "grid larger than 2^31-1 points: split it");
    if (!d_tD || !d_rD || !d_sv || !d_h || !d_dh || !zD || !zLay) return fail(UCF_ERR_BAD_ARGUMENT, "NULL array");
    {
        // Short time vectors (or several split indices): neither lane = time nor lane = Laplace sample fills the waves.
        // With enough points the grid is expanded into the point list it stands for and runs lane = point, in order of
        // radius like every long list (the outputs of a grid are in point order already: point = it * nr + ir).
        const int* j0s = pl->P.j0s;
        const int nsv = (j0s[0] > j0s[1] ? j0s[0] - j0s[1] : j0s[1] - j0s[0]) + 1;
        const int ntiles_ = (nt + UCF_WAVE - 1) / UCF_WAVE;
        const bool lane_time = nsv == 1 && (double)nt / (64.0 * ntiles_) > (double)pl->D.np / (64.0 * ((pl->D.np + 63) / 64)) && !pl->force_layout0;
        ucf_dev_params one = pl->dev;
        one.nz = 1;
        const long long np_grid = (long long)nt * nr;
        if (!lane_time && batch_layout() == 3 && np_grid >= 4 * UCF_WAVE && pl->D.np <= UCF_WAVE && state_item_bytes(pl, one) != 0 &&
            !pl->force_layout0) {
            if (pl->expand_points < (size_t)np_grid) {
                if (pl->d_expand) { (void)hipDeviceSynchronize(); (void)hipFree(pl->d_expand); pl->d_expand = nullptr; pl->expand_points = 0; }
                if (hipMalloc((void**)&pl->d_expand, (size_t)np_grid * (2 * sizeof(double) + sizeof(int))) != hipSuccess)
                    return fail(UCF_ERR_NOMEM, "hipMalloc of the expanded grid (%lld points) failed", np_grid);
                pl->expand_points = (size_t)np_grid;
            }
            double* e_tD = pl->d_expand;
            double* e_rD = e_tD + np_grid;
            int* e_sv = (int*)(e_rD + np_grid);
            int rc = ucf_faithful::launch_expand_grid(nt, nr, d_tD, d_sv, d_rD, e_tD, e_rD, e_sv, stream);
            if (rc) return fail(rc, "grid expansion kernel launch failed");
            return batch_device_impl(pl, (int)np_grid, e_tD, e_rD, e_sv, nz, zD, zLay, d_h, d_dh, d_stats, stream, false);
        }
    }
    // depths in chunks that fit the wave's LDS budget; each chunk is its own launch sequence on the stream
    const int zc = z_chunk(pl);
    for (int z0 = 0; z0 < nz; z0 += zc) {
        const int n = (nz - z0 < zc) ? nz - z0 : zc;
        int rc = grid_device_chunk(pl, nt, d_tD, d_sv, nr, d_rD, n, zD + z0, zLay + z0, nz, z0, d_h, d_dh, d_stats, stream);
        if (rc) return rc;
    }
    return UCF_OK;
}

namespace {
int grid_device_chunk(ucf_plan* pl, int nt, const double* d_tD, const int* d_sv, int nr, const double* d_rD,
                      int nz, const double* zD, const int* zLay, int nz_out, int z_off, double* d_h, double* d_dh,
                      ucf_stats* d_stats, void* stream)
{
    ucf_dev_params dp;
    int rc = fill_call_params(pl, nz, zD, zLay, dp, nz_out, z_off);
    if (rc) return rc;
    const int* j0s = pl->P.j0s;
    const int svmin = j0s[0] < j0s[1] ? j0s[0] : j0s[1];
    const int nsv = (j0s[0] > j0s[1] ? j0s[0] - j0s[1] : j0s[1] - j0s[0]) + 1;     // driver_io.f90:660-664: sv in [min,max]
    const size_t nabs = (size_t)pl->D.nabs;
    const int ntiles_ = (nt + UCF_WAVE - 1) / UCF_WAVE;
    const bool lane_time = nsv == 1 && (double)nt / (64.0 * ntiles_) > (double)pl->D.np / (64.0 * ((pl->D.np + 63) / 64)) && !pl->force_layout0;
    (void)lane_time;
    rc = ensure_work(pl, (size_t)nr * nsv * nabs * 2 * sizeof(double));
    if (rc) return rc;
    rc = ucf_faithful::launch_abscissae(dp, nr * nsv, 0, nsv, svmin, d_rD, d_sv, pl->d_work, stream);
    if (rc) return fail(rc, "abscissa kernel launch failed");
    // lane layout: lane = time (all 64 lanes live, needs one split index for all times) when that fills the
    // wave better than lane = Laplace sample (2M+1 of 64 lanes)
    const int ntiles = (nt + UCF_WAVE - 1) / UCF_WAVE;
    const double fill_time = (double)nt / (64.0 * ntiles);
    const double fill_lap = (double)pl->D.np / (64.0 * ((pl->D.np + 63) / 64));
    if (nsv == 1 && fill_time > fill_lap && !pl->force_layout0) {
        const size_t per_radius = (size_t)nt * nz * pl->D.np * 2 * sizeof(double);
        int nrc = (int)(((size_t)1 << 30) / per_radius);           // <= 1 GiB of workspace per chunk of radii
        const size_t state_per_radius = state_item_bytes(pl, dp) * ntiles * pl->D.np;
        if (state_per_radius && (size_t)nrc > state_budget() / state_per_radius) nrc = (int)(state_budget() / state_per_radius);
        if (nrc < 1) nrc = 1;
        if (nrc > nr) nrc = nr;
        rc = ensure_state(pl, dp, (size_t)nrc * ntiles * pl->D.np);
        if (rc) return rc;
        if (pl->totlap_bytes < per_radius * nrc) {
            if (pl->d_totlap) { (void)hipDeviceSynchronize(); (void)hipFree(pl->d_totlap); pl->d_totlap = nullptr; pl->totlap_bytes = 0; }
            if (hipMalloc((void**)&pl->d_totlap, per_radius * nrc) != hipSuccess)
                return fail(UCF_ERR_NOMEM, "hipMalloc of %zu transform-workspace bytes failed", per_radius * nrc);
            pl->totlap_bytes = per_radius * nrc;
        }
        rc = ensure_glscr(pl, nz);
        if (rc) return rc;
        // timing brackets the transform kernel of a single-chunk call (the dominant kernel of the path)
        void* tev0 = nullptr; void* tev1 = nullptr;
        pl->ev_valid = 0;
        if (pl->timing && nrc == nr) {
            if (!pl->ev0) { hipEvent_t a, b; if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { pl->ev0 = a; pl->ev1 = b; } }
            tev0 = pl->ev0; tev1 = pl->ev1;
            pl->ev_valid = (tev0 && tev1);
            pl->last_kernel = (pl->mode == 1) ? "ucf_fast::integrate[_generic]_kernel<FAMILY, 1>"
                                              : (state_item_bytes(pl, dp) ? "ucf_faithful::integrate_generic_kernel<FAMILY, 1>" : "ucf_faithful::point_kernel<FAMILY, 1>");
        }
        for (int ir0 = 0; ir0 < nr; ir0 += nrc) {
            const int n = (nr - ir0 < nrc) ? nr - ir0 : nrc;
            rc = (pl->mode == 1)
                     ? ucf_fast::launch_grid_transposed(dp, nt, nr, ir0, n, svmin, d_tD, d_rD, pl->d_work, pl->d_totlap, d_h, d_dh, d_stats, stream, tev0, tev1, pl->d_glscr, pl->d_state, pl->d_ndone)
                     : ucf_faithful::launch_grid_transposed(dp, nt, nr, ir0, n, svmin, d_tD, d_rD, pl->d_work, pl->d_totlap, d_h, d_dh, d_stats, stream, tev0, tev1, pl->d_glscr, pl->d_state, pl->d_ndone);
            if (rc == UCF_ERR_UNSUPPORTED) return fail(rc, "this model / size combination has no kernel (model %d, LDS limit)", dp.model);
            if (rc) return fail(rc, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
        }
        return UCF_OK;
    }
    return launch_points_any(pl, dp, nt * nr, 0, nr, nsv, svmin, d_tD, d_rD, d_sv, d_h, d_dh, d_stats, stream);
}
}  // namespace

namespace {
__global__ void iota_kernel(int n, int* v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
__global__ void gather_points_kernel(int n, const int* __restrict__ perm, const double* __restrict__ tD, const int* __restrict__ sv,
                                     double* __restrict__ tDs, int* __restrict__ svs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { tDs[i] = tD[perm[i]]; svs[i] = sv[perm[i]]; }
}
__global__ void scatter_results_kernel(int n, int nz, const int* __restrict__ perm, const double* __restrict__ hs,
                                       const double* __restrict__ dhs, double* __restrict__ h, double* __restrict__ dh)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)n * nz) return;
    const int i = (int)(e / nz), z = (int)(e % nz);
    h[(size_t)perm[i] * nz + z] = hs[e];
    dh[(size_t)perm[i] * nz + z] = dhs[e];
}

int batch_device_all_depths(ucf_plan* pl, int npts, const double* d_tD, const double* d_rD, const int* d_sv,
                            int nz, const double* zD, const int* zLay, double* d_h, double* d_dh, ucf_stats* d_stats, void* stream)
{
    const int zc = z_chunk(pl);
    for (int z0 = 0; z0 < nz; z0 += zc) {
        const int n = (nz - z0 < zc) ? nz - z0 : zc;
        int rc = batch_device_chunk(pl, npts, d_tD, d_rD, d_sv, n, zD + z0, zLay + z0, nz, z0, d_h, d_dh, d_stats, stream);
        if (rc) return rc;
    }
    return UCF_OK;
}

// presorted: the caller (ucf_drawdown_batch) already put the points in order of radius
int batch_device_impl(ucf_plan* pl, int npts, const double* d_tD, const double* d_rD, const int* d_sv,
                      int nz, const double* zD, const int* zLay, double* d_h, double* d_dh, ucf_stats* d_stats, void* stream,
                      bool presorted)
{
    if (presorted || npts < 4 * UCF_WAVE || batch_layout() != 3)
        return batch_device_all_depths(pl, npts, d_tD, d_rD, d_sv, nz, zD, zLay, d_h, d_dh, d_stats, stream);
    // lane = point wants the 64 points of a wave to be neighbours in radius (see ucf_drawdown_batch): sort by radius on
    // the device (rocPRIM radix sort of (rD, index)), evaluate, scatter the results back to the caller's order
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)npts;
    size_t temp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, temp_bytes, d_rD, (double*)nullptr, (int*)nullptr, (int*)nullptr, n, 0, 64, s) != hipSuccess)
        return fail(UCF_ERR_HIP, "rocprim::radix_sort_pairs (size query) failed");
    // one allocation: keys_out | tD_s | h_s | dh_s | idx_in | idx_out | sv_s | sort temp
    const size_t off_keys = 0, off_t = off_keys + n * 8, off_h = off_t + n * 8, off_d = off_h + n * nz * 8, off_i0 = off_d + n * nz * 8,
                 off_i1 = off_i0 + n * 4, off_sv = off_i1 + n * 4, off_tmp = (off_sv + n * 4 + 255) / 256 * 256, total = off_tmp + temp_bytes;
    if (pl->sort_bytes < total) {
        if (pl->d_sort) { (void)hipDeviceSynchronize(); (void)hipFree(pl->d_sort); pl->d_sort = nullptr; pl->sort_bytes = 0; }
        if (hipMalloc((void**)&pl->d_sort, total) != hipSuccess) return fail(UCF_ERR_NOMEM, "hipMalloc of %zu sort-workspace bytes failed", total);
        pl->sort_bytes = total;
    }
    char* base = (char*)pl->d_sort;
    double* keys = (double*)(base + off_keys);
    double* tDs = (double*)(base + off_t);
    double* hs = (double*)(base + off_h);
    double* dhs = (double*)(base + off_d);
    int* i0 = (int*)(base + off_i0);
    int* i1 = (int*)(base + off_i1);
    int* svs = (int*)(base + off_sv);
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, s, npts, i0);
    if (rocprim::radix_sort_pairs(base + off_tmp, temp_bytes, d_rD, keys, i0, i1, n, 0, 64, s) != hipSuccess)
        return fail(UCF_ERR_HIP, "rocprim::radix_sort_pairs failed");
    hipLaunchKernelGGL(gather_points_kernel, dim3(nb), dim3(256), 0, s, npts, i1, d_tD, d_sv, tDs, svs);
    int rc = batch_device_all_depths(pl, npts, tDs, keys, svs, nz, zD, zLay, hs, dhs, d_stats, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(scatter_results_kernel, dim3((unsigned)((n * nz + 255) / 256)), dim3(256), 0, s, npts, nz, i1, hs, dhs, d_h, d_dh);
    return hipGetLastError() == hipSuccess ? UCF_OK : fail(UCF_ERR_HIP, "sort helper kernels failed");
}
}  // namespace

int ucf_drawdown_batch_device(ucf_plan* pl, int npts, const double* d_tD, const double* d_rD, const int* d_sv,
                              int nz, const double* zD, const int* zLay, double* d_h, double* d_dh,
                              ucf_stats* d_stats, void* stream)
{
    if (!pl) return fail(UCF_ERR_BAD_ARGUMENT, "NULL plan");
    if (npts < 0) return fail(UCF_ERR_BAD_ARGUMENT, "npts < 0");
    if (nz < 1) return fail(UCF_ERR_BAD_ARGUMENT, "nz < 1");
    if (npts == 0) return UCF_OK;
    if (!d_tD || !d_rD || !d_sv || !d_h || !d_dh || !zD || !zLay) return fail(UCF_ERR_BAD_ARGUMENT, "NULL array");
    return batch_device_impl(pl, npts, d_tD, d_rD, d_sv, nz, zD, zLay, d_h, d_dh, d_stats, stream, false);
}

namespace {
int batch_device_chunk(ucf_plan* pl, int npts, const double* d_tD, const double* d_rD, const int* d_sv,
                       int nz, const double* zD, const int* zLay, int nz_out, int z_off, double* d_h, double* d_dh,
                       ucf_stats* d_stats, void* stream)
{
    ucf_dev_params dp;
    int rc = fill_call_params(pl, nz, zD, zLay, dp, nz_out, z_off);
    if (rc) return rc;
    return batch_points(pl, dp, npts, d_tD, d_rD, d_sv, d_h, d_dh, d_stats, stream);
}
}  // namespace

namespace {
int check_sv(const ucf_plan* pl, int n, const int* sv)
{
    for (int i = 0; i < n; i++)
        if (sv[i] < 1 || sv[i] + pl->P.nacc > pl->D.nj0z)
            return fail(UCF_ERR_BAD_ARGUMENT, "sv[%d]=%d outside 1..%d", i, sv[i], pl->D.nj0z - pl->P.nacc);
    return UCF_OK;
}
}  // namespace

int ucf_drawdown_batch(ucf_plan* pl, int npts, const double* tD, const double* rD, const int* sv,
                       int nz, const double* zD, const int* zLay, double* h, double* dh, ucf_stats* stats)
{
    if (!pl) return fail(UCF_ERR_BAD_ARGUMENT, "NULL plan");
    if (npts < 0) return fail(UCF_ERR_BAD_ARGUMENT, "npts < 0");
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (npts == 0) return UCF_OK;
    if (!tD || !rD || !sv || !h || !dh) return fail(UCF_ERR_BAD_ARGUMENT, "NULL array");
    int rc = check_sv(pl, npts, sv);
    if (rc) return rc;
    dev_buf b_t, b_r, b_s, b_h, b_d, b_st;
    const size_t nb = sizeof(double) * (size_t)npts;
    if (b_t.alloc(nb) || b_r.alloc(nb) || b_s.alloc(sizeof(int) * (size_t)npts) || b_h.alloc(nb * nz) ||
        b_d.alloc(nb * nz) || b_st.alloc(sizeof(ucf_stats)))
        return fail(UCF_ERR_NOMEM, "device allocation failed for %d points", npts);
    // A long list runs with lane = point: the 64 points of a wave should be neighbours in radius, because the wave
    // leaves the fast evaluators at the first lane that must (small radii reach the overflow regime early).  The list
    // is evaluated in order of radius and the results are put back in the caller's order.
    std::vector<int> perm;
    std::vector<double> tS, rS, hS, dS;
    std::vector<int> sS;
    const bool sorted = npts >= 4 * UCF_WAVE && batch_layout() == 3;
    if (sorted) {
        perm.resize(npts);
        for (int i = 0; i < npts; i++) perm[i] = i;
        std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return rD[x] < rD[y]; });
        tS.resize(npts); rS.resize(npts); sS.resize(npts);
        for (int i = 0; i < npts; i++) { tS[i] = tD[perm[i]]; rS[i] = rD[perm[i]]; sS[i] = sv[perm[i]]; }
        tD = tS.data(); rD = rS.data(); sv = sS.data();
    }
    HIP_TRY(hipMemcpy(b_t.p, tD, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b_r.p, rD, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b_s.p, sv, sizeof(int) * (size_t)npts, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(b_st.p, 0, sizeof(ucf_stats)));
    if (nz < 1 || !zD || !zLay) return fail(UCF_ERR_BAD_ARGUMENT, "bad depths");
    rc = batch_device_impl(pl, npts, (const double*)b_t.p, (const double*)b_r.p, (const int*)b_s.p, nz, zD,
                           zLay, (double*)b_h.p, (double*)b_d.p, stats ? (ucf_stats*)b_st.p : nullptr, nullptr, true);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (sorted) {
        hS.resize((size_t)npts * nz); dS.resize((size_t)npts * nz);
        HIP_TRY(hipMemcpy(hS.data(), b_h.p, nb * nz, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(dS.data(), b_d.p, nb * nz, hipMemcpyDeviceToHost));
        for (int i = 0; i < npts; i++)
            for (int z = 0; z < nz; z++) {
                h[(size_t)perm[i] * nz + z] = hS[(size_t)i * nz + z];
                dh[(size_t)perm[i] * nz + z] = dS[(size_t)i * nz + z];
            }
    } else {
        HIP_TRY(hipMemcpy(h, b_h.p, nb * nz, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(dh, b_d.p, nb * nz, hipMemcpyDeviceToHost));
    }
    if (stats) HIP_TRY(hipMemcpy(stats, b_st.p, sizeof(ucf_stats), hipMemcpyDeviceToHost));
    return UCF_OK;
}

int ucf_drawdown_grid(ucf_plan* pl, int nt, const double* tD, const int* sv, int nr, const double* rD,
                      int nz, const double* zD, const int* zLay, double* h, double* dh, ucf_stats* stats)
{
    if (!pl) return fail(UCF_ERR_BAD_ARGUMENT, "NULL plan");
    if (nt < 0 || nr < 0) return fail(UCF_ERR_BAD_ARGUMENT, "negative grid size");
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (nt == 0 || nr == 0) return UCF_OK;
    if (!tD || !rD || !sv || !h || !dh) return fail(UCF_ERR_BAD_ARGUMENT, "NULL array");
    int rc = check_sv(pl, nt, sv);
    if (rc) return rc;
    const int* j0s = pl->P.j0s;
    const int svmin = j0s[0] < j0s[1] ? j0s[0] : j0s[1], svmax = j0s[0] > j0s[1] ? j0s[0] : j0s[1];
    for (int i = 0; i < nt; i++)
        if (sv[i] < svmin || sv[i] > svmax) return fail(UCF_ERR_BAD_ARGUMENT, "sv[%d]=%d outside the plan's split range %d..%d", i, sv[i], svmin, svmax);
    dev_buf b_t, b_r, b_s, b_h, b_d, b_st;
    const size_t no = sizeof(double) * (size_t)nt * nr * nz;
    if (b_t.alloc(sizeof(double) * nt) || b_r.alloc(sizeof(double) * nr) || b_s.alloc(sizeof(int) * nt) || b_h.alloc(no) ||
        b_d.alloc(no) || b_st.alloc(sizeof(ucf_stats)))
        return fail(UCF_ERR_NOMEM, "device allocation failed for a %d x %d grid", nt, nr);
    HIP_TRY(hipMemcpy(b_t.p, tD, sizeof(double) * nt, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b_r.p, rD, sizeof(double) * nr, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b_s.p, sv, sizeof(int) * nt, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(b_st.p, 0, sizeof(ucf_stats)));
    rc = ucf_drawdown_grid_device(pl, nt, (const double*)b_t.p, (const int*)b_s.p, nr, (const double*)b_r.p, nz, zD, zLay,
                                  (double*)b_h.p, (double*)b_d.p, stats ? (ucf_stats*)b_st.p : nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h, b_h.p, no, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dh, b_d.p, no, hipMemcpyDeviceToHost));
    if (stats) HIP_TRY(hipMemcpy(stats, b_st.p, sizeof(ucf_stats), hipMemcpyDeviceToHost));
    return UCF_OK;
}

int ucf_drawdown_multi(ucf_plan* const* plans, int nplans, int npts, const double* t, const double* r,
                       int nz, const double* z, int dimensionless, double* h, double* dh)
{
    if (!plans || nplans < 1) return fail(UCF_ERR_BAD_ARGUMENT, "no plans");
    if (npts < 0 || nz < 1) return fail(UCF_ERR_BAD_ARGUMENT, "bad sizes");
    if (npts == 0) return UCF_OK;
    if (!t || !r || !z || !h || !dh) return fail(UCF_ERR_BAD_ARGUMENT, "NULL array");
    for (int k = 0; k < nplans; k++) if (!plans[k]) return fail(UCF_ERR_BAD_ARGUMENT, "plans[%d] is NULL", k);
    const size_t np_ = (size_t)npts, tot = (size_t)nplans * np_;
    // the observation points are evaluated in order of radius (see ucf_drawdown_batch) and put back at the end
    std::vector<int> perm(npts);
    for (int i = 0; i < npts; i++) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return r[x] < r[y]; });
    std::vector<double> t_s(npts), r_s(npts);
    for (int i = 0; i < npts; i++) { t_s[i] = t[perm[i]]; r_s[i] = r[perm[i]]; }
    double* const h_user = h;
    double* const dh_user = dh;
    std::vector<double> h_tmp(tot * nz), dh_tmp(tot * nz);
    t = t_s.data(); r = r_s.data(); h = h_tmp.data(); dh = dh_tmp.data();
    struct unsort_at_exit {
        const std::vector<int>& perm; size_t np_, tot; int nplans, nz; const double* hs; const double* ds; double* h; double* dh; bool armed;
        ~unsort_at_exit() {
            if (!armed) return;
            for (int k = 0; k < nplans; k++)
                for (size_t i = 0; i < np_; i++)
                    for (int z = 0; z < nz; z++) {
                        h[(k * np_ + perm[i]) * nz + z] = hs[(k * np_ + i) * nz + z];
                        dh[(k * np_ + perm[i]) * nz + z] = ds[(k * np_ + i) * nz + z];
                    }
        }
    } unsort{perm, np_, tot, nplans, nz, h_tmp.data(), dh_tmp.data(), h_user, dh_user, true};
    // host staging: per plan tD, rD, sv
    std::vector<double> tD(tot), rD(tot), zD((size_t)nplans * nz);
    std::vector<int> sv(tot), zl((size_t)nplans * nz);
    for (int k = 0; k < nplans; k++) {
        const ucf_derived& D = plans[k]->D;
        for (int i = 0; i < npts; i++) { tD[k * np_ + i] = t[i] / D.Tc; rD[k * np_ + i] = r[i] / D.Lc; }
        for (int j = 0; j < nz; j++) zD[(size_t)k * nz + j] = z[j] / D.Lc;
        int rc = ucf_zlay(plans[k], nz, &zD[(size_t)k * nz], &zl[(size_t)k * nz]);
        if (rc) return rc;
        rc = ucf_split_vector(plans[k], npts, &tD[k * np_], &sv[k * np_]);
        if (rc) return rc;
        rc = check_sv(plans[k], npts, &sv[k * np_]);
        if (rc) return rc;
    }
    // Plans that differ only in what the evaluators read (hydraulic / geometric parameters) share one launch
    // sequence: work item = (plan, point), parameter block per plan in device memory.  Anything that shapes the
    // work must agree; otherwise every plan gets its own launches on a pool of streams.
    bool one_launch = (nplans > 1) && (tot <= 0x7fffffffULL);
    for (int k = 0; k < nplans && one_launch; k++) {
        const ucf_plan* a = plans[0];
        const ucf_plan* b = plans[k];
        const ucf_dev_params &x = a->dev, &y = b->dev;
        one_launch = a->mode == 1 && b->mode == 1 && a->device == b->device && a->force_layout0 == b->force_layout0 &&
                     (x.model == 1 || x.model == 3 || x.model == 4 || x.model == 5 || (x.model == 6 && x.MNtype == 2)) &&   // integrate_kernel models
                     x.model == y.model && x.MNtype == y.MNtype && x.order == y.order && x.MoenchM == y.MoenchM &&
                     x.M == y.M && x.k == y.k && x.R == y.R && x.nacc == y.nacc && x.ngl == y.ngl && x.N == y.N &&
                     x.nj0z == y.nj0z && x.alpha == y.alpha && x.logtol == y.logtol &&
                     a->P.j0s[0] == b->P.j0s[0] && a->P.j0s[1] == b->P.j0s[1] && (x.timeType >= 0) == (y.timeType >= 0) &&
                     (x.timeType >= 0 || x.timeType == y.timeType);
    }
    dev_buf b_t, b_r, b_s, b_h, b_d;
    if (b_t.alloc(sizeof(double) * tot) || b_r.alloc(sizeof(double) * tot) || b_s.alloc(sizeof(int) * tot) ||
        b_h.alloc(sizeof(double) * tot * nz) || b_d.alloc(sizeof(double) * tot * nz))
        return fail(UCF_ERR_NOMEM, "device allocation failed for %d plans x %d points", nplans, npts);
    HIP_TRY(hipMemcpy(b_t.p, tD.data(), sizeof(double) * tot, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b_r.p, rD.data(), sizeof(double) * tot, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b_s.p, sv.data(), sizeof(int) * tot, hipMemcpyHostToDevice));
    if (one_launch) {
        ucf_plan* pl = plans[0];
        std::vector<ucf_dev_params> dps(nplans);
        dev_buf b_p;
        if (b_p.alloc(sizeof(ucf_dev_params) * nplans)) return fail(UCF_ERR_NOMEM, "device allocation failed for %d parameter blocks", nplans);
        // one abscissa row per (plan, point), in chunks that keep the table <= 256 MiB
        const size_t row_bytes = (size_t)pl->D.nabs * 2 * sizeof(double);
        size_t chunk_plans = (table_budget() / row_bytes) / np_;      // whole plans per table chunk
        if (chunk_plans < 1) chunk_plans = 1;
        size_t chunk = chunk_plans * np_;
        if (chunk > tot) chunk = tot;
        int rc = ensure_work(pl, chunk * row_bytes);
        if (rc) return rc;
        // depths in the same chunks as every other entry point (LDS budget of the integrate kernels)
        const int zc = z_chunk(pl);
        for (int z0 = 0; z0 < nz && rc == UCF_OK; z0 += zc) {
            const int nzc = (nz - z0 < zc) ? nz - z0 : zc;
            for (int k = 0; k < nplans; k++) {
                rc = fill_call_params(plans[k], nzc, &zD[(size_t)k * nz + z0], &zl[(size_t)k * nz + z0], dps[k], nz, z0);
                if (rc) return rc;
            }
            for (int k = 1; k < nplans; k++) dps[0].any_lay3 |= dps[k].any_lay3;      // plan 0's block drives the launch
            HIP_TRY(hipDeviceSynchronize());                              // the previous chunk still reads the parameter blocks
            HIP_TRY(hipMemcpy(b_p.p, dps.data(), sizeof(ucf_dev_params) * nplans, hipMemcpyHostToDevice));
            for (size_t base = 0; base < tot && rc == UCF_OK; base += chunk) {
                const int n = (int)((tot - base < chunk) ? tot - base : chunk);
                rc = ucf_faithful::launch_abscissae(dps[0], n, 1, 1, 0, (const double*)b_r.p + base, (const int*)b_s.p + base, pl->d_work, nullptr);
                if (rc) return fail(rc, "abscissa kernel launch failed");
                // points [base, base + n) of the flattened (plan, point) index; plan of point q = q / npts
                rc = launch_points_any(pl, dps[0], n, 1, 1, 1, 0, (const double*)b_t.p + base, (const double*)b_r.p + base,
                                       (const int*)b_s.p + base, (double*)b_h.p + base * nz, (double*)b_d.p + base * nz, nullptr, nullptr,
                                       (const ucf_dev_params*)b_p.p, npts, base, (int)tot);
                if (rc) return rc;
                if (base + chunk < tot) HIP_TRY(hipDeviceSynchronize());      // the next chunk rewrites the table
            }
        }
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h, b_h.p, sizeof(double) * tot * nz, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(dh, b_d.p, sizeof(double) * tot * nz, hipMemcpyDeviceToHost));
        if (!dimensionless)
            for (int k = 0; k < nplans; k++) {
                const double Hc = plans[k]->D.Hc;
                for (size_t i = 0; i < np_ * nz; i++) { h[k * np_ * nz + i] *= Hc; dh[k * np_ * nz + i] *= Hc; }
            }
        return UCF_OK;
    }
    const int NS = 8;
    hipStream_t streams[NS];
    int ns = 0;
    for (; ns < NS && ns < nplans; ns++)
        if (hipStreamCreateWithFlags(&streams[ns], hipStreamNonBlocking) != hipSuccess) break;
    if (ns == 0) return fail(UCF_ERR_HIP, "cannot create a HIP stream");
    int rc = UCF_OK;
    for (int k = 0; k < nplans && rc == UCF_OK; k++) {
        rc = ucf_drawdown_batch_device(plans[k], npts, (const double*)b_t.p + k * np_, (const double*)b_r.p + k * np_,
                                       (const int*)b_s.p + k * np_, nz, &zD[(size_t)k * nz], &zl[(size_t)k * nz],
                                       (double*)b_h.p + k * np_ * nz, (double*)b_d.p + k * np_ * nz, nullptr, streams[k % ns]);
    }
    for (int i = 0; i < ns; i++) { (void)hipStreamSynchronize(streams[i]); (void)hipStreamDestroy(streams[i]); }
    if (rc) return rc;
    HIP_TRY(hipMemcpy(h, b_h.p, sizeof(double) * tot * nz, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dh, b_d.p, sizeof(double) * tot * nz, hipMemcpyDeviceToHost));
    if (!dimensionless)
        for (int k = 0; k < nplans; k++) {
            const double Hc = plans[k]->D.Hc;
            for (size_t i = 0; i < np_ * nz; i++) { h[k * np_ * nz + i] *= Hc; dh[k * np_ * nz + i] *= Hc; }
        }
    return UCF_OK;
}

// ---- stage hooks
int ucf_eval_samples(ucf_plan* pl, int n_a, const double* a, double rD, int np, const double* p_re_im,
                     int nz, const double* zD, const int* zLay, double* fp_re_im)
{
    if (!pl || !a || !p_re_im || !fp_re_im || n_a < 1) return fail(UCF_ERR_BAD_ARGUMENT, "bad sample request");
    if (np != pl->D.np) return fail(UCF_ERR_BAD_ARGUMENT, "np=%d but the plan has 2M+1=%d", np, pl->D.np);
    ucf_dev_params dp;
    int rc = fill_call_params(pl, nz, zD, zLay, dp);
    if (rc) return rc;
    dev_buf b_a, b_p, b_f;
    const size_t nf = sizeof(double) * 2 * (size_t)n_a * nz * np;
    if (b_a.alloc(sizeof(double) * n_a) || b_p.alloc(sizeof(double) * 2 * np) || b_f.alloc(nf))
        return fail(UCF_ERR_NOMEM, "device allocation failed");
    HIP_TRY(hipMemcpy(b_a.p, a, sizeof(double) * n_a, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b_p.p, p_re_im, sizeof(double) * 2 * np, hipMemcpyHostToDevice));
    rc = (pl->mode == 1) ? ucf_fast::launch_samples(dp, n_a, (const double*)b_a.p, rD, (const double*)b_p.p, (double*)b_f.p, nullptr)
                         : ucf_faithful::launch_samples(dp, n_a, (const double*)b_a.p, rD, (const double*)b_p.p, (double*)b_f.p, nullptr);
    if (rc) return fail(rc, "sample kernel launch failed");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(fp_re_im, b_f.p, nf, hipMemcpyDeviceToHost));
    return UCF_OK;
}

int ucf_pvalues(const ucf_plan* pl, double tee, double* p_re_im)
{
    if (!pl || !p_re_im) return fail(UCF_ERR_BAD_ARGUMENT, "NULL argument");
    const double PI = 4.0 * std::atan(1.0);
    const double sigma = pl->P.alpha - std::log(pl->P.tol) / (2.0 * tee);      // invlap.f90:165
    for (int i = 0; i <= 2 * pl->P.M; i++) {
        p_re_im[2 * i] = sigma;
        p_re_im[2 * i + 1] = PI * i / tee;                                     // :168
    }
    return UCF_OK;
}

int ucf_dehoog(int n, int M, double alpha, double tol, const double* t, const double* tee, const double* fp, double* ft)
{
    if (n < 1 || !t || !tee || !fp || !ft) return fail(UCF_ERR_BAD_ARGUMENT, "bad de Hoog request");
    if (M < 1 || M > UCF_MAX_LAP_M) return fail(UCF_ERR_UNSUPPORTED, "M=%d outside 1..%d", M, UCF_MAX_LAP_M);
    int rc = require_device();
    if (rc) return rc;
    dev_buf b_t, b_e, b_f, b_o;
    const int np = 2 * M + 1;
    if (b_t.alloc(sizeof(double) * n) || b_e.alloc(sizeof(double) * n) || b_f.alloc(sizeof(double) * 2 * (size_t)n * np) ||
        b_o.alloc(sizeof(double) * n))
        return fail(UCF_ERR_NOMEM, "device allocation failed");
    HIP_TRY(hipMemcpy(b_t.p, t, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b_e.p, tee, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b_f.p, fp, sizeof(double) * 2 * (size_t)n * np, hipMemcpyHostToDevice));
    rc = ucf_faithful::launch_dehoog(n, M, alpha, std::log(tol), (const double*)b_t.p, (const double*)b_e.p,
                                     (const double*)b_f.p, (double*)b_o.p, nullptr);
    if (rc) return fail(rc, "de Hoog kernel launch failed");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(ft, b_o.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    return UCF_OK;
}

int ucf_bessel_k01(int n, const double* z, double* k, int* ierr)
{
    if (n < 1 || !z || !k || !ierr) return fail(UCF_ERR_BAD_ARGUMENT, "bad Bessel request");
    int rc = require_device();
    if (rc) return rc;
    dev_buf b_z, b_k, b_e;
    if (b_z.alloc(sizeof(double) * 2 * n) || b_k.alloc(sizeof(double) * 4 * n) || b_e.alloc(sizeof(int) * n))
        return fail(UCF_ERR_NOMEM, "device allocation failed");
    HIP_TRY(hipMemcpy(b_z.p, z, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    rc = ucf_faithful::launch_bessel(n, (const double*)b_z.p, (double*)b_k.p, (int*)b_e.p, nullptr);
    if (rc) return fail(rc, "Bessel kernel launch failed");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(k, b_k.p, sizeof(double) * 4 * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ierr, b_e.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    return UCF_OK;
}

int ucf_wynn_epsilon(int n, int nterms, const double* series, double* acc, int* status)
{
    if (n < 1 || nterms < 1 || nterms > 64 || !series || !acc || !status) return fail(UCF_ERR_BAD_ARGUMENT, "bad Wynn request");
    int rc = require_device();
    if (rc) return rc;
    dev_buf b_s, b_a, b_st;
    if (b_s.alloc(sizeof(double) * 2 * (size_t)n * nterms) || b_a.alloc(sizeof(double) * 2 * n) || b_st.alloc(sizeof(int) * n))
        return fail(UCF_ERR_NOMEM, "device allocation failed");
    HIP_TRY(hipMemcpy(b_s.p, series, sizeof(double) * 2 * (size_t)n * nterms, hipMemcpyHostToDevice));
    rc = ucf_faithful::launch_wynn(n, nterms, (const double*)b_s.p, (double*)b_a.p, (int*)b_st.p, nullptr);
    if (rc) return fail(rc, "Wynn kernel launch failed");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(acc, b_a.p, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(status, b_st.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    return UCF_OK;
}

int ucf_extraptozero(int n, int R, const double* x, const double* y, double* out)
{
    if (n < 1 || R < 1 || R > UCF_MAX_R || !x || !y || !out) return fail(UCF_ERR_BAD_ARGUMENT, "bad extrapolation request");
    int rc = require_device();
    if (rc) return rc;
    dev_buf b_x, b_y, b_o;
    if (b_x.alloc(sizeof(double) * R) || b_y.alloc(sizeof(double) * 2 * (size_t)n * R) || b_o.alloc(sizeof(double) * 2 * n))
        return fail(UCF_ERR_NOMEM, "device allocation failed");
    HIP_TRY(hipMemcpy(b_x.p, x, sizeof(double) * R, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b_y.p, y, sizeof(double) * 2 * (size_t)n * R, hipMemcpyHostToDevice));
    rc = ucf_faithful::launch_extrap(n, R, (const double*)b_x.p, (const double*)b_y.p, (double*)b_o.p, nullptr);
    if (rc) return fail(rc, "extrapolation kernel launch failed");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, b_o.p, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
    return UCF_OK;
}

}  // extern "C"
