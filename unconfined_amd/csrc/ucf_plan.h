// ucf_plan.h -- internal structures shared by the host API and the kernels.
#pragma once
#include "../../include/ucf.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#define UCF_WAVE 64
#define UCF_MAX_R 16
extern int ucf_grid_slots;     /* workgroups per launch (grid-stride over the work items); each owns a scratch slot */
extern int ucf_finish_part;   /* diagnostic: lanes per scratch part in finish_kernel (16/32/64; 0 = choose from the LDS footprint) */
#define UCF_GRID_SLOTS ucf_grid_slots

// Everything a kernel needs, passed by value as one kernel argument (lives in
// SGPRs / the scalar cache: it is wave-uniform).  Table pointers are device
// pointers into one small per-plan allocation that stays L2/scalar-cache hot.
struct ucf_dev_params {
    int model, MNtype, order, timeType, MoenchM;
    int M, np, k, N, R, nacc, ngl, nz;
    int tab_premul;        // the abscissa table's Gauss-Lobatto entries carry their quadrature weight (fast flavour; abscissa_kernel)
    int nj0z, any_lay3;    // any_lay3: some depth of the launch (of any plan of a parameter batch) lies above the screen top
    int any_lay1;          // ... below the screen bottom
    int any_fold;          // some plan of the launch folds a screen term (fold_dD or fold_lD1) or is model 4 (no screen terms): 0 = the NOFOLD instantiations may run
    int nz_out, z_off;     // depths of the whole call / offset of this launch's chunk: out index = pt*nz_out + z_off + z
    double timePar[2];
    double kappa, alphaD, beta;
    double lD, dD, bD, dD1, lD1;              // dD1 = 1-dD, lD1 = 1-lD (laplace_hankel_solutions.f90:157-158)
    double MoenchInvGamma[UCF_MAX_MOENCH];    // 1.0/gamma_m (:74)
    double alpha, logtol, maxexp;
    // fast flavour: hoisted reciprocals, plan-level exact folds, validity bound of the fast evaluation
    double inv_kappa, inv_bD, fast_eta_max, fast_im_max;
    int fold_dD, fold_lD1, share_g1top, _pad2;
    double g1_delta;       // (dD1 - 1) + dD, exact: the argument of cosh(eta (dD1 - 1)) is -(dD - g1_delta) (share_g1top = 2)
    // Hantush with wellbore storage (:204-301): rDw, CDw (:250), tDb (:253)
    double hs_rDw, hs_CDw, hs_tDb;
    // Mishra/Neuman (Malama form, :404-442): host-evaluated scalar prefactors
    double mn_vartheta, mn_u0, mn_c3;         // mn_c3 = 1 / (kappa u0^2): (eta1 / u0)^2 = (p vartheta + a^2) mn_c3
    // Mishra/Neuman FD (:444-544)
    double fd_h, fd_invhsq, fd_beta0, fd_beta3, fd_expmb2;   // exp(-beta2)
    double fd_isk, fd_gmax;                   // 1/sqrt(K), K = (1/h^2 - beta3/h)/h^2 (0 if K <= 0); 2^(500/order) - 1 (fd_inverse_B2)
    double hv[UCF_MAX_R];                     // Richardson spacings (driver.f90:91)
    double zD[UCF_MAX_NZ];
    int zLay[UCF_MAX_NZ];
    const double* ts_x;    // [N]      tanh(u2)+1 of the densest level (integration.f90:62 without *s/2)
    const double* ts_w;    // [R][N]   normalised weights of level j in row j-1 (first Nv(j) entries)
    const double* gl_x;    // [ngl]
    const double* gl_w;    // [ngl]
    const double* j0z;     // [nj0z]
    const double* fd_e;    // [order]  exp(-beta1*(j-1)*h)
    const double* sched;   // timeType = -n: [n] start times | [n] rate increments | final time | sum of increments
    const double* sc_tab;  // [256] x (sin, cos)(k pi / 128) | [128] x (hi, lo) of 2^(j/128): copied into LDS by the fast flavour's kernels
                           // (sincos_tab_, exp_tab_)
};
#define UCF_SC_ENTRIES (256 + 128)   /* 16-byte units of that table */
#define UCF_IWPB 4             /* waves per workgroup of integrate_kernel: they share the sin/cos table in LDS */

// HIP events around the kernels of the last lane = time grid call issued through a workspace (measurement only):
// bracket i = ev[2i] .. ev[2i+1] around the kernel called name[i].
#define UCF_MAX_TIMED 96          /* brackets per call: kernels of a launch sequence x chunks of radii */
struct ucf_timers {
    void* ev[2 * UCF_MAX_TIMED];     // hipEvent_t, created on first use
    char name[UCF_MAX_TIMED][96];
    int n;                           // brackets recorded by the last timed call
    int open;                        // a bracket is open
};
// begin a bracket (closing the one before it); no-ops on a NULL timer set
static inline void ucf_tm_close(ucf_timers* tm, void* stream)
{
    if (!tm || !tm->open) return;
    (void)hipEventRecord((hipEvent_t)tm->ev[2 * (tm->n - 1) + 1], (hipStream_t)stream);
    tm->open = 0;
}
static inline void ucf_tm_mark(ucf_timers* tm, const char* name, void* stream)
{
    // UCF_TRACE_LAUNCHES=1 (diagnostic): wait for everything launched so far and name the kernel that comes next on stderr,
    // so that the last line before a device fault names the kernel that faulted
    static const bool trace = [] { const char* e = std::getenv("UCF_TRACE_LAUNCHES"); return e && *e && *e != '0'; }();
    if (trace) {
        const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        std::fprintf(stderr, "[ucf] stream %s; next: %s\n", e == hipSuccess ? "clean" : hipGetErrorString(e), name);
        std::fflush(stderr);
    }
    if (!tm) return;
    ucf_tm_close(tm, stream);
    if (tm->n >= UCF_MAX_TIMED) return;
    const int i = tm->n;
    for (int k = 2 * i; k < 2 * i + 2; k++)
        if (!tm->ev[k]) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; tm->ev[k] = e; }
    std::snprintf(tm->name[i], sizeof(tm->name[i]), "%s", name);
    (void)hipEventRecord((hipEvent_t)tm->ev[2 * i], (hipStream_t)stream);
    tm->n = i + 1;
    tm->open = 1;
}

// ucf_debug_stages: what the launcher did (one record per transform launch sequence of the call)
struct ucf_debug_rec {
    int count = 0;                 // launch sequences seen (the hook wants exactly one)
    int layout = -1, nwork = 0, per_point = 0, nr = 0, nt = 0, ir0 = 0, nrc = 0, npts = 0;
    double* d_totlap0 = nullptr;   // LAYOUT 0 keeps no transform: the hook lends it a buffer [npts][nz][np]
};

// Everything a call in flight writes.  A plan keeps one workspace per HIP stream that has called into it, so calls on
// different streams never share scratch; calls that name the same stream are enqueued under the workspace's lock and
// run in stream order.  Buffers only ever grow; a buffer that is outgrown is RETIRED (kept until ucf_plan_reserve /
// ucf_plan_destroy), never freed while kernels may still read it -- no synchronisation inside the *_device entries.
struct ucf_buffer {
    void* p = nullptr;
    size_t bytes = 0;
    void* base = nullptr;          // what hipMalloc returned (== p unless UCF_GUARD places the buffer at the END of its pages)
};
struct ucf_workspace {
    void* stream = nullptr;
    std::mutex mu;                 // held while a call enqueues its work on `stream`
    ucf_buffer work;               // abscissa table: rows x nabs x (a, a*J0(a rD))
    ucf_buffer totlap;             // accelerated transform totlap(t, r, z, m) of the lane = time / lane = point layouts, 16 B each
    ucf_buffer glscr;              // finished J0-interval areas of the resident workgroups: [UCF_GRID_SLOTS][nacc][nz][64] complex
    ucf_buffer expand;             // a grid expanded into the point list it stands for: tD | rD | sv per point
    ucf_buffer sort;               // device-side ordering of a point list by radius: keys, permutation, staged inputs/outputs
    ucf_buffer state;              // integrate kernel -> finish/point kernel: [items][(R+1+nacc)*nz][64] complex
    ucf_buffer ndone;              // abscissae done per item | count of unfinished | unfinished items
    ucf_buffer pblocks;            // parameter blocks of a parameter-batched launch (ucf_drawdown_multi)
    std::vector<void*> retired;    // outgrown buffers
    bool dry = false;              // ucf_plan_reserve: size the buffers, launch nothing
    struct ucf_debug_rec* dbg = nullptr;   // ucf_debug_stages: the transform launches of the call in progress are recorded here
    ucf_timers tm = {};
    int tm_valid = 0;
    const char* tm_names[UCF_MAX_TIMED] = {};
};

struct ucf_plan {
    ucf_params P = {};
    ucf_derived D = {};
    ucf_dev_params dev = {};   // zD/zLay/nz filled per call
    int mode = 0;              // 0 faithful, 1 fast
    int force_layout0 = 0;     // diagnostic: never use the lane = time layout
    int timing = 0;            // bracket the kernels of single-chunk grid calls with events
    int device = 0;
    double* d_tables = nullptr;    // one allocation holding all tables
    size_t tables_bytes = 0;
    size_t o_tsx = 0, o_tsw = 0, o_glx = 0, o_glw = 0, o_j0z = 0, o_fde = 0, o_sched = 0, o_sct = 0;     // offsets (doubles) of the tables in it
    // host copies (for the accessor API)
    double* h_j0z = nullptr;
    double* h_ts_x = nullptr;
    double* h_ts_w = nullptr;      // [R][N]
    double* h_gl_x = nullptr;
    double* h_gl_w = nullptr;
    int Nv[UCF_MAX_R] = {};
    // per-stream workspaces (see ucf_workspace); `mu` guards the list and the counters
    std::mutex mu;
    std::vector<ucf_workspace*> ws;
    ucf_workspace* last_timed = nullptr;
    long long n_alloc = 0;         // device allocations made on behalf of calls (ucf_plan_alloc_count)
    // the stream of the host entry points that run asynchronously inside the library (ucf_drawdown_grid, ucf_drawdown_grid_multi):
    // created on first use, destroyed with the plan -- a workspace is keyed by its stream, so the stream must outlive it
    void* own_stream = nullptr;
};

// launchers implemented in ucf_kernels.hip (one set per build flavour)
namespace ucf_faithful {
// abscissa tables (shared by both flavours): tab[row][nabs] of (a, a*J0(a*rD))
int launch_abscissae(const ucf_dev_params& dp, int nrows, int per_point, int nsv, int svmin, const double* d_rD,
                     const int* d_sv, double* d_tab, void* stream);
int launch_expand_grid(int nt, int nr, const double* d_tD, const int* d_sv, const double* d_rD, double* d_tDp, double* d_rDp,
                       int* d_svp, void* stream);
int launch_points(const ucf_dev_params& dp, int npts, int per_point, int nr, int nsv, int svmin, const double* d_tD,
                  const double* d_rD, const int* d_sv, const double* d_tab, double* d_h, double* d_dh,
                  ucf_stats* d_stats, void* stream, double* d_glscr, double* d_state, int* d_ndone,
                  const ucf_dev_params* d_params = nullptr, int ppp = 1, int pbase = 0, double* d_dbg_totlap = nullptr);
int launch_points_chunked(const ucf_dev_params& dp, int npts, int per_point, int nr, int nsv, int svmin, const double* d_tD,
                          const double* d_rD, const int* d_sv, const double* d_tab, double* d_totlap, double* d_h,
                          double* d_dh, ucf_stats* d_stats, void* stream, double* d_glscr, double* d_state, int* d_ndone,
                  const ucf_dev_params* d_params = nullptr, int ppp = 1, int pbase = 0);
int launch_grid_transposed(const ucf_dev_params& dp, int nt, int nr, int ir0, int nrc, int svmin, const double* d_tD,
                           const double* d_rD, const double* d_tab, double* d_totlap, double* d_h, double* d_dh,
                           ucf_stats* d_stats, void* stream, ucf_timers* tm, double* d_glscr, double* d_state, int* d_ndone);
int launch_points_lanes(const ucf_dev_params& dp, int npts, int ppp, const double* d_tD, const double* d_rD, const int* d_sv,
                        const double* d_tab, double* d_totlap, double* d_h, double* d_dh, ucf_stats* d_stats, void* stream,
                        double* d_state, int* d_ndone, const ucf_dev_params* d_params = nullptr, int pbase = 0);
int launch_samples(const ucf_dev_params& dp, int n_a, const double* d_a, double rD, const double* d_p, double* d_fp,
                   void* stream);
int launch_bessel(int n, const double* d_z, double* d_k, int* d_ierr, void* stream);
size_t state_bytes_per_item(const ucf_dev_params& dp);
int launch_dehoog(int n, int M, double alpha, double logtol, const double* d_t, const double* d_tee,
                  const double* d_fp, double* d_ft, void* stream);
int launch_wynn(int n, int nterms, const double* d_series, double* d_acc, int* d_status, void* stream);
int launch_extrap(int n, int R, const double* d_x, const double* d_y, double* d_out, void* stream);
int launch_debug_gather(const ucf_dev_params& dp, int layout, int nwork, int per_point, int nr, int nt, int ir0, const double* d_state,
                        const int* d_ndone, double* d_out_state, int* d_out_ndone, void* stream);
int launch_wynn_regs(int n, int nterms, const double* d_series, double* d_acc, int* d_status, void* stream);
int launch_dehoog_tiles_hook(const ucf_dev_params& dp, int n, const double* d_tD, const double* d_totlap, double* d_h, double* d_dh, void* stream);
}
namespace ucf_fast {
int launch_points_chunked(const ucf_dev_params& dp, int npts, int per_point, int nr, int nsv, int svmin, const double* d_tD,
                          const double* d_rD, const int* d_sv, const double* d_tab, double* d_totlap, double* d_h,
                          double* d_dh, ucf_stats* d_stats, void* stream, double* d_glscr, double* d_state, int* d_ndone,
                  const ucf_dev_params* d_params = nullptr, int ppp = 1, int pbase = 0);
int launch_grid_transposed(const ucf_dev_params& dp, int nt, int nr, int ir0, int nrc, int svmin, const double* d_tD,
                           const double* d_rD, const double* d_tab, double* d_totlap, double* d_h, double* d_dh,
                           ucf_stats* d_stats, void* stream, ucf_timers* tm, double* d_glscr, double* d_state, int* d_ndone);
int launch_points(const ucf_dev_params& dp, int npts, int per_point, int nr, int nsv, int svmin, const double* d_tD,
                  const double* d_rD, const int* d_sv, const double* d_tab, double* d_h, double* d_dh,
                  ucf_stats* d_stats, void* stream, double* d_glscr, double* d_state, int* d_ndone,
                  const ucf_dev_params* d_params = nullptr, int ppp = 1, int pbase = 0, double* d_dbg_totlap = nullptr);
int launch_points_lanes(const ucf_dev_params& dp, int npts, int ppp, const double* d_tD, const double* d_rD, const int* d_sv,
                        const double* d_tab, double* d_totlap, double* d_h, double* d_dh, ucf_stats* d_stats, void* stream,
                        double* d_state, int* d_ndone, const ucf_dev_params* d_params = nullptr, int pbase = 0);
int launch_samples(const ucf_dev_params& dp, int n_a, const double* d_a, double rD, const double* d_p, double* d_fp,
                   void* stream);
// bytes of integrate_kernel -> point_kernel state per work item (0 where the flavour / model has no integrate_kernel)
size_t state_bytes_per_item(const ucf_dev_params& dp);
size_t lt_table_bytes(const ucf_dev_params& dp, size_t rows);
int launch_wynn_regs(int n, int nterms, const double* d_series, double* d_acc, int* d_status, void* stream);
int launch_dehoog_tiles_hook(const ucf_dev_params& dp, int n, const double* d_tD, const double* d_totlap, double* d_h, double* d_dh, void* stream);
}
