// ucf_plan.h -- internal structures shared by the host API and the kernels.
#pragma once
#include "../../include/ucf.h"

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
    int nj0z, any_lay3;    // any_lay3: some depth of the launch (of any plan of a parameter batch) lies above the screen top
    int nz_out, z_off;     // depths of the whole call / offset of this launch's chunk: out index = pt*nz_out + z_off + z
    double timePar[2];
    double kappa, alphaD, beta;
    double lD, dD, bD, dD1, lD1;              // dD1 = 1-dD, lD1 = 1-lD (laplace_hankel_solutions.f90:157-158)
    double MoenchInvGamma[UCF_MAX_MOENCH];    // 1.0/gamma_m (:74)
    double alpha, logtol, maxexp;
    // fast flavour: hoisted reciprocals, plan-level exact folds, validity bound of the fast evaluation
    double inv_kappa, inv_bD, fast_eta_max, fast_im_max;
    int fold_dD, fold_lD1, share_g1top, _pad2;
    // Hantush with wellbore storage (:204-301): rDw, CDw (:250), tDb (:253)
    double hs_rDw, hs_CDw, hs_tDb;
    // Mishra/Neuman (Malama form, :404-442): host-evaluated scalar prefactors
    double mn_vartheta, mn_u0;
    // Mishra/Neuman FD (:444-544)
    double fd_h, fd_invhsq, fd_beta0, fd_beta3, fd_expmb2;   // exp(-beta2)
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
};

struct ucf_plan {
    ucf_params P;
    ucf_derived D;
    ucf_dev_params dev;        // zD/zLay/nz filled per call
    int mode;                  // 0 faithful, 1 fast
    int force_layout0;         // diagnostic: never use the lane = time layout
    int timing;                // bracket the dominant kernel with events
    void* ev0;                 // hipEvent_t
    void* ev1;
    int ev_valid;
    const char* last_kernel;
    int device;
    double* d_tables;          // one allocation holding all tables
    size_t tables_bytes;
    size_t o_tsx, o_tsw, o_glx, o_glw, o_j0z, o_fde, o_sched;     // offsets (doubles) of the tables in it
    // host copies (for the accessor API)
    double* h_j0z;
    double* h_ts_x;
    double* h_ts_w;            // [R][N]
    double* h_gl_x;
    double* h_gl_w;
    int Nv[UCF_MAX_R];
    // abscissa-table workspace (grown on demand, never shrunk)
    double* d_work;
    size_t work_bytes;
    // LAYOUT 1 workspace: accelerated transform totlap(t, r, z, m), 16 B each
    double* d_totlap;
    size_t totlap_bytes;
    // finished J0-interval areas of the resident workgroups: [UCF_GRID_SLOTS][nacc][nz][64] complex
    double* d_glscr;
    size_t glscr_bytes;
    // a grid expanded into the point list it stands for (small time vectors): tD | rD | sv per point
    double* d_expand;
    size_t expand_points;
    // device-side ordering of a point list by radius (ucf_drawdown_batch_device): keys, permutation, staged inputs/outputs
    void* d_sort;
    size_t sort_bytes;
    // fast flavour: state of every work item between integrate_kernel and point_kernel
    // [items][(R+1+nacc)*nz][64] complex, and the abscissae done per item
    double* d_state;
    size_t state_bytes;
    int* d_ndone;
    size_t ndone_items;
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
                  const ucf_dev_params* d_params = nullptr, int ppp = 1, int pbase = 0);
int launch_points_chunked(const ucf_dev_params& dp, int npts, int per_point, int nr, int nsv, int svmin, const double* d_tD,
                          const double* d_rD, const int* d_sv, const double* d_tab, double* d_totlap, double* d_h,
                          double* d_dh, ucf_stats* d_stats, void* stream, double* d_glscr, double* d_state, int* d_ndone,
                  const ucf_dev_params* d_params = nullptr, int ppp = 1, int pbase = 0);
int launch_grid_transposed(const ucf_dev_params& dp, int nt, int nr, int ir0, int nrc, int svmin, const double* d_tD,
                           const double* d_rD, const double* d_tab, double* d_totlap, double* d_h, double* d_dh,
                           ucf_stats* d_stats, void* stream, void* ev0, void* ev1, double* d_glscr, double* d_state, int* d_ndone);
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
}
namespace ucf_fast {
int launch_points_chunked(const ucf_dev_params& dp, int npts, int per_point, int nr, int nsv, int svmin, const double* d_tD,
                          const double* d_rD, const int* d_sv, const double* d_tab, double* d_totlap, double* d_h,
                          double* d_dh, ucf_stats* d_stats, void* stream, double* d_glscr, double* d_state, int* d_ndone,
                  const ucf_dev_params* d_params = nullptr, int ppp = 1, int pbase = 0);
int launch_grid_transposed(const ucf_dev_params& dp, int nt, int nr, int ir0, int nrc, int svmin, const double* d_tD,
                           const double* d_rD, const double* d_tab, double* d_totlap, double* d_h, double* d_dh,
                           ucf_stats* d_stats, void* stream, void* ev0, void* ev1, double* d_glscr, double* d_state, int* d_ndone);
int launch_points(const ucf_dev_params& dp, int npts, int per_point, int nr, int nsv, int svmin, const double* d_tD,
                  const double* d_rD, const int* d_sv, const double* d_tab, double* d_h, double* d_dh,
                  ucf_stats* d_stats, void* stream, double* d_glscr, double* d_state, int* d_ndone,
                  const ucf_dev_params* d_params = nullptr, int ppp = 1, int pbase = 0);
int launch_points_lanes(const ucf_dev_params& dp, int npts, int ppp, const double* d_tD, const double* d_rD, const int* d_sv,
                        const double* d_tab, double* d_totlap, double* d_h, double* d_dh, ucf_stats* d_stats, void* stream,
                        double* d_state, int* d_ndone, const ucf_dev_params* d_params = nullptr, int pbase = 0);
int launch_samples(const ucf_dev_params& dp, int n_a, const double* d_a, double rD, const double* d_p, double* d_fp,
                   void* stream);
// bytes of integrate_kernel -> point_kernel state per work item (0 where the flavour / model has no integrate_kernel)
size_t state_bytes_per_item(const ucf_dev_params& dp);
}
