/*
 * TEST INFRASTRUCTURE -- CPU restatement ("oracle") of the reference's
 * Laplace-Hankel drawdown path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (unconfined_amd/) never does.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (or to a
 * stated few-ulp tolerance) against outputs of the unmodified reference compiled
 * by oracle/Makefile (`make ref`) through oracle/ref_harness.f90, and end-to-end
 * against the reference binary's .out files; the vectors are committed under
 * tests/golden/ (generator: oracle/gen_golden.py).
 *
 * The same source builds in binary64 (libucf_oracle.so, the oracle proper, glibc
 * libm like the reference's flang build) and in binary128 (libucf_oracle_q.so,
 * a "truth" used only to arbitrate differences at the reference's noise floor).
 * The exported interface is double in both.
 */
#ifndef UCF_ORACLE_H
#define UCF_ORACLE_H

#include "../include/ucf.h"   /* POD structs only (ucf_params, ucf_derived) */

#ifdef __cplusplus
extern "C" {
#endif

/* numerical constants of the reference (constants.f90:51-66) */
double ucfo_maxexp(void);

/* driver_io.f90:159-186 (MNtype==1 overrides) + 531-567 */
int ucfo_nondim(const ucf_params* P, ucf_derived* D);
/* driver_io.f90:575-586 */
void ucfo_zlay(const ucf_derived* D, int nz, const double* zD, int* zLay);
/* driver_io.f90:628-647 */
void ucfo_j0_zeros(int n, double* j0z);
/* driver_io.f90:654-664 */
void ucfo_split_vector(const int j0s[2], int nt, const double* tD, int* sv);
/* utility.f90:34-57 */
void ucfo_linspace(double lo, double hi, int n, double* v);
void ucfo_logspace(int lo, int hi, int n, double* v);

/* invlap.f90:154-172 */
void ucfo_pvalues(double tee, int M, double alpha, double tol, double* p_re_im);
/* invlap.f90:143-152 -> 46-141 (scalar t) */
double ucfo_dehoog(int M, double alpha, double tol, double t, double tee, const double* fp_re_im);
/* integration.f90:31-67: weights w[N] (normalised, sum = 2) and abscissae a[N] on [0,s] */
void ucfo_tanh_sinh(int k, double s, double* w, double* a);
/* integration.f90:70-120: interior nodes/weights x[ord-2], w[ord-2] */
void ucfo_gauss_lobatto(int ord, double* x, double* w);
/* integration.f90:125-189; status 0 ok, 1 truncated, 2 sentinel (<4 terms), 3 early exit */
void ucfo_wynn_epsilon(int n, const double* series_re_im, double* acc_re_im, int* status);
/* integration.f90:192-237 */
void ucfo_extraptozero(int n, const double* x, const double* y_re_im, double* out_re_im);

/* cbesk(z, fnu=0, kode=1, n=2): cbessel.f90:877 -> cbknu :5036; returns ierr */
int ucfo_cbesk01(double zr, double zi, double* k_re_im);

/* laplace_hankel_solutions.f90:30-120; fp[nz][np] complex (column-major like fp(np,nz)) */
int ucfo_lap_hank_soln(const ucf_params* P, const ucf_derived* D, double a, double rD,
                       int np, const double* p_re_im, int nz, const double* zD, const int* zLay,
                       double* fp_re_im);

/* one (t,r) point: body of driver.f90:100-232 with single-point-run semantics
 * (abscissae from this point's own arg; Q5: all-zero -> 0).
 * stage (optional, may be NULL) receives intermediate vectors, see ucf_oracle.c. */
typedef struct ucfo_stage {
    double* p;        /* [np][2]            */
    double* fa;       /* [N][nz][np][2]     densest tanh-sinh samples */
    double* tmp;      /* [R][nz][np][2]     */
    double* finint;   /* [nz][np][2]        */
    double* glarea;   /* [nacc][nz][np][2]  */
    double* infint;   /* [nz][np][2]        */
    double* totlap;   /* [nz][np][2]        */
} ucfo_stage;

int ucfo_point(const ucf_params* P, const ucf_derived* D, const double* j0z,
               double tD, double rD, int sv, int nz, const double* zD, const int* zLay,
               double* h, double* dh, ucfo_stage* stage);

/* counters of the in-band rules taken by ucfo_point / ucfo_batch since the last reset, in the order of ucf_stats
 * (include/ucf.h): nan_scrubbed, zero_vectors, wynn_truncated, wynn_sentinel, wynn_early_exit, wynn_all_zero */
void ucfo_stats_reset(void);
void ucfo_stats_get(long long* out6);

/* many points, OpenMP over points (threads<=0: all cores); returns 0 */
int ucfo_batch(const ucf_params* P, int npts, const double* tD, const double* rD, const int* sv,
               int nz, const double* zD, const int* zLay, double* h, double* dh, int threads);

#ifdef __cplusplus
}
#endif
#endif
