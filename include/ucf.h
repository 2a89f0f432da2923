/*
 * ucf.h -- C ABI of the MI355X-native Laplace-Hankel drawdown engine.
 *
 * This is the drop-in boundary for the one hot path of klkuhlm/unconfined: the
 * per-(t,r) loop body of `program Driver` (reference driver.f90:100-276) and the
 * module procedures it imports (reference driver.f90:28-40).  The reference has
 * no FFI for this path (its only ISO_C_BINDING precedent is arb_J/arb_Y,
 * reference laplace_hankel_solutions.f90:310-325: scalars by value, plain
 * numbers); every entry point below therefore names the reference procedure or
 * loop it replaces.  A reference-side binding (Fortran `bind(C)` interface block)
 * is shown in INTEGRATION.md and shipped in unconfined_amd/fortran/.
 *
 * Conventions
 *   - plain C: pointers + sizes, caller-owned contiguous fp64 / int32 arrays;
 *   - complex vectors travel as interleaved (re,im) doubles;
 *   - every function returns UCF_OK (0) or a negative ucf_status; nothing aborts
 *     (the reference `stop`s on bad input, driver_io.f90:78-378; we return the
 *     matching UCF_ERR_* instead and ucf_last_error() gives the message);
 *   - "in-band" numerical rules of the reference (NaN->0 scrub, Wynn-epsilon
 *     truncation/sentinel, epsilon-table early exit, FD underflow guard) are
 *     part of the numerical contract and are reproduced on the device;
 *   - threading / streams (the reference calls its procedures from OpenMP threads with shared read-only
 *     parameter objects, driver.f90:129-230): any number of host threads may call the drawdown entry points on ONE
 *     plan at the same time.  Everything a call writes lives in a workspace that the plan keeps PER HIP STREAM:
 *     calls on different streams share nothing and run concurrently; calls that name the same stream (the host
 *     point-list entry uses the default stream, the host grid entries a stream that the plan owns for its lifetime)
 *     are enqueued one after the other under the workspace's lock and execute in stream order.  The *_device entry points never synchronise; they allocate only when a workspace
 *     must grow (outgrown buffers are kept until the stream has drained, never freed under a running kernel), and not
 *     at all after ucf_plan_reserve -- which is what capturing them into a hipGraph needs.
 *     What may NOT overlap with calls in flight on the same plan: ucf_plan_update, ucf_plan_set_mode,
 *     ucf_plan_set_timing, ucf_plan_destroy (the caller orders those; ucf_plan_update waits for the plan's own
 *     streams before it rewrites a device table).
 *   - there is NO CPU fallback: every compute entry point fails with
 *     UCF_ERR_NO_DEVICE if no gfx950-capable HIP device is usable.
 */
#ifndef UCF_H
#define UCF_H

#ifdef __cplusplus
extern "C" {
#endif

#define UCF_VERSION 100        /* 0.1.0 */
#define UCF_MAX_MOENCH 16      /* max number of Moench alphas (driver_io.f90:142-151) */
#define UCF_MAX_NZ 32          /* depths per LAUNCH; calls with more depths are walked in chunks by the library */
#define UCF_MAX_SCHEDULE 100    /* steps of a piecewise-constant pumping schedule (time.f90:81-95) */
#define UCF_MAX_LAP_M 127      /* 2M+1 <= 256: the wave-cooperative de Hoog holds up to four samples per lane (the reference takes
                                  any M >= 2, driver_io.f90:306-309; its QD table is numerically void long before M = 127) */

typedef enum ucf_status {
    UCF_OK = 0,
    UCF_ERR_INVALID_MODEL = -1,     /* driver_io.f90:90-97   */
    UCF_ERR_GEOMETRY = -2,          /* driver_io.f90:236-258 */
    UCF_ERR_AQUIFER = -3,           /* driver_io.f90:242-252 */
    UCF_ERR_MISHRA_NEUMAN = -4,     /* driver_io.f90:260-276 */
    UCF_ERR_MALAMA_BETA = -5,       /* driver_io.f90:278-281 */
    UCF_ERR_MOENCH = -6,            /* driver_io.f90:142-149,283-290 */
    UCF_ERR_DEHOOG = -7,            /* driver_io.f90:306-309 */
    UCF_ERR_TANH_SINH = -8,         /* driver_io.f90:316-327 */
    UCF_ERR_GAUSS_LOBATTO = -9,     /* driver_io.f90:329-333 */
    UCF_ERR_UNSUPPORTED = -10,      /* valid in the reference, not built here (see ucf_last_error) */
    UCF_ERR_BAD_ARGUMENT = -11,
    UCF_ERR_NO_DEVICE = -12,
    UCF_ERR_HIP = -13,
    UCF_ERR_NOMEM = -14,
    UCF_ERR_OBSERVATION = -15       /* driver_io.f90:352-383 */
} ucf_status;

/*
 * POD mirror of the reference's parameter types as read from the 18-line deck
 * (types.f90:31-225: well, formation, solution, invLaplace, TanhSinh,
 * invHankel, GaussLobatto).  Dimensional quantities, exactly as on the deck;
 * non-dimensionalisation (driver_io.f90:531-567) happens in ucf_plan_create.
 */
typedef struct ucf_params {
    int model;            /* 0 Theis, 1 Hantush, 2 Hantush+storage, 3 Moench, 4 Malama full, 5 Malama partial, 6 Mishra/Neuman */
    int MNtype;           /* model 6: 0 naive (ARB, unsupported), 1 Malama, 2 finite difference */
    int order;            /* model 6 / MNtype 2: FD nodes in the vadose zone */
    int timeType;         /* pumping-rate time behaviour (time.f90:46): 1..8; -n (-1..-100) = n-step piecewise-CONSTANT
                             schedule; -(100+n) (-101..-200) = n-segment piecewise-LINEAR schedule (time.f90:97-122);
                             the 2n+1 parameters of either are in timeParExt */
    double timePar[2];
    double Q;             /* pumping rate [L^3/T] */
    double l, d;          /* depth to screen bottom / top from aquifer top [L] */
    double rw, rc;        /* well / casing radius [L] */
    double gammaSkin;
    double b;             /* saturated thickness [L] */
    double Kr, kappa;     /* radial K [L/T], Kz/Kr */
    double Ss, Sy;
    double beta;          /* Malama linearisation parameter */
    int MoenchM;
    int _pad0;
    double MoenchAlpha[UCF_MAX_MOENCH];
    double ac, ak, psia, psik, usL;   /* Mishra/Neuman vadose-zone parameters */
    int M;                /* de Hoog: 2M+1 Laplace samples */
    int k;                /* tanh-sinh: N = 2^k - 1 abscissae */
    int R;                /* Richardson levels */
    int nacc;             /* J0-zero intervals accelerated by Wynn-epsilon */
    int ord;              /* Gauss-Lobatto order (ord-2 interior nodes) */
    int j0s[2];           /* min/max J0 zero at which finite/infinite parts split */
    int _pad1;
    double alpha, tol;    /* de Hoog abscissa of convergence, tolerance */
    double rwobs, sF;     /* observation well radius / shape factor (model 2) */
    /* piecewise constant: tpar(1:n) step start times, tpar(n+1) final time, tpar(n+2:2n+1) rates (types.f90:66-70).
     * piecewise linear: tpar(1:n) knot times t_1 < ... < t_n, tpar(n+1) = t_f, tpar(n+2:2n+1) = the rate at t_2, ..., t_n,
     * t_f; the rate is 0 up to t_1, continuous ("no jumps", time.f90:98), linear between knots, constant after t_f.
     * (The reference reads the n rates as y(t_1..t_n) and then indexes y(n+1), one past its array, time.f90:101,115:
     * its transform only ever uses rate DIFFERENCES, i.e. it assumes a rate that starts from 0 at t_1 -- the reading
     * here is the one under which every parameter is used and nothing is read out of bounds) */
    double timeParExt[2 * UCF_MAX_SCHEDULE + 1];
} ucf_params;

/* Derived, dimensionless quantities (driver_io.f90:531-567) -- read back for tests/headers. */
typedef struct ucf_derived {
    double Lc, Tc, Hc;
    double sigma, alphaD, betaD;
    double lD, dD, bD, rDw, rDwobs;
    double acD, akD, lambdaD, psiaD, psikD, usLD, b1, PsiD;
    double MoenchGamma[UCF_MAX_MOENCH];
    double l_eff, d_eff, ac_eff;   /* after the MNtype==1 overrides (driver_io.f90:159-186) */
    int np;       /* 2M+1 */
    int N;        /* 2^k-1 */
    int nj0z;     /* max(j0s)+nacc+1 */
    int nabs;     /* N + nacc*(ord-2): abscissae per point */
} ucf_derived;

/* Counters of the in-band rules taken during a batch. */
typedef struct ucf_stats {
    long long nan_scrubbed;     /* invlap.f90:71-74  : NaN Laplace samples set to 0        */
    long long zero_vectors;     /* invlap.f90:69,139 : all-zero f(p) -> f(t)=0             */
    long long wynn_truncated;   /* integration.f90:150-158 : series cut at first non-finite */
    long long wynn_sentinel;    /* integration.f90:142-149 : < 4 usable terms -> -999999.9  */
    long long wynn_early_exit;  /* integration.f90:169-177 : |denom| <= 2.2e-16             */
    long long wynn_all_zero;    /* driver.f90:209 : every area exactly 0 -> 0 (quirk Q5)   */
} ucf_stats;

typedef struct ucf_plan ucf_plan;

int ucf_version(void);
const char* ucf_last_error(void);          /* thread-local message of the last failure */
const char* ucf_status_string(int status);

/* ---- plan: replaces read_input's numerical half + the `first`-time setup in the driver
 * (driver_io.f90:531-567,628-647; driver.f90:79-91,121-126,138-151,179-183). ---- */
int ucf_plan_create(const ucf_params* P, ucf_plan** out);        /* bound to the HIP device that is current */
int ucf_plan_create_on(const ucf_params* P, int device, ucf_plan** out);   /* bound to HIP device `device` (0-based) */
int ucf_device_count(int* n);                                    /* UCF_ERR_NO_DEVICE if there is none */
void ucf_plan_destroy(ucf_plan* plan);
/* New hydraulic / geometric / schedule parameters for an existing plan (parameter estimation: thousands of
 * parameter sets, one set of numerical settings): everything that depends on them (driver_io.f90:531-567 and the
 * per-model constants) is recomputed, the quadrature tables, workspaces, flavour and timing switches stay.  The model
 * and the numerical settings (M, k, R, nacc, ord, J0 split, FD order, schedule length, number of Moench terms) must
 * not change: UCF_ERR_BAD_ARGUMENT otherwise.  Microseconds instead of the ~0.3 ms of ucf_plan_create. */
int ucf_plan_update(ucf_plan* plan, const ucf_params* P);
int ucf_plan_derived(const ucf_plan* plan, ucf_derived* out);
/* the same quantities without a plan (host arithmetic only, no GPU needed): read_input's checks (driver_io.f90:88-333)
 * and its non-dimensionalisation (:531-567) */
int ucf_nondimensionalise(const ucf_params* P, ucf_derived* out);
int ucf_plan_j0z(const ucf_plan* plan, int n, double* j0z);             /* driver_io.f90:628-647 */
int ucf_plan_tanh_sinh(const ucf_plan* plan, int level /*1..R*/, int n, double* w, double* x_unit /* tanh(u2)+1, level R only, may be NULL */);
int ucf_plan_gauss_lobatto(const ucf_plan* plan, int n, double* x, double* w);
/* execution mode: 0 = faithful (reference operation order, no FMA contraction),
 *                 1 = fast (same algorithm, FMA contraction + shared subexpressions). */
int ucf_plan_set_mode(ucf_plan* plan, int mode);

/* measurement: when enabled, every kernel of a following grid call in the lane = time layout is bracketed by HIP events
 * on the call's stream.  ucf_plan_kernel_times waits for the brackets of the last such call and returns one row per
 * kernel in order of first launch: total duration [ms], number of launches (a call that walks the radii in chunks
 * launches every kernel once per chunk; may be NULL) and name (as rocprofv3 prints it, without "void " and the
 * argument list); ucf_plan_kernel_ms returns the per-launch duration of the kernel with the largest total. */
int ucf_plan_set_timing(ucf_plan* plan, int enable);
int ucf_plan_kernel_times(ucf_plan* plan, int cap, double* ms, int* launches, const char** names, int* n);
int ucf_plan_kernel_ms(ucf_plan* plan, double* ms, const char** kernel_name);

/* Size the workspaces of `stream` (hipStream_t as void*, NULL = default stream) for calls to come -- a grid of nt x nr
 * points (0 x 0: none) and / or a point list of npts points (0: none), nz depths each -- so that the *_device entry
 * points allocate nothing afterwards.  ucf_plan_alloc_count: device allocations made so far on behalf of calls. */
int ucf_plan_reserve(ucf_plan* plan, int nt, int nr, int npts, int nz, void* stream);
long long ucf_plan_alloc_count(const ucf_plan* plan);

/* sha256 (first 16 hex digits) of the kernel and host sources this library was built from: measurement files under
 * profiles/ carry it, and bench.py refuses a profile whose id differs from the library it runs. */
const char* ucf_build_id(void);

/* ---- host-side helpers that the reference computes in read_input ---- */
int ucf_logspace(int lo, int hi, int n, double* out);                   /* utility.f90:51-57 */
int ucf_linspace(double lo, double hi, int n, double* out);             /* utility.f90:34-49 */
int ucf_zlay(const ucf_plan* plan, int nz, const double* zD, int* zLay);            /* driver_io.f90:575-586 */
int ucf_split_vector(const ucf_plan* plan, int nt, const double* tD, int* sv);      /* driver_io.f90:654-664 */

/* ---- the hot path: replaces the body of the (i,k) loop nest, driver.f90:100-232.
 * Points are independent (flattened t x r); per point: tD, rD, sv (1-based index
 * into j0z).  Outputs h, dh are dimensionless, [npts][nz] row-major, *before* the
 * screen averaging of driver.f90:234-243 (see ucf_screen_average). ---- */
int ucf_drawdown_batch(ucf_plan* plan, int npts,
                       const double* tD, const double* rD, const int* sv,
                       int nz, const double* zD, const int* zLay,
                       double* h, double* dh, ucf_stats* stats /* may be NULL */);

/* Same, all per-point arrays already resident in HBM; asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = default stream).  `d_stats` may be NULL.
 * Preconditions the library cannot check on device-resident inputs: 1 <= sv[i] <= nj0z - nacc (sv indexes the
 * J0-zero table; the host entry points check it and return UCF_ERR_BAD_ARGUMENT). */
int ucf_drawdown_batch_device(ucf_plan* plan, int npts,
                              const double* d_tD, const double* d_rD, const int* d_sv,
                              int nz, const double* zD, const int* zLay,
                              double* d_h, double* d_dh, ucf_stats* d_stats, void* stream);

/* The same loop body over the product grid the reference's driver actually walks
 * (do i = 1,nt / do k = 1,nr, driver.f90:100,113): nt times (tD[i], sv[i]) x nr radii rD[k].
 * Outputs [nt][nr][nz] row-major.  Abscissae and a*J0(a*rD) depend only on (rD, sv) and are
 * computed once per radius here instead of once per point. */
int ucf_drawdown_grid(ucf_plan* plan, int nt, const double* tD, const int* sv, int nr, const double* rD,
                      int nz, const double* zD, const int* zLay, double* h, double* dh, ucf_stats* stats);
int ucf_drawdown_grid_device(ucf_plan* plan, int nt, const double* d_tD, const int* d_sv, int nr, const double* d_rD,
                             int nz, const double* zD, const int* zLay, double* d_h, double* d_dh,
                             ucf_stats* d_stats, void* stream);
/* (device-resident sv of a grid must lie in the plan's split range [min(j0s), max(j0s)]: it selects a row of the
 *  per-radius abscissa table; ucf_split_vector produces such values and ucf_drawdown_grid checks them) */

/* ---- the sweep on several GPUs.  Every (t,r) point is independent; the shard axis is the reference's
 * own serial loop nest (do i = 1,nt / do k = 1,nr, driver.f90:100,113): the flattened index i*nr + k is cut into
 * `world` contiguous blocks of whole time rows, B = ceil(nt / world) rows each (the last ones may be short or empty),
 * so that a shard is itself a product grid and its results are one contiguous slice of [nt][nr][nz].
 * No data-path collective; the only exchange is the final gather of the slices. */
int ucf_shard_rows(int nt, int world, int rank, int* lo, int* hi);     /* rows [lo, hi) of shard `rank`; no GPU needed */

/* One process per GPU (the bench, RCCL): the rank computes ITS rows of the sweep and leaves them at their place in the
 * full-size device arrays d_h, d_dh [world*B][nr][nz] (d_tD, d_sv: all nt rows), so that an IN-PLACE all-gather of
 * B*nr*nz doubles per rank (ncclAllGather with sendbuff = recvbuff + rank*count; torch.distributed
 * all_gather_into_tensor on a view) completes the arrays on every rank.  Asynchronous on `stream`. */
int ucf_drawdown_grid_shard_device(ucf_plan* plan, int rank, int world, int nt, const double* d_tD, const int* d_sv,
                                   int nr, const double* d_rD, int nz, const double* zD, const int* zLay,
                                   double* d_h, double* d_dh, ucf_stats* d_stats, void* stream);

/* The same with the gather inside the library: the rank's rows, then one in-place ncclAllGather per array on `stream`
 * (sendbuff = recvbuff + rank * B*nr*nz) over the RCCL communicator `comm` (an ncclComm_t passed as void*: the host's
 * own, or one made by ucf_comm_create).  Asynchronous; d_h, d_dh must hold world*B rows.  RCCL is bound when the first
 * of these entries is called (dlopen: libucf.so itself links no collective library); UCF_ERR_UNSUPPORTED without it.
 * This is the "final RCCL gather over xGMI" of the sweep: the reference writes ONE file from one address space
 * (driver.f90:245-273), every rank ends up holding that whole result. */
int ucf_drawdown_grid_allgather(ucf_plan* plan, int rank, int world, int nt, const double* d_tD, const int* d_sv,
                                int nr, const double* d_rD, int nz, const double* zD, const int* zLay,
                                double* d_h, double* d_dh, ucf_stats* d_stats, void* comm, void* stream);
/* A communicator of the library's own for hosts that have none: rank 0 draws the 128-byte id (ncclGetUniqueId) and hands
 * it to the other ranks by whatever the host has (a file, MPI, torch.distributed ...); every rank then calls
 * ucf_comm_create with the HIP device it computes on current (ncclCommInitRank). */
int ucf_comm_unique_id(unsigned char* id128);
int ucf_comm_create(const unsigned char* id128, int world, int rank, void** comm);
int ucf_comm_destroy(void* comm);

/* One process driving ngpu devices (the Fortran host): plans[g] was created with device g current (ucf_plan_create
 * binds a plan to the current HIP device) from the same parameters.  Host arrays in and out like ucf_drawdown_grid;
 * shard g runs on plans[g]'s device on a stream of its own, all devices at once, and the gather is each device's
 * copy of its slice straight into rows lo..hi of h and dh (one PCIe/xGMI transfer per device: staging the slices
 * through one GPU first would only add a hop).  stats: summed over the shards.  ngpu = 1 is ucf_drawdown_grid. */
int ucf_drawdown_grid_multi(ucf_plan* const* plans, int ngpu, int nt, const double* tD, const int* sv, int nr,
                            const double* rD, int nz, const double* zD, const int* zLay, double* h, double* dh,
                            ucf_stats* stats);

/* The point-list counterpart (ucf_drawdown_batch_multi): block g of the list -- ucf_shard_rows over
 * the npts points -- runs on plans[g]'s device, one host thread per device, each through ucf_drawdown_batch (which
 * orders its block by radius); results land in the caller's order in h, dh [npts][nz].  ngpu = 1 is ucf_drawdown_batch. */
int ucf_drawdown_batch_multi(ucf_plan* const* plans, int ngpu, int npts, const double* tD, const double* rD, const int* sv,
                             int nz, const double* zD, const int* zLay, double* h, double* dh, ucf_stats* stats);

/* Parameter-batched evaluation for inversion / fitting (the tool's real use,
 * reference README.md:45-56): the SAME observation points -- dimensional times t[npts], radii r[npts],
 * depths z[nz] (z up from the aquifer base) -- under nplans parameter sets.  Each plan
 * non-dimensionalises with its own Lc, Tc (driver_io.f90:531-567), gets its own layers and split vector.
 * Fast-flavour plans of the same model and numerical settings (M, k/R, nacc/ord, alpha, tol, J0 split) -- the
 * fitting case: only hydraulic / geometric parameters vary -- share ONE launch sequence over (plan, point) work
 * items with per-plan parameter blocks in device memory; any other mix runs plan by plan on a small pool of
 * HIP streams so that the small launches overlap.  Plan by plan the results ARE the single-plan calls; the shared launch
 * sequence runs other instantiations of the same kernels (parameter blocks in memory, often another lane layout), in
 * which the compiler may contract a product and a sum into an FMA where the single-plan instantiation does not: same
 * formulas, results equal to the rounding noise of the fast flavour -- bit for bit on most decks at their own settings,
 * otherwise as far apart as either is from exact arithmetic (tests/test_gpu_contract.py judges that against binary128).
 * h, dh: [nplans][npts][nz]; dimensional (x Hc of each plan) unless dimensionless != 0. */
int ucf_drawdown_multi(ucf_plan* const* plans, int nplans, int npts, const double* t, const double* r,
                       int nz, const double* z, int dimensionless, double* h, double* dh);

/* driver.f90:234-243 (quirk Q2: not a textbook trapezoid) */
int ucf_screen_average(int npts, int zOrd, const double* h, double* havg);

/* ---- stage hooks (device implementations of the imported procedures; used by the
 * parity tests to compare each stage with the oracle) ---- */
/* lap_hank_soln, laplace_hankel_solutions.f90:30-120: fp[n_a][nz][np] complex */
int ucf_eval_samples(ucf_plan* plan, int n_a, const double* a, double rD,
                     int np, const double* p_re_im, int nz, const double* zD, const int* zLay,
                     double* fp_re_im);
/* deHoog_pvalues, invlap.f90:154-172 */
int ucf_pvalues(const ucf_plan* plan, double tee, double* p_re_im);
/* deHoog_invlap (scalar t), invlap.f90:143-152 -> 46-141; n independent problems */
int ucf_dehoog(int n, int M, double alpha, double tol, const double* t, const double* tee,
               const double* fp_re_im /*[n][2M+1]*/, double* ft);
/* wynn_epsilon, integration.f90:125-189; status: 0 ok, 1 truncated, 2 sentinel, 3 early exit */
int ucf_wynn_epsilon(int n, int nterms, const double* series_re_im /*[n][nterms]*/,
                     double* acc_re_im, int* status);
/* extraptozero, integration.f90:192-237 */
int ucf_extraptozero(int n, int R, const double* x /*[R]*/, const double* y_re_im /*[n][R]*/,
                     double* out_re_im);

/* The intermediate stages (driver.f90:129-216) of the PRODUCTION launch sequence: the call runs exactly what
 * ucf_drawdown_grid (grid != 0: nt times x nr radii) or ucf_drawdown_batch on a list already ordered by radius (grid == 0:
 * nt = nr = number of points, rD per point) would run for these sizes -- the same lane layout, kernel instantiations and
 * launch bounds -- and then reads back what those kernels left in the plan's workspace:
 *   state [npts][2M+1][(R+1+nacc)*nz] complex: per Laplace sample the level sums [R][nz] of the tanh-sinh part (WITHOUT the
 *         factor arg/2 of driver.f90:135,154, which the finishing kernel applies), the area of the interval in progress
 *         [nz], the finished J0-interval areas [nacc][nz] (driver.f90:201-203); zeros where the launch sequence keeps no
 *         state (info[1] = 0: the monolithic kernel);
 *   ndone [npts][2M+1]: abscissae the fast evaluators integrated (< nabs: the item was finished by the reference-order
 *         evaluator, whose accumulators never leave the kernel: its state entries are what the hand-over was);
 *   totlap [npts][nz][2M+1] complex: finint + infint (driver.f90:216);   h, dh [npts][nz].
 * info[0..3] = lane layout used (0 sample, 1 time, 3 point), slots per sample, 2M+1, npts.  Needs nz <= the depths of
 * one launch and sizes that make one launch sequence (UCF_ERR_UNSUPPORTED otherwise). */
int ucf_debug_stages(ucf_plan* plan, int grid, int nt, const double* tD, const int* sv, int nr, const double* rD,
                     int nz, const double* zD, const int* zLay, double* state, int* ndone, double* totlap,
                     double* h, double* dh, int* info);
/* wynn_epsilon as the finishing kernel runs it -- both epsilon columns in registers, at most 12 terms -- in flavour
 * `mode` (0 faithful, 1 fast); same conventions as ucf_wynn_epsilon */
int ucf_debug_wynn(int mode, int n, int nterms, const double* series_re_im, double* acc_re_im, int* status);
/* deHoog_invlap as every grid call and long point list runs it (the tiled kernel: cooperative quotient-difference rhombus,
 * continued fraction per lane): n transforms fp[n][2M+1] at times t[n], T = 2 t (driver.f90:106); h[n] = f(t),
 * dh[n] = t * (inverse of p F(p)) (driver.f90:219-230) */
int ucf_debug_dehoog_tiles(int mode, int n, int M, double alpha, double tol, const double* t, const double* fp_re_im,
                           double* h, double* dh);

/* K0(z), K1(z), Re z >= 0: cbesk(z, fnu=0, kode=1, n=2), cbessel.f90:877 -> cbknu :5036 (model 2);
 * k_re_im[n][2][2] = (K0, K1), ierr[n] as cbesk's IERR */
int ucf_bessel_k01(int n, const double* z_re_im, double* k_re_im, int* ierr);

/* the (sin, cos)(k pi / 128), k = 0..255, table that every plan uploads for the fast flavour's evaluators (host code, no
 * GPU needed; tab[256][2]) */
int ucf_sincos_table(double* tab);
/* and the 2^(j/128), j = 0..127, table behind it (exp of the fast flavour's evaluators; tab[128][2] = (hi, lo)) */
int ucf_exp2_table(double* tab);

/* ---- measurement helper: sustained fp64 FMA rate of the device ---- */
int ucf_fp64_fma_peak(double* tflops);

#ifdef __cplusplus
}
#endif
#endif /* UCF_H */
