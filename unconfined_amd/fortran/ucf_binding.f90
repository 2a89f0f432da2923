! ucf_binding.f90 -- ISO_C_BINDING view of include/ucf.h for Fortran hosts.
!
! This is the binding a maintainer of the reference would add to replace the OpenMP loop
! nest of driver.f90:100-232 by the GPU path (the reference's only precedent for
! ISO_C_BINDING is the arb_J/arb_Y interface, laplace_hankel_solutions.f90:310-325).
! The derived types mirror the C structs field by field (sequence + bind(C)).
module ucf_binding
  use, intrinsic :: iso_c_binding
  implicit none
  private

  integer(c_int), parameter, public :: UCF_MAX_MOENCH = 16
  integer(c_int), parameter, public :: UCF_MAX_SCHEDULE = 100
  integer(c_int), parameter, public :: UCF_OK = 0

  type, bind(C), public :: ucf_params
     integer(c_int) :: model, MNtype, order, timeType
     real(c_double) :: timePar(2)
     real(c_double) :: Q, l, d, rw, rc, gammaSkin, b, Kr, kappa, Ss, Sy, beta
     integer(c_int) :: MoenchM, pad0
     real(c_double) :: MoenchAlpha(UCF_MAX_MOENCH)
     real(c_double) :: ac, ak, psia, psik, usL
     integer(c_int) :: M, k, R, nacc, ord, j0s(2), pad1
     real(c_double) :: alpha, tol, rwobs, sF
     real(c_double) :: timeParExt(2*UCF_MAX_SCHEDULE+1)
  end type ucf_params

  type, bind(C), public :: ucf_derived
     real(c_double) :: Lc, Tc, Hc, sigma, alphaD, betaD, lD, dD, bD, rDw, rDwobs
     real(c_double) :: acD, akD, lambdaD, psiaD, psikD, usLD, b1, PsiD
     real(c_double) :: MoenchGamma(UCF_MAX_MOENCH)
     real(c_double) :: l_eff, d_eff, ac_eff
     integer(c_int) :: np, N, nj0z, nabs
  end type ucf_derived

  type, bind(C), public :: ucf_stats
     integer(c_long_long) :: nan_scrubbed, zero_vectors, wynn_truncated, wynn_sentinel, &
          & wynn_early_exit, wynn_all_zero
  end type ucf_stats

  public :: ucf_version, ucf_last_error, ucf_plan_create, ucf_plan_destroy, ucf_plan_update, ucf_plan_derived, &
       & ucf_plan_set_mode, ucf_logspace, ucf_linspace, ucf_zlay, ucf_split_vector, &
       & ucf_drawdown_grid, ucf_drawdown_batch, ucf_screen_average, ucf_error_message, &
       & ucf_nondimensionalise, ucf_device_count, ucf_plan_create_on, ucf_shard_rows, ucf_drawdown_grid_multi, &
       & ucf_drawdown_batch_multi

  interface
     function ucf_version() bind(C, name='ucf_version') result(v)
       import :: c_int
       integer(c_int) :: v
     end function ucf_version

     function ucf_last_error() bind(C, name='ucf_last_error') result(p)
       import :: c_ptr
       type(c_ptr) :: p
     end function ucf_last_error

     function ucf_plan_create(P, plan) bind(C, name='ucf_plan_create') result(rc)
       import :: c_int, c_ptr, ucf_params
       type(ucf_params), intent(in) :: P
       type(c_ptr), intent(out) :: plan
       integer(c_int) :: rc
     end function ucf_plan_create

     subroutine ucf_plan_destroy(plan) bind(C, name='ucf_plan_destroy')
       import :: c_ptr
       type(c_ptr), value :: plan
     end subroutine ucf_plan_destroy

     function ucf_plan_update(plan, P) bind(C, name='ucf_plan_update') result(rc)
       import :: c_int, c_ptr, ucf_params
       type(c_ptr), value :: plan
       type(ucf_params), intent(in) :: P
       integer(c_int) :: rc
     end function ucf_plan_update

     function ucf_plan_derived(plan, D) bind(C, name='ucf_plan_derived') result(rc)
       import :: c_int, c_ptr, ucf_derived
       type(c_ptr), value :: plan
       type(ucf_derived), intent(out) :: D
       integer(c_int) :: rc
     end function ucf_plan_derived

     function ucf_plan_set_mode(plan, mode) bind(C, name='ucf_plan_set_mode') result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: plan
       integer(c_int), value :: mode
       integer(c_int) :: rc
     end function ucf_plan_set_mode

     function ucf_logspace(lo, hi, n, v) bind(C, name='ucf_logspace') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: lo, hi, n
       real(c_double), intent(out) :: v(*)
       integer(c_int) :: rc
     end function ucf_logspace

     function ucf_linspace(lo, hi, n, v) bind(C, name='ucf_linspace') result(rc)
       import :: c_int, c_double
       real(c_double), value :: lo, hi
       integer(c_int), value :: n
       real(c_double), intent(out) :: v(*)
       integer(c_int) :: rc
     end function ucf_linspace

     function ucf_zlay(plan, nz, zD, zLay) bind(C, name='ucf_zlay') result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: plan
       integer(c_int), value :: nz
       real(c_double), intent(in) :: zD(*)
       integer(c_int), intent(out) :: zLay(*)
       integer(c_int) :: rc
     end function ucf_zlay

     function ucf_split_vector(plan, nt, tD, sv) bind(C, name='ucf_split_vector') result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: plan
       integer(c_int), value :: nt
       real(c_double), intent(in) :: tD(*)
       integer(c_int), intent(out) :: sv(*)
       integer(c_int) :: rc
     end function ucf_split_vector

     ! the (i,k) loop nest of driver.f90:100-232 in one call; h, dh are [nz, nr, nt] in Fortran order
     function ucf_drawdown_grid(plan, nt, tD, sv, nr, rD, nz, zD, zLay, h, dh, stats) &
          & bind(C, name='ucf_drawdown_grid') result(rc)
       import :: c_int, c_double, c_ptr, ucf_stats
       type(c_ptr), value :: plan
       integer(c_int), value :: nt, nr, nz
       real(c_double), intent(in) :: tD(*), rD(*), zD(*)
       integer(c_int), intent(in) :: sv(*), zLay(*)
       real(c_double), intent(out) :: h(*), dh(*)
       type(ucf_stats), intent(out) :: stats
       integer(c_int) :: rc
     end function ucf_drawdown_grid

     function ucf_drawdown_batch(plan, npts, tD, rD, sv, nz, zD, zLay, h, dh, stats) &
          & bind(C, name='ucf_drawdown_batch') result(rc)
       import :: c_int, c_double, c_ptr, ucf_stats
       type(c_ptr), value :: plan
       integer(c_int), value :: npts, nz
       real(c_double), intent(in) :: tD(*), rD(*), zD(*)
       integer(c_int), intent(in) :: sv(*), zLay(*)
       real(c_double), intent(out) :: h(*), dh(*)
       type(ucf_stats), intent(out) :: stats
       integer(c_int) :: rc
     end function ucf_drawdown_batch

     ! read_input's checks and non-dimensionalisation (driver_io.f90:88-333,531-567); host arithmetic, no GPU
     function ucf_nondimensionalise(P, D) bind(C, name='ucf_nondimensionalise') result(rc)
       import :: c_int, ucf_params, ucf_derived
       type(ucf_params), intent(in) :: P
       type(ucf_derived), intent(out) :: D
       integer(c_int) :: rc
     end function ucf_nondimensionalise

     function ucf_device_count(n) bind(C, name='ucf_device_count') result(rc)
       import :: c_int
       integer(c_int), intent(out) :: n
       integer(c_int) :: rc
     end function ucf_device_count

     ! a plan bound to HIP device `device` (0-based)
     function ucf_plan_create_on(P, device, plan) bind(C, name='ucf_plan_create_on') result(rc)
       import :: c_int, c_ptr, ucf_params
       type(ucf_params), intent(in) :: P
       integer(c_int), value :: device
       type(c_ptr), intent(out) :: plan
       integer(c_int) :: rc
     end function ucf_plan_create_on

     ! rows [lo, hi) (0-based) of the nt-row sweep that shard `rank` of `world` owns
     function ucf_shard_rows(nt, world, rank, lo, hi) bind(C, name='ucf_shard_rows') result(rc)
       import :: c_int
       integer(c_int), value :: nt, world, rank
       integer(c_int), intent(out) :: lo, hi
       integer(c_int) :: rc
     end function ucf_shard_rows

     ! the same loop nest on ngpu devices at once: plans(g) bound to device g, rows of the i loop (driver.f90:100) in
     ! contiguous blocks, every device's block copied straight to its place in h, dh ([nz, nr, nt] in Fortran order)
     function ucf_drawdown_grid_multi(plans, ngpu, nt, tD, sv, nr, rD, nz, zD, zLay, h, dh, stats) &
          & bind(C, name='ucf_drawdown_grid_multi') result(rc)
       import :: c_int, c_double, c_ptr, ucf_stats
       type(c_ptr), intent(in) :: plans(*)
       integer(c_int), value :: ngpu, nt, nr, nz
       real(c_double), intent(in) :: tD(*), rD(*), zD(*)
       integer(c_int), intent(in) :: sv(*), zLay(*)
       real(c_double), intent(out) :: h(*), dh(*)
       type(ucf_stats), intent(out) :: stats
       integer(c_int) :: rc
     end function ucf_drawdown_grid_multi

     ! the point-list counterpart: blocks of the list on plans(g)'s devices, h, dh [nz, npts] in Fortran order
     function ucf_drawdown_batch_multi(plans, ngpu, npts, tD, rD, sv, nz, zD, zLay, h, dh, stats) &
          & bind(C, name='ucf_drawdown_batch_multi') result(rc)
       import :: c_int, c_double, c_ptr, ucf_stats
       type(c_ptr), intent(in) :: plans(*)
       integer(c_int), value :: ngpu, npts, nz
       real(c_double), intent(in) :: tD(*), rD(*), zD(*)
       integer(c_int), intent(in) :: sv(*), zLay(*)
       real(c_double), intent(out) :: h(*), dh(*)
       type(ucf_stats), intent(out) :: stats
       integer(c_int) :: rc
     end function ucf_drawdown_batch_multi

     function ucf_screen_average(npts, zOrd, h, havg) bind(C, name='ucf_screen_average') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: npts, zOrd
       real(c_double), intent(in) :: h(*)
       real(c_double), intent(out) :: havg(*)
       integer(c_int) :: rc
     end function ucf_screen_average
  end interface

contains

  ! C string of ucf_last_error() as a Fortran string
  function ucf_error_message() result(msg)
    character(len=:), allocatable :: msg
    type(c_ptr) :: p
    character(kind=c_char), pointer :: s(:)
    integer :: n
    p = ucf_last_error()
    msg = ''
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, [512])
    n = 0
    do while (n < 512)
       if (s(n+1) == c_null_char) exit
       n = n + 1
    end do
    allocate(character(len=n) :: msg)
    msg = transfer(s(1:n), msg)
  end function ucf_error_message

end module ucf_binding
