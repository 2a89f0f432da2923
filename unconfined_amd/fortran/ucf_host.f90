! ucf_host.f90 -- thin Fortran host: deck in, drawdown rows out, all numerics on the GPU.
!
!   ucf_host <deck> [faithful|fast|header] [ngpu]          ("header": parameter echo only, needs no GPU)
!
! Does what `./unconfined <deck>` does for a time-series or contour deck (18-line format of
! input-explanation.txt), with the serial/OpenMP loop nest of driver.f90:100-232 replaced by
! ONE call through ISO_C_BINDING: ucf_drawdown_grid_multi, which cuts the rows of the i loop
! (driver.f90:100) into contiguous blocks over `ngpu` MI355X (default: every visible device, at
! most one per time row) and gathers the blocks into the result arrays.  The result file is the
! reference's: same '#' parameter echo (ucf_output, driver_io.f90:668-845) and rows in the same
! edit descriptors (constants.f90:72-73: ES14.07E2 / ES24.15E4), so outputs diff line by line.
! UCF_HOST_ONE_DEVICE=1 (rehearsal on a one-GPU box): all plans on device 0.
program ucf_host
  use, intrinsic :: iso_c_binding
  use ucf_binding
  use ucf_output
  implicit none

  type(ucf_params) :: P
  type(ucf_derived) :: D
  type(ucf_stats) :: st
  type(c_ptr) :: plan
  type(c_ptr), allocatable :: plans(:)
  type(run_shape) :: shape
  integer(c_int) :: ngpu, ndev, g, ngpu_asked
  logical :: header_only
  character(len=512) :: deckname, modearg, gpuarg, onedev, tfile, sfile, outname
  character(len=8192) :: line
  character(len=64) :: tok(256)
  integer :: ntok, quiet, zOrd, ios, mode, i, k, m, u
  logical :: dimless, timeseries, piezometer, compute
  real(c_double) :: tval, rval, zTop, zBot, sc
  integer(c_int) :: nt, nr, nz, rc
  integer :: minlog, maxlog, ncomp, nfileA, nfileB, nrc, nzc
  real(c_double) :: lo, hi, lo2, hi2
  real(c_double), allocatable :: t(:), r(:), z(:), tD(:), rD(:), zD(:), h(:), dh(:), ha(:), da(:)
  integer(c_int), allocatable :: sv(:), zLay(:)

  call get_command_argument(1, deckname)
  if (len_trim(deckname) == 0) deckname = 'input.dat'
  call get_command_argument(2, modearg)
  mode = 0
  if (trim(modearg) == 'fast') mode = 1
  call get_command_argument(3, gpuarg)
  ngpu = 0
  if (len_trim(gpuarg) > 0) read(gpuarg, *, iostat=ios) ngpu

  open(newunit=u, file=trim(deckname), status='old', action='read', iostat=ios)
  if (ios /= 0) call die('cannot open deck '//trim(deckname))

  ! ---- 18 records, leading tokens only; the rest of each record is commentary
  call rec(u, 5); read(tok(1),*) quiet; read(tok(2),*) P%model
  dimless = truth(tok(3)); timeseries = truth(tok(4)); piezometer = truth(tok(5))
  call rec(u, 1); P%Q = num(tok(1))
  call rec(u, 2); P%l = num(tok(1)); P%d = num(tok(2))
  call rec(u, 2); P%rw = num(tok(1)); P%rc = num(tok(2))
  call rec(u, 1); P%gammaSkin = num(tok(1))
  call rec(u, 1); read(tok(1),*) P%timeType
  P%timePar = 0.0_c_double; P%timeParExt = 0.0_c_double
  if (P%timeType > -1) then
     if (ntok < 3) call die('time behaviour needs two parameters')
     P%timePar(1) = num(tok(2)); P%timePar(2) = num(tok(3))
  else                                   ! -n / -(100+n): 2n+1 schedule parameters (driver_io.f90:119-127)
     k = -P%timeType
     if (k > 100) k = k - 100
     k = 2*k + 1
     if (k > 2*UCF_MAX_SCHEDULE+1 .or. ntok < 1 + k) call die('pumping schedule: wrong number of parameters')
     do i = 1, k
        P%timeParExt(i) = num(tok(1+i))
     end do
     P%timePar(1) = P%timeParExt(1); P%timePar(2) = P%timeParExt(2)
  end if
  call rec(u, 1); P%b = num(tok(1))
  call rec(u, 2); P%Kr = num(tok(1)); P%kappa = num(tok(2))
  call rec(u, 2); P%Ss = num(tok(1)); P%Sy = num(tok(2))
  call rec(u, 2); P%beta = num(tok(1)); read(tok(2),*) P%MoenchM
  P%MoenchAlpha = 0.0_c_double
  if (P%MoenchM > UCF_MAX_MOENCH) call die('too many Moench alphas')
  if (ntok < 2 + max(P%MoenchM, 0)) call die('Moench alphas missing on record 10')
  do i = 1, P%MoenchM
     P%MoenchAlpha(i) = num(tok(2+i))
  end do
  call rec(u, 7)
  P%ac = num(tok(1)); P%ak = num(tok(2)); P%psia = num(tok(3)); P%psik = num(tok(4)); P%usL = num(tok(5))
  read(tok(6),*) P%MNtype; read(tok(7),*) P%order
  call rec(u, 3); read(tok(1),*) P%M; P%alpha = num(tok(2)); P%tol = num(tok(3))
  call rec(u, 2); read(tok(1),*) P%k; read(tok(2),*) P%R
  call rec(u, 4); read(tok(1),*) P%j0s(1); read(tok(2),*) P%j0s(2); read(tok(3),*) P%nacc; read(tok(4),*) P%ord
  call rec(u, 2); tfile = tok(1); tval = num(tok(2))
  call rec(u, 2); sfile = tok(1); rval = num(tok(2))
  call rec(u, 5); zTop = num(tok(1)); zBot = num(tok(2)); read(tok(3),*) zOrd
  P%rwobs = num(tok(4)); P%sF = num(tok(5))
  call rec(u, 1); outname = tok(1)
  close(u)
  P%pad0 = 0; P%pad1 = 0

  rc = ucf_nondimensionalise(P, D)
  if (rc /= UCF_OK) call die('bad deck: '//ucf_error_message())
  header_only = (trim(modearg) == 'header')

  ! one plan per GPU (the plans are replicas: same parameters, each bound to its device)
  ndev = 0
  ngpu_asked = ngpu
  if (header_only) then
     ngpu = 0
     allocate(plans(0))
  else
  rc = ucf_device_count(ndev)
  if (rc /= UCF_OK) call die('ucf_device_count: '//ucf_error_message())
  if (ngpu <= 0) ngpu = ndev
  call get_environment_variable('UCF_HOST_ONE_DEVICE', onedev)
  if (ngpu > ndev .and. trim(onedev) /= '1') call die('more GPUs asked for than are visible')
  allocate(plans(ngpu))
  do g = 1, ngpu
     if (trim(onedev) == '1') then
        rc = ucf_plan_create_on(P, 0_c_int, plans(g))
     else
        rc = ucf_plan_create_on(P, g - 1, plans(g))
     end if
     if (rc /= UCF_OK) call die('ucf_plan_create_on: '//ucf_error_message())
     rc = ucf_plan_set_mode(plans(g), int(mode, c_int))
  end do
  plan = plans(1)
  rc = ucf_plan_derived(plan, D)
  end if

  ! ---- where and when (driver_io.f90:385-523)
  if (timeseries) then
     open(newunit=u, file=trim(tfile), status='old', action='read', iostat=ios)
     if (ios /= 0) call die('cannot open time file '//trim(tfile))
     call rec(u, 2); compute = truth(tok(1)); read(tok(2),*) nfileA
     call rec(u, 3); read(tok(1),*) minlog; read(tok(2),*) maxlog; read(tok(3),*) ncomp
     if (compute) then
        nt = ncomp
        allocate(t(nt))
        rc = ucf_logspace(int(minlog,c_int), int(maxlog,c_int), nt, t)
     else
        nt = nfileA
        allocate(t(nt))
        do i = 1, nt
           call rec(u, 1); t(i) = num(tok(1))
        end do
     end if
     close(u)
     ! the checks of driver_io.f90:355-394,443-449 (the reference prints ERROR and stops)
     if (zTop < zBot) call die('top of monitoring well screen must be at or above bottom')
     if (zTop > P%b .or. zBot < 0.0_c_double) call die('top of monitoring well screen must be above bottom and both between 0 and b')
     if (.not. piezometer .and. zOrd < 1) call die('# of quadrature points at monitoring location must be > 0')
     if (P%rwobs <= 0.0_c_double) call die('monitoring well radius must be >0')
     if (P%sF <= 0.0_c_double) call die('monitoring well shape factor must be >0')
     if (.not. rval > P%rw) call die('r must be > rw')
     if (any(t < 0.0_c_double)) call die('all times must be > 0')
     nr = 1
     allocate(r(1)); r(1) = rval
     if (piezometer) zOrd = 1
     nz = zOrd
     allocate(z(nz))
     rc = ucf_linspace(zBot, zTop, nz, z)
  else
     nt = 1
     allocate(t(1)); t(1) = tval
     open(newunit=u, file=trim(sfile), status='old', action='read', iostat=ios)
     if (ios /= 0) call die('cannot open space file '//trim(sfile))
     call rec(u, 3); compute = truth(tok(1)); read(tok(2),*) nfileA; read(tok(3),*) nfileB
     call rec(u, 3); lo = num(tok(1)); hi = num(tok(2)); read(tok(3),*) nrc
     call rec(u, 3); lo2 = num(tok(1)); hi2 = num(tok(2)); read(tok(3),*) nzc
     if (compute) then
        nr = nrc; nz = nzc
        allocate(r(nr), z(nz))
        rc = ucf_linspace(lo, hi, nr, r)
        rc = ucf_linspace(lo2, hi2, nz, z)
     else
        nr = nfileA; nz = nfileB
        allocate(r(nr), z(nz))
        read(u,*) r(1:nr)
        read(u,*) z(1:nz)
        if (any(r < P%rw)) call die('r must be >= rw')                                  ! driver_io.f90:505-507
     end if
     close(u)
     if (any(z < 0.0_c_double) .or. any(z > P%b)) call die('z must be in range 0<=>b')    ! :494-497,516-519
  end if

  if (header_only) then
     open(newunit=u, file=trim(outname), status='replace', action='write')
     shape%dimless = dimless; shape%timeseries = timeseries; shape%piezometer = piezometer
     shape%zOrd = zOrd; shape%nt = nt; shape%zTop = zTop; shape%zBot = zBot; shape%tval = tval
     shape%r = r; shape%z = z
     call echo_parameters(u, P, D, shape)
     close(u)
     stop
  end if

  allocate(tD(nt), rD(nr), zD(nz), sv(nt), zLay(nz), h(nz*nr*nt), dh(nz*nr*nt))
  tD = t / D%Tc
  rD = r / D%Lc
  zD = z / D%Lc
  rc = ucf_zlay(plan, nz, zD, zLay)
  rc = ucf_split_vector(plan, nt, tD, sv)

  ! ---- the hot path: one call instead of the OpenMP loop nest
  if (ngpu > nt) ngpu = nt
  rc = ucf_drawdown_grid_multi(plans, ngpu, nt, tD, sv, nr, rD, nz, zD, zLay, h, dh, st)
  if (rc /= UCF_OK) call die('ucf_drawdown_grid_multi: '//ucf_error_message())

  sc = D%Hc
  if (dimless) sc = 1.0_c_double

  open(newunit=u, file=trim(outname), status='replace', action='write')
  shape%dimless = dimless; shape%timeseries = timeseries; shape%piezometer = piezometer
  shape%zOrd = zOrd; shape%nt = nt; shape%zTop = zTop; shape%zBot = zBot; shape%tval = tval
  shape%r = r; shape%z = z
  call echo_parameters(u, P, D, shape)
  ! (what the reference's quiet > 0 progress lines would say goes to the terminal, not into the file)
  write(*,'(A,I0,A,6(1X,I0))') 'ucf_host: ', ngpu, ' GPU(s); in-band rules fired (nan_scrubbed zero_vectors '// &
       & 'wynn_truncated wynn_sentinel wynn_early_exit wynn_all_zero):', st%nan_scrubbed, st%zero_vectors, &
       & st%wynn_truncated, st%wynn_sentinel, st%wynn_early_exit, st%wynn_all_zero
  if (timeseries) then
     allocate(ha(nt), da(nt))
     if (nz > 1) then          ! driver.f90:234-243
        rc = ucf_screen_average(nt, nz, h, ha)
        rc = ucf_screen_average(nt, nz, dh, da)
     else
        ha = h(1:nt); da = dh(1:nt)
     end if
     do i = 1, nt
        if (dimless) then
           write(u,'(ES14.07E2,1X,2(ES24.15E4,1X))') tD(i), ha(i), da(i)
        else
           write(u,'(ES14.07E2,1X,2(ES24.15E4,1X))') t(i), ha(i)*sc, da(i)*sc
        end if
     end do
  else
     do k = 1, nr
        do m = 1, nz
           i = m + nz*(k-1)
           if (dimless) then
              write(u,'(2(ES14.07E2,1X),2(ES24.15E4,1X))') zD(m), rD(k), h(i), dh(i)
           else
              write(u,'(2(ES14.07E2,1X),2(ES24.15E4,1X))') z(m), r(k), h(i)*sc, dh(i)*sc
           end if
        end do
     end do
  end if
  close(u)
  do g = 1, size(plans)
     call ucf_plan_destroy(plans(g))
  end do

contains

  subroutine die(msg)
    character(*), intent(in) :: msg
    write(*,'(A)') 'ucf_host: ERROR '//msg
    stop 1
  end subroutine die

  ! read one record and split its first `need` blank-separated tokens
  subroutine rec(unit, need)
    integer, intent(in) :: unit, need
    integer :: p, q, ln, io
    read(unit,'(A)',iostat=io) line
    if (io /= 0) call die('unexpected end of file')
    ntok = 0
    ln = len_trim(line)
    p = 1
    do while (p <= ln .and. ntok < size(tok))
       do while (p <= ln)
          if (line(p:p) /= ' ' .and. line(p:p) /= achar(9)) exit
          p = p + 1
       end do
       if (p > ln) exit
       q = p
       do while (q <= ln)
          if (line(q:q) == ' ' .or. line(q:q) == achar(9)) exit
          q = q + 1
       end do
       ntok = ntok + 1
       tok(ntok) = line(p:q-1)
       p = q
    end do
    if (ntok < need) call die('record too short: '//trim(line))
  end subroutine rec

  function num(s) result(x)
    character(*), intent(in) :: s
    real(c_double) :: x
    integer :: io
    read(s,*,iostat=io) x
    if (io /= 0) call die('bad number '//trim(s))
  end function num

  function truth(s) result(b)
    character(*), intent(in) :: s
    logical :: b
    character :: c
    c = s(1:1)
    if (c == '.') c = s(2:2)
    b = (c == 'T' .or. c == 't')
  end function truth

end program ucf_host
