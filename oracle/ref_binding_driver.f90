! TEST INFRASTRUCTURE -- the reference-side binding of INTEGRATION.md section 2, compiled once.
!
! This is what a maintainer of the reference who keeps driver.f90 / driver_io.f90 would build: the reference's OWN
! modules (types, driver_io, constants: linked from oracle/_ref/<flavour>/, compiled there from /root/reference by
! oracle/Makefile -- nothing of theirs is copied) + the ISO_C_BINDING module that ships with the product
! (unconfined_amd/fortran/ucf_binding.f90).  The reference's read_input fills w, f, s, l, h, gl, ts (driver.f90:71); the
! mapping of those onto ucf_params is the snippet of INTEGRATION.md; the loop nest of driver.f90:100-232 is ONE call.
!
!   ref_binding_driver deck.in nondim        no GPU needed: prints, as hexadecimal bit patterns, every quantity of
!                                            driver_io.f90:531-567 as read_input left it in w/f/s next to what
!                                            ucf_nondimensionalise makes of the same deck (tests/test_ref_binding.py
!                                            requires equal bits)
!   ref_binding_driver deck.in gpu [mode]    the whole run: header by the reference's own writer, the loop nest by
!                                            ucf_drawdown_grid, rows in the reference's format (driver.f90:234-273)
program ref_binding_driver
  use, intrinsic :: iso_c_binding
  use types
  use driver_io, only : read_input, write_timeseries_header, write_contour_header
  use constants, only : DP, EP, RFMT, HFMT
  use ucf_binding
  implicit none

  type(invLaplace) :: l
  type(invHankel) :: h
  type(GaussLobatto) :: gl
  type(TanhSinh) :: ts
  type(well) :: w
  type(formation) :: f
  type(solution) :: s
  integer, parameter :: UNIT = 20

  type(ucf_params)  :: P
  type(ucf_derived) :: D
  type(ucf_stats)   :: st
  type(c_ptr)       :: plan
  real(c_double), allocatable :: hh(:), dd(:), tD(:), rD(:), zD(:), obs(:), dobs(:)
  integer(c_int), allocatable :: zl(:), sv(:)
  character(256) :: what, modearg
  integer :: i, k, m, n, rc, mode
  integer(c_int) :: nt, nr, nz

  call get_command_argument(2, what)
  if (len_trim(what) == 0) what = 'nondim'
  call get_command_argument(3, modearg)

  call read_input(w,f,s,l,h,gl,ts)                 ! the reference's own (driver.f90:71), deck named on the command line

  ! ---- INTEGRATION.md section 2: the deck values (dimensional, as read) into the POD block
  P%model = s%model;  P%MNtype = s%MNtype;  P%order = s%order
  P%timeType = l%timeType
  P%timePar = 0.0_c_double
  P%timeParExt = 0.0_c_double
  if (l%timeType > 0) then
     P%timePar = l%timePar(1:2)
  else
     n = size(l%timePar)
     P%timeParExt(1:n) = l%timePar(1:n)
  end if
  P%Q = w%Q;  P%l = w%l;  P%d = w%d;  P%rw = w%rw;  P%rc = w%rc;  P%gammaSkin = f%gammaSkin
  P%b = f%b;  P%Kr = f%Kr;  P%kappa = f%kappa;  P%Ss = f%Ss;  P%Sy = f%Sy;  P%beta = f%beta
  P%MoenchM = f%MoenchM;  P%pad0 = 0;  P%pad1 = 0
  P%MoenchAlpha = 0.0_c_double
  if (allocated(f%MoenchAlpha)) P%MoenchAlpha(1:f%MoenchM) = f%MoenchAlpha(1:f%MoenchM)
  P%ac = f%ac;  P%ak = f%ak;  P%psia = f%psia;  P%psik = f%psik;  P%usL = f%usL
  P%M = l%M;  P%alpha = l%alpha;  P%tol = l%tol
  P%k = ts%k;  P%R = ts%R;  P%j0s = h%j0s;  P%nacc = gl%nacc;  P%ord = gl%ord
  P%rwobs = s%rwobs;  P%sF = s%sF

  rc = ucf_nondimensionalise(P, D)
  if (rc /= UCF_OK) then
     write(*,'(A,I0,1X,A)') 'ucf_nondimensionalise failed: ', rc, trim(ucf_error_message())
     stop 2
  end if

  if (trim(what) == 'nondim') then
     ! name, the reference's bits, the library's bits
     call pair('Lc', s%Lc, D%Lc);  call pair('Tc', s%Tc, D%Tc);  call pair('Hc', s%Hc, D%Hc)
     call pair('sigma', f%MalamaSigma, D%sigma);  call pair('alphaD', f%alphaD, D%alphaD);  call pair('betaD', f%betaD, D%betaD)
     call pair('lD', w%lD, D%lD);  call pair('dD', w%dD, D%dD);  call pair('bD', w%bD, D%bD);  call pair('rDw', w%rDw, D%rDw)
     if (s%model == 2) call pair('rDwobs', s%rDwobs, D%rDwobs)
     if (s%model == 6) then
        call pair('acD', f%acD, D%acD);  call pair('akD', f%akD, D%akD);  call pair('lambdaD', f%lambdaD, D%lambdaD)
        call pair('psiaD', f%psiaD, D%psiaD);  call pair('psikD', f%psikD, D%psikD);  call pair('usLD', f%usLD, D%usLD)
        call pair('b1', f%b1, D%b1);  call pair('PsiD', f%PsiD, D%PsiD)
     end if
     if (s%model == 3) then
        do m = 1, f%MoenchM
           call pair('MoenchGamma', f%MoenchGamma(m), D%MoenchGamma(m))
        end do
     end if
     write(*,'(A,4(1X,I0))') 'sizes', 2*l%M + 1, 2**ts%k - 1, size(h%j0z), D%nj0z
     ! what the one call receives from read_input (driver_io.f90:397-664): times, radii, depths, layers, split indices
     do i = 1, s%nt
        write(*,'(A,1X,Z16.16,1X,I0)') 'tD', transfer(s%tD(i), 1_c_int64_t), h%sv(i)
     end do
     do i = 1, s%nr
        write(*,'(A,1X,Z16.16,1X,I0)') 'rD', transfer(s%rD(i), 1_c_int64_t), 0
     end do
     do i = 1, s%nz
        write(*,'(A,1X,Z16.16,1X,I0)') 'zD', transfer(s%zD(i), 1_c_int64_t), s%zLay(i)
     end do
     stop
  end if

  ! ---- the whole run on the GPU
  mode = 0
  if (trim(modearg) == 'fast') mode = 1
  rc = ucf_plan_create(P, plan)
  if (rc /= UCF_OK) then
     write(*,'(A,I0,1X,A)') 'ucf_plan_create failed: ', rc, trim(ucf_error_message())
     stop 3
  end if
  rc = ucf_plan_set_mode(plan, int(mode, c_int))
  nt = s%nt;  nr = s%nr;  nz = s%nz
  allocate(hh(nz*nr*nt), dd(nz*nr*nt), zl(nz), sv(nt), tD(nt), rD(nr), zD(nz), obs(1), dobs(1))
  tD = s%tD;  rD = s%rD;  zD = s%zD;  zl = s%zLay;  sv = h%sv
  ! replaces driver.f90:100-232 (both loops, all six OpenMP regions)
  rc = ucf_drawdown_grid(plan, nt, tD, sv, nr, rD, nz, zD, zl, hh, dd, st)
  if (rc /= UCF_OK) then
     write(*,'(A,I0,1X,A)') 'ucf_drawdown_grid failed: ', rc, trim(ucf_error_message())
     stop 4
  end if

  ! the writers of driver.f90:93-97 and :234-273 stay what they are: the reference's own header routines ...
  l%np = 2*l%M + 1
  ts%N = 2**ts%k - 1
  if (s%timeSeries) then
     call write_timeseries_header(w,f,s,l,h,gl,ts,UNIT)
  else
     call write_contour_header(w,f,s,l,h,gl,ts,UNIT)
  end if
  ! ... and its row formats; hh/dd hold totint/totintd of every (m,k,i) as (nz,nr,nt)
  do i = 1, s%nt
     do k = 1, s%nr
        n = ((i - 1)*s%nr + (k - 1))*s%nz
        if (s%timeseries) then
           rc = ucf_screen_average(1_c_int, merge(1_c_int, int(s%zOrd, c_int), s%piezometer .or. s%zOrd <= 1), hh(n+1:n+s%nz), obs)
           rc = ucf_screen_average(1_c_int, merge(1_c_int, int(s%zOrd, c_int), s%piezometer .or. s%zOrd <= 1), dd(n+1:n+s%nz), dobs)
           if (s%dimless) then
              write (UNIT,'('//RFMT//',1X,2('//HFMT//',1X))') s%tD(i), obs(1), dobs(1)
           else
              write (UNIT,'('//RFMT//',1X,2('//HFMT//',1X))') s%t(i), obs(1)*s%Hc, dobs(1)*s%Hc
           end if
        else
           do m = 1, s%nz
              if (s%dimless) then
                 write (UNIT,'(2('//RFMT//',1X),2('//HFMT//',1X))') s%zD(m), s%rD(k), hh(n+m), dd(n+m)
              else
                 write (UNIT,'(2('//RFMT//',1X),2('//HFMT//',1X))') s%z(m), s%r(k), hh(n+m)*s%Hc, dd(n+m)*s%Hc
              end if
           end do
        end if
     end do
  end do
  close(UNIT)
  call ucf_plan_destroy(plan)

contains
  subroutine pair(name, a, b)
    character(*), intent(in) :: name
    real(DP), intent(in) :: a
    real(c_double), intent(in) :: b
    write(*,'(A,1X,Z16.16,1X,Z16.16)') name, transfer(a, 1_c_int64_t), transfer(b, 1_c_int64_t)
  end subroutine pair
end program ref_binding_driver
