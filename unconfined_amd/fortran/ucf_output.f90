! ucf_output.f90 -- the '#' parameter echo at the head of a result file, in the layout of the files that
! `./unconfined deck` writes (reference driver_io.f90:668-758 time series, :760-845 contour maps), so that a result of
! the Fortran host diffs line by line against the reference's.  Formatting only; the same layout as
! unconfined_amd/output.py, which tests/test_output_format.py pins byte for byte against files written by the
! reference binary (tests/test_fortran_host.py compares this module's header with it).
module ucf_output
  use, intrinsic :: iso_c_binding
  use ucf_binding, only : ucf_params, ucf_derived
  implicit none
  private
  public :: echo_parameters, run_shape

  character(*), parameter :: NUM = 'ES14.07E2'          ! constants.f90:72 (RFMT)

  ! what kind of run the deck asks for and where it observes (filled by the host while it reads the deck)
  type :: run_shape
     logical :: dimless = .false., timeseries = .true., piezometer = .true.
     integer :: zOrd = 1, nt = 0
     real(c_double) :: zTop = 0, zBot = 0, tval = 0
     real(c_double), allocatable :: r(:), z(:)
  end type run_shape

  character(15), parameter :: model_name(0:6) = [character(15) :: 'Theis', 'Hantush', 'Hantush w/ stor', 'Moench', &
       & 'Malama full pen', 'Malama part pen', 'Mishra/Neuman']

contains

  pure function behaviour_text(tt) result(s)          ! types.f90:66-75
    integer, intent(in) :: tt
    character(len=:), allocatable :: s
    select case (tt)
    case (1); s = 'step on; tpar(1) = on time; tpar(2) not used'
    case (2); s = 'finite pulse; tpar(1:2) = on/off time'
    case (3); s = 'infinitessimal pulse; tpar(1) = pulse location; tpar(2) not used'
    case (4); s = 'stairs; tpar(1) = time step (Q increase by integer multiples); tpar(2) = off time'
    case (5, 8); s = 'rectified square wave; tpar(1) = 1/2 period of wave; tpar(2) = start time'
    case (6); s = 'cos(omega*t); tpar(1) = omega; tpar(2) = start time'
    case (7); s = 'rectified triangular wave; tpar(1) = 1/4 period of wave; tpar(2) = start time'
    case default; s = 'piecewise constant rate (n steps); tpar(1:n)=ti; tpar(n+1)=tfinal; tpar(n+2:)=Q'
    end select
  end function behaviour_text

  ! label followed by numbers separated by one blank (the reference's n(RFMT,1X) groups leave a trailing blank,
  ! which a formatted record drops again: nothing to reproduce there)
  subroutine put(u, label, v)
    integer, intent(in) :: u
    character(*), intent(in) :: label
    real(c_double), intent(in) :: v(:)
    character(len=32) :: f
    if (size(v) == 0) then
       write(u, '(A)') label
       return
    end if
    write(f, '(A,I0,A)') '(A,', size(v), '('//NUM//',1X))'
    write(u, f) label, v
  end subroutine put

  subroutine echo_parameters(u, P, D, S)
    integer, intent(in) :: u
    type(ucf_params), intent(in) :: P
    type(ucf_derived), intent(in) :: D
    type(run_shape), intent(in) :: S
    character(len=:), allocatable :: kr_label, skin_label, name
    integer :: npar, nseg
    logical :: ts
    ts = S%timeseries
    name = trim(model_name(P%model))
    write(u, '(A)') '# -*-auto-revert-*-'
    if (ts) then
       write(u, '(A,I0,1X,A,I0)') '# model, EP precision :: ', P%model, name//', ', 8
       write(u, '(A,3(L1,1X))') '# dimensionless?, timeseries?, piezometer? :: ', S%dimless, S%timeseries, S%piezometer
       kr_label = '# Kr,kappa (kappa=Kz/Kr) :: '; skin_label = '# gamma (dimensionless skin) :: '
    else
       write(u, '(A,I0,1X,A,I0)') '# model, EP :: ', P%model, name//', ', 8
       write(u, '(A,2(L1,1X))') '# dimensionless?, timeseries? :: ', S%dimless, S%timeseries
       kr_label = '# Kr,kappa (Kz/Kr) :: '; skin_label = '# gamma (dimless skin) :: '
    end if
    call put(u, '# Q (volumetric pumping rate) :: ', [P%Q])
    call put(u, '# b (initial sat thickness) :: ', [P%b])
    call put(u, '# l,d (screen bot & top) :: ', [P%l, P%d])
    call put(u, '# rw,rc (well/casing radii) :: ', [P%rw, P%rc])
    call put(u, kr_label, [P%Kr, P%kappa])
    call put(u, '# Ss,Sy :: ', [P%Ss, P%Sy])
    call put(u, skin_label, [P%gammaSkin])
    if (P%timeType > -1) then
       call put(u, '# pumping well time behavior :: '//itoa(P%timeType)//behaviour_text(P%timeType), P%timePar)
    else
       nseg = -P%timeType
       if (nseg > 100) nseg = nseg - 100
       npar = 2*nseg + 1
       call put(u, '# pumping well time behavior :: '//itoa(P%timeType)//behaviour_text(P%timeType), P%timeParExt(1:npar))
    end if
    call put(u, '# deHoog M, alpha, tol :: '//itoa(P%M), [P%alpha, P%tol])
    write(u, '(A,2(I0,1X))') '# tanh-sinh: k, n extrapolation steps :: ', P%k, P%R
    write(u, '(A,4(I0,1X))') '# GLquad: J0 split, n 0-accel, GL-order :: ', P%j0s, P%nacc, P%ord
    if (ts) then
       if (S%piezometer) then
          call put(u, '# point obs piezometer r,rD,z,zD :: ', [S%r(1), S%r(1)/D%Lc, S%z(1), S%z(1)/D%Lc])
       else
          write(u, '(A,3('//NUM//',1X),I0)') '# screened obs well r,zTop,zBot,zOrd :: ', S%r(1), S%zTop, S%zBot, S%zOrd
          call put(u, '# screened obs well rW,shape factor :: ', [P%rwobs, P%sF])
       end if
    else
       call put(u, '# num r locations, rlocs :: '//itoa(size(S%r))//' ', S%r)
       call put(u, '# num z locations, zlocs :: '//itoa(size(S%z))//' ', S%z)
       call put(u, '# time, tD :: ', [S%tval, S%tval/D%Tc])
    end if
    select case (P%model)
    case (4, 5)
       call put(u, '# Malama beta linearization parameter :: ', [P%beta])
    case (6)
       call put(u, '# Mishra/Neuman ac,ak,psia,psik,b1 ::', [D%ac_eff, P%ak, P%psia, P%psik, D%b1])
       if (P%MNtype == 2) then
          if (ts) then
             call put(u, '# Mishra/Neuman vadose zone finite-difference order, finite-difference spacing ::'// &
                  & itoa(P%order)//' ', [P%usL/(P%order - 1)])
          else
             call put(u, '# Mishra/Neuman finite-difference order, finite-difference mesh spacing ::'// &
                  & itoa(P%order)//' ', [P%usL/(P%order - 1)])
          end if
       else if (P%MNtype == 1 .and. ts) then
          write(u, '(A)') "# NB: Malama's Mishra/Neuman implementation (1) assumes ac=ak and fully penetrating "// &
               & 'pumping well without wellbore storage'
       end if
    end select          ! (model 3: the reference echoes the Moench coefficients to the terminal, not the file)
    if (ts) then
       write(u, '(A,I0)') '# times :: ', S%nt
       call put(u, '# characteristic length, time :: ', [D%Lc, D%Tc])
       if (.not. S%dimless) write(u, '(A,'//NUM//')') '# characteristic head ::', D%Hc
       write(u, '(A)') '#'
       if (S%dimless) then
          write(u, '(A)') '#     t_D              '//name//'             t*dh/d(log(t))'
       else
          write(u, '(A)') '#     t                '//name//'             t*dh/d(log(t))'
       end if
       write(u, '(A)') '#'//repeat('-', 63)
    else
       write(u, '(A)') '#'
       if (S%dimless) then
          write(u, '(A)') '#     z_D           r_D           '//name//'          t*dh/d(log(t))'
       else
          write(u, '(A)') '#      z            r             '//name//'          t*dh/d(log(t))'
       end if
       write(u, '(A)') '#'//repeat('-', 76)
    end if
  end subroutine echo_parameters

  pure function itoa(i) result(s)
    integer, intent(in) :: i
    character(len=:), allocatable :: s
    character(len=16) :: b
    write(b, '(I0)') i
    s = trim(b)
  end function itoa

end module ucf_output
