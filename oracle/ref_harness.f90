! TEST INFRASTRUCTURE (oracle side) -- not part of the shipped product.
!
! ref_harness: a small stand-alone Fortran program (our own code) that links
! against the *unmodified* reference modules compiled by oracle/Makefile into
! oracle/_ref/ and dumps, bit-exactly (IEEE-754 hex), the outputs of the
! reference's public procedures for inputs chosen by oracle/gen_golden.py:
!
!   read_input            (/root/reference/driver_io.f90:30)   -> "params"
!   deHoog_pvalues        (/root/reference/invlap.f90:154)     -> "pvalues"
!   lap_hank_soln         (/root/reference/laplace_hankel_solutions.f90:30) -> "soln"
!   tanh_sinh_setup       (/root/reference/integration.f90:31) -> "tanhsinh"
!   gauss_lobatto_setup   (/root/reference/integration.f90:70) -> "gausslobatto"
!   wynn_epsilon          (/root/reference/integration.f90:125)-> "wynn"
!   extraptozero          (/root/reference/integration.f90:192)-> "extrap"
!   deHoog_invlap         (/root/reference/invlap.f90:34)      -> "dehoog"
!   cbesk                 (/root/reference/cbessel.f90:877)    -> "cbesk"
!
! usage:  ref_harness <deck> < commands > dump
! The deck is parsed by the reference's own read_input (it takes argv(1)), so
! every derived quantity is the reference's.  Floating point numbers travel as
! 16-digit hex bit patterns in both directions, so nothing is lost in text.
program ref_harness
  use constants, only : DP, EP
  use types
  use driver_io, only : read_input
  use laplace_hankel_solutions, only : lap_hank_soln
  use invlap, only : deHoog_invlap, deHoog_pvalues
  use integration, only : tanh_sinh_setup, gauss_lobatto_setup, wynn_epsilon, extraptozero
  use cbessel, only : cbesk
  implicit none

  type(invLaplace) :: l
  type(invHankel) :: h
  type(GaussLobatto) :: gl
  type(TanhSinh) :: ts
  type(well) :: w
  type(formation) :: f
  type(solution) :: s

  character(32) :: cmd
  character(256) :: line
  character(32) :: tok(8)
  integer :: ntok
  integer :: ios, i, n, k, m, ord
  real(DP) :: a, rD, tee, t, arg
  complex(EP), allocatable :: fp(:,:), vec(:), yv(:)
  real(EP), allocatable :: xv(:)
  complex(EP) :: acc
  real(EP) :: ft
  type(TanhSinh) :: t2
  type(GaussLobatto) :: g2
  complex(DP) :: zk, kk(2)
  integer :: nzk, ierrk

  call read_input(w,f,s,l,h,gl,ts)
  l%np = 2*l%M + 1
  allocate(l%p(l%np))
  l%p = (1.0_EP, 0.0_EP)

  do
     read(*,'(A)',iostat=ios) line
     if (ios /= 0) exit
     call split(line)
     if (ntok == 0) cycle
     cmd = tok(1)
     select case (trim(cmd))

     case ('params')
        call dump_params()

     case ('setlap')           ! setlap M alpha(hex) tol(hex)
        read(tok(2),*) m
        l%M = m
        l%alpha = hex2r(tok(3))
        l%tol = hex2r(tok(4))
        l%np = 2*m + 1
        if (allocated(l%p)) deallocate(l%p)
        allocate(l%p(l%np))
        l%p = (1.0_EP, 0.0_EP)

     case ('pvalues')          ! pvalues tee(hex)
        tee = hex2r(tok(2))
        l%p(1:l%np) = deHoog_pvalues(tee, l)
        write(*,'(A,1X,I0)') 'pvalues', l%np
        call dump_cvec(l%p, l%np)

     case ('setp')             ! setp i re(hex) im(hex)
        read(tok(2),*) i
        l%p(i) = cmplx(hex2r(tok(3)), hex2r(tok(4)), EP)

     case ('soln')             ! soln a(hex) rD(hex)  -- uses the current l%p
        a = hex2r(tok(2))
        rD = hex2r(tok(3))
        allocate(fp(l%np, s%nz))
        fp = lap_hank_soln(a, rD, l%np, s%nz, w, f, s, l)
        write(*,'(A,2(1X,I0))') 'soln', l%np, s%nz
        do k = 1, s%nz
           call dump_cvec(fp(:,k), l%np)
        end do
        deallocate(fp)

     case ('tanhsinh')         ! tanhsinh k arg(hex)
        read(tok(2),*) k
        arg = hex2r(tok(3))
        n = 2**k - 1
        allocate(t2%Q(1))
        allocate(t2%Q(1)%w(n), t2%Q(1)%a(n))
        call tanh_sinh_setup(t2, k, arg, 1)
        write(*,'(A,1X,I0)') 'tanhsinh', n
        call dump_rvec(t2%Q(1)%w, n)
        call dump_rvec(t2%Q(1)%a, n)
        deallocate(t2%Q)

     case ('gausslobatto')     ! gausslobatto ord
        read(tok(2),*) ord
        g2%ord = ord
        allocate(g2%x(ord-2), g2%w(ord-2))
        call gauss_lobatto_setup(g2)
        write(*,'(A,1X,I0)') 'gausslobatto', ord-2
        call dump_rvec(g2%x, ord-2)
        call dump_rvec(g2%w, ord-2)
        deallocate(g2%x, g2%w)

     case ('wynn')             ! wynn n ; then n lines "re(hex) im(hex)"
        read(tok(2),*) n
        allocate(vec(n))
        call read_cvec(vec, n)
        acc = wynn_epsilon(vec, 0)
        write(*,'(A)') 'wynn'
        call dump_cvec([acc], 1)
        deallocate(vec)

     case ('extrap')           ! extrap n ; n lines x(hex) ; n lines re im
        read(tok(2),*) n
        allocate(xv(n), yv(n))
        do i = 1, n
           read(*,'(A)') line
           call split(line)
           xv(i) = hex2r(tok(1))
        end do
        call read_cvec(yv, n)
        acc = extraptozero(xv, yv)
        write(*,'(A)') 'extrap'
        call dump_cvec([acc], 1)
        deallocate(xv, yv)

     case ('dehoog')           ! dehoog t(hex) tee(hex) ; then 2M+1 lines re im
        t = hex2r(tok(2))
        tee = hex2r(tok(3))
        allocate(vec(l%np))
        call read_cvec(vec, l%np)
        ft = deHoog_invlap(t, tee, vec, l)
        write(*,'(A)') 'dehoog'
        write(*,'(Z16.16)') r2hex(ft)
        deallocate(vec)

     case ('cbesk')            ! cbesk re(hex) im(hex): K0,K1 by the reference's Amos routine (cbessel.f90:877)
        zk = cmplx(hex2r(tok(2)), hex2r(tok(3)), DP)
        kk = (0.0_DP, 0.0_DP)
        call cbesk(z=zk, fnu=0.0_DP, kode=1, n=2, cy=kk, nz=nzk, ierr=ierrk)
        write(*,'(A,2(1X,I0))') 'cbesk', nzk, ierrk
        call dump_cvec(cmplx(kk, kind=EP), 2)

     case default
        write(*,'(A)') 'unknown '//trim(cmd)
        stop 2
     end select
  end do

contains

  subroutine split(str)
    ! whitespace tokeniser (at most 8 tokens)
    character(*), intent(in) :: str
    integer :: p, q, ln
    ntok = 0
    ln = len_trim(str)
    p = 1
    do while (p <= ln .and. ntok < 8)
       do while (p <= ln)
          if (str(p:p) /= ' ') exit
          p = p + 1
       end do
       if (p > ln) exit
       q = p
       do while (q <= ln)
          if (str(q:q) == ' ') exit
          q = q + 1
       end do
       ntok = ntok + 1
       tok(ntok) = str(p:q-1)
       p = q
    end do
  end subroutine split

  function hex2r(c) result(x)
    character(*), intent(in) :: c
    integer(8) :: b
    real(DP) :: x
    read(c,'(Z16)') b
    x = transfer(b, x)
  end function hex2r

  function r2hex(x) result(b)
    real(DP), intent(in) :: x
    integer(8) :: b
    b = transfer(x, b)
  end function r2hex

  subroutine read_cvec(v, nn)
    integer, intent(in) :: nn
    complex(EP), intent(out) :: v(nn)
    integer :: ii
    do ii = 1, nn
       read(*,'(A)') line
       call split(line)
       v(ii) = cmplx(hex2r(tok(1)), hex2r(tok(2)), EP)
    end do
  end subroutine read_cvec

  subroutine dump_cvec(v, nn)
    integer, intent(in) :: nn
    complex(EP), intent(in) :: v(nn)
    integer :: ii
    do ii = 1, nn
       write(*,'(Z16.16,1X,Z16.16)') r2hex(real(v(ii),DP)), r2hex(real(aimag(v(ii)),DP))
    end do
  end subroutine dump_cvec

  subroutine dump_rvec(v, nn)
    integer, intent(in) :: nn
    real(EP), intent(in) :: v(nn)
    integer :: ii
    do ii = 1, nn
       write(*,'(Z16.16)') r2hex(v(ii))
    end do
  end subroutine dump_rvec

  subroutine dump_r(name, x)
    character(*), intent(in) :: name
    real(DP), intent(in) :: x
    write(*,'(A,1X,Z16.16)') name, r2hex(x)
  end subroutine dump_r

  subroutine dump_i(name, iv)
    character(*), intent(in) :: name
    integer, intent(in) :: iv
    write(*,'(A,1X,I0)') name, iv
  end subroutine dump_i

  subroutine dump_params()
    integer :: ii
    write(*,'(A)') 'params'
    call dump_i('model', s%model)
    call dump_i('MNtype', s%MNtype)
    call dump_i('order', s%order)
    call dump_i('dimless', merge(1,0,s%dimless))
    call dump_i('timeseries', merge(1,0,s%timeseries))
    call dump_i('piezometer', merge(1,0,s%piezometer))
    call dump_i('nt', s%nt)
    call dump_i('nr', s%nr)
    call dump_i('nz', s%nz)
    call dump_i('zOrd', s%zOrd)
    call dump_i('M', l%M)
    call dump_i('timeType', l%timeType)
    call dump_i('k', ts%k)
    call dump_i('R', ts%R)
    call dump_i('j0s1', h%j0s(1))
    call dump_i('j0s2', h%j0s(2))
    call dump_i('nacc', gl%nacc)
    call dump_i('ord', gl%ord)
    call dump_i('MoenchM', f%MoenchM)
    call dump_r('alpha', l%alpha)
    call dump_r('tol', l%tol)
    call dump_r('timePar1', real(l%timePar(1),DP))
    call dump_r('timePar2', real(l%timePar(2),DP))
    call dump_r('Lc', s%Lc)
    call dump_r('Tc', s%Tc)
    call dump_r('Hc', s%Hc)
    call dump_r('lD', w%lD)
    call dump_r('dD', w%dD)
    call dump_r('bD', w%bD)
    call dump_r('rDw', w%rDw)
    call dump_r('l', w%l)
    call dump_r('d', w%d)
    call dump_r('b', f%b)
    call dump_r('Kr', f%Kr)
    call dump_r('kappa', f%kappa)
    call dump_r('Ss', f%Ss)
    call dump_r('Sy', f%Sy)
    call dump_r('beta', f%beta)
    call dump_r('sigma', f%MalamaSigma)
    call dump_r('alphaD', f%alphaD)
    call dump_r('betaD', f%betaD)
    call dump_r('ac', f%ac)
    call dump_r('ak', f%ak)
    call dump_r('psia', f%psia)
    call dump_r('psik', f%psik)
    call dump_r('acD', f%acD)
    call dump_r('akD', f%akD)
    call dump_r('lambdaD', f%lambdaD)
    call dump_r('psiaD', f%psiaD)
    call dump_r('psikD', f%psikD)
    call dump_r('usLD', f%usLD)
    call dump_r('b1', f%b1)
    call dump_r('PsiD', f%PsiD)
    call dump_r('rDwobs', s%rDwobs)
    call dump_r('sF', s%sF)
    write(*,'(A,1X,I0)') 'MoenchGamma', f%MoenchM
    do ii = 1, f%MoenchM
       write(*,'(Z16.16)') r2hex(f%MoenchGamma(ii))
    end do
    write(*,'(A,1X,I0)') 'j0z', size(h%j0z)
    call dump_rvec(h%j0z, size(h%j0z))
    write(*,'(A,1X,I0)') 'sv', s%nt
    do ii = 1, s%nt
       write(*,'(I0)') h%sv(ii)
    end do
    write(*,'(A,1X,I0)') 't', s%nt
    call dump_rvec(real(s%t,EP), s%nt)
    write(*,'(A,1X,I0)') 'tD', s%nt
    call dump_rvec(real(s%tD,EP), s%nt)
    write(*,'(A,1X,I0)') 'rD', s%nr
    call dump_rvec(real(s%rD,EP), s%nr)
    write(*,'(A,1X,I0)') 'zD', s%nz
    call dump_rvec(real(s%zD,EP), s%nz)
    write(*,'(A,1X,I0)') 'zLay', s%nz
    do ii = 1, s%nz
       write(*,'(I0)') s%zLay(ii)
    end do
    write(*,'(A)') 'endparams'
  end subroutine dump_params

end program ref_harness
