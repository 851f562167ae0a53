! ref_multirand_shim.F90 -- bind(C) entry points around the REFERENCE's own
! multirand module (compiled from /root/reference/src/multirand.F90 where it
! lies, see oracle/Makefile).  Test infrastructure only: it lets tests/ and the
! golden-vector generator call the real reference RNG to pin the restatement in
! pic1dp_oracle.c.  This file contains no reference source text.
module ref_multirand_shim
use iso_c_binding
use multirand
implicit none
contains

subroutine ref_multirand_init(al_int, seed_type, mype, warmup, selftest) &
    bind(C, name="ref_multirand_init")
  integer(c_int), value :: al_int, seed_type, mype, warmup, selftest
  call multirand_init(int(al_int), int(seed_type), int(mype), int(warmup), &
    selftest /= 0)
end subroutine

subroutine ref_multirand_int_array64(a, n) bind(C, name="ref_multirand_int_array64")
  integer(c_int64_t), value :: n
  integer(c_int64_t), intent(out) :: a(n)
  call multirand_int_array64(a)
end subroutine

subroutine ref_multirand_real_array64(a, n) bind(C, name="ref_multirand_real_array64")
  integer(c_int64_t), value :: n
  real(c_double), intent(out) :: a(n)
  call multirand_real_array64(a)
end subroutine

subroutine ref_multirand_gaussian_array64(a, n) &
    bind(C, name="ref_multirand_gaussian_array64")
  integer(c_int64_t), value :: n
  real(c_double), intent(out) :: a(n)
  call multirand_gaussian_array64(a)
end subroutine

end module ref_multirand_shim
