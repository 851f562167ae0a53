! host_input.F90 -- compile-time parameters of the Fortran host, under the
! reference's names (input_* of src/pic1dp_input.F90:32-256) so that a user's
! parameter edits carry over one to one.  input_fill() copies them into the
! run-time struct of the C ABI.  A few sizes can be overridden from the
! environment (PIC1DP_NPARTICLE, PIC1DP_NX, PIC1DP_TIME_MAX, PIC1DP_SEED_TYPE)
! so that tests can run a small case without recompiling.
module pic1dp_host_input
use iso_c_binding
use pic1dp_hip
implicit none

integer, parameter :: kpr = c_double

integer(c_int32_t), parameter :: input_ntime_max = 900000
real(kpr), parameter :: input_time_max = 500.0_kpr
integer(c_int32_t), parameter :: input_linear = 0
real(kpr), parameter :: input_lx = 2.0_kpr * 3.1415926535897932384626_kpr / 0.36_kpr
integer(c_int32_t), parameter :: input_iptcldist = 3
integer(c_int32_t), parameter :: input_nspecies = 1
real(kpr), dimension(input_nspecies), parameter :: &
  input_species_charge = (/ -1.0_kpr /), input_species_mass = (/ 1.0_kpr /), &
  input_species_temperature = (/ 1.0_kpr /), input_species_temperature2 = (/ 1.0_kpr /), &
  input_species_density = (/ 0.9_kpr /), input_species_v0 = (/ 5.0_kpr /)
integer(c_int32_t), parameter :: input_nmode = 1
integer(c_int32_t), dimension(0 : input_nmode - 1), parameter :: input_modes = (/ 1 /)
integer(c_int32_t), parameter :: input_init_nmode = 1
integer(c_int32_t), dimension(0 : input_init_nmode - 1), parameter :: input_init_mode = (/ 1 /)
real(kpr), dimension(0 : input_init_nmode - 1), parameter :: &
  input_init_mode_cos = (/ 0.00_kpr /), input_init_mode_sin = (/ 1e-5_kpr /)
integer(c_int32_t), parameter :: input_deltaf = 1
real(kpr), parameter :: input_dt = 0.05_kpr
integer(c_int64_t), parameter :: input_nparticle_max = 6400000_c_int64_t
integer(c_int64_t), dimension(input_nspecies), parameter :: &
  input_species_nparticle_init = input_nparticle_max
integer(c_int32_t), parameter :: input_imarker = 2
real(kpr), parameter :: input_v_max = 8.0_kpr
integer(c_int32_t), parameter :: input_nx = 192
integer(c_int32_t), parameter :: input_nv = 128
integer(c_int32_t), parameter :: input_iptclshape = 4
! marker optimisation: counts, times and thresholds (the lists are implied-do
! expressions of the counts, like the reference's)
integer(c_int32_t), parameter :: input_nmerge = 0
integer(c_int32_t), parameter :: input_nremove = 0
integer(c_int32_t), parameter :: input_typeremove = 2
real(kpr), parameter :: input_remove_frac = 0.9_kpr
integer(c_int32_t), parameter :: input_nsplit = 0
integer(c_int32_t), parameter :: input_split_ngroup = 5
real(kpr), parameter :: input_split_dv_sig_frac = 0.1_kpr
integer(c_int32_t), parameter :: input_multirand_al_int = 3
! the reference ships 3 (/dev/urandom); 1 = constant seeds, reproducible
integer(c_int32_t), parameter :: input_multirand_seed_type = 1
integer(c_int32_t), parameter :: input_multirand_warmup = 5
logical, parameter :: input_multirand_selftest = .true.
integer(c_int32_t), parameter :: input_verbosity = 1
real(kpr), parameter :: input_output_interval = 0.5_kpr
integer(c_int32_t), parameter :: input_nx_opd = 64
integer(c_int32_t), parameter :: input_nv_opd = 64

contains

subroutine env_int64(name, val)
  character(len=*), intent(in) :: name
  integer(c_int64_t), intent(inout) :: val
  character(len=64) :: buf
  integer :: stat, ios
  integer(c_int64_t) :: tmp
  call get_environment_variable(name, buf, status=stat)
  if (stat /= 0) return
  read (buf, *, iostat=ios) tmp
  if (ios == 0) val = tmp
end subroutine env_int64

subroutine env_real(name, val)
  character(len=*), intent(in) :: name
  real(kpr), intent(inout) :: val
  character(len=64) :: buf
  integer :: stat, ios
  real(kpr) :: tmp
  call get_environment_variable(name, buf, status=stat)
  if (stat /= 0) return
  read (buf, *, iostat=ios) tmp
  if (ios == 0) val = tmp
end subroutine env_real

subroutine input_fill(inp)
  type(pic1dp_input_t), intent(out) :: inp
  integer(c_int64_t) :: n, nx, seed
  integer :: s
  call pic1dp_hip_check(pic1dp_hip_input_defaults(inp), 'input_defaults')
  if (pic1dp_hip_input_size() /= int(c_sizeof(inp), c_int)) then
    write (*, '(a)') 'pic1dp_host: struct pic1dp_input layout differs from the library'
    stop 1
  end if
  inp%ntime_max = input_ntime_max
  inp%time_max = input_time_max
  inp%linear = input_linear
  inp%lx = input_lx
  inp%iptcldist = input_iptcldist
  inp%nspecies = input_nspecies
  inp%species_charge(1 : input_nspecies) = input_species_charge
  inp%species_mass(1 : input_nspecies) = input_species_mass
  inp%species_temperature(1 : input_nspecies) = input_species_temperature
  inp%species_temperature2(1 : input_nspecies) = input_species_temperature2
  inp%species_density(1 : input_nspecies) = input_species_density
  inp%species_v0(1 : input_nspecies) = input_species_v0
  inp%nmode = input_nmode
  inp%modes(1 : input_nmode) = input_modes
  inp%init_nmode = input_init_nmode
  inp%init_mode(1 : input_init_nmode) = input_init_mode
  inp%init_mode_cos(1 : input_init_nmode) = input_init_mode_cos
  inp%init_mode_sin(1 : input_init_nmode) = input_init_mode_sin
  inp%deltaf = input_deltaf
  inp%dt = input_dt
  inp%nparticle_max = input_nparticle_max
  inp%species_nparticle_init(1 : input_nspecies) = input_species_nparticle_init
  inp%imarker = input_imarker
  inp%v_max = input_v_max
  inp%nx = input_nx
  inp%nv = input_nv
  inp%iptclshape = input_iptclshape
  inp%multirand_al_int = input_multirand_al_int
  inp%multirand_seed_type = input_multirand_seed_type
  inp%multirand_warmup = input_multirand_warmup
  inp%multirand_selftest = merge(1, 0, input_multirand_selftest)
  inp%output_interval = input_output_interval
  inp%nx_opd = input_nx_opd
  inp%nv_opd = input_nv_opd
  inp%nmerge = input_nmerge
  inp%nremove = input_nremove
  inp%nsplit = input_nsplit
  inp%typeremove = input_typeremove
  inp%remove_frac = input_remove_frac
  inp%split_ngroup = input_split_ngroup
  inp%split_dv_sig_frac = input_split_dv_sig_frac
  do s = 1, input_nmerge
    inp%tmerge(s) = 50.0_kpr + s * 0.5_kpr
    inp%thshmerge(s) = 0.1_kpr / max(input_nmerge, 1) * real(s, kpr)
  end do
  do s = 1, input_nremove
    inp%tremove(s) = 50.0_kpr + s * 0.5_kpr
    inp%thshremove(s) = 0.1_kpr / max(input_nremove, 1) * real(s, kpr)
  end do
  do s = 1, input_nsplit
    inp%tsplit(s) = 50.0_kpr + s * 0.5_kpr
    inp%thshsplit(s) = 1.0_kpr - 0.9_kpr / max(input_nsplit, 1) * real(s, kpr)
  end do
  ! test-size overrides
  n = inp%nparticle_max
  call env_int64('PIC1DP_NPARTICLE', n)
  if (n /= inp%nparticle_max) then
    inp%nparticle_max = n
    do s = 1, input_nspecies
      inp%species_nparticle_init(s) = n
    end do
  end if
  nx = inp%nx
  call env_int64('PIC1DP_NX', nx)
  inp%nx = int(nx, c_int32_t)
  seed = inp%multirand_seed_type
  call env_int64('PIC1DP_SEED_TYPE', seed)
  inp%multirand_seed_type = int(seed, c_int32_t)
  call env_real('PIC1DP_TIME_MAX', inp%time_max)
end subroutine input_fill

end module pic1dp_host_input
