! host_output.F90 -- pic1dp.out in the reference's byte layout (PETSc binary
! viewer: big-endian, int32 / float64; src/pic1dp_output.F90:74-92,173-186,
! 457-474) and the stdout progress lines (:510-526), fed by the GPU engine.
module pic1dp_host_output
use iso_c_binding
use pic1dp_hip
use pic1dp_host_ranks
implicit none

integer, parameter :: output_unit_out = 71
integer(c_int32_t), parameter :: vec_file_classid = 1211214
! PIC1DP_HOST_PROFILE=1: wall clock of output_all split into the library's part (diagnostics on the GPU, their
! way to the host) and the file writes; the counterpart of the reference's global_iwt_output timer
logical :: output_profile = .false.
real(c_double) :: output_lib_s = 0.0_c_double, output_write_s = 0.0_c_double
integer :: output_records = 0

contains

function output_wall() result(t)
  real(c_double) :: t
  integer(c_int64_t) :: cnt, rate
  call system_clock(cnt, rate)
  t = real(cnt, c_double) / real(rate, c_double)
end function output_wall

subroutine output_init(inp)
  type(pic1dp_input_t), intent(in) :: inp
  open (output_unit_out, file='pic1dp.out', access='stream', form='unformatted', &
    status='replace', convert='big_endian')
  write (output_unit_out) inp%nspecies, inp%nmode, inp%nx, inp%nv, inp%nx_opd, inp%nv_opd, &
    inp%modes(1 : inp%nmode)
  write (output_unit_out) inp%lx, inp%v_max
end subroutine output_init

subroutine output_vec(a)
  real(c_double), intent(in) :: a(:)
  write (output_unit_out) vec_file_classid, int(size(a), c_int32_t)
  write (output_unit_out) a
end subroutine output_vec

! output_field + output_ptcldist + output_progress(1)
subroutine output_all(ctx, inp, verbosity)
  type(c_ptr), intent(in) :: ctx
  type(pic1dp_input_t), intent(in) :: inp
  integer(c_int32_t), intent(in) :: verbosity
  real(c_double) :: scal(2 + 3 * inp%nspecies), sums(3 * inp%nspecies)
  real(c_double) :: e(inp%nx), cd(inp%nx), re(inp%nmode), im(inp%nmode)
  real(c_double), allocatable :: mxv(:), txv(:), pxv(:), mv(:), tv(:), pv(:), dist(:)
  integer(c_int32_t) :: s, itime
  real(c_double) :: time, progress(2)
  character :: cprogress
  integer :: nxv, ntot
  real(c_double) :: t0, t1

  t0 = output_wall()
  nxv = inp%nx_opd * inp%nv_opd
  if (ranks_size == 1) then
    ! one rank: everything the record holds in ONE call and one wait (pic1dp_hip_output_all)
    ntot = 3 * nxv + 3 * inp%nv_opd
    allocate (dist(ntot * inp%nspecies))
    call pic1dp_hip_check(pic1dp_hip_output_all(ctx, scal, int(size(scal), c_int32_t), e, cd, re, im, dist), 'output_all')
    t1 = output_wall()
    output_lib_s = output_lib_s + (t1 - t0)
    write (output_unit_out) scal
    call output_vec(re)
    call output_vec(im)
    call output_vec(e)
    call output_vec(cd)
    write (output_unit_out) dist        ! per species: markr_xv, total_xv, pertb_xv, markr_v, total_v, pertb_v -- the file's order
    t0 = output_wall()
    output_write_s = output_write_s + (t0 - t1)
    output_records = output_records + 1
  else
  allocate (mxv(nxv), txv(nxv), pxv(nxv), mv(inp%nv_opd), tv(inp%nv_opd), pv(inp%nv_opd))
  ! VecSum over ranks (src/pic1dp_output.F90:126-151): local sums, reduced to rank 0, finished there
  do s = 0, inp%nspecies - 1
    call pic1dp_hip_check(pic1dp_hip_energy_sums(ctx, s, sums(3 * s + 1 : 3 * s + 3)), 'energy_sums')
  end do
  call ranks_reduce_to_root(sums, 3 * inp%nspecies)
  if (ranks_rank == 0) call pic1dp_hip_check(pic1dp_hip_output_scalars_from(ctx, sums, scal, &
    int(size(scal), c_int32_t)), 'output_scalars_from')
  if (ranks_rank == 0) then
    call pic1dp_hip_check(pic1dp_hip_get_field(ctx, e, cd, re, im), 'get_field')
    t1 = output_wall()
    output_lib_s = output_lib_s + (t1 - t0)
    write (output_unit_out) scal
    call output_vec(re)
    call output_vec(im)
    call output_vec(e)
    call output_vec(cd)
    t0 = output_wall()
    output_write_s = output_write_s + (t0 - t1)
  end if
  do s = 0, inp%nspecies - 1
    ! MPI_Reduce of the six histograms to rank 0 (:333-356), which scales and writes them
    call pic1dp_hip_check(pic1dp_hip_ptcldist(ctx, s, 0_c_int32_t, mxv, txv, pxv, mv, tv, pv), 'ptcldist')
    call ranks_reduce_to_root(mxv, nxv)
    call ranks_reduce_to_root(txv, nxv)
    call ranks_reduce_to_root(pxv, nxv)
    call ranks_reduce_to_root(mv, int(inp%nv_opd))
    call ranks_reduce_to_root(tv, int(inp%nv_opd))
    call ranks_reduce_to_root(pv, int(inp%nv_opd))
    if (ranks_rank == 0) call pic1dp_hip_check(pic1dp_hip_ptcldist_finish(ctx, s, mxv, txv, pxv, mv, tv, pv), &
      'ptcldist_finish')
    if (ranks_rank /= 0) cycle
    t1 = output_wall()
    output_lib_s = output_lib_s + (t1 - t0)
    write (output_unit_out) mxv
    write (output_unit_out) txv
    write (output_unit_out) pxv
    write (output_unit_out) mv
    write (output_unit_out) tv
    write (output_unit_out) pv
    t0 = output_wall()
    output_write_s = output_write_s + (t0 - t1)
  end do
  output_records = output_records + 1
  end if
  if (verbosity == 1) then
    call pic1dp_hip_check(pic1dp_hip_get_time(ctx, itime, time), 'get_time')
    progress(1) = 1e2_c_double * real(itime, c_double) / inp%ntime_max
    progress(2) = 1e2_c_double * time / inp%time_max
    if (maxloc(progress, 1) == 1) then
      cprogress = 'i'
    else
      cprogress = 't'
    end if
    write (*, '(a, f5.1, a, i7, f9.3, es12.3e3)') cprogress, maxval(progress), '%', itime, time, scal(2)
  else if (verbosity >= 2) then
    call pic1dp_hip_check(pic1dp_hip_get_time(ctx, itime, time), 'get_time')
    write (*, '(a, i7, a, f9.3)') 'Info: finished itime = ', itime, ', time = ', time
  end if
end subroutine output_all

! output_progress(2) (src/pic1dp_output.F90:487-543): the line after a merge /
! remove / split event; it is printed inside the time step, hence itime + 1 and
! time + dt
subroutine output_progress_optimized(ctx, inp, verbosity)
  type(c_ptr), intent(in) :: ctx
  type(pic1dp_input_t), intent(in) :: inp
  integer(c_int32_t), intent(in) :: verbosity
  integer(c_int64_t) :: nalloc, np, nparticle_allspec
  integer(c_int32_t) :: s, itime
  real(c_double) :: time, progress(2)
  character :: cprogress

  if (verbosity == 0) return
  nparticle_allspec = 0
  do s = 0, inp%nspecies - 1
    call pic1dp_hip_check(pic1dp_hip_local_sizes(ctx, s, nalloc, np), 'local_sizes')
    nparticle_allspec = nparticle_allspec + np
  end do
  call pic1dp_hip_check(pic1dp_hip_get_time(ctx, itime, time), 'get_time')
  if (verbosity == 1) then
    progress(1) = 1e2_c_double * real(itime, c_double) / inp%ntime_max
    progress(2) = 1e2_c_double * time / inp%time_max
    if (maxloc(progress, 1) == 1) then
      cprogress = 'i'
    else
      cprogress = 't'
    end if
    write (*, '(a, f5.1, a, i7, f9.3, a, i9)') cprogress, maxval(progress), '%', itime + 1, time + inp%dt, &
      ' : optimization performed, current # of particles ', nparticle_allspec
  else
    write (*, '(2a, i9)') 'Info: particle_optimize performed, ', 'current # of particles:', nparticle_allspec
  end if
end subroutine output_progress_optimized

subroutine output_final
  real(c_double) :: t0
  t0 = output_wall()
  close (output_unit_out)
  output_write_s = output_write_s + (output_wall() - t0)
end subroutine output_final

end module pic1dp_host_output
