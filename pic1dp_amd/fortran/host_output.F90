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
! A record is assembled in memory as big-endian 8-byte words (every item of it is one: a float64, or a Vec's two int32 of
! header) and written with ONE native write: through the Fortran runtime's convert='big_endian' the same bytes cost 0.12 ms
! per 103 KB record (element-wise conversion), a fifth of the default run's output cost
integer(c_int64_t), allocatable :: rec_buf(:)
integer :: rec_n = 0

contains

pure elemental function bswap64(x) result(y)
  integer(c_int64_t), intent(in) :: x
  integer(c_int64_t) :: y
  y = ior(ior(ior(ishft(iand(x, int(z'FF', c_int64_t)), 56), ishft(iand(x, int(z'FF00', c_int64_t)), 40)), &
              ior(ishft(iand(x, int(z'FF0000', c_int64_t)), 24), ishft(iand(x, int(z'FF000000', c_int64_t)), 8))), &
          ior(ior(iand(ishft(x, -8), int(z'FF000000', c_int64_t)), iand(ishft(x, -24), int(z'FF0000', c_int64_t))), &
              ior(iand(ishft(x, -40), int(z'FF00', c_int64_t)), iand(ishft(x, -56), int(z'FF', c_int64_t)))))
end function bswap64

pure elemental function bswap32(x) result(y)
  integer(c_int32_t), intent(in) :: x
  integer(c_int32_t) :: y
  y = ior(ior(ishft(iand(x, int(z'FF', c_int32_t)), 24), ishft(iand(x, int(z'FF00', c_int32_t)), 8)), &
          ior(iand(ishft(x, -8), int(z'FF00', c_int32_t)), iand(ishft(x, -24), int(z'FF', c_int32_t))))
end function bswap32

subroutine rec_room(n)
  integer, intent(in) :: n
  integer(c_int64_t), allocatable :: grown(:)
  if (.not. allocated(rec_buf)) allocate (rec_buf(max(4096, 2 * n)))
  if (rec_n + n > size(rec_buf)) then
    allocate (grown(2 * (rec_n + n)))
    grown(1 : rec_n) = rec_buf(1 : rec_n)
    call move_alloc(grown, rec_buf)
  end if
end subroutine rec_room

! the reals of a record, as they stand in the file
subroutine rec_put(a)
  real(c_double), intent(in) :: a(:)
  call rec_room(size(a))
  rec_buf(rec_n + 1 : rec_n + size(a)) = bswap64(transfer(a, 1_c_int64_t, size(a)))
  rec_n = rec_n + size(a)
end subroutine rec_put

! a PETSc Vec: class id, length (int32 each), values
subroutine rec_put_vec(a)
  real(c_double), intent(in) :: a(:)
  integer(c_int32_t) :: hdr(2)
  call rec_room(1)
  hdr(1) = bswap32(vec_file_classid)
  hdr(2) = bswap32(int(size(a), c_int32_t))
  rec_n = rec_n + 1
  rec_buf(rec_n) = transfer(hdr, 1_c_int64_t)
  call rec_put(a)
end subroutine rec_put_vec

subroutine rec_flush
  if (rec_n > 0) write (output_unit_out) rec_buf(1 : rec_n)
  rec_n = 0
end subroutine rec_flush

! The record output_all has assembled goes to the file HERE: the driver calls this after it has handed the next step(s) to
! the GPU, so that the write passes while the device works (and once more before the file is closed)
subroutine output_flush
  real(c_double) :: t0
  if (rec_n == 0) return
  t0 = output_wall()
  call rec_flush
  output_write_s = output_write_s + (output_wall() - t0)
end subroutine output_flush

function output_wall() result(t)
  real(c_double) :: t
  integer(c_int64_t) :: cnt, rate
  call system_clock(cnt, rate)
  t = real(cnt, c_double) / real(rate, c_double)
end function output_wall

subroutine output_init(inp)
  type(pic1dp_input_t), intent(in) :: inp
  open (output_unit_out, file='pic1dp.out', access='stream', form='unformatted', status='replace')
  write (output_unit_out) bswap32(inp%nspecies), bswap32(inp%nmode), bswap32(inp%nx), bswap32(inp%nv), &
    bswap32(inp%nx_opd), bswap32(inp%nv_opd), bswap32(inp%modes(1 : inp%nmode))
  call rec_put([inp%lx, inp%v_max])
  call rec_flush
end subroutine output_init

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
    call rec_put(scal)
    call rec_put_vec(re)
    call rec_put_vec(im)
    call rec_put_vec(e)
    call rec_put_vec(cd)
    call rec_put(dist)                  ! per species: markr_xv, total_xv, pertb_xv, markr_v, total_v, pertb_v -- the file's order
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
    call rec_put(scal)
    call rec_put_vec(re)
    call rec_put_vec(im)
    call rec_put_vec(e)
    call rec_put_vec(cd)
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
    call rec_put(mxv)
    call rec_put(txv)
    call rec_put(pxv)
    call rec_put(mv)
    call rec_put(tv)
    call rec_put(pv)
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
  call output_flush
  t0 = output_wall()
  close (output_unit_out)
  output_write_s = output_write_s + (output_wall() - t0)
end subroutine output_final

end module pic1dp_host_output
