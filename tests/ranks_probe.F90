! ranks_probe.F90 -- test support (tests/test_host_logic.py): the rendez-vous of the Fortran host's ranks
! (pic1dp_amd/fortran/host_ranks.F90: files standing in for the MPI_Allgather / MPI_Bcast / MPI_Reduce the reference makes
! outside its hot path, src/pic1dp.F90:43-52, src/pic1dp_output.F90:333-356) exercised by itself, WITHOUT a GPU, at the
! target's rank count -- eight processes, which the GPU box's limit of six processes on its card keeps the host itself from
! being run as: the all-gather of the 64-byte exchange handles, the broadcast of the 128-byte RCCL id, and the reduction of
! the diagnostics (3 x 4096 + 3 x 64 doubles, src/pic1dp_output.F90:333-356) to rank 0, five rounds.
program ranks_probe
use iso_c_binding
use pic1dp_host_ranks
implicit none
integer, parameter :: nh = 64, nid = 128, nd = 3 * 4096 + 3 * 64
integer(c_signed_char) :: mine(nh), id(nid)
integer(c_signed_char), allocatable :: all(:)
real(c_double), allocatable :: a(:)
real(c_double) :: want
integer :: round, q, i
integer(c_int64_t) :: c0, c1, rate
call ranks_init()
allocate (all(nh * ranks_size), a(nd))
do round = 1, 5
  do i = 1, nh
    mine(i) = int(mod(7 * ranks_rank + 3 * i + round, 120), c_signed_char)
  end do
  call ranks_allgather_handles(mine, nh, all)
  do q = 0, ranks_size - 1
    do i = 1, nh
      if (all(q * nh + i) /= int(mod(7 * q + 3 * i + round, 120), c_signed_char)) then
        write (*, '(a, 3i6)') 'allgather mismatch', round, q, i
        stop 2
      end if
    end do
  end do
  if (ranks_rank == 0) then
    do i = 1, nid
      id(i) = int(mod(5 * i + round, 100), c_signed_char)
    end do
  else
    id = 0
  end if
  call ranks_bcast_bytes(id, nid)
  do i = 1, nid
    if (id(i) /= int(mod(5 * i + round, 100), c_signed_char)) then
      write (*, '(a, 2i6)') 'bcast mismatch', round, i
      stop 3
    end if
  end do
  do i = 1, nd
    a(i) = real(ranks_rank + 1, c_double) * 0.5_c_double ** round + real(i, c_double)
  end do
  call ranks_reduce_to_root(a, nd)
  if (ranks_rank == 0) then
    do i = 1, nd
      want = real(1, c_double) * 0.5_c_double ** round + real(i, c_double)
      do q = 1, ranks_size - 1      ! rank order, as the reduction adds
        want = want + (real(q + 1, c_double) * 0.5_c_double ** round + real(i, c_double))
      end do
      if (a(i) /= want) then
        write (*, '(a, 2i6, 2es24.16)') 'reduce mismatch', round, i, a(i), want
        stop 4
      end if
    end do
  end if
end do
! (the host makes its gathers at set-up, long before ranks_finalize removes a rank's files; here the last round is right in
! front of it: one more all-gather as a barrier -- whoever leaves it knows every rank has fetched everything before it -- and
! a moment for the slowest rank to fetch the barrier's own files)
call ranks_allgather_handles(mine, nh, all)
call system_clock(c0, rate)
do
  call system_clock(c1)
  if (c1 - c0 > rate) exit
end do
call ranks_finalize()
write (*, '(a, i0, a, i0)') 'ranks_probe ok: rank ', ranks_rank, ' of ', ranks_size
end program ranks_probe
