! host_ranks.F90 -- the Fortran host as one of N processes WITHOUT MPI in the image (the reference is an MPI
! program: src/pic1dp.F90:43-52, `mpiexec -n 4`, run/Makefile:41; this image's mpif90 wrappers are unusable).
! Rank and size come from the environment (PIC1DP_RANK, PIC1DP_NRANKS), and the two things the reference does
! with MPI outside the hot path travel through small files in a rendez-vous directory (PIC1DP_RENDEZVOUS -- fresh for
! every run: a file an earlier run left there would be taken for this run's; ranks_finalize removes this run's):
!   ranks_allgather_handles  <->  MPI_Allgather of the 64-byte exchange handles (INTEGRATION.md section 4)
!   ranks_bcast_bytes        <->  MPI_Bcast of RCCL's 128-byte unique id from rank 0 (PIC1DP_ALLREDUCE=rccl)
!   ranks_reduce_to_root     <->  MPI_Reduce(..., MPI_SUM, 0, ...) of the diagnostics (src/pic1dp_output.F90:333-356)
! The charge sum of the hot path (MPI_Allreduce, src/pic1dp_interaction.F90:130-135) does NOT go through files:
! it is the library's one-hop exchange between the GPUs.  With MPI in the image these two routines are the two
! MPI calls named above and nothing else changes.
module pic1dp_host_ranks
use iso_c_binding
use pic1dp_hip, only: pic1dp_hip_abort
implicit none

integer :: ranks_rank = 0, ranks_size = 1
character(len=512) :: ranks_dir = '.'
integer :: ranks_seq = 0
real :: ranks_timeout_s = 120.0

interface
  function c_rename(oldname, newname) bind(C, name="rename") result(rc)
    import
    character(kind=c_char), intent(in) :: oldname(*), newname(*)
    integer(c_int) :: rc
  end function c_rename
  function c_usleep(usec) bind(C, name="usleep") result(rc)
    import
    integer(c_int), value :: usec
    integer(c_int) :: rc
  end function c_usleep
  function c_unlink(path) bind(C, name="unlink") result(rc)
    import
    character(kind=c_char), intent(in) :: path(*)
    integer(c_int) :: rc
  end function c_unlink
end interface

contains

subroutine ranks_init()
  character(len=32) :: buf
  integer :: stat
  call get_environment_variable('PIC1DP_RANK', buf, status=stat)
  if (stat == 0) read (buf, *) ranks_rank
  call get_environment_variable('PIC1DP_NRANKS', buf, status=stat)
  if (stat == 0) read (buf, *) ranks_size
  call get_environment_variable('PIC1DP_RENDEZVOUS', ranks_dir, status=stat)
  if (stat /= 0) ranks_dir = '.'
  if (ranks_size < 1 .or. ranks_rank < 0 .or. ranks_rank >= ranks_size) then
    write (*, '(a)') 'Error: PIC1DP_RANK / PIC1DP_NRANKS out of range'
    stop 1
  end if
end subroutine ranks_init

function ranks_file(tag, seq, rank) result(path)
  character(len=*), intent(in) :: tag
  integer, intent(in) :: seq, rank
  character(len=640) :: path
  write (path, '(a, a, a, a, i0, a, i0, a)') trim(ranks_dir), '/', tag, '_', seq, '_', rank, '.bin'
end function ranks_file

! a file appears whole or not at all: written under a temporary name, then renamed
subroutine ranks_publish(path, bytes, nbytes)
  character(len=*), intent(in) :: path
  integer(c_signed_char), intent(in) :: bytes(*)
  integer, intent(in) :: nbytes
  integer :: u, rc
  open (newunit=u, file=trim(path)//'.tmp', access='stream', form='unformatted', status='replace')
  write (u) bytes(1 : nbytes)
  close (u)
  rc = c_rename(trim(path)//'.tmp'//c_null_char, trim(path)//c_null_char)
  if (rc /= 0) then
    write (*, '(2a)') 'Error: cannot publish ', trim(path)
    stop 1
  end if
end subroutine ranks_publish

subroutine ranks_fetch(path, bytes, nbytes)
  character(len=*), intent(in) :: path
  integer(c_signed_char), intent(out) :: bytes(*)
  integer, intent(in) :: nbytes
  integer :: u, rc, stat
  logical :: there
  real :: waited
  waited = 0.0
  do
    inquire (file=trim(path), exist=there)
    if (there) exit
    rc = c_usleep(2000_c_int)
    waited = waited + 0.002
    if (waited > ranks_timeout_s) then
      write (*, '(3a)') 'Error: rank file ', trim(path), ' did not appear (a peer stopped?)'
      call pic1dp_hip_abort()   ! (stops; a buffered record of pic1dp.out is written first)
    end if
  end do
  open (newunit=u, file=trim(path), access='stream', form='unformatted', status='old', iostat=stat)
  read (u) bytes(1 : nbytes)
  close (u)
end subroutine ranks_fetch

! MPI_Allgather of nbytes per rank: all(1 : nbytes * size) in rank order
subroutine ranks_allgather_handles(mine, nbytes, all)
  integer(c_signed_char), intent(in) :: mine(*)
  integer, intent(in) :: nbytes
  integer(c_signed_char), intent(out) :: all(*)
  integer :: q
  ranks_seq = ranks_seq + 1
  call ranks_publish(ranks_file('gather', ranks_seq, ranks_rank), mine, nbytes)
  do q = 0, ranks_size - 1
    call ranks_fetch(ranks_file('gather', ranks_seq, q), all(q * nbytes + 1 : (q + 1) * nbytes), nbytes)
  end do
end subroutine ranks_allgather_handles

! MPI_Bcast(buf, nbytes, MPI_BYTE, 0): rank 0 publishes, the others fetch (the file goes with ranks_finalize)
subroutine ranks_bcast_bytes(buf, nbytes)
  integer(c_signed_char), intent(inout) :: buf(*)
  integer, intent(in) :: nbytes
  ranks_seq = ranks_seq + 1
  if (ranks_size == 1) return
  if (ranks_rank == 0) then
    call ranks_publish(ranks_file('gather', ranks_seq, 0), buf, nbytes)
  else
    call ranks_fetch(ranks_file('gather', ranks_seq, 0), buf, nbytes)
  end if
end subroutine ranks_bcast_bytes

! MPI_Reduce(a, a, n, MPI_DOUBLE, MPI_SUM, 0): on rank 0 a becomes the sum over ranks in rank order; the
! other ranks only contribute (and go on at once, like MPI_Reduce lets them)
subroutine ranks_reduce_to_root(a, n)
  real(c_double), intent(inout), target :: a(*)
  integer, intent(in) :: n
  real(c_double), allocatable, target :: other(:)
  integer(c_signed_char), pointer :: raw(:)
  integer :: q, rc
  if (ranks_size == 1) return
  ranks_seq = ranks_seq + 1
  if (ranks_rank /= 0) then
    call c_f_pointer(c_loc(a), raw, [8 * n])
    call ranks_publish(ranks_file('reduce', ranks_seq, ranks_rank), raw, 8 * n)
    return
  end if
  allocate (other(n))
  call c_f_pointer(c_loc(other), raw, [8 * n])
  do q = 1, ranks_size - 1
    call ranks_fetch(ranks_file('reduce', ranks_seq, q), raw, 8 * n)
    a(1 : n) = a(1 : n) + other(1 : n)
    rc = c_unlink(trim(ranks_file('reduce', ranks_seq, q))//c_null_char)
  end do
end subroutine ranks_reduce_to_root

! the end of a run that went well: this rank's gather files go (the reduce files went as they were read)
subroutine ranks_finalize()
  integer :: q, rc
  if (ranks_size == 1) return
  do q = 1, ranks_seq
    rc = c_unlink(trim(ranks_file('gather', q, ranks_rank))//c_null_char)
  end do
end subroutine ranks_finalize

end module pic1dp_host_ranks
