! pic1dp_host.F90 -- Fortran host of the MI355X engine.  It walks through the
! sequence of the reference driver (program pic1dp, src/pic1dp.F90:43-125):
! init, load, initial deposit + solve, output, the RK2 loop with the three hot
! call sites, output cadence, finalisation -- every hot call goes through the
! ISO_C_BINDING interface of pic1dp_hip_mod.F90 to the HIP kernels.
!
!   reference call site (src/pic1dp.F90)     here
!   :71,:88  interaction_collect_charge      pic1dp_hip_collect_charge(ctx)
!   :72,:89  field_solve_electric            pic1dp_hip_solve_field(ctx)
!   :80      interaction_push_particle       pic1dp_hip_push(ctx, global_irk)
!   :82      particle_optimize               pic1dp_hip_particle_optimize(ctx, global_irk, flag)
!
! As the reference, the host runs as one of N processes, one per GPU (src/pic1dp.F90:43-52; `make run` starts
! four): rank / size from PIC1DP_RANK / PIC1DP_NRANKS, each rank owns its PETSC_DECIDE block of the markers, the
! charge is summed over the GPUs inside the library, and rank 0 alone writes pic1dp.out and the progress lines from
! diagnostics reduced to it.  How the library sums the charge (MPI_Allreduce, src/pic1dp_interaction.F90:130-135):
!   PIC1DP_ALLREDUCE=p2p  (default for N > 1) the one-hop exchange; the 64-byte handles are all-gathered once
!   PIC1DP_ALLREDUCE=rccl an RCCL all-reduce over xGMI: rank 0 draws the communicator's 128-byte unique id
!                         (pic1dp_hip_comm_unique_id), it is broadcast (host_ranks.F90), every rank joins
!                         (pic1dp_hip_comm_init) -- also with N = 1, where the one-rank communicator runs every launch of
!                         the N-rank RCCL step on one GPU.
!
! With PIC1DP_FUSED=1 in the environment the three calls of a sub-step are
! replaced by the fused pic1dp_hip_substep(ctx, global_irk); with PIC1DP_FUSED=2
! the whole irk loop is one pic1dp_hip_step(ctx, 1) (fastest: the half-step
! state is recomputed instead of stored; the library then advances
! global_itime / global_time itself, exactly as src/pic1dp.F90:92-93); with
! PIC1DP_FUSED=3 all the steps up to the next output_all are ONE call,
! pic1dp_hip_step(ctx, pic1dp_hip_steps_to_output): one launch per time step.
! PIC1DP_HOST_PROFILE=1 prints the wall clock of the run split into the time loop's steps, output_all's
! diagnostics and the writes of pic1dp.out (the steps are then waited for, pic1dp_hip_sync, before the clock is read).
program pic1dp_host
use iso_c_binding
use pic1dp_hip
use pic1dp_host_input
use pic1dp_host_output
use pic1dp_host_ranks
implicit none

type(pic1dp_input_t) :: inp
type(pic1dp_layout_t) :: lay
type(c_ptr) :: ctx
integer(c_int32_t) :: global_irk, global_itime, itermination, due, flag_optimized, nbatch
real(c_double) :: global_time, ms_push, ms_charge, ms_field
real(c_double) :: t_run0, t_loop0, t_a, steps_s, load_s
character(len=8) :: buf
character(len=512) :: dump_path
integer :: stat, verbosity, fail_at
integer(c_int32_t) :: timer_mode
logical :: fused, whole_step, batched, use_rccl, loop_profile
integer(c_signed_char) :: handle(PIC1DP_XCHG_HANDLE_BYTES), comm_id(PIC1DP_COMM_ID_BYTES)
integer(c_signed_char), allocatable :: handles(:)

call input_fill(inp)
call ranks_init()                           ! MPI_Comm_rank / MPI_Comm_size, src/pic1dp.F90:50-52
lay = pic1dp_layout_t(ranks_rank, ranks_size, 0, -1)   ! one process per GPU (device = rank mod visible GPUs)
call pic1dp_hip_check(pic1dp_hip_create(inp, lay, ctx), 'create')      ! particle_init + field_init
call get_environment_variable('PIC1DP_ALLREDUCE', buf, status=stat)
use_rccl = (stat == 0 .and. buf(1:4) == 'rccl')
if (use_rccl) then
  ! the charge sum over ranks (MPI_Allreduce, src/pic1dp_interaction.F90:130-135) as an RCCL all-reduce: the
  ! communicator's unique id from rank 0 to every rank (MPI_Bcast), then every rank joins
  if (ranks_rank == 0) call pic1dp_hip_check(pic1dp_hip_comm_unique_id(comm_id), 'comm_unique_id')
  call ranks_bcast_bytes(comm_id, PIC1DP_COMM_ID_BYTES)
  call pic1dp_hip_check(pic1dp_hip_comm_init(ctx, comm_id), 'comm_init')
  call pic1dp_hip_check(pic1dp_hip_set_allreduce(ctx, 1), 'set_allreduce')
else if (ranks_size > 1) then
  ! ... or as the library's one-hop exchange: every rank's 64-byte handle to every rank, once
  allocate (handles(PIC1DP_XCHG_HANDLE_BYTES * ranks_size))
  call pic1dp_hip_check(pic1dp_hip_xchg_create(ctx, handle), 'xchg_create')
  call ranks_allgather_handles(handle, PIC1DP_XCHG_HANDLE_BYTES, handles)
  call pic1dp_hip_check(pic1dp_hip_xchg_connect(ctx, handles), 'xchg_connect')
  call pic1dp_hip_check(pic1dp_hip_set_allreduce(ctx, 2), 'set_allreduce')
end if
verbosity = input_verbosity
if (ranks_rank /= 0) verbosity = 0          ! PetscPrintf prints on rank 0 only (src/pic1dp_global.F90:71-90)
if (ranks_rank == 0) call output_init(inp)
if (ranks_rank == 0) pic1dp_hip_on_abort => output_final   ! a record assembled but not yet written survives a stop on error
call get_environment_variable('PIC1DP_FUSED', buf, status=stat)
fused = (stat == 0 .and. buf(1:1) == '1')
whole_step = (stat == 0 .and. (buf(1:1) == '2' .or. buf(1:1) == '3'))
batched = (stat == 0 .and. buf(1:1) == '3')
call get_environment_variable('PIC1DP_HOST_PROFILE', buf, status=stat)
output_profile = (stat == 0 .and. buf(1:1) == '1')   ! the split steps | output_all | file writes: one device sync per iteration
loop_profile = (stat == 0 .and. buf(1:1) == '2')     ! the time loop's wall clock only: nothing added to the loop
call get_environment_variable('PIC1DP_TIMERS', buf, status=stat)
timer_mode = 17                                      ! the reference's timers (wtimer): HIP events around the launches of every 17th
                                                     ! block of 64 under a timer id, scaled (1: around every launch -- 10-28 % of a
                                                     ! default-size run; 0: off)
if (stat == 0) read (buf, *, iostat=stat) timer_mode
steps_s = 0.0_c_double
! (tests) PIC1DP_HOST_FAIL_AT=n: at time step n the host makes a call the library refuses -- what a failure in the middle of
! a run looks like: pic1dp_hip_check prints the library's message, the record assembled but not yet written goes to
! pic1dp.out (pic1dp_hip_on_abort), the program stops with code 1
fail_at = -1
call get_environment_variable('PIC1DP_HOST_FAIL_AT', buf, status=stat)
if (stat == 0) read (buf, *, iostat=stat) fail_at

t_run0 = output_wall()
call pic1dp_hip_check(pic1dp_hip_particle_load(ctx), 'particle_load')
if (output_profile) call pic1dp_hip_check(pic1dp_hip_sync(ctx), 'sync')
load_s = output_wall() - t_run0
call pic1dp_hip_check(pic1dp_hip_timers_enable(ctx, timer_mode), 'timers_enable')
! output_all is called at the reference's cadence below: steps it follows take its diagnostics along
call pic1dp_hip_check(pic1dp_hip_set_output_fusion(ctx, 1), 'set_output_fusion')

global_itime = 0
global_time = 0.0_c_double
call pic1dp_hip_check(pic1dp_hip_set_time(ctx, global_itime, global_time), 'set_time')

! solve initial field
call pic1dp_hip_check(pic1dp_hip_collect_charge(ctx), 'collect_charge')
call pic1dp_hip_check(pic1dp_hip_solve_field(ctx), 'solve_field')
if (verbosity == 1) write (*, '(a/a)') 'Info: progress:', 'progrss  itime     time  int E^2 dx'
call output_all(ctx, inp, verbosity)

call pic1dp_hip_check(pic1dp_hip_check_termination(ctx, itermination), 'check_termination')
t_loop0 = output_wall()
do while (itermination == 0)                 ! main time evolution loop
  t_a = output_wall()
  if (global_itime == fail_at) call pic1dp_hip_check(pic1dp_hip_push(ctx, 3_c_int32_t), 'push (forced failure)')
  if (whole_step) then
    nbatch = 1
    if (batched) call pic1dp_hip_check(pic1dp_hip_steps_to_output(ctx, nbatch), 'steps_to_output')
    call pic1dp_hip_check(pic1dp_hip_step(ctx, nbatch), 'step')
    call pic1dp_hip_check(pic1dp_hip_get_time(ctx, global_itime, global_time), 'get_time')
  else
    do global_irk = 1, 2
      if (fused) then
        call pic1dp_hip_check(pic1dp_hip_substep(ctx, global_irk), 'substep')
      else
        call pic1dp_hip_check(pic1dp_hip_push(ctx, global_irk), 'push')
        call pic1dp_hip_check(pic1dp_hip_particle_optimize(ctx, global_irk, flag_optimized), 'particle_optimize')
        if (flag_optimized == 1) call output_progress_optimized(ctx, inp, verbosity)
        call pic1dp_hip_check(pic1dp_hip_collect_charge(ctx), 'collect_charge')
        call pic1dp_hip_check(pic1dp_hip_solve_field(ctx), 'solve_field')
      end if
    end do
    global_itime = global_itime + 1
    global_time = global_time + inp%dt
    call pic1dp_hip_check(pic1dp_hip_set_time(ctx, global_itime, global_time), 'set_time')
  end if
  if (ranks_rank == 0) call output_flush     ! the previous record's file write, while the GPU works on what was just enqueued
  if (output_profile) then
    call pic1dp_hip_check(pic1dp_hip_sync(ctx), 'sync')
    steps_s = steps_s + (output_wall() - t_a)
  end if
  call pic1dp_hip_check(pic1dp_hip_check_termination(ctx, itermination), 'check_termination')
  call pic1dp_hip_check(pic1dp_hip_output_due(ctx, itermination, due), 'output_due')
  if (due == 1) call output_all(ctx, inp, verbosity)
end do

if (ranks_rank == 0) call output_final
if (loop_profile .and. ranks_rank == 0) then
  call pic1dp_hip_check(pic1dp_hip_sync(ctx), 'sync')
  write (*, '(a, f10.3, a, i8, a)') 'Info: host wall clock (s):   time loop', output_wall() - t_loop0, '   (', global_itime, ' steps)'
end if
if (output_profile .and. ranks_rank == 0) then
  t_a = output_wall()
  write (*, '(a)') 'Info: host wall clock (s):'
  write (*, '(a, f10.3, a, f10.3, a, i8, a)') '   particle load', load_s, '   time loop', t_a - t_loop0, &
    '   (', global_itime, ' steps)'
  write (*, '(a, f10.3, a, f10.3, a, f10.3, a, i6, a)') '   steps', steps_s, '   output_all: library', output_lib_s, &
    '   file writes', output_write_s, '   (', output_records, ' records)'
end if
if (verbosity >= 1) then
  call pic1dp_hip_check(pic1dp_hip_timer_ms(ctx, PIC1DP_IWT_PUSH_PARTICLE, ms_push), 'timer')
  call pic1dp_hip_check(pic1dp_hip_timer_ms(ctx, PIC1DP_IWT_COLLECT_CHARGE, ms_charge), 'timer')
  call pic1dp_hip_check(pic1dp_hip_timer_ms(ctx, PIC1DP_IWT_FIELD_ELECTRIC, ms_field), 'timer')
  if (timer_mode > 1) then   ! (the reference's wtimer totals are exact: say so when these are not)
    write (*, '(a, i0, a)') 'Info: timers (GPU, ms; estimates: launches of every ', timer_mode, &
      'th block of 64 timed and scaled, PIC1DP_TIMERS=1 times every launch):'
  else
    write (*, '(a)') 'Info: timers (GPU, ms):'
  end if
  write (*, '(a, f12.3, a, f12.3, a, f12.3)') '   push particle', ms_push, '   collect charge', ms_charge, &
    '   electric field', ms_field
end if
call get_environment_variable('PIC1DP_DUMP_MARKERS', dump_path, status=stat)
if (stat == 0) call dump_markers(trim(dump_path))
call pic1dp_hip_check(pic1dp_hip_destroy(ctx), 'destroy')   ! particle_final + field_final
call ranks_finalize()

contains

! PIC1DP_DUMP_MARKERS=<file>: this rank's markers of species 0 at the end of the run, raw native doubles
! [np as one double | x | v | p | w] -- for tests that compare two runs of the host bit for bit
subroutine dump_markers(path)
  character(len=*), intent(in) :: path
  integer(c_int64_t) :: nalloc, np
  real(c_double), allocatable :: x(:), v(:), p(:), w(:)
  integer :: u
  call pic1dp_hip_check(pic1dp_hip_local_sizes(ctx, 0_c_int32_t, nalloc, np), 'local_sizes')
  allocate (x(nalloc), v(nalloc), p(nalloc), w(nalloc))
  call pic1dp_hip_check(pic1dp_hip_particles_download(ctx, 0_c_int32_t, x, v, p, w, nalloc), 'particles_download')
  open (newunit=u, file=path, access='stream', form='unformatted', status='replace')
  write (u) real(np, c_double)
  write (u) x(1 : np), v(1 : np), p(1 : np), w(1 : np)
  close (u)
end subroutine dump_markers

end program pic1dp_host
