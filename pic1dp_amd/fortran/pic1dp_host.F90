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
! With PIC1DP_FUSED=1 in the environment the three calls of a sub-step are
! replaced by the fused pic1dp_hip_substep(ctx, global_irk); with PIC1DP_FUSED=2
! the whole irk loop is one pic1dp_hip_step(ctx, 1) (fastest: the half-step
! state is recomputed instead of stored; the library then advances
! global_itime / global_time itself, exactly as src/pic1dp.F90:92-93).
program pic1dp_host
use iso_c_binding
use pic1dp_hip
use pic1dp_host_input
use pic1dp_host_output
implicit none

type(pic1dp_input_t) :: inp
type(pic1dp_layout_t) :: lay
type(c_ptr) :: ctx
integer(c_int32_t) :: global_irk, global_itime, itermination, due, flag_optimized
real(c_double) :: global_time, ms_push, ms_charge, ms_field
character(len=8) :: buf
integer :: stat
logical :: fused, whole_step

call input_fill(inp)
lay = pic1dp_layout_t(0, 1, 0, -1)          ! one process, one GPU
call pic1dp_hip_check(pic1dp_hip_create(inp, lay, ctx), 'create')      ! particle_init + field_init
call output_init(inp)
call get_environment_variable('PIC1DP_FUSED', buf, status=stat)
fused = (stat == 0 .and. buf(1:1) == '1')
whole_step = (stat == 0 .and. buf(1:1) == '2')

call pic1dp_hip_check(pic1dp_hip_particle_load(ctx), 'particle_load')
call pic1dp_hip_check(pic1dp_hip_timers_enable(ctx, 1), 'timers_enable')
! output_all is called at the reference's cadence below: steps it follows take its diagnostics along
call pic1dp_hip_check(pic1dp_hip_set_output_fusion(ctx, 1), 'set_output_fusion')

global_itime = 0
global_time = 0.0_c_double
call pic1dp_hip_check(pic1dp_hip_set_time(ctx, global_itime, global_time), 'set_time')

! solve initial field
call pic1dp_hip_check(pic1dp_hip_collect_charge(ctx), 'collect_charge')
call pic1dp_hip_check(pic1dp_hip_solve_field(ctx), 'solve_field')
if (input_verbosity == 1) write (*, '(a/a)') 'Info: progress:', 'progrss  itime     time  int E^2 dx'
call output_all(ctx, inp, input_verbosity)

call pic1dp_hip_check(pic1dp_hip_check_termination(ctx, itermination), 'check_termination')
do while (itermination == 0)                 ! main time evolution loop
  if (whole_step) then
    call pic1dp_hip_check(pic1dp_hip_step(ctx, 1), 'step')
    call pic1dp_hip_check(pic1dp_hip_get_time(ctx, global_itime, global_time), 'get_time')
  else
    do global_irk = 1, 2
      if (fused) then
        call pic1dp_hip_check(pic1dp_hip_substep(ctx, global_irk), 'substep')
      else
        call pic1dp_hip_check(pic1dp_hip_push(ctx, global_irk), 'push')
        call pic1dp_hip_check(pic1dp_hip_particle_optimize(ctx, global_irk, flag_optimized), 'particle_optimize')
        if (flag_optimized == 1) call output_progress_optimized(ctx, inp, input_verbosity)
        call pic1dp_hip_check(pic1dp_hip_collect_charge(ctx), 'collect_charge')
        call pic1dp_hip_check(pic1dp_hip_solve_field(ctx), 'solve_field')
      end if
    end do
    global_itime = global_itime + 1
    global_time = global_time + inp%dt
    call pic1dp_hip_check(pic1dp_hip_set_time(ctx, global_itime, global_time), 'set_time')
  end if
  call pic1dp_hip_check(pic1dp_hip_check_termination(ctx, itermination), 'check_termination')
  call pic1dp_hip_check(pic1dp_hip_output_due(ctx, itermination, due), 'output_due')
  if (due == 1) call output_all(ctx, inp, input_verbosity)
end do

call output_final
if (input_verbosity >= 1) then
  call pic1dp_hip_check(pic1dp_hip_timer_ms(ctx, PIC1DP_IWT_PUSH_PARTICLE, ms_push), 'timer')
  call pic1dp_hip_check(pic1dp_hip_timer_ms(ctx, PIC1DP_IWT_COLLECT_CHARGE, ms_charge), 'timer')
  call pic1dp_hip_check(pic1dp_hip_timer_ms(ctx, PIC1DP_IWT_FIELD_ELECTRIC, ms_field), 'timer')
  write (*, '(a)') 'Info: timers (GPU, ms):'
  write (*, '(a, f12.3, a, f12.3, a, f12.3)') '   push particle', ms_push, '   collect charge', ms_charge, &
    '   electric field', ms_field
end if
call pic1dp_hip_check(pic1dp_hip_destroy(ctx), 'destroy')   ! particle_final + field_final
end program pic1dp_host
