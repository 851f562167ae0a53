! pic1dp_hip_mod.F90 -- ISO_C_BINDING face of the C ABI in include/pic1dp_hip.h.
!
! This is the whole "Fortran side" of the drop-in: a host that today calls the
! reference's interaction_collect_charge / field_solve_electric /
! interaction_push_particle (src/pic1dp.F90:71,72,80,88,89) uses this module and
! calls pic1dp_hip_collect_charge / _solve_field / _push on a context instead.
! Every function returns the error code (0 = ok), which plays the role of
! global_ierr + CHKERRQ (src/pic1dp_global.F90:59).
module pic1dp_hip
use iso_c_binding
implicit none

integer(c_int), parameter :: PIC1DP_ABI_VERSION = 4
integer(c_int), parameter :: PIC1DP_MAX_SPECIES = 8
integer(c_int), parameter :: PIC1DP_MAX_MODES = 4096
integer(c_int), parameter :: PIC1DP_MAX_INIT_MODES = 16
integer(c_int), parameter :: PIC1DP_COMM_ID_BYTES = 128
integer(c_int), parameter :: PIC1DP_XCHG_HANDLE_BYTES = 64
integer(c_int), parameter :: PIC1DP_MAX_OPT = 32

! wall-clock timer ids = the reference's (src/pic1dp_global.F90:38-50)
integer(c_int32_t), parameter :: PIC1DP_IWT_PUSH_PARTICLE = 4, PIC1DP_IWT_COLLECT_CHARGE = 6, &
  PIC1DP_IWT_FIELD_ELECTRIC = 7, PIC1DP_IWT_PARTICLE_OPTIMIZE = 8, PIC1DP_IWT_MPIALLREDU = 21

! struct pic1dp_input: run-time mirror of the parameters of src/pic1dp_input.F90
type, bind(C) :: pic1dp_input_t
  integer(c_int32_t) :: abi_version, ntime_max, linear, iptcldist, nspecies, nmode, init_nmode
  integer(c_int32_t) :: deltaf, imarker, nx, nv, iptclshape, nx_opd, nv_opd
  integer(c_int32_t) :: multirand_al_int, multirand_seed_type, multirand_warmup, multirand_selftest
  integer(c_int64_t) :: nparticle_max
  integer(c_int64_t) :: species_nparticle_init(PIC1DP_MAX_SPECIES)
  real(c_double) :: time_max, lx, dt, v_max, output_interval
  real(c_double) :: species_charge(PIC1DP_MAX_SPECIES), species_mass(PIC1DP_MAX_SPECIES)
  real(c_double) :: species_temperature(PIC1DP_MAX_SPECIES), species_temperature2(PIC1DP_MAX_SPECIES)
  real(c_double) :: species_density(PIC1DP_MAX_SPECIES), species_v0(PIC1DP_MAX_SPECIES)
  integer(c_int32_t) :: modes(PIC1DP_MAX_MODES)
  integer(c_int32_t) :: init_mode(PIC1DP_MAX_INIT_MODES)
  real(c_double) :: init_mode_cos(PIC1DP_MAX_INIT_MODES), init_mode_sin(PIC1DP_MAX_INIT_MODES)
  ! marker optimisation (src/pic1dp_input.F90:141-206)
  integer(c_int32_t) :: nmerge, nremove, nsplit, typeremove, split_ngroup, reserved0
  real(c_double) :: remove_frac, split_dv_sig_frac
  real(c_double) :: tmerge(PIC1DP_MAX_OPT), thshmerge(PIC1DP_MAX_OPT)
  real(c_double) :: tremove(PIC1DP_MAX_OPT), thshremove(PIC1DP_MAX_OPT)
  real(c_double) :: tsplit(PIC1DP_MAX_OPT), thshsplit(PIC1DP_MAX_OPT)
end type pic1dp_input_t

! struct pic1dp_layout: this process in the particle decomposition
type, bind(C) :: pic1dp_layout_t
  integer(c_int32_t) :: rank, nranks, npe, device
end type pic1dp_layout_t

interface
  function pic1dp_hip_last_error_c() bind(C, name="pic1dp_hip_last_error") result(msg)
    import
    type(c_ptr) :: msg
  end function pic1dp_hip_last_error_c
  function pic1dp_hip_abi_version() bind(C, name="pic1dp_hip_abi_version") result(ierr)
    import
    integer(c_int) :: ierr
  end function pic1dp_hip_abi_version
  function pic1dp_hip_tuning_build() bind(C, name="pic1dp_hip_tuning_build") result(ierr)
    import
    integer(c_int) :: ierr
  end function pic1dp_hip_tuning_build
  function pic1dp_hip_device_count() bind(C, name="pic1dp_hip_device_count") result(ierr)
    import
    integer(c_int) :: ierr
  end function pic1dp_hip_device_count
  function pic1dp_hip_input_size() bind(C, name="pic1dp_hip_input_size") result(ierr)
    import
    integer(c_int) :: ierr
  end function pic1dp_hip_input_size
  function pic1dp_hip_input_defaults(inp) bind(C, name="pic1dp_hip_input_defaults") result(ierr)
    import
    type(pic1dp_input_t), intent(out) :: inp
    integer(c_int) :: ierr
  end function pic1dp_hip_input_defaults
  function pic1dp_hip_input_validate(inp, layout) bind(C, name="pic1dp_hip_input_validate") result(ierr)
    import
    type(pic1dp_input_t), intent(in) :: inp
    type(pic1dp_layout_t), intent(in) :: layout
    integer(c_int) :: ierr
  end function pic1dp_hip_input_validate
  function pic1dp_hip_block_sizes(inp, ispecies, mype, npe, nalloc, np) bind(C, name="pic1dp_hip_block_sizes") result(ierr)
    import
    type(pic1dp_input_t), intent(in) :: inp
    integer(c_int32_t), value :: ispecies
    integer(c_int32_t), value :: mype
    integer(c_int32_t), value :: npe
    integer(c_int64_t), intent(out) :: nalloc
    integer(c_int64_t), intent(out) :: np
    integer(c_int) :: ierr
  end function pic1dp_hip_block_sizes
  function pic1dp_hip_host_particle_load(inp, mype, npe, x, v, p, w, nalloc) bind(C, name="pic1dp_hip_host_particle_load") result(ierr)
    import
    type(pic1dp_input_t), intent(in) :: inp
    integer(c_int32_t), value :: mype
    integer(c_int32_t), value :: npe
    real(c_double), intent(inout) :: x(*)
    real(c_double), intent(inout) :: v(*)
    real(c_double), intent(inout) :: p(*)
    real(c_double), intent(inout) :: w(*)
    integer(c_int64_t), value :: nalloc
    integer(c_int) :: ierr
  end function pic1dp_hip_host_particle_load
  function pic1dp_hip_host_multirand_int64(al_int, seed_type, mype, warmup, selftest, out, n) bind(C, name="pic1dp_hip_host_multirand_int64") result(ierr)
    import
    integer(c_int32_t), value :: al_int
    integer(c_int32_t), value :: seed_type
    integer(c_int32_t), value :: mype
    integer(c_int32_t), value :: warmup
    integer(c_int32_t), value :: selftest
    integer(c_int64_t), intent(inout) :: out(*)
    integer(c_int64_t), value :: n
    integer(c_int) :: ierr
  end function pic1dp_hip_host_multirand_int64
  function pic1dp_hip_create(inp, layout, ctx) bind(C, name="pic1dp_hip_create") result(ierr)
    import
    type(pic1dp_input_t), intent(in) :: inp
    type(pic1dp_layout_t), intent(in) :: layout
    type(c_ptr), intent(out) :: ctx
    integer(c_int) :: ierr
  end function pic1dp_hip_create
  function pic1dp_hip_destroy(ctx) bind(C, name="pic1dp_hip_destroy") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int) :: ierr
  end function pic1dp_hip_destroy
  function pic1dp_hip_local_sizes(ctx, ispecies, nalloc, np) bind(C, name="pic1dp_hip_local_sizes") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: ispecies
    integer(c_int64_t), intent(out) :: nalloc
    integer(c_int64_t), intent(out) :: np
    integer(c_int) :: ierr
  end function pic1dp_hip_local_sizes
  function pic1dp_hip_particle_load(ctx) bind(C, name="pic1dp_hip_particle_load") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int) :: ierr
  end function pic1dp_hip_particle_load
  function pic1dp_hip_particles_upload(ctx, ispecies, x, v, p, w, n, np) bind(C, name="pic1dp_hip_particles_upload") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: ispecies
    real(c_double), intent(in) :: x(*)
    real(c_double), intent(in) :: v(*)
    real(c_double), intent(in) :: p(*)
    real(c_double), intent(in) :: w(*)
    integer(c_int64_t), value :: n
    integer(c_int64_t), value :: np
    integer(c_int) :: ierr
  end function pic1dp_hip_particles_upload
  function pic1dp_hip_particles_download(ctx, ispecies, x, v, p, w, n) bind(C, name="pic1dp_hip_particles_download") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: ispecies
    real(c_double), intent(inout) :: x(*)
    real(c_double), intent(inout) :: v(*)
    real(c_double), intent(inout) :: p(*)
    real(c_double), intent(inout) :: w(*)
    integer(c_int64_t), value :: n
    integer(c_int) :: ierr
  end function pic1dp_hip_particles_download
  function pic1dp_hip_particles_download_bak(ctx, ispecies, xb, vb, wb, n) bind(C, name="pic1dp_hip_particles_download_bak") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: ispecies
    real(c_double), intent(inout) :: xb(*)
    real(c_double), intent(inout) :: vb(*)
    real(c_double), intent(inout) :: wb(*)
    integer(c_int64_t), value :: n
    integer(c_int) :: ierr
  end function pic1dp_hip_particles_download_bak
  function pic1dp_hip_collect_charge(ctx) bind(C, name="pic1dp_hip_collect_charge") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int) :: ierr
  end function pic1dp_hip_collect_charge
  function pic1dp_hip_solve_field(ctx) bind(C, name="pic1dp_hip_solve_field") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int) :: ierr
  end function pic1dp_hip_solve_field
  function pic1dp_hip_push(ctx, irk) bind(C, name="pic1dp_hip_push") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: irk
    integer(c_int) :: ierr
  end function pic1dp_hip_push
  function pic1dp_hip_particle_optimize(ctx, irk, flag_optimized) bind(C, name="pic1dp_hip_particle_optimize") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: irk
    integer(c_int32_t), intent(out) :: flag_optimized
    integer(c_int) :: ierr
  end function pic1dp_hip_particle_optimize
  function pic1dp_hip_substep(ctx, irk) bind(C, name="pic1dp_hip_substep") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: irk
    integer(c_int) :: ierr
  end function pic1dp_hip_substep
  function pic1dp_hip_step(ctx, nsteps) bind(C, name="pic1dp_hip_step") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: nsteps
    integer(c_int) :: ierr
  end function pic1dp_hip_step
  function pic1dp_hip_set_step_mode(ctx, mode) bind(C, name="pic1dp_hip_set_step_mode") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: mode
    integer(c_int) :: ierr
  end function pic1dp_hip_set_step_mode
  function pic1dp_hip_set_field_solver(ctx, kind) bind(C, name="pic1dp_hip_set_field_solver") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: kind
    integer(c_int) :: ierr
  end function pic1dp_hip_set_field_solver
  function pic1dp_hip_get_field_half(ctx, electric_half) bind(C, name="pic1dp_hip_get_field_half") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(inout) :: electric_half(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_get_field_half
  function pic1dp_hip_sync(ctx) bind(C, name="pic1dp_hip_sync") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int) :: ierr
  end function pic1dp_hip_sync
  function pic1dp_hip_get_time(ctx, itime, time) bind(C, name="pic1dp_hip_get_time") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), intent(out) :: itime
    real(c_double), intent(out) :: time
    integer(c_int) :: ierr
  end function pic1dp_hip_get_time
  function pic1dp_hip_set_time(ctx, itime, time) bind(C, name="pic1dp_hip_set_time") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: itime
    real(c_double), value :: time
    integer(c_int) :: ierr
  end function pic1dp_hip_set_time
  function pic1dp_hip_check_termination(ctx, flag) bind(C, name="pic1dp_hip_check_termination") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), intent(out) :: flag
    integer(c_int) :: ierr
  end function pic1dp_hip_check_termination
  function pic1dp_hip_output_due(ctx, itermination, flag) bind(C, name="pic1dp_hip_output_due") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: itermination
    integer(c_int32_t), intent(out) :: flag
    integer(c_int) :: ierr
  end function pic1dp_hip_output_due
  function pic1dp_hip_output_all(ctx, scalars, nscalars, electric, chargeden, mode_re, mode_im, dist) &
      bind(C, name="pic1dp_hip_output_all") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(inout) :: scalars(*)
    integer(c_int32_t), value :: nscalars
    real(c_double), intent(inout) :: electric(*)
    real(c_double), intent(inout) :: chargeden(*)
    real(c_double), intent(inout) :: mode_re(*)
    real(c_double), intent(inout) :: mode_im(*)
    real(c_double), intent(inout) :: dist(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_output_all
  function pic1dp_hip_steps_to_output(ctx, nsteps) bind(C, name="pic1dp_hip_steps_to_output") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), intent(out) :: nsteps
    integer(c_int) :: ierr
  end function pic1dp_hip_steps_to_output
  function pic1dp_hip_check_state(ctx, deep) bind(C, name="pic1dp_hip_check_state") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: deep
    integer(c_int) :: ierr
  end function pic1dp_hip_check_state
  function pic1dp_hip_get_field(ctx, electric, chargeden, mode_re, mode_im) bind(C, name="pic1dp_hip_get_field") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(inout) :: electric(*)
    real(c_double), intent(inout) :: chargeden(*)
    real(c_double), intent(inout) :: mode_re(*)
    real(c_double), intent(inout) :: mode_im(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_get_field
  function pic1dp_hip_chargeden_state(ctx, kept_mode_only) bind(C, name="pic1dp_hip_chargeden_state") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), intent(out) :: kept_mode_only
    integer(c_int) :: ierr
  end function pic1dp_hip_chargeden_state
  function pic1dp_hip_set_electric(ctx, electric) bind(C, name="pic1dp_hip_set_electric") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(in) :: electric(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_set_electric
  function pic1dp_hip_set_chargeden(ctx, chargeden) bind(C, name="pic1dp_hip_set_chargeden") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(in) :: chargeden(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_set_chargeden
  function pic1dp_hip_field_energy(ctx, energy) bind(C, name="pic1dp_hip_field_energy") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(out) :: energy
    integer(c_int) :: ierr
  end function pic1dp_hip_field_energy
  function pic1dp_hip_energy_history(ctx, energy, nmax, count) bind(C, name="pic1dp_hip_energy_history") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(inout) :: energy(*)
    integer(c_int64_t), value :: nmax
    integer(c_int64_t), intent(out) :: count
    integer(c_int) :: ierr
  end function pic1dp_hip_energy_history
  function pic1dp_hip_energy_history_reset(ctx) bind(C, name="pic1dp_hip_energy_history_reset") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int) :: ierr
  end function pic1dp_hip_energy_history_reset
  function pic1dp_hip_energy_sums(ctx, ispecies, sums) bind(C, name="pic1dp_hip_energy_sums") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: ispecies
    real(c_double), intent(inout) :: sums(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_energy_sums
  function pic1dp_hip_cell_indices(ctx, ispecies, ix, count) bind(C, name="pic1dp_hip_cell_indices") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: ispecies
    integer(c_int32_t), intent(inout) :: ix(*)
    integer(c_int64_t), intent(inout) :: count(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_cell_indices
  function pic1dp_hip_output_scalars(ctx, realbuf, n) bind(C, name="pic1dp_hip_output_scalars") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(inout) :: realbuf(*)
    integer(c_int32_t), value :: n
    integer(c_int) :: ierr
  end function pic1dp_hip_output_scalars
  function pic1dp_hip_ptcldist(ctx, ispecies, finish, markr_xv, total_xv, pertb_xv, markr_v, total_v, pertb_v) &
      bind(C, name="pic1dp_hip_ptcldist") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: ispecies
    integer(c_int32_t), value :: finish
    real(c_double), intent(inout) :: markr_xv(*), total_xv(*), pertb_xv(*)
    real(c_double), intent(inout) :: markr_v(*), total_v(*), pertb_v(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_ptcldist
  function pic1dp_hip_output_scalars_from(ctx, sums, realbuf, n) bind(C, name="pic1dp_hip_output_scalars_from") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(in) :: sums(*)
    real(c_double), intent(inout) :: realbuf(*)
    integer(c_int32_t), value :: n
    integer(c_int) :: ierr
  end function pic1dp_hip_output_scalars_from
  function pic1dp_hip_ptcldist_finish(ctx, ispecies, markr_xv, total_xv, pertb_xv, markr_v, total_v, pertb_v) &
      bind(C, name="pic1dp_hip_ptcldist_finish") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: ispecies
    real(c_double), intent(inout) :: markr_xv(*), total_xv(*), pertb_xv(*)
    real(c_double), intent(inout) :: markr_v(*), total_v(*), pertb_v(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_ptcldist_finish
  function pic1dp_hip_charge_local(ctx, charge2) bind(C, name="pic1dp_hip_charge_local") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(inout) :: charge2(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_charge_local
  function pic1dp_hip_charge_reduced(ctx, charge1) bind(C, name="pic1dp_hip_charge_reduced") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(in) :: charge1(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_charge_reduced
  function pic1dp_hip_comm_unique_id(id) bind(C, name="pic1dp_hip_comm_unique_id") result(ierr)
    import
    integer(c_signed_char), intent(inout) :: id(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_comm_unique_id
  function pic1dp_hip_comm_init(ctx, id) bind(C, name="pic1dp_hip_comm_init") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_signed_char), intent(in) :: id(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_comm_init
  function pic1dp_hip_predict_kind(ctx, kind) bind(C, name="pic1dp_hip_predict_kind") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), intent(out) :: kind
    integer(c_int) :: ierr
  end function pic1dp_hip_predict_kind
  function pic1dp_hip_set_seed_offset(ctx, offset) bind(C, name="pic1dp_hip_set_seed_offset") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: offset
    integer(c_int) :: ierr
  end function pic1dp_hip_set_seed_offset
  function pic1dp_hip_set_output_fusion(ctx, on) bind(C, name="pic1dp_hip_set_output_fusion") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: on
    integer(c_int) :: ierr
  end function pic1dp_hip_set_output_fusion
  function pic1dp_hip_comm_available() bind(C, name="pic1dp_hip_comm_available") result(ierr)
    import
    integer(c_int) :: ierr
  end function pic1dp_hip_comm_available
  function pic1dp_hip_xchg_create(ctx, handle) bind(C, name="pic1dp_hip_xchg_create") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_signed_char), intent(out) :: handle(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_xchg_create
  function pic1dp_hip_xchg_connect(ctx, handles) bind(C, name="pic1dp_hip_xchg_connect") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_signed_char), intent(in) :: handles(*)
    integer(c_int) :: ierr
  end function pic1dp_hip_xchg_connect
  function pic1dp_hip_set_allreduce(ctx, kind) bind(C, name="pic1dp_hip_set_allreduce") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: kind
    integer(c_int) :: ierr
  end function pic1dp_hip_set_allreduce
  function pic1dp_hip_xchg_info(ctx, memkind, exchanges) bind(C, name="pic1dp_hip_xchg_info") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), intent(out) :: memkind
    integer(c_int64_t), intent(out) :: exchanges
    integer(c_int) :: ierr
  end function pic1dp_hip_xchg_info
  function pic1dp_hip_xchg_time(ctx, ms, exchanges_timed, reset) bind(C, name="pic1dp_hip_xchg_time") result(ierr)
    import
    type(c_ptr), value :: ctx
    real(c_double), intent(out) :: ms
    integer(c_int64_t), intent(out) :: exchanges_timed
    integer(c_int32_t), value :: reset
    integer(c_int) :: ierr
  end function pic1dp_hip_xchg_time
  function pic1dp_hip_timers_enable(ctx, on) bind(C, name="pic1dp_hip_timers_enable") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: on
    integer(c_int) :: ierr
  end function pic1dp_hip_timers_enable
  function pic1dp_hip_timer_ms(ctx, iwt, ms) bind(C, name="pic1dp_hip_timer_ms") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: iwt
    real(c_double), intent(out) :: ms
    integer(c_int) :: ierr
  end function pic1dp_hip_timer_ms
  function pic1dp_hip_timers_reset(ctx) bind(C, name="pic1dp_hip_timers_reset") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int) :: ierr
  end function pic1dp_hip_timers_reset
  function pic1dp_hip_set_launch(ctx, threads, blocks_per_cu) bind(C, name="pic1dp_hip_set_launch") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: threads
    integer(c_int32_t), value :: blocks_per_cu
    integer(c_int) :: ierr
  end function pic1dp_hip_set_launch
  function pic1dp_hip_get_stream(ctx, stream) bind(C, name="pic1dp_hip_get_stream") result(ierr)
    import
    type(c_ptr), value :: ctx
    type(c_ptr), intent(out) :: stream
    integer(c_int) :: ierr
  end function pic1dp_hip_get_stream
  function pic1dp_hip_kernel_stats(ctx, which, ms, launches) bind(C, name="pic1dp_hip_kernel_stats") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: which
    real(c_double), intent(out) :: ms
    integer(c_int64_t), intent(out) :: launches
    integer(c_int) :: ierr
  end function pic1dp_hip_kernel_stats
  function pic1dp_hip_kernel_stats_enable(ctx, on) bind(C, name="pic1dp_hip_kernel_stats_enable") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: on
    integer(c_int) :: ierr
  end function pic1dp_hip_kernel_stats_enable
  function pic1dp_hip_kernel_bytes(ctx, which, read_bytes, written_bytes, carry_bytes, name, name_len) &
      bind(C, name="pic1dp_hip_kernel_bytes") result(ierr)
    import
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: which
    real(c_double), intent(out) :: read_bytes
    real(c_double), intent(out) :: written_bytes
    real(c_double), intent(out) :: carry_bytes
    character(kind=c_char), intent(out) :: name(*)
    integer(c_int32_t), value :: name_len
    integer(c_int) :: ierr
  end function pic1dp_hip_kernel_bytes
end interface

! called by pic1dp_hip_check before it stops the program on an error (a host with buffered output flushes it there)
abstract interface
  subroutine pic1dp_hip_abort_hook()
  end subroutine pic1dp_hip_abort_hook
end interface
procedure(pic1dp_hip_abort_hook), pointer, save :: pic1dp_hip_on_abort => null()

contains

! message of the last failed call as a Fortran string
function pic1dp_hip_last_error() result(text)
  character(len=:), allocatable :: text
  type(c_ptr) :: p
  character(kind=c_char), pointer :: s(:)
  integer :: n, i
  p = pic1dp_hip_last_error_c()
  text = ''
  if (.not. c_associated(p)) return
  call c_f_pointer(p, s, [512])
  n = 0
  do while (n < 512)
    if (s(n + 1) == c_null_char) exit
    n = n + 1
  end do
  allocate (character(len=n) :: text)
  do i = 1, n
    text(i:i) = s(i)
  end do
end function pic1dp_hip_last_error

! the CHKERRQ of this binding: print the message and stop on a non-zero code
subroutine pic1dp_hip_check(ierr, where)
  integer(c_int), intent(in) :: ierr
  character(len=*), intent(in) :: where
  if (ierr /= 0) then
    write (*, '(5a, i0, a)') 'pic1dp_hip: ', where, ': ', pic1dp_hip_last_error(), ' (error ', ierr, ')'
    ! what the host still holds in memory goes out first (the record output_all has assembled: the reference has written
    ! it by the time a later call fails)
    call pic1dp_hip_abort()
  end if
end subroutine pic1dp_hip_check

! stop on an error, after the host's hook has run (once: it is taken off first, a failure inside it cannot come back)
subroutine pic1dp_hip_abort()
  procedure(pic1dp_hip_abort_hook), pointer :: hook
  if (associated(pic1dp_hip_on_abort)) then
    hook => pic1dp_hip_on_abort
    pic1dp_hip_on_abort => null()
    call hook()
  end if
  stop 1
end subroutine pic1dp_hip_abort

end module pic1dp_hip
