"""ctypes binding of libpic1dp_hip.so (the C ABI of include/pic1dp_hip.h).

There is no Python or CPU fallback: if the library is missing or cannot be
loaded this module raises, loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PIC1DP_LIB: alternative build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("PIC1DP_LIB") or os.path.join(HERE, "lib", "libpic1dp_hip.so")

MAX_SPECIES = 8
MAX_MODES = 4096
MAX_INIT_MODES = 16
COMM_ID_BYTES = 128
XCHG_HANDLE_BYTES = 64
ABI_VERSION = 5
MAX_OPT = 32

ERR_NAMES = {1: "ARG", 2: "HIP", 3: "NODEVICE", 4: "STATE", 5: "COMM", 6: "RNG", 7: "NOMEM"}


class Pic1dpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pic1dp_hip error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code


class Input(C.Structure):
    """struct pic1dp_input: the run-time mirror of src/pic1dp_input.F90"""
    _fields_ = [
        ("abi_version", C.c_int32), ("ntime_max", C.c_int32), ("linear", C.c_int32),
        ("iptcldist", C.c_int32), ("nspecies", C.c_int32), ("nmode", C.c_int32),
        ("init_nmode", C.c_int32), ("deltaf", C.c_int32), ("imarker", C.c_int32),
        ("nx", C.c_int32), ("nv", C.c_int32), ("iptclshape", C.c_int32),
        ("nx_opd", C.c_int32), ("nv_opd", C.c_int32), ("multirand_al_int", C.c_int32),
        ("multirand_seed_type", C.c_int32), ("multirand_warmup", C.c_int32),
        ("multirand_selftest", C.c_int32),
        ("nparticle_max", C.c_int64),
        ("species_nparticle_init", C.c_int64 * MAX_SPECIES),
        ("time_max", C.c_double), ("lx", C.c_double), ("dt", C.c_double),
        ("v_max", C.c_double), ("output_interval", C.c_double),
        ("species_charge", C.c_double * MAX_SPECIES),
        ("species_mass", C.c_double * MAX_SPECIES),
        ("species_temperature", C.c_double * MAX_SPECIES),
        ("species_temperature2", C.c_double * MAX_SPECIES),
        ("species_density", C.c_double * MAX_SPECIES),
        ("species_v0", C.c_double * MAX_SPECIES),
        ("modes", C.c_int32 * MAX_MODES),
        ("init_mode", C.c_int32 * MAX_INIT_MODES),
        ("init_mode_cos", C.c_double * MAX_INIT_MODES),
        ("init_mode_sin", C.c_double * MAX_INIT_MODES),
        ("nmerge", C.c_int32), ("nremove", C.c_int32), ("nsplit", C.c_int32),
        ("typeremove", C.c_int32), ("split_ngroup", C.c_int32), ("reserved0", C.c_int32),
        ("remove_frac", C.c_double), ("split_dv_sig_frac", C.c_double),
        ("tmerge", C.c_double * MAX_OPT), ("thshmerge", C.c_double * MAX_OPT),
        ("tremove", C.c_double * MAX_OPT), ("thshremove", C.c_double * MAX_OPT),
        ("tsplit", C.c_double * MAX_OPT), ("thshsplit", C.c_double * MAX_OPT),
    ]


class Layout(C.Structure):
    """struct pic1dp_layout"""
    _fields_ = [("rank", C.c_int32), ("nranks", C.c_int32), ("npe", C.c_int32),
                ("device", C.c_int32)]


_P = C.c_void_p
_D = C.POINTER(C.c_double)
_INP = C.POINTER(Input)

# name -> (argtypes); every function returns int unless noted
SIGNATURES = {
    "pic1dp_hip_abi_version": [],
    "pic1dp_hip_last_error": [],            # returns const char*
    "pic1dp_hip_device_count": [],
    "pic1dp_hip_tuning_build": [],
    "pic1dp_hip_input_defaults": [_INP],
    "pic1dp_hip_input_size": [],
    "pic1dp_hip_input_validate": [_INP, C.POINTER(Layout)],
    "pic1dp_hip_block_sizes": [_INP, C.c_int32, C.c_int32, C.c_int32,
                               C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "pic1dp_hip_host_particle_load": [_INP, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int64],
    "pic1dp_hip_host_multirand_int64": [C.c_int32] * 5 + [_P, C.c_int64],
    "pic1dp_hip_create": [_INP, C.POINTER(Layout), C.POINTER(_P)],
    "pic1dp_hip_destroy": [_P],
    "pic1dp_hip_local_sizes": [_P, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "pic1dp_hip_particle_load": [_P],
    "pic1dp_hip_particles_upload": [_P, C.c_int32, _P, _P, _P, _P, C.c_int64, C.c_int64],
    "pic1dp_hip_particles_download": [_P, C.c_int32, _P, _P, _P, _P, C.c_int64],
    "pic1dp_hip_particles_download_bak": [_P, C.c_int32, _P, _P, _P, C.c_int64],
    "pic1dp_hip_collect_charge": [_P],
    "pic1dp_hip_solve_field": [_P],
    "pic1dp_hip_push": [_P, C.c_int32],
    "pic1dp_hip_particle_optimize": [_P, C.c_int32, C.POINTER(C.c_int32)],
    "pic1dp_hip_substep": [_P, C.c_int32],
    "pic1dp_hip_step": [_P, C.c_int32],
    "pic1dp_hip_set_step_mode": [_P, C.c_int32],
    "pic1dp_hip_set_output_fusion": [_P, C.c_int32],
    "pic1dp_hip_set_seed_offset": [_P, C.c_int32],
    "pic1dp_hip_predict_kind": [_P, C.POINTER(C.c_int32)],
    "pic1dp_hip_get_field_half": [_P, _P],
    "pic1dp_hip_set_field_solver": [_P, C.c_int32],
    "pic1dp_hip_sync": [_P],
    "pic1dp_hip_get_time": [_P, C.POINTER(C.c_int32), _D],
    "pic1dp_hip_set_time": [_P, C.c_int32, C.c_double],
    "pic1dp_hip_check_termination": [_P, C.POINTER(C.c_int32)],
    "pic1dp_hip_output_due": [_P, C.c_int32, C.POINTER(C.c_int32)],
    "pic1dp_hip_steps_to_output": [_P, C.POINTER(C.c_int32)],
    "pic1dp_hip_check_state": [_P, C.c_int32],
    "pic1dp_hip_output_all": [_P, _P, C.c_int32, _P, _P, _P, _P, _P],
    "pic1dp_hip_get_field": [_P, _P, _P, _P, _P],
    "pic1dp_hip_chargeden_state": [_P, C.POINTER(C.c_int32)],
    "pic1dp_hip_set_electric": [_P, _P],
    "pic1dp_hip_set_chargeden": [_P, _P],
    "pic1dp_hip_field_energy": [_P, _D],
    "pic1dp_hip_energy_history": [_P, _P, C.c_int64, C.POINTER(C.c_int64)],
    "pic1dp_hip_energy_history_reset": [_P],
    "pic1dp_hip_energy_sums": [_P, C.c_int32, _P],
    "pic1dp_hip_cell_indices": [_P, C.c_int32, _P, _P],
    "pic1dp_hip_output_scalars": [_P, _P, C.c_int32],
    "pic1dp_hip_ptcldist": [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P],
    "pic1dp_hip_charge_local": [_P, _P],
    "pic1dp_hip_charge_reduced": [_P, _P],
    "pic1dp_hip_comm_unique_id": [_P],
    "pic1dp_hip_comm_init": [_P, _P],
    "pic1dp_hip_comm_available": [],
    "pic1dp_hip_xchg_create": [_P, _P],
    "pic1dp_hip_xchg_connect": [_P, _P],
    "pic1dp_hip_set_allreduce": [_P, C.c_int32],
    "pic1dp_hip_xchg_info": [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int64)],
    "pic1dp_hip_xchg_time": [_P, _D, C.POINTER(C.c_int64), C.c_int32],
    "pic1dp_hip_timers_enable": [_P, C.c_int32],
    "pic1dp_hip_timer_ms": [_P, C.c_int32, _D],
    "pic1dp_hip_timers_reset": [_P],
    "pic1dp_hip_set_launch": [_P, C.c_int32, C.c_int32],
    "pic1dp_hip_get_stream": [_P, C.POINTER(_P)],
    "pic1dp_hip_kernel_stats": [_P, C.c_int32, _D, C.POINTER(C.c_int64)],
    "pic1dp_hip_kernel_stats_enable": [_P, C.c_int32],
    "pic1dp_hip_output_scalars_from": [_P, _P, _P, C.c_int32],
    "pic1dp_hip_ptcldist_finish": [_P, C.c_int32, _P, _P, _P, _P, _P, _P],
    "pic1dp_hip_kernel_bytes": [_P, C.c_int32, _D, _D, _D, C.c_char_p, C.c_int32],
}

_lib = None


def load():
    """load the HIP library; raises if it is absent (no fallback exists)"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python pic1dp_amd/build.py` "
            "(hipcc --offload-arch=gfx950). pic1dp_amd has no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, args in SIGNATURES.items():
        f = getattr(L, name)  # AttributeError here = ABI mismatch, also loud
        f.argtypes = args
        f.restype = C.c_char_p if name == "pic1dp_hip_last_error" else C.c_int
    got = L.pic1dp_hip_abi_version()
    if got != ABI_VERSION:
        raise ImportError("libpic1dp_hip.so ABI %d, Python binding expects %d" % (got, ABI_VERSION))
    if L.pic1dp_hip_input_size() != C.sizeof(Input):
        raise ImportError("struct pic1dp_input: library %d bytes, binding %d bytes"
                          % (L.pic1dp_hip_input_size(), C.sizeof(Input)))
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = load().pic1dp_hip_last_error()
        raise Pic1dpError(rc, msg.decode() if msg else "")
