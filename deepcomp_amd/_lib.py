"""ctypes binding of libdcomp_hip.so (include/dcomp.h).  Fails loudly when the HIP extension is missing:
there is no CPU fallback in the product path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DCOMP_LIB') or os.path.join(_HERE, 'csrc', 'libdcomp_hip.so')   # DCOMP_LIB: tools/ab/ablate.py timing variants

OK, EINVAL, EHIP, EACTION, ETAPE, EPOS, EUNSUPPORTED, EABI = 0, -1, -2, -3, -4, -5, -6, -7
ABI_VERSION = 3                 # include/dcomp.h DCOMP_ABI_VERSION: what the struct declarations below describe
CENTRAL, MULTI = 0, 1
REWARD = {'avg': 0, 'sum': 1, 'min': 2}
SHARING = {'resource-fair': 0, 'rate-fair': 1, 'max-cap': 2, 'proportional-fair': 3}
UTILITY = {'log': 0, 'step': 1}
RNG_TAPE, RNG_PHILOX = 0, 1
MAX_BS, MAX_UE = 64, 1024
SPECIAL_MAX_UE = 256             # up to here the specialised kernels; 257 ... 1 024 UEs per env: the generic kernel (csrc/dcomp_big.h)
MASK32_MAX_BS = 32              # up to here: one 32-bit connection mask per UE and the specialised kernels; beyond: state.conn_hi + the generic kernel

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_fp = ctypes.POINTER(ctypes.c_float)


class DcompCfg(ctypes.Structure):
    _fields_ = [('num_envs', ctypes.c_int32), ('num_ue', ctypes.c_int32), ('num_bs', ctypes.c_int32),
                ('map_w', ctypes.c_int32), ('map_h', ctypes.c_int32), ('env_kind', ctypes.c_int32),
                ('reward_agg', ctypes.c_int32), ('rng_mode', ctypes.c_int32), ('tape_depth', ctypes.c_int32),
                ('device', ctypes.c_int32), ('max_ues', ctypes.c_int32), ('seed', ctypes.c_uint64),
                ('env_id_base', ctypes.c_int64), ('bs_x', _dp), ('bs_y', _dp), ('bs_sharing', _ip),
                ('ue_util', _ip), ('ue_dr_req', _fp), ('ue_vel_lo', _ip), ('ue_vel_hi', _ip),
                ('ue_init_x', _ip), ('ue_init_y', _ip), ('ue_pause_duration', _ip), ('ue_border_buffer', _ip),
                ('ue_velocity', _dp)]


class DcompState(ctypes.Structure):
    _fields_ = [('pos', ctypes.c_void_p), ('mv', ctypes.c_void_p), ('conn', ctypes.c_void_p),
                ('ewma', ctypes.c_void_p), ('flags', ctypes.c_void_p), ('conn_since', ctypes.c_void_p),
                ('uid', ctypes.c_void_p), ('orig_consumed', ctypes.c_void_p), ('conn_hi', ctypes.c_void_p)]


class DcompOut(ctypes.Structure):
    _fields_ = [('obs', ctypes.c_void_p), ('reward', ctypes.c_void_p), ('sum_utility', ctypes.c_void_p),
                ('ue_dr', ctypes.c_void_p), ('ue_utility', ctypes.c_void_p), ('reward_before', ctypes.c_void_p),
                ('obs_compact', ctypes.c_void_p)]


class DcompTape(ctypes.Structure):
    _fields_ = [('pos0', ctypes.c_void_p), ('triples', ctypes.c_void_p), ('num_ids', ctypes.c_int32)]


class DcompRolloutOpts(ctypes.Structure):
    _fields_ = [('every_step', ctypes.c_int32), ('horizon', ctypes.c_int32), ('new_episode_draws', ctypes.c_int32),
                ('policy_loop', ctypes.c_int32), ('ev_n_remove', ctypes.c_void_p), ('ev_n_add', ctypes.c_void_p),
                ('ev_remove_idx', ctypes.c_void_p), ('ev_add_xy', ctypes.c_void_p)]


class DcompEvents(ctypes.Structure):
    _fields_ = [('n_remove', ctypes.c_int32), ('n_add', ctypes.c_int32), ('remove_idx', ctypes.c_void_p),
                ('add_xy', ctypes.c_void_p)]


class DcompPolicy(ctypes.Structure):
    _fields_ = [('policy', ctypes.c_int32), ('obs_kind', ctypes.c_int32), ('num_envs', ctypes.c_int32),
                ('num_ue', ctypes.c_int32), ('num_bs', ctypes.c_int32), ('num_active', ctypes.c_int32),
                ('epsilon', ctypes.c_float), ('cluster_mask', ctypes.c_void_p)]


POLICY = {'3gpp': 0, 'fullcomp': 1, 'dynamic': 2, 'cluster': 3}

EXPORTS = ['dcomp_abi_version', 'dcomp_create_v', 'dcomp_create', 'dcomp_destroy', 'dcomp_state_sizes', 'dcomp_obs_dim', 'dcomp_reset', 'dcomp_step',
           'dcomp_step_dyn', 'dcomp_num_ue',
           'dcomp_rollout', 'dcomp_rollout_ex', 'dcomp_rollout_is_fused', 'dcomp_rollout_fused_for', 'dcomp_lanes_per_env', 'dcomp_needs_conn_hi', 'dcomp_step_kernel_name', 'dcomp_check', 'dcomp_time', 'dcomp_episode', 'dcomp_set_episode', 'dcomp_set_seed', 'dcomp_set_tape', 'dcomp_get_counters', 'dcomp_set_counters', 'dcomp_mt_draw_tape',
           'dcomp_connect_threshold', 'dcomp_connect_boundary_sq', 'dcomp_last_error', 'dcomp_version', 'dcomp_selftest', 'dcomp_heuristic_actions', 'dcomp_set_policy',
           'dcomp_fragment_words', 'dcomp_pack_fragment', 'dcomp_unpack_fragment']

_lib = None


class DcompError(RuntimeError):
    pass


def load():
    """Load the HIP extension; raise (never fall back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build the HIP extension first (python -m deepcomp_amd.build). "
                          "deepcomp_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7: import it first so that this library binds to the SAME HIP
    # runtime (device pointers and streams are shared with torch); loading ours first would pull in /opt/rocm's copy.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.dcomp_create.argtypes = [ctypes.POINTER(DcompCfg), ctypes.POINTER(vp)]        # the ABI-1 symbol: refuses (include/dcomp.h, ABI guard)
    if hasattr(L, 'dcomp_create_v'):
        L.dcomp_create_v.argtypes = [i32] + [ctypes.c_size_t] * 4 + [ctypes.POINTER(DcompCfg), ctypes.POINTER(vp)]
    L.dcomp_destroy.argtypes = [vp]
    L.dcomp_state_sizes.argtypes = [vp] + [ctypes.POINTER(ctypes.c_size_t)] * 6
    L.dcomp_obs_dim.argtypes = [vp, _ip, _ip]
    L.dcomp_reset.argtypes = [vp, ctypes.POINTER(DcompState), ctypes.POINTER(DcompTape), ctypes.POINTER(DcompOut), vp]
    L.dcomp_step.argtypes = [vp, ctypes.POINTER(DcompState), vp, ctypes.POINTER(DcompOut), vp]
    L.dcomp_step_dyn.argtypes = [vp, ctypes.POINTER(DcompState), vp, ctypes.POINTER(DcompOut), ctypes.POINTER(DcompEvents), vp]
    L.dcomp_num_ue.argtypes = [vp]
    L.dcomp_rollout.argtypes = [vp, ctypes.POINTER(DcompState), vp, i32, ctypes.POINTER(DcompOut), vp]
    if os.environ.get('DCOMP_LIB') and not hasattr(L, 'dcomp_rollout_ex'):     # a timing variant built from older sources
        EXPORTS[:] = [n for n in EXPORTS if hasattr(L, n)]
        L.dcomp_rollout_ex = L.dcomp_rollout_is_fused = lambda *a: EUNSUPPORTED
    else:
        L.dcomp_rollout_ex.argtypes = [vp, ctypes.POINTER(DcompState), vp, i32, ctypes.POINTER(DcompOut), ctypes.POINTER(DcompRolloutOpts), vp]
        L.dcomp_rollout_is_fused.argtypes = [vp]
        if hasattr(L, 'dcomp_rollout_fused_for'):
            L.dcomp_rollout_fused_for.argtypes = [vp, i32, i32, i32]
        if hasattr(L, 'dcomp_lanes_per_env'):
            L.dcomp_lanes_per_env.argtypes = [vp]
        if hasattr(L, 'dcomp_needs_conn_hi'):
            L.dcomp_needs_conn_hi.argtypes = [vp]
        if hasattr(L, 'dcomp_step_kernel_name'):
            L.dcomp_step_kernel_name.argtypes = [vp, ctypes.c_char_p, i32]
    L.dcomp_check.argtypes = [vp, ctypes.POINTER(DcompState), vp]
    L.dcomp_time.argtypes = [vp]
    L.dcomp_episode.argtypes = [vp]
    L.dcomp_episode.restype = i64
    L.dcomp_set_episode.argtypes = [vp, i64]
    if hasattr(L, 'dcomp_set_seed'):
        L.dcomp_set_seed.argtypes = [vp, ctypes.c_uint64]
        L.dcomp_set_tape.argtypes = [vp, ctypes.POINTER(DcompTape), i32]
    L.dcomp_get_counters.argtypes = [vp, ctypes.POINTER(i64)]
    L.dcomp_set_counters.argtypes = [vp, ctypes.POINTER(i64)]
    L.dcomp_mt_draw_tape.argtypes = [ctypes.POINTER(DcompCfg), ctypes.POINTER(i64), i32, i32, vp, vp]
    L.dcomp_connect_threshold.restype = ctypes.c_double
    if hasattr(L, 'dcomp_connect_boundary_sq'):
        L.dcomp_connect_boundary_sq.restype = ctypes.c_double
    L.dcomp_last_error.restype = ctypes.c_char_p
    L.dcomp_version.restype = ctypes.c_char_p
    L.dcomp_selftest.argtypes = [i32, i32, vp, vp, vp, i64, vp]
    if hasattr(L, 'dcomp_heuristic_actions'):
        L.dcomp_heuristic_actions.argtypes = [ctypes.POINTER(DcompPolicy), vp, vp, vp]
    if hasattr(L, 'dcomp_set_policy'):
        L.dcomp_set_policy.argtypes = [vp, ctypes.POINTER(DcompPolicy), vp]
    if hasattr(L, 'dcomp_pack_fragment'):
        L.dcomp_fragment_words.argtypes = [i32, i32]
        L.dcomp_pack_fragment.argtypes = [vp, i64, i32, i32, vp, vp, vp]
        L.dcomp_unpack_fragment.argtypes = [vp, i64, i32, i32, vp, vp]
    if os.environ.get('DCOMP_LIB'):              # timing variants built from older sources lack the newest entry points
        EXPORTS[:] = [n for n in EXPORTS if hasattr(L, n)]
    for name in EXPORTS:
        getattr(L, name)
    _lib = L
    return L


def last_error():
    return load().dcomp_last_error().decode()


def create(cfg, handle):
    """dcomp_create_v with the sizes of THIS module's struct declarations: a library built from other headers refuses (DCOMP_EABI)
    instead of reading past a struct."""
    L = load()
    if not hasattr(L, 'dcomp_create_v'):            # DCOMP_LIB timing variants built from ABI-1 sources
        return L.dcomp_create(ctypes.byref(cfg), ctypes.byref(handle))
    return L.dcomp_create_v(ABI_VERSION, ctypes.sizeof(DcompCfg), ctypes.sizeof(DcompState), ctypes.sizeof(DcompOut),
                            ctypes.sizeof(DcompRolloutOpts), ctypes.byref(cfg), ctypes.byref(handle))


def check(rc):
    """Map a negative return code to the exception the reference raises in the same situation."""
    if rc == OK:
        return
    msg = last_error()
    if rc in (EACTION, EPOS):
        raise AssertionError(msg)          # base.py:238, central.py:61, movement.py:165
    if rc == EUNSUPPORTED:
        raise NotImplementedError(msg)     # user.py:92, central.py:73
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == EABI:
        raise ImportError(msg)             # binding and library were built from different headers
    raise DcompError(f"dcomp error {rc}: {msg}")
