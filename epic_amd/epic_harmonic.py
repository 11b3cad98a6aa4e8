"""ctypes binding of the MI355X-native ``libepic.so`` (epic_amd/lib/libepic.so).

Host-side mirror of the reference's binding module (libepic/python/epic/epic_harmonic.py:38-124): the same
``EpicHarmonic`` structure (field order/types of libepic/include/epic/harmonic/harmonic.h:44-64) and the same
``_epic.<function>.argtypes`` table, so code written against ``epic_harmonic._epic`` keeps working.  The library is
the product: if it is missing this module raises at import -- there is no Python or CPU stand-in for the GPU path.
"""
import ctypes as ct
import os

_HERE = os.path.dirname(os.path.realpath(__file__))
# EPIC_LIB: another build of the same library (A/B timing of kernel variants, tools/ab_bench.sh)
LIB_PATH = os.environ.get("EPIC_LIB") or os.path.join(_HERE, "lib", "libepic.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "epic_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C epic_amd/csrc` (hipcc, --offload-arch=gfx950)" % LIB_PATH)



def _preload_hip_runtime():
    """One HIP runtime per process.  libepic.so needs ``libamdhip64.so.7``; PyTorch-ROCm ships its own copy with
    that SONAME and loads it by file name.  Whoever comes second must find the first one already loaded, so when
    PyTorch is installed its copy is loaded here, before libepic.so (the dynamic loader then satisfies libepic's
    NEEDED entry by SONAME, and a later ``import torch`` finds its own file already mapped).  Without PyTorch the
    system runtime under /opt/rocm is used as for any drop-in libepic.so."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ct.CDLL(cand, mode=ct.RTLD_GLOBAL)
        except OSError:
            pass


_preload_hip_runtime()
_epic = ct.CDLL(LIB_PATH)


class EpicHarmonic(ct.Structure):
    """The C struct Harmonic (80 bytes)."""

    _fields_ = [("n", ct.c_uint),
                ("m", ct.POINTER(ct.c_uint)),
                ("u", ct.POINTER(ct.c_float)),
                ("locked", ct.POINTER(ct.c_uint)),
                ("epsilon", ct.c_float),
                ("delta", ct.c_float),
                ("numIterationsToStaggerCheck", ct.c_uint),
                ("currentIteration", ct.c_uint),
                ("d_m", ct.POINTER(ct.c_uint)),
                ("d_u", ct.POINTER(ct.c_float)),
                ("d_locked", ct.POINTER(ct.c_uint)),
                ("d_delta", ct.POINTER(ct.c_float)),
                ]


assert ct.sizeof(EpicHarmonic) == 80

_H = ct.POINTER(EpicHarmonic)
_UP = ct.POINTER(ct.c_uint)

# name -> argtypes; every function returns int.  Grouped as the reference's headers are.
_SIGNATURES = {
    # harmonic_cpu.h
    "harmonic_complete_cpu": (_H,),
    "harmonic_update_cpu": (_H,),
    "harmonic_update_and_check_cpu": (_H,),
    # harmonic_gpu.h
    "harmonic_complete_gpu": (_H, ct.c_uint),
    "harmonic_initialize_gpu": (_H, ct.c_uint),
    "harmonic_execute_gpu": (_H, ct.c_uint),
    "harmonic_uninitialize_gpu": (_H,),
    "harmonic_update_gpu": (_H, ct.c_uint),
    "harmonic_update_and_check_gpu": (_H, ct.c_uint),
    "harmonic_get_potential_values_gpu": (_H,),
    # harmonic_model_gpu.h
    "harmonic_initialize_dimension_size_gpu": (_H,),
    "harmonic_uninitialize_dimension_size_gpu": (_H,),
    "harmonic_initialize_potential_values_gpu": (_H,),
    "harmonic_uninitialize_potential_values_gpu": (_H,),
    "harmonic_initialize_locked_gpu": (_H,),
    "harmonic_uninitialize_locked_gpu": (_H,),
    "harmonic_update_model_gpu": (_H,),
    # harmonic_utilities_{cpu,gpu}.h
    "harmonic_utilities_set_cells_2d_cpu": (_H, ct.c_uint, _UP, _UP),
    "harmonic_utilities_set_cells_2d_gpu": (_H, ct.c_uint, ct.c_uint, _UP, _UP),
    # harmonic_path_cpu.h (reference parameters are pointers at the ABI level)
    "harmonic_compute_potential_2d_cpu": (_H, ct.c_float, ct.c_float, ct.POINTER(ct.c_float)),
    "harmonic_compute_gradient_2d_cpu": (_H, ct.c_float, ct.c_float, ct.c_float, ct.POINTER(ct.c_float),
                                         ct.POINTER(ct.c_float)),
    "harmonic_compute_path_2d_cpu": (_H, ct.c_float, ct.c_float, ct.c_float, ct.c_float, ct.c_uint, _UP,
                                     ct.POINTER(ct.POINTER(ct.c_float))),
    "harmonic_free_path_cpu": (ct.POINTER(ct.POINTER(ct.c_float)),),
    # harmonic_legacy_cpu.h / harmonic_legacy_path_cpu.h
    "harmonic_legacy_sor_2d_float_cpu": (ct.c_uint, ct.c_uint, ct.c_float, ct.c_float, _UP, ct.POINTER(ct.c_float), _UP),
    "harmonic_legacy_sor_2d_double_cpu": (ct.c_uint, ct.c_uint, ct.c_double, ct.c_double, _UP, ct.POINTER(ct.c_double),
                                          _UP),
    "harmonic_legacy_sor_2d_long_double_cpu": (ct.c_uint, ct.c_uint, ct.c_longdouble, ct.c_longdouble, _UP,
                                               ct.POINTER(ct.c_longdouble), _UP),
    "harmonic_legacy_compute_potential_2d_cpu": (ct.c_uint, ct.c_uint, _UP, ct.POINTER(ct.c_double), ct.c_double,
                                                 ct.c_double, ct.POINTER(ct.c_double)),
    "harmonic_legacy_compute_gradient_2d_cpu": (ct.c_uint, ct.c_uint, _UP, ct.POINTER(ct.c_double), ct.c_double,
                                                ct.c_double, ct.c_double, ct.POINTER(ct.c_double),
                                                ct.POINTER(ct.c_double)),
    "harmonic_legacy_compute_path_2d_cpu": (ct.c_uint, ct.c_uint, _UP, ct.POINTER(ct.c_double), ct.c_double,
                                            ct.c_double, ct.c_double, ct.c_double, ct.c_uint, ct.c_int, _UP,
                                            ct.POINTER(ct.POINTER(ct.c_double))),
    "harmonic_legacy_free_path_cpu": (ct.POINTER(ct.POINTER(ct.c_double)),),
    # include/epic_hip.h (extensions)
    "epic_hip_device_count": (),
    "epic_hip_update_n_gpu": (_H, ct.c_uint, ct.c_int),
    "epic_hip_timed_sweeps_gpu": (_H, ct.c_uint, ct.c_uint, ct.POINTER(ct.c_float)),
    "epic_hip_set_rows_per_task": (_H, ct.c_uint),
    "epic_hip_iterations_per_pass": (_H,),
    "epic_hip_tile_iterations": (_H,),
    "epic_hip_fused_rows_per_task": (_H,),
    "epic_hip_finish_iteration": (_H,),
    "epic_hip_set_math_mode": (_H, ct.c_int),
    "epic_hip_set_scheme": (_H, ct.c_int),
    "epic_hip_set_activity_tracking": (_H, ct.c_int),
    "epic_hip_compute_paths_2d_gpu": (_H, ct.c_uint, ct.POINTER(ct.c_float), ct.c_float, ct.c_float, ct.c_uint, _UP,
                                      ct.POINTER(ct.c_int), ct.POINTER(ct.c_float)),
    "epic_hip_compute_path_2d_gpu": (_H, ct.c_float, ct.c_float, ct.c_float, ct.c_float, ct.c_uint, _UP,
                                     ct.POINTER(ct.POINTER(ct.c_float))),
    "epic_hip_activity_stats": (_H, ct.POINTER(ct.c_ulonglong), ct.POINTER(ct.c_ulonglong)),
    "epic_hip_activity_stats2": (_H, ct.POINTER(ct.c_ulonglong), ct.POINTER(ct.c_ulonglong), ct.POINTER(ct.c_ulonglong)),
    "epic_hip_work_done": (_H, ct.POINTER(ct.c_double), ct.c_int),
    "epic_hip_eval_math": (ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int, ct.c_void_p),
    "epic_hip_get_layout": (_H, _UP, ct.POINTER(ct.c_size_t), ct.POINTER(ct.c_size_t)),
    "epic_hip_device_layout": (_H, ct.c_int, ct.POINTER(ct.c_int), _UP, _UP, _UP),
    "epic_hip_multi_report": (_H, ct.c_char_p, ct.c_size_t),
    "epic_hip_config_dump": (_H, ct.c_char_p, ct.c_size_t),
    "epic_hip_config_reload": (_H,),
    "epic_hip_pack_mask_2d": (ct.c_void_p, ct.c_uint, ct.c_uint, ct.c_uint, ct.c_int, ct.c_int, ct.c_void_p,
                              ct.c_void_p),
    "epic_hip_sweep_2d": (ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_uint, ct.c_uint, ct.c_uint, ct.c_uint,
                          ct.c_uint, ct.c_int, ct.c_void_p, ct.c_void_p),
    "epic_hip_sweep2_2d": (ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_uint, ct.c_uint, ct.c_uint, ct.c_int, ct.c_void_p),
    "epic_hip_fuse_masks_2d": (ct.c_void_p, ct.c_uint, ct.c_uint, ct.c_void_p, ct.c_void_p),
    "epic_hip_sweeps_2d": (ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_uint, ct.c_uint, ct.c_uint, ct.c_uint, ct.c_uint,
                           ct.c_int, ct.POINTER(ct.c_int), ct.c_void_p),
    "epic_hip_sweep_rb_2d": (ct.c_void_p, ct.c_void_p, ct.c_uint, ct.c_uint, ct.c_uint, ct.c_uint, ct.c_uint, ct.c_int,
                             ct.c_int, ct.c_void_p, ct.c_void_p),
}

for _name, _args in _SIGNATURES.items():
    if os.environ.get("EPIC_LIB") and _name.startswith("epic_hip_") and not hasattr(_epic, _name):
        continue  # an older build named for A/B timing (tools/ab_bench.sh) may lack a newer extension entry point
    _fn = getattr(_epic, _name)  # AttributeError here = the library does not export what its headers declare
    _fn.argtypes = _args
    _fn.restype = ct.c_int

_epic.epic_hip_version.argtypes = ()
_epic.epic_hip_version.restype = ct.c_char_p
_epic.epic_hip_mask_words_2d.argtypes = (ct.c_uint, ct.c_uint)
_epic.epic_hip_mask_words_2d.restype = ct.c_size_t
_epic.epic_hip_mask_words_fused_2d.argtypes = (ct.c_uint, ct.c_uint)
_epic.epic_hip_mask_words_fused_2d.restype = ct.c_size_t
_epic.epic_hip_pitch_for_cols.argtypes = (ct.c_uint,)
_epic.epic_hip_pitch_for_cols.restype = ct.c_uint

# return codes (include/epic/epic_abi.h; reference libepic/include/epic/error_codes.h:31-46)
EPIC_SUCCESS = 0
EPIC_SUCCESS_AND_CONVERGED = 1
EPIC_ERROR_INVALID_DATA = 2
EPIC_ERROR_INVALID_CUDA_PARAM = 3
EPIC_ERROR_DEVICE_MALLOC = 4
MATH_PRECISE = 0
MATH_FAST = 1
MATH_TOL = 4
SCHEME_JACOBI = 0
SCHEME_REDBLACK = 1
EPIC_CELL_TYPE_GOAL = 0
EPIC_CELL_TYPE_OBSTACLE = 1
EPIC_CELL_TYPE_FREE = 2


def config_dump(h):
    """epic_hip_config_dump as a dict: the context's Config (every EPIC_HIP_* knob as read when it was created), its state and the
    path a batch of plain iterations takes now; None without device state."""
    import json

    buf = ct.create_string_buffer(1 << 14)
    n = _epic.epic_hip_config_dump(h, buf, len(buf))
    return json.loads(buf.value.decode()) if n > 0 else None
