"""ctypes binding of the C ABI declared in include/aadff.h (libaadff.so, HIP/gfx950).

There is NO CPU fallback: every compute entry point needs the HIP library and a GPU and
raises RuntimeError otherwise.  torch is imported first so that libaadff.so resolves
`libamdhip64.so.7` to the runtime instance torch already loaded (streams and device
pointers are then shared).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AADFF_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libaadff.so")   # AADFF_LIB: A/B builds (tools/)

ABI_VERSION = 9
MAX_GRID, MAX_KS, MAX_SURF, MAX_AI = 64, 51, 32, 8
SURF_STOP, SURF_SPHERIC, SURF_ASPHERIC = 0, 1, 2


class Surface(C.Structure):
    _fields_ = [("d", C.c_float), ("c", C.c_float), ("k", C.c_float), ("r", C.c_float),
                ("r2", C.c_float), ("r2_shape", C.c_float),
                ("eta_fwd", C.c_float), ("eta_fwd2", C.c_float), ("eta_bwd", C.c_float), ("eta_bwd2", C.c_float),
                ("kind", C.c_int), ("n_ai", C.c_int), ("refract_fwd", C.c_int), ("refract_bwd", C.c_int),
                ("k_gt_m1", C.c_int), ("ai", C.c_float * MAX_AI), ("dai", C.c_float * MAX_AI),
                ("cos2_min_fwd", C.c_float), ("cos2_min_bwd", C.c_float), ("newton_step_tol", C.c_float)]


class LensState(C.Structure):
    _fields_ = [("d_sensor", C.c_float), ("hfov", C.c_float), ("tan_hfov", C.c_float), ("foclen", C.c_float),
                ("fnum", C.c_float), ("n_focus_rays", C.c_int), ("flags", C.c_int), ("pad", C.c_int)]


class LensConst(C.Structure):
    _fields_ = [("n_surf", C.c_int), ("r_last", C.c_float), ("sensor_w", C.c_float), ("sensor_h", C.c_float),
                ("pixel_size", C.c_float), ("enp_z", C.c_float), ("enp_r", C.c_float), ("enp_r2", C.c_float),
                ("enp_r2_shrunk", C.c_float), ("exp_z", C.c_float), ("exp_r_shrunk", C.c_float),
                ("first_d", C.c_float), ("first_r2", C.c_float)]


class Stage(C.Structure):
    """aadff_stage_t: upload folded into a psf_points launch."""
    _fields_ = [("src_host", C.c_void_p), ("dst_dev", C.c_void_p), ("slice_stride", C.c_long),
                ("first_slice", C.c_int), ("generation", C.c_uint), ("counters", C.c_void_p)]


FIT_MAX_LAYERS = 16


class FitNet(C.Structure):
    """aadff_fit_net: the PSF network and its buffers for aadff_fit_chain (offsets in elements)."""
    _L, _L1 = C.c_int * FIT_MAX_LAYERS, C.c_int * (FIT_MAX_LAYERS + 1)
    _fields_ = [("n_layers", C.c_int), ("batch", C.c_int), ("ld_batch", C.c_int),
                ("k", _L), ("n", _L), ("ld_k", _L), ("ld_n", _L), ("off_w", _L), ("off_wt", _L), ("off_b", _L),
                ("off_xt", _L1), ("off_dzt", _L1), ("off_gw", _L), ("off_gb", _L),
                ("param_bf16", C.c_void_p), ("scratch_bf16", C.c_void_p), ("grad", C.c_void_p),
                ("inp", C.c_void_p), ("target", C.c_void_p), ("pred", C.c_void_p)]


class Levels(C.Structure):
    """aadff_levels_t: the short levels of a strict / edge stack (csrc/stack_host.cpp)."""
    _fields_ = [("S", C.c_int), ("n_surf", C.c_int), ("n_tables", C.c_int), ("jobs_max", C.c_int), ("fov_rays", C.c_int), ("J1", C.c_int), ("J2", C.c_int),
                ("pad", C.c_int), ("tables_dev", C.c_void_p), ("bt_green", C.c_void_p), ("zeros", C.c_void_p),
                ("h_par1", C.c_void_p), ("d_par1", C.c_void_p), ("h_res1", C.c_void_p), ("d_res1", C.c_void_p),
                ("h_par2", C.c_void_p), ("d_par2", C.c_void_p), ("h_res2", C.c_void_p), ("d_res2", C.c_void_p),
                ("h_pupil", C.c_void_p), ("d_pupil", C.c_void_p), ("curved", C.c_ubyte * MAX_SURF)]


class EdgeStack(C.Structure):
    """aadff_edge_stack_t: the edge-exact psf_map level of a stack."""
    _fields_ = [("S", C.c_int), ("L", C.c_int), ("N", C.c_int), ("spp", C.c_int), ("ks", C.c_int), ("n_surf", C.c_int), ("n_tables", C.c_int),
                ("t_green", C.c_int), ("cap", C.c_int), ("pad", C.c_int), ("per", C.c_long), ("per_l", C.c_long), ("o_main", C.c_long), ("n_pm", C.c_long),
                ("delta", C.c_float), ("pixel_size", C.c_float), ("lc", LensConst), ("tables_dev", C.c_void_p),
                ("h_u", C.c_void_p), ("d_u", C.c_void_p), ("h_focus", C.c_void_p), ("d_focus", C.c_void_p), ("d_pts", C.c_void_p), ("states_prov", C.c_void_p),
                ("raw", C.c_void_p), ("slope", C.c_void_p), ("count", C.c_void_p), ("list", C.c_void_p), ("h_back", C.c_void_p),
                ("h_par3", C.c_void_p), ("d_par3", C.c_void_p), ("pset", C.c_void_p), ("bt_main", C.c_void_p),
                ("h_pupil_main", C.c_void_p), ("d_pupil_main", C.c_void_p)]


assert C.sizeof(Stage) == 40
assert C.sizeof(Surface) == 136 and C.sizeof(LensState) == 32 and C.sizeof(LensConst) == 52

_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_long

# name -> argtypes; every function returns int (0 = ok).  Kept in one table so the
# symbol-export test can walk it (tests/test_abi_symbols.py).
PROTOTYPES = {
    "aadff_render_psf_map": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "aadff_render_psf_map_stack": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "aadff_render_psf_map_stack_strided": [_P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _P],
    "aadff_render_psf_map_stack_layered": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "aadff_time_next_launch": [_P, _P],
    "aadff_render_psf": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "aadff_local_psf_render": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "aadff_thinlens_render": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _F, _F, _P],
    "aadff_trace_rays": [_P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P, _P],
    "aadff_trace_rays_strict": [_P, _P, _P, _I, _P, _I, _I, _I, _I, _F, _P, _P, _P],
    "aadff_trace_rays_strict_batched": [_P, _P, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "aadff_trace_rays_strict_fused": [_P, _P, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "aadff_strict_psf_points": [_P, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P, _F, _I, _I, _P, _P, _P, _P, _P],
    "aadff_strict_psf_points_alt": [_P, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "aadff_selftest_strict_ops": [_P, _P, _I, _I, _P, _P],
    "aadff_strict_centroid": [_P, _P, _I, _I, _I, _P, _P, _P],
    "aadff_trace_points": [_P, _I, _P, _P, _I, _F, _F, _P, _I, _P, _P, _P, _P, _P],
    "aadff_psf_splat": [_P, _P, _P, _I, _I, _F, _I, _P, _P, _P],
    "aadff_psf_points": [_P, _I, _I, _I, _P, _P, LensConst, _P, _P, _I, _L, _L, _P, _I, _L, _L, _I, _I, _I, _P, _P, _P, _P],
    "aadff_psf_points_staged": [_P, _I, _I, _I, _P, _P, LensConst, _P, _P, _I, _L, _L, _P, _I, _L, _L, _I, _I, _I, _P, _P, _P,
                                C.POINTER(Stage), _P],
    "aadff_psf_points_edge": [_P, _I, _I, _I, _P, _P, LensConst, _P, _P, _I, _L, _L, _P, _I, _L, _L, _I, _F, _P, _P, _P, _P, _P, _I, _P, _P],
    "aadff_strict_edge_retrace": [_P, _I, _I, _P, _P, _I, _I, _P, _P, _P, _I, _P, _F, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P],
    "aadff_psf_normalise": [_P, _I, _I, _I, _F, _I, _I, _P, _P],
    "aadff_psfnet_forward": [_P, _L, _P, _P, _I, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "aadff_psfnet_render_rgbd": [_P, _P, _P, _P, _F, _F, _L, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "aadff_refocus": [_P, _I, _P, _I, _L, _P, LensConst, _P, _P],
    "aadff_refocus_staged": [_P, _I, _P, _P, _L, _I, _L, _P, LensConst, _P, _P, _P],
    "aadff_post_computation": [_I, _P, LensConst, _P, _P],
    "aadff_publish_flags": [_P, _P, _P],
    "aadff_relu_bwd_bias": [_P, _P, _P, _P, _I, _I, _I, _P],
    "aadff_psfnet_head_loss_grad": [_P, _P, _P, _P, _I, _I, _I, _P],
    "aadff_fit_gemm_nt": [_P, _I, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P],
    "aadff_fit_chain": [_P, _P, _P, _F, _I, _F, _F, _F, _P],
    "aadff_fit_layer_bwd": [_P, _I, _I, _P, _I, _I, _I, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P],
    "aadff_fit_input": [_P, _P, _I, _P, _I, _I, _I, _P],
    "aadff_fit_head": [_P, _I, _P, _P, _P, _I, _P, _I, _P, _I, _I, _P, _P, _F, _I, _F, _F, _F, _P],
    "aadff_fit_adamw": [_P, _P, _P, _P, _P, _P, _P, _L, _P, _F, _F, _F, _P],
    "aadff_adamw_step": [_P, _P, _I, _P, _P, _P, _L, _P, _P, _F, _I, _F, _F, _F, _F, _P],
    "aadff_host_mt19937_uniform_f32": [_P, _L, _L, _P],
    "aadff_host_mt19937_discard": [_P, _L, _L],
    "aadff_host_mt19937_rows": [_P, _L, _I, _L, _L, _P, _P, _I],
    "aadff_strict_replay_threads": [_I],
    "aadff_host_device_pointer": [_P, _P],
    "aadff_host_masked_mean_f32": [_P, _P, _L, _L, _P, _P],
    "aadff_levels_focus_submit": [C.POINTER(Levels), _P, _P, _F, _F, _F, _P, _P, _P, _P, _I, _P, _P],
    "aadff_levels_focus_finish": [C.POINTER(Levels), _P, _P, _P],
    "aadff_levels_fov_submit": [C.POINTER(Levels), _P, _F, _I, _P, _P],
    "aadff_levels_fov_finish": [C.POINTER(Levels), _P, _P, _P],
    "aadff_edge_provisional": [C.POINTER(EdgeStack), _P, _P, _P],
    "aadff_edge_finish": [C.POINTER(EdgeStack), _P, _P, _P, _F, _F, _F, _P, _P, _P, _P, _P],
    "aadff_host_pupil_points": [_P, _L, _P, _P, _L, _F, _F, _F, _P, _P, _P, _P, _I],
}
OTHER_SYMBOLS = ["aadff_abi_version", "aadff_last_error", "aadff_device_info"]

_lib = None


def load_library(path=None):
    """dlopen libaadff.so and set prototypes; loud failure if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"aadff: HIP library not found at {p}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C aberration-aware-depth-from-focus_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.aadff_abi_version.restype = C.c_int
    lib.aadff_last_error.restype = C.c_char_p
    lib.aadff_device_info.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
    lib.aadff_device_info.restype = C.c_int
    if lib.aadff_abi_version() != ABI_VERSION:
        raise RuntimeError(f"aadff: ABI version mismatch: library {lib.aadff_abi_version()} != binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


_gpu_ok = False


def require_gpu():
    global _gpu_ok
    if _gpu_ok:                                       # (torch.cuda.is_available() re-reads the environment on every call: 2.5 us)
        return _lib if _lib is not None else load_library()
    if not torch.cuda.is_available():
        raise RuntimeError("aadff: no HIP device visible. This package runs its hot path on MI355X only; "
                           "there is no CPU fallback (the CPU restatement under oracle/ is test infrastructure).")
    lib = load_library()
    _gpu_ok = True
    return lib


def call(name, *args):
    """Invoke an ABI function; raise with the library's message on failure."""
    lib = require_gpu()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.aadff_last_error().decode(errors="replace")
        if rc == -1 and ("should be" in msg or "Input image" in msg):
            raise AssertionError(msg)     # the reference raises AssertionError for these (render_psf.py:43-53)
        raise RuntimeError(f"{name} failed (rc={rc}): {msg}")


class on_device:
    """`with torch.cuda.device(dev)` that does nothing when dev already is the current device (the context manager costs 5-8 us per
    entry, three entries per slice of the per-call API)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            return self.ctx.__exit__(*a)


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "aadff ABI needs contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def f32c(t, device):
    """`.contiguous().float()` on the target device (SURVEY.md §8b ownership row)."""
    return t.to(device=device, dtype=torch.float32).contiguous()
