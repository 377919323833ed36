"""ctypes binding of libvoge_hip.so (the C ABI declared in include/voge_hip.h).

There is exactly one implementation of the hot path: the HIP library.  If it cannot be
loaded, or a tensor is not on a HIP device, the ops raise -- there is no CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VOGE_HIP_LIB points at another build of the same ABI (kernel tuning variants)
LIB_PATH = os.environ.get("VOGE_HIP_LIB") or os.path.join(_HERE, "libvoge_hip.so")
ABI_VERSION = 7
MAX_K = 256      # VOGE_MAX_K of include/voge_hip.h: the top-K lists of a tile live in LDS

_c_void_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_long = ctypes.c_long
_c_float = ctypes.c_float
_c_size_t = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/voge_hip.h one to one
SIGNATURES = {
    "voge_abi_version": (_c_int, []),
    "voge_error_string": (ctypes.c_char_p, [_c_int]),
    "voge_trace_workspace_bytes": (_c_size_t, [_c_int] * 4),
    "voge_trace_pool_usage": (_c_int, [_c_void_p, _c_size_t] + [_c_int] * 4 + [_c_void_p] * 2),
    "voge_trace_topk_fwd": (_c_int, [_c_void_p] * 5 + [_c_int] * 5 + [_c_float, _c_void_p, _c_size_t]
                            + [_c_void_p] * 6),
    "voge_trace_topk_list_fwd": (_c_int, [_c_void_p] * 4 + [_c_int] * 9 + [_c_float] + [_c_void_p] * 6),
    "voge_trace_bwd": (_c_int, [_c_void_p] * 8 + [_c_int, _c_long, _c_int, _c_int, _c_void_p, _c_size_t] + [_c_void_p] * 4),
    "voge_trace_bwd_workspace_bytes": (_c_size_t, [_c_int]),
    "voge_trace_topk_fwd_iso": (_c_int, [_c_void_p] * 5 + [_c_int] * 5 + [_c_float, _c_void_p, _c_size_t]
                                + [_c_void_p] * 6),
    "voge_trace_bwd_iso": (_c_int, [_c_void_p] * 8 + [_c_int, _c_long, _c_int, _c_int, _c_void_p, _c_size_t] + [_c_void_p] * 4),
    "voge_trace_bwd_iso_workspace_bytes": (_c_size_t, [_c_int]),
    "voge_trace_topk_fwd_iso_view": (_c_int, [_c_void_p] * 3 + [_c_int] * 2 + [_c_void_p] * 3 + [_c_int] * 5
                                     + [_c_float, _c_void_p, _c_size_t] + [_c_void_p] * 6),
    "voge_trace_bwd_iso_view": (_c_int, [_c_void_p] * 3 + [_c_int] * 2 + [_c_void_p] * 6
                                + [_c_int, _c_int, _c_long, _c_int, _c_int, _c_void_p, _c_size_t] + [_c_void_p] * 4),
    "voge_bin_gaussians": (_c_int, [_c_void_p] * 3 + [_c_int] * 4 + [_c_void_p] + [_c_int] * 2 + [_c_void_p] * 2),
    "voge_fragments_fwd": (_c_int, [_c_void_p] * 5 + [_c_int] * 5 + [_c_float, _c_float, _c_void_p, _c_size_t] + [_c_void_p] * 8),
    "voge_fragments_fwd_iso": (_c_int, [_c_void_p] * 5 + [_c_int] * 5 + [_c_float, _c_float, _c_void_p, _c_size_t] + [_c_void_p] * 9),
    "voge_fragments_fwd_iso_view": (_c_int, [_c_void_p] * 3 + [_c_int] * 2 + [_c_void_p] * 3 + [_c_int] * 5
                                    + [_c_float, _c_float, _c_void_p, _c_size_t] + [_c_void_p] * 9),
    "voge_frame_trace_fwd_iso": (_c_int, [_c_void_p] * 2 + [_c_int] * 2 + [_c_void_p] * 4 + [_c_int] * 9
                                 + [_c_float, _c_void_p, _c_size_t] + [_c_void_p] * 7),
    "voge_frame_shade_fwd_iso": (_c_int, [_c_void_p] * 5 + [_c_float, _c_void_p, _c_void_p, _c_float, _c_long, _c_int, _c_int, _c_long]
                                 + [_c_void_p] * 7 + [_c_size_t, _c_void_p]),
    "voge_frame_shade_bwd_iso": (_c_int, [_c_void_p] * 2 + [_c_int] * 2 + [_c_void_p] * 9 + [_c_float, _c_void_p, _c_long, _c_long, _c_float]
                                 + [_c_int, _c_int, _c_long, _c_int, _c_int, _c_int, _c_long, _c_void_p, _c_size_t] + [_c_void_p] * 4),
    "voge_frame_merge_bwd_iso": (_c_int, [_c_void_p] * 2 + [_c_int] * 2 + [_c_void_p] * 7 + [_c_long, _c_long, _c_void_p, _c_void_p, _c_float]
                                 + [_c_int, _c_int, _c_long, _c_int, _c_int, _c_int, _c_long, _c_void_p, _c_size_t] + [_c_void_p] * 4),
    "voge_frame_bwd_acc_bytes": (_c_size_t, [_c_int]),
    "voge_frame_bwd_gen_acc_bytes": (_c_size_t, [_c_int]),
    "voge_frame_trace_fwd_gen": (_c_int, [_c_void_p] * 2 + [_c_int] * 3 + [_c_void_p] * 4 + [_c_int] * 9
                                 + [_c_float, _c_void_p, _c_size_t] + [_c_void_p] * 7),
    "voge_frame_shade_fwd_rec": (_c_int, [_c_int] + [_c_void_p] * 5 + [_c_float, _c_void_p, _c_void_p, _c_float, _c_long, _c_int, _c_int, _c_long]
                                 + [_c_void_p] * 9 + [_c_size_t, _c_void_p]),
    "voge_frame_bwd_gen": (_c_int, [_c_int, _c_void_p] + [_c_int] * 3 + [_c_void_p] * 11 + [_c_float, _c_void_p, _c_long, _c_long, _c_void_p, _c_float]
                           + [_c_int, _c_int, _c_long, _c_int, _c_int, _c_int, _c_long, _c_void_p, _c_size_t, _c_int] + [_c_void_p] * 4),
    "voge_fragment_bwd_workspace_bytes": (_c_size_t, [_c_int]),
    "voge_fragment_act_dsd_iso": (_c_int, [_c_void_p] * 5 + [_c_long, _c_int, _c_int] + [_c_void_p] * 3),
    "voge_composite_shade_fwd_iso": (_c_int, [_c_void_p] * 5 + [_c_float, _c_void_p, _c_void_p, _c_float, _c_long, _c_int, _c_int, _c_long]
                                     + [_c_void_p] * 6),
    "voge_composite_shade_fwd_rec": (_c_int, [_c_void_p] * 5 + [_c_float, _c_void_p, _c_void_p, _c_float, _c_long, _c_int, _c_int, _c_long]
                                     + [_c_void_p] * 8),
    "voge_composite_fwd_rec": (_c_int, [_c_void_p] * 5 + [_c_float, _c_long, _c_int] + [_c_void_p] * 5),
    "voge_trace_lean_fwd": (_c_int, [_c_void_p] * 5 + [_c_int] * 5 + [_c_float, _c_void_p, _c_size_t] + [_c_void_p] * 5),
    "voge_fragment_merge_bwd": (_c_int, [_c_void_p] * 11 + [_c_long, _c_long, _c_void_p, _c_float, _c_int, _c_long, _c_int, _c_int, _c_int,
                                                             _c_long, _c_void_p, _c_size_t] + [_c_void_p] * 4),
    "voge_composite_fwd_iso": (_c_int, [_c_void_p] * 5 + [_c_float, _c_long, _c_int] + [_c_void_p] * 3),
    "voge_fragment_shade_bwd": (_c_int, [_c_void_p] * 13 + [_c_float, _c_void_p, _c_long, _c_long, _c_float]
                                + [_c_int, _c_long, _c_int, _c_int, _c_int, _c_long, _c_void_p, _c_size_t] + [_c_void_p] * 4),
    "voge_fragment_shade_bwd_iso": (_c_int, [_c_void_p] * 2 + [_c_int] * 2 + [_c_void_p] * 11 + [_c_float, _c_void_p, _c_long, _c_long, _c_float]
                                    + [_c_int, _c_int, _c_long, _c_int, _c_int, _c_int, _c_long, _c_void_p, _c_size_t] + [_c_void_p] * 4),
    "voge_fragment_merge_bwd_iso": (_c_int, [_c_void_p] * 2 + [_c_int] * 2 + [_c_void_p] * 9 + [_c_long, _c_long, _c_void_p, _c_float]
                                    + [_c_int, _c_int, _c_long, _c_int, _c_int, _c_int, _c_long, _c_void_p, _c_size_t] + [_c_void_p] * 4),
    "voge_fragment_bwd_iso": (_c_int, [_c_void_p] * 2 + [_c_int] * 2 + [_c_void_p] * 8 + [_c_long, _c_long, _c_void_p, _c_float]
                              + [_c_int, _c_int, _c_long, _c_int, _c_int, _c_void_p, _c_size_t] + [_c_void_p] * 3),
    "voge_fragment_bwd": (_c_int, [_c_void_p] * 10 + [_c_long, _c_long, _c_void_p, _c_float]
                          + [_c_int, _c_long, _c_int, _c_int, _c_void_p, _c_size_t] + [_c_void_p] * 3),
    "voge_composite_fwd": (_c_int, [_c_void_p] * 5 + [_c_float, _c_long, _c_int] + [_c_void_p] * 3),
    "voge_composite_bwd": (_c_int, [_c_void_p] * 6 + [_c_float, _c_long, _c_int] + [_c_void_p] * 4),
    "voge_merge_fwd": (_c_int, [_c_void_p] * 4 + [_c_long, _c_int, _c_int, _c_long, _c_int] + [_c_void_p] * 2),
    "voge_merge_bwd": (_c_int, [_c_void_p] * 5 + [_c_long, _c_int, _c_int, _c_int, _c_long] + [_c_void_p] * 3),
    "voge_blend_fwd": (_c_int, [_c_void_p] * 3 + [_c_float, _c_long, _c_int, _c_int] + [_c_void_p] * 3),
    "voge_shade_fwd": (_c_int, [_c_void_p] * 5 + [_c_float, _c_long, _c_int, _c_int, _c_long, _c_int] + [_c_void_p] * 5),
    "voge_shade_bwd": (_c_int, [_c_void_p] * 7 + [_c_float, _c_void_p, _c_long, _c_int, _c_int, _c_int, _c_long] + [_c_void_p] * 3),
    "voge_rays_fwd": (_c_int, [_c_void_p] * 4 + [_c_int] * 4 + [_c_void_p] * 4),
    "voge_rays_striped_fwd": (_c_int, [_c_void_p] * 4 + [_c_int] * 6 + [_c_void_p] * 4),
    "voge_rays_striped_bwd": (_c_int, [_c_void_p] * 6 + [_c_int] * 6 + [_c_void_p] * 6),
    "voge_cones_floats": (_c_size_t, [_c_int] * 3),
    "voge_ray_cones": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "voge_general_preamble_fwd": (_c_int, [_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p] * 3),
    "voge_general_preamble_bwd": (_c_int, [_c_void_p] * 2 + [_c_int] * 5 + [_c_void_p] * 3),
    "voge_rays_bwd": (_c_int, [_c_void_p] * 6 + [_c_int] * 4 + [_c_void_p] * 6),
    "voge_ray_dense_fwd": (_c_int, [_c_void_p] * 3 + [_c_int, _c_long] + [_c_void_p] * 4),
    "voge_ray_dense_bwd": (_c_int, [_c_void_p] * 6 + [_c_int, _c_long] + [_c_void_p] * 4),
    "voge_find_nearest_k": (_c_int, [_c_void_p] * 3 + [_c_float, _c_int, _c_int, _c_long] + [_c_void_p] * 5),
    "voge_find_nearest_k_bwd": (_c_int, [_c_void_p] * 4 + [_c_int, _c_int, _c_long] + [_c_void_p] * 4),
    "voge_scatter_max": (_c_int, [_c_void_p] * 2 + [_c_long, _c_long] + [_c_void_p] * 2),
    "voge_silhouette_fwd": (_c_int, [_c_void_p, _c_long, _c_int] + [_c_void_p] * 3),
    "voge_silhouette_bwd": (_c_int, [_c_void_p] * 2 + [_c_long] + [_c_void_p] * 2),
    "voge_blend_bwd": (_c_int, [_c_void_p] * 3 + [_c_float, _c_void_p, _c_long, _c_int, _c_int] + [_c_void_p] * 3),
}

_lib = None


class VogeHipError(RuntimeError):
    """Raised when libvoge_hip.so is missing or one of its entry points returns non-zero
    (the reference raises RuntimeError through AT_CUDA_CHECK, ray_trace_voge.cu:278)."""


def load():
    """dlopen libvoge_hip.so and type its entry points.  Raises VogeHipError if the library is
    absent -- build it with `python -c 'import __graft_entry__ as g; g.build()'` or
    `make -C voge_amd/csrc`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VogeHipError(
            f"{LIB_PATH} not found: the VoGE hot path has no CPU fallback. "
            "Build the HIP library with `make -C voge_amd/csrc` (hipcc, gfx950).")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise VogeHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VogeHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.voge_abi_version()
    if got != ABI_VERSION:
        raise VogeHipError(f"libvoge_hip.so ABI {got} != expected {ABI_VERSION}")
    _lib = lib
    return lib


AB_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvoge_hip_ab.so")


class using:
    """`with _lib.using(path) as lib:` -- every entry point resolves to another build of the library inside the block (the
    -DVOGE_AB build with round 3's sweep and its switch, for the bit-for-bit test; tools' timing builds).  The product
    library is untouched: it is a second dlopen with its own globals."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global _lib
        if not os.path.exists(self.path):
            raise VogeHipError(f"{self.path} not found (make -C voge_amd/csrc builds it)")
        lib = ctypes.CDLL(self.path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        self.prev, _lib = _lib, lib
        return lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def check(code, what):
    if code != 0:
        msg = load().voge_error_string(code).decode()
        raise VogeHipError(f"{what} failed with code {code}: {msg}")
