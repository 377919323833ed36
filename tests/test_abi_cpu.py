"""CPU tests of the drop-in boundary: libvoge_hip.so loads without a GPU, exports every symbol
include/voge_hip.h declares, and the ctypes table mirrors the header.  No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "voge_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t|const char \*)\s*(voge_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(2)] = n
    return out


@pytest.fixture(scope="module")
def lib():
    from voge_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_header_declares_the_path():
    fns = header_functions()
    for name in ("voge_trace_topk_fwd", "voge_trace_topk_list_fwd", "voge_trace_bwd", "voge_composite_fwd",
                 "voge_composite_bwd", "voge_merge_fwd", "voge_merge_bwd", "voge_blend_fwd", "voge_blend_bwd",
                 "voge_shade_fwd", "voge_shade_bwd", "voge_rays_fwd", "voge_rays_bwd", "voge_abi_version",
                 "voge_error_string", "voge_trace_workspace_bytes", "voge_trace_bwd_workspace_bytes"):
        assert name in fns, name
    # every entry point cites the reference interface it replaces
    text = open(HEADER).read()
    assert text.count("Replaces") >= 7 and "ray_trace_voge.cu" in text and "Aggregation.py" in text


def test_library_exports_every_declared_symbol(lib):
    from voge_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    fns = header_functions()
    for name, nargs in fns.items():
        assert hasattr(raw, name), f"{name} declared in voge_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} missing from the ctypes table"
        assert len(_lib.SIGNATURES[name][1]) == nargs, f"{name}: header has {nargs} args"
    assert set(_lib.SIGNATURES) == set(fns)


def test_version_errors_and_sizes_without_gpu(lib):
    assert lib.voge_abi_version() == 7
    assert lib.voge_error_string(0) == b"success"
    assert b"K exceeds" in lib.voge_error_string(-3)
    assert b"workspace" in lib.voge_error_string(-2)
    n = lib.voge_trace_workspace_bytes(1, 50000, 512, 512)
    assert n >= 50000 * 64 and n % 256 == 0
    assert lib.voge_trace_workspace_bytes(2, 1000, 64, 64) > lib.voge_trace_workspace_bytes(1, 1000, 64, 64)
    assert lib.voge_trace_bwd_workspace_bytes(1000) == 1000 * 112
    # argument validation happens before any HIP call
    assert lib.voge_composite_fwd(None, None, None, None, None, 1.0, 10, 0, None, None, None) == -1
    assert lib.voge_trace_topk_fwd(None, None, None, None, None, 1, 10, 8, 8, 1000, 4.6, None, 0, None, None, None, None, None, None) == -3
    # the cone hierarchy: 21 records of 8 floats per 32x32-pixel super-tile (its own, 4 quads', 16 tiles')
    assert lib.voge_cones_floats(2, 65, 33) == 2 * 3 * 2 * 21 * 8
    # the trace's scratch is sized for a CHUNK of the batch: never below one view, capped at 1 GiB beyond that
    # (eight 1024^2 views of 200k Gaussians asked for 6.2 GB)
    one = lib.voge_trace_workspace_bytes(1, 200000, 1024, 1024)
    assert 700e6 < one < 900e6 and lib.voge_trace_workspace_bytes(8, 200000, 1024, 1024) == one < 1.5e9
    v512 = lib.voge_trace_workspace_bytes(1, 50000, 512, 512)
    assert lib.voge_trace_workspace_bytes(8, 50000, 512, 512) <= (1 << 30) < 8 * v512
    assert lib.voge_trace_workspace_bytes(5, 2562, 128, 128) > 3 * lib.voge_trace_workspace_bytes(1, 2562, 128, 128)   # small batches: one chunk (the list pool's floor does not scale)


def test_frame_entries_validate_before_any_hip_call(lib):
    """ABI 7's frame entries refuse bad arguments on the host (no GPU in this container: anything that reached HIP would fail
    differently)."""
    P = 1234      # (a non-NULL pointer value: nothing is dereferenced before validation is through)
    tr = (P, P, 1, 1, P, P, P, P, 0, 64, 0, 0, 1, 100, 64, 64)
    assert lib.voge_frame_trace_fwd_iso(*tr, 1000, 4.6, P, 1 << 30, P, P, P, P, P, None, None) == -3      # K above VOGE_MAX_K
    assert lib.voge_frame_trace_fwd_iso(P, P, 1, 1, None, P, P, P, 0, 64, 0, 0, 1, 100, 64, 64, 16, 4.6, P, 1 << 30, P, P, P, P, P, None, None) == -1   # no R
    assert lib.voge_frame_trace_fwd_iso(*tr, 16, 4.6, P, 1000, P, P, P, P, P, None, None) == -2          # scratch too small
    assert lib.voge_frame_trace_fwd_iso(P, P, 1, 3, P, P, P, P, 0, 64, 0, 0, 1, 100, 64, 64, 16, 4.6, P, 1 << 30, P, P, P, P, P, None, None) == -1   # sigma rule 3
    assert lib.voge_frame_trace_fwd_gen(P, P, 1, 1, 3, P, P, P, P, 0, 64, 0, 0, 1, 100, 64, 64, 16, 4.6, P, 1 << 30, P, P, P, P, P, None, None) == -1   # kind 3
    assert lib.voge_frame_bwd_acc_bytes(1000) == 32000 and lib.voge_frame_bwd_gen_acc_bytes(1000) == 64000
    # the backward: K beyond a pixel's lanes in one wave, a scratch that cannot hold the accumulator, an unknown form
    bw = (P, P, 1, 1, P, P, P, P, P, P, P, P, P, -1.0, P, 3, 1, 1.0, 1, 100, 64, 64)
    assert lib.voge_frame_shade_bwd_iso(*bw, 200, 3, 100, P, 3200, P, P, P, None) == -3
    assert lib.voge_frame_shade_bwd_iso(*bw, 16, 3, 100, P, 100, P, P, P, None) == -2
    assert lib.voge_frame_bwd_gen(7, P, 1, 1, 1, P, P, P, P, P, None, P, None, P, P, P, -1.0, P, 3, 1, None, 1.0, 1, 100, 64, 64, 16, 3, 100, P,
                                  6400, 1, P, P, P, None) == -1
    assert lib.voge_frame_bwd_gen(0, P, 1, 1, 1, P, P, P, P, P, P, P, P, P, P, P, -1.0, P, 3, 1, None, 1.0, 1, 100, 64, 64, 16, 3, 100, P,
                                  6400, 1, P, P, P, None) == -1      # per-axis records keep no act / dsd
    # the composite that zeroes the accumulator: 16-byte granularity
    assert lib.voge_frame_shade_fwd_iso(P, P, P, P, P, 1.0, P, P, -1.0, 4096, 16, 3, 100, P, P, P, P, P, None, P + 4, 3200, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    from voge_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libvoge_hip.so")
    with pytest.raises(_lib.VogeHipError, match="no CPU fallback"):
        _lib.load()
