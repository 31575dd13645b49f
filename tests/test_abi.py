"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, and exports
every symbol include/vda.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from video_depth_anything_amd import build
    return build.build()


def declared_symbols():
    text = open(os.path.join(REPO, "include", "vda.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vda_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for need in ("vda_gemm_f16", "vda_attention_f16", "vda_layernorm_f32_f16", "vda_groupnorm_nhwc_f16",
                 "vda_temporal_attention_f16", "vda_bilinear_nhwc_f16", "vda_last_error",
                 # fp32-operand twins
                 "vda_gemm_f32", "vda_attention_f32", "vda_layernorm_f32_f32", "vda_groupnorm_nhwc_f32", "vda_temporal_attention_f32",
                 # handle API
                 "vda_create", "vda_destroy", "vda_load_weight", "vda_finalize_weights", "vda_workspace_bytes", "vda_forward"):
        assert need in syms


def test_library_exports_every_declared_symbol(libpath):
    import torch  # noqa: F401  (its bundled libamdhip64.so must be the process's one HIP runtime, see _lib._load)
    ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(libpath)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"libvda_hip.so lacks {missing}"
    assert lib.vda_abi_version() == 8


def test_binding_table_matches_header(libpath):
    from video_depth_anything_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert ctypes.sizeof(_lib.GemmArgs) == 12 * 8 + 22 * 4
    assert ctypes.sizeof(_lib.Config) == 16 * 4              # vda_config: what examples/host_demo.cpp reads from model.bin


def test_refusal_needs_no_gpu(libpath):
    """Argument validation happens before any HIP call."""
    from video_depth_anything_amd import _lib
    a = _lib.GemmArgs()
    assert _lib.lib.vda_gemm_f16(ctypes.byref(a), None) != 0
    assert b"null" in _lib.lib.vda_last_error()


def test_cpp_host_demo_builds_and_links(libpath):
    """examples/host_demo.cpp (a C++ host of the handle API) compiles against include/vda.h and links against the library;
    argument checking happens before any HIP call."""
    import subprocess
    from video_depth_anything_amd import build
    exe = build.build_host_demo()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "usage" in r.stderr


def test_new_entry_points_refuse_bad_arguments_without_a_gpu(libpath):
    """vda_conv3x3_up2_f16 / vda_rope_qk_f16 validate before any HIP call (null pointers, widths the kernels are not built for)."""
    from video_depth_anything_amd import _lib
    lib = _lib.lib
    assert lib.vda_conv3x3_up2_f16(None, None, None, None, 1, 4, 4, 64, 32, 32, None) != 0
    assert b"null" in lib.vda_last_error()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.cast(ctypes.addressof(buf) + (-ctypes.addressof(buf)) % 16, ctypes.c_void_p)
    assert lib.vda_conv3x3_up2_f16(p, p, None, p, 1, 4, 4, 24, 32, 32, None) != 0          # C not a multiple of 16
    assert b"multiple of 16" in lib.vda_last_error()
    assert lib.vda_conv3x3_up2_f16(p, p, None, p, 1, 4, 4, 64, 256, 256, None) != 0        # N > 128: the implicit GEMM's job
    assert b"at most 128" in lib.vda_last_error()
    assert lib.vda_rope_qk_f16(p, 4, 4, 12, None) != 0                                      # C not a multiple of 8
    assert b"multiple of 8" in lib.vda_last_error()
