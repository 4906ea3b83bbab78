"""CPU: libaadff.so loads and exports every symbol include/aadff.h declares; the ctypes
structs match the C layout.  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from aadff import _abi


@pytest.fixture(scope="module")
def header(repo_root):
    return open(os.path.join(repo_root, "include", "aadff.h")).read()


def test_library_built():
    assert os.path.exists(_abi.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported(header):
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(aadff_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 13
    lib = C.CDLL(_abi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in aadff.h but not exported"
    assert declared == set(_abi.PROTOTYPES) | set(_abi.OTHER_SYMBOLS), "ctypes binding and header drifted apart"


def test_abi_version_and_error_string():
    lib = _abi.load_library()
    assert lib.aadff_abi_version() == _abi.ABI_VERSION == 9
    assert isinstance(lib.aadff_last_error(), bytes)


def test_struct_layouts_match_c(tmp_path, repo_root):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "aadff.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(aadff_surface_t),sizeof(aadff_lens_state_t),sizeof(aadff_lens_const_t),"
                   "offsetof(aadff_surface_t,ai),offsetof(aadff_lens_state_t,flags),offsetof(aadff_lens_const_t,first_r2),"
                   "sizeof(aadff_fit_net),offsetof(aadff_fit_net,off_xt),offsetof(aadff_fit_net,param_bf16),"
                   "sizeof(aadff_levels_t),offsetof(aadff_levels_t,curved),sizeof(aadff_edge_stack_t),offsetof(aadff_edge_stack_t,lc),"
                   "offsetof(aadff_edge_stack_t,h_pupil_main));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(repo_root, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_abi.Surface), C.sizeof(_abi.LensState), C.sizeof(_abi.LensConst),
            _abi.Surface.ai.offset, _abi.LensState.flags.offset, _abi.LensConst.first_r2.offset,
            C.sizeof(_abi.FitNet), _abi.FitNet.off_xt.offset, _abi.FitNet.param_bf16.offset,
            C.sizeof(_abi.Levels), _abi.Levels.curved.offset, C.sizeof(_abi.EdgeStack), _abi.EdgeStack.lc.offset, _abi.EdgeStack.h_pupil_main.offset]
    assert got == want


def test_argument_errors_need_no_gpu():
    """Argument validation happens before any HIP call."""
    lib = _abi.load_library()
    rc = lib.aadff_render_psf_map(None, None, None, 1, 3, 8, 8, 2, 3, None)
    assert rc == -1 and b"NULL" in lib.aadff_last_error()
    rc = lib.aadff_render_psf_map(C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), 1, 3, 64, 64, 2, 4, None)
    assert rc == -1 and b"odd" in lib.aadff_last_error()
    rc = lib.aadff_render_psf_map(C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), 1, 3, 64, 64, 100, 3, None)
    assert rc == -1 and b"grid" in lib.aadff_last_error()
