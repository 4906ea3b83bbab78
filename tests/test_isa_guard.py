"""The library's gfx950 code objects hold no packed float32 instruction of the form that is unreliable beside another kernel's MFMA
waves on MI355X (tools/check_isa.py, profiles/r06_concurrency_probe_grid.txt): the build refuses such an object, this test keeps the
built libraries honest and pins what the pattern does and does not match."""
import importlib.util
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_isa", os.path.join(REPO, "tools", "check_isa.py"))
check_isa = importlib.util.module_from_spec(spec)
spec.loader.exec_module(check_isa)


def test_pattern_matches_the_measured_forms_only():
    bad = ["v_pk_mul_f32 v[66:67], v[66:67], v[42:43] op_sel:[0,1]",
           "v_pk_add_f32 v[12:13], v[12:13], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]",
           "v_pk_mul_f32 v[2:3], v[2:3], v[4:5] op_sel:[1,1]"]                      # measured clean, refused all the same: never emitted, one rule
    good = ["v_pk_mul_f32 v[6:7], v[8:9], s[30:31] op_sel_hi:[1,0]",
            "v_pk_mul_f32 v[28:29], v[28:29], s[66:67] op_sel:[0,1]",               # an odd scalar register broadcast: measured clean
            "v_pk_mul_f32 v[24:25], s[16:17], v[6:7] op_sel:[1,0]",
            "v_pk_fma_f32 v[0:1], s[2:3], v[4:5], v[0:1] op_sel:[1,0,0] op_sel_hi:[1,1,1]",
            "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]",
            "v_pk_mov_b32 v[24:25], v[6:7], s[16:17] op_sel:[1,0]",
            "v_pk_add_f32 v[6:7], v[6:7], 1.0 op_sel_hi:[1,0]"]
    for line in bad:
        assert check_isa.BAD.search(line), line
    for line in good:
        assert not check_isa.BAD.search(line), line


@pytest.mark.parametrize("lib", ["libaadff.so", "libaadff_latestage.so", "libaadff_literal.so"])
def test_built_libraries_are_clean(lib):
    path = os.path.join(REPO, "aberration-aware-depth-from-focus_amd", "csrc", lib)
    assert os.path.exists(path), f"{path} is not built (python -c 'import __graft_entry__ as g; g.build()')"
    found = check_isa.offenders(path)
    assert found is not None, "no gfx950 code object found in " + lib
    assert found == [], found[:5]
