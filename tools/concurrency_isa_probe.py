#!/usr/bin/env python3
"""Which instruction forms of the strict (packed float32) trace are reliable on MI355X while ANOTHER kernel runs on a second stream?

Victims: tools/conc_victims.hip - one instruction class, one csrc/strict_math2.h function or one exact inline-asm sequence per kernel,
in a long dependent chain on seeded inputs; two launches on the same inputs must agree bit for bit.  Each victim is launched beside a
load on a side stream and compared with a quiet launch.  Loads: none | conv = aadff_render_psf_map_stack (the MFMA convolution) |
agg0 = a bare v_mfma_f32_16x16x32_f16 chain | agg1 = 26 KB of LDS in use | agg2 = plain VALU | agg3 = MFMA holding 26 KB of LDS.

Result (profiles/r06_concurrency_probe_grid.txt): v_pk_mul_f32 / v_pk_add_f32 with op_sel on the SECOND source (low lane reads the high
half of src1: op_sel:[0,1], also with op_sel_hi:[1,0]) differ in every launch beside an MFMA load and never alone; op_sel on the
first source, op_sel:[1,1], op_sel_hi-only forms, every v_pk_fma_f32 form, SGPR sources, v_pk_mov_b32, trans ops, float64, LDS and
scalar / coherent loads never differ.  tools/check_isa.py refuses that form at build time.

    python tools/concurrency_isa_probe.py [launches per cell]     # OPS=0,1,... LOADS=none,conv,agg0,... select; builds the victims if needed"""
import os, sys, ctypes as C, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff import _abi
dev = torch.device("cuda:0")
SO = os.path.join(REPO, "tools", "conc_victims.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SO[:-3] + ".hip"):
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-I" + os.path.join(REPO, "include"),
                    "-I" + os.path.join(REPO, "aberration-aware-depth-from-focus_amd", "csrc"), SO[:-3] + ".hip", "-o", SO], check=True)
lib = C.CDLL(SO)
lib.victim_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.aggressor_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
torch.manual_seed(0)
n = 1 << 20
inp = torch.rand(n, device=dev) + 0.5
table = torch.rand(1024, device=dev)
from deeplens.optics import Lensgroup
from deeplens.basics import WAVE_RGB
lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(512, 512), device="cuda:0", parity="strict")
surf_table = lens._table(list(WAVE_RGB))
BL = 4096
img_big = torch.rand(1, 3, 1024, 1024, device=dev)
psf = torch.rand(10, 3, 121, 121, device=dev); psf /= psf.sum()
out_big = torch.empty(1, 3, 10, 1024, 1024, device=dev)
agg_out = torch.empty(8192 * 256, device=dev)
side = torch.cuda.Stream()
NAMES = ["pk_fma", "pk_mul_add", "fma", "rcp", "sqrt", "ieee_div", "div_fixup", "lds", "s_load", "fresh_load", "f64_pow", "asph_terms", "div2", "sqrt2", "normalize32", "rcp_pair_pk", "rcp_pair_nop_pk", "sag_poly", "dsag_poly", "dsag", "sag_dsag2", "residual2", "sag_dsag2_nopoly", "asm_cvt_pkmov_mulf64_pkmul", "asm_pkmov_nop_pkmul_then_f64", "asm_pkmov_mulf64_nop0_pkmul", "asm_pkmov_mulf64_nop4_pkmul", "asm_pk_first_then_f64", "asm_pkmov_pkmul_nof64", "asm_pkmov_sgpr_only", "asm_pkmul_opsel01_only", "asm_pkmov_vgpr_only", "asm_pkmul_opselhi10_only", "pkmul_opsel10", "pkmul_sgpr_opsel10", "pkfma_opsel100", "pkfma_sgpr_opsel100", "pkfma_opsel001", "pkadd_opsel01", "pkadd_swap", "pkmul_swap", "pkmul_opsel11", "pkmul_opselhi00", "pkmul_sgpr_src1_opsel01"]
ITERS = [4000, 3000, 4000, 1500, 1500, 300, 1500, 1500, 1500, 600, 400, 200, 400, 400, 150, 1500, 1500, 150, 150, 100, 100, 100, 100, 1500, 1500, 1500, 1500, 1500, 2000, 2000, 2000, 2000, 2000, 2000, 2000, 2000, 2000, 2000, 2000, 2000, 2000, 2000, 2000, 2000]
NIT = int(sys.argv[1]) if len(sys.argv) > 1 else 200
def victim(op):
    out = torch.empty(2 * BL * 256, device=dev)
    rc = lib.victim_launch(op, inp.data_ptr(), out.data_ptr(), n, BL, ITERS[op], (surf_table if op >= 17 else table).data_ptr(), _abi.stream_ptr(dev))
    assert rc == 0, rc
    return out
def load(kind):
    with torch.cuda.stream(side):
        s = C.c_void_p(side.cuda_stream)
        if kind == "conv":
            for _ in range(6): _abi.call("aadff_render_psf_map_stack", _abi.ptr(img_big), _abi.ptr(psf), _abi.ptr(out_big), 1, 3, 10, 1024, 1024, 11, 11, s)
        elif kind.startswith("agg"):
            k = int(kind[3:])
            for _ in range(4): lib.aggressor_launch(k, agg_out.data_ptr(), 8192, [5000, 1500, 10000, 5000][k], s)
for kind in os.environ.get("LOADS", "none,conv,agg0,agg1,agg2,agg3").split(","):
    for op in [int(x) for x in os.environ.get('OPS', ','.join(str(i) for i in range(len(NAMES)))).split(',')]:
        ref = victim(op); torch.cuda.synchronize()
        bad = 0; worst = 0
        for it in range(NIT):
            load(kind)
            got = victim(op)
            torch.cuda.synchronize()
            nd = int((got.view(torch.int32) != ref.view(torch.int32)).sum())
            bad += nd > 0; worst = max(worst, nd)
        print(f"load {kind:5s} victim {NAMES[op]:10s}: {bad} of {NIT} launches differ (most differing words in one launch {worst} of {ref.numel()})", flush=True)
