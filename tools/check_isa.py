#!/usr/bin/env python3
"""Build guard: no gfx950 code object of the library may contain a two-operand packed float32 instruction whose LOW lane reads the HIGH
half of its SECOND source when that source is a VGPR pair (`v_pk_mul_f32 / v_pk_add_f32 d, a, v[n:n+1] op_sel:[x,1]`, with or without
op_sel_hi).

Why: on MI355X that form returns wrong low-lane results now and then while a wave of ANOTHER kernel issues MFMA instructions on the same
SIMD (our own convolution on a second stream is enough); alone on the chip it is exact.  Measured with tools/concurrency_probe.py
(profiles/r06_concurrency_probe_grid.txt): v_pk_mul_f32 / v_pk_add_f32 with op_sel:[0,1] (also the swapped form op_sel:[0,1]
op_sel_hi:[1,0]) differ in every launch beside a bare MFMA chain; op_sel on the first source, op_sel:[1,1], every op_sel_hi-only form,
every v_pk_fma_f32 form and SGPR sources (also an SGPR pair as the second source with op_sel:[0,1], the compiler's broadcast of an odd
scalar register) do not.  The compiler only produces the form when its SLP vectoriser re-packs scalar code that
follows explicit float2 arithmetic (the aspheric polynomial of csrc/strict_math.h after csrc/strict_math2.h's conic part), so the
strict translation units are built with -fno-slp-vectorize; this check keeps every object honest.

    python tools/check_isa.py obj.o [obj.o ...]      # exit 1 and list the offenders if any"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("AADFF_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
BAD = re.compile(r"\bv_pk_(mul|add|min|max)_f32\s+[^,]+,\s*[^,]+,\s*v\[\d+:\d+\]\s+op_sel:\[[01],1\]")      # second source a VGPR pair


def offenders(path):
    tmp = tempfile.mkdtemp(prefix="aadff_isa_")
    try:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", os.path.basename(local)], cwd=tmp, check=True, capture_output=True)
        cos = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not cos:
            return None
        out = []
        for co in cos:
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], cwd=tmp, check=True, capture_output=True, text=True).stdout
            fn = "?"
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    fn = m.group(1)
                elif BAD.search(line):
                    out.append((fn, line.split("//")[0].strip()))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main(paths):
    rc = 0
    for p in paths:
        found = offenders(p)
        if found is None:
            print(f"check_isa: {p}: no device code object (host-only object)")
            continue
        if found:
            rc = 1
            print(f"check_isa: {p}: {len(found)} packed float32 instruction(s) whose low lane reads the high half of the second source:")
            for fn, ins in found[:20]:
                print(f"    {fn}: {ins}")
        else:
            print(f"check_isa: {p}: clean")
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
