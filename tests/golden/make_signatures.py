#!/usr/bin/env python3
"""Fixture G16: the call signatures of the reference's public functions and methods, taken from its source by `ast`
(no import: torchvision / cv2 are not installed here).  Run in the build container (needs /root/reference):

    python tests/golden/make_signatures.py      ->  tests/golden/g16_signatures.json

{module: {qualified name: [[parameter, default source text or null], ...]}} - names and defaults only, no bodies;
tests/test_host_logic.py compares every entry this package also defines (drop-in check of SURVEY.md 8b)."""
import ast
import json
import os

REF = "/root/reference"
MODULES = ["deeplens/optics.py", "deeplens/psfnet.py", "deeplens/psfnet_arch.py", "deeplens/render_psf.py", "deeplens/basics.py",
           "deeplens/surfaces.py", "deeplens/monte_carlo.py", "dff/utils.py", "dff/dataset.py", "dff/factory.py"]


def sig(fn):
    a = fn.args
    pos = a.posonlyargs + a.args
    defaults = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
    out = [[p.arg, d] for p, d in zip(pos, defaults)]
    if a.vararg:
        out.append(["*" + a.vararg.arg, None])
    out += [[k.arg, None if d is None else ast.unparse(d)] for k, d in zip(a.kwonlyargs, a.kw_defaults)]
    if a.kwarg:
        out.append(["**" + a.kwarg.arg, None])
    return out


def main():
    res = {}
    for rel in MODULES:
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        entries = {}
        for node in tree.body:
            if isinstance(node, ast.FunctionDef):
                entries[node.name] = sig(node)
            elif isinstance(node, ast.ClassDef):
                for sub in node.body:
                    if isinstance(sub, ast.FunctionDef):
                        entries[f"{node.name}.{sub.name}"] = sig(sub)
        res[rel[:-3].replace("/", ".")] = entries
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g16_signatures.json")
    json.dump(res, open(out, "w"), indent=0, sort_keys=True)
    print(out, {m: len(v) for m, v in res.items()})


if __name__ == "__main__":
    main()
