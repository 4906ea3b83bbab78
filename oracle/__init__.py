"""ORACLE — test infrastructure only.

CPU restatement (torch fp32 / numpy / plain C) of the reference's focal-stack rendering
hot path, pinned against golden vectors generated from the imported reference
(tests/golden/make_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import, link or execute anything in this directory; the product
package (aberration-aware-depth-from-focus_amd/) never does and fails loudly when its
HIP library is missing.
"""
