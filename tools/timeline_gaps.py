#!/usr/bin/env python3
"""Idle gaps and overlaps between the kernels of consecutive M1 stacks from a rocprofv3 kernel trace (csv)."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    tag = "psf" if "psf_points" in n else "conv" if "conv_psf_map" in n else "refocus" if "refocus" in n else None
    if tag:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag))
rows.sort()
rows = rows[len(rows) // 2:]                                  # steady state
by = collections.defaultdict(list)
for s, e, t in rows:
    by[t].append((s, e))
dur = {t: sum(e - s for s, e in v) / len(v) / 1e3 for t, v in by.items()}
print("mean duration us:", {k: round(v, 1) for k, v in dur.items()})
psf, conv, ref = by["psf"], by["conv"], by["refocus"]
period = [(psf[i + 1][0] - psf[i][0]) / 1e3 for i in range(len(psf) - 1)]
period = [p for p in period if p < 2 * sorted(period)[len(period) // 2]]
print("period us (psf start to next psf start): median %.1f" % sorted(period)[len(period) // 2])
g1, g2 = [], []
for c in conv:
    prv = [p for p in psf if p[1] <= c[0] + 1000]
    nxt = [p for p in psf if p[0] >= c[1] - 1000]
    if prv: g1.append(c[0] - prv[-1][1])
    if nxt: g2.append(nxt[0][0] - c[1])
med = lambda v: sorted(v)[len(v) // 2] / 1e3
print("gap psf end -> conv start: median %.2f us; conv end -> next psf start: median %.2f us" % (med(g1), med(g2)))
ov = []
for s, e in ref:
    inside = [c for c in conv if c[0] < e and c[1] > s] + [p for p in psf if p[0] < e and p[1] > s]
    ov.append(1 if inside else 0)
print("refocus launches overlapping another kernel: %d of %d" % (sum(ov), len(ov)))
oc = sum(1 for s, e in ref if any(c[0] < e and c[1] > s for c in conv))
print("refocus launches overlapping a convolution: %d of %d" % (oc, len(ref)))
solo = [c[1] - c[0] for c in conv if not any(r[0] < c[1] and r[1] > c[0] for r in ref)]
both = [c[1] - c[0] for c in conv if any(r[0] < c[1] and r[1] > c[0] for r in ref)]
print("conv duration us: alone %.1f (n=%d), beside a refocus %.1f (n=%d)" % (sum(solo) / max(1, len(solo)) / 1e3, len(solo), sum(both) / max(1, len(both)) / 1e3, len(both)))
for s, e in ref[:4]:
    prev_psf = [p for p in psf if p[0] <= s]
    if prev_psf:
        print("  refocus start %.1f us after psf start (psf lasts %.1f), refocus lasts %.1f" % ((s - prev_psf[-1][0]) / 1e3, (prev_psf[-1][1] - prev_psf[-1][0]) / 1e3, (e - s) / 1e3))
