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
n = min(len(psf), len(conv)) - 1
g1 = [conv[i][0] - psf[i][1] for i in range(n) if conv[i][0] > psf[i][0]]
period = [(psf[i + 1][0] - psf[i][0]) / 1e3 for i in range(n)]
print("period us (psf start to next psf start): mean %.1f" % (sum(period) / len(period)))
print("gap psf end -> conv start: mean %.2f us" % (sum(g1) / len(g1) / 1e3))
g2 = []
for i in range(n):
    nxt = [p for p in psf if p[0] >= conv[i][0]]
    if nxt: g2.append(nxt[0][0] - conv[i][1])
print("gap conv end -> next psf start: mean %.2f us" % (sum(g2) / len(g2) / 1e3))
ov = []
for s, e in ref:
    inside = [c for c in conv if c[0] < e and c[1] > s] + [p for p in psf if p[0] < e and p[1] > s]
    ov.append(1 if inside else 0)
print("refocus launches overlapping another kernel: %d of %d" % (sum(ov), len(ov)))
for s, e in ref[:4]:
    prev_psf = [p for p in psf if p[0] <= s]
    if prev_psf:
        print("  refocus start %.1f us after psf start (psf lasts %.1f), refocus lasts %.1f" % ((s - prev_psf[-1][0]) / 1e3, (prev_psf[-1][1] - prev_psf[-1][0]) / 1e3, (e - s) / 1e3))
