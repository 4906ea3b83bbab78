import csv, sys, glob
kt = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
mc = glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True)
rows = []
for r in csv.DictReader(open(kt)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
if mc:
    for r in csv.DictReader(open(mc[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
# take a window in the last quarter
n = len(rows)
w = rows[int(n * 0.6):int(n * 0.6) + 45]
t0 = w[0][0]
prev_end = t0
for s, e, name in w:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {name}")
    prev_end = max(prev_end, e)
