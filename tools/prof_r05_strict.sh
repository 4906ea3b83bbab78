# Round-5 evidence for the fused strict stack (gpurun -- 'bash tools/prof_r05_strict.sh r05_a'): kernel stats + VALU counters of tools/strict_profile.py
R=$GRAFT_REPO_ROOT; TAG=${1:-r05_a}
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_strict_stats -- python3 $R/tools/strict_profile.py 12 > $R/gpurun_out/${TAG}_strict_profile.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS SQ_INSTS_SALU -d $R/gpurun_out/${TAG}_strict_pmc -- python3 $R/tools/strict_profile.py 4 > /dev/null 2>&1
cd $R
find gpurun_out/${TAG}_strict_stats -name "*kernel_stats.csv" -exec head -8 {} \;
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/${TAG}_strict_pmc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
    for k in acc:
        if "fused" in k: print(k, n[k], {c: v / max(n[k], 1) for c, v in acc[k].items()})
PY
