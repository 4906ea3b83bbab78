# L2 / L1 counters of the fused PSF-network kernel on the small workload (tools/m2_small.py); 4 counters per pass, bounded
R=$GRAFT_REPO_ROOT; TAG=${1:-r05_m2l2}
cd /tmp; export TMPDIR=/tmp
PM="timeout 150 rocprofv3 --kernel-trace --output-format csv"
CMD="python3 $R/tools/m2_small.py"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- $CMD > /dev/null 2>&1
$PM --pmc TCC_REQ TCC_HIT TCC_MISS TCC_READ -d $R/gpurun_out/${TAG}_a -- $CMD > /dev/null 2>&1; echo "pass a rc=$?"
$PM --pmc TCC_EA0_RDREQ TCC_BUSY TCC_CYCLE TCC_TAG_STALL -d $R/gpurun_out/${TAG}_b -- $CMD > /dev/null 2>&1; echo "pass b rc=$?"
$PM --pmc TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES -d $R/gpurun_out/${TAG}_c -- $CMD > /dev/null 2>&1; echo "pass c rc=$?"
$PM --pmc SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/${TAG}_d -- $CMD > /dev/null 2>&1; echo "pass d rc=$?"
cd $R
python3 - <<PY
import csv, glob, collections
for t in "abcd":
    for f in glob.glob("gpurun_out/${TAG}_%s/**/*counter_collection.csv" % t, recursive=True):
        acc, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            if "psfnet_fused" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        print(t, {k: acc[k] / cnt[k] for k in acc})
for f in glob.glob("gpurun_out/${TAG}_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "psfnet_fused" in r["Name"]: print("stats", r["Calls"], r["AverageNs"])
PY
