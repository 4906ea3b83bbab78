# Round-5 counter evidence for the fused PSF-network kernel: gpurun -- 'bash tools/prof_r05_m2.sh r05_m2'
# Every PMC set is its own rocprofv3 run (no trace domains besides --kernel-trace); tools/summarise_m2.py digests them.
R=$GRAFT_REPO_ROOT; TAG=${1:-r05_m2}
cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --mode m2 --steps 6 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- $CMD > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
PM="timeout 300 rocprofv3 --kernel-trace --output-format csv"
$PM --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/${TAG}_pmc1 -- $CMD > /dev/null 2>&1
$PM --pmc TCC_REQ TCC_READ TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_BUSY TCC_CYCLE TCC_TAG_STALL -d $R/gpurun_out/${TAG}_pmc2 -- $CMD > /dev/null 2>&1
$PM --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAVES -d $R/gpurun_out/${TAG}_pmc3 -- $CMD > /dev/null 2>&1
$PM --pmc TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCC_WRITE_REQ -d $R/gpurun_out/${TAG}_pmc4 -- $CMD > /dev/null 2>&1
cd $R
python3 tools/summarise_m2.py $TAG
