# Round-4 evidence for profiles/ (run on the GPU box: gpurun -- 'bash tools/prof_r04.sh r04_a'); every rocprofv3 pass is its own run.
#  <tag>_stats       kernel stats of the DEFAULT bench command (two stacks in flight; the solo-leg launches have their own rows:
#                    conv_psf_map_sbatch_kernel<..., true> / psf_points_kernel<true> = the launches bench.py's roofline / trace blocks time)
#  <tag>_stats_s1    kernel stats of `bench.py --streams 1` (every kernel alone on the device throughout)
#  <tag>_single_stats kernel stats of tools/kbench.py (lone-slice render_psf_map: conv_psf_map_blk_kernel) 
#  <tag>_strict_stats kernel stats of a strict-parity stack (tools/strict_profile.py)
#  PMC passes on `--streams 1` (counters of concurrently running kernels cannot be told apart)
R=$GRAFT_REPO_ROOT; TAG=${1:-r04_a}
cd /tmp; export TMPDIR=/tmp
RP="timeout 300 rocprofv3 --kernel-trace --stats --output-format csv"
$RP -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
$RP -d $R/gpurun_out/${TAG}_stats_s1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 200 --warmup 20 > /dev/null 2>&1
$RP -d $R/gpurun_out/${TAG}_fit_stats -- python3 $R/bench.py --mode fit --steps 100 > /dev/null 2>&1
$RP -d $R/gpurun_out/${TAG}_single_stats -- python3 $R/tools/kbench.py --rounds 5 --iters 20 > /dev/null 2>&1
$RP -d $R/gpurun_out/${TAG}_strict_stats -- python3 $R/tools/strict_profile.py 6 > /dev/null 2>&1
PM="timeout 300 rocprofv3 --kernel-trace --output-format csv"
$PM --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/${TAG}_psf_pmc1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 --solo-steps 4 > /dev/null 2>&1
$PM --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC -d $R/gpurun_out/${TAG}_psf_pmc2 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 --solo-steps 4 > /dev/null 2>&1
$PM --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/${TAG}_conv_pmc -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 --solo-steps 4 > /dev/null 2>&1
$PM --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 > /dev/null 2>&1
$PM --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 > /dev/null 2>&1
$PM --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/${TAG}_single_pmc -- python3 $R/tools/kbench.py --rounds 1 --iters 3 > /dev/null 2>&1
$PM --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_single_fetch -- python3 $R/tools/kbench.py --rounds 1 --iters 3 > /dev/null 2>&1
$PM --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_single_write -- python3 $R/tools/kbench.py --rounds 1 --iters 3 > /dev/null 2>&1
cd $R
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_1stream.json 2>/dev/null
timeout 300 python bench.py --mode c3 > gpurun_out/${TAG}_bench_c3.json 2>/dev/null
timeout 300 python bench.py --mode fit > gpurun_out/${TAG}_bench_fit.json 2>/dev/null
timeout 300 python bench.py --mode m2 > gpurun_out/${TAG}_bench_m2.json 2>/dev/null
AADFF_FORCE_GROUP=1 timeout 300 python bench.py --gather --no-cpu-baseline > gpurun_out/${TAG}_bench_rccl1_gather.json 2>/dev/null
timeout 300 python tools/conv_timeline.py --json gpurun_out/${TAG}_conv_timeline.json > /dev/null 2>&1
timeout 300 python tools/conv_single_timeline.py --json gpurun_out/${TAG}_conv_single_timeline.json > /dev/null 2>&1
AADFF_CONV_PATH=toeplitz timeout 300 python tools/conv_single_timeline.py --json gpurun_out/${TAG}_conv_single_timeline_toeplitz.json > /dev/null 2>&1
timeout 300 python tools/kbench.py --rounds 7 --iters 20 > gpurun_out/${TAG}_kbench.txt 2>&1
AADFF_CONV_PATH=toeplitz timeout 300 python tools/kbench.py --rounds 7 --iters 20 > gpurun_out/${TAG}_kbench_toeplitz.txt 2>&1
timeout 300 python tools/strict_profile.py 8 > gpurun_out/${TAG}_strict_profile.txt 2>&1
timeout 300 python tools/latency_breakdown.py > gpurun_out/${TAG}_latency_breakdown.txt 2>&1
timeout 600 python tools/soak.py > gpurun_out/${TAG}_soak.txt 2>&1
# gpurun merges at most 64 MiB back: keep the summaries (kernel stats, counter collections), drop traces and raw dumps
find gpurun_out -path "*${TAG}_*" \( -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*_raw.npy" -o -name "*.db" \) -delete
du -s gpurun_out/* | sort -n | tail -6; du -sh gpurun_out | tail -1; ls gpurun_out | grep ${TAG} | wc -l; cut -c1-200 gpurun_out/${TAG}_bench.json
