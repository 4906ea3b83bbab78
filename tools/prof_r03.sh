# Round-3 evidence for profiles/ (run on the GPU box: gpurun -- 'bash tools/prof_r03.sh r03_a'); every rocprofv3 pass is its own run.
#  <tag>_stats       kernel stats of the DEFAULT bench command (two stacks in flight; the solo-leg launches have their own rows:
#                    conv_psf_map_sbatch_kernel<..., true, ...> / psf_points_kernel<true> = the launches bench.py's roofline / trace blocks time)
#  <tag>_stats_s1    kernel stats of `bench.py --streams 1` (every kernel alone on the device throughout)
#  PMC passes on `--streams 1` (counters of concurrently running kernels cannot be told apart)
R=$GRAFT_REPO_ROOT; TAG=${1:-r03_a}
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_s1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 200 --warmup 20 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_fit_stats -- python3 $R/bench.py --mode fit --steps 100 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_lp_stats -- python3 $R/tools/local_psf_bench.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_psf_pmc1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_psf_pmc2 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_conv_pmc -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 > /dev/null 2>&1
cd $R
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --streams 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_1stream.json 2>/dev/null
python bench.py --mode c3 > gpurun_out/${TAG}_bench_c3.json 2>/dev/null
python bench.py --mode fit > gpurun_out/${TAG}_bench_fit.json 2>/dev/null
python bench.py --mode m2 > gpurun_out/${TAG}_bench_m2.json 2>/dev/null
python tools/conv_timeline.py --json gpurun_out/${TAG}_conv_timeline.json > /dev/null 2>&1
AADFF_CONV_PAIR=1 python tools/conv_timeline.py --json gpurun_out/${TAG}_conv_timeline_paired.json > /dev/null 2>&1
python tools/latency_breakdown.py > gpurun_out/${TAG}_latency_breakdown.txt 2>&1
python tools/parity_per_slice.py --label shipped > gpurun_out/${TAG}_parity_per_slice_shipped.json 2>/dev/null
AADFF_LIB=$R/aberration-aware-depth-from-focus_amd/csrc/libaadff_literal.so python tools/parity_per_slice.py --label literal > gpurun_out/${TAG}_parity_per_slice_literal.json 2>/dev/null
python tools/parity_per_slice.py --strict --label strict > gpurun_out/${TAG}_parity_per_slice_strict.json 2>/dev/null
python tools/soak.py > gpurun_out/${TAG}_soak.txt 2>&1
find gpurun_out/${TAG}_* -name "*_kernel_stats.csv" | head; cut -c1-200 gpurun_out/${TAG}_bench.json
