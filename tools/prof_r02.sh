# Round-2 evidence for profiles/: kernel stats of the M1 bench, the fit loop and the gather kernels; PMC passes (separate runs).
R=$GRAFT_REPO_ROOT; TAG=${1:-r02_a}
cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 > /dev/null 2>&1
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_fit_stats -- python3 $R/bench.py --mode fit --steps 100 > /dev/null 2>&1
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_lp_stats -- python3 $R/tools/local_psf_bench.py > /dev/null 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_lp_fetch -- python3 $R/tools/local_psf_bench.py > /dev/null 2>&1
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_psf_pmc1 -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_psf_pmc2 -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --spinup-s 0 > /dev/null 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --spinup-s 0 > /dev/null 2>&1
cd $R; python bench.py --mode fit > gpurun_out/${TAG}_bench_fit.json 2>/dev/null; python bench.py > gpurun_out/${TAG}_bench.json 2>/dev/null
python bench.py --streams 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_2streams.json 2>/dev/null
AADFF_REFOCUS_OVERLAP=1 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_refocus_overlap.json 2>/dev/null
python bench.py --mode m2 > gpurun_out/${TAG}_bench_m2.json 2>/dev/null
python tools/soak.py > gpurun_out/${TAG}_soak.txt 2>&1; AADFF_REFOCUS_OVERLAP=1 python tools/soak.py >> gpurun_out/${TAG}_soak.txt 2>&1
find gpurun_out/${TAG}_* -name "*_kernel_stats.csv" | head; cut -c1-200 gpurun_out/${TAG}_bench.json
