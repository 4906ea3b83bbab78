# Evidence for profiles/: bench JSON lines (M1 default, M2), rocprofv3 kernel stats of the M1 bench, conv traffic PMC.
R=$GRAFT_REPO_ROOT; TAG=${1:-r01_j}
cd $R; python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --mode m2 > gpurun_out/${TAG}_bench_m2.json 2>> gpurun_out/${TAG}_bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --spinup-s 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --spinup-s 0 > /dev/null 2>&1
cat $R/gpurun_out/${TAG}_bench.json | cut -c1-260
