# HBM traffic of the conv kernel in bench.py: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01h_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01h_write -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01h_stats -- python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 > /dev/null 2>&1
cd $R; python bench.py > gpurun_out/r01_h_bench.json 2> gpurun_out/r01_h_bench.err; cat gpurun_out/r01_h_bench.json | cut -c1-300
