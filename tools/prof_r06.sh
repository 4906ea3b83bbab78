# Round-6 evidence for profiles/ (run on the GPU box: gpurun -- 'bash tools/prof_r06.sh r06_a'); every rocprofv3 pass is its own run.
# The hashes of the sources and of the library that ran are recorded FIRST (collection time), tools/summarise_profiles.py stamps the
# digests with them.
R=$GRAFT_REPO_ROOT; TAG=${1:-r05_a}
mkdir -p $R/gpurun_out
python3 - <<PY
import hashlib, json, os
R = "$R"; c = os.path.join(R, "aberration-aware-depth-from-focus_amd", "csrc")
files = {"conv.hip": os.path.join(c, "conv.hip"), "trace.hip": os.path.join(c, "trace.hip"), "strict_fused.hip": os.path.join(c, "strict_fused.hip"),
         "strict.hip": os.path.join(c, "strict.hip"), "strict_math.h": os.path.join(c, "strict_math.h"), "strict_math2.h": os.path.join(c, "strict_math2.h"),
         "psfnet.hip": os.path.join(c, "psfnet.hip"), "common.h": os.path.join(c, "common.h"), "aadff.h": os.path.join(R, "include", "aadff.h"),
         "libaadff.so": os.path.join(c, "libaadff.so")}
json.dump({k: hashlib.sha256(open(v, "rb").read()).hexdigest() for k, v in files.items()}, open(os.path.join(R, "gpurun_out", "${TAG}_code_sha256.json"), "w"), indent=1)
PY
cd /tmp; export TMPDIR=/tmp
RP="timeout 300 rocprofv3 --kernel-trace --stats --output-format csv"
$RP -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
$RP -d $R/gpurun_out/${TAG}_stats_s1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 200 --warmup 20 > /dev/null 2>&1
$RP -d $R/gpurun_out/${TAG}_single_stats -- python3 $R/tools/kbench.py --rounds 5 --iters 20 > /dev/null 2>&1
$RP -d $R/gpurun_out/${TAG}_strict_stats -- python3 $R/tools/strict_profile.py 12 --render > $R/gpurun_out/${TAG}_strict_profile.txt 2>&1
$RP -d $R/gpurun_out/${TAG}_edge_stats -- python3 $R/tools/edge_bench.py 12 > $R/gpurun_out/${TAG}_edge_bench_under_rocprof.txt 2>&1
$RP -d $R/gpurun_out/${TAG}_m1l_stats -- python3 $R/bench.py --mode m1l --no-cpu-baseline --steps 20 > $R/gpurun_out/${TAG}_bench_m1l_under_rocprof.json 2>/dev/null
$RP -d $R/gpurun_out/${TAG}_dropin_stats -- python3 $R/tools/dropin_bench.py 20 > $R/gpurun_out/${TAG}_dropin.txt 2>/dev/null
PM="timeout 300 rocprofv3 --kernel-trace --output-format csv"
$PM --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/${TAG}_psf_pmc1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 --solo-steps 4 > /dev/null 2>&1
$PM --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC -d $R/gpurun_out/${TAG}_psf_pmc2 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 --solo-steps 4 > /dev/null 2>&1
$PM --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/${TAG}_conv_pmc -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 --solo-steps 4 > /dev/null 2>&1
$PM --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 > /dev/null 2>&1
$PM --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 5 --warmup 2 --spinup-s 0 > /dev/null 2>&1
$PM --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS SQ_INSTS_SALU -d $R/gpurun_out/${TAG}_strict_pmc -- python3 $R/tools/strict_profile.py 4 > /dev/null 2>&1
cd $R
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_1stream.json 2>/dev/null
timeout 300 python bench.py --mode c3 > gpurun_out/${TAG}_bench_c3.json 2>/dev/null
timeout 300 python bench.py --mode fit > gpurun_out/${TAG}_bench_fit.json 2>/dev/null
timeout 300 python bench.py --mode m2 > gpurun_out/${TAG}_bench_m2.json 2>/dev/null
timeout 300 python bench.py --mode m1l > gpurun_out/${TAG}_bench_m1l.json 2>/dev/null
timeout 300 python tools/conv_ks_sweep.py > gpurun_out/${TAG}_conv_ks_sweep.txt 2>/dev/null
timeout 300 python tools/kbench.py > gpurun_out/${TAG}_kbench.txt 2>/dev/null
timeout 300 python tools/conv_blkw_probe.py 2>/dev/null | tail -22 > gpurun_out/${TAG}_conv_blkw_probe.txt
timeout 200 python tools/conv_single_timeline.py --ks 21 --grid 7 --json gpurun_out/${TAG}_conv_blkw_timeline_ks21.json > /dev/null 2>&1
AADFF_CONV_BLKW_RB=24 timeout 200 python tools/conv_single_timeline.py --ks 21 --grid 7 --json gpurun_out/${TAG}_conv_blkw_timeline_ks21_rb24.json > /dev/null 2>&1
PROBE_DEPTHS=3 timeout 300 python tools/strict_pipe_probe.py 30 2>/dev/null | grep -v "^/opt" > gpurun_out/${TAG}_strict_pipe_probe.txt
PROBE_DEPTHS=2,4 timeout 300 python tools/edge_bench.py 40 2>/dev/null | grep -v "^/opt" > gpurun_out/${TAG}_edge_bench.txt
AADFF_CALL_ZERO_COPY=0 timeout 300 python tools/dropin_bench.py 20 > gpurun_out/${TAG}_dropin_with_copies.txt 2>/dev/null
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^/opt" > gpurun_out/${TAG}_gputest.txt
timeout 900 python tools/parity_seeds.py --cases 4 --first 5 > gpurun_out/${TAG}_parity_seeds.json 2> gpurun_out/${TAG}_parity_seeds.err
for M in load pipeline; do for P in strict edge; do MODE=$M PARITY=$P timeout 300 python tools/concurrency_probe.py 500 2>&1 | tail -1; done; done > gpurun_out/${TAG}_concurrency_probe.txt
# the raw rocprofv3 directories exceed what gpurun copies back (64 MiB): reduce them HERE (tools/summarise_profiles.py writes the judged
# summaries into profiles/), ship the summaries in gpurun_out/${TAG}_profiles/ and drop the raw traces
python3 tools/summarise_profiles.py ${TAG} > gpurun_out/${TAG}_summarise.log 2>&1
mkdir -p gpurun_out/${TAG}_profiles
cp profiles/${TAG}_* gpurun_out/${TAG}_profiles/ 2>/dev/null
cp profiles/conv_traffic.json profiles/psf_kernel_pmc.json profiles/strict_kernel_pmc.json gpurun_out/${TAG}_profiles/ 2>/dev/null
for d in stats stats_s1 single_stats strict_stats edge_stats m1l_stats dropin_stats psf_pmc1 psf_pmc2 conv_pmc fetch write strict_pmc; do rm -rf gpurun_out/${TAG}_$d; done
du -sh gpurun_out > gpurun_out/${TAG}_size.txt
