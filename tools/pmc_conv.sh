R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/sb_pmc1 -- python3 $R/tools/kbench.py --rounds 1 --iters 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $R/gpurun_out/sb_pmc2 -- python3 $R/tools/kbench.py --rounds 1 --iters 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/sb_pmc3 -- python3 $R/tools/kbench.py --rounds 1 --iters 2 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
for d in ("sb_pmc1", "sb_pmc2", "sb_pmc3"):
    f = glob.glob(f"{R}/gpurun_out/{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sbatch" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
