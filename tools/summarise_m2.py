#!/usr/bin/env python3
"""Digest of tools/prof_r05_m2.sh: per launch of psfnet_fused_kernel the rocprofv3 duration and the PMC sums, plus the derived
figures DESIGN.md section 4.5 quotes (matrix-pipe busy share, L2 request rate and hit rate, LDS conflict share).  Writes
profiles/<tag>_kernel_pmc.json and copies the kernel stats."""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05_m2"
KEY = "psfnet_fused_kernel"
out = {"kernel": KEY, "command": "bench.py --mode m2 --steps 6 --warmup 2", "counters_per_launch": {}}
for f in glob.glob(os.path.join(REPO, "gpurun_out", f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(REPO, "profiles", f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(f)):
        if KEY in r["Name"]:
            out["rocprofv3"] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 1), "min_us": round(float(r["MinNs"]) / 1e3, 1)}
for n in (1, 2, 3, 4):
    for f in glob.glob(os.path.join(REPO, "gpurun_out", f"{tag}_pmc{n}", "**", "*counter_collection.csv"), recursive=True):
        acc, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            if KEY in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
        for k in acc:
            out["counters_per_launch"][k] = acc[k] / cnt[k]
c = out["counters_per_launch"]
d = {}
if "SQ_BUSY_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
    d["mfma_busy_over_4x_sq_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * c["SQ_BUSY_CYCLES"]), 4)
if "SQ_INSTS_MFMA" in c and "rocprofv3" in out:
    t = out["rocprofv3"]["avg_us"] * 1e-6
    # v_mfma_f32_16x16x32_f16: 16 cycles of one SIMD's matrix pipe; 1024 SIMDs at 2.4 GHz
    d["mfma_pipe_share_from_instruction_count"] = round(c["SQ_INSTS_MFMA"] * 16 / (1024 * 2.4e9 * t), 4)
    d["mfma_instructions_per_launch"] = c["SQ_INSTS_MFMA"]
if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
    d["wave_cycles_waiting_share"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)
if "SQ_WAIT_INST_LDS" in c and "SQ_WAVE_CYCLES" in c:
    d["wave_cycles_waiting_on_lds_share"] = round(c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"], 4)
if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
    d["lds_bank_conflict_share_of_lds_cycles"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
if "TCC_REQ" in c and "rocprofv3" in out:
    t = out["rocprofv3"]["avg_us"] * 1e-6
    d["l2_requests_per_launch"] = c["TCC_REQ"]
    d["l2_request_rate_G_per_s"] = round(c["TCC_REQ"] / t / 1e9, 2)
    d["l2_bandwidth_TB_per_s_at_128B_per_request"] = round(c["TCC_REQ"] * 128 / t / 1e12, 2)
    if "TCC_HIT" in c and "TCC_MISS" in c and c["TCC_HIT"] + c["TCC_MISS"]:
        d["l2_hit_rate"] = round(c["TCC_HIT"] / (c["TCC_HIT"] + c["TCC_MISS"]), 4)
    if "TCC_EA0_RDREQ" in c:
        d["l2_fabric_read_requests_per_launch"] = c["TCC_EA0_RDREQ"]
    if "TCC_BUSY" in c and "TCC_CYCLE" in c and c["TCC_CYCLE"]:
        d["l2_busy_share"] = round(c["TCC_BUSY"] / c["TCC_CYCLE"], 4)
if "TCP_TCC_READ_REQ_LATENCY" in c and c.get("TCP_TCC_READ_REQ"):
    d["l1_to_l2_read_latency_cycles"] = round(c["TCP_TCC_READ_REQ_LATENCY"] / c["TCP_TCC_READ_REQ"], 1)
out["derived"] = d
src = os.path.join(REPO, "aberration-aware-depth-from-focus_amd", "csrc", "psfnet.hip")
out["code_sha256"] = hashlib.sha256(open(src, "rb").read()).hexdigest()[:12]
path = os.path.join(REPO, "profiles", f"{tag}_kernel_pmc.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
