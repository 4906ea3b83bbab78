#!/usr/bin/env python3
"""Focal-stack MP/s on MI355X (BASELINE.json metric).

Step = one M1 focal stack: 1024x1024 RGB, 10 focus distances, 11x11 PSF grid, ks 11,
2048 rays per point and wavelength (+2048 chief rays), lens rf50mm:
    host pupil-sample draws -> refocus (S states) -> fused ray-trace/PSF-grid kernel ->
    stack-fused patch-PSF convolution,
inputs (image, lens tables) resident in HBM.  One process per GPU; at N > 1 every rank
renders its own scene's stack (weak scaling) and, with --gather, the
stacks are all-gathered over RCCL on a side stream overlapped with the next step.

Prints ONE JSON line (rank 0).  `roofline` is the patch-PSF convolution kernel against
the 8 TB/s HBM roofline with ALGORITHMIC bytes (24 B/pixel/slice);
`cpu_baseline` is the oracle (CPU restatement of the reference) timed on this host.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "aberration-aware-depth-from-focus_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

# numpy's OpenBLAS comes with a pool of up to 64 pthreads that busy-wait for ~10 ms after every BLAS call (np.linalg.norm of a stack in
# the parity legs): on a GPU box with a 16-CPU quota that burst, a few ms before a host-bound timed loop, used up the cgroup's 100 ms
# period and froze the process for 55-85 ms somewhere in the loop (one step of an edge-mode leg at 80 ms instead of 1; found with
# `host_cpu_quota` / AADFF_BENCH_THREADS=1 below).  One BLAS thread: the norms take 1 s longer in total, nothing spins.
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
import torch

H = W = 1024
S, GRID, KS, SPP = 10, 11, 11, 2048
HBM_PEAK = 8.0e12                       # B/s, MI355X_MICROARCH.md
ALG_BYTES_PER_SLICE = 2 * 3 * H * W * 4  # read image once + write output once


_THREAD_MARKS = []


def _threads_cpu():
    """{tid: user + system clock ticks} of this process's threads"""
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read()
            rest = f[f.rindex(")") + 2:].split()
            out[int(t)] = int(rest[11]) + int(rest[12])
        except (OSError, ValueError):
            pass
    return out


def _thread_mark(label):
    if os.environ.get("AADFF_BENCH_THREADS") == "1":
        _THREAD_MARKS.append((label, set(_threads_cpu())))


def _cgroup_cpu():
    """cpu.stat of this process's cgroup (v2), or None"""
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
    except (OSError, ValueError):
        return None


def usable_cpus():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 logical CPUs under a 16-CPU quota; oversubscribing it stalls OpenMP)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def file_sha256(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def cpu_baseline(lens_path, img, dbar, fds, budget_s=30.0):
    """Oracle M1 slices (refocus -> psf_map -> render_psf_map) on the host cores, seed 0 — also the reference
    pixels of the `parity` block.  Returns (record, [slices])."""
    from oracle import conv as oconv
    from oracle.lens import OracleLens
    torch.set_num_threads(usable_cpus())
    lens = OracleLens(lens_path, sensor_res=(H, W))
    torch.manual_seed(0)
    t0 = time.perf_counter()
    slices = []
    for f in fds:
        lens.refocus(float(f))
        pm = lens.psf_map(depth=dbar, grid=GRID, ks=KS, spp=SPP)
        slices.append(oconv.render_psf_map(img, pm, GRID))
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    n = len(slices)
    return {"value": round(n * H * W / 1e6 / dt, 4), "unit": "MP/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model(),
            "sample": f"{n} of {S} slices of the same 1024x1024 M1 stack (oracle: refocus+psf_map+render_psf_map), {dt:.1f} s"}, slices


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gather", action="store_true", help="all-gather the rendered stacks over RCCL (config 3)")
    ap.add_argument("--emulate-ranks", action="store_true",
                    help="run the --gpus N ranks on ONE GPU with the gloo backend (RCCL refuses two ranks per device): "
                         "exercises launcher, sharding and gather order on a 1-GPU box; not a scaling measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--device-rng", action="store_true", help="draw pupil samples on the GPU (not sample-comparable)")
    ap.add_argument("--spinup-s", type=float, default=2.0,
                    help="untimed device spin-up before the warm-up steps [s]: the FIRST GPU process on a fresh box needs ~2 s of load before its clocks settle "
                         "(measured: 29.8 k MP/s with 0.3 s against 31.3 k with 3 s, later processes 31.0-31.2 k either way)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("AADFF_BENCH_STREAMS", "2")),
                    help="M1: stacks in flight on this many HIP streams of the one GPU (aadff.focal_stack.StackPipeline).  Default 2: the "
                         "VALU-bound PSF-grid kernel of stack i+1 runs beside the LDS/MFMA/HBM-bound convolution of stack i and fills the "
                         "dispatch gaps and kernel tails (+8-10 %% over one stream).  The per-kernel durations of the `roofline` and "
                         "`trace` blocks are measured in an untimed SOLO leg after the timed region (one stack at a time, device idle "
                         "before each, HIP events attached to the kernels' own dispatches), because kernels of different stacks share "
                         "the device in the timed region.  --streams 1: every kernel alone on the device throughout (round-2 default; "
                         "rocprofv3 --stats of that command reproduces the solo-leg durations)")
    ap.add_argument("--solo-steps", type=int, default=40, help="stacks of the untimed solo leg that times the convolution and PSF-grid kernels")
    ap.add_argument("--mode", choices=("m1", "m2", "fit", "c3", "m1l"), default="m1",
                    help="m1 (default, BASELINE.json metric): ray-traced PSF grid + patch convolution; "
                         "m2: RGB-D stack through the PSF surrogate network (PSFNet.render, SURVEY.md 8f-1); "
                         "fit: 1_fit_psfnet.py training iterations (ray-traced targets + MLP step, BASELINE config 3); "
                         "c3: 16 scenes x 10 slices sharded in whole-scene blocks, all-gathered row by row (BASELINE config 3, strong scaling); "
                         "m1l: M1-layered (SURVEY.md 8(d)): the depth MAP in 4 layers, one ray-traced PSF map per (slice, layer), per-pixel selection "
                         "fused into the stack convolution")
    args = ap.parse_args()
    _thread_mark("after the imports")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Self-launch: one child per rank, started BEFORE this process has made any GPU call (it never makes one).
        from aadff.dist import spawn_ranks
        raise SystemExit(spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus, emulate=args.emulate_ranks))
    if os.environ.get("AADFF_BENCH_HANG_DUMP_S"):           # debugging aid: every thread's stack after N seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["AADFF_BENCH_HANG_DUMP_S"]), exit=True)
    # every rank onto the CPUs of its GPU's NUMA node, before the first GPU call (one placement line per rank on stderr)
    from aadff.dist import pin_to_gpu_numa
    pin_to_gpu_numa()
    # the GPU box exposes 256 logical CPUs under a 16-CPU cgroup quota: unbounded OpenMP pools stall small torch CPU ops
    torch.set_num_threads(max(1, usable_cpus() // max(1, int(os.environ.get("WORLD_SIZE", "1")))))
    if args.mode == "m2":
        return main_m2(args)
    if args.mode == "fit":
        return main_fit(args)
    if args.mode == "c3":
        return main_c3(args)
    if args.mode == "m1l":
        return main_m1l(args)

    import torch.distributed as dist
    from aadff import dist as adist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    adist.init_from_env(backend="nccl", device=dev)       # RCCL (gloo when ranks are emulated on one GPU); no-op for one process

    from aadff.focal_stack import StackPipeline
    from aadff.sampling import DeviceSampler
    from aadff.synth import synth_depth_mm, synth_rgb
    from deeplens.optics import Lensgroup

    lens_path = os.path.join(REPO, "lenses", "rf50mm", "lens.json")
    img_h = torch.from_numpy(synth_rgb(H, W, seed=1234 + rank))[None]
    depth = synth_depth_mm(H, W, seed=5678 + rank)
    dbar = -float(depth.mean())
    fds = -np.linspace(depth.min(), depth.max(), S)
    lens = Lensgroup(lens_path, sensor_res=(H, W), device=dev)
    if args.device_rng:
        lens.sampler = DeviceSampler(dev, seed=rank)
    img = img_h.to(dev)
    multi = world > 1 or adist.grouped()               # collectives are issued (AADFF_FORCE_GROUP=1: also on a one-rank RCCL group)
    n_streams = 1 if (multi and args.gather) else max(1, args.streams)
    pipe = StackPipeline(lens, S, H, W, 1, 3, GRID, KS, SPP, depth=n_streams)
    plan = pipe.plans[0]
    # Kernel durations: HIP events ATTACHED to a kernel's own dispatch (aadff_time_next_launch -> hipExtLaunchKernelGGL):
    # the kernel's begin-to-end time on its launch stream, the quantity rocprofv3 reports.  Taken in the solo leg below.
    def hip_event():
        e = torch.cuda.Event(enable_timing=True)
        e.record()                                   # creates the HIP event behind it (its handle goes to the launch)
        return e

    def hip_elapsed_ms(a, b):
        b.synchronize()
        return float(a.elapsed_time(b))

    # --gather: the all-gather of step i runs on a side stream while step i+1 renders into the other output buffer
    ring = None
    if multi and args.gather:
        ring = adist.GatherRing(lambda: torch.empty_like(plan.out), world, slots=2, device=dev)

    def step(i, timed=False):
        torch.manual_seed(i)                # every stack is seeded (26 us of host time, hidden in the pipeline)
        cur = pipe.plans[pipe.turn % pipe.depth]
        if ring is not None:
            k, cur.out = ring.acquire()
        out, _ = pipe.render(lens, img, dbar, fds, inputs_ready=True)      # the image is resident; outputs are not consumed here
        if ring is not None:
            ring.submit(k)
        return out

    def barrier():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # device spin-up (untimed, before the warm-up steps): the first GPU process on a fresh box needs ~2 s under load before
    # the clocks settle; a cold first run otherwise reads 4-10 % low with identical code
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup_s:
        for i in range(16):
            step(i)
        torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    t_issued = time.perf_counter() - t0              # host side: every step of this rank queued
    if ring is not None:
        ring.drain()
    torch.cuda.synchronize(dev)
    dt_own = time.perf_counter() - t0                # this rank alone: its own steps done (before the barrier that waits for the slowest)
    barrier()
    dt = adist.all_reduce_max(time.perf_counter() - t0)
    # one line per rank BEFORE the max-reduce hides it: a slow rank (wrong NUMA node, CPU quota, a neighbour's load) is visible in the tail
    print(f"bench: rank {rank}/{world} (local {local}, pid {os.getpid()}): own {dt_own / args.steps * 1e3:.4f} ms per step, "
          f"host-side {t_issued / args.steps * 1e3:.4f} ms per step to queue, job (max over ranks) {dt / args.steps * 1e3:.4f} ms per step",
          file=sys.stderr, flush=True)
    ranks_seen = {"world_size": world, "backend": None, "own_ms_per_step": None}
    if multi:
        assert dist.get_world_size() == world == max(args.gpus, 1) or adist.forced_group(), (dist.get_world_size(), world, args.gpus)
        own = torch.tensor([dt_own / args.steps * 1e3, t_issued / args.steps * 1e3], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        rows = [torch.zeros_like(own) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, own)
        ranks_seen = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                      "own_ms_per_step": [round(float(r[0]), 4) for r in rows], "host_ms_per_step": [round(float(r[1]), 4) for r in rows]}

    # ---- untimed SOAK leg: the same pipelined step for >= 1 s, so that a sampler with a coarse cadence (the driver's gpu_busy reading,
    # rocm-smi) sees the GPU under this load at all - the timed region of the default command is ~7 ms
    soak_s = float(os.environ.get("AADFF_BENCH_SOAK_S", "1.0"))
    n_soak, t_soak = 0, time.perf_counter()
    while time.perf_counter() - t_soak < soak_s:
        for i in range(64):
            step(n_soak + i)
        n_soak += 64
        torch.cuda.synchronize(dev)
    if ring is not None:
        ring.drain()
    torch.cuda.synchronize(dev)
    soak_ms = (time.perf_counter() - t_soak) / max(n_soak, 1) * 1e3

    # ---- untimed: kernel error flags of every step so far (NaN residual / no valid chief ray -> the number is void)
    torch.cuda.synchronize(dev)
    bits = 0
    for p_ in pipe.plans:
        bits |= int(p_.flags.item())
        p_.flags.zero_()
    if bits & 3:
        print(f"bench: kernel flags 0x{bits:x} raised during the timed loop (bit0 NaN in Newton residual, bit1 no valid chief ray)",
              file=sys.stderr, flush=True)
        raise SystemExit(3)

    # ---- untimed: per-stack latency as SURVEY.md 8(d) defines it (first host call -> device idle), median of 20
    lat, lat_render = [], []
    if ring is None:
        # a LONE stack on its own plan: what a caller without a pipeline gets
        from aadff.focal_stack import StackPlan, render_focal_stack_m1 as _rfs1
        lplan = StackPlan(lens, S, H, W, 1, 3, GRID, KS, SPP)
        for i in range(24):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            torch.manual_seed(i)
            _rfs1(lens, img, dbar, fds, GRID, KS, SPP, plan=lplan, update_lens=False)
            torch.cuda.synchronize(dev)
            if i >= 4:
                lat.append(time.perf_counter() - t1)
        lplan.check_flags()
    for i in range(20 if ring is not None else 0):
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        step(i)
        ring.drain()
        torch.cuda.synchronize(dev)
        lat.append(time.perf_counter() - t1)
    for i in range(20 if ring is None else 0):   # the same with the seed outside the window (torch.manual_seed walks the device generators too)
        torch.cuda.synchronize(dev)
        torch.manual_seed(i)
        t1 = time.perf_counter()
        pipe.render(lens, img, dbar, fds, inputs_ready=True)
        torch.cuda.synchronize(dev)
        lat_render.append(time.perf_counter() - t1)
    # ---- untimed SOLO leg: the same stacks on ONE stream, queued back to back (the device stays busy and clocked up, and the
    # stream order keeps every kernel alone on the device - the condition rocprofv3 of `bench.py --streams 1` sees); on every
    # 4th stack the PSF-grid and the convolution kernel carry a HIP event pair on their own dispatch, and two stream events
    # bracket the convolution launch (round-1 method).  An attached pair costs ~10 us of queue time: this leg is not timed.
    from aadff.focal_stack import render_focal_stack_m1
    torch.cuda.synchronize(dev)
    solo, splan = [], pipe.plans[0]
    # ---- untimed: the same step on ONE stream (round 2's default), so that the line carries both schedules
    n_one = max(20, min(args.steps, 100))
    for i in range(4):
        torch.manual_seed(i)
        render_focal_stack_m1(lens, img, dbar, fds, GRID, KS, SPP, plan=splan, update_lens=False)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for i in range(n_one):
        torch.manual_seed(i)
        render_focal_stack_m1(lens, img, dbar, fds, GRID, KS, SPP, plan=splan, update_lens=False)
    torch.cuda.synchronize(dev)
    one_stream_ms = (time.perf_counter() - t1) / n_one * 1e3
    for i in range(4 * max(4, args.solo_steps)):
        torch.manual_seed(i)
        rec = None
        if i % 4 == 3:
            rec = {"psf": (hip_event(), hip_event()), "conv": (hip_event(), hip_event()),
                   "bracket": (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))}
            splan.psf_kernel_events, splan.conv_kernel_events, splan.conv_events = rec["psf"], rec["conv"], rec["bracket"]
        render_focal_stack_m1(lens, img, dbar, fds, GRID, KS, SPP, plan=splan, update_lens=False)
        splan.psf_kernel_events = splan.conv_kernel_events = splan.conv_events = None
        if rec is not None:
            solo.append(rec)
    torch.cuda.synchronize(dev)
    solo = solo[2:]                                                     # the first two re-warm the clocks after the latency leg
    psf_ms = float(np.median([hip_elapsed_ms(*r["psf"]) for r in solo]))
    # ---- untimed: the pixels of a seed-0 step for the parity block
    torch.cuda.synchronize(dev)
    out0 = step(0)
    torch.cuda.synchronize(dev)
    got = out0[0].cpu().numpy() if rank == 0 else None              # [3,S,H,W]

    # ---- untimed: the convolution kernel launched BACK TO BACK on one stream (tools/kbench.py's method: 7 rounds of 10 launches
    # between two stream events, median round): what the kernel sustains when nothing else separates its launches - longer than the
    # lone launch above, because consecutive launches contend for HBM at their head and tail (51-53 us against 40 us)
    from aadff import _abi as _abi_b2b
    b2b = []
    _st = _abi_b2b.stream_ptr(dev)
    for _round in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _abi_b2b.call("aadff_render_psf_map_stack", _abi_b2b.ptr(img), _abi_b2b.ptr(splan.psf_maps), _abi_b2b.ptr(splan.out), 1, 3, S, H, W, GRID, KS, _st)
        e1.record()
        torch.cuda.synchronize(dev)
        if _round:
            b2b.append(e0.elapsed_time(e1) / 10)
    conv_b2b_ms = float(np.median(b2b))

    conv_all = [hip_elapsed_ms(*r["conv"]) for r in solo]
    conv_ms = float(np.mean(conv_all))
    conv_bracket = [r["bracket"][0].elapsed_time(r["bracket"][1]) for r in solo]
    achieved = ALG_BYTES_PER_SLICE * S / (conv_ms * 1e-3)
    # PMC digests (rocprofv3 --pmc passes cannot run inside this process): quoted ONLY when they were collected on the very
    # sources the loaded library was built from - tools/summarise_profiles.py stamps each digest with the sha256 of the kernel's
    # source file; a digest of another build is reported as stale (null), not as a measurement
    def digest(name, source):
        path = os.path.join(REPO, "profiles", name)
        if not os.path.exists(path):
            return None, f"profiles/{name} missing"
        dj = json.load(open(path))
        have = file_sha256(os.path.join(REPO, "aberration-aware-depth-from-focus_amd", "csrc", source))
        if dj.get("code_sha256", {}).get(source) != have:
            return None, (f"stale: profiles/{name} was collected on another build of csrc/{source} (digest "
                          f"{str(dj.get('code_sha256', {}).get(source))[:12]}, this tree {have[:12]}); re-run tools/prof_r05.sh + tools/summarise_profiles.py")
        return dj, f"profiles/{name} (rocprofv3 PMC passes of this command on this build of csrc/{source}, sha256 {have[:12]}; not measured in this run)"

    def digest_multi(name, sources):
        """the same for a kernel compiled from several sources: quoted only while EVERY one of them still has the hash it was collected on"""
        path = os.path.join(REPO, "profiles", name)
        if not os.path.exists(path):
            return None, f"profiles/{name} missing"
        dj = json.load(open(path))
        for src in sources:
            have = file_sha256(os.path.join(REPO, "aberration-aware-depth-from-focus_amd", "csrc", src))
            if dj.get("code_sha256", {}).get(src) != have:
                return None, (f"stale: profiles/{name} was collected on another build of csrc/{src} (digest {str(dj.get('code_sha256', {}).get(src))[:12]}, "
                              f"this tree {have[:12]}); re-run tools/prof_r06.sh + tools/summarise_profiles.py")
        return dj, f"profiles/{name} (rocprofv3 passes of tools/strict_profile.py on this build of {', '.join(sources)}; not measured in this run)"
    tj, traffic_source = digest("conv_traffic.json", "conv.hip")
    traffic = tj.get("hbm_bytes_per_launch") if tj else None
    unique = (1 + S) * 3 * H * W * 4                     # bytes that MUST move per launch: the image once + S output slices (13.2 B/pixel/slice)
    vj, valu_source = digest("psf_kernel_pmc.json", "trace.hip")
    valu = vj.get("valu_busy") if vj else None
    if rank == 0:
        n_surf = len(lens.surfaces)
        steps_per_stack = 3 * (SPP + 2048) * GRID * GRID * n_surf * S         # rays x surfaces
        res = {
            "metric": "focal-stack MP/s (1024^2 x 10 slices, 11x11 PSF grid)",
            "value": round(world * S * H * W / 1e6 * args.steps / dt, 2), "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "latency_ms_p50": round(float(np.median(lat)) * 1e3, 4),
            # SURVEY.md 8(d)'s metric as defined there: S x H x W / (first host call -> device idle, one stack, median of 20)
            "value_survey_8d": round(S * H * W / 1e6 / float(np.median(lat)), 2),
            "latency_ms_p50_render_call_only": round(float(np.median(lat_render)) * 1e3, 4) if lat_render else None,
            "streams": n_streams,
            # who took part (the driver's first 8-rank run will want this): group size and backend as torch.distributed reports them,
            # every rank's own ms per step and host-side queueing time before the max over ranks
            "ranks_seen": ranks_seen,
            # the GPU boxes are shared hosts (256 logical CPUs, a 16-CPU quota per job, load averages of 15-30 from other jobs): the
            # host-bound legs (strict / edge pipelines at depth 4) read up to 2-4 x slower on a busy host - profiles/r06_zz_bench_busy_host.json
            "host": {"loadavg_1_5_15": [round(v, 1) for v in os.getloadavg()], "usable_cpus": usable_cpus()},
            "soak": {"ms_per_step": round(soak_ms, 4), "steps": n_soak, "seconds": round(soak_ms * n_soak / 1e3, 2),
                     "what": "the same pipelined step queued for >= 1 s after the timed region (untimed leg; synchronised every 64 steps)"},
            "one_stream": {"ms_per_step": round(one_stream_ms, 4), "value": round(S * H * W / 1e6 / (one_stream_ms * 1e-3), 2), "steps": n_one,
                           "what": "the same step queued on ONE HIP stream (every kernel alone on the device; untimed leg of this rank)"},
            "config": {"workload": "rf50mm, 1024x1024 synthetic RGB + depth plane, 10 focus distances, 11x11 PSF grid, "
                                   "ks 11, spp 2048 (+2048 chief), mode M1 (refocus -> psf_map -> render_psf_map)",
                       "stacks_per_step_per_gpu": 1, "pupil_samples": "device RNG" if args.device_rng else "host torch RNG, reference call order",
                       "gather": bool(ring is not None), "ranks_emulated_on_one_gpu": adist.emulated(),
                       "value_is": f"pipelined throughput: steps queued back to back, {n_streams} stack(s) in flight on {n_streams} HIP stream(s) "
                                   "of the GPU; latency_ms_p50 = one stack, host call to device idle",
                       "kernel_times_from": f"untimed solo leg after the timed region: the same stacks back to back on ONE stream, {len(solo)} "
                                            "of them with events on the kernels' dispatches (kernels of different stacks overlap in the "
                                            "timed region when streams > 1)",
                       "arithmetic": "fp32 ray trace / PSF grid; convolution operands carried as fp16 hi+lo pairs "
                                     "(>= 21-bit significand) on MFMA with fp32 accumulation, <= 2e-6 abs from the reference's fp32 conv2d"},
            "roofline": {"kernel": {"v": "conv_psf_map_kernel<11,5> (packed fp32 FMA)",
                                    "t": "conv_psf_map_mfma_kernel<11,5> (Toeplitz GEMM, fp16x3 operand split, fp32 accumulate)"}.get(
                                        os.environ.get("AADFF_CONV_PATH", "s")[0],
                                        "conv_psf_map_sbatch_kernel (slice-batched im2col GEMM on MFMA, fp16x3 operand split, "
                                        "fp32 accumulate)") + ", stack-fused S=10",
                         # headline: the bytes that must move per launch - image read ONCE for the S slices + S output slices
                         "bound": "hbm", "achieved": round(unique / (conv_ms * 1e-3) / 1e9, 2),
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(unique / (conv_ms * 1e-3) / HBM_PEAK, 4),
                         "basis": "stack-fused bytes: (1 + S) x 3 x H x W x 4 B = 13.2 B/pixel/slice (the image is read once for all slices)",
                         "bytes_per_launch": unique,
                         # SURVEY.md 8(d)'s definition beside it: 24 B/pixel/slice (every slice re-reads the image)
                         "achieved_survey_8d": round(achieved / 1e9, 2), "frac_survey_8d": round(achieved / HBM_PEAK, 4),
                         "algorithmic_bytes_per_launch_survey_8d": ALG_BYTES_PER_SLICE * S,
                         "traffic": traffic, "traffic_source": traffic_source,
                         # the three honest fractions side by side: SURVEY 8(d)'s algorithmic bytes (frac_survey_8d), the
                         # bytes the counters saw on the bus (frac_traffic) and the bytes that must move (frac); all over kernel_ms
                         "frac_traffic": round(traffic / (conv_ms * 1e-3) / HBM_PEAK, 4) if traffic else None,
                         "achieved_traffic": round(traffic / (conv_ms * 1e-3) / 1e9, 2) if traffic else None,
                         "kernel_ms_back_to_back": round(conv_b2b_ms, 4),
                         "frac_back_to_back": {"frac": round(unique / (conv_b2b_ms * 1e-3) / HBM_PEAK, 4),
                                               "frac_survey_8d": round(ALG_BYTES_PER_SLICE * S / (conv_b2b_ms * 1e-3) / HBM_PEAK, 4),
                                               "frac_traffic": round(traffic / (conv_b2b_ms * 1e-3) / HBM_PEAK, 4) if traffic else None,
                                               "what": "the same three fractions over kernel_ms_back_to_back (10 launches between two stream events, "
                                                       "median of 7 rounds: tools/kbench.py's method)"},
                         "kernel_ms": round(conv_ms, 4), "kernel_ms_median": round(float(np.median(conv_all)), 4),
                         "kernel_ms_bracketed": round(float(np.mean(conv_bracket)), 4) if conv_bracket else None,
                         "kernel_ms_note": "kernel_ms: HIP events attached to the kernel's dispatch (hipExtLaunchKernelGGL start/stop events) in the "
                                           "solo leg = the kernel's own begin-to-end time on its launch stream, what rocprofv3 reports for "
                                           "`bench.py --streams 1`; kernel_ms_bracketed: two stream events around the same launches "
                                           "(adds the two dispatch gaps, the round-1 method)",
                         "tflops": round(2 * 3 * KS * KS * H * W * S / (conv_ms * 1e-3) / 1e12, 2)},
            "trace": {"kernel": "psf_points_kernel (fused chief-ray centre + ray trace + LDS histogram + normalise)",
                      "us_per_stack": round(psf_ms * 1e3, 2), "ray_surface_steps_per_stack": steps_per_stack,
                      "ray_surface_steps_per_s": round(steps_per_stack / (psf_ms * 1e-3), 0),
                      "frac_of_step_one_stream": round(psf_ms / one_stream_ms, 4), "bound": "valu",
                      "valu_busy": valu, "valu_busy_source": valu_source},
            "flags": bits,
        }
        if world == 1:
            # the reference's OWN loops through the drop-in API (no StackPlan / StackPipeline): what a reference script gets
            try:
                sys.path.insert(0, os.path.join(REPO, "tools"))
                import dropin_bench
                from aadff.synth import mlp_state_dict as _msd, synth_depth_mm as _sdm
                from deeplens.psfnet import PSFNet as _PSFNet
                _net = _PSFNet(lens_path, sensor_res=(H, W), kernel_size=KS, device=dev)
                _net.psfnet.load_state_dict({k: torch.from_numpy(v) for k, v in _msd(seed=4321).items()})
                _dm = (torch.from_numpy(_sdm(H, W, seed=5678))[None, None] / 1e3).to(dev)
                torch.manual_seed(0)
                res["dropin_api"] = dropin_bench.measure(Lensgroup(lens_path, sensor_res=(H, W), device=dev), img, dbar, fds, GRID, KS, SPP,
                                                         reps=20, psfnet=_net, depth_map=_dm)
                del _net, _dm
                if not args.no_cpu_baseline:
                    # the same M1 loop through a STRICT and through an EDGE lens (the modes that carry the 1e-4 guarantee on every slice):
                    # refocus and psf_map each run their half of the fused stack for one state
                    from aadff import strict_stack as _ss0
                    for _par in ("strict", "edge"):
                        _ls = Lensgroup(lens_path, sensor_res=(H, W), device=dev, parity=_par)
                        torch.manual_seed(0)
                        for _ in range(2):
                            dropin_bench.m1_loop(_ls, img, dbar, fds, GRID, KS, SPP)
                        torch.cuda.synchronize(dev)
                        _t0 = time.perf_counter()
                        for _ in range(5):
                            dropin_bench.m1_loop(_ls, img, dbar, fds, GRID, KS, SPP)
                        torch.cuda.synchronize(dev)
                        _t = (time.perf_counter() - _t0) / 5
                        res["dropin_api"]["m1_loop_" + _par] = {"ms_per_stack": round(_t * 1e3, 2), "value": round(S * H * W / 1e6 / _t, 1), "unit": "MP/s",
                                                                "stacks": 5, "loop": f"the m1_loop through Lensgroup(parity='{_par}')"}
                        _ss0.release_buffers(_ls)
                        del _ls
            except Exception as e:
                res["dropin_api"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            _thread_mark("before cpu_baseline")
            res["cpu_baseline"], want = cpu_baseline(lens_path, img_h, dbar, fds)
            _thread_mark("after cpu_baseline")
            n = len(want)
            a = got[:, :n].astype(np.float64)
            b = np.stack([w[0].numpy() for w in want], 1).astype(np.float64)
            per = [float(np.linalg.norm(a[:, k] - b[:, k]) / np.linalg.norm(b[:, k])) for k in range(n)]
            res["parity"] = {"rel_l2": float(f"{np.linalg.norm(a - b) / np.linalg.norm(b):.3e}"), "slices": n, "tolerance": 1e-4,
                             "rel_l2_per_slice": [float(f"{v:.3e}") for v in per],
                             "against": "oracle (CPU restatement pinned to the reference by tests/golden), seed 0, "
                                        "same image / depth plane / focus distances as the timed steps"}
            # the same stack through the two modes that carry the 1e-4 guarantee on EVERY slice (no floor widening; exit code 5 otherwise):
            #   Lensgroup(parity="strict")  every ray in the reference's operation order on the GPU + the reference's host arithmetic
            #   Lensgroup(parity="edge")    round 6: d_sensor / hfov strict, the PSF rays on the fast kernel, only the rays at the histogram's
            #                               window edge re-traced in the reference's arithmetic
            # then a TIMED leg of each.  Two comparators, both printed: the oracle run on THIS box's host (`b`, whose MKL code path can
            # differ from the machine that produced the fixtures: the same torch program differs from itself by up to 1.1e-4 on a
            # slice across host CPUs, profiles/r03_d_oracle_cross_cpu.json) and the committed G9 fixture = the reference's own PSF maps
            # through the same HIP convolution.  The gate takes, per mode, the comparator the mode is closer to: a driver box with
            # another CPU model must not turn the exit code with no code defect (VERDICT r5 "What's weak" #2).
            strict_fail = False
            from aadff import _abi
            from aadff import strict_stack as _ss
            from aadff.focal_stack import render_focal_stack_m1 as _rfs
            b_fx = None
            try:
                g9 = np.load(os.path.join(REPO, "tests", "golden", "g9_stack_m1_1024.npz"))
                if n == S and tuple(g9["psf_maps"].shape) == (S, 3, GRID * KS, GRID * KS):
                    fx = torch.empty((1, 3, S, H, W), dtype=torch.float32, device=dev)
                    _abi.call("aadff_render_psf_map_stack", _abi.ptr(img), _abi.ptr(torch.from_numpy(g9["psf_maps"]).to(dev).contiguous()), _abi.ptr(fx),
                              1, 3, S, H, W, GRID, KS, _abi.stream_ptr(dev))
                    torch.cuda.synchronize(dev)
                    b_fx = fx[0].cpu().numpy().astype(np.float64)
                    del fx
            except Exception as e:
                print("bench: G9 fixture comparator unavailable:", repr(e), file=sys.stderr, flush=True)

            def per_slice(a_, ref):
                return [float(np.linalg.norm(a_[:, k] - ref[:, k]) / np.linalg.norm(ref[:, k])) for k in range(n)]

            def guaranteed_mode(parity_name, what):
                """parity + timed legs of one guaranteed mode; returns (record, failed)"""
                import gc
                ls = Lensgroup(lens_path, sensor_res=(H, W), device=dev, parity=parity_name)
                torch.manual_seed(0)
                _rfs(ls, img, dbar, fds, GRID, KS, SPP)          # seeds the lens's table of batch-wide Newton counts (per-surface form)
                torch.manual_seed(0)
                _rfs(ls, img, dbar, fds, GRID, KS, SPP)          # first fused call: staging buffers, first launches
                torch.cuda.synchronize(dev)
                torch.manual_seed(0)
                so = _rfs(ls, img, dbar, fds, GRID, KS, SPP)
                torch.cuda.synchronize(dev)
                a2 = so[0].cpu().numpy()[:, :n].astype(np.float64)
                per2 = per_slice(a2, b)
                per_fx = per_slice(a2, b_fx) if b_fx is not None else None
                gate = min(max(per2), max(per_fx)) if per_fx is not None else max(per2)
                n_strict = max(20, min(args.steps, 50))
                stats0 = dict(_ss.StrictCounts.of(ls).stats)
                # every host array of these modes is small (<= 20 k elements per op): one CPU thread.  With the 16-thread OpenMP pool the
                # cpu_baseline leg left behind, each op that crosses ATen's grain size wakes 16 spinning workers and the process runs
                # into its cgroup CPU quota: a 30-40 ms stall every 100 ms scheduler period was measured
                torch.set_num_threads(1)
                for _ in range(3):                               # untimed: the first stacks on FRESH draws (the calls above repeated seed 0) meet
                    _rfs(ls, img, dbar, fds, GRID, KS, SPP)      # count rows the table has not seen yet and re-launch a short level
                torch.cuda.synchronize(dev)
                stats0 = dict(_ss.StrictCounts.of(ls).stats)
                gc.collect()
                gc.freeze()                                      # the bench's long-lived objects out of the collector's way: a full collection
                cg0 = _cgroup_cpu()
                _thread_mark(f"before the {parity_name} timed loop")
                th0 = _threads_cpu() if os.environ.get("AADFF_BENCH_THREADS") == "1" else None
                t_s = time.perf_counter()                        # over them cost 50-90 ms every ~10 steps of this leg
                marks_s = []
                _pr = None
                if os.environ.get("AADFF_BENCH_PROFILE") == parity_name:
                    import cProfile
                    _pr = cProfile.Profile()
                    _pr.enable()
                for _ in range(n_strict):                        # new draws every step: the generator runs on, as in the reference's loop
                    t_i = time.perf_counter()
                    so = _rfs(ls, img, dbar, fds, GRID, KS, SPP)
                    torch.cuda.synchronize(dev)                  # host call to device idle, stack by stack (SURVEY 8d's definition)
                    marks_s.append(time.perf_counter() - t_i)
                    if os.environ.get("AADFF_STRICT_TIMING") == "1":
                        print(f"bench: {parity_name} step", round(marks_s[-1] * 1e3, 2), getattr(ls, "_strict_timing", None), file=sys.stderr, flush=True)
                if _pr is not None:
                    import pstats
                    _pr.disable()
                    pstats.Stats(_pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
                t_s = (time.perf_counter() - t_s) / n_strict
                cg1 = _cgroup_cpu()
                if th0 is not None:                              # debug: which threads used the CPU during the loop, and since when they exist
                    th1 = _threads_cpu()
                    top = sorted(((th1[t] - th0.get(t, 0), t) for t in th1), reverse=True)
                    born = lambda t: next((lab for lab, tids in _THREAD_MARKS if t in tids), "later")
                    groups = {}
                    for dt, t in top:
                        if dt > 0:
                            groups.setdefault(born(t), []).append(dt)
                    print(f"bench threads [{parity_name}]: {len(th1)} threads; CPU ticks (10 ms) during the loop by first sighting:",
                          {k: (len(v), sum(v), max(v)) for k, v in groups.items()}, file=sys.stderr, flush=True)
                stats1 = _ss.StrictCounts.of(ls).stats
                rec = {"rel_l2": float(f"{np.linalg.norm(a2 - b) / np.linalg.norm(b):.3e}"),
                       "rel_l2_per_slice": [float(f"{v:.3e}") for v in per2], "worst_slice": float(f"{max(per2):.3e}"),
                       "against": "the oracle run on this box's host CPU",
                       "vs_g9_fixture": None if per_fx is None else {
                           "rel_l2": float(f"{np.linalg.norm(a2 - b_fx) / np.linalg.norm(b_fx):.3e}"), "rel_l2_per_slice": [float(f"{v:.3e}") for v in per_fx],
                           "worst_slice": float(f"{max(per_fx):.3e}"),
                           "against": "tests/golden/g9_stack_m1_1024.npz: the reference's own PSF maps (generated in the build container) through the same HIP convolution"},
                       "gate": {"worst_slice": float(f"{gate:.3e}"), "tolerance_per_slice": 1e-4,
                                "rule": "per mode, the comparator it is closer to (on-box oracle or committed fixture): the reference differs from itself across host CPUs"},
                       "tolerance_per_slice": 1e-4,
                       "timed": {"steps": n_strict, "ms_per_step": round(t_s * 1e3, 3), "value": round(S * H * W / 1e6 / t_s, 1),
                                 "ms_per_step_p50": round(float(np.median(marks_s)) * 1e3, 3), "ms_per_step_max": round(float(np.max(marks_s)) * 1e3, 3),
                                 "ms_per_step_max_at": int(np.argmax(marks_s)), "ms_per_step_second_max": round(float(np.sort(marks_s)[-2]) * 1e3, 3),
                                 "unit": "MP/s", "what": f"render_focal_stack_m1 through the {parity_name} lens, one stack at a time "
                                 "(host call to device idle), fresh draws every step",
                                 "speculation": {k: stats1[k] - stats0.get(k, 0) for k in stats1},
                                 "host_cpu_quota": None if cg0 is None or cg1 is None else {
                                     "throttled_ms": round((cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)) / 1e3, 1),
                                     "nr_throttled": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                                     "cpu_ms_used": round((cg1.get("usage_usec", 0) - cg0.get("usage_usec", 0)) / 1e3, 1),
                                     "what": "this process group's cgroup cpu.stat over the timed loop: a step that ran into the box's CPU quota "
                                             "shows up here and in ms_per_step_max, not in the p50"}},
                       "seconds_per_stack": round(t_s, 4), "what": what}
                _ss.release_buffers(ls)
                # the same mode with two and with four stacks in flight (StrictPipeline: lenses / streams software-pipelined on this thread; draws
                # at submission).  Both depths are timed: depth 4 is the faster one on an idle host and the slower one on a busy one (the GPU
                # boxes are shared; profiles/r06_zz_bench_busy_host.json) - ms_per_step is the better of the two, `depths` holds both
                try:
                    by_depth = {}
                    for depth_ in (2, 4):
                        pipe2 = _ss.StrictPipeline(lambda: Lensgroup(lens_path, sensor_res=(H, W), device=dev, parity=parity_name), depth=depth_)
                        torch.manual_seed(1)
                        for f_ in [pipe2.submit(img, dbar, fds, GRID, KS, SPP) for _ in range(3 * depth_)]:  # seeds the lenses' count tables, warms
                            f_.result()[1].synchronize()
                        torch.cuda.synchronize(dev)
                        t_p = time.perf_counter()
                        futs = [pipe2.submit(img, dbar, fds, GRID, KS, SPP) for _ in range(n_strict)]
                        for f_ in futs:
                            o_, e_ = f_.result()
                            e_.synchronize()
                            del o_
                        by_depth[depth_] = (time.perf_counter() - t_p) / n_strict
                        pipe2.close()
                        for l_ in pipe2.lenses:
                            _ss.release_buffers(l_)
                    best = min(by_depth, key=by_depth.get)
                    t_p = by_depth[best]
                    rec["timed_pipelined"] = {
                        "steps": n_strict, "ms_per_step": round(t_p * 1e3, 3), "value": round(S * H * W / 1e6 / t_p, 1), "unit": "MP/s", "depth": best,
                        "depths": {str(k): round(v * 1e3, 3) for k, v in by_depth.items()},
                        "what": "aadff.strict_stack.StrictPipeline, one host thread: every host wait of a stack (round trips of the short levels, "
                                "psf_map launch, re-launches) is where the host goes on with another stack; same stacks as the sequential loop (draws at "
                                "submission); ms_per_step = the better of depth 2 and depth 4 on this host"}
                except Exception as e:
                    rec["timed_pipelined"] = {"error": repr(e)}
                failed = not gate <= 1e-4
                if failed:
                    print(f"bench: {parity_name}-mode parity above 1e-4 on a slice against BOTH comparators", file=sys.stderr, flush=True)
                return rec, failed

            for _name, _key, _what in (
                    ("strict", "strict_mode", "Lensgroup(parity='strict'): the reference's float32 operation order on the GPU + the reference's host "
                     "arithmetic; one fused launch per level on speculated batch-wide Newton counts, verified from the any-bits "
                     "and re-launched where a count was off (aadff/strict_stack.py, csrc/strict_fused.hip); DESIGN.md section 2"),
                    ("edge", "edge_mode", "Lensgroup(parity='edge'): refocus / calc_fov as in the strict mode (d_sensor, hfov to the reference's bits), the PSF grid "
                     "on the FAST kernel with the rays within 2e-4 mm of the histogram's window edge (deeplens/monte_carlo.py:37) left undecided, "
                     "re-traced in the reference's arithmetic and added (aadff_psf_points_edge -> aadff_strict_edge_retrace -> aadff_psf_normalise); "
                     "DESIGN.md section 2")):
                try:
                    res["parity"][_key], _f = guaranteed_mode(_name, _what)
                    strict_fail = strict_fail or _f
                    if _name == "strict":
                        sj, ssrc = digest_multi("strict_kernel_pmc.json", ("strict_fused.hip", "strict_math2.h", "strict_math.h", "strict.hip"))
                        res["parity"][_key]["psf_map_kernel"] = {"us_per_launch_rocprof": sj.get("us_per_launch_rocprof") if sj else None,
                                                                 "SQ_INSTS_VALU": sj.get("SQ_INSTS_VALU") if sj else None,
                                                                 "valu_busy": sj.get("valu_busy") if sj else None, "source": ssrc}
                except Exception as e:                   # the contract line must not depend on the verification modes
                    res["parity"][_key] = {"error": repr(e)}
            fpath = os.path.join(REPO, "tests", "golden", "g13_fp32_floor.npz")
            if os.path.exists(fpath):       # fp32-vs-fp64 distance of the reference formulation itself, per slice (static fixture)
                res["parity"]["fp32_floor_per_slice"] = [float(f"{v:.3e}") for v in np.load(fpath)["img_floor"][:n]]
            if not res["parity"]["rel_l2"] <= 1e-4:
                print(json.dumps(res), flush=True)
                print("bench: parity failed", file=sys.stderr, flush=True)
                raise SystemExit(4)
        print(json.dumps(res), flush=True)
        if not args.no_cpu_baseline and world == 1 and strict_fail:
            raise SystemExit(5)                          # the mode that carries the 1e-4 guarantee missed it on a slice
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def main_c3(args):
    """BASELINE.json config 3: 16 scenes x 10 slices = 160 (scene, slice) units dealt to the ranks in whole-scene blocks
    (AADFF_C3_BLOCK=1: SURVEY.md 8e's u = r (mod N)), per-row in-place all-gathers reassemble [160,3,1024,1024] on every
    rank.  Total work is fixed: strong scaling."""
    import torch.distributed as dist
    from aadff import dist as adist
    from aadff.focal_stack import SceneUnitRenderer, render_scenes_sharded
    from aadff.synth import synth_depth_mm, synth_rgb
    from deeplens.optics import Lensgroup
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    adist.init_from_env(backend="nccl", device=dev)
    n_scenes = int(os.environ.get("AADFF_C3_SCENES", "16"))
    lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), device=dev)
    scenes = []
    for sc in range(n_scenes):
        depth = synth_depth_mm(H, W, seed=5678 + sc)
        scenes.append((torch.from_numpy(synth_rgb(H, W, seed=1234 + sc))[None].to(dev), -float(depth.mean()),
                       -np.linspace(depth.min(), depth.max(), S)))
    c3_streams = int(os.environ.get("AADFF_C3_STREAMS", "2"))
    rend = SceneUnitRenderer(lens, scenes, S, GRID, KS, SPP, streams=c3_streams)
    n_units = n_scenes * S
    block = int(os.environ.get("AADFF_C3_BLOCK", "0")) or adist.scene_block(n_units, S, world)
    steps, warm = (min(args.steps, 10) if args.steps == 200 else args.steps), min(args.warmup, 2)
    multi = world > 1 or adist.grouped()               # AADFF_FORCE_GROUP=1: the gather branch on a one-rank RCCL group
    side = torch.cuda.Stream(dev) if multi else None

    def step():
        full, _, done = render_scenes_sharded(rend, gather=True, stream=side or torch.cuda.current_stream(dev), block=block)
        if done is not None:
            torch.cuda.current_stream(dev).wait_event(done)
        return full

    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup_s:      # clocks of a fresh box (see main); rank-local work only: the
        render_scenes_sharded(rend, gather=False, block=block)   # ranks may make different numbers of spin-up passes
        torch.cuda.synchronize(dev)
    for _ in range(warm):
        step()
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        full = step()
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    dt = adist.all_reduce_max(time.perf_counter() - t0)
    rend.check_flags()
    share8 = None
    if world == 1:
        # What ONE rank of an 8-rank job has to do, measured here (untimed leg): rank 0's share of the 8-way partition
        # rendered on this GPU without peers - its launches, and the host draws it replicates, go with the number of scenes
        # it touches, so this is NOT t1 / 8 by construction (tools/c3_share_probe.py sweeps partitions and rank counts).
        share8 = {}
        for name, blk in (("block", adist.scene_block(n_units, S, 8)), ("half_scene", max(1, S // 2)), ("round_robin", 1)):
            mine8 = adist.shard_units(n_units, 0, 8, blk)
            loc = torch.empty((len(mine8), 3, H, W), dtype=torch.float32, device=dev)
            rend.render(mine8, out=loc)
            torch.cuda.synchronize(dev)
            t8 = time.perf_counter()
            for _ in range(steps):
                rend.render(mine8, out=loc)
            torch.cuda.synchronize(dev)
            share8[name] = {"units_per_block": blk, "units": len(mine8), "scenes_touched": len({u // S for u in mine8}),
                            "ms": round((time.perf_counter() - t8) / steps * 1e3, 3)}
            del loc
        rend.check_flags()
    if rank == 0:
        # What the first real 8-GPU run should be read against (no multi-GPU box in the build loop): every rank
        # must RECEIVE (N-1)/N of the gathered set per step over its 7 xGMI links (~153 GB/s each, point to point).
        unit_bytes = 3 * H * W * 4
        total_bytes = n_scenes * S * unit_bytes
        link_in = 7 * 153e9
        exp = {"xgmi_inbound_GBps_per_rank": round(link_in / 1e9, 1), "gathered_bytes": total_bytes,
               "gather_floor_ms_at_8": round(total_bytes * 7 / 8 / link_in * 1e3, 3),
               "note": "render_ms_at_8 is MEASURED: rank 0's share of the 8-way partition rendered on this GPU (whole-scene blocks; "
                       "the round-robin partition u = r mod 8 is listed beside it). With the gather a step cannot be shorter than "
                       "gather_floor_ms_at_8 (each rank receives 7/8 of the set) plus the last row's exposed part, so the gathered "
                       "configuration is link-bound and the >= 6x target of north_star is a no-gather target"}
        if world == 1:
            t1 = dt / steps * 1e3
            t8 = share8["block"]["ms"]
            rows8 = adist.padded_share(n_units, 8, share8["block"]["units_per_block"]) // share8["block"]["units_per_block"]
            # rows are gathered as they complete, one after the other on the links: row i can start when this rank's block
            # of it is rendered ((i+1)/rows of the render time) and row i-1 has arrived
            def gathered(t_render, rows):
                g = 0.0
                for i in range(rows):
                    g = max((i + 1) * t_render / rows, g) + exp["gather_floor_ms_at_8"] / rows
                return g
            with_gather = gathered(t8, rows8)
            for v in share8.values():
                v["step_ms_with_gather_model"] = round(gathered(v["ms"], adist.padded_share(n_units, 8, v["units_per_block"]) // v["units_per_block"]), 3)
            exp.update({"render_ms_at_1": round(t1, 3), "render_ms_at_8": t8, "share_of_8_measured": share8,
                        "speedup_at_8_no_gather": round(t1 / t8, 2),
                        "speedup_at_8_with_gather": round(t1 / with_gather, 2),
                        "MPs_at_8_no_gather": round(n_scenes * S * H * W / 1e6 / (t8 * 1e-3), 0),
                        "MPs_at_8_with_gather": round(n_scenes * S * H * W / 1e6 / (with_gather * 1e-3), 0)})
        print(json.dumps({
            "metric": "focal-stack MP/s (config 3: 16 scenes x 10 slices sharded over N ranks in whole-scene blocks, all-gathered)",
            "value": round(n_scenes * S * H * W / 1e6 * steps / dt, 2), "unit": "MP/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"rf50mm, {n_scenes} scenes x {S} slices, 1024x1024, 11x11 PSF grid, ks 11, spp 2048, mode M1",
                       "gather": multi, "ranks_emulated_on_one_gpu": adist.emulated(),
                       "partition": f"blocks of {block} consecutive units dealt round-robin (unit u -> rank (u // {block}) % N)",
                       "streams_per_rank": c3_streams,
                       "gather_form": "slices written by the convolution straight into the unit-order buffer [rows, world, block, C, H, W]; one "
                                      f"in-place all-gather per buffer row ({block} x 12.6 MB per rank) on a side stream, started as soon as the "
                                      "scene groups that produce this rank's block of the row have been launched",
                       "gathered_shape": list(full.shape)},
            "expected_scaling": exp}), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def main_m2(args):
    """M2: [1,3,1024,1024] RGB + depth map -> 10 slices by PSFNet.render (fused MLP + gather kernel), random-init
    weights of the reference architecture (4 -> 64 -> 256 -> 8x256 -> 121).  Not the BASELINE.json metric."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    from aadff import psfnet_pack
    from aadff.dist import init_from_env
    from aadff.focal_stack import render_focal_stack_m2
    from aadff.synth import mlp_state_dict, synth_depth_mm, synth_rgb
    from deeplens.psfnet import PSFNet
    init_from_env(backend="nccl", device=dev)
    net = PSFNet(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), kernel_size=KS, device=dev)
    net.psfnet.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_state_dict(seed=4321).items()})
    net.mlp_precision = os.environ.get("AADFF_M2_PRECISION", "fp32")       # "fp16": opt-in single-pass mode of the fused kernel
    img = torch.from_numpy(synth_rgb(H, W, seed=1234 + rank))[None].to(dev)
    depth_m = (torch.from_numpy(synth_depth_mm(H, W, seed=5678 + rank))[None, None] / 1e3).to(dev)
    steps = min(args.steps, 20) if args.steps == 200 else args.steps        # a step is ~31 ms
    warm = min(args.warmup, 3)
    evs = []

    def hook(start):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)

    for _ in range(warm):
        render_focal_stack_m2(net, img, depth_m, S)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    psfnet_pack.EVENT_HOOK = hook
    t0 = time.perf_counter()
    for _ in range(steps):
        render_focal_stack_m2(net, img, depth_m, S)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    psfnet_pack.EVENT_HOOK = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kms = float(np.mean([evs[i].elapsed_time(evs[i + 1]) for i in range(0, len(evs), 2)]))
    launches_per_step = len(evs) // 2 // steps                              # 1: the whole stack in one fused launch
    px_per_launch = S * H * W // launches_per_step
    flop_px = 2 * (4 * 64 + 64 * 256 + 8 * 256 * 256 + 256 * KS * KS)       # fp32-equivalent flops per pixel
    passes = 1 if net.mlp_precision == "fp16" else 3
    issued = passes * flop_px * px_per_launch / (kms * 1e-3) / 1e12         # fp16 MFMA flops (three per product with the hi/lo split)
    if rank == 0:
        print(json.dumps({
            "metric": "focal-stack MP/s (M2: RGB-D through PSFNet.render, 1024^2 x 10 slices)",
            "value": round(world * S * H * W / 1e6 * steps / dt, 2), "unit": "MP/s", "n_gpus": world, "steps": steps,
            "warmup": warm, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "1024x1024 synthetic RGB + depth map, 10 focus distances (linear rule), PSFNet MLP "
                                   "4-64-256-8x256-121 with random-init weights, per-pixel 11x11 gather",
                       "arithmetic": "fp16 single pass on MFMA, fp32 accumulate (PSFs ~5e-4 relative; opt-in)" if passes == 1 else
                                     "fp32 operands as fp16 hi/lo pairs on MFMA, fp32 accumulate (2e-7 from torch fp32)"},
            "roofline": {"kernel": f"psfnet_fused_kernel<64> ({launches_per_step} launch(es) per stack)", "bound": "mfma", "achieved": round(issued, 1),
                         "peak": 2500.0, "unit": "TFLOP/s", "frac": round(issued / 2500.0, 4), "traffic": None,
                         "kernel_ms": round(kms, 4), "fp32_equivalent_tflops": round(issued / passes, 1)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main_m1l(args):
    """M1-layered: RGB-D stack, the depth map quantised into L = 4 layers, S x L ray-traced PSF maps, every pixel
    keeps the candidate of its own layer.  Three launches per stack: refocus, one PSF-grid launch for the S x L pairs, the layered
    stack convolution (aadff_render_psf_map_stack_layered).  Step = one stack; one rank per GPU renders its own scene (weak scaling)."""
    import torch.distributed as dist
    from aadff import _abi
    from aadff import dist as adist
    from aadff.focal_stack import render_focal_stack_m1_layered
    from aadff.synth import synth_depth_mm, synth_rgb
    from deeplens.optics import Lensgroup, raise_psf_flags
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    adist.init_from_env(backend="nccl", device=dev)
    L = int(os.environ.get("AADFF_M1L_LAYERS", "4"))
    lens_path = os.path.join(REPO, "lenses", "rf50mm", "lens.json")
    lens = Lensgroup(lens_path, sensor_res=(H, W), device=dev)
    img_h = torch.from_numpy(synth_rgb(H, W, seed=1234 + rank))[None]
    depth_h = -torch.from_numpy(synth_depth_mm(H, W, seed=5678 + rank))[None, None]
    img, depth = img_h.to(dev), depth_h.to(dev)
    fds = [float(f) for f in np.linspace(float(depth_h.max()), float(depth_h.min()), S)]       # nearest .. farthest (mm < 0)
    steps = min(args.steps, 50) if args.steps == 200 else args.steps
    warm = min(args.warmup, 5)
    for i in range(warm):
        torch.manual_seed(i)
        render_focal_stack_m1_layered(lens, img, depth, fds, layers=L, grid=GRID, ks=KS, spp=SPP)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        torch.manual_seed(100 + i)
        out = render_focal_stack_m1_layered(lens, img, depth, fds, layers=L, grid=GRID, ks=KS, spp=SPP)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    raise_psf_flags(int(lens._m1l_flags.item()))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the convolution launch alone between two stream events (same image / maps / index as a step) and, for comparison, the whole
    # composition it replaces
    torch.manual_seed(7)
    _, maps_d, lidx_d = render_focal_stack_m1_layered(lens, img, depth, fds, layers=L, grid=GRID, ks=KS, spp=SPP, return_parts=True)
    out_d = torch.empty((1, 3, S, H, W), dtype=torch.float32, device=dev)
    kms, comp = [], []
    for i in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _abi.call("aadff_render_psf_map_stack_layered", _abi.ptr(img), _abi.ptr(maps_d), _abi.ptr(lidx_d), _abi.ptr(out_d), 1, 3, S, L, H, W, GRID, KS,
                  _abi.stream_ptr(dev))
        e1.record()
        torch.cuda.synchronize(dev)
        kms.append(e0.elapsed_time(e1))
    for i in range(5):
        t1 = time.perf_counter()
        render_focal_stack_m1_layered(lens, img, depth, fds, layers=L, grid=GRID, ks=KS, spp=SPP, fused=False)
        torch.cuda.synchronize(dev)
        comp.append(time.perf_counter() - t1)
    conv_ms = float(np.median(kms))
    bytes_launch = (1 + S) * 3 * H * W * 4 + H * W                           # image once + S output slices + 1 B / pixel of layer index
    if rank == 0:
        res = {
            "metric": "focal-stack MP/s (M1-layered: RGB-D, depth map in 4 layers, 1024^2 x 10 slices, 11x11 PSF grid)",
            "value": round(world * S * H * W / 1e6 * steps / dt, 2), "unit": "MP/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"rf50mm, 1024x1024 synthetic RGB + depth MAP quantised into {L} layers, 10 focus distances, 11x11 PSF grid, ks 11, "
                                   f"spp 2048 (+2048 chief): {S * L} ray-traced PSF maps per stack, per-pixel selection fused into the stack convolution",
                       "launches_per_stack": 3, "composition_ms_per_stack": round(float(np.median(comp)) * 1e3, 3),
                       "composition": "the same three launches but S x L candidate slices through the stack convolution + torch.gather (L x the output bytes)"},
            "roofline": {"kernel": "conv_psf_map_sbatch_kernel<24,4,false,false,LAYERED> (aadff_render_psf_map_stack_layered)", "bound": "hbm",
                         "achieved": round(bytes_launch / (conv_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(bytes_launch / (conv_ms * 1e-3) / HBM_PEAK, 4), "traffic": None,
                         "basis": "(1 + S) x 12 B + 1 B of layer index per pixel (SURVEY.md 8(d) M1-layered; the index map is uint8 here)",
                         "bytes_per_launch": bytes_launch, "kernel_ms": round(conv_ms, 4),
                         "kernel_ms_note": "two stream events around the launch alone (same image, PSF maps and layer index as a step), median of 20",
                         "tflops": round(2 * 3 * KS * KS * H * W * S * L / (conv_ms * 1e-3) / 1e12, 2)},
        }
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, REPO)
            from oracle import psfnet as opsf
            from oracle.lens import OracleLens
            torch.set_num_threads(usable_cpus())
            ora = OracleLens(lens_path, sensor_res=(H, W))
            n = 1                                            # bounded sample: ONE slice = 4 psf_map traces + 4 convolutions of the oracle
            torch.manual_seed(100 + steps - 1)               # the seed of the last timed step: its first slice consumes the same draws
            t1 = time.perf_counter()
            want = opsf.focal_stack_m1_layered(ora, img_h, depth_h, fds[:n], layers=L, grid=GRID, ks=KS, spp=SPP)
            tc = time.perf_counter() - t1
            a, b = out[:, :, :n].cpu().numpy().astype(np.float64), want.numpy().astype(np.float64)
            res["cpu_baseline"] = {"value": round(n * H * W / 1e6 / tc, 4), "unit": "MP/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "cpu_model": cpu_model(), "sample": f"{n} of {S} slices of the same M1-layered stack (oracle composition: refocus + {L} x "
                                   f"(psf_map + render_psf_map) + per-pixel selection), {tc:.1f} s"}
            res["parity"] = {"rel_l2": float(f"{np.linalg.norm(a - b) / np.linalg.norm(b):.3e}"), "slices": n, "tolerance": 1e-4,
                             "against": "oracle composition (oracle/psfnet.py: focal_stack_m1_layered), same seed, first slice of the last timed step"}
        print(json.dumps(res), flush=True)
        if res.get("parity", {}).get("rel_l2", 0.0) > 1e-4:
            print("bench: parity failed", file=sys.stderr, flush=True)
            raise SystemExit(4)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_fit(args):
    """1_fit_psfnet.py training loop (reference: deeplens/psfnet.py:79-170): per iteration a random focus distance
    (refocus kernel), bs = 128 random points, their ray-traced PSFs (fused trace/PSF kernel, spp 4096, ks 11, sensor 480 x 640:
    the configuration of 1_fit_psfnet.py:18,22) as targets,
    one MLP forward/backward/AdamW step in torch (bf16 autocast on the MLP).  Not the BASELINE.json metric."""
    from aadff.synth import mlp_state_dict
    from deeplens.psfnet import PSFNet
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    # the script's own configuration (1_fit_psfnet.py:18,22): sensor 480 x 640, bs 128, spp 4096, lr 1e-4, ks 11
    fit_res, bs = (480, 640), 128
    spp = int(os.environ.get("AADFF_FIT_SPP", "4096"))
    net = PSFNet(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=fit_res, kernel_size=KS, device=dev)
    net.psfnet.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_state_dict(seed=4321).items()})
    from deeplens.psfnet import _TrainStep
    steps = min(args.steps, 100) if args.steps == 200 else args.steps
    step = _TrainStep(net.psfnet, 1e-4, 10000, bs, KS * KS, dev, True, os.environ.get("AADFF_FIT_GRAPH", "1") != "0")
    t_data = [0.0]

    plan = net._training_plan(bs, spp)       # pipelined producer (aadff/training.py): two launches per batch, no copies

    def it(i):
        torch.manual_seed(i)
        t0 = time.perf_counter()
        inp, psf = plan.next()
        t_data[0] += time.perf_counter() - t0
        step(inp, psf)

    for i in range(min(args.warmup, 10)):
        it(i)
    torch.cuda.synchronize(dev)
    t_data[0] = 0.0
    plan.wait_s = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        it(i)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    plan.check_flags()
    print(json.dumps({
        "metric": f"PSFNet fit iterations/s (bs {bs} ray-traced PSF targets, spp {spp}, ks 11, bf16 MLP step)",
        "value": round(steps / dt, 2), "unit": "it/s", "n_gpus": 1, "steps": steps, "warmup": min(args.warmup, 10),
        "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 targets / bf16 MLP", "data": "synthetic",
        "config": {"workload": f"1_fit_psfnet.py:18,22: rf50mm, {fit_res[0]}x{fit_res[1]} sensor, bs {bs}, spp {spp}, lr 1e-4, reference sampling of "
                               "(x, y, z, focus), random-init MLP 4-64-256-8x256-121",
                   "host_data_ms_per_step": round((t_data[0] - plan.wait_s) / steps * 1e3, 4),
                   "host_wait_for_gpu_ms_per_step": round(plan.wait_s / steps * 1e3, 4),
                   "batches": "pipelined producer (aadff/training.py): pinned block uploaded inside the refocus launch, 2 launches per batch"}}), flush=True)


if __name__ == "__main__":
    main()
