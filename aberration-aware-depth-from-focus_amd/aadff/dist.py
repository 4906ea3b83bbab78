"""One process per GPU: shard (scene, slice) units across ranks (blocks of consecutive units dealt round-robin) and,
when a single consumer needs the whole result, reassemble it with one in-place all-gather per buffer row (RCCL over
xGMI on the GPU box, gloo in the CPU tests and in the one-GPU rank emulation).  The path has no other exchange step: units are
independent.

Launching: `spawn_ranks` starts N copies of a script BEFORE anything has touched the GPU (the parent
never makes a HIP call; a child owns its device).  `--emulate-ranks` puts all N ranks on GPU 0 with
the gloo backend, because RCCL refuses two ranks on one device: the launcher, the
sharding and the gather order are then exercised on a one-GPU box; xGMI is not.
"""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(argv, world, emulate=False, env=None, timeout=None):
    """Run `python argv...` as `world` rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and wait.
    Returns the largest exit code (the first failing rank's when one fails: its peers are then torn down; 124 when
    `timeout` seconds - one deadline for the whole job - expire).  The caller must not have initialised the GPU (no HIP call, no
    torch.cuda.is_available()): on this pool a GPU-initialised parent must not start GPU children by exec, and a
    parent that holds a context would also take memory on device 0."""
    port = free_port()
    procs = []
    for r in range(world):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0" if emulate else str(r),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this host driver
        if emulate:
            e["AADFF_EMULATE_RANKS"] = "1"
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
    # One poll loop over ALL ranks with ONE overall deadline (torchrun's behaviour): the first rank that exits non-zero,
    # or the deadline, ends the job - the remaining ranks (which would otherwise sit in init_process_group or in a
    # collective, holding their GPUs until the backend's own 10-30 min timeout) are terminated, then killed.
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    rc = max(rc, abs(code))
            if live and rc == 0:
                if deadline is not None and time.monotonic() > deadline:
                    rc = 124
                    break
                time.sleep(0.02)
    finally:
        left = [p for p in procs if p.poll() is None]
        for p in left:
            p.terminate()                # exact PIDs we started
        t_kill = time.monotonic() + 5.0
        for p in left:
            try:
                p.wait(timeout=max(0.0, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def emulated():
    return os.environ.get("AADFF_EMULATE_RANKS", "0") == "1"


def forced_group():
    """AADFF_FORCE_GROUP=1: a process group is formed even for ONE rank, and the sharded renderers / bench take their
    collective branch through it.  A one-rank RCCL group on a one-GPU box executes everything that only exists under RCCL -
    `init_process_group("nccl", device_id=...)`, the IN-PLACE row gather whose send buffer aliases the receive buffer,
    all_reduce on a device tensor, barrier, teardown - so the first real multi-GPU launch cannot fail on something one GPU
    would have shown (tests/test_gpu_dist.py::test_rccl_branch_on_a_one_rank_group)."""
    return os.environ.get("AADFF_FORCE_GROUP", "0") == "1"


def grouped():
    """True when collectives are to be issued: more than one rank, or a forced one-rank group."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced_group())


def _cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_cpus(local_rank, sysfs="/sys"):
    """(pci address, NUMA node, CPUs of that node) of the local_rank-th AMD GPU in PCI-bus order - the order HIP enumerates in -
    read from sysfs WITHOUT touching the GPU; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES index lists are honoured.  None when the
    tree does not say (no amdgpu devices, NUMA node -1, ...)."""
    # Pin only when the mapping local_rank -> PCI device is unambiguous: CUDA_VISIBLE_DEVICES (HIP honours it too), UUID or other
    # non-numeric visible-device lists and emulated ranks (every LOCAL_RANK on GPU 0) are not decoded here, and a rank pinned to
    # the WRONG node is worse off than an unpinned one.
    if emulated() or os.environ.get("CUDA_VISIBLE_DEVICES", "").strip():
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        vis = os.environ.get(var, "").strip()
        if vis and not all(v.strip().isdigit() for v in vis.split(",")):
            return None
    base = os.path.join(sysfs, "bus", "pci", "drivers", "amdgpu")
    try:
        addrs = sorted(a for a in os.listdir(base) if a.count(":") == 2)
    except OSError:
        return None
    gpus = []
    for a in addrs:
        try:
            cls = open(os.path.join(base, a, "class")).read().strip()
        except OSError:
            cls = ""
        if cls.startswith("0x03") or cls.startswith("0x12") or not cls:      # display controllers / processing accelerators
            gpus.append(a)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        vis = os.environ.get(var, "").strip()
        if vis and all(v.strip().isdigit() for v in vis.split(",")):
            idx = [int(v) for v in vis.split(",")]
            if all(i < len(gpus) for i in idx):
                gpus = [gpus[i] for i in idx]
    if not 0 <= local_rank < len(gpus):
        return None
    addr = gpus[local_rank]
    try:
        node = int(open(os.path.join(base, addr, "numa_node")).read().strip())
        if node < 0:
            return addr, node, set()
        cpus = _cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read())
    except (OSError, ValueError):
        return None
    return addr, node, cpus


def pin_to_gpu_numa(local_rank=None, sysfs="/sys", announce=True):
    """Restrict this process (every thread it starts later) to the CPUs of the NUMA node its GPU hangs off - BEFORE the first GPU
    call, so that the runtime's helper threads and the pinned staging blocks land there too: a rank spends ~0.3 ms of host work per
    stack and its kernels read pinned memory over PCIe.  Leaves the affinity alone when sysfs does not name a node or the node's
    CPUs are outside the cgroup's set.  One placement line per rank goes to stderr.  AADFF_NUMA_PIN=0 switches it off."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else int(local_rank)
    rank = os.environ.get("RANK", "0")
    line = f"aadff placement: rank {rank} (local {local_rank}) pid {os.getpid()}"
    try:
        have = os.sched_getaffinity(0)
        info = None if os.environ.get("AADFF_NUMA_PIN", "1") == "0" else gpu_numa_cpus(local_rank, sysfs)
        if info is None:
            line += f": no NUMA information, affinity unchanged ({len(have)} CPUs)"
        else:
            addr, node, cpus = info
            use = cpus & have
            if node < 0 or not use:
                line += f": GPU {addr} NUMA node {node}, affinity unchanged ({len(have)} CPUs)"
            else:
                os.sched_setaffinity(0, use)
                line += f": GPU {addr} on NUMA node {node}, pinned to {len(use)} of its {len(cpus)} CPUs ({min(use)}..{max(use)})"
    except (OSError, AttributeError, ValueError) as e:      # never a reason to fail a run
        line += f": affinity unchanged ({e!r})"
    if announce:
        print(line, file=sys.stderr, flush=True)
    return line


INIT_EXIT_CODE = 17


def init_from_env(backend=None, device=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.
    Returns (rank, world).  Single-process runs need no group (AADFF_FORCE_GROUP=1 forms one all the same: `forced_group`).
    Ranks emulated on one GPU use gloo."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or forced_group()) and not dist.is_initialized():
        if emulated():
            backend = "gloo"
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        # A rendezvous that does not complete (a peer that never started, a wrong MASTER_ADDR, RCCL stuck bringing a link up) must
        # end the job with a message and a non-zero status, not hang the launcher: a watchdog thread bounds the BRING-UP (TCP
        # rendezvous + first communicator) and nothing else.  No `timeout=` goes to init_process_group: in torch that value is the
        # default timeout of EVERY later collective on the group (a rank-0-only CPU baseline or checkpoint that keeps the peers in
        # a barrier for more than AADFF_INIT_TIMEOUT_S would be aborted by the backend's own watchdog); the backend defaults
        # (10 min RCCL, 30 min gloo) stay.  The process EXITS (a fresh child is the launcher's business); it never re-executes
        # itself - it may already have touched the GPU.
        import threading
        limit = float(os.environ.get("AADFF_INIT_TIMEOUT_S", "120"))
        done = threading.Event()

        def watchdog():
            if not done.wait(limit):
                print(f"aadff: rank {rank} of {world}: process group ({backend}) not up after {limit:.0f} s "
                      f"(MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')}); giving up", file=sys.stderr, flush=True)
                os._exit(INIT_EXIT_CODE)

        threading.Thread(target=watchdog, daemon=True, name="aadff-init-watchdog").start()
        try:
            _init_group(backend, rank, world, device, kw)
        except Exception as e:
            print(f"aadff: rank {rank} of {world}: init_process_group({backend}) failed: {e!r} "
                  f"(MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')})", file=sys.stderr, flush=True)
            done.set()
            raise SystemExit(INIT_EXIT_CODE)
        done.set()
    return rank, world


def _init_group(backend, rank, world, device, kw):
    if backend == "nccl":
        # RCCL writes a version banner ("RCCL version : ...", five lines) to STDOUT through C stdio when its first communicator
        # comes up - behind whatever Python has printed by then, since C stdio is flushed at exit.  bench.py's contract is ONE
        # JSON line on stdout: bring the communicator up here with fd 1 pointing at stderr, flush C stdio, restore.
        import ctypes
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
            t = torch.zeros(1, device=device if device is not None else "cuda")
            dist.all_reduce(t)
            torch.cuda.synchronize()
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(keep, 1)
            os.close(keep)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)


def shard_units(n_units, rank, world, block=1):
    """Blocks of `block` consecutive units dealt round-robin: unit u belongs to rank (u // block) % world.  block = 1 is
    SURVEY.md 8e's u = r (mod world); block = S hands out whole scenes (`scene_block`)."""
    block = int(block)
    nblocks = (n_units + block - 1) // block
    return [u for b in range(rank, nblocks, world) for u in range(b * block, min(n_units, (b + 1) * block))]


def padded_share(n_units, world, block=1):
    """Units per rank after padding to equal shares of whole blocks (all_gather needs equal shapes)."""
    nblocks = (n_units + block - 1) // block
    return (nblocks + world - 1) // world * block


def scene_block(n_units, S, world):
    """Block size for (scene, slice) units: the largest divisor of S (blocks must not straddle scenes) that still gives
    every rank a block.  A rank's launches, and the host draws it replicates per scene, go with the number of SCENES it
    touches: with u = r (mod 8) a rank of config 3 touches all 16 scenes for 1-2 slices each and measures 3.3x slower than
    its share of the work (tools/c3_share_probe.py); with block = S = 10 it renders 2 whole scenes at the one-GPU rate."""
    best = 1
    for d in range(1, S + 1):
        if S % d == 0 and d * world <= max(n_units, world):
            best = d
    return best


def _host_backend():
    return dist.get_backend() == "gloo"


def all_gather_into(full, local):
    """dist.all_gather_into_tensor on flat views (gloo only accepts the concatenated form) that also works for
    device tensors under gloo (staged through the host; only the emulation and the CPU tests take that road)."""
    assert full.is_contiguous() and local.is_contiguous() and full.numel() == local.numel() * dist.get_world_size()
    if local.is_cuda and _host_backend():
        h = torch.empty(full.numel(), dtype=full.dtype)
        dist.all_gather_into_tensor(h, local.reshape(-1).cpu())
        full.view(-1).copy_(h)
    else:
        dist.all_gather_into_tensor(full.view(-1), local.view(-1))


def all_reduce_max(value):
    """max over ranks of a Python float (timing): a host tensor under gloo, a device tensor under RCCL."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if _host_backend() else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def unit_order(full, n_units, world, share):
    """rank-major [world*share, ...] (what ONE monolithic all-gather delivers) -> unit order u = i*world + r.  (The sharded
    renderers no longer need it: they gather row by row into a buffer that is in unit order already.)"""
    shape = tuple(full.shape[1:])
    return full.reshape((world, share) + shape).transpose(0, 1).reshape((world * share,) + shape)[:n_units]


class GatherRing:
    """Overlaps the all-gather of step i with the rendering of step i+1.

    The renderer writes into `slots` output buffers in turn; `submit(buf)` enqueues the all-gather of that
    buffer on a side stream behind the work already queued on the current stream and returns at once.
    Before a buffer is handed out again (`acquire`), the current stream waits for the gather that last read
    it.  With 2 slots the compute stream therefore never waits for the gather of the step just finished."""

    def __init__(self, make_local, world, slots=2, device=None):
        self.dev = device
        self.local = [make_local() for _ in range(slots)]
        self.full = [torch.empty((world,) + tuple(self.local[0].shape), dtype=self.local[0].dtype, device=self.local[0].device)
                     for _ in range(slots)]
        self.done = [None] * slots
        self.stream = torch.cuda.Stream(device)
        self.turn = 0

    def acquire(self):
        k = self.turn % len(self.local)
        if self.done[k] is not None:
            torch.cuda.current_stream(self.dev).wait_event(self.done[k])
        return k, self.local[k]

    def submit(self, k):
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            all_gather_into(self.full[k], self.local[k])
            e = torch.cuda.Event()
            e.record(self.stream)
        self.done[k] = e
        self.turn += 1
        return self.full[k], e

    def drain(self):
        torch.cuda.current_stream(self.dev).wait_stream(self.stream)


def gather_row(row, rank):
    """Complete one row `[world, ...]` of the unit-order buffer: rank r's block sits at row[r]; ONE all-gather fills the
    others.  RCCL gathers in place (send buffer = receive buffer + rank x count); gloo gets a copy of the own block."""
    all_gather_into(row, row[rank] if not _host_backend() else row[rank].clone())


def render_sharded(n_units, render_unit, unit_shape, dtype=torch.float32, device="cpu", gather=True, stream=None, block=1):
    """Render this rank's units with `render_unit(u) -> tensor[unit_shape]` and, if `gather`,
    return the full `[n_units, *unit_shape]` tensor on every rank (the rank's own `[share, *unit_shape]` units otherwise:
    the consumer is rank-local, e.g. DDP training).  The gather buffer is `[rows, world, block, *unit_shape]` - row i holds
    the units (i*world)*block .. (i*world + world)*block - 1, i.e. unit order - and row i is completed by one all-gather
    as soon as this rank's block of that row is rendered (per-row chunks, no reorder).  With `stream` the
    gathers run on that side stream
    and the call ALWAYS returns three values `(out, mine, done)`: `done` is the event the consumer must wait
    on (the caller may render the next batch meanwhile), or None when nothing ran on the side stream (one rank, or
    `gather=False`).  Without `stream` it returns `(out, mine)` and the result is ready on the current stream."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    share = padded_share(n_units, world, block)
    mine = shard_units(n_units, rank, world, block)
    if not gather or (world == 1 and not grouped()):
        local = torch.zeros((share,) + tuple(unit_shape), dtype=dtype, device=device)
        for i, u in enumerate(mine):
            local[i].copy_(render_unit(u))
        out = local[:n_units] if (gather and world == 1) else local
        return (out, mine) if stream is None else (out, mine, None)
    full = torch.zeros((share * world,) + tuple(unit_shape), dtype=dtype, device=device)
    n_rows = share // block
    rows = full.view((n_rows, world, block) + tuple(unit_shape))
    on_gpu = full.is_cuda
    cur = torch.cuda.current_stream(device) if on_gpu else None
    if stream is not None and on_gpu:
        full.record_stream(stream)
    for i in range(n_rows):
        for j in range(block):
            u = (i * world + rank) * block + j
            if u < n_units:
                full[u].copy_(render_unit(u))
        if stream is not None and on_gpu:
            ev = torch.cuda.Event()
            ev.record(cur)
            stream.wait_event(ev)
            with torch.cuda.stream(stream):
                gather_row(rows[i], rank)
        else:
            gather_row(rows[i], rank)
    if stream is None:
        return full[:n_units], mine
    done = None
    if on_gpu:
        done = torch.cuda.Event()
        done.record(stream)
    return full[:n_units], mine, done
