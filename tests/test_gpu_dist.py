"""Multi-rank GPU path on a ONE-GPU box (run with `-m gpu`): the self-launcher, the block-round-robin sharding of (scene, slice)
units through the HIP stack renderer and the all-gather order, with the ranks emulated on GPU 0 over gloo (RCCL refuses two
ranks per device; the same code takes the RCCL branch when every rank has its own GPU).

This module never touches the GPU itself: every GPU process is a child started before any HIP call of this process
(alphabetically it also runs before the other -m gpu modules)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from aadff.dist import spawn_ranks          # noqa: E402  (imports torch, makes no GPU call)

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


@pytest.mark.timeout(900)
def test_sharded_units_two_emulated_ranks_equal_single_rank(tmp_path):
    worker = os.path.join(HERE, "dist_gpu_worker.py")
    args = [worker, "--out", str(tmp_path), "--scenes", "4", "--res", "128", "--slices", "10"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    assert subprocess.call([sys.executable] + args, env=env, timeout=600) == 0                 # world 1
    # world 2 on one GPU: whole-scene blocks (the default), SURVEY 8e's u = r (mod 2) - every rank owns every other slice
    # of every scene and SKIPS the draws of the rest -, and half-scene blocks with the scene groups alternating over 2 streams
    variants = (("", []), ("_rr", ["--block", "1"]), ("_b5s2", ["--block", "5", "--streams", "2"]))
    for tag, extra in variants:
        assert spawn_ranks(args + extra + ["--tag", tag], 2, emulate=True, env=env, timeout=600) == 0, tag
    one = np.load(tmp_path / "full_w1.npy")
    plain = np.load(tmp_path / "plain_stacks.npy")
    assert one.shape == plain.shape == (40, 3, 128, 128)
    # a unit's inputs do not depend on who renders it; the PSF histogram uses float atomics (sum-order noise ~1e-7),
    # the convolution of identical maps is deterministic
    d1p = np.abs(one - plain).reshape(40, -1).max(1)
    assert d1p.max() <= 5e-6
    for tag, _ in variants:
        two = np.load(tmp_path / f"full_w2{tag}.npy")
        assert two.shape == one.shape
        d12 = np.abs(one - two).reshape(40, -1).max(1)
        print(f"\nsharded (2 emulated ranks{tag}) vs 1 rank: max |d| per unit <= {d12.max():.2e}; 1 rank units vs plain per-scene stacks <= {d1p.max():.2e}")
        assert d12.max() <= 5e-6, tag
    assert float(np.abs(one).mean()) > 0.05                                                      # real pixels, not zeros


@pytest.mark.timeout(1500)
def test_config3_at_its_real_size_one_rank_and_two_emulated_ranks(tmp_path):
    """BASELINE.json config 3 at the size it names - 16 scenes x 10 slices x 1024^2, 11x11 PSF grid, spp 2048 - through the
    zero-copy sharded renderer (slices written by the convolution straight into the unit-order gather buffer, one in-place
    all-gather per buffer row): every one of the 160 gathered units equals the plain per-scene stack (<= 5e-6: float atomics
    of the PSF histogram), on one rank and on two ranks emulated on the one GPU, in every rank."""
    worker = os.path.join(HERE, "dist_gpu_worker.py")
    args = [worker, "--out", str(tmp_path), "--scenes", "16", "--res", "1024", "--slices", "10", "--grid", "11", "--spp", "2048", "--check-inproc"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    assert subprocess.call([sys.executable] + args, env=env, timeout=900) == 0
    assert spawn_ranks(args + ["--streams", "2"], 2, emulate=True, env=env, timeout=1200) == 0       # whole-scene blocks, 2 streams per rank
    for name in ("check_w1_r0.json", "check_w2_r0.json", "check_w2_r1.json"):
        rec = json.load(open(tmp_path / name))
        assert rec["units"] == 160 and rec["shape"] == [160, 3, 1024, 1024] and rec["worst_abs_diff"] <= 5e-6 and rec["mean_abs_pixel"] > 0.05
        print(f"\n{name}: 160 units, worst |gathered - plain| = {rec['worst_abs_diff']:.2e}")


@pytest.mark.timeout(900)
def test_config3_partition_at_eight_emulated_ranks(tmp_path):
    """Config 3's rank count through the HIP renderer: 16 scenes x 10 slices = 160 units over EIGHT ranks (emulated on the one
    GPU over gloo, small images), whole-scene blocks (rank r renders scenes r and r + 8) and SURVEY 8e's round robin (every
    rank owns 1-2 slices of every scene and skips the draws of the rest): in every rank every gathered unit equals the plain
    per-scene stack."""
    worker = os.path.join(HERE, "dist_gpu_worker.py")
    args = [worker, "--out", str(tmp_path), "--scenes", "16", "--res", "64", "--slices", "10", "--grid", "3", "--spp", "256", "--check-inproc"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    for extra in ([], ["--block", "1"]):
        for f in os.listdir(tmp_path):
            os.remove(tmp_path / f)
        assert spawn_ranks(args + extra, 8, emulate=True, env=env, timeout=800) == 0, extra
        recs = [json.load(open(tmp_path / f"check_w8_r{r}.json")) for r in range(8)]
        assert all(rec["units"] == 160 and rec["shape"] == [160, 3, 64, 64] and rec["worst_abs_diff"] <= 5e-6 and rec["mean_abs_pixel"] > 0.05 for rec in recs)
        print(f"\n8 emulated ranks {extra or ['whole-scene blocks']}: worst |gathered - plain| over the ranks = {max(r['worst_abs_diff'] for r in recs):.2e}")


@pytest.mark.timeout(900)
def test_bench_self_launches_two_emulated_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for extra in ([], ["--gather"]):
        p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--emulate-ranks", "--steps", "6",
                            "--warmup", "2", "--spinup-s", "0.05"] + extra, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["config"]["ranks_emulated_on_one_gpu"] and rec["config"]["gather"] == bool(extra)
        assert rec["value"] > 0 and rec["flags"] & 3 == 0


@pytest.mark.timeout(900)
def test_bench_c3_mode_one_rank_and_two_emulated_ranks():
    """`bench.py --mode c3` (config 3: units sharded u = r mod N, gathered per buffer row) runs on one rank and on two
    emulated ranks and reports the expected 8-GPU figures with and without the gather next to the measured line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["AADFF_C3_SCENES"] = "4"
    for gpus in (1, 2):
        extra = ["--gpus", "2", "--emulate-ranks"] if gpus == 2 else []
        p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--mode", "c3", "--steps", "3", "--warmup", "1"] + extra,
                           env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
        assert rec["n_gpus"] == gpus and rec["scaling"] == "strong" and rec["value"] > 0
        assert rec["config"]["gathered_shape"] == [40, 3, 1024, 1024] and "expected_scaling" in rec


@pytest.mark.timeout(900)
def test_config5_ddp_consumer_on_hip_rendered_stacks_two_emulated_ranks(tmp_path):
    """BASELINE.json config 5's data flow on the GPU: two ranks (emulated on the one GPU: gloo instead of RCCL) each render
    the focal stacks of their own mini-batch with the HIP renderer (PSFNet.render_stack, one fused launch per stack) and
    train a DistributedDataParallel consumer; no data-path collective, and DDP leaves identical weights on both ranks although
    they rendered different scenes (examples/config5_ddp_render.py)."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    script = os.path.join(REPO, "examples", "config5_ddp_render.py")
    # the miniature (64 x 64, 5 slices) and the configuration's own size: configs/aber_aware_dff_dfv.yml:19-21 = bs 2, n_stack 8, 480 x 640
    for extra in ([], ["--hw", "480", "640", "--n-stack", "8", "--batch", "2", "--steps", "2"]):
        assert spawn_ranks([script, "--save", str(tmp_path), "--steps", "3"] + extra, 2, emulate=True, env=env, timeout=600) == 0, extra
        a, b = (torch.load(tmp_path / f"config5_rank{r}.pt") for r in range(2))
        assert all(torch.equal(x, y) for x, y in zip(a["params"], b["params"])), "DDP ranks ended with different consumer weights"
        assert np.isfinite(a["loss"]) and np.isfinite(b["loss"]) and a["loss"] != b["loss"]      # different scenes per rank


@pytest.mark.timeout(900)
def test_rccl_branch_on_a_one_rank_group(tmp_path):
    """Everything that only exists under RCCL, executed once on the one GPU (AADFF_FORCE_GROUP=1 forms a ONE-rank `nccl` group):
    `init_process_group("nccl", device_id=...)`, the IN-PLACE per-row all-gather whose send buffer aliases the receive buffer
    (aadff.dist.gather_row), the side-stream form, GatherRing, all_reduce_max on a device tensor, barrier and teardown - through
    the sharded renderer (every gathered unit equal to the plain per-scene stack) and through `bench.py --gather` / `--mode c3`.
    The multi-rank tests above run over gloo (RCCL refuses two ranks per device); this one makes sure that the first real
    `bench.py --gpus 8 [--mode c3] [--gather]` cannot fail on something a single GPU would have shown."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AADFF_EMULATE_RANKS")}
    env.update(AADFF_FORCE_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(HERE, "dist_gpu_worker.py")
    for extra in ([], ["--block", "1", "--streams", "2"]):
        args = [worker, "--out", str(tmp_path), "--scenes", "4", "--res", "128", "--slices", "10", "--check-inproc"] + extra
        p = subprocess.run([sys.executable] + args, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        rec = json.load(open(tmp_path / "check_w1_r0.json"))
        assert rec["units"] == 40 and rec["worst_abs_diff"] <= 5e-6 and rec["mean_abs_pixel"] > 0.05
        print(f"\none-rank RCCL group {extra}: 40 units gathered in place, worst |gathered - plain| = {rec['worst_abs_diff']:.2e}")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gather", "--steps", "6", "--warmup", "2", "--spinup-s", "0.05",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[:600]            # ONE JSON line: RCCL's version banner must not land on stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["config"]["gather"] is True and not rec["config"]["ranks_emulated_on_one_gpu"] and rec["value"] > 0
    # the driver's own launch line for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), here with one process: the
    # group comes from torchrun's RANK / WORLD_SIZE / MASTER_* and the line must still be the only thing on stdout
    from aadff.dist import free_port
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--spinup-s", "0.05", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1, p.stdout[:600]
    env["AADFF_C3_SCENES"] = "4"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--mode", "c3", "--steps", "3", "--warmup", "1", "--spinup-s", "0.05"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert rec["config"]["gather"] is True and rec["config"]["gathered_shape"] == [40, 3, 1024, 1024] and rec["value"] > 0
