"""GPU: data parallelism of the HIP engine with LOCAL BatchNorm (the default DP mode) — two ranks: RCCL
("nccl") with one device per rank when the box has >= 2 GPUs, gloo with both ranks on cuda:0 otherwise.

  * engine level: each rank runs its shard through the HIP engine, the flat gradient buffers are all-reduced; the
    result must equal the T4 oracle "run the CPU restatement on each shard from identical weights, average the
    gradients" — forward quantities per shard within 1e-4, the averaged gradients within 5e-3 (relative L2), and both
    ranks must hold identical post-Adam weights;
  * entry-point level: `train_soft_intro_vae()` itself under a 2-process torchrun-style launch (WORLD_SIZE = 2 in the
    environment): per-rank batch = batch_size / 2, rank-0-only checkpoint, identical weights on both ranks at the end.
"""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _multi_gpu(world=2):
    return torch.cuda.device_count() >= world


def _backend(world=2):
    return "nccl" if _multi_gpu(world) else "gloo"


def _device(rank, world=2):
    return torch.device("cuda", rank if _multi_gpu(world) else 0)


def _env(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SIVAE_DP_SAME_DEVICE="0" if _multi_gpu(world) else "1",
                      SIVAE_DP_BACKEND=_backend(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (REPO, os.path.join(REPO, "soft-intro-vae-pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _engine_worker(rank, world, port, out_dir):
    _env(rank, world, port)
    import torch.distributed as dist
    from oracle import sivae_oracle as O
    import train_soft_intro_vae as T
    from sivae_hip import dp
    from sivae_hip.engine import SoftIntroEngine
    from sivae_hip.optim import FlatAdam
    dp.init(backend=_backend(world))
    dev = _device(rank, world)
    torch.cuda.set_device(dev)
    cdim, zdim, channels, image_size, B = 3, 16, [16, 32, 64], 32, 16
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1e-8)
    P = O.init_params(cdim, zdim, channels, image_size, seed=0)
    model = T.SoftIntroVAE(cdim=cdim, zdim=zdim, channels=channels, image_size=image_size)
    # rank 1 starts from different weights: the broadcast must overwrite them with rank 0's
    model.load_state_dict({k: (v.clone() if rank == 0 else torch.randn_like(v) if v.is_floating_point() else v.clone())
                           for k, v in P.items()}, strict=True)
    model = model.to(dev).train()
    oe, od = FlatAdam(model.encoder.parameters(), lr=2e-4), FlatAdam(model.decoder.parameters(), lr=2e-4)
    dp.broadcast_([oe.flat, od.flat] + [b for b in model.buffers()])
    sync = dp.GradSync()
    eng = SoftIntroEngine(model, oe, od, beta_kl=1.0, beta_rec=1.0, beta_neg=256.0, gamma_r=1e-8, grad_sync=sync)
    g = torch.Generator().manual_seed(1234)
    real = torch.rand(B, cdim, image_size, image_size, generator=g)
    noise = torch.randn(B, zdim, generator=g)
    eps = [torch.randn(B, zdim, generator=g) for _ in range(5)]
    lo, n = dp.shard_batch(B, world, rank)
    sl = slice(lo, lo + n)
    captured = {}
    orig = oe.step

    def step(grad_scale=1.0):
        captured["flat_grad"] = oe.flat_grad.detach().clone() * grad_scale  # all-reduced SUM x 1/world = the mean
        orig(grad_scale)
    oe.step = step
    es = eng.e_step(real[sl].to(dev), noise[sl].to(dev), [e[sl].to(dev) for e in eps[:3]], keep=True)
    torch.cuda.synchronize()
    # oracle on THIS shard (local BatchNorm = per-shard statistics), gradients to be averaged over the shards
    ref = O.e_step(P, real[sl], noise[sl], [e[sl] for e in eps[:3]], hp, channels, image_size)
    worst = max(float((es["kept"][k].double().cpu() - v.detach().double()).abs().max() / (v.detach().abs().max() + 1e-30))
                for k, v in ref.items())
    keys = O.trainable_keys(P, "encoder.")
    gshard = torch.cat([P[k].grad.reshape(-1) for k in keys])
    if dist.get_backend() == "nccl":  # (RCCL reduces device tensors)
        gdev = gshard.to(dev)
        dist.all_reduce(gdev)
        gshard = gdev.cpu()
    else:
        dist.all_reduce(gshard)
    gmean = gshard / world
    hip = torch.cat([p.reshape(-1) for p in [captured["flat_grad"].cpu()]])
    # FlatAdam's flat buffer is in encoder.parameters() order == state_dict order of the trainable keys
    order = [k for k, _ in model.encoder.named_parameters()]
    assert ["encoder." + k for k in order] == keys
    rel2 = float((hip.double() - gmean.double()).norm() / gmean.double().norm())
    torch.save(dict(worst_fwd=worst, grad_rel2=rel2, flat=oe.flat.detach().cpu(), calls=sync.calls,
                    overlapped=sync.overlapped),
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_local_bn_two_ranks_match_per_shard_oracle(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_engine_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(str(tmp_path / "rank0.pt")), torch.load(str(tmp_path / "rank1.pt"))
    for r in (r0, r1):
        assert r["worst_fwd"] <= 1e-4, r["worst_fwd"]   # per-shard forward / loss parity vs the per-shard oracle
        assert r["grad_rel2"] <= 5e-3, r["grad_rel2"]   # all-reduced mean gradient == mean of the shard-oracle gradients
        assert r["calls"] == 1
        assert r["overlapped"] == 1                     # the tail bucket was all-reduced DURING the backward
    assert torch.equal(r0["flat"], r1["flat"])          # replicas stay bit-identical after the Adam step


def _train_worker(rank, world, port, out_dir):
    _env(rank, world, port)
    os.chdir(out_dir)
    os.environ["SIVAE_SYNTHETIC_IMAGES"] = "64"
    import torch.distributed as dist
    import train_soft_intro_vae as T
    model = T.train_soft_intro_vae(dataset="synthetic-cifar10", z_dim=16, batch_size=16, num_workers=0, num_epochs=2,
                                   num_vae=0, beta_kl=1.0, beta_rec=1.0, beta_neg=256, seed=5, test_iter=3,
                                   save_interval=1, device=_device(rank, world))
    sd = model.state_dict()
    torch.save({k: v.cpu() for k, v in sd.items()}, os.path.join(out_dir, "sd_rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_train_entry_point_under_two_process_launch(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_train_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(str(tmp_path / "sd_rank0.pt")), torch.load(str(tmp_path / "sd_rank1.pt"))
    n_same = 0
    for k in a:
        if k.endswith(("running_mean", "running_var")):
            continue  # local BatchNorm: running statistics are per rank (each saw its own shards)
        assert torch.equal(a[k], b[k]), k  # weights: replicated, bit-identical after 8 DP iterations
        n_same += 1
    assert n_same > 20
    assert all(torch.isfinite(v).all() for v in a.values() if v.is_floating_point())
    # 64 images / global batch 16 = 4 iterations per epoch (each rank 8 images per iteration), 2 epochs; per iteration
    # 5 encoder passes, + the constructor probe, + deterministic dumps at cur_iter 0, 3, 6 and the final one
    assert int(a["encoder.main.1.num_batches_tracked"]) == 1 + 8 * 5 + 3 + 1
    ckpts = [f for f in os.listdir(tmp_path / "saves")]
    assert ckpts and all(f.endswith(".pth") for f in ckpts)  # written once (rank 0), not once per rank


def test_bench_two_rank_launch_reports_strong_and_weak_lines(tmp_path):
    """the driver's multi-GPU launch shape on one device: `torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`
    (gloo, both ranks on cuda:0 — the box has one GPU) prints ONE JSON line whose headline is the BASELINE configuration
    as stated (global batch 128 split over the ranks: 64 images per GPU, "strong") and which carries the
    fixed-per-GPU-work variant beside it (128 images per GPU, "weak"); the two flat-gradient all-reduces per iteration
    ran (finite losses on the all-reduced gradients)."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend",
           "gloo", "--same-device", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 128 and d["config"]["per_gpu_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["metric"].endswith("256x256 bs128") and d["unit"] == "img/s" and d["value"] > 0
    assert d["weak"]["scaling"] == "weak" and d["weak"]["global_batch"] == 256 and d["weak"]["per_gpu_batch"] == 128
    assert d["weak"]["value"] > 0
    import math
    assert all(math.isfinite(v) for v in d["config"]["final_stats"].values())
    assert "cpu_baseline" not in d  # (rank 0 of a multi-GPU launch does not time the CPU loop)


def test_bench_plain_python_start_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment (the shape of the driver's N = 1 command)
    re-executes itself under torch.distributed.run, one process per rank, and still prints ONE JSON line from rank 0."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
           "--scaling", "strong", "--steps", "1", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "dp2"
    assert d["config"]["per_gpu_batch"] == 64 and d["value"] > 0
