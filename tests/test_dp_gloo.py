"""T4 plumbing on CPU: the data-parallel layer (sivae_hip.dp) under gloo with world_size 2.

DP semantics of this engine: per-image batch shards, replicated weights, ONE all-reduce(SUM) of the flat
gradient buffer per network per backward, 1/world folded into the optimizer. The oracle for it: run the CPU
oracle on each shard from identical weights, average the gradients — which the 2-process run must equal."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import sivae_oracle as O
    from sivae_hip import dp
    w, r, _ = dp.init(backend="gloo")
    assert (w, r) == (world, rank) and dp.world_size() == world
    channels, image_size, zdim, gB = [8, 16], 16, 8, 8
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=64.0, gamma_r=1e-8)
    # rank 1 starts from DIFFERENT weights: broadcast_ must overwrite them with rank 0's
    P = O.init_params(3, zdim, channels, image_size, seed=0 if rank == 0 else 99)
    enc_keys = O.trainable_keys(P, "encoder.")
    flat = torch.cat([P[k].reshape(-1) for k in P if P[k].is_floating_point()])
    dp.broadcast_([flat])
    off = 0
    for k in P:
        if P[k].is_floating_point():
            n = P[k].numel()
            P[k] = flat[off:off + n].view(P[k].shape).clone()
            off += n
    g = torch.Generator().manual_seed(7)
    real = torch.rand(gB, 3, image_size, image_size, generator=g)
    noise = torch.randn(gB, zdim, generator=g)
    eps = [torch.randn(gB, zdim, generator=g) for _ in range(3)]
    start, per = dp.shard_batch(gB, world, rank)
    sl = slice(start, start + per)
    O.e_step(P, real[sl], noise[sl], [e[sl] for e in eps], hp, channels, image_size)
    flat_grad = torch.cat([P[k].grad.reshape(-1) for k in enc_keys])
    sync = dp.GradSync()
    sync(flat_grad)
    flat_grad.mul_(sync.grad_scale)
    assert sync.calls == 1 and sync.bytes == flat_grad.numel() * 4
    dp.barrier()
    np.save(os.path.join(out_dir, "grad_rank%d.npy" % rank), flat_grad.numpy())
    dist.destroy_process_group()


def test_two_rank_gradient_mean_matches_per_shard_oracle(tmp_path):
    from oracle import sivae_oracle as O
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0 = np.load(tmp_path / "grad_rank0.npy")
    g1 = np.load(tmp_path / "grad_rank1.npy")
    assert np.array_equal(g0, g1), "ranks disagree after the all-reduce"
    # single-process oracle of the DP semantics: per-shard gradients from identical weights, averaged
    torch.set_num_threads(2)
    channels, image_size, zdim, gB = [8, 16], 16, 8, 8
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=64.0, gamma_r=1e-8)
    g = torch.Generator().manual_seed(7)
    real = torch.rand(gB, 3, image_size, image_size, generator=g)
    noise = torch.randn(gB, zdim, generator=g)
    eps = [torch.randn(gB, zdim, generator=g) for _ in range(3)]
    acc = None
    for r in range(world):
        P = O.init_params(3, zdim, channels, image_size, seed=0)
        sl = slice(r * gB // world, (r + 1) * gB // world)
        O.e_step(P, real[sl], noise[sl], [e[sl] for e in eps], hp, channels, image_size)
        fg = torch.cat([P[k].grad.reshape(-1) for k in O.trainable_keys(P, "encoder.")])
        acc = fg if acc is None else acc + fg
    ref = (acc / world).numpy()
    assert np.abs(g0 - ref).max() <= 1e-5 * (np.abs(ref).max() + 1e-30)


def test_shard_batch_rules():
    from sivae_hip import dp
    assert dp.shard_batch(128, 8, 3) == (48, 16)
    assert dp.shard_batch(64, 8, 7) == (56, 8)
    with pytest.raises(ValueError):
        dp.shard_batch(10, 4, 0)
    assert dp.env_world()[0] >= 1


def _decision_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sivae_hip import dp
    dp.init(backend="gloo")
    # an abort condition seen by ONE rank must be seen by all (NaN loss / negative KL difference: a rank raising alone
    # would leave the others in the next all-reduce)
    seen = dp.any_rank(rank == 1)
    none = dp.any_rank(False)
    # the unseeded run's stream seed: rank 0's draw everywhere (DistributedSampler needs ONE seed)
    seed = dp.broadcast_int(1234567 + 1000 * rank)
    # epoch statistics: every rank logs the global mean
    means = dp.mean_over_ranks([float(rank), 10.0 + rank])
    # the sampler partitions the dataset with the broadcast seed: no sample repeated, none dropped
    ds = torch.utils.data.TensorDataset(torch.arange(64))
    sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=seed)
    idx = torch.tensor(list(iter(sampler)))
    torch.save(dict(seen=seen, none=none, seed=seed, means=means, idx=idx), os.path.join(out_dir, "dec%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_collective_decisions_and_seed_broadcast(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_decision_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / ("dec%d.pt" % k)) for k in range(world)]
    assert all(x["seen"] for x in r) and not any(x["none"] for x in r)
    assert r[0]["seed"] == r[1]["seed"] == 1234567
    assert r[0]["means"] == r[1]["means"] == [0.5, 10.5]
    both = torch.cat([r[0]["idx"], r[1]["idx"]])
    assert sorted(both.tolist()) == list(range(64))


def test_gradsync_early_bucket_is_the_early_final_half():
    """the overlapped bucket = the second half of the parameter list (final in the first half of the backward); for the
    decoder that is the shallow blocks, never the fc / 512-channel gradients that only become final at the end"""
    sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
    from sivae_hip import dp

    class P:
        def __init__(self, n):
            self._n = n
            self.__dict__["_sivae_on_grad"] = None

        def numel(self):
            return self._n

        def register_post_accumulate_grad_hook(self, h):
            pass

    class Opt:
        def __init__(self, sizes):
            self.params = [P(n) for n in sizes]
            self.flat_grad = torch.zeros(sum(sizes))
    sync = dp.GradSync(overlap=False)
    dec = Opt([1000, 900, 800, 50, 40, 30, 20, 10])  # decoder-like: big tensors first
    split, tail = sync._plan(dec)
    assert split == 1000 + 900 + 800 + 50 and len(tail) == 4
    enc = Opt([10, 20, 30, 40, 50, 800, 900, 1000])  # encoder-like: big tensors last
    split, tail = sync._plan(enc)
    assert split == 10 + 20 + 30 + 40 and sum(p.numel() for p in tail) == 2750


def test_bench_without_world_size_becomes_its_own_launcher(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE set must start one process per rank itself (torch.distributed.run on
    127.0.0.1) instead of refusing: without a GPU each rank then stops at the device check — the message that proves the
    ranks were started (the old behaviour was a 'launch with torch.distributed.run' exit in the parent)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONDONTWRITEBYTECODE="1")
    if torch.cuda.is_available():
        pytest.skip("the GPU twin of this test is tests/test_dp_gpu.py::test_bench_plain_python_start_spawns_its_own_ranks")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode != 0
    assert "needs a ROCm device" in out.stderr, out.stderr[-1500:]
    assert "launch with torch.distributed.run" not in out.stderr
