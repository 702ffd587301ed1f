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
