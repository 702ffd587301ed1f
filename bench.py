#!/usr/bin/env python
"""Headline benchmark: Soft-IntroVAE training images/sec at 256x256, global batch 128 (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full Soft-IntroVAE iteration of soft_intro_vae/train_soft_intro_vae.py:547-624 (5 encoder +
8 decoder forwards, both backwards, both Adam updates) on a synthetic U[0,1) NCHW batch already resident in
HBM, fp32, random-init weights of the CelebA-HQ-256 network (channels [64,128,256,512,512,512], z 512).
Data parallel runs keep the per-GPU work fixed at the N = 1 workload (128 images per GPU, global batch 128*N:
"weak" scaling, the per-image shard rule of this path); `--scaling strong` instead fixes the global batch at 128
(128/N images per GPU).  Gradients of the encoder and of the decoder are each all-reduced once per iteration
over RCCL (two flat fp32 buffers, 109.5 MB + 83.8 MB).

Prints ONE JSON line (rank 0). Besides the contract keys it carries
  roofline      the dominant MFMA kernel timed with HIP events around every launch in the timed region
                (algorithmic conv FLOPs / measured time vs the 157.3 TFLOP/s fp32 matrix peak), plus the
                whole-step algorithmic TFLOP/s (13 F_E + 19 F_D - 3 f_conv0 per image)
  cpu_baseline  the CPU oracle (oracle/sivae_oracle.py, a torch-CPU restatement pinned to the reference by
                golden vectors) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X fp32 matrix peak (MI355X_MICROARCH.md)

CONFIGS = {
    # name: (image_size, channels, zdim, global_batch, betas(kl, rec, neg), gamma_r)
    "celeb256": (256, [64, 128, 256, 512, 512, 512], 512, 128, (1.0, 0.5, 1024.0), 1e-8),
    "celeb128": (128, [64, 128, 256, 512, 512], 256, 128, (1.0, 0.5, 1024.0), 1e-8),
    "cifar10": (32, [64, 128, 256], 128, 256, (1.0, 1.0, 256.0), 1e-8),
}


def forward_flops(channels, image_size, zdim, cdim=3):
    """algorithmic conv+linear FLOPs (2/MAC) of one encoder / decoder forward per image, and the stem conv"""
    def conv(ci, co, k, hw):
        return 2.0 * ci * co * k * k * hw * hw
    fe = conv(cdim, channels[0], 5, image_size)
    f0 = fe
    cc, sz = channels[0], image_size // 2
    for ch in channels[1:]:
        if cc != ch:
            fe += conv(cc, ch, 1, sz)
        fe += conv(cc, ch, 3, sz) + conv(ch, ch, 3, sz)
        cc, sz = ch, sz // 2
    fe += 2 * conv(cc, cc, 3, sz)
    nfeat = cc * sz * sz
    fe += 2.0 * nfeat * 2 * zdim
    fd = 2.0 * zdim * nfeat
    cc, sz = channels[-1], sz
    for ch in channels[::-1]:
        if cc != ch:
            fd += conv(cc, ch, 1, sz)
        fd += conv(cc, ch, 3, sz) + conv(ch, ch, 3, sz)
        cc, sz = ch, sz * 2
    fd += 2 * conv(cc, cc, 3, sz) + conv(cc, cdim, 5, sz)
    return fe, fd, f0


def cpu_baseline(cfg, batch, iters, threads):
    """time the CPU oracle on a bounded sample of the same workload (rank 0, N = 1 only)"""
    from oracle import sivae_oracle as O
    image_size, channels, zdim, _, (bk, br, bn), gr = cfg
    if threads:
        torch.set_num_threads(threads)
    hp = dict(beta_rec=br, beta_kl=bk, beta_neg=bn, gamma_r=gr)
    P = O.init_params(3, zdim, channels, image_size, seed=0)
    opt_e = O.Adam(P, O.trainable_keys(P, "encoder."), 2e-4)
    opt_d = O.Adam(P, O.trainable_keys(P, "decoder."), 2e-4)
    g = torch.Generator().manual_seed(1234)
    real = torch.rand(batch, 3, image_size, image_size, generator=g)

    def one():
        noise = torch.randn(batch, zdim, generator=g)
        eps = [torch.randn(batch, zdim, generator=g) for _ in range(5)]
        O.train_iteration(P, opt_e, opt_d, real, noise, eps, hp, channels, image_size)

    one()  # warm-up
    t0 = time.time()
    for _ in range(iters):
        one()
    dt = time.time() - t0
    return dict(value=batch * iters / dt, unit="img/s", cores=torch.get_num_threads(), kind="port",
                sample="%d timed iterations at batch %d of the same %dx%d network after 1 warm-up (%.1f s); CPU "
                       "throughput is batch-insensitive" % (iters, batch, image_size, image_size, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="celeb256", choices=sorted(CONFIGS))
    ap.add_argument("--global-batch", type=int, default=None, help="override the global batch")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: config batch PER GPU (global = batch*N); strong: config batch is the global batch")
    ap.add_argument("--bootstrap", action="store_true", help="soft_intro_vae_bootstrap variant (config 5)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="fp32: the parity path (headline); bf16: config 3's build-defined mode (bf16 storage + bf16 MFMA "
                         "convs, fp32 accumulation / statistics / master weights)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-iters", type=int, default=1)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true",
                    help="testing only: every rank uses cuda:0 (with --backend gloo) to exercise the DP path on one GPU")
    ap.add_argument("--sync-bn", action="store_true",
                    help="synchronised BatchNorm across ranks (opt-in; default local BatchNorm as DDP would do)")
    ap.add_argument("--hip-graph", action="store_true",
                    help="capture the whole iteration into a HIP graph and time replays (single GPU; pays off where the "
                         "host launch rate is the limit: 32x32 nets, small batches)")
    ap.add_argument("--no-reuse", action="store_true",
                    help="re-execute the two D-step decoder forwards whose inputs and weights are unchanged since the "
                         "E-step (the reference does); default: replay them from the E-step's activations")
    args = ap.parse_args()

    from sivae_hip import dp, ops, rng
    from sivae_hip.engine import SoftIntroEngine
    from sivae_hip.optim import FlatAdam
    import train_soft_intro_vae as T
    import train_soft_intro_vae_bootstrap as TB

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the Soft-IntroVAE HIP engine has no CPU path)")
    world, rank, local = dp.init(backend=args.backend)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    dev = torch.device("cuda", local if (world > 1 and not args.same_device) else 0)
    torch.cuda.set_device(dev)

    cfg = CONFIGS[args.config]
    image_size, channels, zdim, gbatch, (bk, br, bn), gr = cfg
    if args.global_batch:
        gbatch = args.global_batch
    elif args.scaling == "weak":
        gbatch = gbatch * world
    if args.bootstrap:
        gr = 1.0
    _, per = dp.shard_batch(gbatch, world, rank)

    torch.manual_seed(0)
    model = (TB if args.bootstrap else T).SoftIntroVAE(cdim=3, zdim=zdim, channels=channels, image_size=image_size)
    model = model.to(dev).train()
    opt_e = FlatAdam(model.encoder.parameters(), lr=2e-4)
    opt_d = FlatAdam(model.decoder.parameters(), lr=2e-4)
    dp.broadcast_([opt_e.flat, opt_d.flat] + [b for b in model.buffers()])
    sync = dp.GradSync() if world > 1 else None
    sync_bn = dp.enable_sync_bn(args.sync_bn)
    eng = SoftIntroEngine(model, opt_e, opt_d, beta_kl=bk, beta_rec=br, beta_neg=bn, gamma_r=gr,
                          bootstrap=args.bootstrap, grad_sync=sync, reuse_decoder_forward=not args.no_reuse,
                          compute_dtype=args.dtype)
    rng.manual_seed(0, rank)
    g = torch.Generator().manual_seed(1234 + rank)
    real = torch.rand(per, 3, image_size, image_size, generator=g).to(dev)

    for _ in range(args.warmup):
        eng.soft_intro_step(real)
    if args.hip_graph:
        if world > 1:
            raise SystemExit("--hip-graph is single-GPU")
        eng.capture(real, warmup=1)
        step_fn = eng.replay
        args.no_kernel_timing = True  # (HIP events cannot be recorded per launch inside a graph)
    else:
        step_fn = eng.soft_intro_step
    torch.cuda.synchronize()
    dp.barrier()
    if not args.no_kernel_timing:
        ops.TIMER = ops.KernelTimer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step_fn(real)
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    stats = dict(zip(("lossE", "lossD", "loss_rec", "kl_real", "kl_fake", "kl_rec", "expelbo_rec", "expelbo_fake"),
                     [float(v) for v in last["stats"].cpu()]))

    if rank != 0:
        return
    fe, fd, f0 = forward_flops(channels, image_size, zdim)
    flops_img = 13 * fe + (17 if args.bootstrap else 19) * fd - 3 * f0
    ms_per_step = 1e3 * dt / args.steps
    value = gbatch * args.steps / dt
    step_tflops = flops_img * gbatch * args.steps / dt / 1e12 / world  # per GPU
    roof = dict(bound="mfma", achieved=None, peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=None, traffic=None)
    if timer is not None:
        summ = timer.summary()
        key = max(summ, key=lambda k: summ[k]["total_ms"])
        d = summ[key]
        ach = d["flops"] / (d["total_ms"] * 1e-3) / 1e12
        ach_ex = d["executed_flops"] / (d["total_ms"] * 1e-3) / 1e12
        conv_ms = sum(v["total_ms"] for v in summ.values())
        conv_fl = sum(v["flops"] for v in summ.values())
        roof.update(kernel=key, achieved=round(ach, 2), frac=round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                    # `achieved` counts the reference op's ALGORITHMIC FLOPs (2*B*H*W*Co*Ci*k*k); the Winograd
                    # F(2x2,3x3) kernels issue 16/36 of them to the matrix pipe, hence frac can exceed 1 —
                    # executed_* is what the MFMA pipe actually ran (its utilisation)
                    executed_tflops=round(ach_ex, 2), executed_frac=round(ach_ex / PEAK_FP32_MFMA_TFLOPS, 4),
                    launches=d["launches"], avg_launch_ms=round(d["avg_ms"], 4),
                    kernel_share_of_step=round(d["total_ms"] / (1e3 * dt), 4),
                    all_mfma_kernels=dict(tflops=round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                                          share_of_step=round(conv_ms / (1e3 * dt), 4)),
                    per_kernel={k: dict(launches=v["launches"], avg_ms=round(v["avg_ms"], 4),
                                        tflops=round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 2),
                                        executed_tflops=round(v["executed_flops"] / (v["total_ms"] * 1e-3) / 1e12, 2))
                                for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])})
    if roof.get("kernel") and args.config == "celeb256" and per == 128 and not args.bootstrap:
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of THIS workload
        # (tools/pmc_step.py + tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in separate passes, KiB units, read
        # side x2.0 as calibrated on a 1 GiB copy in the same pass) — counters cannot be read from inside the bench
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_pmc_traffic.json")
        if os.path.exists(tpath):
            tk = json.load(open(tpath))["kernels"]
            hit = [v for k, v in tk.items() if k.startswith(roof["kernel"][:-1] + ",")]
            if hit:
                roof["traffic"] = hit[0]["hbm_bytes"]
                roof["traffic_unit"] = "HBM bytes per launch (rocprofv3 PMC, profiles/r1_pmc_traffic.json)"
    executed_img = flops_img - (0.0 if args.no_reuse else 2 * fd)  # two decoder forwards replayed, not re-executed
    roof["step"] = dict(algorithmic_gflop_per_image=round(flops_img / 1e9, 1),
                        tflops_per_gpu=round(step_tflops, 2), frac=round(step_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
                        executed_gflop_per_image=round(executed_img / 1e9, 1),
                        executed_tflops_per_gpu=round(step_tflops * executed_img / flops_img, 2),
                        executed_frac=round(step_tflops * executed_img / flops_img / PEAK_FP32_MFMA_TFLOPS, 4))
    if timer is not None:
        # what the matrix pipe really ran per image (timer totals: direct kernels count 1:1, Winograd 16/36)
        issued = sum(v["executed_flops"] for v in summ.values()) / (per * args.steps)
        roof["step"].update(mfma_issued_gflop_per_image=round(issued / 1e9, 1),
                            mfma_issued_tflops_per_gpu=round(issued * per * args.steps / dt / 1e12, 2),
                            mfma_issued_frac=round(issued * per * args.steps / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4))
    out = {
        # BASELINE.json's metric is quoted at 256x256 with 128 images per GPU; any other run says what it ran
        "metric": "training images/sec (whole node) at 256x256 bs128" if (args.config == "celeb256" and per == 128)
        else "training images/sec (whole node) at %dx%d bs%d" % (image_size, image_size, per),
        "value": round(value, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": "soft_intro_vae%s %s %dx%d zdim=%d channels=%s, full E-step + D-step iteration"
                               % ("_bootstrap" if args.bootstrap else "", args.config, image_size, image_size, zdim,
                                  channels),
                   "global_batch": gbatch, "per_gpu_batch": per, "parallelism": "dp%d" % world,
                   "betas": {"kl": bk, "rec": br, "neg": bn}, "gamma_r": gr, "lr": 2e-4,
                   "decoder_forward_reuse": not args.no_reuse, "hip_graph": bool(args.hip_graph),
                   "batchnorm": "sync" if sync_bn else "local",
                   "final_stats": stats},
        "roofline": roof,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_batch, args.cpu_iters, args.cpu_threads)
        out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
