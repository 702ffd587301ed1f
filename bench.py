#!/usr/bin/env python
"""Headline benchmark: Soft-IntroVAE training images/sec at 256x256, global batch 128 (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full Soft-IntroVAE iteration of soft_intro_vae/train_soft_intro_vae.py:547-624 (5 encoder +
8 decoder forwards, both backwards, both Adam updates) on a synthetic U[0,1) NCHW batch already resident in
HBM, fp32, random-init weights of the CelebA-HQ-256 network (channels [64,128,256,512,512,512], z 512).
Data-parallel runs (N > 1) report the BASELINE configuration as it is stated — global batch 128 split per image over the
N GPUs (128/N images per GPU, "scaling": "strong") — as the headline `value`, and ALSO time the fixed-per-GPU-work
variant (128 images per GPU, global batch 128*N) in the same invocation, reported under "weak"; each names its
per-GPU batch.  `--scaling strong|weak` runs just one of them.  Gradients of the encoder and of the decoder are each
all-reduced once per iteration over RCCL (two flat fp32 buffers, 109.5 MB + 83.8 MB).

Prints ONE JSON line (rank 0). Besides the contract keys it carries
  roofline      the dominant MFMA kernel timed with HIP events around every launch in the timed region.  `achieved` /
                `frac` are PHYSICAL: FLOPs the kernel issued to the matrix pipe (Winograd F(2x2,3x3) issues 16/36 of a
                3x3 conv's multiplies) / measured time, against the dense peak of the dtype (157.3 TFLOP/s fp32 MFMA,
                2.5 PFLOP/s bf16) — always <= 1; `mfma_busy_pmc` is the same kernel's matrix-pipe busy fraction from
                the committed rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES); the reference op's ALGORITHMIC FLOPs per
                second (what SURVEY 8d's formula counts, can exceed the peak for Winograd) are under `algorithmic_*`;
                `step` carries the whole-iteration figures (13 F_E + 19 F_D - 3 f_conv0 per image)
  cpu_baseline  the CPU oracle (oracle/sivae_oracle.py, a torch-CPU restatement pinned to the reference by
                golden vectors) timed on this host's cores on a bounded sample of the same workload.
  value_untimed the same workload re-timed in the same process WITHOUT the per-launch HIP events (`value` is the
                instrumented run the roofline object comes from: ~1 % slower in fp32, ~5 % in bf16 mode)
  also          (N = 1, default invocation) the other BASELINE.json configurations and the per-GPU shards of the
                data-parallel ones, each measured in this same process: celeb256 at the 16-image shard (config 4 on 8
                GPUs), soft_intro_vae_bootstrap at batch 64 and at its 8-image shard (config 5), cifar10 batch 256
                (config 2), celeb128 batch 128 in bf16 mode (config 3, with its HBM fraction), and the bf16 mode at the
                headline workload (celeb256 batch 128 and its 16-image shard).  `value` there is the untimed rate; `mfma_issued_frac` comes from a few instrumented iterations of the same engine.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide; the 5 PF headline figure includes 2:1 sparsity)
PEAK_HBM_TBS = 8.0
PROFILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")

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


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, batch, iters, threads):
    """time the CPU oracle on a bounded sample of the same workload (rank 0, N = 1 only): >= 2 timed iterations after
    one warm-up at the best thread count of a short sweep (oversubscribing a big host is SLOWER: 128 threads gave 0.235
    img/s in round 1 where 8 gave 0.365)"""
    from oracle import sivae_oracle as O
    image_size, channels, zdim, _, (bk, br, bn), gr = cfg
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

    ncpu = os.cpu_count() or 1
    physical = max(1, ncpu // 2)
    sweep = {}
    if threads:
        cand = [threads]
    else:
        cand = sorted({t for t in (8, 16, 32, 64, physical) if t <= ncpu})
    torch.set_num_threads(cand[0])
    one()  # warm-up (allocator, oneDNN primitive caches)
    for t in cand:
        torch.set_num_threads(t)
        t0 = time.time()
        one()
        sweep[t] = batch / (time.time() - t0)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    iters = max(2, iters)
    t0 = time.time()
    for _ in range(iters):
        one()
    dt = time.time() - t0
    return dict(value=batch * iters / dt, unit="img/s", cores=best, kind="port", cpu_model=_cpu_model(),
                host_logical_cpus=ncpu, thread_sweep_img_per_s={str(k): round(v, 4) for k, v in sweep.items()},
                sample="%d timed iterations at batch %d of the same %dx%d network after 1 warm-up + a %d-point thread "
                       "sweep (%.1f s timed); CPU throughput is batch-insensitive" % (iters, batch, image_size,
                                                                                    image_size, len(sweep), dt))


def _pmc_match(kernel_key, kernels):
    """the entry of a PMC summary (keys = demangled kernel names without spaces) that a KernelTimer key stands for:
    exact / template-prefix match (fp32 kernels), or — bf16 timer keys name the tile class, e.g.
    "bf16_conv_kernel<3,co128>" — the heaviest instantiation <KS, WM, WN, WVM, WVN, ...> with WVM*WM*32 == 128"""
    pref = kernel_key[:-1] + ","  # "conv_wino_kernel<1,4,false>" -> "conv_wino_kernel<1,4,false,"
    hit = [v for k, v in kernels.items() if k == kernel_key or k.startswith(pref)]
    if hit:
        return hit[0]
    if kernel_key.startswith("bf16_") and "<" in kernel_key:
        base, targs = kernel_key[:-1].split("<")
        targs = targs.split(",")
        cand = []
        for k, v in kernels.items():
            if not k.startswith(base + "<" + targs[0] + ","):
                continue
            a = k[k.index("<") + 1:-1].split(",")
            if len(targs) > 1 and targs[1].startswith("co"):
                if int(a[1]) * int(a[3]) * 32 != int(targs[1][2:]):
                    continue
            cand.append(v)
        if cand:
            # several instantiations share a timer key (small / big pixel tile): the one with the largest share of the step
            return max(cand, key=lambda v: (v.get("share_of_gpu_active", 0.0), v.get("launches", 0) * v.get("hbm_bytes", 0)))
    return None


def _pmc_files(kind, config, dtype):
    """committed PMC summaries of a workload, newest round first: profiles/r<N>_pmc_<kind>_<workload>*.json"""
    import glob
    import re
    tag = {("celeb256", "fp32"): ("celeb256_bs128_fp32", "bs128", "final"), ("celeb128", "bf16"): ("celeb128_bf16",)}.get(
        (config, dtype), ())
    out = []
    for path in glob.glob(os.path.join(PROFILES, "r*_pmc_%s*.json" % kind)):
        name = os.path.basename(path)
        m = re.match(r"r(\d+)_pmc_%s_?(.*)\.json" % kind, name)
        if not m:
            continue
        rest = m.group(2)
        if kind == "traffic" and rest in ("", "direct_kernels"):
            rest = "bs128" if rest == "" else "skip"
        if config == "celeb256" and "bf16" in rest:
            continue
        if any(rest == t or rest.startswith(t) for t in tag):
            out.append((int(m.group(1)), name))
    return [n for _, n in sorted(out, key=lambda rn: (-rn[0], rn[1]))]


def _lib_sha16():
    from sivae_hip import lib
    return lib.sha256()[:16]


def _stamp(d, source):
    """provenance of a counter figure: the file it came from, the library build it was taken with, and whether that is
    the build this run timed (`stale`: true when it is another build or the file carries no stamp — counters cannot be
    read from inside the bench, so a kernel edit after the PMC pass must show)"""
    sha = d.get("lib_sha256_16")
    return dict(source="profiles/" + source, lib_sha256_16=sha, stale=(sha is None or sha != _lib_sha16()))


def _pmc_busy(kernel_key, config="celeb256", dtype="fp32"):
    """matrix-pipe busy fraction and MFMA op count of `kernel_key` from the newest committed rocprofv3 --pmc pass of the
    workload (tools/pmc_mfma_busy.py over SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE / SQ_INSTS_VALU_MFMA_MOPS_F32)"""
    for name in _pmc_files("mfma_busy", config, dtype):
        d = json.load(open(os.path.join(PROFILES, name)))
        hit = _pmc_match(kernel_key, d["kernels"])
        if hit:
            out = dict(value=hit["mfma_busy_frac"], mfma_mops_per_launch=hit.get("mfma_mops_f32_per_launch"))
            out.update(_stamp(d, name))
            return out
    return None


def _pmc_traffic(kernel_key, config, dtype):
    """HBM bytes per launch of `kernel_key` + whole-step bytes from the newest committed FETCH_SIZE / WRITE_SIZE passes"""
    for name in _pmc_files("traffic", config, dtype):
        d = json.load(open(os.path.join(PROFILES, name)))
        hit = _pmc_match(kernel_key, d["kernels"])
        out = dict(kernel_bytes=hit["hbm_bytes"] if hit else None, step_bytes=d.get("step_total_hbm_bytes"))
        out.update(_stamp(d, name))
        return out
    return None


def measure(args, cfg, gbatch, world, rank, dev, scaling, steps=None, warmup=None, untimed_steps=0):
    """build a fresh model + engine for `gbatch` images over `world` ranks, run warm-up + timed steps
    -> dict(dt, per, stats, timer_summary); untimed_steps > 0: a second region of that many steps without the per-launch
    HIP events (dt_untimed)"""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    from sivae_hip import dp, ops, rng
    from sivae_hip.engine import SoftIntroEngine
    from sivae_hip.optim import FlatAdam
    import train_soft_intro_vae as T
    import train_soft_intro_vae_bootstrap as TB
    image_size, channels, zdim, _, (bk, br, bn), gr = cfg
    if args.bootstrap:
        gr = 1.0
    _, per = dp.shard_batch(gbatch, world, rank)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):  # the constructor prints like the reference's: stdout carries ONE JSON line
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

    for _ in range(warmup):
        eng.soft_intro_step(real)
    no_timing = args.no_kernel_timing
    if args.hip_graph:
        if world > 1:
            raise SystemExit("--hip-graph is single-GPU")
        eng.capture(real, warmup=1)
        step_fn = eng.replay
        no_timing = True  # (HIP events cannot be recorded per launch inside a graph)
    else:
        step_fn = eng.soft_intro_step
    torch.cuda.synchronize()
    dp.barrier()
    if not no_timing:
        ops.TIMER = ops.KernelTimer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step_fn(real)
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    dt_untimed = None
    if untimed_steps > 0:
        torch.cuda.synchronize()
        dp.barrier()
        t1 = time.perf_counter()
        for _ in range(untimed_steps):
            last = step_fn(real)
        torch.cuda.synchronize()
        dp.barrier()
        torch.cuda.synchronize()
        dt_untimed = time.perf_counter() - t1
    # a persistent BatchNorm backward that gave up at its grid barrier leaves garbage gradients behind and a FAST iteration:
    # never report such a run (collective under DP, like the training loop's drain())
    poisoned = ops.bn_fused_poisoned()
    if dp.any_rank(poisoned is not None, dev if world > 1 else None):
        raise SystemExit(poisoned or "another rank's one-pass BatchNorm backward gave up at its grid barrier")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    stats = dict(zip(("lossE", "lossD", "loss_rec", "kl_real", "kl_fake", "kl_rec", "expelbo_rec", "expelbo_fake"),
                     [float(v) for v in last["stats"].cpu()]))
    summ = timer.summary() if timer is not None else None
    del eng, model, opt_e, opt_d, real, last
    torch.cuda.empty_cache()
    return dict(dt=dt, per=per, gbatch=gbatch, stats=stats, summ=summ, sync_bn=sync_bn, scaling=scaling, gamma_r=gr,
                steps=steps, dt_untimed=dt_untimed, untimed_steps=untimed_steps,
                paired=os.environ.get("SIVAE_PAIR_PASSES", "auto") != "0")


def also_legs(args, world, rank, dev):
    """the other BASELINE.json configurations / per-GPU shards, measured in this process (N = 1): -> dict"""
    import copy
    legs = [  # name, config, global batch, bootstrap, dtype, untimed steps, instrumented steps, warm-up
        ("celeb256_fp32_bs16_shard", "celeb256", 16, False, "fp32", 16, 4, 3),
        ("bootstrap256_fp32_bs64", "celeb256", 64, True, "fp32", 6, 2, 2),
        ("bootstrap256_fp32_bs8_shard", "celeb256", 8, True, "fp32", 20, 4, 3),
        ("cifar10_fp32_bs256", "cifar10", 256, False, "fp32", 25, 5, 5),
        ("celeb128_bf16_bs128", "celeb128", 128, False, "bf16", 14, 4, 4),
        # SURVEY 8 gives config 4 as "fp32 parity / bf16 perf": the build-defined bf16 mode at the headline workload
        ("celeb256_bf16_bs128", "celeb256", 128, False, "bf16", 6, 2, 2),
        ("celeb256_bf16_bs16_shard", "celeb256", 16, False, "bf16", 12, 3, 3),
    ]
    out = {}
    for name, config, gb, boot, dtype, n_untimed, n_timed, n_warm in legs:
        a = copy.copy(args)
        a.bootstrap, a.dtype, a.config, a.hip_graph, a.no_kernel_timing = boot, dtype, config, False, False
        cfg = CONFIGS[config]
        image_size, channels, zdim = cfg[0], cfg[1], cfg[2]
        r = measure(a, cfg, gb, world, rank, dev, "strong", steps=n_timed, warmup=n_warm, untimed_steps=n_untimed)
        peak = PEAK_FP32_MFMA_TFLOPS if dtype == "fp32" else PEAK_BF16_MFMA_TFLOPS
        fe, fd, f0 = forward_flops(channels, image_size, zdim)
        flops_img = 13 * fe + (17 if boot else 19) * fd - 3 * f0
        value = gb * n_untimed / r["dt_untimed"]
        summ = r["summ"]
        issued = sum(v["executed_flops"] for v in summ.values()) / r["dt"] / 1e12
        key = max(summ, key=lambda k: summ[k]["total_ms"])
        d = summ[key]
        leg = {"value": round(value, 2), "unit": "img/s", "ms_per_step": round(1e3 * r["dt_untimed"] / n_untimed, 3),
               "steps": n_untimed, "warmup": n_warm + n_timed, "global_batch": gb, "dtype": "f32" if dtype == "fp32" else "bf16",
               "workload": "soft_intro_vae%s %s %dx%d zdim=%d" % ("_bootstrap" if boot else "", config, image_size,
                                                                 image_size, zdim),
               "algorithmic_tflops": round(flops_img * value / 1e12, 2),
               "mfma_issued_frac": round(issued / peak, 4), "mfma_peak_tflops": peak,
               "instrumented": {"value": round(gb * n_timed / r["dt"], 2), "steps": n_timed, "dominant_kernel": key,
                                "dominant_kernel_issued_frac": round(d["executed_flops"] / (d["total_ms"] * 1e-3) / 1e12 / peak, 4),
                                "dominant_kernel_avg_ms": round(d["avg_ms"], 4)}}
        if dtype == "bf16":
            tr = _pmc_traffic(key, config, dtype)
            if tr is not None and tr["step_bytes"]:
                tbs = tr["step_bytes"] / (r["dt_untimed"] / n_untimed) / 1e12
                leg["hbm"] = dict(step_bytes=tr["step_bytes"], achieved_tbs=round(tbs, 3), peak_tbs=PEAK_HBM_TBS,
                                  frac=round(tbs / PEAK_HBM_TBS, 4), source=tr["source"],
                                  lib_sha256_16=tr["lib_sha256_16"], stale=tr["stale"])
        out[name] = leg
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="celeb256", choices=sorted(CONFIGS))
    ap.add_argument("--global-batch", type=int, default=None, help="override the global batch")
    ap.add_argument("--scaling", default="both", choices=["both", "weak", "strong"],
                    help="strong: the config's batch is the GLOBAL batch (BASELINE config 4: 128 over N GPUs); weak: "
                         "the config's batch PER GPU; both (default): strong is the headline value, weak is reported "
                         "beside it (identical runs for N = 1)")
    ap.add_argument("--bootstrap", action="store_true", help="soft_intro_vae_bootstrap variant (config 5)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="fp32: the parity path (headline); bf16: config 3's build-defined mode (bf16 storage + bf16 MFMA "
                         "convs, fp32 accumulation / statistics / master weights)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-iters", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0: short thread sweep, the best count is used")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the `also` object (the other configurations / shard sizes measured in the same process) and "
                         "the untimed re-run of the headline")
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

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N` (the shape of the N = 1 command): become the launcher — one process
        # per GPU under torch.distributed.run on the loopback address, rank 0 prints the ONE JSON line.  (The reference's own
        # multi-GPU precedent self-spawns too: style_soft_intro_vae/launcher.py:126-129.)
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    from sivae_hip import dp

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the Soft-IntroVAE HIP engine has no CPU path)")
    if args.same_device:
        os.environ["SIVAE_DP_SAME_DEVICE"] = "1"  # (persistent kernels of several processes on one GPU: see ops.BN_FUSED)
    world, rank, local = dp.init(backend=args.backend)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    dev = torch.device("cuda", local if (world > 1 and not args.same_device) else 0)
    torch.cuda.set_device(dev)

    cfg = CONFIGS[args.config]
    image_size, channels, zdim, cfg_batch, (bk, br, bn), gr = cfg
    base = args.global_batch if args.global_batch else cfg_batch
    runs = []
    if world == 1 or args.global_batch:
        runs.append(("strong" if args.scaling != "weak" else "weak", base))
    else:
        if args.scaling in ("both", "strong"):
            runs.append(("strong", base))
        if args.scaling in ("both", "weak"):
            runs.append(("weak", base * world))
    headline_run = (world == 1 and args.config == "celeb256" and not args.bootstrap and args.dtype == "fp32"
                    and not args.global_batch and not args.hip_graph and not args.no_kernel_timing and not args.no_also)
    results = [measure(args, cfg, gb, world, rank, dev, sc, untimed_steps=8 if (headline_run and i == 0) else 0)
               for i, (sc, gb) in enumerate(runs)]
    also = also_legs(args, world, rank, dev) if headline_run else None
    if rank != 0:
        return
    head = results[0]
    dt, per, gbatch, stats, summ = head["dt"], head["per"], head["gbatch"], head["stats"], head["summ"]

    peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "fp32" else PEAK_BF16_MFMA_TFLOPS
    fe, fd, f0 = forward_flops(channels, image_size, zdim)
    flops_img = 13 * fe + (17 if args.bootstrap else 19) * fd - 3 * f0
    ms_per_step = 1e3 * dt / args.steps
    value = gbatch * args.steps / dt
    step_tflops = flops_img * gbatch * args.steps / dt / 1e12 / world  # per GPU, algorithmic
    roof = dict(bound="mfma", achieved=None, peak=peak, unit="TFLOP/s", frac=None, traffic=None)
    if summ is not None:
        key = max(summ, key=lambda k: summ[k]["total_ms"])
        d = summ[key]
        alg = d["flops"] / (d["total_ms"] * 1e-3) / 1e12
        issued = d["executed_flops"] / (d["total_ms"] * 1e-3) / 1e12
        conv_ms = sum(v["total_ms"] for v in summ.values())
        conv_fl = sum(v["flops"] for v in summ.values())
        conv_ex = sum(v["executed_flops"] for v in summ.values())
        roof.update(kernel=key,
                    # PHYSICAL: what the matrix pipe ran / time / dense peak of the dtype  (<= 1)
                    achieved=round(issued, 2), frac=round(issued / peak, 4),
                    basis="MFMA-issued FLOPs of the dominant kernel (Winograd F(2x2,3x3) issues 16/36, the upsample-phase "
                          "kernels 9/36 of the reference conv's multiplies) / HIP-event time / dense %s MFMA peak"
                          % ("fp32" if args.dtype == "fp32" else "bf16"),
                    algorithmic_tflops=round(alg, 2), algorithmic_frac=round(alg / peak, 4),
                    launches=d["launches"], avg_launch_ms=round(d["avg_ms"], 4),
                    kernel_share_of_step=round(d["total_ms"] / (1e3 * dt), 4),
                    mfma_busy_pmc=_pmc_busy(key, args.config, args.dtype),
                    all_mfma_kernels=dict(issued_tflops=round(conv_ex / (conv_ms * 1e-3) / 1e12, 2),
                                          issued_frac=round(conv_ex / (conv_ms * 1e-3) / 1e12 / peak, 4),
                                          algorithmic_tflops=round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                                          share_of_step=round(conv_ms / (1e3 * dt), 4)),
                    per_kernel={k: dict(launches=v["launches"], avg_ms=round(v["avg_ms"], 4),
                                        issued_tflops=round(v["executed_flops"] / (v["total_ms"] * 1e-3) / 1e12, 2),
                                        algorithmic_tflops=round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 2))
                                for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])})
        pb = roof.get("mfma_busy_pmc")
        if pb and pb.get("mfma_mops_per_launch") and args.dtype == "fp32":
            # the counter twin of `frac`: MFMA ops the PMC pass counted per launch (x 512 FLOPs per v_mfma_f32_32x32x2 "mop")
            # over THIS run's HIP-event time per launch
            roof["frac_pmc"] = round(pb["mfma_mops_per_launch"] * 512.0 / (d["avg_ms"] * 1e-3) / 1e12 / peak, 4)
        tr = _pmc_traffic(key, args.config, args.dtype) if per == cfg_batch and not args.bootstrap else None
        if tr is not None:
            roof["traffic"] = tr["kernel_bytes"]
            roof["traffic_unit"] = "HBM bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, %s)" % tr["source"]
            roof["traffic_provenance"] = {k: tr[k] for k in ("source", "lib_sha256_16", "stale")}
            if tr["step_bytes"]:
                # the whole iteration against the HBM roofline: fabric bytes per iteration / measured iteration time
                roof["hbm"] = dict(step_bytes=tr["step_bytes"], achieved_tbs=round(tr["step_bytes"] / (dt / args.steps) / 1e12, 3),
                                   peak_tbs=PEAK_HBM_TBS, frac=round(tr["step_bytes"] / (dt / args.steps) / 1e12 / PEAK_HBM_TBS, 4),
                                   note="PMC bytes of one iteration (separate profiled run of this workload) over this "
                                        "run's iteration time; bf16 mode is the HBM-borderline one (SURVEY 8d)")
    executed_img = flops_img - (0.0 if args.no_reuse else 2 * fd)  # two decoder forwards replayed, not re-executed
    roof["step"] = dict(algorithmic_gflop_per_image=round(flops_img / 1e9, 1),
                        algorithmic_tflops_per_gpu=round(step_tflops, 2),
                        algorithmic_frac=round(step_tflops / peak, 4),
                        executed_gflop_per_image=round(executed_img / 1e9, 1))
    if summ is not None:
        # what the matrix pipe really ran per image (timer totals: direct kernels count 1:1, Winograd 16/36)
        issued_img = sum(v["executed_flops"] for v in summ.values()) / (per * args.steps)
        roof["step"].update(mfma_issued_gflop_per_image=round(issued_img / 1e9, 1),
                            mfma_issued_tflops_per_gpu=round(issued_img * per * args.steps / dt / 1e12, 2),
                            mfma_issued_frac=round(issued_img * per * args.steps / dt / 1e12 / peak, 4))
    headline = args.config == "celeb256" and gbatch == 128 and not args.bootstrap
    out = {
        # BASELINE.json's metric is quoted at 256x256 with a global batch of 128; any other run says what it ran
        "metric": "training images/sec (whole node) at 256x256 bs128" if headline
        else "training images/sec (whole node) at %dx%d global batch %d" % (image_size, image_size, gbatch),
        "value": round(value, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": head["scaling"], "vs_baseline": None,
        "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": "soft_intro_vae%s %s %dx%d zdim=%d channels=%s, full E-step + D-step iteration"
                               % ("_bootstrap" if args.bootstrap else "", args.config, image_size, image_size, zdim,
                                  channels),
                   "global_batch": gbatch, "per_gpu_batch": per, "parallelism": "dp%d" % world,
                   "betas": {"kl": bk, "rec": br, "neg": bn}, "gamma_r": head["gamma_r"], "lr": 2e-4,
                   "decoder_forward_reuse": not args.no_reuse, "hip_graph": bool(args.hip_graph),
                   "batchnorm": "sync" if head["sync_bn"] else "local",
                   "final_stats": stats},
        "roofline": roof,
    }
    if head["dt_untimed"]:
        out["value_untimed"] = round(gbatch * head["untimed_steps"] / head["dt_untimed"], 3)
        out["value_untimed_note"] = ("%d further iterations of the same engine without the per-launch HIP events; `value` "
                                     "is the instrumented region the roofline object is measured in" % head["untimed_steps"])
    if also is not None:
        ref = out.get("value_untimed", value)
        for k in ("celeb256_fp32_bs16_shard",):
            also[k]["rate_vs_bs128"] = round(also[k]["value"] / ref, 4)
        also["bootstrap256_fp32_bs8_shard"]["rate_vs_bs64"] = round(
            also["bootstrap256_fp32_bs8_shard"]["value"] / also["bootstrap256_fp32_bs64"]["value"], 4)
        out["also"] = also
    for r in results[1:]:
        out[r["scaling"]] = {"value": round(r["gbatch"] * args.steps / r["dt"], 3), "unit": "img/s",
                             "ms_per_step": round(1e3 * r["dt"] / args.steps, 3), "global_batch": r["gbatch"],
                             "per_gpu_batch": r["per"], "scaling": r["scaling"],
                             "algorithmic_tflops_per_gpu": round(flops_img * r["gbatch"] * args.steps / r["dt"] / 1e12 / world, 2)}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_batch, args.cpu_iters, args.cpu_threads)
        out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    if also is not None:
        # the LAST key of the line (a truncated stdout tail still carries it): the shard-size legs and their ratios
        out["zz_shard_summary"] = {
            "headline_img_s": round(out.get("value_untimed", value), 2),
            "celeb256_bs16_shard_img_s": also["celeb256_fp32_bs16_shard"]["value"],
            "rate_vs_bs128": also["celeb256_fp32_bs16_shard"]["rate_vs_bs128"],
            "bootstrap256_bs64_img_s": also["bootstrap256_fp32_bs64"]["value"],
            "bootstrap256_bs8_shard_img_s": also["bootstrap256_fp32_bs8_shard"]["value"],
            "rate_vs_bs64": also["bootstrap256_fp32_bs8_shard"]["rate_vs_bs64"],
            **{k + "_img_s": also[k]["value"] for k in also if k.startswith(("cifar10", "celeb128", "celeb256_bf16"))}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
