#!/usr/bin/env python3
"""bench.py — rays/s of one Pano-NeRF training step on N MI355X (contract: see the task prompt).

A "step" = draw a batch of rays from the HBM-resident synthetic 512x1024 panorama pool -> PanoMipNeRF
forward (128 coarse + 128 fine samples, density-gradient normals, 10x10 env-light gather, Lambertian
surface) -> tone-mapped loss (coarse + fine + surface + chromaticity + orientation) -> backward (first- and
second-order) -> one all-reduce of the 2.455 MB flat gradient (N > 1) -> Adam.  The global batch is fixed
at 4096 rays (BASELINE.json configs[3]/[4]) and split evenly over the ranks: strong scaling.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA (v_mfma_f32_32x32x16_bf16 / _f16), same guide
PEAK_HBM_GBPS = 8000.0  # HBM3E, same guide
DEFAULT_MODE = "fused_f16x2"
# algorithmic GEMM FLOPs (MACs x 2) per ray per train step, Pano N=128
F_PANO, F_GRAD = 1222656.0, 1016320.0


def flop_per_ray_step(n, d=10, ne=10):
    return 3.0 * (2 * n * F_PANO + n * F_GRAD + d * ne * F_PANO)


def analytic_radiance(viewdirs, origins):
    d, o = viewdirs, origins
    f = torch.stack([1.5 + torch.sin(3 * d[:, 0] + o[:, 0]) + torch.cos(2 * d[:, 1]),
                     1.0 + torch.sin(2 * d[:, 1] + o[:, 1]) * torch.cos(d[:, 2]),
                     0.5 + torch.cos(4 * d[:, 2] + o[:, 2]) + d[:, 1]], -1)
    return torch.clamp(torch.nn.functional.softplus(f), 0, 1000).float()


def cpu_baseline(n_samples, rays_cpu, rgbs_cpu, env_cpu, b_cpu, reps=3):
    """The oracle (CPU restatement of the reference, 'faithful' = vmap(jacrev) normals like upstream) timed on the host
    cores: one full-size warm-up step, then the median of `reps` training steps on `b_cpu` rays of the same synthetic
    batch; the 'fast' (grad-of-sum normals) port once, so that the algorithmic and the hardware part of
    the speed-up separate."""
    from oracle import pano_oracle as orc
    params = {k: v.clone().requires_grad_(True) for k, v in orc.init_params(4, 5).items()}
    opt = torch.optim.Adam(list(params.values()), lr=2e-4)

    def step(rays, rgbs, mode):
        gen = torch.Generator().manual_seed(0)
        b, s = rays.origins.shape[0], n_samples + 1
        noise = dict(t_rand=torch.rand(b, s, generator=gen),
                     u_rand=torch.rand(b, s, generator=gen) * (1.0 / s - 1.2e-7),
                     env_rand=torch.rand(1, 11, generator=gen))
        outs = orc.pano_forward(params, rays, env_cpu, num_samples=n_samples, noise=noise, normals_mode=mode)
        loss = orc.pano_loss(outs, rays.lossmult, rgbs)
        opt.zero_grad()
        loss.backward()
        opt.step()

    rays = orc.Rays(*[x[:b_cpu] for x in rays_cpu])
    rgbs = rgbs_cpu[:b_cpu]
    step(rays, rgbs, "faithful")  # warm-up at full size (thread pools, allocator, functorch caches)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        step(rays, rgbs, "faithful")
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    t0 = time.perf_counter()
    step(rays, rgbs, "fast")
    t_fast = time.perf_counter() - t0
    out = {"value": b_cpu / med, "unit": "rays/s", "cores": torch.get_num_threads(), "threads": torch.get_num_threads(),
           "cores_is": "torch intra-op threads the port ran on (= physical cores of the box; host_cpus counts hardware threads)",
           "host_cpus": os.cpu_count(),
           "kind": "port",
           "sample": f"median of {reps} train steps (fwd+bwd+Adam, faithful vmap(jacrev) normals) after 1 warm-up, each on "
                     f"{b_cpu} rays x {n_samples}+{n_samples} samples of the same synthetic batch, fp32",
           "samples_s": [round(t, 3) for t in times],
           "fast_normals_port": {"value": b_cpu / t_fast, "unit": "rays/s", "seconds": round(t_fast, 3),
                                 "note": "same step with grad-of-sum normals (one reverse sweep, the algorithm the HIP path uses)"}}
    cal = os.path.join(ROOT, "tests", "golden", "ref_cpu_timing.json")
    if os.path.exists(cal):  # port-vs-imported-reference speed ratio measured in the build container (8 vCPU)
        try:
            runs = [r for r in json.load(open(cal))["runs"] if r["model"] == "pano"]
            out["port_over_reference_speed"] = max(r["oracle_faithful_over_reference"] for r in runs)
        except Exception:
            pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--global-batch", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--cpu-rays", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inference", action="store_true", help="skip the full-panorama inference leg (N = 1 only)")
    ap.add_argument("--no-cfg2", action="store_true", help="skip the BASELINE configs[1] (256x512, bf16, 512 rays) leg (N = 1 only)")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="capture forward+loss+backward (and Adam when N=1) in one HIP graph and replay it per step; auto = up "
                         "to 2048 rays per GPU (eager fallback if capture fails); off: eager launches, per-launch HIP events "
                         "inside the timed region")
    ap.add_argument("--mlp-mode", choices=("fused", "fused_f16x2", "fused_f16x2_t32", "fused_bf16", "layerwise"),
                    default=os.environ.get("PN_MLP_MODE", DEFAULT_MODE),
                    help="fused_f16x2 (default): on-chip MLP chains, fp16 pairs with power-of-two scaling, 3 partial products "
                         "(fp32-class accuracy); fused: the same kernels with the exact 3-term bf16 split, 6 partial products; "
                         "fused_bf16: plain bf16 operands (BASELINE configs[1]); layerwise: one fp32 GEMM launch per layer")
    ap.add_argument("--streams", default="auto",
                    help="sub-batches of a rank's rays run concurrently on this many HIP streams; auto = 1: with the "
                         "three-product kernels one chain is as fast as two at every size (512 rays: 136.0 k vs 132.7 k "
                         "rays/s, 1024: 151.4 k vs 153.1 k, 2048: 162.1 k vs 162.3 k)")
    ap.add_argument("--overlap", choices=("on", "off"), default="off",
                    help="weight-gradient GEMMs on a side stream beside the next evaluation's chains, each kernel family on its "
                         "share of the CUs (on), or in line on the main stream, one batched job per layer (off)")
    ap.add_argument("--chain-wgs", type=int, default=None, help="--overlap on: workgroups (CUs) of a chain kernel while weight gradients run beside it")
    ap.add_argument("--wgrad-wgs", type=int, default=None, help="--overlap on: CUs of a weight-gradient job while a chain runs beside it")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    import torch.distributed as dist
    # PN_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (ranks then
    # share devices round-robin and the gradient all-reduce goes through the host); never used for reported numbers
    backend = os.environ.get("PN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import pano_nerf_amd as pn
    from pano_nerf_amd import _lib
    from pano_nerf_amd.dist import shard_bounds
    split = False  # (the layer-wise split GEMM mode of round 1 is gone: the fused chains supersede it)
    peak_nt = PEAK_F32_MFMA_TFLOPS
    fused = args.mlp_mode != "layerwise"
    q24_on = args.mlp_mode == "fused_f16x2" and bool(int(_lib.load().pn_chain_q24_slots(2, 1, 0)))
    # fused chains on the 16-bit matrix cores (2.5 PF dense, bf16 and fp16 alike): six partial products per fp32-equivalent
    # product (bf16 three-term split), three (fp16 pair) or one (plain bf16)
    np_ = {"fused": 3, "fused_f16x2": 2, "fused_f16x2_t32": 2, "fused_bf16": 1}.get(args.mlp_mode, 3)  # template argument of the fused kernels
    peak_chain = PEAK_BF16_MFMA_TFLOPS / {3: 6.0, 2: 3.0, 1: 1.0}[np_]
    # names as rocprofv3 prints them (profiles/*_kernel_stats.csv, profiles/r02_pmc_summary.json)
    CLASSES = ((0, "k_gemm_nt"), (1, "k_gemm_tn"), (2, f"k_chain_fwd<{np_}>"), (3, f"k_chain_dgrad<{np_}>"),
               (4, f"k_chain_tangent<{np_}>"), (5, f"k_chain_bwd<{np_}>"),
               (6, f"k_chain_wgrad<{np_}, 2, 4, 4, 2, false, false>"), (7, f"k_chain_wgrad<{np_}, 1, 3, 8, 1, false, false>"),
               (8, f"k_chain_wgrad<{np_}, 1, 3, 4, 3, false, false>"), (9, f"k_chain_wgrad<{np_}, 1, 2, 1, 4, false, false>"),
               (10, f"k_chain_wgrad<{np_}, 1, 1, 1, 4, false, false>"),
               # the 256 x 256 tile with operand tensors in three bytes per element ("Q24": Y only / X and Y)
               (11, f"k_chain_wgrad<{np_}, 2, 4, 4, 2, false, true>"), (12, f"k_chain_wgrad<{np_}, 2, 4, 4, 2, true, true>"))

    def read_prof():
        res = {}
        for cls, name in CLASSES:
            ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
            _lib.load().pn_prof_read(cls, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
            if n.value:
                res[name] = (ms.value, n.value, fl.value)
        return res

    def peak_of(name):
        if name.startswith("k_chain"):
            return peak_chain
        return peak_nt if name == "k_gemm_nt" else PEAK_F32_MFMA_TFLOPS

    # ---- synthetic scene: 3 identity-rotation cameras, analytic HDR radiance, rays made by K1
    torch.manual_seed(4)
    cams = []
    gcpu = torch.Generator().manual_seed(4)
    for _ in range(3):
        m = torch.eye(4)
        m[:3, 3] = torch.rand(3, generator=gcpu) - 0.5
        cams.append(m.numpy())
    # no ray pool is stored: a batch is regenerated from (camera, pixel) by pn_sample_pano_rays; only the
    # target colours live in HBM (the materialised pool below is a temporary to evaluate them once)
    ray_pool = pn.DeviceRayPool(args.height, args.width, cams, near=0.0, far=10.0, device=dev)
    pool = ray_pool.rays
    gt_pool = analytic_radiance(pool.viewdirs, pool.origins)
    del pool
    ray_pool.rgbs = gt_pool
    env = ray_pool.lit_rays(10)
    n_pool = len(ray_pool)

    model = pn.PanoMipNeRF(num_samples=args.samples, rgb_activation="softplus", rgb_padding=0,
                           mlp_num_density_channels=5, num_env_samples=10).to(dev)
    model.mlp_mode = args.mlp_mode
    model.overlap_weight_grads = args.overlap == "on"
    if args.chain_wgs is not None:
        model.overlap_chain_wgs = args.chain_wgs
    if args.wgrad_wgs is not None:
        model.overlap_wgrad_wgs = args.wgrad_wgs
    if world > 1:  # identical replicas
        dist.broadcast(model.mlp.flat_params(), 0)
    opt = pn.FlatAdam(model.mlp, lr=2e-4)
    lo, hi = shard_bounds(args.global_batch, rank, world)
    nb = hi - lo
    torch.manual_seed(1234 + rank)  # every rank draws ITS slice of the global batch and its own jitter noise
    lr_dev = torch.zeros(1, device=dev)

    def fwd_bwd():
        """Sample this rank's rays from the HBM pool, render, loss, backward -> (loss, flat gradient)."""
        rays, gt = ray_pool.sample(nb)
        opt.zero_grad()
        kw = dict(env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
        if state["streams"] > 1:  # the rank's rays as concurrent sub-batches on separate HIP streams (same gradient)
            loss, g, outs = pn.concurrent_step(model, pn.pano_loss, rays, gt, parts=state["streams"], **kw)
            pred = outs[1][0].detach()
            return loss, pred, gt[:pred.shape[0]], g
        outs = model(rays=rays, **kw)
        loss, _ = pn.pano_loss(outs, rays.lossmult, gt)
        loss.backward()
        return loss.detach(), outs[1][0].detach(), gt, model.mlp.last_flat_grad

    graph = None
    n_streams = 1 if args.streams == "auto" else max(1, int(args.streams))
    state = {"streams": n_streams, "replay_check": None, "last_g": None}

    def step(i, local=False):
        """One training step.  `local=True` (rank-0-only measurement legs after the timed region) skips the gradient
        all-reduce: a collective issued by one rank alone would never complete."""
        lr_dev.fill_(pn.mip_lr(i))
        if graph is not None:
            graph.replay()
            loss, pred, gt, g = state["out"]
            if world == 1:
                state["last_g"] = g
                return loss, pred, gt  # Adam is part of the graph
        else:
            loss, pred, gt, g = fwd_bwd()
        if world > 1 and not local:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        state["last_g"] = g
        opt.step_dev(g, lr_dev, grad_scale=1.0 if local else 1.0 / world)
        return loss, pred, gt

    def try_capture():
        """Whole-step capture: ~700 launches become one graph launch (the step is launch-latency sensitive at
        512 rays per GPU).  The decision to replay is COLLECTIVE: a capture that throws on one rank, or a replayed step that
        does not reproduce an eager step on one rank, sends EVERY rank back to eager launches - every rank reaches the same
        all-reduce of the verdict whatever happened locally (a rank that skipped it would pair its next collective, the
        614 k-float gradient all-reduce, with the others' 1-float verdict)."""
        nonlocal graph
        ok, why, check = True, None, None
        try:
            if os.environ.get("PN_BENCH_FAIL_CAPTURE_RANK") == str(rank):  # test hook (tests/test_gpu_bench_dist.py)
                raise RuntimeError("forced capture failure on this rank (PN_BENCH_FAIL_CAPTURE_RANK)")
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    out = fwd_bwd()
                    opt.step_dev(out[3], lr_dev, grad_scale=1.0 / world) if world == 1 else None
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g_ = torch.cuda.CUDAGraph()
            # thread_local: RCCL's helper threads may touch HIP while we capture; they must not invalidate the capture
            with torch.cuda.graph(g_, capture_error_mode="thread_local" if world > 1 else "global"):
                out = fwd_bwd()
                if world == 1:
                    opt.step_dev(out[3], lr_dev, grad_scale=1.0)
            state["out"] = out
            # Self-check (the first RCCL run captures with RCCL's helper threads alive): one replay must reproduce one eager
            # step on the SAME batch - same generator state, hence the same rays and jitter - in loss and flat gradient.
            rng = torch.cuda.get_rng_state(dev)
            l_e, _, _, g_e = fwd_bwd()
            l_e, g_e = l_e.clone(), g_e.clone()
            torch.cuda.set_rng_state(rng, dev)
            g_.replay()
            l_g, g_g = state["out"][0], state["out"][3]
            torch.cuda.synchronize()
            scale = float(g_e.abs().max())
            ok = (bool(torch.isfinite(l_g)) and abs(float(l_g) - float(l_e)) <= 1e-6 * abs(float(l_e))
                  and float((g_g - g_e).abs().max()) <= 1e-6 * scale and scale > 0)
            check = {"loss_eager": float(l_e), "loss_replay": float(l_g),
                     "max_grad_diff_over_max_grad": float((g_g - g_e).abs().max()) / max(scale, 1e-30)}
            if not ok:
                why = f"a replayed step does not reproduce the eager step ({check})"
        except Exception as e:  # this rank cannot replay: the others must learn it through the verdict below
            ok, why, g_ = False, f"graph capture unavailable ({type(e).__name__}: {e})", None
            torch.cuda.synchronize()
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # reached by every rank on every path
        ok_all = bool(flag.item() > 0.5)
        state["replay_check"] = dict(check or {}, ok_all_ranks=ok_all, ok_this_rank=ok, reason=why)
        graph = g_ if ok_all else None
        if not ok_all:
            print(f"[bench] rank {rank}: {why or 'another rank cannot replay the step'}; every rank runs eagerly", file=sys.stderr)
            if args.graph == "on" and not ok and not os.environ.get("PN_BENCH_FAIL_CAPTURE_RANK"):
                raise RuntimeError(why)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Graph replay pays where the step is launch-latency sensitive: 512 rays per GPU 136.0 k rays/s replayed against 132.8 k
    # eager, nothing at 1024 / 2048 (151 k / 162 k either way) or 4096 (163 k either way): `auto` replays up to 2048 rays per
    # GPU and keeps the 4096-ray run eager, so that its launches are timed live with HIP events inside the timed region;
    # it falls back to eager launches if capture fails, or if a replayed step does not reproduce an eager one (try_capture).
    # (Replayed steps had looked up to 17 % faster for a while - the library cleared two small tables with hipMemsetAsync and
    # some replays ran on uncleared tables: the fast runs were the corrupted ones.  The captured graph did hold the edges
    # memset -> accumulating kernel (profiles/r03_graph_memset_nodes.txt); the tables are cleared by a kernel now.)
    use_graph = args.graph == "on" or (args.graph == "auto" and nb <= 2048)
    if use_graph:
        try_capture()  # (falls back to eager launches on every rank together; see there)
    for i in range(args.warmup):
        step(i)
    fence()
    _lib.load().pn_prof_enable(1)
    t0 = time.perf_counter()
    # HIP events cannot be recorded inside a replayed graph, and with concurrent sub-batches a launch's wall duration
    # includes the other chain's share of the matrix cores: in both cases the roofline leg runs right after the region
    prof_live = graph is None and n_streams == 1
    if not prof_live:
        _lib.load().pn_prof_enable(0)
    for i in range(args.steps):
        loss, pred, gt = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    used_graph = graph is not None
    # audit trail of the data-parallel step: every rank's ray count and last loss, and the norm of the gradient block as it
    # stood after the last all-reduce (identical on all ranks by construction; rank 0 reports it)
    audit = {"rays_per_rank": [nb], "last_loss_per_rank": [float(loss)],
             "allreduced_grad_l2": float(state["last_g"].double().norm()) if state["last_g"] is not None else None}
    if world > 1:
        rec = torch.tensor([float(nb), float(loss), audit["allreduced_grad_l2"] or 0.0], device=dev, dtype=torch.float64)
        recs = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(recs, rec)
        audit = {"rays_per_rank": [int(r[0].item()) for r in recs], "last_loss_per_rank": [float(r[1].item()) for r in recs],
                 "allreduced_grad_l2": float(recs[0][2].item()),
                 "allreduced_grad_l2_spread_over_ranks": float(max(r[2].item() for r in recs) - min(r[2].item() for r in recs))}
    prof = read_prof()
    _lib.load().pn_prof_enable(0)
    if not prof_live:  # graph mode: measure the per-launch figures on eager steps right after the timed region
        eager_graph, graph = graph, None
        state["streams"] = 1  # one chain: with concurrent sub-batches a launch's wall duration is not its cost
        step(args.warmup + args.steps)  # one untimed eager step first (the allocator and the caches settle after the replays)
        torch.cuda.synchronize()
        _lib.load().pn_prof_enable(1)
        for i in range(3):
            step(args.warmup + args.steps + 1 + i)
        torch.cuda.synchronize()
        prof = read_prof()
        _lib.load().pn_prof_enable(0)
        graph = None
    # the same kernels with the side stream off: per-launch durations without time-sharing (not part of `value`)
    iso = {}
    if rank == 0 and args.overlap == "on":
        model.overlap_weight_grads = False
        step(args.warmup + args.steps, local=True)
        torch.cuda.synchronize()
        _lib.load().pn_prof_enable(1)
        for i in range(2):
            step(args.warmup + args.steps + 1 + i, local=True)
        torch.cuda.synchronize()
        for name, (ms_, n_, fl_) in read_prof().items():
            iso[name] = {"avg_launch_us": 1e3 * ms_ / max(n_, 1), "tflops": fl_ / max(ms_, 1e-9) / 1e9,
                         "frac": fl_ / max(ms_, 1e-9) / 1e9 / peak_of(name)}
        _lib.load().pn_prof_enable(0)
        model.overlap_weight_grads = args.overlap == "on"
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = args.global_batch * args.steps / elapsed
    psnr = pn.loss.hdr_to_ldr_psnr(pred, gt)

    if not bool(torch.isfinite(loss)):  # a step that ran fast on garbage is not a measurement (seen once: see try_capture)
        raise RuntimeError(f"bench: non-finite loss {float(loss)} after the timed region on rank {rank}")
    if rank == 0:
        dom = max(prof, key=lambda k: prof[k][0])
        ms, n, fl = prof[dom]
        avg_us = 1e3 * ms / max(n, 1)
        achieved = fl / max(ms, 1e-9) / 1e9  # TFLOP/s  (FLOP / ms / 1e9)
        # this path is priced against the MFMA roof.  `achieved` = the 16-bit matrix-core FLOP/s the kernel
        # ISSUES (algorithmic 2 M N K FLOPs x partial products per fp32 product) against the dense 16-bit peak (2.5 PF);
        # `algorithmic_tflops` = the fp32-equivalent figure (frac_algorithmic = that / 2.5 PF).  The HBM view of the same
        # kernel (algorithmic operand bytes / duration against 8 TB/s) is kept as `roofline.hbm`.
        prod = {3: 6.0, 2: 3.0, 1: 1.0}[np_] if dom.startswith("k_chain") else 1.0
        peak_dom = PEAK_BF16_MFMA_TFLOPS if dom.startswith("k_chain") else PEAK_F32_MFMA_TFLOPS
        roof_mfma = {"achieved": achieved * prod, "peak": peak_dom, "unit": "TFLOP/s", "frac": achieved * prod / peak_dom,
                     "algorithmic_tflops": achieved, "frac_algorithmic": achieved / peak_dom,
                     "partial_products_per_fp32_product": prod}
        roof_hbm = None
        if dom.startswith("k_chain_wgrad") and ", 2, 4, 4, 2," in dom:
            # the 256 x 256 weight-gradient tile reads (256 ex + 256 ey) bytes per sample row for 2 * 256 * 256 FLOP: 4 bytes per
            # element of an fp32 tensor, 3 of a Q24 tensor, 2 with bf16 tensors
            esz = 2.0 if args.mlp_mode == "fused_bf16" else 4.0
            ex, ey = (3.0 if dom.endswith("true, true>") else esz), (3.0 if dom.endswith("true>") else esz)
            row_bytes = 256 * (ex + ey)
            gbs = fl / (2.0 * 256 * 256) * row_bytes / max(ms, 1e-9) / 1e6  # algorithmic bytes / time, GB/s
            roof_hbm = {"achieved": gbs, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBPS,
                        "algorithmic_bytes_per_launch": fl / max(n, 1) / (2.0 * 256 * 256) * row_bytes}
        bound = "mfma"  # this path is priced against the MFMA roof (top-level figure); `nearer_roof` says which roof
        top = roof_mfma  # the dominant kernel is actually nearer to (the 256 x 256 weight-gradient tile: HBM)
        nearer = "hbm" if (roof_hbm is not None and roof_hbm["frac"] > roof_mfma["frac"]) else "mfma"
        traffic, pmc_tab, pmc_src = None, {}, None
        for cand in (["r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json"] if fused else ["r01_pmc_summary.json"]):
            pmc = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(pmc):
                try:
                    pmc_tab = json.load(open(pmc))
                    traffic = pmc_tab.get(dom, {}).get("hbm_bytes_per_launch")
                    pmc_src = {"file": "profiles/" + cand, "taken_on": pmc_tab.get("_taken_on"), "note": pmc_tab.get("_note")}
                    break
                except Exception:
                    traffic, pmc_tab, pmc_src = None, {}, None

        def hbm_rate(k, avg_us):
            """HBM GB/s of kernel k: PMC bytes per launch (profiles/r02_pmc_summary.json, same workload) / live duration;
            only at the profiled size (4096 rays on one GPU)."""
            b = pmc_tab.get(k, {}).get("hbm_bytes_per_launch") if (world == 1 and args.global_batch == 4096) else None
            return None if not b else b / max(avg_us, 1e-9) / 1e3
        out = {
            "metric": "rays/sec (train step)", "value": value, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": {"fused": "f32 (3xbf16 split products, fp32 accumulate)",
                      "fused_f16x2": "f32 (2xfp16 split products, fp32 accumulate; weight-gradient operands h_l / delta_l / r_l / hdot_l "
                                     "stored with 16 significant bits, 'Q24')" if q24_on else
                                     "f32 (2xfp16 split products, fp32 accumulate)",
                      "fused_f16x2_t32": "f32 (2xfp16 split products, fp32 accumulate; every stored tensor fp32)",
                      "fused_bf16": "bf16 (fp32 accumulate)"}.get(args.mlp_mode, "f32"), "data": "synthetic",
            "config": {"workload": f"panonerf.yaml train step, synthetic {args.height}x{args.width} pano pool x3 cams, "
                                   f"{args.samples} coarse + {args.samples} fine samples, 10x10 env-light rays, "
                                   f"surface+chrom+ort loss, Adam; global batch {args.global_batch} rays "
                                   f"({hi - lo} per GPU)",
                       "global_batch": args.global_batch, "rays_per_gpu": hi - lo, "num_samples": args.samples,
                       "parallelism": f"dp{world} (rays sharded, one 2.455 MB gradient all-reduce/step)",
                       "launch": ("hip-graph replay" if used_graph else
                                  ("eager (graph capture failed or a replayed step did not reproduce the eager step on some rank: every rank fell back)"
                                   if (state["replay_check"] and not state["replay_check"]["ok_all_ranks"]) else "eager")),
                       "replay_check": state["replay_check"],
                       "data_parallel_audit": audit,
                       "schedule": (f"{n_streams} concurrent sub-batches of {(hi - lo + n_streams - 1) // n_streams} rays on "
                                    f"{n_streams} HIP streams per GPU" if n_streams > 1 else "one chain per GPU"),
                       "mlp_mode": args.mlp_mode,
                       "gemm_mode": ("fused on-chip chains (one kernel per forward / reverse sweep / tangent sweep / "
                                     "backward pass, activations in registers, weights by LDS-DMA ring) + T32 weight-gradient "
                                     "GEMMs, " + {"fused": "x = h + m + l (bf16), six partial products on "
                                                           "v_mfma_f32_*_bf16, fp32 accumulate (fp32 accuracy)",
                                                  "fused_f16x2": "x 2^e = h + l (fp16, |error| < 2^-24 |x|; one power-of-two "
                                                                 "scale per weight matrix and per sample or tensor), three "
                                                                 "partial products on v_mfma_f32_*_f16, fp32 accumulate"
                                                                 + ("; the 256-wide tensors that only the weight-gradient GEMMs read "
                                                                    "back (h0..h6, hdot0..6, delta_l and r_l for l = 1-4, 6, 7) are "
                                                                    "STORED rounded to 16 significant bits, three bytes per element "
                                                                    "(Q24): outputs are unaffected, weight gradients carry a 2^-17 "
                                                                    "relative rounding per operand (mlp_mode fused_f16x2_t32 keeps "
                                                                    "them fp32)" if q24_on else ""),
                                                  "fused_f16x2_t32": "as fused_f16x2 with every stored tensor fp32 (no Q24)",
                                                  "fused_bf16": "plain bf16 operands on v_mfma_f32_*_bf16, fp32 accumulate"
                                                  }.get(args.mlp_mode, ""))
                                    if fused else
                                    ("split: x = h + m + l (bf16), six partial products on v_mfma_f32_32x32x16_bf16, fp32 "
                                     "accumulate; weight gradients on fp32 MFMA" if split else "exact fp32 MFMA")},
            "roofline": {"bound": bound, "nearer_roof": nearer, "kernel": dom, "achieved": top["achieved"],
                         "peak": top["peak"],
                         "unit": top["unit"], "frac": top["frac"],
                         "mfma": roof_mfma, "hbm": roof_hbm,
                         "traffic": traffic, "traffic_source": pmc_src,
                         "avg_launch_us": avg_us, "launches": n,
                         "measured": "HIP events around every GEMM launch " + ("during the timed region" if prof_live else
                                     "on 3 eager single-chain steps (after one untimed) right after the timed region (events cannot be recorded "
                                     "inside a replayed graph, and concurrent sub-batches time-share the matrix cores; same "
                                     "kernels; with sub-batches the timed region's GEMMs have 1/streams of these rows)"),
                         "note": ("timed region runs the weight-gradient GEMMs (k_gemm_tn) on a side stream, concurrently "
                                  "with the k_gemm_nt chain: per-launch durations include time sharing; `isolated` = the "
                                  "same kernels in 2 extra steps with the side stream off.  " if args.overlap == "on" else
                                  "all launches on one stream (no time sharing between kernels).  ") +
                                 ("top-level `achieved` / `peak` / `frac` are the SURVEY.md 8(d) MFMA figure: 16-bit matrix-core "
                                  "FLOP/s issued (algorithmic 2*M*N*K FLOPs per launch x partial products per fp32 product: six "
                                  "for the bf16 three-term split, three for the fp16 pair) against the dense 16-bit MFMA peak "
                                  "(2.5 PF); `mfma.algorithmic_tflops` is the fp32-equivalent rate; `hbm` is the same kernel's "
                                  "algorithmic operand bytes against 8 TB/s; `other[*].tflops` are fp32-equivalent"
                                  if args.mlp_mode in ("fused", "fused_f16x2", "fused_f16x2_t32") else
                                  ("`peak` is the dense bf16 MFMA figure (2.5 PF)" if args.mlp_mode == "fused_bf16" else
                                   "`peak` is the 2.4 GHz datasheet figure; under the power cap the same k_gemm_nt binary runs "
                                   "126 TF on zero/constant operands and 101 TF on N(0,1) operands (tools/experiments/bench_clock.py, "
                                   "profiles/r01_clock_vs_data.txt) and sustains a 2.10 GHz shader clock inside its K loop "
                                   "(tools/experiments/trace_nt.py, profiles/r01_nt_phase_trace_K256.txt), i.e. 137.5 TF are available")),
                         "isolated": iso,
                         "flop_per_launch": fl / max(n, 1),
                         "other": {k: {"total_ms": v[0], "launches": v[1], "avg_launch_us": 1e3 * v[0] / max(v[1], 1),
                                       "tflops": v[2] / max(v[0], 1e-9) / 1e9,
                                       "frac": v[2] / max(v[0], 1e-9) / 1e9 / peak_of(k),
                                       "frac_is": "fp32-equivalent TFLOP/s over (2.5 PF / partial products) = issued 16-bit FLOP/s over 2.5 PF",
                                       "hbm_gbs_from_pmc_traffic": hbm_rate(k, 1e3 * v[0] / max(v[1], 1))}
                                   for k, v in prof.items()},
                         "end_to_end_frac": value / world * flop_per_ray_step(args.samples) /
                                            ((peak_chain if fused else PEAK_F32_MFMA_TFLOPS) * 1e12),
                         "end_to_end_frac_is": "rays/s/GPU x 1.696 GFLOP (SURVEY.md 8d) x partial products / 2.5 PF",
                         "end_to_end_frac_of_fp32_mfma_peak": value / world * flop_per_ray_step(args.samples) /
                                                              (PEAK_F32_MFMA_TFLOPS * 1e12)},
            "psnr_batch_db": psnr, "loss": float(loss),
        }
        if world == 1 and not args.no_inference:
            # inference rays/s for one full panorama: the first camera's H x W rays through render_image
            # (normals + env light + surface shading, 32768-ray chunks), after the timed training region
            hw = args.height * args.width
            cam0 = pn.generate_pano_rays(args.height, args.width, cams[0], 0.0, 10.0, device=dev)
            with torch.no_grad():
                nw = min(4 * 32768, hw)  # warm-up over four chunks: both chunk streams allocate their buffers once
                pn.render_image(model, pn.Rays(*[p[:nw] for p in cam0]), env, 1, nw, chunk_size=32768)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                pn.render_image(model, cam0, env, args.height, args.width, chunk_size=32768)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
            out["inference"] = {"pano": f"{args.height}x{args.width}", "num_samples": args.samples, "chunk_size": 32768,
                                "seconds_per_pano": dt, "rays_per_s": hw / dt}
            # the drop-in the way the REFERENCE calls it (systems/panonerf_system.py:133-192, configs/panonerf.yaml:22
            # val.chunk_size 512): rearrange_render_image(rays, 512), then 1 024 model calls under no_grad on ONE stream,
            # outputs appended per chunk, concatenated and reshaped - the caller's loop, unchanged, restated here line by line
            def reference_loop(chunk):
                img_rays = pn.Rays(*[x.view(1, args.height, args.width, -1) for x in cam0])
                single_image_rays, _ = pn.rearrange_render_image(img_rays, chunk)
                keep = [[] for _ in range(8)]
                with torch.no_grad():
                    for batch_rays in single_image_rays:
                        (c_rgb, c_dep, *_), (f_rgb, f_dep, _, f_nor, alb, rhn, sf_rgb, _, sd) = model(
                            rays=batch_rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
                        for lst, v in zip(keep, (c_rgb, f_rgb, c_dep, f_dep, f_nor, alb, sf_rgb, sd)):
                            lst.append(v)
                dims = (3, 3, 1, 1, 3, 3, 3, 3)
                return [torch.cat(x, 0).view(1, args.height, args.width, dm).permute(0, 3, 1, 2) for x, dm in zip(keep, dims)]
            loop = {}
            for tag, replay in (("replayed", True), ("eager", False)):
                model.replay_inference = replay
                reference_loop(512) if tag == "replayed" else None  # (first pass: captures the chunk's launch sequence)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                imgs = reference_loop(512)
                torch.cuda.synchronize()
                loop[tag] = time.perf_counter() - t1
            model.replay_inference = False
            big = pn.render_image(model, cam0, env, args.height, args.width, chunk_size=32768)
            same = all(torch.equal(a, b) for a, b in zip(imgs, [big[i] for i in (0, 1, 2, 3, 4, 5, 7, 8)]))
            out["inference"]["reference_loop"] = {
                "chunk_size": 512, "model_calls": (hw + 511) // 512,
                "seconds_per_pano": loop["eager"], "rays_per_s": hw / loop["eager"],
                "seconds_per_pano_graph_replay_per_chunk": loop["replayed"],
                "over_big_chunk_render": loop["eager"] / dt,
                "images_bit_identical_to_the_32768_chunk_render": bool(same),
                "note": "the caller's loop of systems/panonerf_system.py:133-192 with val.chunk_size 512, one stream, eager "
                        "launches (the module's default); `graph_replay_per_chunk` = the same loop with "
                        "model.replay_inference = True (a HIP graph captured once per chunk size inside the module, render.py "
                        "_replayed: six input copies + one graph launch + ten output clones per call)"}
        if world == 1 and not args.no_cfg2:
            # BASELINE.json configs[1] (panonerf.yaml, 256x512 pano, 128 samples, bf16) on this GPU, after the timed region
            # and outside `value`: its own 3-camera 256x512 pool, a fresh model in plain-bf16 arithmetic, 512-ray batches
            # (configs/panonerf.yaml:4), 5 timed eager steps after 2 warm-up steps
            pool2 = pn.DeviceRayPool(256, 512, cams, near=0.0, far=10.0, device=dev)
            full2 = pool2.rays
            pool2.rgbs = analytic_radiance(full2.viewdirs, full2.origins)
            del full2
            env2 = pool2.lit_rays(10)
            torch.manual_seed(4)
            m2 = pn.PanoMipNeRF(num_samples=128, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5,
                                num_env_samples=10).to(dev)
            m2.mlp_mode = "fused_bf16"
            opt2 = pn.FlatAdam(m2.mlp, lr=2e-4)

            lr2 = torch.zeros(1, device=dev)

            def fb2():
                r2, g2 = pool2.sample(512)
                opt2.zero_grad()
                o2 = m2(rays=r2, env_rays=env2, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
                l2, _ = pn.pano_loss(o2, r2.lossmult, g2)
                l2.backward()
                opt2.step_dev(m2.mlp.last_flat_grad, lr2, grad_scale=1.0)
                return l2.detach()

            # the same launch mode as the main run at this size (`auto` replays up to 2048 rays per GPU): one HIP graph over
            # sample -> forward -> loss -> backward -> Adam, checked against an eager step on the same batch before it is timed
            g2 = None
            launch2 = "eager"
            if args.graph != "off":
                try:
                    for i in range(2):
                        lr2.fill_(pn.mip_lr(i))
                        fb2()
                    torch.cuda.synchronize()
                    side2 = torch.cuda.Stream(device=dev)
                    side2.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side2):
                        fb2()
                    torch.cuda.current_stream().wait_stream(side2)
                    torch.cuda.synchronize()
                    gg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gg):
                        lcap = fb2()
                    # replay self-check on one batch: restore parameters, Adam state and RNG between the two runs
                    snap = (m2.mlp.flat_params().clone(), opt2.exp_avg.clone(), opt2.exp_avg_sq.clone(), opt2.step_dev_t.clone())
                    rng = torch.cuda.get_rng_state(dev)
                    le = fb2().clone()
                    pe = m2.mlp.flat_params().clone()
                    m2.mlp.flat_params().copy_(snap[0]); opt2.exp_avg.copy_(snap[1]); opt2.exp_avg_sq.copy_(snap[2]); opt2.step_dev_t.copy_(snap[3])
                    torch.cuda.set_rng_state(rng, dev)
                    gg.replay()
                    torch.cuda.synchronize()
                    if float((m2.mlp.flat_params() - pe).abs().max()) <= 1e-6 * float(pe.abs().max()) and abs(float(lcap) - float(le)) <= 1e-6 * abs(float(le)):
                        g2, launch2 = gg, "hip-graph replay"
                    else:
                        launch2 = "eager (a replayed step did not reproduce the eager step)"
                except Exception as e:
                    launch2 = f"eager (graph capture unavailable: {type(e).__name__}: {e})"
                    torch.cuda.synchronize()

            def step2(i):
                lr2.fill_(pn.mip_lr(i))
                if g2 is not None:
                    g2.replay()
                    return lcap
                return fb2()
            for i in range(3):
                step2(2 + i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n2 = 20
            for i in range(n2):
                l2 = step2(5 + i)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            if not bool(torch.isfinite(l2)):
                raise RuntimeError("bench: non-finite loss in the configs[1] (bf16) leg")
            # its roofline: HIP events around every chain / GEMM launch of 3 eager steps (events cannot be recorded in a replay)
            _lib.load().pn_prof_enable(1)
            for i in range(3):
                lr2.fill_(pn.mip_lr(30 + i))
                fb2()
            torch.cuda.synchronize()
            prof2 = {}
            for cls, name in CLASSES:
                name = name.replace(f"<{np_}", "<1")
                ms_, n_, fl_ = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
                _lib.load().pn_prof_read(cls, ctypes.byref(ms_), ctypes.byref(n_), ctypes.byref(fl_))
                if n_.value:
                    prof2[name] = (ms_.value, n_.value, fl_.value)
            _lib.load().pn_prof_enable(0)
            dom2 = max(prof2, key=lambda k: prof2[k][0])
            ms_, n_, fl_ = prof2[dom2]
            tf2 = fl_ / max(ms_, 1e-9) / 1e9
            pmc2 = {}
            try:
                pmc2 = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_summary_cfg2_bf16.json")))
            except Exception:
                pmc2 = {}
            b2 = pmc2.get(dom2, {}).get("hbm_bytes_per_launch")
            step_bytes2 = pmc2.get("_step_total_bytes")
            out["extra"] = {"cfg2_bf16": {"workload": "panonerf.yaml train step, 256x512 pano pool x3 cams, 128 + 128 samples, "
                                                      "512-ray batches, mlp_mode fused_bf16 (bf16 operands, fp32 accumulate)",
                                          "launch": launch2,
                                          "rays_per_s": 512 * n2 / dt2, "ms_per_step": 1e3 * dt2 / n2, "steps": n2, "warmup": 3,
                                          "loss": float(l2),
                                          "roofline": {"bound": "mfma", "kernel": dom2, "achieved": tf2, "peak": PEAK_BF16_MFMA_TFLOPS,
                                                       "unit": "TFLOP/s", "frac": tf2 / PEAK_BF16_MFMA_TFLOPS,
                                                       "avg_launch_us": 1e3 * ms_ / max(n_, 1), "launches": n_,
                                                       "traffic": b2,
                                                       "hbm_gbs_of_that_kernel": (b2 / (1e3 * ms_ / max(n_, 1)) / 1e3) if b2 else None,
                                                       "hbm_gbs_whole_step": (step_bytes2 / (1e3 * dt2 / n2) / 1e6) if step_bytes2 else None,
                                                       "end_to_end_frac": 512 * n2 / dt2 * flop_per_ray_step(128) / (PEAK_BF16_MFMA_TFLOPS * 1e12),
                                                       "measured": "HIP events around every chain / GEMM launch of 3 eager steps after the "
                                                                   "timed replays; bf16 operands: issued = algorithmic FLOPs; traffic from "
                                                                   "profiles/r04_pmc_summary_cfg2_bf16.json (separate --pmc passes of this mode "
                                                                   "at this size) or null",
                                                       "other": {k: {"avg_launch_us": 1e3 * v[0] / max(v[1], 1), "launches": v[1],
                                                                     "tflops": v[2] / max(v[0], 1e-9) / 1e9,
                                                                     "frac": v[2] / max(v[0], 1e-9) / 1e9 / PEAK_BF16_MFMA_TFLOPS}
                                                                 for k, v in prof2.items()}}}}
            del m2, opt2, pool2
        if world == 1 and not args.no_cpu_baseline:
            k = args.cpu_rays
            rays_k, gt_k = ray_pool.take(torch.arange(0, k * 997, 997, device=dev) % n_pool)
            rays_cpu = pn.Rays(*[x.cpu() for x in rays_k])
            gt_cpu = gt_k.cpu()
            from oracle import pano_oracle as orc
            env_cpu = orc.Rays(*[x.cpu() for x in env])
            out["cpu_baseline"] = cpu_baseline(args.samples, orc.Rays(*rays_cpu), gt_cpu, env_cpu, k)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
