#!/usr/bin/env python
"""bench.py -- EMLight regression training throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one synthetic batch: DenseNet-BC forward (HIP
kernels), spherical mover's Sinkhorn loss (HIP), backward (HIP), Adam.  Workload at N=1:
BASELINE configs[1] -- "RegressionNetwork train.py, batch 64, 128 anchors, Sinkhorn blur .05"
on 240x320 crops.  N>1: one process per GPU (torchrun), the batch dimension shards, gradients
all-reduce over RCCL/xGMI (DDP); per-GPU work is fixed -> weak scaling.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the dense-layer
conv1x1 forward/backward family: f32 MFMA-bound), measured live with HIP events on the
launch stream; `cpu_baseline` is the oracle (torch-CPU restatement, parity-pinned to the
reference) timed on this box's host cores at a bounded batch.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_240x320 = 43.775      # algorithmic conv FLOPs per image, forward
STEP_GFLOP_240x320 = 131.2      # forward + dgrad + wgrad (minus conv0 dgrad)
F32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32


def conv1x1_flops(B, crop_hw):
    """Algorithmic FLOPs of ALL dense-layer + transition conv1x1 forward launches of one step."""
    h, w = crop_hw
    c, tot = 24, 0.0
    for _ in range(3):
        for l in range(16):
            tot += 2.0 * (c + 12 * l) * 48 * h * w
        c_tot = c + 192
        tot += 2.0 * c_tot * (c_tot // 2) * (h // 2) * (w // 2)
        c, h, w = c_tot // 2, h // 2, w // 2
    return tot * B


def time_kernel_family(trainer, batch, steps):
    """Average duration of the conv1x1 forward launches of one training step, by bracketing the
    encoder's forward pass kernels with HIP events on the launch stream (no profiler)."""
    from emlight_amd import _lib
    L = _lib.lib()
    orig = L.eml_dense_conv1x1_fwd_f32
    events = []

    class Timed:
        def __call__(self, *a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = orig(*a)
            e1.record()
            events.append((e0, e1))
            return rc
    L.eml_dense_conv1x1_fwd_f32 = Timed()
    try:
        for _ in range(steps):
            trainer.step(batch)
        torch.cuda.synchronize()
    finally:
        L.eml_dense_conv1x1_fwd_f32 = orig
    total_ms = sum(a.elapsed_time(b) for a, b in events)
    return total_ms / steps, len(events) // steps


def cpu_baseline(anchors, crop_hw, blur, batch=2, steps=1):
    """Oracle (port of the reference's PyTorch-CPU maths) training step on the host cores."""
    import oracle
    torch.set_num_threads(os.cpu_count())
    from emlight_amd.RegressionNetwork.data import synthetic_batch
    net = oracle.OracleDenseNet(anchors=anchors, crop_hw=crop_hw).train()
    M = oracle.anchor_cost_matrix(anchors)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    b = synthetic_batch(batch, anchors, crop_hw, seed=1234)
    emd = lambda x, y: oracle.samples_loss(x, y, M, blur=blur)

    def one():
        loss, _ = oracle.regression_loss(net(b["crop"]), b, emd, anchors)
        opt.zero_grad()
        loss.backward()
        opt.step()
    one()  # warm-up
    t0 = time.time()
    for _ in range(steps):
        one()
    dt = (time.time() - t0) / steps
    return {"value": batch / dt, "unit": "images/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d step(s) of batch %d, %dx%d crops, %d anchors, oracle torch-CPU f32, %d threads"
                      % (steps, batch, crop_hw[0], crop_hw[1], anchors, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--anchors", type=int, default=128)
    ap.add_argument("--crop_hw", type=int, nargs=2, default=(240, 320))
    ap.add_argument("--blur", type=float, default=.05)
    ap.add_argument("--engine", default="hip", choices=["hip", "aten"])
    ap.add_argument("--no_cpu_baseline", action="store_true")
    args = ap.parse_args()

    from emlight_amd.RegressionNetwork.engine import RegressionTrainer, init_distributed
    from emlight_amd.RegressionNetwork.data import synthetic_batch
    rank, local, world = init_distributed()
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node %d" % args.gpus
    dev = "cuda:%d" % local
    crop_hw = tuple(args.crop_hw)
    tr = RegressionTrainer(anchors=args.anchors, crop_hw=crop_hw, blur=args.blur, device=dev,
                           engine=args.engine, world=world)
    batch = synthetic_batch(args.batch, args.anchors, crop_hw, seed=1234 + rank, device=dev)

    for _ in range(args.warmup):
        tr.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = args.batch * world * args.steps / dt
        out = {
            "metric": "training images/sec (regression step: DenseNet-BC fwd + Sinkhorn loss + bwd + Adam)",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RegressionNetwork train step, BASELINE configs[1]", "per_gpu_batch": args.batch,
                       "global_batch": args.batch * world, "crop_hw": list(crop_hw), "anchors": args.anchors,
                       "sinkhorn_blur": args.blur, "engine": args.engine,
                       "parallelism": "dp%d" % world if world > 1 else "single"},
            "step_tflops": round(STEP_GFLOP_240x320 * value / 1e3, 2) if crop_hw == (240, 320) else None,
            "step_frac_of_f32_mfma_peak": round(STEP_GFLOP_240x320 * value / world / 1e3 / F32_MFMA_PEAK_TFLOPS, 4)
            if crop_hw == (240, 320) else None,
        }
        if args.engine == "hip":
            fam_ms, n_launch = time_kernel_family(tr, batch, 2)
            fl = conv1x1_flops(args.batch, crop_hw)
            ach = fl / (fam_ms * 1e-3) / 1e12
            out["roofline"] = {"kernel": "conv1x1_fwd_kernel (BN1+ReLU fused f32-MFMA 1x1 conv; %d launches/step)" % n_launch,
                               "bound": "mfma", "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                               "avg_launch_ms": round(fam_ms / max(n_launch, 1), 4)}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.anchors, crop_hw, args.blur)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
