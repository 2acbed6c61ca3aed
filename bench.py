#!/usr/bin/env python
"""bench.py -- EMLight training throughput on MI355X (BASELINE.json metric).

Headline (`value`): the regression training step -- DenseNet-BC forward (HIP kernels), spherical mover's Sinkhorn loss
(HIP), backward (HIP), Adam -- at BASELINE configs[1] ("RegressionNetwork train.py, batch 64, 128 anchors, Sinkhorn
blur .05", 240x320 crops).  The same JSON line carries the other legs of BASELINE's metric ("regression+projector;
Sinkhorn ms/iter") as objects, each timed in this process on the launch stream:
  `projector`  GenProjector G step + D step at configs[2] (B=32/GPU, 128x256)        img/s, fraction of the f32-MFMA roof
  `joint`      regression -> rasteriser -> projector step at configs[3] (32/GPU)     img/s, fraction of the f32-MFMA roof
  `sinkhorn`   ms per eps-step at configs[1] and at configs[4]'s shape (B=16, N=256)  fraction of the 8 TB/s HBM roof
  `rasteriser` SG lobes -> 128x256 panorama at configs[2]'s shape                     ms, GB/s on output bytes, Gexp/s
A "step" is one pass of the hot path over one synthetic batch resident in HBM.

N>1: one process per GPU, the batch dimension shards, gradients all-reduce over RCCL/xGMI (DDP); per-GPU work is
fixed -> weak scaling.  Launch either way:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
    python bench.py --gpus N          (no WORLD_SIZE in the environment: bench.py starts the N ranks itself)

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel family of the headline step, measured live with
HIP events on the launch stream; `cpu_baseline` is the oracle (torch-CPU restatement, parity-pinned to the reference)
timed on this box's host cores at a bounded batch.
"""
import argparse
import json
import os
import sys
import time

# kernel arguments in device memory: read by the HIP runtime when it initialises, so chosen here, before torch touches the GPU
# (emlight_amd/_runtime.py: an entry point's choice, never a side effect of importing the package)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from emlight_amd import _runtime  # noqa: E402
_runtime.entry_point_defaults()

import glob
import socket
import subprocess

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_240x320 = 43.775      # algorithmic conv FLOPs per image, forward
STEP_GFLOP_240x320 = 131.2      # forward + dgrad + wgrad (minus conv0 dgrad)
# The engine pools BEFORE the three transition convolutions (pool_act_kernel: the 1x1 convolution commutes with the 2x2 mean),
# so a quarter of their algorithmic FLOPs (DenseNet.py:14-21 counts them at the pre-pool resolution: 5.87 G forward, x3 in a
# step) is what the matrix pipe executes.  `value * STEP` is the algorithmic credit SURVEY 8d defines; `EXECUTED` is the honest
# utilisation of the pipe -- both are on the line.
TRANSITION_FWD_GFLOP_240x320 = 2.0 * (216 * 108 * 120 * 160 + 300 * 150 * 60 * 80 + 342 * 171 * 30 * 40) * 4 / 1e9   # at pre-pool pixels
EXECUTED_STEP_GFLOP_240x320 = STEP_GFLOP_240x320 - 3 * 0.75 * TRANSITION_FWD_GFLOP_240x320
F32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E, 8 stacks


def conv_flops(B, crop_hw):
    """Algorithmic FLOPs (2*Cout*Cin*k*k*Hout*Wout) of one pass over all conv1x1 launches
    (dense bottlenecks + transitions) and over all conv3x3 launches of the encoder, for batch B."""
    h, w = crop_hw
    c, f1, f3 = 24, 0.0, 0.0
    for _ in range(3):
        for l in range(16):
            f1 += 2.0 * (c + 12 * l) * 48 * h * w
            f3 += 2.0 * 48 * 12 * 9 * h * w
        c_tot = c + 192
        f1 += 2.0 * c_tot * (c_tot // 2) * (h // 2) * (w // 2)
        c, h, w = c_tot // 2, h // 2, w // 2
    return f1 * B, f3 * B


def family_bytes(B, crop_hw):
    """Algorithmic HBM bytes per training step of every kernel family (DESIGN.md section 3 states the
    per-pixel figures): each operand a launch must read or write once, f32, nothing for halos or re-reads."""
    h, w = crop_hw
    c = 24
    by = {k: 0.0 for k in FAMILIES}
    for _ in range(3):
        P = float(B * h * w)
        for l in range(16):
            k = c + 12 * l
            kp = (k + 15) // 16 * 16
            by["eml_dense_conv1x1_fwd_f32"] += ((k + 48) * 4 + kp / 8) * P   # X[:, :k] in, Z out, ReLU bit mask out
            by["eml_dense_conv1x1_bwd_weight_f32"] += (k + 96 + 48) * 4 * P   # X[:, :k], dzn, Z in; dz out (materialised)
            if l % 2 == 1:   # upper layer of a pair: the narrow pass rides on it (G slice in, compact N12 out)
                by["eml_dense_conv1x1_bwd_weight_f32"] += (12 + 12) * 4 * P
            by["eml_dense_conv3x3_fwd_f32"] += (48 + 12) * 4 * P         # Z in, 12 new channels out
            # G, X (deferred affine), Z in; dzn, GF out; the lower layer of a pair also reads the compact N12
            by["eml_dense_conv3x3_bwd_data_f32"] += (12 + 12 + 48 + 48 + 12 + (12 if l % 2 == 0 else 0)) * 4 * P
            by["eml_dense_conv3x3_bwd_weight_f32"] += (12 + 48) * 4 * P  # dY, Z in
            # the two above in one pass (round 4): every operand once -- Z serves the statistics and the weight gradient
            by["eml_dense_conv3x3_bwd_fused_f32"] += (12 + 12 + 48 + 48 + 12 + (12 if l % 2 == 0 else 0)) * 4 * P
            if l % 2 == 0:  # layers (l+1, l): narrow pass over l's 12 output channels, fused pass over [0, k)
                # old G in, new G out, 2 x dz in, the two layers' ReLU bit masks in (round 2, late: X is no longer read)
                by["eml_dense_conv1x1_bwd_data_multi_f32"] += ((2 * k + 96) * 4 + 2 * kp / 8) * P
        ct = c + 192
        by["eml_dense_pool_act_f32"] += ((ct + ct // 4) * 4 + ct / 8) * P   # transition: X in, pooled activation A + ReLU bits out
        by["eml_dense_conv1x1_fwd_f32"] += (ct // 4 + ct // 8) * 4 * P   # transition conv on A: A in, pooled out
        by["eml_dense_conv1x1_bwd_weight_f32"] += (ct // 4 + ct // 8) * 4 * P
        by["eml_dense_conv1x1_bwd_data_f32"] += ((ct + ct // 8) * 4 + ct / 8) * P   # dY(pooled), ReLU bits in, G out (no X)
        c, h, w = ct // 2, h // 2, w // 2
    return by


def wgrad_operand_minimal_bytes(B, crop_hw):
    """What a 1x1 weight gradient needs at the very least: x and dz once.  `family_bytes` books the
    BN2-backward rebuild that is fused into the kernel (dzn and Z in, dz out) to it as well: (k + 144) floats per pixel against
    the (k + 48) counted here -- the bench line carries the fraction of the HBM roof on BOTH counts."""
    h, w = crop_hw
    c, tot = 24, 0.0
    for _ in range(3):
        P = float(B * h * w)
        tot += sum((c + 12 * l + 48) * 4 * P for l in range(16))
        ct = c + 192
        tot += (ct // 4 + ct // 8) * 4 * P       # transition: pooled activation A in, pooled dY in
        c, h, w = ct // 2, h // 2, w // 2
    return tot


# launcher -> (kernel family, which algorithmic FLOP count one pass over its launches performs)
FAMILIES = {
    "eml_dense_conv1x1_fwd_f32": ("conv1x1_fwd_kernel (BN1+ReLU fused into the MFMA operand load)", 0),
    "eml_dense_conv1x1_bwd_weight_f32": ("conv1x1_bwd_weight_kernel (wgrad, dz rebuilt in LDS and materialised; the upper "
                                         "layer of a pair also runs the narrow data pass on that dz tile)", 0),
    "eml_dense_conv1x1_bwd_data_multi_f32": ("conv1x1_bwd_data_multi_kernel (dgrad of 2 dense layers per pass; ReLU mask "
                                             "from the forward's bits, no x read; BN1-backward accumulate)", 0),
    "eml_dense_conv1x1_bwd_data_f32": ("transition_bwd_data_kernel (transition dgrad: un-pool, ReLU mask, BN backward accumulate)", 2),
    "eml_dense_pool_act_f32": ("pool_act_kernel (transition operand: 2x2 mean of relu(bn(x)))", 3),
    "eml_dense_conv3x3_fwd_f32": ("conv3x3 forward: conv3x3_fwd_tp_kernel (block 1: the 9 taps x 12 channels as MFMA rows, 84 "
                                  "MFMAs per 16 pixels, shift-and-add of accumulators) + conv3x3_fwd_kernel (blocks 2, 3: halo tile, 108)", 1),
    "eml_dense_conv3x3_bwd_data_f32": ("conv3x3_bwd_data_kernel", 1),
    "eml_dense_conv3x3_bwd_weight_f32": ("conv3x3_bwd_weight_kernel", 1),
    "eml_dense_conv3x3_bwd_fused_f32": ("conv3x3_bwd_fused_tp_kernel (data gradient + weight gradient of a layer in one pass over "
                                        "the tiles: replaces the two rows above, whose algorithmic bytes / FLOPs then count "
                                        "for launches that did not happen)", 4),
}


# launchers whose calls are booked to another launcher's family (same layer, another kernel)
FAMILY_ALIASES = {"eml_dense_conv3x3_fwd_tp_f32": "eml_dense_conv3x3_fwd_f32",
                  # the same pass with its top 24 columns leaving as compact tensors (same bytes, other addresses)
                  "eml_dense_conv1x1_bwd_data_multi_top_f32": "eml_dense_conv1x1_bwd_data_multi_f32"}


def time_kernel_families(trainer, batch, steps, B, crop_hw):
    """Per-family GPU time of one training step, measured live: every dense-engine launcher call is
    bracketed by HIP events on the stream it launches on (torch's current stream)."""
    from emlight_amd import _lib
    L = _lib.lib()
    events = {k: [] for k in FAMILIES}
    dispatches = {k: 0 for k in FAMILIES}
    orig = {k: getattr(L, k) for k in list(FAMILIES) + list(FAMILY_ALIASES)}
    # kernel dispatches of the family's MAIN kernel per launcher call: the 1x1 launchers run one dispatch per 48-wide
    # output chunk (a transition's 108 / 150 / 171 output channels = 3 / 4 / 4 dispatches); Cout is argument 10 / 17
    chunks = {"eml_dense_conv1x1_fwd_f32": lambda a: (a[10] + 47) // 48,
              "eml_dense_conv1x1_bwd_weight_f32": lambda a: (a[17] + 47) // 48}

    def timed(launcher):
        name = FAMILY_ALIASES.get(launcher, launcher)
        fn, nd = orig[launcher], chunks.get(launcher)

        def call(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            events[name].append((e0, e1))
            dispatches[name] += nd(a) if nd else 1
            return rc
        return call
    for k in orig:
        setattr(L, k, timed(k))
    try:
        for _ in range(steps):
            trainer.step(batch)
        torch.cuda.synchronize()
    finally:
        for k in orig:
            setattr(L, k, orig[k])
    f1, f3 = conv_flops(B, crop_hw)
    h, w = crop_hw
    ftr, c = 0.0, 24
    for _ in range(3):  # transitions only (they are part of f1 as well)
        ct = c + 192
        ftr += 2.0 * ct * (ct // 2) * (h // 2) * (w // 2) * B
        c, h, w = ct // 2, h // 2, w // 2
    flops = (f1, f3, ftr, 0.0, 2.0 * f3)
    nbytes = family_bytes(B, crop_hw)
    rows = []
    for k, (label, which) in FAMILIES.items():
        ms = sum(a.elapsed_time(b) for a, b in events[k]) / steps
        n = len(events[k]) // steps
        nd = dispatches[k] // steps
        rows.append({"kernel": label, "launches_per_step": n, "dispatches_per_step": nd, "ms_per_step": round(ms, 3),
                     "avg_launch_ms": round(ms / max(n, 1), 4), "avg_dispatch_ms": round(ms / max(nd, 1), 4),
                     "tflops": round(flops[which] / (ms * 1e-3) / 1e12, 2) if ms > 0 and flops[which] else None,
                     "algorithmic_GB_per_step": round(nbytes[k] / 1e9, 2),
                     "algorithmic_GBps": round(nbytes[k] / (ms * 1e-3) / 1e9, 1) if ms > 0 else None})
        if k == "eml_dense_conv1x1_bwd_weight_f32":
            mb = wgrad_operand_minimal_bytes(B, crop_hw)
            rows[-1]["operand_minimal_GB_per_step"] = round(mb / 1e9, 2)
            rows[-1]["operand_minimal_GBps"] = round(mb / (ms * 1e-3) / 1e9, 1) if ms > 0 else None
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def pmc_traffic(kernel_label):
    """HBM bytes per launch of a kernel family from the newest committed PMC summary (`profiles/rNN_pmc_summary.csv`:
    separate rocprofv3 --pmc passes of this same command; FETCH_SIZE under-reports 16-B/lane reads 2x on gfx950) and
    where that number comes from (file + the commit stamped in its sidecar), or (None, None) if not collected."""
    import csv
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.csv")))
    if not files:
        return None, None
    path = files[-1]
    name = kernel_label.split(" ")[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["kernel"].startswith(name):
            d = int(r["dispatches"])
            tot += d * (2.0 * float(r["mean_FETCH_SIZE"]) + float(r["mean_WRITE_SIZE"])) * 1024.0
            n += d
    src = {"file": os.path.relpath(path, ROOT), "collected_at_commit": None,
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch, mean over the family's dispatches"}
    meta = path[:-4] + ".meta.json"
    if os.path.exists(meta):
        src.update(json.load(open(meta)))
    return (round(tot / n) if n else None), src


def _events(fn, reps):
    """Mean GPU milliseconds of `fn()` over `reps` back-to-back calls, HIP events on torch's current stream (the stream
    every launcher of this library enqueues on)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def time_sinkhorn(B, anchors, blur, dev, reps=50):
    """BASELINE's second metric: Sinkhorn ms per eps-step (1 iter = 4 softmin sweeps + averaging,
    sinkhorn_divergence.py:87-97) of the HIP loss.  Outputs are pre-allocated once, so the `reps` loss calls are
    back-to-back kernel launches (loop kernel + finishing kernel) between two HIP events -- the figure is the kernels'."""
    from emlight_amd.RegressionNetwork.geomloss import SamplesLoss
    from emlight_amd.RegressionNetwork.geomloss.samples_loss import sinkhorn_outputs
    g = torch.Generator().manual_seed(7)
    x = torch.softmax(torch.randn(B, anchors, generator=g), 1).view(B, anchors, 1).to(dev)
    y = torch.softmax(3 * torch.randn(B, anchors, generator=g), 1).view(B, anchors, 1).to(dev)
    crit = SamplesLoss("sinkhorn", p=2, blur=blur, anchors=anchors)
    out = sinkhorn_outputs(B, anchors, dev, True, True)
    n_eps = int(crit.forward_raw(x, y, out=out)["n_eps"].item())
    ms = _events(lambda: crit.forward_raw(x, y, out=out), reps)
    alg_bytes = 4.0 * (B * anchors * anchors * 4 + 2 * B * anchors * 4)  # per eps-step, SURVEY 8d
    per_step = ms / (n_eps + 2)
    return {"batch": B, "anchors": anchors, "blur": blur, "ms_per_loss_call": round(ms, 4), "n_eps": n_eps,
            "sweeps": n_eps + 2, "ms_per_eps_step": round(per_step, 5),
            "algorithmic_GBps": round(alg_bytes / (per_step * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak_8TBps": round(alg_bytes / (per_step * 1e-3) / 8e12, 4),
            "note": "loss call = schedule + loop + finish kernels (forward and unit gradients), outputs pre-allocated; "
                    "algorithmic bytes = the reference's materialised-cost traffic 4*(B*N*N*4 + 2*B*N*4) per eps-step"}


def time_rasteriser(B, anchors, pano_hw, dev, reps=50):
    """SG lobes -> equirect panorama (util.py:222-245) at configs[2]'s shape: one launch, B*3*H*W*4 output bytes +
    B*7N*4 input bytes (12.6 MB at B=32), B*N*H*W exponentials; forward and the colour gradient."""
    from emlight_amd.RegressionNetwork.util import convert_to_panorama, sphere_points
    H, W = pano_hw
    g = torch.Generator().manual_seed(11)
    dirs = torch.from_numpy(sphere_points(anchors)).float().view(1, 3 * anchors).repeat(B, 1).to(dev)
    sizes = torch.full((B, anchors), 0.0025, device=dev)
    colors = torch.rand(B, 3 * anchors, generator=g).to(dev).requires_grad_(True)
    ms_api = _events(lambda: convert_to_panorama(dirs, sizes, colors.detach(), pano_hw=pano_hw), reps)
    # the launch itself, back to back into a pre-allocated panorama through the C ABI (like the Sinkhorn legs): at < 20 us per
    # launch the Python entry point above (a torch.empty + the autograd node per call) is bound by the host, not by the kernel
    from emlight_amd import _lib
    pano, cd = torch.empty(B, 3, H, W, device=dev), colors.detach().contiguous()
    L_, p_ = _lib.lib(), _lib.ptr
    ms_f = _events(lambda: _lib.check(L_.eml_sg_rasterise_f32(p_(dirs), p_(sizes), p_(cd), p_(pano), B, anchors, H, W,
                                                              _lib.current_stream()), "eml_sg_rasterise_f32"), reps)
    out = convert_to_panorama(dirs, sizes, colors, pano_hw=pano_hw)
    gout = torch.rand_like(out)
    ms_b = _events(lambda: torch.autograd.grad(out, colors, gout, retain_graph=True), reps)
    # the colour gradient's two launches (per-patch light lists + the tile sum) back to back, and the round-1 kernel beside it
    work = torch.empty(max(1, L_.eml_sg_rasterise_bwd_work_floats(B, anchors, H, W)), device=dev)
    gcol, gq = torch.empty(B, 3 * anchors, device=dev), gout.contiguous()
    ms_bl = _events(lambda: _lib.check(L_.eml_sg_rasterise_bwd_colors_ex_f32(p_(dirs), p_(sizes), p_(gq), p_(gcol), p_(work), B,
                                                                             anchors, H, W, 0, _lib.current_stream()),
                                       "eml_sg_rasterise_bwd_colors_ex_f32"), reps)
    ms_b1 = _events(lambda: _lib.check(L_.eml_sg_rasterise_bwd_colors_f32(p_(dirs), p_(sizes), p_(gq), p_(gcol), B, anchors, H, W,
                                                                          _lib.current_stream()),
                                       "eml_sg_rasterise_bwd_colors_f32"), reps)
    nbytes = B * 3 * H * W * 4 + B * 7 * anchors * 4
    nexp = float(B) * anchors * H * W
    from emlight_amd.RegressionNetwork.util import rasterise_raw
    _, executed = rasterise_raw(dirs, sizes, colors.detach(), pano_hw=pano_hw, count=True)
    return {"batch": B, "anchors": anchors, "pano_hw": [H, W], "ms_fwd": round(ms_f, 4), "ms_fwd_python_entry": round(ms_api, 4),
            "ms_bwd_colors": round(ms_b, 4), "ms_bwd_colors_launches": round(ms_bl, 4),
            "ms_bwd_colors_round1_kernel": round(ms_b1, 4),
            "algorithmic_MB": round(nbytes / 1e6, 2), "GBps_on_algorithmic_bytes": round(nbytes / (ms_f * 1e-3) / 1e9, 1),
            "reference_exponentials": int(nexp), "executed_exponentials": executed,
            "executed_fraction": round(executed / nexp, 4),
            "Gexp_per_s_reference": round(nexp / (ms_f * 1e-3) / 1e9, 1),
            "Gexp_per_s_executed": round(executed / (ms_f * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak_8TBps": round(nbytes / (ms_f * 1e-3) / 8e12, 4),
            "note": "ms_fwd = eml_sg_rasterise_f32 launched back to back into a pre-allocated panorama (HIP events); "
                    "ms_fwd_python_entry = through convert_to_panorama (allocation + autograd node per call: host-bound at this "
                    "size).  Instruction-bound (exp2 + fma per surviving (pixel, light) pair), not HBM-bound: 12 B per pixel leave; "
                    "per 16x8-pixel patch only the lights whose lobe can be non-zero there are evaluated (hierarchical "
                    "cull, bit-identical to the exhaustive loop): reference_exponentials = B*N*H*W is what "
                    "util.py:239-244 evaluates, executed_exponentials what this launch did (device counter)"}


def _cpu_baseline_worker(anchors, crop_hw, blur, batch, threads, q):
    import oracle
    torch.set_num_threads(threads)
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
    t0 = time.time()
    one()  # warm-up (also the fallback sample if the box is slow)
    warm = time.time() - t0
    steps, dt = 1, warm
    if warm < 8.0:
        steps = 2
        t0 = time.time()
        for _ in range(steps):
            one()
        dt = (time.time() - t0) / steps
    q.put((batch / dt, steps))


def cpu_baseline(anchors, crop_hw, blur, batch=2, budget_s=150):
    """Oracle (port of the reference's PyTorch-CPU maths) training step on this box's host cores:
    a BOUNDED sample (batch 2, <= 3 steps) in a child process with a wall-clock budget.  Thread count
    is capped at 32: at batch 2 the reference's ATen CPU kernels slow down, not up, beyond that
    (measured: 256 threads -> 292 s per step on the MI355X host).  Deviation from SURVEY 8d's protocol (B=4, 1+3
    steps, all cores), chosen so that the default bench run stays within minutes; recorded in DESIGN.md section 5."""
    import multiprocessing as mp
    threads = min(os.cpu_count() or 1, 32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_cpu_baseline_worker, args=(anchors, crop_hw, blur, batch, threads, q))
    pr.start()
    pr.join(budget_s)
    if pr.is_alive():
        pr.terminate()
        pr.join()
        return {"value": None, "unit": "images/s", "cores": threads, "kind": "port",
                "sample": "oracle step of batch %d did not finish within %d s" % (batch, budget_s)}
    try:
        value, steps = q.get(timeout=10)
    except Exception:
        return {"value": None, "unit": "images/s", "cores": threads, "kind": "port",
                "sample": "oracle child process exited without a result (exit code %s)" % pr.exitcode}
    return {"value": round(value, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "%d timed step(s) of batch %d, %dx%d crops, %d anchors, oracle torch-CPU f32 training step "
                      "(fwd + Sinkhorn + bwd + Adam), %d threads of %d host cpus"
                      % (steps, batch, crop_hw[0], crop_hw[1], anchors, threads, os.cpu_count() or 1)}


def run_timed(step_fn, steps, warmup, world, device):
    """`warmup` untimed steps, then EXACTLY `steps` timed ones bracketed by barrier + device sync on both
    sides; returns the MAX over ranks of the elapsed seconds (identical on every rank)."""
    is_cuda = str(device).startswith("cuda")
    sync = torch.cuda.synchronize if is_cuda else (lambda: None)
    for _ in range(warmup):
        step_fn()
    sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device if is_cuda else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


G_FWD_GFLOP, D_PAIR_GFLOP = 154.1, 4.71   # SPADE generator forward / discriminator forward on a fake+real pair
# VGG19 to relu5_1 at 128x256, algorithmic conv FLOPs per image forward (2*Cout*Cin*9*H*W over the 13 convolutions):
# 0.113 + 2.416 + 2 * (1.208 + 2.416) ... = 23.6; the G step runs it on fake and on real and takes the data gradient
# through fake (loss.py:108-114, pix2pix_model.py:119-120)
VGG_FWD_GFLOP = sum(2.0 * co * ci * 9 * (128 >> lvl) * (256 >> lvl) for lvl, ci, co in
                    [(0, 3, 64), (0, 64, 64), (1, 64, 128), (1, 128, 128), (2, 128, 256), (2, 256, 256), (2, 256, 256),
                     (2, 256, 256), (3, 256, 512), (3, 512, 512), (3, 512, 512), (3, 512, 512), (4, 512, 512)]) / 1e9
# G step: G fwd + bwd (3x) and D fwd/bwd-data on the pair (2x); D step: G fwd (no grad) + D fwd/bwd (3x)
PROJECTOR_STEP_GFLOP = 3 * G_FWD_GFLOP + 2 * D_PAIR_GFLOP + G_FWD_GFLOP + 3 * D_PAIR_GFLOP
VGG_STEP_GFLOP = 3 * VGG_FWD_GFLOP


# projector launchers -> (label, FLOP count of one call from its arguments, or None for the streaming kernels); launchers
# with the same label share a row
_FWD_LABEL = ("gather-GEMM forward (gather_gemm2_kernel / sphere_conv_fwd_fused_kernel: taps gathered into the LDS operand of an "
              "f32-MFMA implicit GEMM: SphereConv2D forward with the residual sum / ReLU or SPADE's modulation in the epilogue; "
              "VGG19's 3x3 convolutions use it with a planar tap table)")
_SPADE_BWD_LABEL = "spade_norm_modulate_bwd_kernel (+ BatchNorm partials, + the gamma|beta bias gradient)"
_NARROW_LABEL = "sphere_conv_narrow_{fwd,dgrad,wgrad}_kernel (conv_img 64 -> 3, the discriminators' heads: one pass each way)"
PROJECTOR_FAMILIES = {
    "eml_sphere_conv_fwd_fused_ex_f32": (_FWD_LABEL, lambda a: 2.0 * a[6] * a[8] * 9 * a[9] * a[10]),     # B * Po * 9C * O
    # the same kernel with SPADE's modulation as its epilogue (the gamma | beta heads: O = 2 * Cn): same family row
    "eml_sphere_conv_spade_fwd_f32": (_FWD_LABEL, lambda a: 2.0 * a[10] * a[11] * a[12] * 9 * a[13] * 2 * a[14]),
    "eml_sphere_conv_dgrad_fused_f32": ("sphere_conv_fwd_fused_kernel on the transposed tap table (input gradient)",
                                        lambda a: 2.0 * a[7] * a[8] * 9 * a[10] * a[11]),  # B * HW * 9C * O
    "eml_sphere_conv_wgrad_fused_f32": ("sphere_conv_wgrad_fused_kernel (weight gradient, split-K over pixels)",
                                        lambda a: 2.0 * a[6] * a[8] * 9 * a[9] * a[10]),
    "eml_sphere_im2col_f32": ("sphere_im2col_kernel (gather for the layers that keep the library GEMM)", None),
    "eml_sphere_col2im_f32": ("sphere_col2im_kernel (transpose gather of the unfused input gradient)", None),
    "eml_spade_norm_modulate_fwd_f32": ("spade_norm_modulate_fwd_kernel (BN + modulation + LeakyReLU)", None),
    "eml_spade_norm_modulate_up2_fwd_f32": ("spade_norm_modulate_fwd_kernel<up2> (the block's x2 upsample folded in)", None),
    "eml_bn_bwd_apply_up2_f32": ("bn_bwd_apply_up2_kernel", None),
    "eml_spade_norm_modulate_bwd_cols_f32": (_SPADE_BWD_LABEL, None),
    "eml_spade_norm_modulate_bwd_y_f32": (_SPADE_BWD_LABEL, None),
    "eml_sphere_conv_narrow_fwd_f32": (_NARROW_LABEL, None),
    "eml_sphere_conv_narrow_dgrad_f32": (_NARROW_LABEL, None),
    "eml_sphere_conv_narrow_wgrad_f32": (_NARROW_LABEL, None),
    "eml_bn_stats_f32": ("bn_stats_kernel (SPADE batch statistics)", None),
    "eml_bn_bwd_apply_f32": ("bn_bwd_apply_kernel", None),
    "eml_sphere_conv_small_fwd_f32": ("sphere_conv_small_fwd_kernel (3 -> 64/128 input layers + ReLU, one pass)", None),
    "eml_sphere_conv_small_wgrad_f32": ("sphere_conv_small_wgrad_kernel (their dW, db and ReLU backward, one pass)", None),
    "eml_instance_norm_act_fwd_f32": ("instance_norm_*_kernel<fwd> (InstanceNorm + LeakyReLU of the discriminator / encoder)", None),
    "eml_instance_norm_act_bwd_f32": ("instance_norm_*_kernel<bwd>", None),
}


# encoder launchers as they appear in the JOINT step's table: (label, FLOP count of one call from its arguments)
ENCODER_FAMILIES = {
    "eml_dense_conv1x1_fwd_f32": ("encoder conv1x1_fwd_kernel", lambda a: 2.0 * a[2] * a[6] * a[10]),                  # P * Kp * Cout
    "eml_dense_conv1x1_bwd_weight_f32": ("encoder conv1x1_bwd_weight_kernel", lambda a: 2.0 * a[2] * a[7] * a[17]),  # P * Cin * Cout
    "eml_dense_conv1x1_bwd_data_multi_f32": ("encoder conv1x1_bwd_data_multi_kernel (two layers per pass)",
                                             lambda a: 2.0 * a[0] * a[15] * (a[17] - a[16]) * 48),      # layers * P * channels * 48
    "eml_dense_conv1x1_bwd_data_multi_top_f32": ("encoder conv1x1_bwd_data_multi_kernel (two layers per pass)",
                                                 lambda a: 2.0 * 2 * a[6] * a[7] * 48),                   # layers * P * k_hi * 48
    "eml_dense_conv1x1_bwd_data_f32": ("encoder transition_bwd_data_kernel", lambda a: 2.0 * a[15] * a[19] * a[7]),  # P * Kp * Ko
    "eml_dense_conv3x3_fwd_f32": ("encoder conv3x3 forward (tap-packed in block 1)", lambda a: 2.0 * a[7] * a[8] * a[9] * 9 * 48 * 12),
    "eml_dense_conv3x3_fwd_tp_f32": ("encoder conv3x3 forward (tap-packed in block 1)", lambda a: 2.0 * a[7] * a[8] * a[9] * 9 * 48 * 12),
    "eml_dense_conv3x3_bwd_data_f32": ("encoder conv3x3_bwd_data_kernel", lambda a: 2.0 * a[8] * a[9] * a[10] * 9 * 48 * 12),
    "eml_dense_conv3x3_bwd_weight_f32": ("encoder conv3x3_bwd_weight_kernel", lambda a: 2.0 * a[6] * a[7] * a[8] * 9 * 48 * 12),
    "eml_dense_conv3x3_bwd_fused_f32": ("encoder conv3x3_bwd_fused_tp_kernel (data + weight gradient in one pass, taps packed into the MFMA rows)",
                                        lambda a: 4.0 * a[8] * a[9] * a[10] * 9 * 48 * 12),
}


def time_projector_families(trainer, data, steps, families=None, other_label=None):
    """Per-family GPU time of one projector (or joint) step (HIP events around every launcher call, on the stream it
    launches on), sorted by time; `other` = the step's remaining GPU time (library GEMMs of the unfused layers, ATen glue,
    Adam).  `families`: launcher -> (label, flops-of-a-call or None); default: the projector's."""
    from emlight_amd import _lib
    L = _lib.lib()
    FAM = families or PROJECTOR_FAMILIES
    events = {k: [] for k in FAM}
    flops = {k: 0.0 for k in FAM}
    orig = {k: getattr(L, k) for k in FAM}

    def timed(name):
        fn, fl = orig[name], FAM[name][1]

        def call(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            events[name].append((e0, e1))
            if fl is not None:
                flops[name] += fl(a)
            return rc
        return call
    for k in FAM:
        setattr(L, k, timed(k))
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the instrumented iterations run the discriminator step's generator pass eagerly: a replayed graph makes no launcher calls
    from emlight_amd.GenProjector.pix2pix_model import Pix2PixModel
    graph_was = Pix2PixModel.graph_dstep
    Pix2PixModel.graph_dstep = False
    try:
        torch.cuda.synchronize()
        t0.record()
        for _ in range(steps):
            trainer.step(data)
        t1.record()
        torch.cuda.synchronize()
    finally:
        Pix2PixModel.graph_dstep = graph_was
        for k in FAM:
            setattr(L, k, orig[k])
    total = t0.elapsed_time(t1) / steps
    rows, acc, by_label = [], 0.0, {}
    for k, (label, fl) in FAM.items():
        ms = sum(a.elapsed_time(b) for a, b in events[k]) / steps
        acc += ms
        t = by_label.setdefault(label, [0, 0.0, 0.0, False])
        t[0] += len(events[k])
        t[1] += ms
        t[2] += flops[k]
        t[3] = t[3] or fl is not None
    for label, (n, ms, fl_sum, has_fl) in by_label.items():
        rows.append({"kernel": label, "launches_per_step": n // steps, "ms_per_step": round(ms, 3),
                     "tflops": round(fl_sum / steps / (ms * 1e-3) / 1e12, 2) if (has_fl and ms > 0) else None})
    rows.sort(key=lambda r: -r["ms_per_step"])
    rows.append({"kernel": other_label or "other (library GEMMs of the unfused layers, ATen elementwise / pooling / norms, Adam)",
                 "launches_per_step": None, "ms_per_step": round(total - acc, 3), "tflops": None})
    return rows


def other_breakdown(step_fn, world):
    """What the `other` row of a per-family table is made of: one step under torch.profiler, every device
    kernel that is NOT one of this repo's bucketed by its name -- library GEMM (rocBLAS / Tensile `Cijk_*`; with the FLOPs of
    the aten::mm / addmm / bmm calls that launched them: 2*M*K*N from the recorded shapes), ATen (elementwise, reductions,
    copies issued by torch), MIOpen, the fused Adam, runtime copies / memsets, RCCL.  Kernel durations are the profiler's (the
    profiled step's HOST side is slower; its kernels are not).  Returns a list of rows, or None when profiling fails."""
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            step_fn()
            torch.cuda.synchronize()
        rows = prof.key_averages(group_by_input_shape=True)
    except Exception as e:   # noqa: BLE001 -- evidence, not the product: never fail the bench line over it
        return [{"error": repr(e)[:200]}]
    dev_us = lambda e: getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0) or 0
    is_kernel = lambda e: str(getattr(e, "device_type", "")).endswith("CUDA")
    buckets = {"library GEMM (rocBLAS / Tensile Cijk_*)": 0.0, "ATen (elementwise, reductions, layout copies)": 0.0,
               "MIOpen (the crop encoder's / discriminator's stock convolutions)": 0.0, "fused Adam": 0.0,
               "runtime copies / memsets": 0.0, "RCCL": 0.0, "this repo's kernels (for reference)": 0.0, "unclassified": 0.0}
    counts = {k: 0 for k in buckets}
    unknown = []
    for e in rows:
        if not is_kernel(e):
            continue
        n, us = e.key, dev_us(e)
        if "DistributedDataParallel" in n or n.startswith("Optimizer.") or n.startswith("ProfilerStep"):
            continue   # user annotations mirrored on the device track (their span covers kernels counted below), not kernels
        if n.startswith("Cijk_") or "Tensile" in n or "rocblas" in n.lower():
            k = "library GEMM (rocBLAS / Tensile Cijk_*)"
        elif "multi_tensor_apply" in n or "FusedAdam" in n or "fused_adam" in n.lower():
            k = "fused Adam"
        elif "miopen" in n.lower() or "MIOpen" in n or "naive_conv" in n or "igemm" in n.lower() or "gridwise" in n.lower():
            k = "MIOpen (the crop encoder's / discriminator's stock convolutions)"
        elif "at::native" in n or "at_cuda_detail" in n or "cub::" in n or "rocprim" in n or "elementwise" in n or "vectorized" in n:
            k = "ATen (elementwise, reductions, layout copies)"
        elif "rocclr" in n or "copyBuffer" in n or "fillBuffer" in n or n.lower().startswith("memcpy") or n.lower().startswith("memset"):
            k = "runtime copies / memsets"
        elif "nccl" in n.lower() or "rccl" in n.lower() or "AllReduce" in n or "Broadcast" in n or "msccl" in n.lower():
            k = "RCCL"
        elif "(anonymous namespace)::" in n or "gg2::" in n or "_kernel" in n and "void " in n and "at::" not in n:
            k = "this repo's kernels (for reference)"
        else:
            k = "unclassified"
            unknown.append((us / 1e3, n[:60]))
        buckets[k] += us / 1e3
        counts[k] += e.count
    gemm_flops = 0.0
    for e in rows:
        if is_kernel(e) or e.key not in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm"):
            continue
        sh = [s for s in (e.input_shapes or []) if len(s) >= 2]
        if len(sh) < 2:
            continue
        a, b = sh[-2], sh[-1]
        batch = a[0] if len(a) == 3 else 1
        gemm_flops += 2.0 * batch * a[-2] * a[-1] * b[-1] * e.count
    out = []
    for k, ms in buckets.items():
        if ms <= 0:
            continue
        row = {"bucket": k, "ms_per_step": round(ms, 3), "launches_per_step": counts[k]}
        if k.startswith("library GEMM") and gemm_flops:
            row["tflops"] = round(gemm_flops / (ms * 1e-3) / 1e12, 1)
            row["gflop_per_step"] = round(gemm_flops / 1e9, 1)
        if k == "unclassified" and unknown:
            row["largest"] = ["%.3f ms %s" % u for u in sorted(unknown, reverse=True)[:3]]
        out.append(row)
    return out


def collectives_of_a_step(step_fn, ddps):
    """What ONE iteration puts on the wire, observed in this run: the ranks RCCL sees, every explicit
    ``dist.all_reduce`` of the step (SPADE's synchronised BatchNorm sums, the Sinkhorn diameter: count and bytes) and DDP's gradient
    buckets per network (count and bytes from the reducer's own record: ``GradientBuckets.describe`` or DDP's logging data).  None without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    seen = {"calls": 0, "bytes": 0}
    orig = dist.all_reduce

    def counting(t, *a, **k):
        if not k.get("async_op"):    # the asynchronous ones are GradientBuckets' gradient buckets: reported per network below
            seen["calls"] += 1
            seen["bytes"] += t.numel() * t.element_size()
        return orig(t, *a, **k)
    dist.all_reduce = counting
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        dist.all_reduce = orig
    out = {"backend": dist.get_backend(), "rccl_ranks_seen": dist.get_world_size(),
           "explicit_all_reduces_per_step": seen["calls"], "explicit_all_reduce_bytes_per_step": seen["bytes"], "ddp": {}}
    for name, d in ddps:
        if d is None:
            continue
        try:
            if hasattr(d, "describe"):    # emlight_amd._dist.GradientBuckets (the default reducer)
                out["ddp"][name] = d.describe()
                continue
            info = d._get_ddp_logging_data()
            sizes = [int(x) for x in str(info.get("rebuilt_bucket_sizes") or info.get("bucket_sizes") or "").replace(",", " ").split()]
            out["ddp"][name] = {"buckets": len(sizes), "bytes": sum(sizes), "backend": info.get("backend_name"),
                                "world_size": info.get("world_size")}
        except Exception as e:   # noqa: BLE001 -- evidence only
            out["ddp"][name] = {"error": repr(e)[:120]}
    return out


def _free_gpu():
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


def _gemm_selection_status():
    """Whether the plain library GEMMs of this run used the recorded selection (emlight_amd/_gemm_selection.py)."""
    from emlight_amd import _gemm_selection
    return _gemm_selection.status()


def leg_projector(args, rank, world, dev, steps, warmup):
    """SURVEY 8d metric (ii): GenProjector training images/sec, one G step + one D step (GenProjector/train.py:33-37)
    at BASELINE configs[2] (B=32 per GPU, 128x256 panoramas).  SphereConv2D and SPADE's modulation run on the HIP
    kernels; the object prices the WHOLE step against the f32 MFMA peak (the convolutions' GEMMs dominate)."""
    os.environ.setdefault("MIOPEN_FIND_MODE", "2")  # fast find: the first step must not spend minutes tuning
    from emlight_amd.GenProjector.networks import default_options
    from emlight_amd.GenProjector.model_trainer import Trainer
    from emlight_amd.GenProjector.data import projector_batch
    B = args.projector_batch
    import warnings
    data = projector_batch(B, dev, ln=args.anchors, seed=1234 + rank)

    def run(no_vgg):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")   # "random VGG features": stated in the JSON instead
            tr = Trainer(default_options(no_vgg_loss=no_vgg, vgg_random=True), device=dev, world=world)
        dt = run_timed(lambda: tr.step(data), steps, warmup, world, dev)
        # EVERY rank runs the instrumented step (it contains DDP's all-reduces and SPADE's statistics all-reduce); rank 0 reports
        fams = time_projector_families(tr, data, 1) if not no_vgg else None
        other = other_breakdown(lambda: tr.step(data), world) if not no_vgg else None
        peak = round(torch.cuda.max_memory_allocated(dev) / 1e9, 1)
        del tr
        _free_gpu()
        return dt, peak, fams, other
    # headline: the reference's step, VGG perceptual term included (pix2pix_model.py:119-120) -- torchvision's weights are
    # not obtainable offline, so the feature stack holds seeded random weights: same work, not the reference's loss value
    dt, peak, fams, other = run(False)
    dt0, _, _, _ = run(True)
    value, value0 = B * world * steps / dt, B * world * steps / dt0
    gflop = PROJECTOR_STEP_GFLOP + VGG_STEP_GFLOP
    tf = gflop * value / world / 1e3
    out = {"metric": "training images/sec (projector step: SPADE generator + PatchGAN discriminator + VGG19 perceptual "
                     "term, G step + D step)",
           "value": round(value, 2), "unit": "images/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 3),
           "config": {"workload": "GenProjector train step (G+D), BASELINE configs[2]", "per_gpu_batch": B,
                      "global_batch": B * world, "pano_hw": [128, 256], "ngf": 64, "ndf": 64,
                      "vgg": "VGG19 to relu5_1 on fake and real, seeded random weights (pretrained ones are not obtainable "
                             "offline; injectable via opt.vgg_weights)",
                      "library_gemm_selection": _gemm_selection_status()},
           "without_vgg": {"value": round(value0, 2), "ms_per_step": round(dt0 / steps * 1e3, 3),
                           "frac_of_f32_mfma_peak": round(PROJECTOR_STEP_GFLOP * value0 / world / 1e3 / F32_MFMA_PEAK_TFLOPS, 4),
                           "note": "the round-1/2 configuration (no_vgg_loss=True): a step lighter than the reference's"},
           "peak_hbm_GB": peak,
           "step_frac_of_f32_mfma_peak": round(tf / F32_MFMA_PEAK_TFLOPS, 4),
           "step_note": "algorithmic conv FLOPs per image = 4*%.1f (G: fwd+bwd in the G step, fwd in the D step) + 5*%.2f (D) "
                        "+ 3*%.1f (VGG19: fake, real, data gradient) = %.1f GFLOP" % (G_FWD_GFLOP, D_PAIR_GFLOP, VGG_FWD_GFLOP, gflop)}
    if fams and rank == 0:
        dom = fams[0]
        out["roofline"] = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom["tflops"], "peak": F32_MFMA_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": round((dom["tflops"] or 0.0) / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                           "launches_per_step": dom["launches_per_step"], "ms_per_step": dom["ms_per_step"],
                           "note": "dominant kernel family of the projector step by HIP-event time on the launch stream (one "
                                   "instrumented step); achieved = the algorithmic conv FLOPs of exactly those launches "
                                   "(2*M*K*N from the launcher arguments) / their summed duration"}
        out["kernel_families"] = fams
        out["other_breakdown"] = other   # what the table's last row is made of (one step under torch.profiler)
    del data
    _free_gpu()
    return out


def leg_joint(args, rank, world, dev, steps, warmup):
    """SURVEY 8d metric (iii) / BASELINE configs[3]: the joint regression -> rasteriser -> projector step
    (emlight_amd/joint.py) at 32 images per GPU (256 over 8), 240x320 crops, 128 anchors, 128x256 panoramas."""
    os.environ.setdefault("MIOPEN_FIND_MODE", "2")
    from emlight_amd.joint import JointTrainer, joint_batch
    B, crop_hw = args.joint_batch, tuple(args.crop_hw)
    import warnings
    from emlight_amd.GenProjector.networks import default_options
    batch = joint_batch(B, dev, args.anchors, crop_hw, seed=1234 + rank)

    host_cpu = [None]

    def run(no_vgg):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tr = JointTrainer(default_options(no_vgg_loss=no_vgg, vgg_random=True), anchors=args.anchors, crop_hw=crop_hw, blur=args.blur,
                              device=dev, world=world)
        dt = run_timed(lambda: tr.step(batch), steps, warmup, world, dev)
        # host CPU seconds (all threads of this process) an iteration costs: the 2 500 launches of an iteration leave the host
        # ~110 us each on the clock but it must never fall behind -- on a loaded host it does, and the step time follows
        # (measured from a drained GPU: how long the host takes to enqueue one iteration, and how much GPU work is still queued
        #  when it is done -- the margin the host has per iteration before the GPU would wait for it)
        enq = tail = 0.0
        for _ in range(3):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            tr.step(batch)
            h1 = time.perf_counter()
            torch.cuda.synchronize()
            enq += h1 - h0
            tail += time.perf_counter() - h1
        host_cpu[0] = (enq / 3 * 1e3, tail / 3 * 1e3)
        # every rank runs the instrumented iteration (collectives inside); rank 0 reports
        fams = None if no_vgg else time_projector_families(
            tr, batch, 1, {**ENCODER_FAMILIES, **PROJECTOR_FAMILIES},
            "other (encoder BN / pooling / head passes, Sinkhorn, rasteriser, library GEMMs of the unfused layers, ATen glue, Adam)")
        other = None if no_vgg else other_breakdown(lambda: tr.step(batch), world)
        coll = None if no_vgg else collectives_of_a_step(
            lambda: tr.step(batch), [("encoder", tr.reg.buckets or tr.reg.ddp),
                                     ("generator", tr.proj.bucketsG or getattr(tr.proj, "_ddpG", None)),
                                     ("discriminator", tr.proj.bucketsD or getattr(tr.proj, "_ddpD", None))])
        peak = round(torch.cuda.max_memory_allocated(dev) / 1e9, 1)
        del tr
        _free_gpu()
        return dt, peak, fams, other, coll
    dt, peak, fams, other, coll = run(False)     # the generator losses include the VGG19 perceptual term, as in the reference (random weights)
    host_cpu_ms = host_cpu[0]
    dt0 = run(True)[0]
    value, value0 = B * world * steps / dt, B * world * steps / dt0
    enc_gflop = STEP_GFLOP_240x320 if crop_hw == (240, 320) else 0.0
    enc_exec = EXECUTED_STEP_GFLOP_240x320 if crop_hw == (240, 320) else 0.0
    gflop = enc_gflop + PROJECTOR_STEP_GFLOP + VGG_STEP_GFLOP
    gflop_exec = enc_exec + PROJECTOR_STEP_GFLOP + VGG_STEP_GFLOP
    tf = gflop * value / world / 1e3
    out = {"metric": "training images/sec (joint step: DenseNet -> SG rasteriser -> SPADE generator + PatchGAN, "
                     "encoder+G step and D step)",
           "value": round(value, 2), "unit": "images/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 3),
           "config": {"workload": "joint regression+projector train step, BASELINE configs[3] (256 over 8 GPUs)",
                      "per_gpu_batch": B, "global_batch": B * world, "crop_hw": list(crop_hw), "anchors": args.anchors,
                      "pano_hw": [128, 256], "ngf": 64, "ndf": 64, "vgg": "VGG19 perceptual term on, seeded random weights",
                      "library_gemm_selection": _gemm_selection_status()},
           "without_vgg": {"value": round(value0, 2), "ms_per_step": round(dt0 / steps * 1e3, 3),
                           "frac_of_f32_mfma_peak": round((enc_gflop + PROJECTOR_STEP_GFLOP) * value0 / world / 1e3
                                                          / F32_MFMA_PEAK_TFLOPS, 4)},
           "peak_hbm_GB": peak,
           "host_enqueue_ms_per_step": round(host_cpu_ms[0], 1) if host_cpu_ms else None,
           "gpu_still_queued_when_host_is_done_ms": round(host_cpu_ms[1], 1) if host_cpu_ms else None,
           "algorithmic_gflop_per_image": round(gflop, 1), "executed_gflop_per_image": round(gflop_exec, 1),
           "step_frac_of_f32_mfma_peak": round(tf / F32_MFMA_PEAK_TFLOPS, 4),
           "step_frac_executed": round(gflop_exec * value / world / 1e3 / F32_MFMA_PEAK_TFLOPS, 4),
           "roofline": {"kernel": "whole step", "bound": "mfma", "achieved": round(tf, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                        "note": "algorithmic conv FLOPs per image = %.1f (encoder train step) + %.1f (projector G+D step) + "
                                "%.1f (VGG19 on fake, real, data gradient) GFLOP; per-kernel figures: the `projector` and "
                                "regression legs" % (enc_gflop, PROJECTOR_STEP_GFLOP, VGG_STEP_GFLOP)}}
    if fams and rank == 0:
        out["kernel_families"] = fams
        out["other_breakdown"] = other
    if coll is not None:
        out["collectives"] = coll
    del batch
    _free_gpu()
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks (one process per GPU) ourselves and
    relay their output; the children see WORLD_SIZE and take the normal path."""
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get("EML_SHARE_GPUS") != "1":   # (tests: ranks may share a device over gloo)
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (args.gpus, n_dev))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch of the regression step (configs[1])")
    ap.add_argument("--projector_batch", type=int, default=32, help="per-GPU batch of the projector leg (configs[2])")
    ap.add_argument("--joint_batch", type=int, default=32, help="per-GPU batch of the joint leg (configs[3]: 256 / 8)")
    ap.add_argument("--anchors", type=int, default=128)
    ap.add_argument("--crop_hw", type=int, nargs=2, default=(240, 320))
    ap.add_argument("--blur", type=float, default=.05)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--legs", default="all",
                    help="comma list of the extra objects on the line: projector,joint,sinkhorn,rasteriser,families "
                         "(default all; 'none' = the headline regression step only)")
    ap.add_argument("--workload", default="regression", choices=["regression", "projector", "joint"],
                    help="regression = BASELINE's headline line with every leg as an object (default); projector / joint = "
                         "a line whose `value` is that leg alone (profiling runs)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    from emlight_amd.RegressionNetwork.engine import RegressionTrainer, init_distributed
    from emlight_amd.RegressionNetwork.data import synthetic_batch
    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree"
                         % (args.gpus, world))
    dev = "cuda:%d" % local
    crop_hw = tuple(args.crop_hw)
    legs = {"projector", "joint", "sinkhorn", "rasteriser", "families"} if args.legs == "all" else \
        set(x for x in args.legs.split(",") if x and x != "none")
    par = "dp%d" % world if world > 1 else "single"

    if args.workload in ("projector", "joint"):
        leg = (leg_projector if args.workload == "projector" else leg_joint)(args, rank, world, dev, args.steps, args.warmup)
        if rank == 0:
            leg.update({"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                        "data": "synthetic"})
            leg["config"]["parallelism"] = par
            print(json.dumps(leg), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    tr = RegressionTrainer(anchors=args.anchors, crop_hw=crop_hw, blur=args.blur, device=dev, world=world)
    batch = synthetic_batch(args.batch, args.anchors, crop_hw, seed=1234 + rank, device=dev)

    dt = run_timed(lambda: tr.step(batch), args.steps, args.warmup, world, dev)

    # live per-kernel timing: EVERY rank runs the instrumented steps (they contain DDP's all-reduce)
    fams = time_kernel_families(tr, batch, 2, args.batch, crop_hw) if "families" in legs else None
    peak_gb = round(torch.cuda.max_memory_allocated(dev) / 1e9, 1)
    reg_coll = collectives_of_a_step(lambda: tr.step(batch), [("encoder", tr.buckets or tr.ddp)])   # every rank steps; rank 0 reports
    del tr, batch
    _free_gpu()
    # the other legs of BASELINE's metric; every rank takes part (DDP collectives inside), rank 0 reports
    short = (min(args.steps, 5), min(args.warmup, 2))
    extra = {}
    # (the joint leg -- north_star's end-to-end step -- before the projector leg: the legs run back to back on one chip, and
    # the later one meets it warmer: the same joint step measured 279-280 ms on its own and 286 ms behind both other legs)
    if "joint" in legs:
        extra["joint"] = leg_joint(args, rank, world, dev, *short)
    if "projector" in legs:
        extra["projector"] = leg_projector(args, rank, world, dev, *short)

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
                       "sinkhorn_blur": args.blur, "parallelism": par,
                       "runtime_env": _runtime.status()},
            "peak_hbm_GB": peak_gb,
            "rccl_ranks_seen": dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1,
            "collectives": reg_coll,
            "step_tflops": round(STEP_GFLOP_240x320 * value / 1e3, 2) if crop_hw == (240, 320) else None,
            "step_frac_of_f32_mfma_peak": round(STEP_GFLOP_240x320 * value / world / 1e3 / F32_MFMA_PEAK_TFLOPS, 4)
            if crop_hw == (240, 320) else None,
            # the same with the FLOPs the matrix pipe really executes (transition convolutions at the pooled resolution)
            "algorithmic_gflop_per_image": STEP_GFLOP_240x320 if crop_hw == (240, 320) else None,
            "executed_gflop_per_image": round(EXECUTED_STEP_GFLOP_240x320, 1) if crop_hw == (240, 320) else None,
            "step_frac_executed": round(EXECUTED_STEP_GFLOP_240x320 * value / world / 1e3 / F32_MFMA_PEAK_TFLOPS, 4)
            if crop_hw == (240, 320) else None,
        }
        if fams:
            dom = fams[0]  # the dominant kernel family of the step, by measured GPU time (rank 0)
            # the roof that binds is the one the kernel sits closer to: f32 activations make the dense-layer
            # passes HBM-bound (O(L^2) re-reads of the concatenated block buffer), not MFMA-bound
            f_hbm = dom["algorithmic_GBps"] / HBM_PEAK_GBPS
            f_mfma = (dom["tflops"] or 0.0) / F32_MFMA_PEAK_TFLOPS
            hbm = f_hbm >= f_mfma
            traffic, source = pmc_traffic(dom["kernel"])
            out["roofline"] = {"kernel": dom["kernel"], "bound": "hbm" if hbm else "mfma",
                               "achieved": dom["algorithmic_GBps"] if hbm else dom["tflops"],
                               "peak": HBM_PEAK_GBPS if hbm else F32_MFMA_PEAK_TFLOPS,
                               "unit": "GB/s" if hbm else "TFLOP/s",
                               "frac": round(max(f_hbm, f_mfma), 4),
                               "traffic": traffic, "traffic_source": source,
                               "algorithmic_bytes_per_launch": round(dom["algorithmic_GB_per_step"] * 1e9 /
                                                                     max(dom["launches_per_step"], 1)),
                               # like for like with `traffic` (a mean over KERNEL DISPATCHES: a transition's launcher call
                               # is 3-4 dispatches of the family's kernel)
                               "algorithmic_bytes_per_dispatch": round(dom["algorithmic_GB_per_step"] * 1e9 /
                                                                       max(dom["dispatches_per_step"], 1)),
                               "traffic_over_algorithmic": (round(traffic / (dom["algorithmic_GB_per_step"] * 1e9 /
                                                                             max(dom["dispatches_per_step"], 1)), 3)
                                                            if traffic else None),
                               # what the kernel actually moves (PMC traffic per dispatch over the live dispatch duration): the
                               # 1x1 kernels stream at the part's ceiling on THIS count; the gap to `achieved` is whole-line
                               # fetches of row prefixes that end inside a 128-byte line
                               "traffic_GBps": (round(traffic / (dom["avg_dispatch_ms"] * 1e-3) / 1e9, 1)
                                                if traffic and dom.get("avg_dispatch_ms") else None),
                               "frac_on_traffic": (round(traffic / (dom["avg_dispatch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                                                   if traffic and dom.get("avg_dispatch_ms") and hbm else None),
                               "dispatches_per_step": dom["dispatches_per_step"],
                               # the strict count: only what the family's arithmetic needs (for the 1x1 weight gradient x and
                               # dz once, without the BN2-backward rebuild fused into it), and the fraction of the roof on it
                               "operand_minimal_bytes_per_dispatch": (round(dom["operand_minimal_GB_per_step"] * 1e9 /
                                                                            max(dom["dispatches_per_step"], 1))
                                                                      if dom.get("operand_minimal_GB_per_step") else None),
                               "frac_operand_minimal": (round(dom["operand_minimal_GBps"] / HBM_PEAK_GBPS, 4)
                                                        if dom.get("operand_minimal_GBps") else None),
                               "other_roof_frac": round(min(f_hbm, f_mfma), 4),
                               "launches_per_step": dom["launches_per_step"], "avg_launch_ms": dom["avg_launch_ms"],
                               "note": "achieved = algorithmic bytes (or conv FLOPs) of the family's launches / their "
                                       "summed HIP-event duration, live in this run; peaks: HBM3E 8 TB/s, f32 MFMA "
                                       "(v_mfma_f32_16x16x4_f32) 157.3 TFLOP/s; traffic = HBM bytes per launch from "
                                       "separate rocprofv3 --pmc passes of this command (PMC cannot be read inside a "
                                       "timed run): see traffic_source"}
            out["kernel_families"] = fams
        if "sinkhorn" in legs:
            out["sinkhorn"] = time_sinkhorn(args.batch, args.anchors, args.blur, dev)
            out["sinkhorn_n256"] = time_sinkhorn(16, 256, args.blur, dev)   # configs[4]'s per-GPU shape: B=16, N=256
        if "rasteriser" in legs:
            out["rasteriser"] = time_rasteriser(args.projector_batch, args.anchors, (128, 256), dev)
            out["rasteriser_n256"] = time_rasteriser(16, 256, (256, 512), dev)   # configs[4]'s per-GPU shape
        out.update(extra)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.anchors, crop_hw, args.blur)
        # the headline numbers of EVERY leg in one compact object: inside `roofline` (an object the driver's record keeps
        # whole) and once more as the LAST key of the line (what a truncated stdout tail still shows)
        legs_summary = {"regression": {"value": out["value"], "ms_per_step": out["ms_per_step"],
                                       "step_frac_of_f32_mfma_peak": out["step_frac_of_f32_mfma_peak"],
                                       "step_frac_executed": out["step_frac_executed"]}}
        for k in ("projector", "joint"):
            if k in out:
                legs_summary[k] = {"value": out[k]["value"], "ms_per_step": out[k]["ms_per_step"],
                                   "step_frac_of_f32_mfma_peak": out[k].get("step_frac_of_f32_mfma_peak",
                                                                            out[k].get("roofline", {}).get("frac")),
                                   "step_frac_executed": out[k].get("step_frac_executed"),
                                   "dominant_kernel_frac": out[k].get("roofline", {}).get("frac"),
                                   "without_vgg": out[k].get("without_vgg", {}).get("value")}
        for k in ("sinkhorn", "sinkhorn_n256"):
            if k in out and isinstance(out[k], dict):
                legs_summary[k] = {kk: out[k][kk] for kk in ("ms_per_loss_call", "ms_per_eps_step", "n_eps",
                                                             "frac_of_hbm_peak_8TBps") if kk in out[k]}
        if "roofline" in out:
            out["roofline"]["legs"] = legs_summary
        out["legs_summary"] = legs_summary
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
