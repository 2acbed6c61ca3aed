"""Diagnostics for tests/test_gpu_projector.py::test_three_iterations_follow_the_stock_op_trajectory: how far the HIP path's
parameter UPDATES and buffers are from the all-stock CPU trainer's after 3 iterations."""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import oracle
from emlight_amd.GenProjector import data, networks
from emlight_amd.GenProjector.model_trainer import Trainer
torch.manual_seed(5)
opt = networks.default_options(ngf=8, ndf=8)
cpu = Trainer(opt, device="cpu")
cpu.model.netG.load_state_dict(oracle.deterministic_projector_state_dict(cpu.model.netG.state_dict(), seed=11))
cpu.model.netD.load_state_dict(oracle.deterministic_projector_state_dict(cpu.model.netD.state_dict(), seed=12))
hip = Trainer(opt, device="cuda")
hip.model.netG.load_state_dict(cpu.model.netG.state_dict())
hip.model.netD.load_state_dict(cpu.model.netD.state_dict())
w0 = {n: {k: v.detach().clone().double() for k, v in net.state_dict().items()} for n, net in (("G", cpu.model.netG), ("D", cpu.model.netD))}
for it in range(3):
    batch = data.projector_batch(2, "cuda", seed=30 + it)
    hip.step(batch)
    with oracle.stock_sphere_ops():
        cpu.step({k: v.cpu() for k, v in batch.items()})
    lh, lc = hip.get_latest_losses(), cpu.get_latest_losses()
    print(it, {k: "%.3e" % abs(float(lh[k].detach().mean()) / float(lc[k].detach().mean()) - 1) for k in lc})
worst = {}
for name, nh, nc in (("G", hip.model.netG, cpu.model.netG), ("D", hip.model.netD, cpu.model.netD)):
    sh, sc = nh.state_dict(), nc.state_dict()
    for k in sc:
        if not sc[k].dtype.is_floating_point:
            continue
        a, b, z = sh[k].detach().cpu().double(), sc[k].detach().double(), w0[name][k]
        kind = k.split(".")[-1]
        du = float(((a - z) - (b - z)).norm() / ((b - z).norm() + 1e-30))
        dv = float((a - b).norm() / (b.norm() + 1e-30))
        w = worst.setdefault((name, kind), [0, 0, ""])
        if du > w[0]:
            w[0], w[2] = du, k
        w[1] = max(w[1], dv)
for k, v in sorted(worst.items()):
    print("%s %-14s update rel diff max %.3e   value rel diff max %.3e   (%s)" % (k[0], k[1], v[0], v[1], v[2]))
