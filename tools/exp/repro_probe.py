"""Bitwise reproducibility of the projector's networks: the same input through netD / netG twice (forward maps and every
parameter gradient compared bit for bit), per network -- localises run-to-run differences of the joint step."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from emlight_amd.GenProjector import networks
from emlight_amd.joint import JointTrainer, joint_batch


def flat(o):
    if torch.is_tensor(o):
        return [o]
    out = []
    for q in o:
        out += flat(q)
    return out


def run(net, args, reps=4, x_grad=True):
    import copy
    res = []
    sd = copy.deepcopy(net.state_dict())
    for _ in range(reps):
        net.load_state_dict(sd)   # spectral norm's power iteration and the running statistics move with every forward
        torch.manual_seed(5)   # the generator draws its code
        for q in net.parameters():
            q.grad = None
        a = [t.detach().clone().requires_grad_(x_grad and t.is_floating_point()) for t in args]
        outs = flat(net(*a))
        loss = sum((o * o).mean() for o in outs)
        loss.backward()
        res.append(([o.detach().clone() for o in outs], [(n, q.grad.clone()) for n, q in net.named_parameters() if q.grad is not None],
                    [t.grad.clone() for t in a if t.grad is not None]))
    bad = 0
    for r in res[1:]:
        for i, (u, v) in enumerate(zip(res[0][0], r[0])):
            if not torch.equal(u, v):
                print("  forward map", i, tuple(u.shape), "differs: max", float((u - v).abs().max())); bad += 1
        for (n, u), (_, v) in zip(res[0][1], r[1]):
            if not torch.equal(u, v):
                print("  grad", n, tuple(u.shape), "differs: max", float((u - v).abs().max()), "of", float(u.abs().max())); bad += 1
        for i, (u, v) in enumerate(zip(res[0][2], r[2])):
            if not torch.equal(u, v):
                print("  input grad", i, tuple(u.shape), "differs: max", float((u - v).abs().max())); bad += 1
    return bad


if __name__ == "__main__":
    small = len(sys.argv) > 1 and sys.argv[1] == "small"
    torch.manual_seed(0)
    if small:
        tr = JointTrainer(networks.default_options(ngf=4, ndf=4), anchors=32, crop_hw=(64, 96), device="cuda:0")
        batch = joint_batch(2, "cuda:0", 32, (64, 96), seed=9)
    else:
        tr = JointTrainer(networks.default_options(), device="cuda:0")
        batch = joint_batch(2, "cuda:0", seed=9)
    tr.step(batch)
    m = tr.proj.model
    H, W = tr.pano_hw
    g = torch.Generator(device="cuda").manual_seed(3)
    inp = torch.randn(2, 3, H, W, device="cuda", generator=g)
    img = torch.randn(2, 3, H, W, device="cuda", generator=g)
    both = torch.cat([torch.cat([inp, img], 1), torch.cat([inp, img.flip(0)], 1)], 0)
    print("netD:", run(m.netD, [both]), "differences")
    crop = torch.randn(2, 3, *batch["crop"].shape[-2:], device="cuda", generator=g) if "crop" in batch else None
    try:
        print("netG:", run(m.netG, [inp, crop]), "differences")
    except Exception as e:
        print("netG probe failed:", repr(e))
    # whole steps from the same state: losses of N repeated steps from a restored state
    import copy
    st = [copy.deepcopy(n.state_dict()) for n in (tr.reg.model, m.netG, m.netD)]
    print("batch keys:", list(batch.keys()))
