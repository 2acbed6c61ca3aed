"""``emlight_amd._dist.GradientBuckets`` on the CPU, without a process group (the collectives themselves: tests/test_ddp_gloo.py
with two gloo ranks, tests/test_gpu_ddp_two_ranks.py): packing, bucket views as ``.grad``, parameters without a gradient, and
``grad_slot`` -- a producer writing straight into its parameter's place in the bucket."""
import copy

import torch

from emlight_amd import _dist
from emlight_amd._dist import GradientBuckets, grad_slot


def _inside(t, flat):
    return flat.data_ptr() <= t.data_ptr() < flat.data_ptr() + flat.numel() * flat.element_size()


def test_buckets_pack_views_and_unused_parameters():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    spare = torch.nn.Linear(4, 4)                      # registered with the reducer, never used in the forward
    ref = copy.deepcopy(net)
    params = list(net.parameters()) + list(spare.parameters())
    red = GradientBuckets(params, world=2, cap_mb=0)   # cap 0: one bucket per parameter -- exercises the in-order launch
    assert len(red.buckets) == len(params) and red.describe()["bytes"] == sum(4 * p.numel() for p in params)
    assert [b["params"][0] is p for b, p in zip(red.buckets, reversed(params))] == [True] * len(params)   # reverse registration order
    x = torch.randn(5, 8)
    for it in range(2):                                # the second iteration reuses the buckets
        for p in params:
            p.grad = None
        net(x).square().sum().backward()
        assert red.next == 0 or red.next <= len(params)
        red.finish()
        ref.zero_grad(set_to_none=True)
        ref(x).square().sum().backward()
        for p, q in zip(net.parameters(), ref.parameters()):
            b = red.buckets[red.where[id(p)]]
            assert _inside(p.grad, b["flat"]) and p.grad.shape == p.shape
            torch.testing.assert_close(p.grad, q.grad, rtol=0, atol=0)
        for p in spare.parameters():                   # no gradient in this backward: zeros, in the bucket
            assert _inside(p.grad, red.buckets[red.where[id(p)]]["flat"]) and float(p.grad.abs().max()) == 0.0
        assert red.next == 0 and not red.launched and all(b["pending"] == len(b["params"]) for b in red.buckets)


class _Lin(torch.autograd.Function):
    """x @ w.T whose backward asks for the weight gradient's place in the bucket, like the HIP producers do."""
    got = []

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.wptr = w.data_ptr()
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gw = g.t() @ x
        slot = grad_slot(ptr=ctx.wptr)
        _Lin.got.append(slot is not None)
        if slot is not None:
            slot.copy_(gw)
            gw = slot
        return g @ w, gw


def test_grad_slot_is_adopted_without_packing_and_refused_the_second_time():
    torch.manual_seed(1)
    w = torch.nn.Parameter(torch.randn(6, 8))
    w2 = torch.nn.Parameter(torch.randn(6, 8))
    red = GradientBuckets([w, w2], world=2, cap_mb=64)
    flat = red.buckets[0]["flat"]
    x = torch.randn(3, 8)
    packed = []
    inner = torch._foreach_copy_
    try:
        torch._foreach_copy_ = lambda dst, src, *a, **k: (packed.append(len(src)), inner(dst, src, *a, **k))[1]
        _Lin.got.clear()
        (_Lin.apply(x, w).sum() + _Lin.apply(x, w2).square().sum()).backward()
        assert _Lin.got == [True, True]
        assert _inside(w.grad, flat) and _inside(w2.grad, flat)      # autograd adopted the returned views
        red.finish()
        assert packed == []                                           # nothing left to pack
        torch.testing.assert_close(w.grad, torch.ones(3, 6).t() @ x, rtol=0, atol=0)
        # a parameter used twice in one forward: the second backward call gets no place (autograd has to sum two tensors)
        w.grad = w2.grad = None
        _Lin.got.clear()
        (_Lin.apply(x, w).sum() + _Lin.apply(2 * x, w).sum() + _Lin.apply(x, w2).sum()).backward()
        assert sorted(_Lin.got) == [False, True, True]
        red.finish()
        assert _inside(w.grad, flat)
        torch.testing.assert_close(w.grad, 3 * (torch.ones(3, 6).t() @ x), rtol=1e-6, atol=1e-6)
    finally:
        torch._foreach_copy_ = inner
    assert grad_slot(ptr=12345) is None and grad_slot(torch.nn.Parameter(torch.zeros(2))) is None
    with torch.enable_grad():
        assert grad_slot(w) is None                                   # a graph-building backward gets no place


def test_places_of_a_deleted_reducer_are_not_handed_out():
    import gc
    w = torch.nn.Parameter(torch.randn(4, 4))
    red = GradientBuckets([w], world=2)
    with torch.no_grad():
        assert grad_slot(w) is not None
    ptr = w.data_ptr()
    for h in red.hooks:
        h.remove()
    del red
    gc.collect()
    with torch.no_grad():
        assert grad_slot(w) is None and grad_slot(ptr=ptr) is None
    GradientBuckets([torch.nn.Parameter(torch.zeros(1))], world=2)    # construction purges the dead entries
    assert ("ptr", ptr) not in _dist._SLOTS
