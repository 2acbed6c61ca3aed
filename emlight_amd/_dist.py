"""When do the data-parallel collectives of the training path run?

Normally: a ``torch.distributed`` process group with more than one rank (one process per GPU under torchrun; backend "nccl" IS
RCCL on ROCm).  ``EML_DIST_SINGLE=1`` is the single-GPU dry run of that path: the entry points then
initialise the process group with ONE rank too, attach the gradient reducer (GradientBuckets below, or DDP) and issue every collective of a
multi-rank iteration (gradient buckets, SPADE's synchronised BatchNorm sums, the Sinkhorn diameter) through RCCL on the one
device -- same results as without it (a one-rank all-reduce is the identity), but RCCL's initialisation, its kernels on the
HIP stream and DDP's hooks have then run under this code before an 8-GPU node ever sees it."""
import os
import weakref


def single_rank_dry_run():
    return os.environ.get("EML_DIST_SINGLE") == "1"


def dp_wrap(world):
    """Should a trainer built for ``world`` ranks all-reduce its gradients (GradientBuckets / DistributedDataParallel)?"""
    if world > 1:
        return True
    import torch.distributed as dist
    return single_rank_dry_run() and dist.is_available() and dist.is_initialized()


def dp_active():
    """Should a collective of the data-parallel path (sync-BN sums, global diameter) be issued now?"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or single_rank_dry_run()


def own_buckets():
    """EML_DP_BUCKETS=0: A/B knob -- torch's DistributedDataParallel wrappers instead of ``GradientBuckets``."""
    from ._knobs import knob_flag
    return knob_flag("EML_DP_BUCKETS", True)


_SLOTS = {}   # id(parameter) and ("ptr", data_ptr) -> (reducer, bucket index, index in the bucket)


def grad_slot(param=None, ptr=None):
    """Where a kernel that PRODUCES the gradient of a parameter may write it: a fresh view of the parameter's place in its
    gradient bucket (returned by a backward, autograd adopts it as ``.grad`` without a copy and the bucket needs no packing for
    it), or None -- no reducer, graph-building backward, or the place was already handed out in this backward (a module used
    twice: autograd has to sum two tensors).  The producer must write every element."""
    import torch
    hit = _SLOTS.get(id(param)) if param is not None else None
    if hit is None and ptr is not None:
        hit = _SLOTS.get(("ptr", ptr))
    if hit is None or torch.is_grad_enabled():
        return None
    red, bi, pi = hit[0](), hit[1], hit[2]
    if red is None:          # its trainer is gone (the registry holds weak references; an address can be reused)
        return None
    b = red.buckets[bi]
    if param is None and b["params"][pi].data_ptr() != ptr:
        return None
    if b["flat"] is None or pi in b["taken"]:
        return None
    b["taken"].add(pi)
    q = b["params"][pi]
    o = b["offs"][pi]
    return b["flat"][o:o + q.numel()].view(q.shape)


class GradientBuckets:
    """The gradient all-reduce of one network, without DistributedDataParallel's per-parameter copies.

    DDP with ``gradient_as_bucket_view`` still launches one copy kernel per parameter and backward: autograd hands every leaf
    a fresh gradient tensor, which the reducer copies into its bucket view -- ~500 kernels of 5-7 us per joint iteration
    (encoder 364 parameters, generator ~400, discriminator ~30: +5.9 ms of a 272 ms iteration with one rank,
    ``tools/scale_check.sh rccl``), i.e. a 2 % tax on every multi-GPU run before a byte has crossed xGMI.  Here a bucket
    (parameters in reverse registration order -- roughly the order their gradients become final -- up to ``cap_mb``) is packed
    by ONE multi-tensor copy when its last gradient has been accumulated (``register_post_accumulate_grad_hook``) and averaged
    over the ranks asynchronously on RCCL's stream while backward continues (``ReduceOp.AVG`` inside the collective; gloo: a
    division pass, then SUM); ``.grad`` of its parameters then ARE views of the bucket (the optimizer reads the reduced values in
    place, nothing is copied back).  The kernels that produce the large gradients (spectral norm's backward, the encoder's
    backward) write them straight into their places (``grad_slot``): those need no packing at all.  Buckets are launched
    strictly in index order on every rank.  ``finish()`` -- before the optimizer step -- launches what is left (a parameter
    without a gradient in this backward counts as zero: every rank runs the same graph) and makes the current stream wait for
    the collectives.  Like DDP, construction broadcasts rank 0's parameters (and buffers, once).  One backward per optimizer
    step with ``zero_grad(set_to_none=True)`` in between, as the trainers do: there is no ``no_sync`` accumulation mode."""

    def __init__(self, params, world, cap_mb=64, name="", buffers=()):
        import torch
        import torch.distributed as dist
        for k in [k for k, v in _SLOTS.items() if v[0]() is None]:   # places of trainers that no longer exist
            del _SLOTS[k]
        self.name, self.world = name, int(world)
        self.params = [p for p in params if p.requires_grad]
        self.buckets, self.where = [], {}
        cap, cur, size = int(cap_mb) << 20, [], 0
        for p in reversed(self.params):
            n = p.numel() * p.element_size()
            if cur and (size + n > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self._add(cur)
                cur, size = [], 0
            cur.append(p)
            size += n
        if cur:
            self._add(cur)
        for b in self.buckets:
            self._materialise(b)
        self.next = 0          # the next bucket to launch (strictly in order, so that every rank issues the same sequence)
        self.launched = []
        self.hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            with torch.no_grad():
                for t in self.params + [b for b in buffers if b is not None]:   # (once; DDP re-broadcasts buffers every
                    dist.broadcast(t.data, src=0)                                #  forward: here nothing rank-local feeds them)

    def _add(self, ps):
        self.where.update({id(p): len(self.buckets) for p in ps})
        self.buckets.append({"params": list(ps), "pending": len(ps), "flat": None, "views": None, "work": None, "offs": None,
                             "taken": set()})
        for i, p in enumerate(ps):
            if p.is_contiguous():
                _SLOTS[id(p)] = _SLOTS[("ptr", p.data_ptr())] = (weakref.ref(self), len(self.buckets) - 1, i)

    def _materialise(self, b):
        import torch
        offs, n = [], 0
        for p in b["params"]:
            offs.append(n)
            n += (p.numel() + 31) // 32 * 32        # every view starts on a 128-byte line
        p0 = b["params"][0]
        b["offs"] = offs
        b["flat"] = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        b["views"] = [b["flat"][o:o + p.numel()].view(p.shape) for o, p in zip(offs, b["params"])]

    def _on_grad(self, p):
        b = self.buckets[self.where[id(p)]]
        b["pending"] -= 1
        while self.next < len(self.buckets) and self.buckets[self.next]["pending"] <= 0:
            self._launch(self.buckets[self.next])
            self.next += 1

    def _launch(self, b):
        import torch
        import torch.distributed as dist
        if b["flat"] is None:
            self._materialise(b)
        src, dst = [], []
        with torch.no_grad():
            for p, v in zip(b["params"], b["views"]):
                g = p.grad
                if g is None:
                    v.zero_()
                elif g.data_ptr() != v.data_ptr():
                    src.append(g)
                    dst.append(v)
            if src:
                torch._foreach_copy_(dst, src)
            for p, v in zip(b["params"], b["views"]):
                p.grad = v
            if dp_active():
                if dist.get_backend() == "nccl":     # RCCL averages inside the collective: no division pass over the bucket
                    b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.AVG, async_op=True)
                else:
                    if dist.get_world_size() > 1:
                        b["flat"].div_(dist.get_world_size())
                    b["work"] = dist.all_reduce(b["flat"], async_op=True)
        self.launched.append(b)

    def finish(self):
        """Call between ``backward()`` and ``optimizer.step()``."""
        if any(b["pending"] < len(b["params"]) for b in self.buckets) or self.launched:
            while self.next < len(self.buckets):    # parameters that received nothing in this backward: zeros
                self._launch(self.buckets[self.next])
                self.next += 1
        for b in self.launched:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
        self.launched, self.next = [], 0
        for b in self.buckets:
            b["pending"] = len(b["params"])
            b["taken"].clear()

    def describe(self):
        import torch.distributed as dist
        sizes = [sum(p.numel() * p.element_size() for p in b["params"]) for b in self.buckets]
        on = dist.is_available() and dist.is_initialized()
        return {"buckets": len(sizes), "bytes": sum(sizes), "backend": dist.get_backend() if on else None,
                "world_size": dist.get_world_size() if on else 1, "reducer": "GradientBuckets (one packed copy per bucket)"}
