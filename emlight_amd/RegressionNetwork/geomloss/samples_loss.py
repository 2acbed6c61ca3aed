"""``SamplesLoss`` -- EMLight's spherical mover's (Sinkhorn) loss on the MI355X.

Same constructor and ``forward`` as the reference class
(``RegressionNetwork/geomloss/samples_loss.py:12-92``); the body is one call into
``libemlight_hip.so`` (``eml_sinkhorn_schedule_f32`` + ``eml_sinkhorn_fwd_f32``) instead of
four materialised (B,N,N) cost tensors and ~250 ATen launches.  Extra, optional arguments
lift what the reference hard-codes: ``anchors`` (reference: 96, ``geomloss/utils.py:66``)
and ``cost_matrix`` (a runtime (N,N) ground cost -- gives the GMLight variant,
``gmloss/utils.py:63-108``, for free).  ``batchsize`` is accepted and ignored: the chord
matrix is stored once, not ``batchsize`` times (``utils.py:80-81``).
"""
import torch
from torch.nn import Module

from ... import _lib
from ..util import sphere_points


EML_SINKHORN_NO_SPLIT, EML_SINKHORN_FORCE_SPLIT, EML_SINKHORN_TEST_STALL = 1, 2, 4   # include/emlight_hip.h


def split_eligible(N):
    """Shapes whose small batches may take the split kernel (csrc/sinkhorn.hip; the only path that has a status word)."""
    return 192 <= N <= 512 and N % 64 == 0


class _SplitWatch:
    """Host side of the split kernel's fail-safe.  The library already made the result right on the device (the tiled
    kernel recomputes a batch whose split kernel gave up: slices not co-resident -- a CU-masked or shared GPU, another
    stream's kernel); what is left for the host is to NOTICE, so that the next calls stop paying the 50 ms give-up: the
    status word of a call is copied to pinned memory without a sync (first three calls, then every 64th) and read when a
    later call finds the copy complete -- the pattern of the dense engine's dgamma ring.  Once raised, this process passes
    EML_SINKHORN_NO_SPLIT from then on."""

    def __init__(self):
        self.disabled = False
        self.fallbacks = 0
        self.calls = 0
        self.pending = None   # (pinned int32 tensor, event)

    def flags(self):
        self.poll()
        return EML_SINKHORN_NO_SPLIT if self.disabled else 0

    def poll(self, wait=False):
        if self.pending is None:
            return
        host, ev = self.pending
        if wait:
            ev.synchronize()
        elif not ev.query():
            return
        self.pending = None
        if int(host[0]) != 0:
            self.fallbacks += 1
            if not self.disabled:
                import warnings
                warnings.warn("emlight_amd: the split Sinkhorn kernel's workgroups were not co-resident (CU mask, shared "
                              "GPU or a concurrent kernel); the tiled kernel recomputed the batch. The split path is "
                              "disabled for the rest of this process.", RuntimeWarning, stacklevel=3)
            self.disabled = True

    def after_call(self, work, B, N):
        self.calls += 1
        if self.disabled or self.pending is not None or not (self.calls <= 3 or self.calls % 64 == 0):
            return
        host = torch.empty(1, dtype=torch.int32).pin_memory()
        with torch.cuda.device(work.device):   # copy and event on the device (and its current stream) that ran the call
            host.copy_(work[24 * B * N:24 * B * N + 1].view(torch.int32), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        self.pending = (host, ev)


_split_watches = {}


def split_watch(device):
    """The watch of one device: whether the slices of the split kernel are co-resident is a property of the device (and of
    what shares it), so a give-up on one GPU must not disable the path on the others of a process."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _split_watches:
        _split_watches[idx] = _SplitWatch()
    return _split_watches[idx]


def sinkhorn_outputs(B, N, dev, need_gx=True, need_gy=False):
    """Device buffers one ``eml_sinkhorn_fwd_ex_f32`` call writes (the caller owns every buffer, include/emlight_hip.h)."""
    f32 = dict(dtype=torch.float32, device=dev)
    n_work = max(int(_lib.lib().eml_sinkhorn_work_floats(B, N)), 8 * B * N)
    work = torch.empty(n_work, **f32)
    if split_eligible(N) and n_work > 24 * B * N:
        work[24 * B * N:].zero_()   # the status word: only calls that take the split path reset it
    return {"eps_s": torch.empty(64, **f32), "n_eps": torch.empty(1, dtype=torch.int32, device=dev),
            "diameter": torch.empty(1, **f32), "loss": torch.empty(B, **f32),
            "gx": torch.empty(B, N, **f32) if need_gx else None, "gy": torch.empty(B, N, **f32) if need_gy else None,
            # scratch size from the library (duals, expectation rows, the small-batch kernel's exchange buffer and status
            # word); never less than the (8,B,N) planes every kernel writes
            "work": work}


def global_range(x, y):
    """(2,) tensor (min, max) of x U y over the batch of EVERY data-parallel rank, without a host sync: the local extrema,
    then ONE 2-float all-reduce (MIN over (min, -max)).  ``max_diameter`` (sinkhorn_divergence.py:9-18) takes the range
    over the whole batch, so a rank that only scanned its own shard would derive a different eps-schedule than the
    single-process run with the global batch.  With one rank this is just the local range."""
    import torch.distributed as dist
    lo = torch.minimum(x.detach().amin(), y.detach().amin())
    hi = torch.maximum(x.detach().amax(), y.detach().amax())
    r = torch.stack([lo, -hi]).float()
    from ..._dist import dp_active
    if dp_active():
        dist.all_reduce(r, op=dist.ReduceOp.MIN)
    return torch.stack([r[0], -r[1]])


def sinkhorn_raw(x, y, alpha, beta, M, Mt, p, blur, scaling, diameter, need_gx=True, need_gy=False, out=None,
                 range_lo_hi=None, flags=None):
    """One call into the HIP library; returns every device-side output (no autograd).  ``out``: buffers from
    ``sinkhorn_outputs`` to write into (a timing loop passes them so that no allocation sits between launches).
    ``range_lo_hi``: device (2,) tensor from ``global_range`` -- the kernel folds it into its own scan.
    ``flags``: EML_SINKHORN_* of the C ABI; default: whatever the split kernel's watch says (``_SplitWatch``)."""
    L = _lib.lib()
    B, N = x.shape
    o = out if out is not None else sinkhorn_outputs(B, N, x.device, need_gx, need_gy)
    watched = flags is None and split_eligible(N)
    if flags is None:
        flags = split_watch(x.device).flags() if watched else 0
    _lib.check(L.eml_sinkhorn_fwd_ex_f32(
        _lib.ptr(x), _lib.ptr(y), _lib.ptr(M), _lib.ptr(Mt), _lib.ptr(alpha), _lib.ptr(beta),
        float(blur), float(scaling), int(p), float(diameter) if diameter is not None else -1.0,
        _lib.ptr(range_lo_hi), _lib.ptr(o["eps_s"]), _lib.ptr(o["n_eps"]), _lib.ptr(o["diameter"]), _lib.ptr(o["loss"]), _lib.ptr(o["gx"]),
        _lib.ptr(o["gy"]), _lib.ptr(o["work"]), B, N, int(flags), _lib.current_stream()), "eml_sinkhorn_fwd_ex_f32")
    if watched and o["work"].numel() > 24 * B * N:
        split_watch(x.device).after_call(o["work"], B, N)
    return {"loss": o["loss"], "gx": o["gx"], "gy": o["gy"], "eps_s": o["eps_s"], "n_eps": o["n_eps"],
            "diameter": o["diameter"], "duals": o["work"][:4 * B * N].view(4, B, N), "work": o["work"]}


class _SinkhornDivergence(torch.autograd.Function):
    """loss (B,) = S_eps(alpha@x, beta@y).  Backward = analytic gradient of the last
    extrapolation (``sinkhorn_divergence.py:101-107``), produced by the forward kernel."""

    @staticmethod
    def forward(ctx, x, y, alpha, beta, M, Mt, p, blur, scaling, diameter, range_lo_hi=None):
        r = sinkhorn_raw(x, y, alpha, beta, M, Mt, p, blur, scaling, diameter,
                         ctx.needs_input_grad[0], ctx.needs_input_grad[1], range_lo_hi=range_lo_hi)
        ctx.save_for_backward(r["gx"], r["gy"])
        return r["loss"]

    @staticmethod
    def backward(ctx, gloss):
        gx_u, gy_u = ctx.saved_tensors
        L = _lib.lib()
        gloss = gloss.contiguous()
        out = [None, None]
        for k, gu in enumerate((gx_u, gy_u)):
            if gu is None:
                continue
            B, N = gu.shape
            go = torch.empty_like(gu)
            _lib.check(L.eml_sinkhorn_bwd_f32(_lib.ptr(gloss), _lib.ptr(gu), _lib.ptr(go), B, N,
                                              _lib.current_stream()), "eml_sinkhorn_bwd_f32")
            out[k] = go
        return out[0], out[1], None, None, None, None, None, None, None, None, None


class SamplesLoss(Module):
    """Debiased Sinkhorn divergence between sampled measures on the N sphere anchors.

    ``SamplesLoss(loss="sinkhorn", p=2, blur=.05, reach=None, diameter=None, scaling=.5,
    batchsize=None)`` -- reference signature ``samples_loss.py:22``; ``forward(x, y)`` with
    ``x, y`` of shape ``(B, N, 1)`` returns ``(B,)`` (``samples_loss.py:35-46``).
    """

    def __init__(self, loss="sinkhorn", p=2, blur=.05, reach=None, diameter=None, scaling=.5,
                 batchsize=None, anchors=96, cost_matrix=None, sync_diameter=False):
        super().__init__()
        if loss != "sinkhorn":
            raise ValueError("only loss='sinkhorn' exists in EMLight's geomloss fork")
        if reach is not None:
            raise NotImplementedError("unbalanced OT (reach) is not on EMLight's path (rho=None)")
        self.loss, self.p, self.blur, self.reach = loss, p, blur, reach
        self.diameter, self.scaling = diameter, scaling
        # data-parallel training: derive the eps-schedule from the range of the GLOBAL batch (one 2-float all-reduce per
        # call, no host sync) -- what the single-process reference sees; ignored when ``diameter`` is given
        self.sync_diameter = bool(sync_diameter)
        self.N = int(anchors) if cost_matrix is None else int(cost_matrix.shape[-1])
        # anchors as the reference builds them: float64 Fibonacci sphere cast to f32 (utils.py:67-69)
        self.register_buffer("anchors", torch.from_numpy(sphere_points(self.N)).float(), persistent=False)
        if cost_matrix is not None:
            M = torch.as_tensor(cost_matrix, dtype=torch.float32).reshape(self.N, self.N).clone()
            self.register_buffer("M", M, persistent=False)
            self.register_buffer("Mt", M.t().contiguous(), persistent=False)
        else:
            self.M = None
            self.Mt = None

    def cost_matrix(self, device):
        """(N,N) chord matrix on ``device`` -- built once by ``eml_emd_anchor_cost_f32``."""
        if self.M is None or self.M.device != device:
            if self.M is not None:  # user-supplied matrix: just move it
                self.M, self.Mt = self.M.to(device), self.Mt.to(device)
            else:
                a = self.anchors.to(device).contiguous()
                M = torch.empty(self.N, self.N, dtype=torch.float32, device=device)
                _lib.check(_lib.lib().eml_emd_anchor_cost_f32(_lib.ptr(a), _lib.ptr(M), self.N,
                                                              _lib.current_stream()), "eml_emd_anchor_cost_f32")
                self.M, self.Mt = M, M  # symmetric: the transpose aliases
        return self.M, self.Mt

    @staticmethod
    def generate_weights(x):
        if x.dim() == 3:
            B, N, _ = x.shape
            return torch.ones(B, N).type_as(x) / N
        raise ValueError("Input samples 'x' and 'y' should be encoded as (B,N,D) (batch) tensors.")

    def process_args(self, *args):
        if len(args) == 6:
            _, a, x, _, b, y = args
            return a, x, b, y
        if len(args) == 4:
            return args
        if len(args) == 2:
            x, y = args
            return None, x, None, y  # uniform 1/N weights are generated inside the kernel
        raise ValueError("A SamplesLoss accepts two (x, y), four (a, x, b, y) or six (l_x, a, x, l_y, b, y) arguments.")

    def forward(self, *args):
        a, x, b, y = self.process_args(*args)
        if x.dim() != 3 or x.shape[-1] != 1 or y.shape != x.shape or x.shape[1] != self.N:
            raise ValueError("expected x, y of shape (B, %d, 1), got %s and %s"
                             % (self.N, tuple(x.shape), tuple(y.shape)))
        B = x.shape[0]
        x2 = _lib.require_gpu_tensor(x.reshape(B, self.N), "x")
        y2 = _lib.require_gpu_tensor(y.reshape(B, self.N), "y")
        if B == 0:
            return x2.new_zeros(0) + 0.0 * (x2.sum() + y2.sum())
        a2 = None if a is None else _lib.require_gpu_tensor(a.reshape(B, self.N), "alpha")
        b2 = None if b is None else _lib.require_gpu_tensor(b.reshape(B, self.N), "beta")
        M, Mt = self.cost_matrix(x2.device)
        rng = global_range(x2, y2) if (self.sync_diameter and self.diameter is None) else None
        return _SinkhornDivergence.apply(x2, y2, a2, b2, M, Mt, self.p, self.blur, self.scaling, self.diameter, rng)

    def forward_raw(self, x, y, need_gx=True, need_gy=True, out=None, flags=None):
        """Every device output of one call (loss, unit grads, schedule, duals) -- for parity tests and timing."""
        B = x.shape[0]
        x2 = _lib.require_gpu_tensor(x.reshape(B, self.N), "x")
        y2 = _lib.require_gpu_tensor(y.reshape(B, self.N), "y")
        M, Mt = self.cost_matrix(x2.device)
        rng = global_range(x2, y2) if (self.sync_diameter and self.diameter is None) else None
        return sinkhorn_raw(x2, y2, None, None, M, Mt, self.p, self.blur, self.scaling, self.diameter,
                            need_gx, need_gy, out, range_lo_hi=rng, flags=flags)
