"""GPU parity: HIP Sinkhorn (through the C ABI) vs the oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

import oracle
from tests.golden.make_golden import SINKHORN_CASES

pytestmark = pytest.mark.gpu

LOSS_ATOL = 1e-6      # SURVEY 8d parity gate: loss <= 1e-6 abs at loss values ~1e-4
GRAD_RTOL = 1e-4      # grad <= 1e-4 rel (of the largest component)


def _crit(n, blur, diameter=None):
    from emlight_amd.RegressionNetwork.geomloss import SamplesLoss
    return SamplesLoss("sinkhorn", p=2, blur=blur, diameter=diameter, anchors=n)


def test_anchor_cost_matrix_matches_oracle():
    for n in (96, 128, 256, 50):
        crit = _crit(n, .05)
        M, _ = crit.cost_matrix(torch.device("cuda"))
        np.testing.assert_allclose(M.cpu().numpy(), oracle.anchor_cost_matrix(n).numpy(), rtol=0, atol=3e-7)


@pytest.mark.parametrize("case", [c[0] for c in SINKHORN_CASES])
def test_golden_cases(golden_sinkhorn, case):
    c = golden_sinkhorn.case(case)
    B, n = c["x"].shape
    diam = None if c["fixed_diameter"] < 0 else float(c["fixed_diameter"])
    crit = _crit(n, float(c["blur"]), diam)
    x = torch.from_numpy(c["x"]).cuda().view(B, n, 1)
    y = torch.from_numpy(c["y"]).cuda().view(B, n, 1)
    r = crit.forward_raw(x, y)
    n_eps = int(r["n_eps"].item())
    assert n_eps == len(c["eps_s"])
    np.testing.assert_allclose(r["eps_s"][:n_eps].cpu().numpy(), c["eps_s"].astype(np.float32), rtol=2e-7)
    assert abs(float(r["diameter"].item()) - float(c["diameter"])) <= 1e-7 * max(1.0, float(c["diameter"]))
    scale = max(1.0, float(np.abs(c["loss"]).max()) / 1e-4)
    np.testing.assert_allclose(r["loss"].cpu().numpy(), c["loss"], rtol=0, atol=LOSS_ATOL * scale)
    duals = r["duals"].cpu().numpy()
    np.testing.assert_allclose(duals, c["duals"], rtol=0, atol=2e-6 * max(1.0, np.abs(c["duals"]).max()))
    gref = c["grad_x"]
    np.testing.assert_allclose(r["gx"].cpu().numpy(), gref, rtol=GRAD_RTOL, atol=GRAD_RTOL * np.abs(gref).max())


@pytest.mark.parametrize("B,n,blur", [(64, 128, .05), (7, 96, .025), (5, 33, .05), (3, 64, .05), (2, 8, .05), (3, 200, .05), (16, 256, .05),
                                      (4, 132, .05), (3, 384, .05), (2, 512, .025), (2, 202, .05), (2, 516, .05),
                                      (24, 256, .05), (40, 256, .05), (20, 384, .05), (3, 192, .05), (3, 320, .025), (2, 448, .05)])
def test_autograd_vs_oracle(B, n, blur):
    """Seeded inputs at BASELINE cfg2/cfg5 shapes + ragged N; loss, d/dx and d/dy through autograd.  N <= 128: register-
    resident costs (N = 128, 64, 8 divide the workgroup size: the strided chord-matrix staging; 96, 33: the general one);
    N % 64 == 0 in [192, 512] at small batch: the SPLIT kernel, rows of a sample over 8 workgroups with the duals exchanged
    through global memory every sweep ((16, 256) = cfg5's per-GPU shape, (3, 192/320/384), (2, 448/512)), over 4 at
    (24, 256); 132 / 200 and the larger batches (40, 256) (two lanes per row), (20, 384) (one lane per row): the LDS-tiled
    kernel; 202 / 516 (N % 4 != 0 or N > 512): the streaming kernel."""
    g = torch.Generator().manual_seed(1234)
    x_c = torch.softmax(torch.randn(B, n, generator=g), 1).view(B, n, 1)
    y_c = torch.softmax(3 * torch.randn(B, n, generator=g), 1).view(B, n, 1)
    w = torch.rand(B, generator=g) + 0.5
    xo, yo = x_c.clone().requires_grad_(True), y_c.clone().requires_grad_(True)
    M = oracle.anchor_cost_matrix(n)
    lo = oracle.samples_loss(xo, yo, M, blur=blur)
    (lo * w).sum().backward()
    # the reference detaches y inside the cost (utils.py:88) but y still gets gradient through
    # the yy / yx problems' first argument -- the HIP backward returns exactly that.
    crit = _crit(n, blur)
    xg, yg = x_c.cuda().requires_grad_(True), y_c.cuda().requires_grad_(True)
    lg = crit(xg, yg)
    (lg * w.cuda()).sum().backward()
    scale = max(1.0, float(lo.abs().max()) / 1e-4)
    np.testing.assert_allclose(lg.detach().cpu().numpy(), lo.detach().numpy(), rtol=0, atol=LOSS_ATOL * scale)
    for got, want in ((xg.grad, xo.grad), (yg.grad, yo.grad)):
        want = want.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=GRAD_RTOL, atol=GRAD_RTOL * np.abs(want).max())


@pytest.mark.parametrize("B,n", [(3, 96), (3, 256), (40, 256), (2, 384), (2, 202)])
def test_weighted_four_argument_form_and_zero_weights(B, n):
    """(alpha, x, beta, y) form incl. zero-mass anchors (log-weight -1e5, sinkhorn_divergence.py:47-50) on every loop kernel:
    register-resident (96), split (3 x 256, 2 x 384), LDS-tiled (40 x 256), streaming (202)."""
    g = torch.Generator().manual_seed(5)
    x = torch.softmax(torch.randn(B, n, generator=g), 1).view(B, n, 1)
    y = torch.softmax(torch.randn(B, n, generator=g), 1).view(B, n, 1)
    a = torch.rand(B, n, generator=g)
    a[:, ::7] = 0
    a = a / a.sum(1, keepdim=True)
    b = torch.rand(B, n, generator=g)
    b = b / b.sum(1, keepdim=True)
    M = oracle.anchor_cost_matrix(n)
    C = lambda p, q: oracle.spherical_cost(p, q, M)
    eps_s = oracle.epsilon_schedule(2, oracle.max_diameter(x, y), .05, .5)
    duals = oracle.sinkhorn_loop(oracle.log_weights(a), oracle.log_weights(b), C(x, x), C(y, y), C(x, y), C(y, x), eps_s)
    want = oracle.sinkhorn_cost(a, b, *duals).numpy()
    got = _crit(n, .05)(a.cuda(), x.cuda(), b.cuda(), y.cuda()).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=LOSS_ATOL * max(1.0, np.abs(want).max() / 1e-4))


def test_properties_at_full_size():
    """Size-independent properties at BASELINE cfg4 per-GPU batch: S(x,x)=0, symmetry, determinism."""
    B, n = 256, 128
    g = torch.Generator().manual_seed(9)
    x = torch.softmax(torch.randn(B, n, generator=g), 1).view(B, n, 1).cuda()
    y = torch.softmax(2 * torch.randn(B, n, generator=g), 1).view(B, n, 1).cuda()
    crit = _crit(n, .05, diameter=1.0)
    sxx = crit(x, x)
    assert float(sxx.abs().max()) <= 2e-7
    sxy, syx = crit(x, y), crit(y, x)
    np.testing.assert_allclose(sxy.cpu().numpy(), syx.cpu().numpy(), rtol=0, atol=1e-6)
    assert float(sxy.min()) > 0
    assert torch.equal(crit(x, y), sxy)  # bitwise run-to-run


def test_empty_batch_and_errors():
    crit = _crit(96, .05)
    out = crit(torch.empty(0, 96, 1, device="cuda"), torch.empty(0, 96, 1, device="cuda"))
    assert out.shape == (0,)
    with pytest.raises(ValueError):
        crit(torch.rand(2, 64, 1, device="cuda"), torch.rand(2, 64, 1, device="cuda"))


def test_standalone_schedule_launcher_matches_numpy():
    """eml_sinkhorn_schedule_f32 (the schedule as its own launch) vs sinkhorn_divergence.py:21-25 in numpy."""
    from emlight_amd import _lib
    L, p = _lib.lib(), _lib.ptr
    g = torch.Generator().manual_seed(3)
    for scale, blur, diam in ((1.0, .05, None), (3.0, .025, None), (1.0, .05, 1.0), (1e-3, .05, None)):
        x = (torch.randn(5, 96, generator=g) * scale).cuda()
        y = (torch.randn(5, 96, generator=g) * scale).cuda()
        eps = torch.zeros(64, device="cuda")
        n_eps = torch.zeros(1, dtype=torch.int32, device="cuda")
        d = torch.zeros(1, device="cuda")
        _lib.check(L.eml_sinkhorn_schedule_f32(p(x), p(y), x.numel(), blur, .5, 2, -1.0 if diam is None else diam, None, p(eps),
                                               p(n_eps), p(d), _lib.current_stream()), "schedule")
        want_d = diam if diam is not None else oracle.max_diameter(x.cpu().view(-1, 96, 1), y.cpu().view(-1, 96, 1))
        want = oracle.epsilon_schedule(2, want_d, blur, .5)
        assert int(n_eps.item()) == len(want)
        np.testing.assert_allclose(eps[:len(want)].cpu().numpy(), np.asarray(want, np.float32), rtol=2e-7)


def test_gmloss_samples_loss_matches_reference_golden():
    """gmloss.SamplesLoss.forward(x, y, geometry) (GMLight): anchors, chord matrix (HIP, one launch
    instead of the N^2 Python loop) and the divergence / gradient against vectors of the REAL reference package."""
    from tests.conftest import Golden
    from tests.golden.make_golden import gmloss_inputs
    from emlight_amd.RegressionNetwork.gmloss import SamplesLoss
    g = Golden("gmloss")
    for name, B in (("b3_blur05", 3), ("b2_blur025", 2)):
        x_np, y_np, depth = gmloss_inputs(B, 17)
        x = torch.from_numpy(x_np).view(B, 128, 1).cuda().requires_grad_(True)
        y = torch.from_numpy(y_np).view(B, 128, 1).cuda()
        crit = SamplesLoss("sinkhorn", p=2, blur=float(g[name + "/blur"]), batchsize=B)
        loss = crit(x, y, depth)
        np.testing.assert_allclose(crit.anchors.cpu().numpy(), g[name + "/anchors"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(crit.M[::8].cpu().numpy(), g[name + "/M_rows8"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(loss.detach().cpu().numpy(), g[name + "/loss"], rtol=0, atol=1e-6)
        loss.sum().backward()
        np.testing.assert_allclose(x.grad.cpu().numpy().reshape(B, 128), g[name + "/grad_x"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("B,n", [(4, 128), (3, 96), (3, 256), (2, 384), (2, 202)])
def test_global_range_is_folded_into_the_diameter_scan(B, n):
    """``range_lo_hi`` of the C ABI (the other data-parallel ranks' (min, max), all-reduced by the caller): the kernel's
    own scan of x U y is widened by it, so the eps-schedule and the loss are those of the GLOBAL batch
    (sinkhorn_divergence.py:9-18) -- register-resident (128, 96), LDS-tiled (256, 384) and streaming (202) kernels."""
    from emlight_amd.RegressionNetwork.geomloss.samples_loss import sinkhorn_raw
    g = torch.Generator().manual_seed(77 + n)
    x = torch.softmax(torch.randn(B, n, generator=g), 1)
    y = torch.softmax(3 * torch.randn(B, n, generator=g), 1)
    lo, hi = float(min(x.min(), y.min())), float(max(x.max(), y.max()))
    crit = _crit(n, .05)
    M, Mt = crit.cost_matrix(torch.device("cuda"))
    Mo = oracle.anchor_cost_matrix(n)
    for rng in ((lo - 0.05, hi + 0.4), (lo + 0.01, hi - 0.01), (lo - 0.2, hi - 0.01)):   # wider, narrower (no effect), one-sided
        d = np.float32(max(hi, rng[1])) - np.float32(min(lo, rng[0]))       # f32 subtraction like max_diameter
        r = sinkhorn_raw(x.cuda(), y.cuda(), None, None, M, Mt, 2, .05, .5, None,
                         range_lo_hi=torch.tensor(rng, dtype=torch.float32, device="cuda"))
        want, aux = oracle.samples_loss(x.view(B, n, 1), y.view(B, n, 1), Mo, blur=.05, diameter=float(d), return_aux=True)
        n_eps = int(r["n_eps"].item())
        assert n_eps == len(aux["eps_s"])
        np.testing.assert_allclose(r["eps_s"][:n_eps].cpu().numpy(), np.asarray(aux["eps_s"], np.float32), rtol=2e-7)
        assert abs(float(r["diameter"].item()) - float(d)) <= 1e-7
        np.testing.assert_allclose(r["loss"].cpu().numpy(), want.numpy(), rtol=0, atol=LOSS_ATOL)


def _masked_stream(n_keep):
    """A HIP stream that may only use the first ``n_keep`` CUs (hipExtStreamCreateWithCUMask)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(n_keep):
        mask[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
    if rc != 0:
        pytest.skip("hipExtStreamCreateWithCUMask unavailable (%d)" % rc)
    return torch.cuda.ExternalStream(h.value, device=torch.device("cuda", 0))


def _split_case(B=16, n=256, blur=.05):
    g = torch.Generator().manual_seed(77)
    x = torch.softmax(torch.randn(B, n, generator=g), 1).view(B, n, 1)
    y = torch.softmax(3 * torch.randn(B, n, generator=g), 1).view(B, n, 1)
    return x, y, _crit(n, blur)


def _status_word(r, B, n):
    return int(r["work"][24 * B * n:24 * B * n + 1].view(torch.int32).item())


def test_split_kernel_is_not_chosen_on_a_stream_with_too_few_cus():
    """VERDICT r3 weak #3 / ADVICE r3 (medium): the split kernel needs 2*B*S co-resident workgroups.  On a stream whose CU
    mask leaves 32 CUs the launcher must size that from the STREAM's CUs (hipExtStreamGetCUMask), i.e. run the tiled
    kernel: same result as the oracle, status word untouched, and -- the discriminating part -- no 50 ms give-up."""
    from emlight_amd.RegressionNetwork.geomloss.samples_loss import EML_SINKHORN_NO_SPLIT
    B, n = 16, 256
    x, y, crit = _split_case(B, n)
    want = oracle.samples_loss(x, y, oracle.anchor_cost_matrix(n), blur=.05)
    xc, yc = x.cuda(), y.cuda()
    crit.cost_matrix(xc.device)
    ref = crit.forward_raw(xc, yc, flags=EML_SINKHORN_NO_SPLIT)     # the tiled kernel, default stream
    torch.cuda.synchronize()
    s = _masked_stream(32)
    with torch.cuda.stream(s):
        crit.forward_raw(xc, yc, flags=0)                            # warm-up (module load, LDS attribute)
        s.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        r = crit.forward_raw(xc, yc, flags=0)
        t1.record()
        s.synchronize()
    assert _status_word(r, B, n) == 0
    assert t0.elapsed_time(t1) < 20.0, "the launcher took the split path on a 32-CU stream and waited for its give-up"
    scale = max(1.0, float(want.abs().max()) / 1e-4)
    np.testing.assert_allclose(r["loss"].cpu().numpy(), want.numpy(), rtol=0, atol=LOSS_ATOL * scale)
    # 32 CUs for 32 workgroups of the tiled kernel: same kernel as the reference run -> bitwise
    assert torch.equal(r["loss"], ref["loss"]) and torch.equal(r["gx"], ref["gx"])


def test_split_kernel_gives_up_and_the_tiled_kernel_recomputes():
    """The on-device half of the fail-safe.  What the exchange needs co-resident is the S = 8 slices of one (sample, role):
    consecutive workgroup ids, which the dispatcher places together -- forced onto streams masked down to 32 and to 2 CUs the
    256-workgroup launch still completed (measured, round 4): the failure cannot be provoked from outside on an idle device.
    EML_SINKHORN_TEST_STALL makes one slice of every group withhold its granules, which is exactly what its partners see when
    it is not resident: they must give up within the 50 ms bound and raise the status word, and the gated tiled launch behind
    the kernel must recompute the batch -- the caller gets the tiled kernel's numbers, never NaN, never a hang."""
    from emlight_amd.RegressionNetwork.geomloss.samples_loss import (EML_SINKHORN_FORCE_SPLIT, EML_SINKHORN_NO_SPLIT,
                                                                     EML_SINKHORN_TEST_STALL)
    B, n = 16, 256
    x, y, crit = _split_case(B, n)
    xc, yc = x.cuda(), y.cuda()
    ref = crit.forward_raw(xc, yc, flags=EML_SINKHORN_NO_SPLIT)
    torch.cuda.synchronize()
    # control: the split kernel completes on its own (status 0) and agrees with the tiled kernel -- also on a 2-CU stream
    ok = crit.forward_raw(xc, yc, flags=EML_SINKHORN_FORCE_SPLIT)
    torch.cuda.synchronize()
    assert _status_word(ok, B, n) == 0
    np.testing.assert_allclose(ok["loss"].cpu().numpy(), ref["loss"].cpu().numpy(), rtol=0, atol=LOSS_ATOL)
    s = _masked_stream(2)
    with torch.cuda.stream(s):
        ok2 = crit.forward_raw(xc, yc, flags=EML_SINKHORN_FORCE_SPLIT)
        s.synchronize()
    assert torch.isfinite(ok2["loss"]).all()
    np.testing.assert_allclose(ok2["loss"].cpu().numpy(), ref["loss"].cpu().numpy(), rtol=0, atol=LOSS_ATOL)
    # the give-up
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    r = crit.forward_raw(xc, yc, flags=EML_SINKHORN_FORCE_SPLIT | EML_SINKHORN_TEST_STALL)
    t1.record()
    torch.cuda.synchronize()
    assert _status_word(r, B, n) == 1, "a slice withheld its granules and nobody noticed"
    assert 40.0 < t0.elapsed_time(t1) < 500.0    # one 50 ms give-up generation, not 2 s per poll and not a hang
    assert torch.isfinite(r["loss"]).all() and torch.isfinite(r["gx"]).all() and torch.isfinite(r["duals"]).all()
    assert torch.equal(r["loss"], ref["loss"]) and torch.equal(r["gx"], ref["gx"])   # the tiled kernel's own numbers
    # and the next healthy call on the same buffers clears the word
    again = crit.forward_raw(xc, yc, flags=EML_SINKHORN_FORCE_SPLIT, out={k: v for k, v in r.items() if k != "duals"} | {"work": r["work"]})
    torch.cuda.synchronize()
    assert _status_word(again, B, n) == 0


def test_split_watch_disables_the_split_path_after_a_give_up():
    """Host half: the status word of a call reaches the host without a sync (pinned copy + event) and switches later
    calls to EML_SINKHORN_NO_SPLIT."""
    from emlight_amd.RegressionNetwork.geomloss import samples_loss as sl
    watch = sl._SplitWatch()
    B, n = 16, 256
    work = torch.zeros(24 * B * n + 4, device="cuda")
    watch.after_call(work, B, n)
    watch.poll(wait=True)
    assert not watch.disabled and watch.flags() == 0
    work[24 * B * n:24 * B * n + 1].view(torch.int32).fill_(1)
    watch.after_call(work, B, n)
    with pytest.warns(RuntimeWarning, match="split Sinkhorn"):
        watch.poll(wait=True)
    assert watch.disabled and watch.flags() == sl.EML_SINKHORN_NO_SPLIT and watch.fallbacks == 1
