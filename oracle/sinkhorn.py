"""Oracle (test infrastructure): EMLight's "spherical mover's" Sinkhorn divergence.

CPU f32 restatement of ``RegressionNetwork/geomloss/`` (a stripped fork of geomloss
0.2.3, tensorized backend, balanced OT).  Differences from the reference are
interface-only: the anchor count ``N`` is a parameter (reference hard-codes 96,
``geomloss/utils.py:66``), the chord matrix ``M`` is stored once as ``(N, N)``
instead of ``B`` times, nothing calls ``.cuda()``, and the global
``set_grad_enabled`` side effect (``sinkhorn_divergence.py:74,99``) is replaced by
a local ``torch.no_grad()`` block.
"""
import numpy as np
import torch


def sphere_points(n=128):
    """Fibonacci-sphere anchors, float64 ``(n, 3)``.

    Follows ``RegressionNetwork/util.py:286-299`` (= ``geomloss/utils.py:48-61``):
    z = linspace(1 - 1/n, 1/n - 1, n), azimuth = k * pi * (3 - sqrt 5).
    """
    k = np.arange(n)
    azimuth = (np.pi * (3.0 - np.sqrt(5.0))) * k
    z = np.linspace(1.0 - 1.0 / n, 1.0 / n - 1.0, n)
    r = np.sqrt(1.0 - z * z)
    return np.stack([r * np.cos(azimuth), r * np.sin(azimuth), z], axis=1)


def anchor_cost_matrix(n):
    """Chord-length ground cost ``M_ij = ||a_i - a_j||_2`` on f32 anchors, ``(n, n)`` f32.

    Follows ``geomloss/utils.py:65-76`` (the N^2 ``torch.norm`` loop), vectorised.
    The anchors are cast to f32 *before* the subtraction, as the reference does.
    """
    a = torch.from_numpy(sphere_points(n)).float()
    d = a[:, None, :] - a[None, :, :]
    m = torch.sqrt((d * d).sum(-1))
    m[m < 0] = 0
    return m


def geometric_points(n, anchor_depth):
    """GMLight's depth-scaled anchors, float64 ``(n, 3)``: the Fibonacci azimuths and heights of ``sphere_points`` with
    the horizontal radius replaced by ``anchor_depth`` (``gmloss/utils.py:63-74``)."""
    k = np.arange(n)
    azimuth = (np.pi * (3.0 - np.sqrt(5.0))) * k
    z = np.linspace(1.0 - 1.0 / n, 1.0 / n - 1.0, n)
    r = np.asarray(anchor_depth, dtype=np.float64)
    return np.stack([r * np.cos(azimuth), r * np.sin(azimuth), z + 0.0 * r], axis=1)


def cost_matrix_of(anchors):
    """``M_ij = ||a_i - a_j||_2`` for arbitrary anchors cast to f32 first (``gmloss/utils.py:76-92``, the same N^2
    ``torch.norm`` loop as ``geomloss/utils.py:65-76``)."""
    a = torch.as_tensor(np.asarray(anchors)).float()
    d = a[:, None, :] - a[None, :, :]
    m = torch.sqrt((d * d).sum(-1))
    m[m < 0] = 0
    return m


def spherical_cost(x, y, M):
    """``C = 0.5 * (0.1 * (|x_i|^2 - 2 x_i.y_j + |y_j|^2) + M_ij)``, ``(B, N, N)``.

    Follows ``geomloss/utils.py:85-99`` and the ``/ 2`` lambda at
    ``geomloss/samples_loss.py:82``.  ``y`` is detached as in the reference
    (``utils.py:88``), so gradients reach only the first argument.
    """
    y = y.detach()
    d_xx = (x * x).sum(-1).unsqueeze(2)
    d_xy = torch.matmul(x, y.permute(0, 2, 1))
    d_yy = (y * y).sum(-1).unsqueeze(1)
    return ((d_xx - 2 * d_xy + d_yy) * 0.1 + M.detach().unsqueeze(0)) / 2


def max_diameter(x, y):
    """Range of the 1-D point cloud x U y.  Follows ``sinkhorn_divergence.py:9-18``."""
    D = x.shape[-1]
    xf, yf = x.reshape(-1, D), y.reshape(-1, D)
    mins = torch.minimum(xf.min(dim=0)[0], yf.min(dim=0)[0])
    maxs = torch.maximum(xf.max(dim=0)[0], yf.max(dim=0)[0])
    return (maxs - mins).norm().item()


def epsilon_schedule(p, diameter, blur, scaling):
    """eps list ``[d^p] + [exp(e) for e in arange(p ln d, p ln blur, p ln s)] + [blur^p]``.

    Follows ``sinkhorn_divergence.py:21-25`` (float64 python/numpy arithmetic).
    """
    start, stop, step = p * np.log(diameter), p * np.log(blur), p * np.log(scaling)
    mid = [float(np.exp(e)) for e in np.arange(start, stop, step)]
    return [float(diameter ** p)] + mid + [float(blur ** p)]


def log_weights(a):
    """``log(a)`` with -1e5 where ``a <= 0``.  Follows ``sinkhorn_divergence.py:47-50``."""
    out = a.log()
    out[a <= 0] = -100000
    return out


def softmin(eps, C, h):
    """``-eps * logsumexp_j(h_j - C_ij / eps)``.  Follows ``samples_loss.py:75-77``."""
    B = C.shape[0]
    return -eps * (h.view(B, 1, -1) - C / eps).logsumexp(2).view(B, -1)


def sinkhorn_loop(a_log, b_log, C_xx, C_yy, C_xy, C_yx, eps_s):
    """Symmetrised eps-scaling Sinkhorn iterations + last differentiable extrapolation.

    Follows ``sinkhorn_divergence.py:72-109`` with rho=None (dampening == 1).
    Returns ``(a_x, b_y, a_y, b_x)``; only the last four softmins carry grad.
    """
    with torch.no_grad():
        eps = eps_s[0]
        a_x = softmin(eps, C_xx, a_log)
        b_y = softmin(eps, C_yy, b_log)
        a_y = softmin(eps, C_yx, a_log)
        b_x = softmin(eps, C_xy, b_log)
        for eps in eps_s:
            at_x = softmin(eps, C_xx, a_log + a_x / eps)
            bt_y = softmin(eps, C_yy, b_log + b_y / eps)
            at_y = softmin(eps, C_yx, a_log + b_x / eps)
            bt_x = softmin(eps, C_xy, b_log + a_y / eps)
            a_x, b_y = .5 * (a_x + at_x), .5 * (b_y + bt_y)
            a_y, b_x = .5 * (a_y + at_y), .5 * (b_x + bt_x)
    h_xx = (a_log + a_x / eps).detach()
    h_yy = (b_log + b_y / eps).detach()
    h_yx = (a_log + b_x / eps).detach()
    h_xy = (b_log + a_y / eps).detach()
    return (softmin(eps, C_xx, h_xx), softmin(eps, C_yy, h_yy),
            softmin(eps, C_yx, h_yx), softmin(eps, C_xy, h_xy))


def sinkhorn_cost(a, b, a_x, b_y, a_y, b_x):
    """``<a, b_x - a_x> + <b, a_y - b_y>`` per sample.  ``sinkhorn_divergence.py:65-69``."""
    B = a.shape[0]
    return ((a.view(B, -1) * (b_x - a_x).view(B, -1)).sum(1)
            + (b.view(B, -1) * (a_y - b_y).view(B, -1)).sum(1))


def samples_loss(x, y, M, blur=.05, scaling=.5, p=2, diameter=None,
                 return_aux=False):
    """``SamplesLoss("sinkhorn", p, blur, scaling=..)(x, y)`` -> ``(B,)``.

    Follows ``geomloss/samples_loss.py:35-46,79-92`` (2-argument form: uniform
    weights 1/N, ``:64-72``).  ``x, y`` are ``(B, N, 1)``; ``M`` is ``(N, N)``.
    """
    B, N, _ = x.shape
    a = torch.ones(B, N, dtype=x.dtype) / N
    b = torch.ones(B, y.shape[1], dtype=y.dtype) / y.shape[1]
    C_xx, C_yy = spherical_cost(x, x.detach(), M), spherical_cost(y, y.detach(), M)
    C_xy, C_yx = spherical_cost(x, y.detach(), M), spherical_cost(y, x.detach(), M)
    if diameter is None:
        diameter = max_diameter(x, y)
    eps_s = epsilon_schedule(p, diameter, blur, scaling)
    duals = sinkhorn_loop(log_weights(a), log_weights(b), C_xx, C_yy, C_xy, C_yx, eps_s)
    loss = sinkhorn_cost(a, b, *duals)
    if return_aux:
        return loss, {"eps_s": eps_s, "diameter": diameter, "duals": duals}
    return loss


def samples_loss_grad_analytic(x, y, M, eps_s):
    """d loss_b / d x_i written out (no autograd), used to pin the HIP backward.

    Gradient reaches x only through the final softmins of the xx and xy problems
    (``sinkhorn_divergence.py:102-107``: duals detached) and only through the cost's
    first argument (``utils.py:88``):  with P = softmax_j(h_j - C_ij/eps),
    ``dL/dx_i = (1/N) [ sum_j P^xy_ij 0.1 (x_i - y_j) - sum_j P^xx_ij 0.1 (x_i - x_j) ]``.
    """
    with torch.no_grad():
        B, N, _ = x.shape
        a = torch.ones(B, N, dtype=x.dtype) / N
        a_log = log_weights(a)
        C_xx, C_yy = spherical_cost(x, x, M), spherical_cost(y, y, M)
        C_xy, C_yx = spherical_cost(x, y, M), spherical_cost(y, x, M)
        eps = eps_s[0]
        a_x, b_y = softmin(eps, C_xx, a_log), softmin(eps, C_yy, a_log)
        a_y, b_x = softmin(eps, C_yx, a_log), softmin(eps, C_xy, a_log)
        for eps in eps_s:
            at_x = softmin(eps, C_xx, a_log + a_x / eps)
            bt_y = softmin(eps, C_yy, a_log + b_y / eps)
            at_y = softmin(eps, C_yx, a_log + b_x / eps)
            bt_x = softmin(eps, C_xy, a_log + a_y / eps)
            a_x, b_y = .5 * (a_x + at_x), .5 * (b_y + bt_y)
            a_y, b_x = .5 * (a_y + at_y), .5 * (b_x + bt_x)
        P_xx = torch.softmax((a_log + a_x / eps).view(B, 1, N) - C_xx / eps, dim=2)
        P_xy = torch.softmax((a_log + a_y / eps).view(B, 1, N) - C_xy / eps, dim=2)
        xv, yv = x.view(B, N), y.view(B, N)
        g_xy = (P_xy * 0.1 * (xv[:, :, None] - yv[:, None, :])).sum(2)
        g_xx = (P_xx * 0.1 * (xv[:, :, None] - xv[:, None, :])).sum(2)
        return ((g_xy - g_xx) / N).view(B, N, 1)
