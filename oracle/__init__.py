"""oracle/ -- CPU restatement of the EMLight hot path.  TEST INFRASTRUCTURE ONLY.

This package is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``emlight_amd/`` imports it, and the product path raises if the HIP
library is missing instead of falling back to anything here.

The reference (fnzhan/EMLight) is pure Python on PyTorch; every function here is a
clean-room restatement in f32 torch-CPU / numpy of one reference function, with
the reference ``file:line`` it follows in its docstring.  The restatement is
parameterised where the reference hard-codes (anchor count N, panorama size) and
never calls ``.cuda()``.

Parity pinning: ``tests/golden/make_golden.py`` imports the real reference modules
from ``/root/reference`` in the build container (CPU), runs them on seeded inputs
and commits the input/output vectors as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks every oracle function against those vectors
(<= 1e-6), so the oracle is pinned to the reference, and the HIP kernels are then
checked against the oracle on the GPU box (where ``/root/reference`` does not
exist).
"""
from .sinkhorn import (sphere_points, anchor_cost_matrix, geometric_points, cost_matrix_of, spherical_cost,
                       epsilon_schedule, max_diameter, log_weights, softmin,
                       sinkhorn_loop, sinkhorn_cost, samples_loss,
                       samples_loss_grad_analytic)
from .rasteriser import pano_grid, convert_to_panorama
from .representation import ExtractMesh
from .projector import (sampling_grid, sphere_conv, spade_modulate, spade_norm_modulate, stock_sphere_ops,
                        StockVGG19, seeded_vgg19_state_dict)
from .joint import (predicted_gaussian_map, joint_step, joint_generator_step, joint_discriminator_step,
                    stock_rasteriser)
from .densenet import (OracleDenseNet, deterministic_state_dict, regression_loss,
                       deterministic_projector_state_dict)

__all__ = [
    "sphere_points", "anchor_cost_matrix", "geometric_points", "cost_matrix_of", "spherical_cost", "epsilon_schedule",
    "max_diameter", "log_weights", "softmin", "sinkhorn_loop", "sinkhorn_cost",
    "samples_loss", "samples_loss_grad_analytic", "pano_grid",
    "convert_to_panorama", "ExtractMesh", "OracleDenseNet", "deterministic_state_dict",
    "regression_loss", "deterministic_projector_state_dict",
    "StockVGG19", "seeded_vgg19_state_dict",
    "predicted_gaussian_map", "joint_step", "joint_generator_step", "joint_discriminator_step", "stock_rasteriser",
    "sampling_grid", "sphere_conv", "spade_modulate", "spade_norm_modulate", "stock_sphere_ops",
]
