"""Host-side mirror of the reference's ``GenProjector/`` package.

SPADE generator + multiscale PatchGAN discriminator keep the reference's ``nn.Module`` forward
signatures and ``state_dict`` keys; in this round they run on stock PyTorch-ROCm ops (MIOpen
convolutions, ``grid_sample``) -- the north_star asks hand-written HIP only for the Sinkhorn loss,
the SG rasteriser and the DenseNet block, all of which live in ``RegressionNetwork``.  The one HIP
kernel on this path is the ground-truth panorama: ``LavalIndoorDataset.__getitem__`` rasterises the
GT lobes per sample on the GPU (``GenProjector/data.py:86-102``); here it is one batched
``eml_sg_rasterise_f32`` call in the training step (``data.projector_batch``).
Fused SphereConv / SPADE kernels are the "next" rows.
"""
