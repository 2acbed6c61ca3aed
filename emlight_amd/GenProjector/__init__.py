"""Host-side mirror of the reference's ``GenProjector/`` package.

SPADE generator + multiscale PatchGAN discriminator keep the reference's ``nn.Module`` forward
signatures and ``state_dict`` keys.  Every ``SphereConv2D`` (73 of the generator's 78 convolutions, all of
the discriminator's) and SPADE's modulation run on the hand-written gfx950 kernels of
``csrc/sphere_conv.hip`` / ``csrc/spade.hip`` (``spherenet.py``); the ground-truth Gaussian map that the
reference rasterises per sample inside ``LavalIndoorDataset.__getitem__`` (``GenProjector/data.py:86-102``)
is one batched ``eml_sg_rasterise_f32`` call (``data.gaussian_map``).  The small stock modules that remain
(ConvEncoder's five strided convs, InstanceNorm, losses) are PyTorch-ROCm library calls.
"""
