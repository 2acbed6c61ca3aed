"""emlight_amd -- MI355X (gfx950) native hot path of fnzhan/EMLight.

Host side: Python on PyTorch-ROCm mirroring the reference's own modules
(``RegressionNetwork.DenseNet``, ``RegressionNetwork.geomloss.SamplesLoss``,
``RegressionNetwork.util.convert_to_panorama`` ...).  Device side: hand-written HIP in
``csrc/`` behind the C ABI of ``include/emlight_hip.h`` (``libemlight_hip.so``, loaded
with ctypes by ``_lib``).  There is no CPU fallback: ops raise if the library is missing.
"""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory (a HIP runtime switch, read when the runtime initialises:
# effective when this package is imported before the process first touches the GPU; a value set by the user wins).  A training
# step here is 800-2 500 launches, many of them 5-10 us kernels whose first wavefront otherwise starts by fetching its arguments
# over the host link: measured on the MI355X, same box, alternating runs -- regression step 530.1 / 530.4 -> 534.0 / 533.5 img/s,
# joint step 291.3 / 289.0 -> 288.7 / 287.8 ms, kernel time of one traced joint iteration 285.8 -> 282.5 ms
# (profiles/r05_ab_dev_kernarg.txt).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
