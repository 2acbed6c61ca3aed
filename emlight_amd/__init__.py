"""emlight_amd -- MI355X (gfx950) native hot path of fnzhan/EMLight.

Host side: Python on PyTorch-ROCm mirroring the reference's own modules
(``RegressionNetwork.DenseNet``, ``RegressionNetwork.geomloss.SamplesLoss``,
``RegressionNetwork.util.convert_to_panorama`` ...).  Device side: hand-written HIP in
``csrc/`` behind the C ABI of ``include/emlight_hip.h`` (``libemlight_hip.so``, loaded
with ctypes by ``_lib``).  There is no CPU fallback: ops raise if the library is missing.
"""
# Importing this package changes nothing in the host process: the runtime settings an entry point may choose
# (kernel arguments in device memory, HIP_FORCE_DEV_KERNARG) live in ``_runtime.entry_point_defaults()``, called by bench.py and
# the train / test / joint mains; the recorded library-GEMM selection is opt-in (``_gemm_selection``).

__version__ = "0.1.0"
