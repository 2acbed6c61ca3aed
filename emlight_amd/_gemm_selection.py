"""Which library kernel runs the projector's plain GEMMs.

The wide low-resolution SphereConv layers (1024 -> 1024 @8x16 ... 128 -> 2048 @16x32) and the few dense layers
of the networks are im2col + ``torch.mm`` / ``bmm`` / ``addmm`` products: 59 ms of a joint step on hipBLASLt's default
(heuristic) selection, 130-135 TF/s.  PyTorch's TunableOp can time every rocBLAS / hipBLASLt solution per GEMM shape and
record the fastest; ``tuned_gemms_gfx950.csv`` next to this file is that record for the shapes of BASELINE's projector and
joint steps (B = 32 per GPU), made once on an MI355X by ``tools/tune_gemms.sh``.  ``ensure()`` switches TunableOp on in
LOOK-UP mode (no tuning at run time, nothing written): shapes in the record run the recorded solution, every other shape the
library default.  TunableOp is PROCESS-WIDE -- it steers every ``torch.mm`` of the process, not only this package's -- so it is
an entry point's choice (``_runtime.entry_point_defaults()``: bench.py, the train / test / joint mains, the test session), never
a side effect of importing the package or of loading the HIP library; a host application opts in by calling ``ensure()``.  The record carries the PyTorch / ROCm /
rocBLAS / hipBLASLt versions and the architecture it was made with; on any other stack TunableOp rejects it and this module
switches TunableOp off again.  Every recorded solution is a plain Tensile kernel without atomics (checked by
``tests/test_gpu_projector.py``: repeated products are bit-identical).

Not used when ``EML_TUNED_GEMMS=0``, or when the process drives TunableOp itself through ``PYTORCH_TUNABLEOP_*`` variables
(which is how the record is made)."""
import os

from ._knobs import knob_flag

CSV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gemms_gfx950.csv")
_state = {"done": False, "active": False, "why": "not requested (a host application opts in with ensure())", "entries": 0,
          "requested": False}


def request():
    """An entry point's opt-in: ``ensure()`` then runs when the HIP library is first loaded (``_lib.lib()``)."""
    _state["requested"] = True
    if not _state["done"]:
        _state["why"] = "requested by the entry point; takes effect when the HIP library is first loaded"


def ensure_if_requested():
    return ensure() if _state.get("requested") else False


def ensure():
    """Idempotent.  Returns True when recorded selections are in effect."""
    if _state["done"]:
        return _state["active"]
    _state["done"] = True
    import torch
    if not knob_flag("EML_TUNED_GEMMS", True):
        _state["why"] = "EML_TUNED_GEMMS=0"
    elif any(k.startswith("PYTORCH_TUNABLEOP_") for k in os.environ):
        _state["why"] = "TunableOp is driven by the process's own PYTORCH_TUNABLEOP_* variables"
    elif not torch.cuda.is_available():
        _state["why"] = "no GPU"
    elif not os.path.exists(CSV):
        _state["why"] = "no record at %s" % CSV
    else:
        tn = None
        try:
            import torch.cuda.tunable as tn   # inside the try: a build without the module keeps the library defaults
            getattr(tn, "write_file_on_exit", lambda v: None)(False)   # look-up only: the packaged record is never rewritten
            tn.enable(True)
            tn.tuning_enable(False)
            tn.record_untuned_enable(False)
            tn.set_filename(CSV, insert_device_ordinal=False)
            ok = bool(tn.read_file(CSV))
            n = len(tn.get_results()) if ok else 0
        except Exception as e:   # an API drift must not take the step down: the default selection is always correct
            ok, n = False, 0
            _state["why"] = "TunableOp refused: %s" % e
        if ok and n > 0:
            _state.update(active=True, entries=n, why="%d recorded GEMM shapes from %s" % (n, os.path.basename(CSV)))
        else:
            try:
                if tn is not None:
                    tn.enable(False)
            except Exception:   # noqa: BLE001 -- nothing to switch off then
                pass
            if not _state["why"].startswith("TunableOp refused"):
                _state["why"] = "the record was made on another software stack (validators differ): library defaults"
    return _state["active"]


def status():
    """What the bench line reports: {"active": bool, "why": str, "entries": int}."""
    return {"active": _state["active"], "why": _state["why"], "entries": _state["entries"]}
