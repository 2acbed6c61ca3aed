"""Process-wide runtime settings that only an ENTRY POINT may choose (importing a library must not change the
HIP runtime of its host process).

``HIP_FORCE_DEV_KERNARG=1`` puts kernel arguments in device memory instead of host-coherent memory; the HIP runtime reads it
once, when it initialises.  A training step here is 800-2 500 launches, many of them 5-10 us kernels whose first wavefront
otherwise starts by fetching its arguments over the host link: measured on the MI355X, same box, alternating runs -- regression
step 530.1 / 530.4 -> 534.0 / 533.5 img/s, joint step 291.3 / 289.0 -> 288.7 / 287.8 ms (profiles/r05_ab_dev_kernarg.txt).
``bench.py`` and the ``train.py`` / ``test.py`` / ``joint.py`` mains call ``entry_point_defaults()`` first thing; a value the
user exported wins; ``EML_DEV_KERNARG=0`` leaves the variable alone.  ``status()`` says what happened -- including "set too
late" when the GPU had already been touched -- and the bench line carries it."""
import os

_state = {"HIP_FORCE_DEV_KERNARG": None, "how": "entry_point_defaults() not called: the runtime's own default"}


def entry_point_defaults():
    from ._knobs import knob_flag
    if "HIP_FORCE_DEV_KERNARG" in os.environ:
        _state.update(HIP_FORCE_DEV_KERNARG=os.environ["HIP_FORCE_DEV_KERNARG"], how="set by the user's environment")
    elif not knob_flag("EML_DEV_KERNARG", True):
        _state.update(how="EML_DEV_KERNARG=0: left to the runtime's default")
    else:
        late = False
        try:
            import torch
            late = torch.cuda.is_initialized()
        except Exception:   # noqa: BLE001 -- no torch yet: certainly not initialised
            pass
        os.environ["HIP_FORCE_DEV_KERNARG"] = "1"
        _state.update(HIP_FORCE_DEV_KERNARG="1",
                      how="set by the entry point AFTER the GPU was first touched: without effect in this process" if late
                      else "set by the entry point before the HIP runtime initialised")
    # the recorded library-GEMM selection (TunableOp in look-up mode): process-wide too, hence REQUESTED here; it is switched
    # on when the HIP library is first loaded -- after the rank has chosen its device (reading the record queries the device)
    from . import _gemm_selection
    _gemm_selection.request()
    return status()


def status():
    from . import _gemm_selection
    return dict(_state, library_gemm_selection=_gemm_selection.status())
