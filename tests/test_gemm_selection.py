"""The recorded library-GEMM selection (emlight_amd/_gemm_selection.py, tuned_gemms_gfx950.csv): the record is well formed and
the switch is inert without a GPU (CPU); on the MI355X it is in effect, deterministic and numerically a plain f32 GEMM."""
import os
import re

import numpy as np
import pytest
import torch


def _fresh():
    from emlight_amd import _gemm_selection as gs
    gs._state.update(done=False, active=False, why="not initialised", entries=0, requested=False)
    return gs


def test_record_is_well_formed():
    from emlight_amd import _gemm_selection as gs
    lines = [l.rstrip("\n") for l in open(gs.CSV)]
    validators = {l.split(",")[1]: l.split(",")[2] for l in lines if l.startswith("Validator,")}
    assert set(validators) == {"PT_VERSION", "HIP_VERSION", "HIPBLASLT_VERSION", "GCN_ARCH_NAME", "ROCBLAS_VERSION"}
    assert validators["GCN_ARCH_NAME"].startswith("gfx950")
    entries = [l.split(",") for l in lines if l and not l.startswith("Validator,")]
    assert len(entries) >= 60
    keys = set()
    for op, key, solution, ms in entries:
        assert re.fullmatch(r"Gemm(AndBias|Strided)?(Batched)?TunableOp_float_[NT][NT]", op), op
        assert re.fullmatch(r"[nt][nt]_\d+_\d+_\d+(_B_\d+)?_ld_\d+_\d+_\d+", key), key
        assert re.fullmatch(r"Default|Gemm_(Hipblaslt|Rocblas)_-?\d+", solution), solution
        assert float(ms) > 0
        assert (op, key) not in keys
        keys.add((op, key))
    # the products the record is for: the 1024 x 1024 x 3 x 3 layers of the generator at 8 x 16, 32 per GPU (forward, weight
    # and input gradient)
    assert any("_1024_4096_9216_" in k or "_4096_1024_9216_" in k for _, k in keys)
    assert any("_9216_" in k and "_4096_" in k for _, k in keys)


def test_switch_is_inert_without_a_gpu_and_obeys_its_knobs(monkeypatch):
    gs = _fresh()
    if torch.cuda.is_available():
        pytest.skip("CPU-side behaviour")
    assert gs.ensure() is False and gs.status() == {"active": False, "why": "no GPU", "entries": 0}
    gs = _fresh()
    monkeypatch.setenv("EML_TUNED_GEMMS", "0")
    assert gs.ensure() is False and gs.status()["why"] == "EML_TUNED_GEMMS=0"
    monkeypatch.delenv("EML_TUNED_GEMMS")
    gs = _fresh()
    monkeypatch.setenv("PYTORCH_TUNABLEOP_TUNING", "1")   # the process that MAKES the record must not be steered by the old one
    assert gs.ensure() is False and "PYTORCH_TUNABLEOP" in gs.status()["why"]
    assert gs.ensure() is False   # idempotent
    _fresh()


@pytest.mark.gpu
def test_recorded_selection_is_in_effect_deterministic_and_a_plain_f32_gemm():
    from emlight_amd import _gemm_selection as gs, _lib, _runtime
    _runtime.entry_point_defaults()   # what bench.py / the train mains / this test session do (tests/conftest.py)
    _lib.lib()
    gs.ensure_if_requested()          # (a library loaded before the request switches on here)
    st = gs.status()
    if not st["active"]:
        assert "validators" in st["why"] or "TUNED_GEMMS" in st["why"] or "PYTORCH_TUNABLEOP" in st["why"], st
        pytest.skip("recorded selection not in effect here: %s" % st["why"])
    import torch.cuda.tunable as tn
    assert tn.is_enabled() and not tn.tuning_is_enabled() and st["entries"] >= 60
    g = torch.Generator(device="cuda").manual_seed(1)
    M, C, O = 4096, 1024, 1024   # G_middle's convolutions at 8 x 16, B = 32
    a9 = torch.randn(M, 9 * C, device="cuda", generator=g)
    w2 = torch.randn(O, 9 * C, device="cuda", generator=g) * 0.02
    gy = torch.randn(M, O, device="cuda", generator=g)
    bias = torch.randn(O, device="cuda", generator=g)
    prods = {"forward": lambda: torch.addmm(bias, a9, w2.t()), "weight gradient": lambda: a9.t() @ gy, "input gradient": lambda: gy @ w2}
    refs = {"forward": lambda: bias.double() + a9.double() @ w2.double().t(), "weight gradient": lambda: a9.double().t() @ gy.double(),
            "input gradient": lambda: gy.double() @ w2.double()}
    for name, f in prods.items():
        y0, y1, y2 = f(), f(), f()
        assert torch.equal(y0, y1) and torch.equal(y0, y2), name            # no atomics in the recorded solutions
        ref = refs[name]()
        err = float((y0.double() - ref).norm() / ref.norm())
        assert err < 2e-6, (name, err)                                      # f32 accumulation over K = 9216 / 4096: ~3e-7
    n_before = len(tn.get_results())
    torch.randn(37, 53, device="cuda") @ torch.randn(53, 29, device="cuda")   # a shape not in the record: library default,
    assert len(tn.get_results()) == n_before                                  # nothing tuned, nothing recorded


class _FakeTunable:
    """Stands in for torch.cuda.tunable: records the switches, answers read_file / get_results as told."""

    def __init__(self, read_ok=True, n=82, boom=None):
        self.read_ok, self.n, self.boom, self.calls, self.on = read_ok, n, boom, [], False

    def enable(self, v=True):
        self.calls.append(("enable", v))
        self.on = bool(v)

    def tuning_enable(self, v=True):
        self.calls.append(("tuning_enable", v))

    def record_untuned_enable(self, v=True):
        self.calls.append(("record_untuned_enable", v))

    def write_file_on_exit(self, v):
        self.calls.append(("write_file_on_exit", v))

    def set_filename(self, name, insert_device_ordinal=False):
        self.calls.append(("set_filename", os.path.basename(name), insert_device_ordinal))

    def read_file(self, name=None):
        if self.boom:
            raise RuntimeError(self.boom)
        return self.read_ok

    def get_results(self):
        return [()] * self.n


@pytest.mark.parametrize("case", ["ok", "other stack", "empty", "raises"])
def test_switch_falls_back_to_library_defaults_when_the_record_is_refused(monkeypatch, case):
    """On a GPU whose software stack is not the one the record was made with TunableOp refuses the file (read_file False),
    an API drift may raise: in every such case TunableOp ends up OFF and the reason is reported -- never a half-enabled state
    that would start tuning inside a training step."""
    import sys
    import types
    gs = _fresh()
    for k in [k for k in os.environ if k.startswith("PYTORCH_TUNABLEOP_")]:
        monkeypatch.delenv(k)
    monkeypatch.delenv("EML_TUNED_GEMMS", raising=False)
    fake = {"ok": _FakeTunable(), "other stack": _FakeTunable(read_ok=False), "empty": _FakeTunable(n=0),
            "raises": _FakeTunable(boom="no such attribute")}[case]
    mod = types.ModuleType("torch.cuda.tunable")
    for name in ("enable", "tuning_enable", "record_untuned_enable", "set_filename", "read_file", "get_results", "write_file_on_exit"):
        setattr(mod, name, getattr(fake, name))
    monkeypatch.setitem(sys.modules, "torch.cuda.tunable", mod)
    monkeypatch.setattr(torch.cuda, "tunable", mod, raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    active = gs.ensure()
    st = gs.status()
    assert ("tuning_enable", False) in fake.calls and ("record_untuned_enable", False) in fake.calls
    assert ("set_filename", "tuned_gemms_gfx950.csv", False) in fake.calls   # one file for every rank: no device ordinal
    assert ("write_file_on_exit", False) in fake.calls                        # the packaged record is never rewritten
    if case == "ok":
        assert active and fake.on and st["entries"] == 82
    else:
        assert not active and not fake.on and st["entries"] == 0, st
        assert ("validators" in st["why"]) if case != "raises" else ("refused" in st["why"])
    _fresh()
