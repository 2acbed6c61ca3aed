"""Which call refuses stream capture?  (round 5's graph of the discriminator step's no-grad generator
pass died with hipErrorStreamCaptureUnsupported from a call that was never located.)

The generator's forward runs under ``torch.cuda.graph`` with every C-ABI launcher of libemlight_hip.so and every ATen op
followed by ``hipStreamIsCapturing`` on the capturing stream: the first call after which the status is no longer ACTIVE (or
that raises) is printed with its arguments' shapes.  One line per capture mode.
    python tools/capture_probe.py [B] [ngf]"""
import ctypes
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
from emlight_amd import _lib  # noqa: E402
from emlight_amd.GenProjector.data import projector_batch  # noqa: E402
from emlight_amd.GenProjector.model_trainer import Trainer  # noqa: E402
from emlight_amd.GenProjector.networks import default_options  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NGF = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hip = ctypes.CDLL("libamdhip64.so")
real = _lib.lib()
state = {"stream": None, "first": None, "calls": 0, "last": None}


def status():
    st = ctypes.c_int(-1)
    rc = hip.hipStreamIsCapturing(ctypes.c_void_p(state["stream"]), ctypes.byref(st))
    return rc, st.value   # status: 0 none, 1 active, 2 invalidated


def note(kind, name, detail):
    if state["stream"] is None or state["first"] is not None:
        return
    state["calls"] += 1
    state["last"] = "%s %s %s" % (kind, name, detail)
    rc, st = status()
    if rc != 0 or st != 1:
        state["first"] = "%s %s %s -> hipStreamIsCapturing rc=%d status=%d (after %d calls)" % (kind, name, detail, rc, st,
                                                                                             state["calls"])


class Spy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not callable(fn) or not name.startswith("eml_"):
            return fn

        def call(*a):
            rc = fn(*a)
            note("C-ABI", name, "rc=%r ints=%s" % (rc, [v for v in a if isinstance(v, int)][:8]))
            return rc
        return call


class AtenWatch(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        note("ATen", str(func), [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:4])
        return out


def main():
    dev = "cuda:0"
    tr = Trainer(default_options(ngf=NGF, ndf=NGF, no_vgg_loss=True), device=dev)
    data = projector_batch(B, dev)
    for _ in range(2):
        tr.step(data)
    torch.cuda.synchronize()
    net = tr.model.netG
    inp, crop = data["input"].clone(), data["crop"].clone()
    with torch.no_grad():
        ref = net(inp, crop).clone()      # warm: geometry tables, LDS attributes, BLAS handles -- all outside the capture
    torch.cuda.synchronize()
    _lib.lib = lambda: Spy()
    for mode in ("global", "thread_local", "relaxed"):
        state.update(stream=None, first=None, calls=0, last=None)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            a = torch.ones(64, 64, device=dev)
            torch.mm(a, a)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        err = None
        try:
            with torch.no_grad(), torch.cuda.graph(graph, stream=side, capture_error_mode=mode):
                state["stream"] = torch.cuda.current_stream().cuda_stream
                with AtenWatch():
                    out = net(inp, crop)
                state["stream_done"] = True
                rc, st = status()
                state["stream"] = None
        except Exception as e:   # noqa: BLE001
            import traceback
            tb = [ln for ln in traceback.format_exc().splitlines() if ln.strip()]
            mine = [i for i, ln in enumerate(tb) if "emlight_amd" in ln or "capture_probe" in ln]
            err = "%s: %s\n    last watched call: %s\n    %s" % (type(e).__name__, str(e).splitlines()[0][:200], state["last"],
                                                                 "\n    ".join(tb[max(0, (mine[-1] if mine else len(tb)) - 8):][:14]))
            state["stream"] = None
        torch.cuda.synchronize()
        if err is None and state["first"] is None:
            graph.replay()
            torch.cuda.synchronize()
            # a replay advances BatchNorm's running statistics and the power iteration like an eager call: compare loosely
            d = float((out - ref).abs().max() / ref.abs().max())
            print("capture_probe mode=%s: CAPTURED, %d calls watched, replay vs eager rel diff %.3g" % (mode, state["calls"], d))
        else:
            print("capture_probe mode=%s: FAILED  first offender: %s | exception: %s" % (mode, state["first"], err))
        del graph
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
