"""A/B knobs of the HIP engine (``EML_*`` environment variables), parsed once per distinct value and validated.

They exist for same-box timing experiments (tools/, profiles/); a production run leaves them unset.  A malformed value does
not break ``import emlight_amd`` -- it is reported and the default is used -- and every non-default setting is announced once
on stderr, so that a knob left over from an experiment cannot silently change the dispatch."""
import os
import sys

_seen = {}   # (name, raw value) -> parsed value: a test or an A/B driver that changes the variable in-process is honoured


def _note(name, value, default):
    if value != default:
        print("emlight_amd: A/B knob %s=%r (default %r)" % (name, value, default), file=sys.stderr)


def knob_int(name, default, lo=None, hi=None):
    raw = os.environ.get(name)
    if ("int", name, raw) not in _seen:
        val = default
        if raw is not None:
            try:
                val = int(raw)
                if (lo is not None and val < lo) or (hi is not None and val > hi):
                    raise ValueError("out of range")
            except ValueError:
                print("emlight_amd: ignoring %s=%r (want an integer%s); using %r"
                      % (name, raw, "" if lo is None else " >= %d" % lo, default), file=sys.stderr)
                val = default
        _note(name, val, default)
        _seen[("int", name, raw)] = val
    return _seen[("int", name, raw)]


def knob_flag(name, default):
    """'0' / '1' switches; anything else is reported and ignored."""
    raw = os.environ.get(name)
    if ("flag", name, raw) not in _seen:
        val = bool(default)
        if raw is not None:
            if raw in ("0", "1"):
                val = raw == "1"
            else:
                print("emlight_amd: ignoring %s=%r (want 0 or 1); using %d" % (name, raw, int(default)), file=sys.stderr)
        _note(name, val, bool(default))
        _seen[("flag", name, raw)] = val
    return _seen[("flag", name, raw)]


def knob_choice(name, default, choices):
    raw = os.environ.get(name)
    if ("choice", name, raw) not in _seen:
        val = default
        if raw is not None:
            if raw in choices:
                val = raw
            else:
                print("emlight_amd: ignoring %s=%r (want one of %s); using %r" % (name, raw, "/".join(choices), default),
                      file=sys.stderr)
        _note(name, val, default)
        _seen[("choice", name, raw)] = val
    return _seen[("choice", name, raw)]
