"""For every kernel in an assembly file: how many MFMAs lie between a global load and the first s_waitcnt vmcnt that can be
waiting for it (the next vmcnt wait after it)?  Loads whose wait follows within a few MFMAs expose their round trip.
    python tools/exp/load_wait_distance.py /tmp/x.s [kernel-name-substring]"""
import re, sys
txt = open(sys.argv[1]).read()
sel = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S):
    name = m.group(1)
    if sel not in name:
        continue
    lines = m.group(2).split("\n")
    mf, events = 0, []
    for i, l in enumerate(lines):
        b = l.split(";")[0]
        if "v_mfma" in b:
            mf += 1
        elif re.search(r"\b(global_load|buffer_load)_(dword|ushort|ubyte|short)", b) and "lds" not in b:
            events.append(("L", i, mf))
        elif "s_waitcnt" in b and "vmcnt" in b:
            events.append(("W", i, mf, re.search(r"vmcnt\((\d+)\)", b).group(1)))
    if mf < 50:
        continue
    print("== %s: %d MFMAs" % (name[:90], mf))
    pend = []
    for e in events:
        if e[0] == "L":
            pend.append(e)
        else:
            n = int(e[3])
            done = pend[:len(pend) - n] if n < len(pend) else []
            pend = pend[len(pend) - n:] if n < len(pend) else pend
            if done:
                d = [e[2] - x[2] for x in done]
                flag = "  <-- SHORT" if min(d) < 12 else ""
                print("   wait vmcnt(%d) at line %d (MFMA %d): %d loads, issued %d..%d MFMAs earlier%s" % (n, e[1], e[2], len(done), min(d), max(d), flag))
