"""The vector-memory timeline of a kernel's assembly: every global load / store / scratch access / vmcnt wait with the number
of MFMAs issued before it, per basic block (so that a load whose wait follows a few MFMAs later stands out).
    python tools/exp/vm_timeline.py /tmp/x.s <kernel-name-substring> [min MFMAs per block to print]"""
import re, sys
txt = open(sys.argv[1]).read()
sel = sys.argv[2]
minmf = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S):
    if sel not in m.group(1):
        continue
    print("==", m.group(1)[:100])
    blocks, cur = [], ["entry", []]
    for l in m.group(2).split("\n"):
        if re.match(r"\.LBB\d+_\d+:", l):
            blocks.append(cur)
            cur = [l.split(":")[0], []]
        else:
            cur[1].append(l)
    blocks.append(cur)
    for name, ls in blocks:
        mf = sum("v_mfma" in x.split(";")[0] for x in ls)
        if mf < minmf:
            continue
        print("-- block %s: %d MFMAs" % (name, mf))
        c = 0
        for l in ls:
            b = l.split(";")[0].strip()
            if "v_mfma" in b:
                c += 1
            elif re.match(r"(global_|buffer_|scratch_)", b) or ("s_waitcnt" in b and "vmcnt" in b) or "s_barrier" in b:
                print("   %4d  %s" % (c, " ".join(b.split())[:80] + ("  ; spill" if "Folded" in l else "")))
