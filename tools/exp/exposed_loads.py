"""Exposed round trips in an assembly file: a global (or scratch) load whose own s_waitcnt follows within a few instructions --
the pattern a sunk / rematerialised load leaves.  Prints kernel, line and whether the site is inside a loop.
    python tools/exp/exposed_loads.py /tmp/x.s [window=12]"""
import re, sys
txt = open(sys.argv[1]).read()
win = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S):
    name, lines = m.group(1), m.group(2).split("\n")
    inloop = False
    hits = []
    for i, l in enumerate(lines):
        if "Loop Header" in l or "in Loop:" in l:
            inloop = True
        elif re.match(r"\.LBB\d+_\d+:\s*$", l):
            inloop = False
        b = l.split(";")[0]
        if re.search(r"\b(global_load_dword|scratch_load_dword)", b) and "lds" not in b:
            dst = re.search(r"v\[?(\d+)", b)
            for j in range(i + 1, min(i + 1 + win, len(lines))):
                bj = lines[j].split(";")[0]
                if re.search(r"\b(global_load|global_store|scratch_)", bj):
                    continue
                mm = re.search(r"vmcnt\((\d+)\)", bj)
                if mm and int(mm.group(1)) <= 1:
                    hits.append((i, inloop, " ".join(b.split())[:60], "spill" if "Folded" in l else ""))
                    break
                if "v_mfma" in bj:
                    break
    loop_hits = [h for h in hits if h[1]]
    if loop_hits:
        import subprocess
        print("==", subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:110])
        for h in loop_hits:
            print("   line %5d  %s %s" % (h[0], h[2], h[3]))
