"""ISA check for conv3x3_bwd_fused_tp_kernel: between the asm-issued staging loads (global_load_dwordx4 written by hand, no
compiler record) and the s_waitcnt vmcnt(0) of g_wait() no instruction may read or write their destination registers.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/bwd.s emlight_amd/csrc/dense_bwd.hip
    python tools/exp/check_asm_loads.py /tmp/bwd.s"""
import re, sys
txt = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r"\n(_ZN\S*conv3x3_bwd_fused_tp_kernelILb1\w*):[^\n]*\n(.*?)s_endpgm", txt, re.S):
    lines = m.group(2).split("\n")
    i = 0
    while i < len(lines):
        mm = re.match(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\[", lines[i])
        if not mm:
            i += 1
            continue
        # a run of staging loads: collect until the next vmcnt(0)
        regs, j = set(), i
        while j < len(lines) and "vmcnt(0)" not in lines[j]:
            m2 = re.match(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\[", lines[j])
            if m2:
                regs |= set(range(int(m2.group(1)), int(m2.group(2)) + 1))
            else:
                body = lines[j].split(";")[0]
                used = set()
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", body):
                    used |= set(range(int(a), int(b) + 1))
                used |= set(int(x) for x in re.findall(r"\bv(\d+)\b", body))
                hit = used & regs
                if hit:
                    bad += 1
                    print("line %d touches in-flight registers %s: %s" % (j, sorted(hit), lines[j].strip()))
            j += 1
        print("run of loads at line %d: %d registers in flight until the wait at line %d" % (i, len(regs), j))
        i = j + 1
print("OK" if not bad else "%d hazards" % bad)
sys.exit(1 if bad else 0)
