#!/usr/bin/env python3
"""Per-kernel register / LDS / spill summary of a hipcc --save-temps .s file."""
import re, sys
txt = open(sys.argv[1]).read()
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = get("name")
    print("%-90s vgpr %s agpr %s sgpr %s spill %s lds %s scratch %s" % (name[:90], get("vgpr_count"), blk.split()[0], get("sgpr_count"),
          get("vgpr_spill_count"), get("group_segment_fixed_size"), get("private_segment_fixed_size")))
