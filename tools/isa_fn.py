#!/usr/bin/env python3
"""cut one kernel out of hipcc's -save-temps assembly and tally its instructions:
   python tools/isa_fn.py FILE.s 'index_runs_kernelILb1ELi17ELi1E' [out.s]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.endswith(":") is False and re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = start
while not s[end].startswith(".Lfunc_end"):
    end += 1
body = s[start:end]
c = Counter()
for l in body:
    l = l.strip()
    if not l or l[0] in ".;" or l.split()[0].endswith(":"):
        continue
    c[l.split()[0]] += 1
print(len(body), "lines,", sum(c.values()), "instructions;", "VALU", sum(v for k, v in c.items() if k.startswith("v_")), "SALU", sum(v for k, v in c.items() if k.startswith("s_")),
      "LDS", sum(v for k, v in c.items() if k.startswith("ds_")), "VMEM", sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_"))))
for k, v in c.most_common(40):
    print(f"  {k:28s}{v}")
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(body))
