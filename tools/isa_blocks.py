"""Instruction mix per basic block of one kernel in a hipcc -S listing:  python tools/isa_blocks.py file.s <kernel-name-substring> [min-instructions]"""
import collections
import re
import sys

src, pat = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])


def cat(op):
    for pre, name in (("v_mfma", "mfma"), ("v_readlane", "rdlane"), ("v_writelane", "wrlane"), ("v_exp", "trans"), ("v_rcp", "trans"),
                      ("v_rsq", "trans"), ("v_sqrt", "trans"), ("v_log", "trans"), ("v_", "valu"), ("s_waitcnt", "wait"), ("s_nop", "nop"),
                      ("s_cbranch", "br"), ("s_branch", "br"), ("s_barrier", "barrier"), ("s_", "salu"), ("ds_", "lds"),
                      ("buffer_", "vmem"), ("global_", "vmem")):
        if op.startswith(pre):
            return name
    return "other"


blocks, cur = [], ["entry", []]
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append(cur)
        cur = [t.split(":")[0], []]
        continue
    if t.startswith("."):
        continue
    cur[1].append(t)
blocks.append(cur)
tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter(cat(i.split()[0]) for i in ins)
    tot.update(c)
    if len(ins) >= minlen:
        print(f"{name:10s} {len(ins):4d}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("total", dict(tot))
