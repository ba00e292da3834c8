"""Instruction histogram of one kernel in a hipcc -S dump: tools/isa_stats.py file.s <substring of mangled name> [loop]"""
import collections
import re
import sys

text = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(_Z\S*" + re.escape(key) + r"\S*):.*?$(.*?)^\s*s_endpgm", text, re.S | re.M)
if not m:
    sys.exit("kernel not found")
body = m.group(2)
lines = [l.strip() for l in body.split("\n")]
lines = [l for l in lines if l and not l.startswith((".", ";", "//"))]
print("kernel", m.group(1)[:100])


def hist(ls, title):
    ops = collections.Counter(l.split()[0] for l in ls if not l.endswith(":"))
    tot = sum(ops.values())
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    salu = sum(v for k, v in ops.items() if k.startswith("s_"))
    print(f"-- {title}: total {tot}  VALU {valu}  SALU {salu}  DS {sum(v for k, v in ops.items() if k.startswith('ds_'))}  "
          f"VMEM {sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))}")
    print("   " + "  ".join(f"{k}:{v}" for k, v in ops.most_common(28)))


hist(lines, "whole kernel")
# loops: a backward branch to a label defines a loop body
labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
for i, l in enumerate(lines):
    mm = re.match(r"s_cbranch_\w+\s+(\S+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i and i - labels[mm.group(1)] > 100:
        hist(lines[labels[mm.group(1)]:i + 1], f"loop {mm.group(1)} ({i - labels[mm.group(1)]} lines)")
