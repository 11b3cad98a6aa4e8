"""Instruction mix of one kernel in a gfx950 .s file (make -C epic_amd/csrc asm): counts per opcode over the whole function
and over its hottest loop (the largest backward-branch body).  usage: isa_mix.py <file.s> <mangled-name-substring>"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and re.match(r"^_Z\w+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    def mix(seg):
        c = collections.Counter()
        for l in seg:
            m = re.match(r"\s+([a-z_0-9]+)\s", l + " ")
            if m and not l.strip().startswith((";", ".")):
                c[m.group(1)] += 1
        return c
    def show(name, c):
        valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_readfirstlane"))
        print("%s: %d instructions, %d VALU, %d SALU, %d ds, %d buffer/global, %d s_load" % (
            name, sum(c.values()), valu, sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_load", "s_waitcnt", "s_nop", "s_buffer"))),
            sum(v for k, v in c.items() if k.startswith("ds_")), sum(v for k, v in c.items() if k.startswith(("buffer_", "global_"))),
            sum(v for k, v in c.items() if k.startswith("s_load"))))
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
            if k.startswith("v_") or k.startswith("ds_"):
                print("    %-28s %d" % (k, v))
    show("whole function", mix(body))
    if loops:
        n, a, b = max(loops)
        print("hottest loop: lines %d..%d of the function" % (a, b))
        show("loop body", mix(body[a:b + 1]))


if __name__ == "__main__":
    main()
