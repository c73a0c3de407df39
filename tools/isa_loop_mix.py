#!/usr/bin/env python3
"""Static instruction mix of every loop of one kernel in a hipcc -S listing (which loop is issue-bound and on what).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -S --cuda-device-only X.hip -o /tmp/x.s
    python tools/isa_loop_mix.py /tmp/x.s attn_bwd4_kernel
"""
import collections
import re
import sys


def cat(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(sys.argv[2]), l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), 1 << 30) < i:
            loops.append((labels[m.group(1)], i))
    print(f"{sys.argv[2]}: {len(body)} lines, {len(loops)} loops")
    for a, b in sorted(loops):
        c, vops, lops = collections.Counter(), collections.Counter(), collections.Counter()
        for l in body[a:b + 1]:
            l = l.strip()
            if not l or l[0] in ";.":
                continue
            op = l.split()[0]
            c[cat(op)] += 1
            if cat(op) == "valu":
                vops[op] += 1
            if cat(op) == "lds":
                lops[op] += 1
        print(f"  lines {a}-{b}: {dict(c)}")
        if b - a > 100:
            print("     valu:", vops.most_common(12))
            print("     lds: ", lops.most_common(8))


if __name__ == "__main__":
    main()
