"""List compiler- or hand-placed `s_waitcnt vmcnt(0)` INSIDE loops of the device assembly of a .hip file (a full drain of the vector-memory
queue in a main loop defeats any prefetch distance: the compiler places one when a register it believes in flight - it cannot count asm
loads - is used inside the loop).  usage: scan_loop_waits.py file.s [...]"""
import re, sys
for path in sys.argv[1:]:
    name, labels, lines = None, {}, open(path).read().split("\n")
    # pass 1: label line numbers
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    loops = []          # (start, end) line ranges of backward branches
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i: loops.append((labels[m.group(1)], i))
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m: name = m.group(1)
        if re.search(r"s_waitcnt.*vmcnt\(0\)", l):
            inside = [(a, b) for a, b in loops if a <= i <= b]
            if inside:
                a, b = min(inside, key=lambda r: r[1] - r[0])
                mf = sum("v_mfma" in x for x in lines[a:b])
                print(f"{path}: {name[:90]} line {i + 1}: loop of {b - a} lines, {mf} MFMAs: {l.strip()}")
