"""Full drains of the vector-memory queue - `s_waitcnt vmcnt(0)` - and scratch (spill) accesses INSIDE the innermost MFMA loops of the device
assembly.  hipcc cannot count loads issued from inline asm, so wherever IT needs a vector-memory result inside such a loop (a global load it
issued, a spill reload) it waits for the whole queue - every LDS-DMA in flight included - and the prefetch distance of the loop is gone
(DESIGN.md 5, round 4).
usage: scan_loop_waits.py file.s [...]                         list every finding
       scan_loop_waits.py --fail <kernel-name regex> file.s    exit code 1 if a matching kernel has one, 2 if NO kernel matches the regex
                                                               (used by __graft_entry__.build())"""
import re, sys


def scan(path, name_rx=None, min_mfma=12):
    lines = open(path).read().split("\n")
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    inner = [(a, b) for a, b in loops if not any((c, d) != (a, b) and a <= c and d <= b for c, d in loops)]
    starts = sorted((i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(lines) if re.match(r"^(_Z\w+):", l))
    found = []
    for a, b in inner:
        mf = sum("v_mfma" in x for x in lines[a:b])
        if mf < min_mfma:
            continue
        name = [n for i, n in starts if i <= a][-1] if starts and starts[0][0] <= a else "?"
        if name_rx and not re.search(name_rx, name):
            continue
        for i in range(a, b):
            if re.search(r"s_waitcnt.*vmcnt\(0\)", lines[i]) or "scratch_" in lines[i]:
                found.append((name, i + 1, mf, lines[i].strip()))
    return found


def matching_kernels(path, name_rx):
    return [m.group(1) for m in re.finditer(r"^(_Z\w+):", open(path).read(), re.M) if re.search(name_rx, m.group(1))]


if __name__ == "__main__":
    args = sys.argv[1:]
    rx = None
    if args and args[0] == "--fail":
        rx, args = args[1], args[2:]
    rc = 0
    if rx and not any(matching_kernels(p, rx) for p in args):      # exit code 2: the pattern matches nothing (stale after a rename / template change)
        print(f"no kernel matches /{rx}/ in {args}")
        sys.exit(2)
    for p in args:
        for name, line, mf, text in scan(p, rx):
            print(f"{p}:{line}: {name[:100]}: innermost loop with {mf} MFMAs: {text}")
            rc = 1 if rx else 0
    sys.exit(rc)
