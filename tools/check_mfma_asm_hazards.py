"""Static check for the hazard the compiler does NOT cover: a VALU instruction issued from INLINE ASM that reads a register an MFMA is still writing.
On gfx950 a VALU read of an MFMA result needs software wait states (passes + 3: 11 behind an 8-pass v_mfma_f32_32x32x16_*, 7 behind a 4-pass 16x16x32); the
hazard recogniser inserts the `s_nop`s for the compiler's own instructions, but the body of an asm statement is opaque to it - the instruction reads
whatever the accumulator registers hold at that moment (round 6: the tile-0 row maximum of attn_fwd_pp_kernel read partial scores; results stayed inside
the tolerances - any softmax reference is a valid one - but differed from run to run).

For every MFMA of every kernel in the file this walks forward - through fall-through code and into the targets of branches - until WINDOW wait states have
passed (an instruction = 1, `s_nop N` = N + 1, an MFMA = its passes... counted as 4, the shortest) and reports an instruction between `;;#ASMSTART` and
`;;#ASMEND` that names one of the MFMA's destination registers as a SOURCE.  Empty asm statements (register launders) contain no instruction and are fine.

Two more hazards of the same kind (software wait states the recogniser places for its own instructions only) are checked on the way:
  * a transcendental result (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos) read by an asm VALU instruction in the very next issue slot (gfx940+: 1 wait state);
  * an SGPR written by a VALU instruction (v_readfirstlane / v_readlane / a compare into an SGPR pair) read by an asm vector-memory instruction - the base of an
    LDS-DMA - within 5 wait states.

usage: python3 tools/check_mfma_asm_hazards.py <file.s> [kernel-name-regex]      (exit status 1 on a finding)"""
import re
import sys

WINDOW = 19            # the longest requirement (16 passes + 3)
VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
KERNEL = re.compile(r"^(_Z\w+):")
LABEL = re.compile(r"^(\.LBB\w+):")
BRANCH = re.compile(r"^s_(c?branch\w*)\s+(\.LBB\w+)")
NOP = re.compile(r"^s_nop\s+(\d+)")
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")
TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_")
VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def linear_window(code, i, budget):
    """the instructions that issue within `budget` wait states behind code[i], fall-through only (labels skipped, stops at a branch)"""
    out, j = [], i + 1
    while j < len(code) and budget > 0:
        l, in_asm = code[j]
        if not LABEL.match(l):
            out.append((j, l, in_asm))
            m = NOP.match(l)
            budget -= int(m.group(1)) + 1 if m else 1
            if BRANCH.match(l) or l.startswith("s_endpgm"):
                break
        j += 1
    return out


def regs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def parse(path, pat):
    """-> {kernel: [(text, in_asm)]} with labels kept as their own entries"""
    kernels, cur, in_asm = {}, None, False
    for raw in open(path):
        l = raw.split(";")[0].strip() if not raw.lstrip().startswith(";;#ASM") else raw.strip().lstrip(";")
        m = KERNEL.match(raw)
        if m:
            cur = kernels.setdefault(m.group(1), []) if re.search(pat, m.group(1)) else None
            in_asm = False
            continue
        if cur is None or not l:
            continue
        if l.startswith("#ASMSTART"):
            in_asm = True
            continue
        if l.startswith("#ASMEND"):
            in_asm = False
            continue
        if l.startswith(".") and not LABEL.match(l):
            if l.startswith(".Lfunc_end") or l.startswith(".section"):
                cur = None
            continue
        cur.append((l, in_asm))
    return kernels


def walk(code, labels, i, budget, dst, seen, found):
    while i < len(code) and budget > 0:
        l, in_asm = code[i]
        if LABEL.match(l):
            i += 1
            continue
        key = (i, budget)
        if key in seen:
            return
        seen.add(key)
        ops = l.split(None, 1)
        srcs = regs(ops[1].split(",", 1)[1]) if len(ops) > 1 and "," in ops[1] else set()
        if l.startswith("v_") and not l.startswith("v_mfma") and in_asm and srcs & dst:
            found.append((i, l))
        if l.startswith("v_mfma") and regs(ops[1].split(",")[0]) & dst:
            return                                   # the chain's next MFMA: its own walk covers what follows
        if not in_asm and len(ops) > 1 and re.match(r"^(v_|ds_read|global_load|buffer_load|scratch_load)", l) and not l.startswith("v_cmp"):
            w = regs(ops[1].split(",")[0])
            if w & dst:                              # a compiler instruction overwrites the register (it places its own wait states): later reads see that value
                dst = dst - w
                if not dst:
                    return
        m = NOP.match(l)
        budget -= int(m.group(1)) + 1 if m else (4 if l.startswith("v_mfma") else 1)
        m = BRANCH.match(l)
        if m:
            if m.group(2) in labels:
                walk(code, labels, labels[m.group(2)], budget, dst, seen, found)
            if m.group(1) == "branch":
                return
        if l.startswith("s_endpgm"):
            return
        i += 1


def main():
    path, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "."
    kernels = parse(path, pat)
    bad = 0
    for name, code in kernels.items():
        labels = {LABEL.match(l).group(1): i for i, (l, _) in enumerate(code) if LABEL.match(l)}
        for i, (l, in_asm0) in enumerate(code):
            ops0 = l.split(None, 1)
            if TRANS.match(l) and len(ops0) > 1:
                d = regs(ops0[1].split(",")[0])
                for j, t, in_asm in linear_window(code, i, 1):
                    o = t.split(None, 1)
                    if in_asm and t.startswith("v_") and len(o) > 1 and "," in o[1] and regs(o[1].split(",", 1)[1]) & d:
                        bad += 1
                        print(f"{name}: `{t}` (inline asm) reads the result of `{l}` in the next issue slot (transcendental forwarding: 1 wait state)")
            if l.startswith("v_") and not in_asm0 and len(ops0) > 1:
                d = sregs(ops0[1].split(",")[0]) if not l.startswith("v_cmpx") else set()
                if d:
                    for j, t, in_asm in linear_window(code, i, 5):
                        if in_asm and VMEM.match(t) and sregs(t) & d:
                            bad += 1
                            print(f"{name}: `{t}` (inline asm) reads an SGPR `{l}` wrote {j - i} instruction(s) earlier (VALU -> SGPR -> vector memory: 5 wait states)")
                        o = t.split(None, 1)
                        if not in_asm and len(o) > 1 and re.match(r"^(s_|v_readfirstlane|v_readlane|v_cmp)", t) and sregs(o[1].split(",")[0]) & d:
                            d = d - sregs(o[1].split(",")[0])      # rewritten in between
            if not l.startswith("v_mfma"):
                continue
            dst = regs(l.split(None, 1)[1].split(",")[0])
            found = []
            walk(code, labels, i + 1, WINDOW, dst, set(), found)
            for j, t in found:
                bad += 1
                print(f"{name}: `{t}` (inline asm) reads a destination of `{l}` {j - i} instruction(s) behind it, inside its {WINDOW} wait states")
    print(f"check_mfma_asm_hazards: {len(kernels)} kernel(s) of {path} scanned, {bad} inline-asm read(s) of a result in flight")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
