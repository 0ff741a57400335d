"""Static check of kernels whose global loads are issued by inline asm with hand-counted `s_waitcnt vmcnt` (gemm.cuh::NtLoopDeep): the
compiler does not know those registers are in flight, so a register copy, spill or reuse it places between a load and the wait that
covers it reads or clobbers garbage.  This scans the device assembly (hipcc -S) of every matching kernel linearly from its entry to its
last MFMA, keeps the in-order queue of outstanding vector-memory operations (gfx9: loads and stores share vmcnt), retires entries at
each `s_waitcnt vmcnt(N)`, and fails if any other instruction names a VGPR that is still the destination of an outstanding load (past the last MFMA the scan
continues until no inline-asm load is in flight any more).
Only loads issued from INLINE ASM (between the `;;#ASMSTART` / `;;#ASMEND` markers of hipcc -S) are tracked as hazards: the compiler
counts its own loads itself, and a linear scan of a loop it generated would report them falsely; they still occupy their slot in the
vmcnt queue, as stores do.

usage: python3 tools/check_vmem_hazards.py <file.s> <kernel-name-regex>      (exit status 1 on a violation)"""
import re
import sys

WIDTH = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4, "ubyte": 1, "sbyte": 1, "ushort": 1, "sshort": 1, "short_d16": 1,
         "short_d16_hi": 1, "ubyte_d16": 1, "ubyte_d16_hi": 1, "sbyte_d16": 1, "sbyte_d16_hi": 1}
VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
LOAD = re.compile(r"^(global|buffer|flat|scratch)_load_(\w+)\s+(.*)$")
STORE = re.compile(r"^(global|buffer|flat|scratch)_(store|atomic)")
WAIT = re.compile(r"^s_waitcnt\b(.*)$")


def regs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_kernel(name, lines):
    last = max((i for i, l in enumerate(lines) if l.startswith("v_mfma")), default=-1)
    queue, bad = [], []
    in_asm = False
    for i, l in enumerate(lines):
        # past the last MFMA the scan goes on only while an inline-asm load is still in flight (gemm_pp.hip: the epilogue operands are fetched in the last load
        # phase of a tile and waited for behind its last MFMA)
        if i > last and not any(queue):
            break
        if l == "#ASMSTART":
            in_asm = True
            continue
        if l == "#ASMEND":
            in_asm = False
            continue
        m = WAIT.match(l)
        if m:
            c = re.search(r"vmcnt\((\d+)\)", m.group(1))
            if c:
                n = int(c.group(1))
                while len(queue) > n:
                    queue.pop(0)
            continue
        m = LOAD.match(l)
        if m and "_lds_" not in l:
            kind = m.group(2)
            ops = m.group(3)
            dst = regs(ops.split(",")[0])
            addr = regs(",".join(ops.split(",")[1:]))
            inflight = set().union(*[q for q in queue if q]) if queue else set()
            if (dst | addr) & inflight:
                bad.append((i, l))
            assert kind in WIDTH or kind.startswith("lds"), (name, l)
            queue.append(dst if in_asm else set())
            continue
        if m:                                   # LDS-DMA: no register destination, one slot in the queue
            queue.append(set())
            continue
        if STORE.match(l):
            inflight = set().union(*[q for q in queue if q]) if queue else set()
            if regs(l) & inflight:
                bad.append((i, l))
            queue.append(set())
            continue
        inflight = set().union(*[q for q in queue if q]) if queue else set()
        if inflight and regs(l) & inflight:
            bad.append((i, l))
    return bad


def main(path, pattern):
    rx = re.compile(pattern)
    cur, body, kernels = None, [], {}
    for raw in open(path):
        l = raw.strip()
        if raw.startswith("_Z") and l.endswith(":") or (raw.startswith("_Z") and ":" in l and "; @" in l):
            cur = l.split(":")[0]
            body = kernels.setdefault(cur, [])
            continue
        if cur is not None and l in (";;#ASMSTART", ";;#ASMEND"):
            body.append(l[2:])
            continue
        if cur is None or not l or l.startswith(";") or l.startswith(".") and not l.startswith(".LBB"):
            if l.startswith(".Lfunc_end"):
                cur = None
            continue
        if l.startswith("s_endpgm"):
            cur = None
            continue
        body.append(l.split(";")[0].strip())
    n = 0
    rc = 0
    for k, lines in kernels.items():
        if not rx.search(k):
            continue
        n += 1
        bad = check_kernel(k, lines)
        if bad:
            rc = 1
            print(f"HAZARD in {k}:")
            for i, l in bad[:10]:
                print(f"   +{i}: {l}")
    print(f"check_vmem_hazards: {n} kernel(s) matching /{pattern}/ scanned, {'VIOLATIONS' if rc else 'no register named while its load is in flight'}")
    if n == 0:
        rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
