"""Lint of the inline-asm statements of the HIP sources: a statement must declare everything it writes.
  * scalar-ALU instructions that write SCC (s_and / s_or / s_xor / s_add / s_sub / s_cmp / s_lshl / s_lshr / s_ashr / s_bfe / s_min / s_max / s_abs /
    s_cselect does not, s_mov does not) need an "scc" clobber - better: not be there (with the clobber the compiler turned the scalar selects around
    the statement into vector code; DESIGN.md 5, round 4: v_cmpx instead of v_cmp + s_and_b64 exec);
  * vcc written (v_cmp / v_cmpx e32 forms, v_add_co ...) needs a "vcc" clobber;
  * m0 and exec, when written, must be restored from a saved copy inside the same statement;
  * a vector-memory STORE of more than 64 bits (global_store_dwordx3 / x4, buffer_store_dwordx3 / x4) must be followed by wait states (s_nop) inside
    the statement: the hardware reads its data registers late, the next write of those registers needs >= 1 wait state (2 with an SGPR offset), and the
    compiler - which inserts them for the stores it knows - cannot see into the template.  Round 5: without them the second phase of the fc1 epilogue
    overwrote ~8,000 gelu' values per launch that were still waiting to be read (tests/test_precision_gpu.py::test_linear_fwd caught it).
usage: check_inline_asm.py file.hip [...]   (exit code 1 on a finding)"""
import re, sys

SCC_WRITERS = re.compile(r"\b(s_and|s_or|s_xor|s_nand|s_nor|s_xnor|s_andn2|s_orn2|s_add|s_addc|s_sub|s_subb|s_cmp|s_cmpk|s_lshl|s_lshr|s_ashr|s_bfe|s_bfm|s_min|s_max|s_abs|"
                         r"s_not|s_wqm|s_quadmask|s_bitcmp|s_absdiff|s_and_saveexec|s_or_saveexec)\w*\b")
WIDE_STORE = re.compile(r"\b(global|buffer|flat|scratch)_store_dwordx[34]\b[^\n]*(?:\n\s*([^\n]*))?")
VCC_WRITERS = re.compile(r"\b(v_cmp\w*|v_cmpx\w*|v_add_co\w*|v_sub_co\w*|v_div_scale\w*)\s+vcc\b")


def statements(text):
    """(line number, asm template string, clobber text) of every asm statement"""
    for m in re.finditer(r"asm\s+volatile\s*\(", text):
        i, depth = m.end(), 1
        while depth and i < len(text):
            depth += text[i] == "("
            depth -= text[i] == ")"
            i += 1
        body = text[m.end():i - 1]
        strs = re.findall(r'"((?:[^"\\]|\\.)*)"', body)
        # the template = the string literals in front of the first ':' that is outside a literal
        parts = re.split(r':(?=(?:[^"]*"[^"]*")*[^"]*$)', body)
        template = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', parts[0]))
        clobbers = parts[3] if len(parts) > 3 else ""
        yield text.count("\n", 0, m.start()) + 1, template.replace("\\n", "\n").replace("\\t", " "), clobbers


def check(path):
    bad = []
    for line, tmpl, clob in statements(open(path).read()):
        if not tmpl.strip():
            continue
        if SCC_WRITERS.search(tmpl) and '"scc"' not in clob:
            bad.append((line, "writes SCC without an \"scc\" clobber: " + SCC_WRITERS.search(tmpl).group(0)))
        for m in WIDE_STORE.finditer(tmpl):
            if not (m.group(2) or "").strip().startswith("s_nop"):
                bad.append((line, "vector-memory store of more than 64 bits without an s_nop behind it (late read of the data registers)"))
        if VCC_WRITERS.search(tmpl) and '"vcc"' not in clob:
            bad.append((line, "writes vcc without a \"vcc\" clobber"))
        for reg, save in (("m0", r"s_mov_b32\s+%\d+,\s*m0"), ("exec", r"s_mov_b64\s+%\d+,\s*exec")):
            sets = len(re.findall(r"\bs_mov_b(?:32|64)\s+" + reg + r"\s*,", tmpl))          # moves INTO the register (the last one must be the restore)
            written = sets > 0 or (reg == "exec" and re.search(r"v_cmpx", tmpl))
            if written:
                need = 2 if (reg == "m0" or not re.search(r"v_cmpx", tmpl)) else 1          # set + restore (exec set by v_cmpx: the restore alone)
                last = [l for l in tmpl.split("\n") if re.search(r"\b" + reg + r"\b", l)][-1]
                if not re.search(save, tmpl) or sets < need or not re.search(r"s_mov_b(?:32|64)\s+" + reg + r"\s*,\s*%\d+", last):
                    bad.append((line, f"writes {reg} without saving and restoring it inside the statement"))
    return bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        for line, msg in check(p):
            print(f"{p}:{line}: {msg}")
            rc = 1
    sys.exit(rc)
