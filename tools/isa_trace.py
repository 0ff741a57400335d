"""Condensed instruction-class trace of one kernel in a hipcc -S file: python3 tools/isa_trace.py file.s <kernel-substring>
(classes: MFMA, DSR / DSW, DMA (LDS-DMA), GLD / GST, SCR_LD / SCR_ST (spills), BAR, waits, labels and branches; run-length encoded)."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and sub in l and l.rstrip().split(';')[0].strip().endswith(':'))
out = []
for l in lines[start + 1:]:
    l = l.strip()
    if l.startswith('.Lfunc_end'): break
    if not l or l.startswith(';'): continue
    op = l.split()[0]
    if op.startswith('scratch_'): c = 'SCR_' + ('ST' if 'store' in op else 'LD')
    elif op.startswith('v_mfma'): c = 'MFMA'
    elif op.startswith('s_barrier'): c = 'BAR'
    elif op.startswith('ds_read') or op.startswith('ds_load'): c = 'DSR'
    elif op.startswith('ds_write') or op.startswith('ds_store'): c = 'DSW'
    elif op.startswith('global_load_lds'): c = 'DMA'
    elif op.startswith('global_store'): c = 'GST'
    elif op.startswith('global_load'): c = 'GLD'
    elif op.startswith('s_waitcnt'): c = 'W:' + l.split(None, 1)[1].split(';')[0].replace(' ', '')
    elif op.startswith('.LBB'): c = '\n' + op
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): c = 'BR->' + l.split()[-1]
    elif op.startswith('s_endpgm'): c = 'END'
    else: continue
    out.append(c)
rl = []
for c in out:
    if rl and rl[-1][0] == c: rl[-1][1] += 1
    else: rl.append([c, 1])
print(' '.join(f"{c}x{n}" if n > 1 else c for c, n in rl))
