#!/usr/bin/env python3
"""ISA check for kernels that issue global loads from inline assembly and wait for them by hand (attn_bwd_fused32p_kernel):
no instruction may touch a load's destination registers between the load and the next hand-written `s_waitcnt vmcnt`
that covers it.  Usage: check_async_loads.py attention.s <kernel-name-substring>.  Conservative: after a load of registers R,
every later instruction that names a register of R before ANY `s_waitcnt vmcnt(` is reported (branches are followed
linearly, i.e. every path is treated as fall-through -- more than enough for straight-line request / deposit code)."""
import re, sys

def regs_of(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()

def main():
    text = open(sys.argv[1]).read()
    pat = sys.argv[2]
    bad = 0
    for m in re.finditer(r'^(_Z\w*%s\w*):' % re.escape(pat), text, re.M):
        name = m.group(1)
        body = text[m.end():text.index('s_endpgm', m.end())].split('\n')
        pending = {}      # reg -> line of the load
        in_asm = False
        for ln, l in enumerate(body):
            t = l.strip()
            if t.startswith(';;#ASMSTART'): in_asm = True; continue
            if t.startswith(';;#ASMEND'): in_asm = False; continue
            if not t or t.startswith(';') or t.startswith('.'): continue
            if 's_waitcnt' in t and 'vmcnt' in t:
                # a hand-written wait (inside an asm block) or the compiler's: either way everything older than N has landed;
                # we only accept vmcnt(0) as clearing all, and vmcnt(N) as clearing all but the N youngest loads
                n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
                order = sorted(set(pending.values()))
                keep = set(order[len(order) - n:]) if n else set()
                pending = {r: k for r, k in pending.items() if k in keep}
                continue
            toks = re.findall(r'v\[\d+:\d+\]|v\d+', t)
            if in_asm and t.startswith('global_load'):
                for r in regs_of(toks[0]): pending[r] = ln
                continue
            used = set()
            for tok in toks: used |= regs_of(tok)
            hit = used & set(pending)
            if hit:
                bad += 1
                print(f"{name[:60]}: line {ln}: `{t}` touches v{sorted(hit)} loaded at line {[pending[r] for r in sorted(hit)][0]}")
        print(f"{name[:70]}: checked {len(body)} lines")
    print("violations:", bad)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
