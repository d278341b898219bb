#!/usr/bin/env python3
"""Linear-scan check of s_waitcnt coverage in a kernel's ISA: flags an instruction that reads (or overwrites) a VGPR which is the
destination of a DS read / VMEM load that the preceding s_waitcnt values do not prove complete.  Control flow is ignored (the
listing is walked top to bottom, counters carried across labels), so a flag is a place to look at, not a proof.
usage: isa_wait_check.py file.s kernel-name-substring"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = [x for x in re.finditer(r'^(_Z\w+):', s, re.M) if pat in x.group(1)]
for mm in m:
    name = mm.group(1)
    body = s[mm.end():s.index('.Lfunc_end', mm.end())].split('\n')
    lg = []   # outstanding LDS ops: (set of dest regs or empty)
    vm = []
    def regs(tok):
        r = re.match(r'v\[(\d+):(\d+)\]', tok)
        if r: return set(range(int(r.group(1)), int(r.group(2)) + 1))
        r = re.match(r'v(\d+)$', tok)
        if r: return {int(r.group(1))}
        return set()
    flags = 0
    for ln, l in enumerate(body):
        l = re.sub(r';.*', '', l).strip()
        if not l or l.endswith(':') or l.startswith('.'): continue
        op, _, rest = l.partition(' ')
        toks = [t.strip() for t in rest.split(',')]
        if op == 's_waitcnt':
            r = re.search(r'lgkmcnt\((\d+)\)', rest)
            if r: n = int(r.group(1)); lg = lg[len(lg) - n:] if n else []
            r = re.search(r'vmcnt\((\d+)\)', rest)
            if r: n = int(r.group(1)); vm = vm[len(vm) - n:] if n else []
            continue
        used = set()
        for t in toks:
            for tt in re.findall(r'v\[\d+:\d+\]|v\d+\b', t): used |= regs(tt)
        for q, what in ((lg, 'lds'), (vm, 'vmem')):
            for d in q:
                if d & used:
                    flags += 1
                    if flags <= 40 and not l.startswith("global_load"): print(f'{name[:50]} line {ln}: {l}   <- {what} result v{min(d)}.. may be pending')
                    break
        if op.startswith('ds_'):
            lg.append(regs(toks[0]) if ('read' in op or 'load' in op) else set())
        elif op.startswith('global_load') or op.startswith('buffer_load') or op.startswith('scratch_load'):
            vm.append(regs(toks[0]))
        elif op.startswith('global_store') or op.startswith('buffer_store') or op.startswith('scratch_store') or op.startswith('global_atomic'):
            vm.append(set())
        elif op.startswith('s_load') or op.startswith('s_buffer_load'):
            lg.append(set())
    print(name[:70], 'flags', flags)
