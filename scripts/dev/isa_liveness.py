"""Dev: approximate VGPR/AGPR liveness over a kernel's ISA (straight-line approximation, loop body iterated twice): where is the register
pressure?  usage: python scripts/dev/isa_liveness.py file.s kernel_substring"""
import re
import sys

s = open(sys.argv[1]).read()
i = s.index(sys.argv[2])
body = s[i:]
body = body[:body.index('.Lfunc_end')]
lines = [l for l in body.splitlines() if re.match(r'^\s+[a-z]', l) and not l.strip().startswith(('.', ';'))]
reg = re.compile(r'\b([va])(?:\[(\d+):(\d+)\]|(\d+))\b')


def regs(tok):
    out = set()
    for m in reg.finditer(tok):
        if m.group(4) is not None:
            out.add((m.group(1), int(m.group(4))))
        else:
            out.update((m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


NODEF = ('global_store', 'scratch_store', 'ds_write', 'buffer_store', 's_', 'v_cmp', 'v_cmpx', 'global_atomic', 'v_readfirstlane', 'v_readlane')
ins = []
for l in lines:
    t = l.split(';')[0].strip()
    mn = t.split()[0]
    ops = t[len(mn):].split(',')
    if mn.startswith(NODEF):
        d, u = set(), set().union(*[regs(o) for o in ops]) if ops else set()
    else:
        d = regs(ops[0]) if ops else set()
        u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        if 'mfma' in mn:
            pass
        if mn.startswith(('v_fma_mixhi', 'v_fma_mixlo', 'v_mac', 'v_fmac', 'v_pk_fmac', 'v_writelane', 'v_dot2c')):
            u |= d                                  # read-modify-write destinations
    ins.append((mn, d, u, l))
live = set()
prof = [0] * len(ins)
for _ in range(2):
    for k in range(len(ins) - 1, -1, -1):
        mn, d, u, _l = ins[k]
        live -= d
        live |= u
        prof[k] = len(live)
mx = max(prof)
print('instructions', len(ins), 'max live', mx)
bar = 0
seg_max = {}
for k, (mn, d, u, l) in enumerate(ins):
    if mn == 's_barrier':
        bar += 1
    seg_max[bar] = max(seg_max.get(bar, 0), prof[k])
print('max live per barrier segment:', seg_max)
if len(sys.argv) > 3:                                   # list what is live at the first instruction of barrier segment N, by defining instruction
    want = int(sys.argv[3])
    bar = 0
    at = None
    for k, (mn, d, u, l) in enumerate(ins):
        if mn == 's_barrier':
            bar += 1
            if bar == want:
                at = k + 1
                break
    live = set()
    for _ in range(2):
        for k in range(len(ins) - 1, at - 1, -1):
            mn, d, u, _l = ins[k]
            live -= d
            live |= u
        if _ == 0:
            # continue around the loop: instructions before `at`
            for k in range(at - 1, -1, -1):
                pass
    # defining instruction of each live register: scan backwards from `at`
    defs = {}
    for r in live:
        for k in range(at - 1, -1, -1):
            if r in ins[k][1]:
                defs[r] = k
                break
    import collections
    by = collections.Counter()
    for r, k in defs.items():
        by[(ins[k][0], k // 50 * 50)] += 1
    print('live at segment', want, ':', len(live), 'undefined-before:', len(live) - len(defs))
    for (mn, pos), n in sorted(by.items(), key=lambda kv: kv[0][1]):
        print(f'  defined by {mn:28s} near instruction {pos:5d}: {n}')
