"""Dev: instruction mix of the largest loop of a kernel in a hipcc -save-temps .s file.  usage: loop_mix.py file.s kernel_substring"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\w*' + re.escape(sys.argv[2]) + r'\w*):', s, re.M)
body = s[m.end():]
body = body[:body.index('.Lfunc_end')]
lines = body.splitlines()
# the extent of the largest loop: hipcc comments every block of a loop with "in Loop: Header=BBn_m"
hdrs = collections.Counter(re.findall(r'Header=(BB\d+_\d+) Depth=1', body))
hdr = hdrs.most_common(1)[0][0]
first = next(i for i, l in enumerate(lines) if re.match(r'^\.L' + hdr + ':', l))
last = max(i for i, l in enumerate(lines) if 'Header=' + hdr + ' ' in l or 'Header=' + hdr + '\n' in l + '\n')
# ... up to the end of the last block of the loop
while last + 1 < len(lines) and not re.match(r'^\.LBB', lines[last + 1]):
    last += 1
best = (first, last)
loop = lines[best[0]:best[1]]
ins = collections.Counter()
for l in loop:
    mm = re.match(r'^\s+([a-z_0-9]+)', l)
    if mm:
        ins[mm.group(1)] += 1
cat = collections.Counter()
for k, v in ins.items():
    c = ('mfma' if k.startswith('v_mfma') else 'trans' if re.match(r'v_(exp|rcp|log|sqrt|rsq|sin|cos)', k) else 'valu' if k.startswith('v_')
         else 'lds' if k.startswith('ds_') else 'vmem' if re.match(r'(global|buffer|flat|scratch)_', k) else 'waitcnt' if k.startswith('s_waitcnt')
         else 'nop' if k.startswith('s_nop') else 'smem' if re.match(r's_(load|buffer)', k) else 'salu' if k.startswith('s_') else 'other')
    cat[c] += v
print(f"loop lines {best}: {dict(cat)}")
print("  ".join(f"{k}={v}" for k, v in ins.most_common(40)))
