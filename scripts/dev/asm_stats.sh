#!/bin/bash
# usage: scripts/dev/asm_stats.sh magnet_amd/csrc/file.hip kernel_name_substring  -> register / spill / instruction mix summary
f=$1; k=$2
out=$(mktemp /tmp/asmXXXXXX.s)
hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only $ASM_STATS_FLAGS -S "$f" -o "$out" 2>/dev/null   # ASM_STATS_FLAGS: e.g. "-mllvm -pragma-unroll-threshold=200000" for readout_kernels.hip
python3 - "$out" "$k" <<'PY'
import re,sys,collections
s=open(sys.argv[1]).read(); key=sys.argv[2]
for m in re.finditer(r'^(_Z\w*'+re.escape(key)+r'\w*):', s, re.M):
    name=m.group(1)
    body=s[m.end():]; body=body[:body.index('.Lfunc_end')]
    ins=collections.Counter()
    for line in body.splitlines():
        mm=re.match(r'^\s+([a-z_0-9]+)\s',line)
        if mm: ins[mm.group(1)]+=1
    meta=s[s.index('.name:           '+name):][:1500]
    g=lambda k:(re.search(k+r':\s+(\d+)',meta) or [0,0])[1]
    sel=['v_accvgpr_read_b32','v_accvgpr_write_b32','v_mov_b32_e32','scratch_load_dwordx4','scratch_store_dwordx4','v_exp_f32_e32','v_rcp_f32_e32','s_nop','s_waitcnt']
    print(name[:60], 'vgpr',g('.vgpr_count'),'agpr',g('.agpr_count'),'spill',g('.vgpr_spill_count'),'scratch',g('.private_segment_fixed_size'),'total',sum(ins.values()), ' '.join(f"{k.replace('v_accvgpr_','acc_').replace('_b32','').replace('_e32','')}={ins[k]}" for k in sel))
PY
rm -f "$out"
