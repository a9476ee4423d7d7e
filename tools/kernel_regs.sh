#!/bin/bash
# VGPR / SGPR / LDS / scratch of every kernel of one HIP source (cross-compiles, no GPU needed): tools/kernel_regs.sh csrc/kernels_mg.hip
cd "$(dirname "$0")/../fluid-simulation_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S "$1" -o /tmp/kregs.s 2>/dev/null || exit 1
python3 - <<'PY'
import re, subprocess
out={}; cur=None
for line in open('/tmp/kregs.s'):
    if re.match(r'\s+- \.\w+:', line): cur={}          # a new kernel's metadata block (keys are sorted: .name comes late)
    m=re.match(r'\s+(?:- )?\.name:\s+(\S+)',line)
    if m and cur is not None and line.startswith('    .name') : out[m.group(1)]=cur
    for k in ('vgpr_count','sgpr_count','group_segment_fixed_size','private_segment_fixed_size','vgpr_spill_count'):
        m=re.match(r'\s+(?:- )?\.%s:\s+(\d+)'%k,line)
        if m and cur is not None: cur[k]=int(m.group(1))
for n,v in out.items():
    d=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
    d=re.sub(r'\(.*','',d.replace('void ','').replace('(anonymous namespace)::','').replace('fl::',''))
    print(f"{d:60s} vgpr {v.get('vgpr_count',0):4d} sgpr {v.get('sgpr_count',0):4d} lds {v.get('group_segment_fixed_size',0):7d} scratch {v.get('private_segment_fixed_size',0):5d} spill {v.get('vgpr_spill_count',0)}")
PY
