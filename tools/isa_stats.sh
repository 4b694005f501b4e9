#!/bin/bash
# Build container (no GPU needed): compile the library for gfx950 with the given extra flags and print, per kernel,
# registers, spills, occupancy, LDS and a few instruction counts from the ISA listing.
#   tools/isa_stats.sh [-DFLAG ...]      (listing left in /tmp/isa/mi355rast-hip-amdgcn-amd-amdhsa-gfx950.s)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/isa && cd /tmp/isa || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 --save-temps "$@" \
    -o /tmp/isa/lib.so "$ROOT/py-numpy-renderer_amd/csrc/mi355rast.hip" 2>/dev/null || { echo "compile failed"; exit 1; }
python3 - <<'PY'
import re
s = open('/tmp/isa/mi355rast-hip-amdgcn-amd-amdhsa-gfx950.s').read()
# a kernel: its label, its body up to .Lfunc_end, then the "; Kernel info:" comment block
for m in re.finditer(r'^(_ZN2mr\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:\n(.*?)COMPUTE_PGM_RSRC2:USER_SGPR', s, re.S | re.M):
    name, body, info = m.group(1), m.group(2), m.group(3)
    if '.amdhsa_kernel' not in body and 'Kernel info' not in info:
        continue
    short = re.sub(r'^_ZN2mr\d+', '', name)[:24]
    def g(key):
        r = re.search(r'; %s: (\d+)' % key, info)
        return int(r.group(1)) if r else -1
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    valu = sum(1 for i in ins if i.startswith('v_'))
    f64 = sum(1 for i in ins if i.startswith('v_') and 'f64' in i)
    rl = sum(1 for i in ins if i.startswith('v_readlane') or i.startswith('v_writelane'))
    sm = sum(1 for i in ins if i.startswith('s_load') or i.startswith('s_buffer_load'))
    vm = sum(1 for i in ins if i.split('_')[0] in ('global', 'buffer', 'flat', 'scratch'))
    ds = sum(1 for i in ins if i.startswith('ds_'))
    print(f"{short:24s} vgpr {g('NumVgprs'):3d} sgpr {g('TotalNumSgprs'):3d} scratch {g('ScratchSize'):3d} occ {g('Occupancy'):2d} "
          f"lds {g('LDSByteSize'):6d} | instr {len(ins):5d} valu {valu:5d} f64 {f64:4d} rd/wrlane {rl:4d} smem {sm:3d} vmem {vm:3d} ds {ds:3d}")
PY
