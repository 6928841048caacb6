#!/bin/bash
# usage (here, no GPU): tools/peel_isa.sh [out.s] — the gfx950 ISA of k_peel<false> with phase markers (; C2AMARK ...) at the
# points where the statistics build reads the clock; for counting instructions per phase of the chain step
R=/root/repo; out=${1:-/tmp/isa/peel_m.s}; mkdir -p $(dirname $out)
tmp=$(mktemp -d); mkdir -p $tmp/p/q $tmp/p/include; cp -r $R/circom-2-arithc_amd/csrc $tmp/p/q/csrc; cp $R/include/c2a.h $tmp/p/include/
python3 - $tmp/p/q/csrc/c2a_peel.h <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
for k in ("ph0","ph1","ph2","ph3"):
    s=s.replace('const ull %s = STATS ? c2a_now() : 0;'%k,'asm volatile("; C2AMARK %s"); const ull %s = STATS ? c2a_now() : 0;'%(k,k))
s=s.replace('            ++processed;\n','            ++processed; asm volatile("; C2AMARK ph4");\n')
open(p,'w').write(s)
PY
cd $tmp/p/q/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=None $EXTRA -S --cuda-device-only -o $tmp/c2a.s c2a_api.hip 2>/dev/null
awk '/^_ZN3c2a6k_peelILb0ELb0EEEvNS_8PeelArgsE:/,/\.Lfunc_end/' $tmp/c2a.s > $out
grep -n "C2AMARK" $out
rm -rf $tmp
