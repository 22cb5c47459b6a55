#!/bin/bash
# Development tool: registers, spills and LDS of every kernel in a built library.
#   usage: tools/kernel_regs.sh [lib.so] [name filter]
LIB=${1:-ray_tracer_amd/csrc/libmirt.so}; FILTER=${2:-k_paths}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$LIB --output=$TMP/dev.co --unbundle 2>/dev/null ||
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $LIB && /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk).group(1)
    if '$FILTER' not in name: continue
    g=lambda k: re.search(r'\.'+k+r':\s+(\S+)',blk).group(1)
    import subprocess
    dem=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    print(f\"vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} spill_v {g('vgpr_spill_count'):>3} spill_s {g('sgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}  {dem[:110]}\")
"
rm -rf $TMP
