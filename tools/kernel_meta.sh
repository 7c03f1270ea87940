#!/bin/bash
# Registers / LDS / spills of the kernels of one csrc/*.hip unit (device-only compile, no GPU needed) and its gfx950 disassembly:
#   bash tools/kernel_meta.sh cnn.hip [name filter]      -> table on stdout, /tmp/iss_meta/<unit>.s
ROOT=$(cd "$(dirname "$0")/.." && pwd)
U=${1:-cnn.hip}; F=${2:-.}
B=$(basename $U .hip); O=/tmp/iss_meta; mkdir -p $O
LL=/opt/rocm/lib/llvm/bin
cd $ROOT/inaspeechsegmenter_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -mllvm -pragma-unroll-threshold=100000 \
    -I../../include --cuda-device-only -c $U -o $O/$B.bundle 2>/dev/null || exit 1
$LL/clang-offload-bundler --unbundle --type=o --input=$O/$B.bundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$O/$B.elf
$LL/llvm-objdump -d --mcpu=gfx950 $O/$B.elf > $O/$B.s 2>/dev/null
$LL/llvm-readelf --notes $O/$B.elf | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|spill|group_segment_fixed|agpr_count" | paste - - - - - - - | sed 's/  */ /g' | grep -E "$F" |
  awk '{for(i=1;i<=NF;i++){if($i==".name:")n=$(i+1); if($i==".vgpr_count:")v=$(i+1); if($i==".agpr_count:")a=$(i+1); if($i==".group_segment_fixed_size:")l=$(i+1); if($i==".vgpr_spill_count:")vs=$(i+1); if($i==".sgpr_spill_count:")ss=$(i+1); if($i==".sgpr_count:")sg=$(i+1)} printf "%s vgpr %3d agpr %3d sgpr %3d lds %6d spill v%d s%d\n", n, v, a, sg, l, vs, ss}' | c++filt | sed 's/issk::ConvArgs/ConvArgs/; s/(anonymous namespace):://'
