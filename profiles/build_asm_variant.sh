#!/bin/bash
# Build a libmi355pose variant whose igemm DEVICE code comes from a (hand-edited) assembly file: ISA-level bisection of the
# round-2 dropped-addend fault.  usage: build_asm_variant.sh <igemm source dir> <device.s> <out.so>
#   <igemm source dir>: a copy of csrc/ whose igemm.hip matches the .s (host stubs and kernel names must agree)
set -e
SRC=$1; ASM=$2; OUT=$3
LL=/opt/rocm/lib/llvm/bin
B=/root/repo/domain-adaptative-hand-pose-estimation_amd/csrc/_build
T=$(mktemp -d)
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $ASM -o $T/dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
(cd $SRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c igemm.hip -o $T/igemm.o 2>&1 | grep -v "warning\|^$" || true)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $T/igemm.o $B/api.o $B/pgemm.o $B/igemm_fp8.o $B/bn.o $B/pool_layout.o $B/pw21.o $B/heatmap.o $B/optim.o
rm -rf $T
echo built $OUT
