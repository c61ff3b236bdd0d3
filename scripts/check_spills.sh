#!/bin/bash
# dev tool: where do the register spills of the default stage-1 kernel sit?  A scratch reload inside the
# count phase makes hipcc wait for vmcnt(0), i.e. for the LDS-DMA prefetch of the NEXT rounds: -9 %.
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icsv-simd_amd/csrc $EXTRA -S --cuda-device-only \
    csv-simd_amd/csrc/stage1_kernels.hip -o /tmp/spillcheck.s 2>/dev/null
for k in "ILb1ELi0ELi0ELb0E" "ILb0ELi0ELi0ELb0E" "ILb1ELi0ELi1ELb0E" "ILb1ELi0ELi2ELb0E" "ILb1ELi0ELi3ELb0E" "ILb1ELi0ELi0ELb1E"; do
  awk "/^_ZN7csvsimd13stage1_kernel${k}EEvNS_10KernelArgsE:/,/s_endpgm/" /tmp/spillcheck.s > /tmp/spillcheck_k.s
  first=$(grep -n "buffer_load_dwordx4.* lds" /tmp/spillcheck_k.s | head -1 | cut -d: -f1)
  last=$(grep -n "buffer_load_dwordx4.* lds" /tmp/spillcheck_k.s | tail -1 | cut -d: -f1)
  end=$(awk -v l="$last" 'NR>l && /s_barrier/ {print NR; exit}' /tmp/spillcheck_k.s)
  inside=$(awk -v a="$first" -v b="$end" 'NR>=a && NR<=b && /scratch_/' /tmp/spillcheck_k.s | wc -l)
  total=$(grep -c "scratch_" /tmp/spillcheck_k.s || true)
  echo "stage1_kernel<$k>: count phase = lines $first..$end, scratch ops inside: $inside (of $total)"
done
