#!/bin/bash
# Remove A/B build leftovers from csrc/ so that a gpurun push carries only libvbs.so + libvbs_dbg.so and their objects.
# Keeps: <source>.o, <source>_dbg.o, libvbs.so, libvbs_dbg.so, k_blur.s.ok, k_blur_dbg.s.ok.
set -e
cd "$(dirname "$0")/../vision-basedsensor_amd/csrc"
for f in *.o *.so *.s.ok *.s; do
  [ -e "$f" ] || continue
  base="${f%%.*}"
  case "$base" in
    libvbs|libvbs_dbg) continue ;;
  esac
  src="${base%_dbg}"
  if [ -e "$src.hip" ]; then continue; fi
  rm -f -- "$f"
done
rm -rf .pytest_cache
ls *.so
