#!/bin/bash
# Dev: the C5 E-level chain with and without the tape, product library against variant builds (MAGNET_LIB_PATH), alternating.
cd "$(dirname "$0")/../.."
for r in 1 2; do
  for lib in "" $(ls magnet_amd/csrc/libmagnet_hip_mlp*.so 2>/dev/null); do
    for t in "" "--tape"; do
      MAGNET_LIB_PATH=$lib python scripts/dev/bench_chain_abl_c5.py $t 2>&1 | tail -1
    done
  done
done
