# Build variant libraries of the chain kernel source (csrc/mlp_bf16_kernels.hip):
#   magnet_amd/csrc/variants/libmagnet_hip_bl<k>.so  =  the product objects with that one source recompiled with
#     k = 1   biases from LDS WITHOUT the wait states behind the inline-asm split: the round-2 failure (scripts/dev/stress_bias_lds.py)
#     k = 11  the same WITH them (= the product kernel)          k = 0   biases from global memory, wait states on (A/B)
#   (k = 2 .. 10 were the variants of the root-cause hunt, profiles/r3/bias_lds_root_cause.txt; their code is no longer in the source)
# usage: bash scripts/dev/build_bias_lds_variants.sh 1 11 0      (after `python -m magnet_amd.build`)
cd "$(dirname "$0")/../../magnet_amd/csrc" || exit 1
mkdir -p variants
OTHERS=$(ls *.o | grep -v '\.dev\.o$' | grep -v '^mlp_bf16_kernels\.o$')
for k in "$@"; do
  D="-DMAGNET_BIAS_LDS=$k -DMAGNET_MIX_TAIL=\"\""                 # the experiments ran without the wait states that fix it ...
  [ $k = 0 ] && D="-DMAGNET_BIAS_LDS=0"                            # 0 = biases from global memory, wait states on (A/B against the product)
  [ $k = 11 ] && D="-DMAGNET_BIAS_LDS=1"                           # ... 11 = variant 1 WITH them (common.cuh: MAGNET_MIX_TAIL)
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $D -c mlp_bf16_kernels.hip -o variants/mlp_bf16_kernels_bl$k.o 2>/dev/null \
    && hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS variants/mlp_bf16_kernels_bl$k.o -o variants/libmagnet_hip_bl$k.so && echo built bl$k ) &
  if [ $(jobs -r | wc -l) -ge 4 ]; then wait -n; fi
done
wait
rm -f variants/*.o
ls -la variants
