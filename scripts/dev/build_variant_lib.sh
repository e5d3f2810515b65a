# Build a complete variant of the library from scratch with extra compile flags (objects under /tmp, library under csrc/variants/):
#   bash scripts/dev/build_variant_lib.sh notail '-DMAGNET_MIX_TAIL=""'
NAME=$1; shift
SRC="$(cd "$(dirname "$0")/../../magnet_amd/csrc" && pwd)"; OBJ=/tmp/variant_$NAME; mkdir -p $OBJ $SRC/variants
for f in $SRC/*.hip; do
  b=$(basename ${f%.hip}); extra=""; [ $b = readout_kernels ] && extra="-mllvm -pragma-unroll-threshold=200000"
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $extra "$@" -c $f -o $OBJ/$b.o 2>/dev/null || echo FAILED $b ) &
  if [ $(jobs -r | wc -l) -ge 6 ]; then wait -n; fi
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 $OBJ/*.o -o $SRC/variants/libmagnet_hip_$NAME.so && echo built $SRC/variants/libmagnet_hip_$NAME.so
