# Dev: variant libraries of k_linear_bwd with one piece compiled out each (LB_ABL bits: 1 loads, 2 dx stores, 4 weight-gradient MFMAs,
# 8 data GEMM, 16 x operand reads + splits, 32 barriers); timing only - the results are wrong
cd /root/repo/magnet_amd/csrc
OBJS=$(ls *.o | grep -v "\.dev\.o" | grep -v "^linear_bwd" | grep -v "^_")
for A in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DLB_ABL=$A -c linear_bwd.hip -o _lb_abl$A.o && \
  hipcc -shared -fPIC --offload-arch=gfx950 $OBJS _lb_abl$A.o -o libmagnet_hip_abl$A.so && echo built abl$A &
done; wait; rm -f _lb_abl*.o
