# Dev: variant libraries of k_mpnn_edge_bwd_co with pieces compiled out (CO_ABL bits: 1 barriers, 2 dW2 MFMAs, 4 GEMM 1 / 2 MFMAs,
# 8 dp stores, 16 transcendentals; extra -D... after the list as CO_DEFS="..."); timing only - the results are wrong
cd /root/repo/magnet_amd/csrc
OBJS=$(ls *.o | grep -v "\.dev\.o" | grep -v "^mpnn_edge_bwd_co" | grep -v "^_" | grep -v "^mpnn_edge_rows\|^mpnn_edge32t\|^mpnn_edge_ws")
for A in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -fno-slp-vectorize -DCO_ABL=$A $CO_DEFS -c mpnn_edge_bwd_co.hip -o _co_abl$A.o && \
  hipcc -shared -fPIC --offload-arch=gfx950 $OBJS _co_abl$A.o -o libmagnet_hip_co$A.so && echo built co$A &
done; wait; rm -f _co_abl*.o
