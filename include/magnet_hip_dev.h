/* Development-only entry points of libmagnet_hip.so.  They exist ONLY in a library built with -DMAGNET_DEV_VARIANTS
 * (`python -m magnet_amd.build --dev`); the product library (what `__graft_entry__.build()` makes and the tests / bench.py
 * load) exports neither symbol and contains none of the kernel variants they select.  Some of those variants are ablations
 * that produce INVALID results; they exist to attribute time (scripts/bench_edge.py, scripts/bench_mlp_abl.py). */
#ifndef MAGNET_HIP_DEV_H
#define MAGNET_HIP_DEV_H
#ifdef __cplusplus
extern "C" {
#endif
/* which = 0: kernel behind magnet_mpnn_edge_f32 (102 = product); which = 1: generation / ablation of the split-GEMM edge kernels;
 * which = 2: ablation flags of k_mpnn_edge32t (1 no MFMA, 2 no epilogue transcendentals, 4 no input Swish / split, 8 L1-resident
 * gathers, 16 no LDS weight reads; only the combinations instantiated in mpnn_edge32t.hip) */
int magnet_debug_set_variant(int which, int value);
/* run-time ablation flags of the split-GEMM MLP kernel */
int magnet_debug_set_mlp_ablation(int value);
/* cycle trace (s_memtime stamps + phase tags) of wave 0 of workgroup `wg` of the split-GEMM MLP kernel into buf[1000] (uint64) */
int magnet_debug_set_mlp_trace(void* buf, int wg);
/* run-time ablation flags of the fused MPNN edge backward kernel (csrc/mpnn_edge_bwd.hip) */
int magnet_debug_set_edge_bwd_ablation(int value);
#ifdef __cplusplus
}
#endif
#endif
