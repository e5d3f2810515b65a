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
/* ---- earlier generations of the fused MPNN edge op (GNN_Layer.message + mean, models/mpnn.py:73-79,46), all measured slower than or
 * equal to the two the product library keeps (k_mpnn_edge16, k_mpnn_edge_stream32) - DESIGN.md 11.1, 14.2; csrc/mpnn_edge32t.hip,
 * mpnn_edge_ws.hip, mpnn_edge_rows.hip, mpnn_edge_rowsr.hip, mpnn_edge_rowsp.hip are compiled into the development library only. */
/* development bits of magnet_mpnn_edge_stream_fp16x3_f32's flags: four gather sets in flight; s_setprio 1 for the younger wave */
#define MAGNET_STREAM_DEV_GATHER4 4
#define MAGNET_STREAM_DEV_PRIO 8
/* The fp16x3 edge op by the TRANSPOSED 32-edge-tile kernel (k_mpnn_edge32t, generation 9): the tile's edges are
 * the rows of the 32x32x16 MFMA, a lane owns an output feature, the per-target mean is a running sum over accumulator registers
 * (no LDS transposition), and the three products of the split share ONE accumulator.  Same result contract as
 * magnet_mpnn_edge_fp16x3_f32.  w2_t32: fp16 planes [8 k slots][4 output blocks][2 planes][64 lanes][8 fp16], both pre-scaled by
 * 2^11 (magnet_amd/packing.py:pack_edge32t_fp16x2; needs |w| < 31.98); b2 the plain bias; a_bm_log2e_scaled as for _t16.
 * partials: scratch, ceil(n_edges / 32) * 2 * 128 floats (two rows per wave range, at most one range per tile). */
int magnet_mpnn_edge_fp16x3_t32_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                    const int32_t* rowptr, int n_edges, int n_rows, const void* w2_t32, const float* b2,
                                    float* agg, float* partials, int a_bm_log2e_scaled, void* stream);
/* The same op, WAVE-SPECIALISED (k_mpnn_edge_ws, generation 10): eight lean waves per workgroup - two gather whole
 * rows and apply the first Swish, four run the transposed GEMM with their slice of W2 resident in registers, two apply the second
 * Swish and the running mean - pipelined over consecutive tiles through LDS, two workgroups per CU.  Arguments and result contract
 * as magnet_mpnn_edge_fp16x3_t32_f32 (w2_t32 from packing.pack_edge32t_fp16x2; a_bm_log2e_scaled 0 / 1). */
int magnet_mpnn_edge_fp16x3_ws_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                   const int32_t* rowptr, int n_edges, int n_rows, const void* w2_t32, const float* b2,
                                   float* agg, float* partials, int a_bm_log2e_scaled, void* stream);
/* The fp16x3 edge op on tiles ALIGNED TO CSR ROWS (k_mpnn_edge_rows32, generation 11) - GNN_Layer.message + the mean
 * of models/mpnn.py:73-79,46 for the rows that own a tile.  tiles: n_tiles descriptors of four int32 {first edge, row, edges in the
 * tile (1..32) | (edges of the row outside the tile) << 8, degree of the row}, at most one per row, rows ascending.  strag_row /
 * strag_src: the STRAGGLERS - (target, source) of the one edge a row has beyond its tile when there is exactly one (the 33rd edge of
 * an MPNN row) - rows ascending, strag_ptr[r] = stragglers of the rows before r ([n_rows + 1]; NULL = none); every wave sends the
 * stragglers of its own rows through the same pipeline first and leaves their messages in agg[row].  Rows whose tile leaves two or more edges outside must hold the MEAN over those edges in agg[row] on entry
 * (magnet_mpnn_edge_fp16x3_t16_f32 on the remainder edge list leaves exactly that).  On return agg[row] is the mean over all of the
 * row's edges for every row with a tile; rows without a tile are not touched.  src: the full CSR's sources [n_edges];
 * w2_t32 / b2 / a_bm_log2e_scaled (bit 0) as for magnet_mpnn_edge_fp16x3_t32_f32.  No scratch.
 * magnet_row_tiles_count / _fill build the plan. */
int magnet_mpnn_edge_rows_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const void* tiles, int n_tiles,
                                     const int32_t* strag_ptr, const int32_t* strag_row, const int32_t* strag_src,
                                     const void* w2_t32, const float* b2, float* agg, int a_bm_log2e_scaled, void* stream);
/* The same op with message_net_2's planes resident in the accumulation half of the register file (k_mpnn_edge_rowsr, generation 12;
 * DESIGN.md 14.2): no LDS weight-fragment reads, one wave per SIMD.  Arguments and contract exactly as
 * magnet_mpnn_edge_rows_fp16x3_f32. */
int magnet_mpnn_edge_rowsr_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const void* tiles, int n_tiles,
                                      const int32_t* strag_ptr, const int32_t* strag_row, const int32_t* strag_src,
                                      const void* w2_t32, const float* b2, float* agg, int a_bm_log2e_scaled, void* stream);
/* Generation 12b (k_mpnn_edge_rowsg): as magnet_mpnn_edge_rowsr_fp16x3_f32 with the stragglers GROUPED WITH THEIR ROWS - every wave
 * runs the 33rd edges of its next 32 rows as one tile right in front of those rows' tiles and parks their messages in LDS; the
 * straggler lists are not needed (a tile descriptor with exactly one edge outside the tile says where that edge is: first edge + 32).
 * has_stragglers = 0: no row has one (the straggler tiles are left out).  Rows with two or more edges outside their tile: the mean
 * over those in agg[row] on entry, as for the two kernels above. */
int magnet_mpnn_edge_rowsg_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const void* tiles, int n_tiles,
                                      int has_stragglers, const void* w2_t32, const float* b2, float* agg, int a_bm_log2e_scaled,
                                      void* stream);
/* Generation 13 (k_mpnn_edge_rowsp): W2's planes in the accumulation registers, one wave per SIMD, the tile's 96 MFMAs in
 * four sub-phases with the previous tile's epilogue and the next tile's input production between them; every row parks the activations
 * of its 33rd edge in LDS and a group's 32 parked edges run as one tile behind the group.  Arguments and contract as
 * magnet_mpnn_edge_rowsg_fp16x3_f32. */
int magnet_mpnn_edge_rowsp_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const void* tiles, int n_tiles,
                                      int has_stragglers, const void* w2_t32, const float* b2, float* agg, int a_bm_log2e_scaled,
                                      void* stream);
/* The tile plan of a CSR graph for the four kernels above.  _count, per row r: has_tile[r] = (deg_r >= pmin); has_strag[r] = the row has a
 * tile and exactly one edge beyond it; n_rem[r] = its edges for the remainder pass (those outside the tile unless that is the one
 * straggler; all of them without a tile).  _fill: given the exclusive prefix sums tile_ptr / strag_ptr / rem_ptr [n_rows + 1] of the
 * three arrays, writes the descriptors, the straggler list and the remainder edge list
 * (src_rem, tgt_rem; still sorted by target, rem_ptr is its rowptr). */
int magnet_row_tiles_count(const int32_t* rowptr, int n_rows, int pmin, int32_t* has_tile, int32_t* has_strag, int32_t* n_rem,
                           void* stream);
/* The plan's three prefix arrays in one call (no host-side scans): ptrs [3][n_rows + 1] int32 receives the exclusive prefix sums of the
 * per-row has_tile / has_strag / n_rem of magnet_row_tiles_count; ptrs[k][n_rows] = number of tiles / stragglers / remainder edges.
 * Feed ptrs, ptrs + (n_rows + 1), ptrs + 2 (n_rows + 1) to magnet_row_tiles_fill. */
int magnet_row_tiles_ptrs(const int32_t* rowptr, int n_rows, int pmin, int32_t* ptrs, void* stream);
int magnet_row_tiles_fill(const int32_t* rowptr, const int32_t* src, int n_rows, const int32_t* tile_ptr, const int32_t* strag_ptr,
                          const int32_t* rem_ptr, void* tiles, int32_t* strag_row, int32_t* strag_src, int32_t* src_rem,
                          int32_t* tgt_rem, void* stream);
#ifdef __cplusplus
}
#endif
#endif
