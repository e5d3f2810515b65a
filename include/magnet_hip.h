/* magnet_hip.h - C ABI of libmagnet_hip.so: MI355X (gfx950) kernels for the MAgNet rollout hot path.
 *
 * Boundary.  The reference (jaggbow/magnet) is pure Python; its hot path reaches native code only
 * through third-party wheels (ATen, torch_scatter, torch_cluster, PyG).  This library sits where
 * those wheels sit: plain device pointers + sizes + a hipStream_t (passed as void*), nothing from
 * torch in any signature.  Every function
 *   - enqueues on `stream` and returns immediately (never allocates, never synchronises),
 *   - returns 0 on success, a negative value for an unsupported argument, or a hipError_t,
 *   - borrows all buffers for the duration of the enqueued work; the caller owns them.
 * All floating point is fp32 (the reference runs precision=32, configs/trainer/default.yaml:5);
 * indices are int32 on the device (the reference's int64 edge_index is converted at the boundary).
 * Citations are file:line in the reference repository.
 */
#ifndef MAGNET_HIP_H
#define MAGNET_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Packed weight layout ("fragment order") used by every Linear in this library.
 * A torch Linear weight W[N_out][K] (row-major, models/backbones/mlp.py:16-21) is zero-padded to
 * NT*32 x KT*32 and stored as float4 Wp[KT][NT][4][64]:
 *     Wp[kt][nt][g][lane] = W[nt*32 + (lane & 31)][kt*32 + 8*g + 4*(lane >> 5) + 0..3]
 * (one 1 KiB coalesced wave load per [kt][nt][g]); the bias is zero-padded to NT*32 floats.
 * magnet_amd/packing.py builds it; a C caller can use the same two lines of index arithmetic.
 * ---------------------------------------------------------------------------------------------- */

#define MAGNET_MAX_KT 36
#define MAGNET_MAX_LAYERS 6
enum { MAGNET_ACT_NONE = 0, MAGNET_ACT_SWISH = 1, MAGNET_ACT_RELU = 2 };
enum { MAGNET_EPI_STORE = 0, MAGNET_EPI_STORE_RES = 1, MAGNET_EPI_SEGMEAN = 2,
       MAGNET_EPI_STORE_MASK = 3 /* nl == 1: y = res > 0 ? x W^T : 0 - a backward-data GEMM with the ReLU mask of the layer below */,
       MAGNET_EPI_QUADBLEND = 4 /* y[r / 4] = sum of res[row] * mlp(x)[row] over the rows 4 (r / 4) .. + 3 (res: [M] weights, M % 4 == 0):
                                   the area blend of the 2-D grid INR decoder, models/magnet_cnn_2d.py:285-290; split-GEMM kernels only */ };

/* One source feeding a 32-wide k tile of the first Linear's input:
 *   value[row][k] = scale * p[(idx ? idx[row] : row / rdiv) * ld + col + k],  k = 0..31
 * p, ld and col must keep every 4-float group 16-byte aligned. */
typedef struct {
    const float* p;
    const int32_t* idx;
    int32_t ld, col;
    float scale;
    int32_t rdiv;
} magnet_src_t;
typedef struct {
    magnet_src_t s[2]; /* n == 2: the two sources are summed (u_i - u_j style differences) */
    int32_t n;
    int32_t pad_[3];
} magnet_ktile_t;
typedef struct {
    const void* w;  /* packed weights */
    const float* b; /* padded bias    */
} magnet_layer_t;
typedef struct {
    int32_t M;   /* rows (nodes, CSR-ordered edges or INR samples) */
    int32_t nkt; /* k tiles of the first layer (K_padded / 32) */
    magnet_ktile_t kt[MAGNET_MAX_KT];
    magnet_layer_t layer[MAGNET_MAX_LAYERS];
    const float* ln_g; /* LayerNorm gain / bias over the OT*32 outputs (eps = ln_eps) */
    const float* ln_b;
    float ln_eps;
    float* y; /* STORE / STORE_RES: [M][ldy], first ncols columns.  SEGMEAN: agg [N][ncols] */
    int32_t ldy, ncols;
    const float* res; /* STORE_RES: y = mlp(x) + res */
    int32_t ldres;
    const int32_t* tgt;    /* SEGMEAN: target row of every edge (non-decreasing) */
    const int32_t* rowptr; /* SEGMEAN: CSR row pointer [N+1] */
    float* partials;       /* SEGMEAN: scratch [ceil(M/32)][2][ncols] */
    int32_t ncolgroups;    /* nl == 1 only, 0/1 = off: G > 1 launches G workgroup columns; group g multiplies by layer[g]
                            * (the packed slice of its OT*32 output columns) and writes y / reads res at column g*OT*32 */
    int32_t groups_in_block; /* 0/1: the G groups are workgroup columns (gridDim.y); G: every workgroup runs its rows through
                              * the G groups one after the other (the input rows are then read from HBM once) */
    int64_t y_group_stride;  /* floats between the outputs of consecutive groups; 0 = OT*32 (neighbouring columns) */
    magnet_src_t pre[2];     /* nl > 1 only, p == NULL = off: rows added to the first
                              * Linear's pre-activation, pre[s].p[(idx ? idx[row] : row) * ld + 0 .. first-layer width).  This is
                              * how a Linear over torch.cat([x_i, x_j, e]) (InteractionNetwork.message, models/magnet_gnn.py:79-83)
                              * runs with K = width(e) only: W [x_i, x_j, e] + b = ((Wa x)[i] + b) + (Wb x)[j] + Wc e (pre[0] carries the bias: with both
                              * sources set the fp16x3 kernel starts its accumulators from the gathered rows), the two
                              * per-node products computed once per node instead of once per edge (col / scale / rdiv unused) */
    float* x0_out;           /* optional (NULL = off) tape outputs for a training backward (no reference counterpart: torch autograd
                              * keeps these tensors alive instead): the first Linear's assembled input rows [M][nkt*32], */
    float* hid[MAGNET_MAX_LAYERS]; /* ... hid[l], l < nl-1: layer l's output after its activation [M][ht*32]; hid[nl-1]: the last
                              * Linear's raw output before last_act / LayerNorm [M][ot*32]; nl == 2 (split-GEMM kernels): hid[2] =
                              * the hidden layer's PRE-activation [M][ht*32] (the MPNN update net's backward needs Swish'(z1)) */
    uint64_t x0_skip;        /* bit kt set: k tile kt of x0_out is NOT written (a backward that walks gathered k tiles back at
                              * source-row granularity never reads them) */
    int32_t* status;         /* fp16x3 / fp16x3s kernels, NULL = off: device status word that receives `status_bit` (atomic OR) when an
                              * activation BETWEEN two Linears of the chain, or the launch's output when no LayerNorm bounds it (it
                              * may feed another fp16 launch: EDSR's convolutions), reaches 65504 - the high fp16 piece saturates there,
                              * and a ReLU can turn the resulting -inf of the next layer into a finite (wrong) output that no check of
                              * the chain's inputs or outputs sees.  The host repeats such a rollout in bf16x6 (ops.RangeGuard). */
    int32_t status_bit;
    int32_t pad2_;
} magnet_mlp_args_t;

/* Fused gather/concat -> NL x (Linear [+ activation]) -> [LayerNorm] -> store | +residual | CSR mean.
 * nl = number of Linears (1, 2 or 5), ht/ot = hidden/output width in 32-column tiles.
 * Replaces the torch.cat + nn.Linear + activation chains of models/mpnn.py:77-78,85-88,191-192 and
 * models/backbones/mlp.py:23-28 (+ nn.LayerNorm, + torch_scatter mean for SEGMEAN) as called from
 * models/magnet_gnn.py:38-42,79-90,135-137,339.  SEGMEAN must be followed by
 * magnet_segmean_fixup_f32 on the same stream. */
int magnet_mlp_f32(const magnet_mlp_args_t* args, int nl, int ht, int ot, int act, int last_act, int ln, int epi,
                   void* stream);
int magnet_mlp_supported(int nl, int ht, int ot, int act, int last_act, int ln, int epi);
/* Same contract with every Linear evaluated on the bf16 matrix cores through the exact hi/mid/lo split (six partial
 * products, fp32-class accuracy).  layer[l].w: the three bf16 planes of the padded weight in
 * fragment order [kt][nt][m][plane][64 lanes][8 bf16] (magnet_amd/packing.py:pack_linear_bf16x3); the planes of one
 * k tile are DMA'd into LDS and shared by the 4 waves of a workgroup. */
int magnet_mlp_bf16x6_f32(const magnet_mlp_args_t* args, int nl, int ht, int ot, int act, int last_act, int ln, int epi,
                          void* stream);
/* Same again with the two-piece fp16 split (x = h + l/2048, three fp16 MFMA products, cross terms in their own
 * accumulator): half the matrix-pipe work of bf16x6, fp32-class accuracy, |values| < 65504 required.
 * layer[l].w: planes [kt][nt][m][2][64 lanes][8 fp16] (magnet_amd/packing.py:pack_linear_fp16x2). */
int magnet_mlp_fp16x3_f32(const magnet_mlp_args_t* args, int nl, int ht, int ot, int act, int last_act, int ln, int epi,
                          void* stream);
/* The same three fp16 products in ONE accumulator per output block (no cross accumulator to zero and fold: fewer VALU instructions
 * and registers per layer): layer[l].w = the fp16x3 planes with the HIGH plane times 2^11, layer[l].b = 2^11 b
 * (packing mode "fp16x3s", |w| < 31.98), the activation's low piece is the unscaled fp16 remainder.  Instantiated for the 5-Linear
 * chains (models/backbones/mlp.py:9-28 with mlp_layers = 4) only: -3 for any other shape, magnet_mlp_fp16x3s_supported says which. */
int magnet_mlp_fp16x3s_f32(const magnet_mlp_args_t* args, int nl, int ht, int ot, int act, int last_act, int ln, int epi,
                           void* stream);
int magnet_mlp_fp16x3s_supported(int nl, int ht, int ot, int act, int last_act, int ln, int epi);
/* REDUCED precision, opt-in (BASELINE.json config 2 "bf16"): inputs and weights rounded to ONE bf16 piece (RNE), one MFMA
 * product, fp32 accumulate; ~1e-2 relative deviation from the fp32 result.  layer[l].w: one bf16 plane
 * [kt][nt][m][1][64 lanes][8 bf16] (packing.py:pack_linear_bf16x1). */
int magnet_mlp_bf16_f32(const magnet_mlp_args_t* args, int nl, int ht, int ot, int act, int last_act, int ln, int epi,
                        void* stream);
int magnet_segmean_fixup_f32(const int32_t* rowptr, int n_rows, int d, const float* partials, float* agg,
                             void* stream);

/* out[r][:] = mean(msg[rowptr[r] : rowptr[r+1]][:]), 0 for empty rows; h in {32, 64, 128, 256}.
 * Replaces torch_scatter scatter(reduce='mean') behind PyG MessagePassing(aggr='mean')
 * (models/mpnn.py:46, models/magnet_gnn.py:54) for CSR-sorted messages. */
int magnet_segment_mean_f32(const float* msg, const int32_t* rowptr, int n_rows, int h, float* out, void* stream);

/* Range guard of the fp16x3 GEMM mode (no reference counterpart: the reference computes in fp32 throughout, configs/
 * trainer/default.yaml:5).  *status |= range_bit if some limit <= |x[i]| < inf, |= nonfinite_bit if some x[i] is inf or NaN
 * (the latter only while *status does not carry unless_bit: a non-finite prediction fed back as an input is not bad data);
 * sticky, never cleared by the library.  The models check the arrays entering the GEMM chains against 65504 (the two-piece
 * fp16 split overflows there) and the predicted window for non-finite values, and raise on the host (magnet_amd/models). */
int magnet_status_check_f32(const float* x, long n, float limit, int range_bit, int nonfinite_bit, int unless_bit, int* status,
                            void* stream);

/* ---- MPNN / MPNN_2d (hidden width 128) ---- */
/* node_in[n] = [u[n,0:tw] | pos[n,1]/L[0..dpos) | pos[n,0]/tmax | 0-pad to kin]; models/mpnn.py:180-191,
 * models/mpnn_2d.py:179 (the x/Lx, x/Ly quirk).  L and tmax are device scalars/vectors. */
int magnet_mpnn_node_input_f32(const float* u, const float* pos, int pos_ld, const float* Lvec, const float* tmax,
                               int n, int tw, int dpos, int kin, float* out, void* stream);
/* agg[i] = mean_{e: tgt(e)=i} Swish(W2 * Swish(A[i] + Bm[src(e)]) + b2), 0 for rows without edges; GNN_Layer.message +
 * aggregate, models/mpnn.py:73-79,46, with message_net_1 split per node.  Edges CSR-sorted by target
 * (rowptr [n_rows + 1]).  The call enqueues the fused edge kernel and the small kernel that assembles the rows crossing
 * work-partition boundaries: agg is complete when the stream reaches the end of the call.  partials: scratch,
 * ceil(n_edges / 32) * 2 * 128 floats. */
int magnet_mpnn_edge_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                         const int32_t* rowptr, int n_edges, int n_rows, const void* w2_packed, const float* b2,
                         float* agg, float* partials, void* stream);
/* Same contract as magnet_mpnn_edge_f32, message_net_2 evaluated on the bf16 matrix cores through the exact
 * hi/mid/lo split (six partial products, fp32-class accuracy).  w2_split: the three bf16 planes of
 * W2 in fragment order [kt][nt][m][plane][64 lanes][8 bf16] (magnet_amd/packing.py:pack_linear_bf16x3). */
int magnet_mpnn_edge_bf16x6_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                const int32_t* rowptr, int n_edges, int n_rows, const void* w2_split, const float* b2,
                                float* agg, float* partials, void* stream);
/* ... through the two-piece fp16 split (three MFMA products; the default, k_mpnn_edge7) */
int magnet_mpnn_edge_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                const int32_t* rowptr, int n_edges, int n_rows, const void* w2_split, const float* b2,
                                float* agg, float* partials, void* stream);
/* reduced-precision variant of the edge kernel (one bf16 piece, see magnet_mlp_bf16_f32) */
int magnet_mpnn_edge_bf16_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                              const int32_t* rowptr, int n_edges, int n_rows, const void* w2_bf16, const float* b2,
                              float* agg, float* partials, void* stream);
/* The same two kernels for callers that fold log2(e) into the packed per-node halves of message_net_1: A_l2e = log2(e) * A,
 * Bm_l2e = log2(e) * Bm.  The hardware exponential is base 2, so both Swish evaluations then cost one multiply less; the
 * factor rides through W2 (linear) and is divided out together with the degree.  agg is the same quantity as above
 * (models/mpnn.py:73-79,46); w2 / b2 are the unscaled parameters. */
int magnet_mpnn_edge_fp16x3_l2e_f32(const float* A_l2e, const float* Bm_l2e, const int32_t* src, const int32_t* tgt,
                                    const int32_t* rowptr, int n_edges, int n_rows, const void* w2_split, const float* b2,
                                    float* agg, float* partials, void* stream);
int magnet_mpnn_edge_bf16_l2e_f32(const float* A_l2e, const float* Bm_l2e, const int32_t* src, const int32_t* tgt,
                                  const int32_t* rowptr, int n_edges, int n_rows, const void* w2_bf16, const float* b2,
                                  float* agg, float* partials, void* stream);
/* The fp16x3 edge op on 16-edge tiles with two waves per SIMD (k_mpnn_edge16).  Same result contract as
 * magnet_mpnn_edge_fp16x3_f32; w2_t16: the two fp16 planes of W2 in v_mfma_f32_16x16x32_f16 fragment order
 * [4 k steps][8 output tiles][2 planes][64 lanes][8 fp16] (magnet_amd/packing.py:pack_edge16_fp16x2);
 * a_bm_log2e_scaled != 0: A and Bm carry the factor log2(e) as for the _l2e entry points.
 * partials: scratch, ceil(n_edges / 16) * 2 * 128 floats (twice the 32-edge kernels' bound). */
int magnet_mpnn_edge_fp16x3_t16_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                    const int32_t* rowptr, int n_edges, int n_rows, const void* w2_t16, const float* b2,
                                    float* agg, float* partials, int a_bm_log2e_scaled, void* stream);
/* Generation 14 (k_mpnn_edge_stream32, csrc/mpnn_edge_stream.hip): the same op (models/mpnn.py:73-79,46 /
 * models/mpnn_2d.py:73-79,46) with 32-edge tiles cut from the edge STREAM of row-aligned chunks - a tile holds the tail of row r and the
 * head of row r + 1, a wave carries the open row's sum from tile to tile - so that no edge is left to a straggler / remainder pass.
 * tiles [n_chunks][tpc] int4 + chunk_ntiles [n_chunks] from magnet_stream_tiles_plan; n_chunks a multiple of 8.  Rows without edges
 * are not written (the plan's flags say whether there are any: zero-fill agg first).
 * w2_t32: fp16 planes [8 k slots][4 output blocks][2 planes][64 lanes][8 fp16], both pre-scaled by 2^11
 * (magnet_amd/packing.py:pack_edge32t_fp16x2; needs |w| < 31.98); b2 the plain bias.
 * flags: MAGNET_STREAM_L2E (bit 0) = A and Bm carry the factor log2(e), as for the _l2e entry points;
 *        MAGNET_STREAM_GATHER3 (bit 1) = three sets of gathered Bm columns in flight instead of two (what magnet_amd passes);
 *        every other bit must be 0 in the product library (-2); the development library (include/magnet_hip_dev.h) defines two more. */
#define MAGNET_STREAM_L2E 1
#define MAGNET_STREAM_GATHER3 2
int magnet_mpnn_edge_stream_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const void* tiles,
                                       const int32_t* chunk_ntiles, int n_chunks, int tpc, int n_rows, const void* w2_t32,
                                       const float* b2, float* agg, int flags, void* stream);
/* The plan of the kernel above: the rows are cut into n_chunks ranges of about equal edge count (chunk c starts at the first row whose
 * edges begin at or after c E / n_chunks); every chunk's edges are cut into tiles of 32 consecutive edges that touch at most two rows
 * (descriptor: x = first edge, y = row of that edge, z = cnt | split << 8 | ends0 << 16 | ends1 << 17, w = deg(row) | deg(row + 1) << 16).
 * flags [2]: [0] bit 0 = some chunk needs more than tpc tiles (its chunk_ntiles is 0: do not launch), bit 1 = some row has no edge,
 * bit 2 = some row has 65 536 edges or more (do not launch); [1] = the number of tiles of all chunks. */
int magnet_stream_tiles_plan(const int32_t* rowptr, int n_rows, int n_chunks, int tpc, void* tiles, int32_t* chunk_ntiles,
                             int32_t* flags, void* stream);
/* out[n][k] = u[n][tw-1] + dt*(k+1) * Conv1d(8->1,k2)([Swish](Conv1d(1->8,k1,stride s1)(h[n])))[k];
 * models/mpnn.py:197-201 with the conv specs of :140-163. */
int magnet_mpnn_readout_f32(const float* h, const float* u, int n, int tw, const float* w1, const float* b1, int k1,
                            int s1, const float* w2, const float* b2, int k2, int use_swish, const float* dt,
                            float* out, void* stream);
/* The same read-out fused with the InstanceNorm apply of the last hop (models/mpnn.py:70 followed by :197-201): h is the raw update
 * output, normalised with mean / var [n_graphs][128] (magnet_instance_norm_stats_f32) while it is loaded; mean == NULL: h as it is.
 * Lane-per-node kernels for the conv specs of models/mpnn.py:140-163 only, n >= 8192, at most 8 graphs; otherwise -2 (normalise with
 * magnet_instance_norm_f32 and call magnet_mpnn_readout_f32). */
int magnet_mpnn_readout_norm_f32(const float* h, const float* mean, const float* var, float eps, int n_graphs, int rows_per_graph,
                                 const float* u, int n, int tw, const float* w1, const float* b1, int k1, int s1, const float* w2,
                                 const float* b2, int k2, int use_swish, const float* dt, float* out, void* stream);
/* One 3 x 3 (tap [9][P]) or k = 3 (tap [3][P]) "same" convolution of EDSR (models/backbones/edsr.py:32-60: conv + ReLU / + skip) on a SMALL channels-last map as a
 * gathered-tap GEMM with the weights resident in LDS: y[p][32 g .. 32 g + 32) = act(sum_t W_t fmap[tap[t][p]] + b) (+ res[p]) for the
 * ncg <= 4 column groups g.  fmap [P + 1][width] (row P all zero: the padding), width in {32, 128}, tap [9][P] from
 * magnet_conv_tap_index, w[g] / b[g] the fp16x3 packs of rows 32 g .. 32 g + 32 of the [Co][9 * width] weight
 * (packing.pack_linear_fp16x2).  -2 for other shapes: use magnet_mlp_fp16x3_f32 (same arithmetic, streamed weights).
 * status (NULL = off): device word that receives status_bit when an output reaches 65504 - the next convolution splits its input
 * into fp16 pieces (see magnet_mlp_args_t.status). */
int magnet_conv_resident_f32(const float* fmap, int width, const int32_t* tap, int ntaps, int P, const void* const* w,
                             const float* const* b, int ncg, int relu, const float* res, float* y, int ldy, int32_t* status,
                             int status_bit, void* stream);
/* PyG InstanceNorm(x, batch) with batch = contiguous blocks of rows_per_graph rows (models/mpnn.py:63,70).
 * workspace: (2*n_graphs*ceil(rows_per_graph/64) + 2*n_graphs) * H floats. */
int magnet_instance_norm_f32(const float* x, int n_graphs, int rows_per_graph, int H, float eps, float* y,
                             float* workspace, void* stream);
/* The same normalisation for graphs of UNEQUAL size (PyG InstanceNorm(x, batch) with an arbitrary sorted batch vector,
 * models/mpnn.py:63,70): graph g owns rows gptr[g] .. gptr[g+1]; mean / var [n_graphs][H] are outputs.  One workgroup per
 * (graph, 32 channels): the generality path, not the fast one. */
int magnet_instance_norm_ragged_f32(const float* x, const int32_t* gptr, int n_graphs, int H, float eps, float* y, float* mean,
                                    float* var, void* stream);
/* The same InstanceNorm in two steps, for callers that fuse the normalisation into the next consumer's loads:
 * statistics only - mean [n_graphs][H] at workspace + n_graphs*ceil(rows_per_graph/64)*H, biased variance right behind it
 * (same workspace size as above). */
int magnet_instance_norm_stats_f32(const float* x, int n_graphs, int rows_per_graph, int H, float* workspace, void* stream);
/* One MPNN hop's per-node halves of message_net_1 (models/mpnn.py:73-79: the Linear over cat(x_i, x_j, u_i - u_j, pos_i - pos_j,
 * variables) splits into a target half A and a source half Bm, evaluated once per NODE) fused with the InstanceNorm apply of the
 * previous hop (models/mpnn.py:70 `x = self.norm(x, batch)`):
 *   hn = mean ? (h - mean[g]) / sqrt(var[g] + eps) : h      (g = row / rows_per_graph)
 *   A  = Wa [hn | node_in] + ba,   Bm = Wb [hn | node_in] + bb,   h_norm = hn (if mean and h_norm)
 * h [n][128], node_in [n][kin], kin in {0, 32, 64} (0: no node_in, the halves are Linears over h alone - MAgNet's InteractionNetwork,
 * models/magnet_gnn.py:76-90 factored per node); wa / wb: fp16x3 planes of the two [128][128 + kin] weight halves in the layout
 * magnet_mlp_fp16x3_f32 takes (magnet_amd/packing.py:pack_linear_fp16x2).  Returns -2 for an unsupported shape (kin, more than
 * 8 graphs, rows_per_graph not a multiple of 32): the caller then uses magnet_instance_norm_f32 + magnet_mlp_fp16x3_f32. */
/* GNN_Layer.update (models/mpnn.py:81-90), residual branch, as ONE persistent launch for large graphs, optionally fused with the
 * InstanceNorm apply of the previous hop (models/mpnn.py:70) on its h input:
 *   hn = mean ? (h - mean[g]) / sqrt(var[g] + eps) : h          (g = row / rows_per_graph)
 *   y  = hn + Swish(W2 Swish(W1 [hn | agg | node_in] + b1) + b2)
 * h, agg, y [n][128] (y must not alias h), node_in [n][kin], kin in {32, 64}, at most 8 graphs (else -2: use
 * magnet_mlp_fp16x3_f32); w1 / w2: fp16x3 planes of update_net_1 ([128][256 + kin]) / update_net_2 ([128][128]) as
 * packing.pack_linear_fp16x2 lays them out.
 * part_sum / part_m2 (both or neither; [n_rows / 128][128], n_rows % 128 == 0): per 128-row tile the column sums of y and the
 * squares centred on the tile's own column means - the InstanceNorm statistics of y (models/mpnn.py:70) without another pass over
 * it; magnet_colstat_final_f32 turns them into mean / variance. */
int magnet_mpnn_node_update_f32(const float* h, const float* agg, const float* node_in, int kin, long n_rows, const float* mean,
                                const float* var, float eps, int n_graphs, int rows_per_graph, const void* w1_f16,
                                const float* b1, const void* w2_f16, const float* b2, float* y, float* part_sum, float* part_m2,
                                void* stream);
/* The same update with the ONE-accumulator packs (packing mode "fp16x3s": planes and biases of both Linears carry 2^11, |w| < 31.98)
 * by k_node_update2: two workgroups per CU (the planes of update_net_2 are streamed through LDS like those of update_net_1 instead of
 * living there, <= 256 registers per wave, the residual rows are read a second time).  kin = 32 only (-2 otherwise: the caller uses
 * magnet_mpnn_node_update_f32).  Same arguments, same result contract. */
int magnet_mpnn_node_update2_f32(const float* h, const float* agg, const float* node_in, int kin, long n_rows, const float* mean,
                                const float* var, float eps, int n_graphs, int rows_per_graph, const void* w1_f16,
                                const float* b1, const void* w2_f16, const float* b2, float* y, float* part_sum, float* part_m2,
                                void* stream);
/* The same update followed by the hop's InstanceNorm (GNN_Layer.forward, models/mpnn.py:69-70: x = self.norm(update, batch)) in ONE
 * launch, for a batch that fits one workgroup (BASELINE.json configs[0]: 2 x 50 nodes - a rollout bound by its number of launches):
 *   y_norm = InstanceNorm(h + Swish(W2 Swish(W1 [h | agg | node_in] + b1) + b2))   over blocks of rows_per_graph rows
 * (biased variance, eps inside the square root; the last block may be short).  Equal to magnet_mpnn_node_update_f32 followed by
 * magnet_instance_norm_f32 to rounding.  -2 unless kin == 32, n_rows <= 128, rows_per_graph >= 1.  y_norm must not alias h. */
int magnet_mpnn_node_update_norm_small_f32(const float* h, const float* agg, const float* node_in, int kin, long n_rows,
                                           int rows_per_graph, float eps, const void* w1_f16, const float* b1, const void* w2_f16,
                                           const float* b2, float* y_norm, void* stream);
/* mean / biased variance [n_graphs][H] from per-chunk partials of chunk_rows rows (rows_per_graph % chunk_rows == 0; the chunks of a
 * graph consecutive): the merge step of PyG InstanceNorm's statistics (pairwise-variance identity, fixed order). */
int magnet_colstat_final_f32(const float* part_sum, const float* part_m2, int n_graphs, int rows_per_graph, int chunk_rows, int H,
                             float* mean, float* var, void* stream);
int magnet_mpnn_node_halves_f32(const float* h, const float* node_in, int kin, long n_rows, const float* mean, const float* var,
                                float eps, int n_graphs, int rows_per_graph, const void* wa_f16, const float* ba,
                                const void* wb_f16, const float* bb, float* A, float* Bm, float* h_norm, void* stream);
/* The same launch on two waves per SIMD (k_node_halves2, round 5): wa_f16s / wb_f16s and ba_s / bb_s are the packs of packing mode "fp16x3s"
 * (one accumulator per output block: planes and biases times 2^11, |w| < 31.98; magnet_pack_linear_f32 mode 4 flags larger weights).
 * Same arguments, same results to the last bits of the fp16x3 split (models/mpnn.py:73-79 / models/mpnn_2d.py:73-79). */
int magnet_mpnn_node_halves2_f32(const float* h, const float* node_in, int kin, long n_rows, const float* mean, const float* var,
                                float eps, int n_graphs, int rows_per_graph, const void* wa_f16s, const float* ba_s,
                                const void* wb_f16s, const float* bb_s, float* A, float* Bm, float* h_norm, void* stream);

/* ---- neighbour search / CSR (torch_cluster radius_graph / knn call sites: models/mpnn.py:245,
 *      models/magnet_gnn.py:247,293) ---- */
/* nbr[q*cap + 0..cnt[q]) = first `cap` nodes (ascending index, same graph) with dist^2 < r2 */
int magnet_radius_search_f32(const float* x, int dim, int n_nodes, const int32_t* graph_ptr, int n_graphs, float r2,
                             int cap, int32_t* nbr, int32_t* cnt, void* stream);
/* identical output, two-level bounding-box pruning (chunk_ptr[g] = first 64-node chunk of graph g; bbox_scratch:
 * (n_chunks + ceil(n_chunks / 64)) * 6 floats - one box per chunk, then one per 64 chunks) */
int magnet_radius_search_bbox_f32(const float* x, int dim, int n_nodes, const int32_t* graph_ptr,
                                  const int32_t* chunk_ptr, int n_graphs, int n_chunks, float r2, int cap,
                                  float* bbox_scratch, int32_t* nbr, int32_t* cnt, void* stream);
int magnet_radius_offsets(const int32_t* nbr, const int32_t* cnt, int n_nodes, int cap, int loop, int32_t* qptr,
                          void* stream);
int magnet_radius_emit_pairs(const int32_t* nbr, const int32_t* cnt, const int32_t* qptr, int n_nodes, int cap,
                             int loop, int32_t* found, int32_t* query, void* stream);
int magnet_csr_by_target(const int32_t* src, const int32_t* tgt, int n_edges, int n_nodes, int32_t* rowptr,
                         int32_t* cursor, int32_t* eid_scratch, int32_t* perm, int32_t* src_out, int32_t* tgt_out,
                         void* stream);
int magnet_i64_to_i32(const int64_t* a, int n, int32_t* b, void* stream);
int magnet_pairs_to_edge_index(const int32_t* row0, const int32_t* row1, int n_edges, int64_t* edge_index,
                               void* stream);
int magnet_knn_f32(const float* x, const int32_t* x_ptr, const float* y, const int32_t* y_ptr, int n_graphs, int dim,
                   int n_queries, int k, int32_t* out, void* stream);

/* magnet_inr_grid_prepare_f32 for d = 2 in the row order the fused blend epilogue needs (MAGNET_EPI_QUADBLEND): the four shifts of a
 * (query, step) are consecutive rows, row = ((b Nq + q) T + i) 4 + s; blend_w[row] = area[opposite shift] / (sum of the four areas)
 * (models/magnet_cnn_2d.py:248-290).  cell_idx / blend_w [B Nq T 4], tail [B Nq T 4][32]. */
int magnet_inr_grid_prepare_blend_f32(const float* coords, const float* cell, const float* x_t, const float* t, int t_ld, int n_graphs,
                                      int Nq, int T, int W, int32_t* cell_idx, float* tail, float* blend_w, void* stream);
/* ---- MAgNet feature builders / INR decoders (magnet_kernels.hip) ---- */
int magnet_node_features_f32(const float* u, const float* x, const float* t_last, int n_graphs, int nodes_per_graph,
                             int T, int d, int ld, float* out, void* stream);
int magnet_euler_update_f32(const float* ret, int ld_ret, const float* last, const float* t, int t_ld, int n_graphs,
                            int nodes_per_graph, int T, int T_out, float* out, void* stream);
int magnet_inr_knn_prepare_f32(const float* lr_coords, const float* hr_coords, const int32_t* nn, int k, int d,
                               int n_graphs, int L, int Nq, int32_t* nn2, float* rel, float* w, void* stream);
int magnet_inr_knn_blend_f32(const float* base, const float* w, const float* vals, const int32_t* nn2,
                             const float* t, int t_ld, const float* wv, const float* wt, int n_graphs, int L, int Nq,
                             int T, int C, int mode, float* z, void* stream);
int magnet_inr_grid_prepare_f32(const float* coords, const float* cell, const float* x_t, const float* t, int t_ld,
                                int n_graphs, int Nq, int T, int W, int d, int32_t* cell_idx, float* tail,
                                float* area, void* stream);
int magnet_inr_grid_blend_f32(const float* p, const float* area, int n_rows, int T, int C, int nshift, float* z,
                              void* stream);

/* y[n_images][Ho][Wo] = F.interpolate(x[n_images][Hi][Wi], mode='bilinear' (d = 2) | 'linear' (d = 1, Hi = Ho = 1),
 * align_corners=False): the rollout feedback of MAgNet[CNN], models/magnet_cnn.py:456-461, magnet_cnn_2d.py:466-472. */
int magnet_resize_linear_f32(const float* x, int n_images, int Hi, int Wi, int Ho, int Wo, int d, float* y, void* stream);

/* EDSR convolutions (models/backbones/edsr.py:32-60) run as gathered-tap GEMMs through magnet_mlp_*: a "same" k x k
 * (d = 2) or k (d = 1, H = 1) convolution over a channels-last map [P = n_images*H*W rows (+1 all-zero row), C] is one
 * Linear over the k^d shifted neighbours.  tap[t*P + p] = row of pixel p's neighbour for tap t (row-major dy, dx), or
 * P (the zero row) outside the image. */
int magnet_conv_tap_index(int n_images, int H, int W, int k, int d, int32_t* tap, void* stream);

/* ---- backward pass of the MPNN training step (train_kernels.hip).  Replaces what torch autograd does for the
 *      reference's training_step (models/mpnn.py:254-297 -> forward :175-203, GNN_Layer :65-90); the Linear
 *      backward-data products reuse magnet_mlp_* with transposed packed weights. ---- */
int magnet_swish_fwd_f32(const float* z, float* r, long n, void* stream);                       /* mpnn.py:15-24 */
int magnet_swish_bwd_f32(const float* dr, const float* z, float* dz, long n, void* stream);
/* the same, and amax256[b & 255] = max(amax256[b & 255], max |dz| of workgroup b): 256 floats ZEROED by the caller - the amax_in of
 * magnet_linear_bwd128_f32 for this dz (n_amax_in = 256) without a pass of its own */
int magnet_swish_bwd_amax_f32(const float* dr, const float* z, float* dz, long n, float* amax256, void* stream);
int magnet_axpy_f32(const float* a, const float* b, float alpha, float* out, long n, void* stream);
int magnet_mul_f32(const float* a, const float* b, float* out, long n, void* stream);
int magnet_copy_cols_f32(const float* src, int ld_src, int scol, float* dst, int ld_dst, int dcol, int w, long rows,
                         int accumulate, void* stream);
/* dst[r, dcol:dcol+w] (+)= scale * src[r, scol:scol+w] with 16-byte accesses (w, the column offsets and the row strides multiples of 4,
 * 16-byte aligned bases; -2 otherwise) */
int magnet_copy_cols_scaled_f32(const float* src, int ld_src, int scol, float* dst, int ld_dst, int dcol, int w, long rows,
                                float scale, int accumulate, void* stream);
/* p[e] = A[tgt[e]] + Bm[src[e]]: pre-activation of message_net_1 (mpnn.py:75-76), recomputed for the backward */
int magnet_edge_pre_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt, long n_edges, float* p,
                        void* stream);
/* dm[e] = dagg[tgt[e]] / deg(tgt[e]): backward of aggr='mean' (mpnn.py:46) */
int magnet_edge_mean_bwd_f32(const float* dagg, const int32_t* tgt, const int32_t* rowptr, long n_edges, float* dm,
                             void* stream);
/* out[r] = sum_{k in row r} x[perm ? perm[k] : k], 128 columns (scatter-add of edge gradients to nodes) */
int magnet_segment_sum128_f32(const float* x, const int32_t* rowptr, const int32_t* perm, int n_rows, float* out,
                              void* stream);
/* Two column blocks of one Linear's weight gradient in one pass over dY (torch autograd of a Linear over torch.cat([h, node_in]),
 * models/mpnn.py:73-90): dW[0:128, wcol : wcol+128] (+)= dY^T X[:, xcol : xcol+128] and dW[0:128, wcol2 : wcol2+32] (+)= dY^T X2[:, x2col :
 * x2col+32]; db (+)= colsum(dY) (db may be NULL).  workspace: nblk * (128 * 160 + 128) + 128 * 160 floats, nblk = min(768, ceil(M / 32)). */
int magnet_wgrad_pair_f32(const float* dY, int ldy, const float* X, int ldx, int xcol, const float* X2, int ldx2, int x2col, long M,
                          float* dW, int ldw, int wcol, int wcol2, float* db, float* workspace, int accumulate, void* stream);
/* dW[0:128, wcol:wcol+kc] (+)= dY[M,128]^T X[M, xcol:xcol+kc]; db[128] (+)= column sums of dY (db may be null);
 * kc in {32,64,96,128}; workspace: nblk*(128*kc+128) + 128*kc floats, nblk = min(768, ceil(M/32)) */
int magnet_wgrad_f32(const float* dY, int ldy, const float* X, int ldx, int xcol, int kc, long M, float* dW, int ldw,
                     int wcol, float* db, float* workspace, int accumulate, void* stream);
/* general form: dY has n_out in {32, 64, 128} columns, dW is [n_out][ldw] (same workspace formula) */
int magnet_wgrad_n_f32(const float* dY, int ldy, int n_out, const float* X, int ldx, int xcol, int kc, long M, float* dW,
                       int ldw, int wcol, float* db, float* workspace, int accumulate, void* stream);

/* magnet_wgrad_n_f32 for `ntile` neighbouring column tiles of kc columns each - X[:, xcol + t kc ..) -> dW[:, wcol + t kc ..) - in one
 * launch per kernel (the taps of a convolution, the column blocks of a wide first layer).  workspace: ntile * nblk * n_out * kc +
 * nblk * n_out + ntile * n_out * kc floats, nblk = min(768, ceil(M / 32)). */
int magnet_wgrad_tiles_f32(const float* dY, int ldy, int n_out, const float* X, int ldx, int xcol, int kc, int ntile, long M,
                           float* dW, int ldw, int wcol, float* db, float* workspace, int accumulate, void* stream);
int magnet_reduce_partials_f32(const float* partial, int nblk, long n, float scale, float* out, int accumulate,
                               void* stream);

/* ---- backward of the generic fused MLP op and of the MAgNet[GNN]-specific kernels (magnet_train_kernels.hip): what torch
 *      autograd does for models/magnet_gnn.py:312-376 and models/backbones/mlp.py:23-28 + nn.LayerNorm in the reference. ---- */
/* x0[M][nkt*32] = the first-layer input magnet_mlp_* streams from its k-tile descriptors (materialised for the backward) */
int magnet_mlp_materialize_input_f32(const magnet_mlp_args_t* args, float* x0, int ldx, void* stream);
/* Host only, no launch: fills args->nkt and args->kt[] with the descriptors of `ntaps` gathered taps of one source block (tap t, block
 * k: src[idx[t * n_rows + row]][32 k .. 32 k + 32), idx from magnet_conv_tap_index) - a convolution of models/backbones/edsr.py:32-60 as
 * a gathered-tap Linear.  Nothing else in *args is touched.  -2: width not a multiple of 32, more than MAGNET_MAX_KT tiles. */
int magnet_mlp_args_set_taps(magnet_mlp_args_t* args, const float* src, int ld, int width, const int32_t* idx, long n_rows, int ntaps);
int magnet_relu_bwd_f32(const float* dy, const float* y, float* dz, long n, void* stream);
/* LayerNorm backward over rows of D <= 256 columns from the pre-norm z; part: [min(2048, ceil(rows/4))][2][D] partial
 * (dgamma | dbeta) sums for magnet_reduce_partials_f32 */
int magnet_layernorm_bwd_f32(const float* z, int ldz, const float* dy, int ldd, const float* gamma, long rows, int D, float eps,
                             float* dz, int ldo, float* part, void* stream);
/* the same, plus amax [min(2048, ceil(rows/4))]: per-workgroup maxima of |dz| - the amax_in of magnet_linear_bwd128_f32 / 64 (NULL:
 * not wanted); tgt / rowptr != NULL: the rows are CSR-ordered edges and dy [n_nodes][ldd] is the gradient of their per-target mean - row
 * r takes dy[tgt[r]] / deg(tgt[r]) (magnet_edge_mean_bwd_cols_f32 fused into the load) */
int magnet_layernorm_bwd_amax_f32(const float* z, int ldz, const float* dy, int ldd, const float* gamma, long rows, int D, float eps,
                                  float* dz, int ldo, float* part, float* amax, const int32_t* tgt, const int32_t* rowptr,
                                  void* stream);
/* dm[e][0:D] = dagg[tgt[e]][0:D] / deg(tgt[e]) */
int magnet_edge_mean_bwd_cols_f32(const float* dagg, const int32_t* tgt, const int32_t* rowptr, long n_edges, int D, float* dm,
                                  int ldm, void* stream);
/* out[r][ocol:ocol+width] (+)= scale * sum_{k in CSR row r} x[perm ? perm[k] : k][xcol:xcol+width]; width in {4..256}, a
 * power of two times 4: the scatter-add that undoes a row gather */
int magnet_segment_sum_cols_f32(const float* x, int ldx, int xcol, const int32_t* rowptr, const int32_t* perm, int n_rows,
                                int width, float scale, float* out, int ldo, int ocol, int accumulate, void* stream);
/* the same for n_runs <= 9 gathers of the same source rows in one launch (host arrays of n_runs device pointers / column offsets: the
 * taps of a convolution): out[r] (+)= scale * sum_u sum_{k in row r of run u} x[perms[u] ? perms[u][k] : k][xcols[u] : +width], runs added
 * in index order */
int magnet_segment_sum_cols_runs_f32(const float* x, int ldx, const int32_t* const* rowptrs, const int32_t* const* perms,
                                     const int32_t* xcols, int n_runs, int n_rows, int width, float scale, float* out, int ldo, int ocol,
                                     int accumulate, void* stream);
/* backward of magnet_inr_knn_blend_f32: dbase [2*nq][C], dvq [2*nq][T] (per (query, neighbour), to be scatter-added to the
 * LR nodes), part: [min(1024, ceil(nq/4))][2][C] partial (dwv | dwt) sums */
int magnet_inr_knn_blend_bwd_f32(const float* dz, const float* w, const float* vals, const int32_t* nn2, const float* t,
                                 int t_ld, const float* wv, int n_graphs, int L, int Nq, int T, int C, int mode, float* dbase,
                                 float* dvq, float* part, void* stream);
/* backward of magnet_inr_grid_blend_f32: dp [n_rows*nshift*T][C] from dz [n_rows*T][C] (the area weights are constants) */
int magnet_inr_grid_blend_bwd_f32(const float* dz, const float* area, int n_rows, int T, int C, int nshift, float* dp,
                                  void* stream);
int magnet_euler_update_bwd_f32(const float* dout, const float* t, int t_ld, int n_graphs, int nodes_per_graph, int T,
                                int T_out, float* dret, int ld_ret, float* dlast, void* stream);
/* backward of InstanceNorm (mpnn.py:63,70) from the kept output yhat and biased variance var [n_graphs][H];
 * workspace: (2*n_graphs*ceil(rows/128) + 2*n_graphs)*H floats */
int magnet_instance_norm_bwd_f32(const float* dy, const float* yhat, const float* var, int n_graphs, int rows_per_graph,
                                 int H, float eps, float* dx, float* workspace, void* stream);
/* ... with the Swish backward of the layer whose output was normalised in the same pass (mpnn.py:81-90, :63): dx as above, dz = dx * Swish'(z);
 * amax256: NULL or 256 zeroed floats that receive the block maxima of |dz| (as magnet_swish_bwd_amax_f32).  dx has the bits of magnet_instance_norm_bwd_f32's, dz is within a few ulp of the two calls'. */
int magnet_instance_norm_bwd_swish_f32(const float* dy, const float* yhat, const float* var, int n_graphs, int rows_per_graph, int H,
                                       float eps, const float* z, float* dx, float* dz, float* amax256, float* workspace, void* stream);
/* ... for graphs of unequal size: graph g owns rows gptr[g] .. gptr[g + 1] (PyG's InstanceNorm(x, batch), models/mpnn.py:63,70); yhat and var
 * [n_graphs][H] as magnet_instance_norm_ragged_f32 left them.  No workspace. */
int magnet_instance_norm_ragged_bwd_f32(const float* dy, const float* yhat, const float* var, const int32_t* gptr, int n_graphs, int H,
                                        float eps, float* dx, void* stream);
/* backward of the Conv1d read-out + Euler update (mpnn.py:197-201): dh [n,128], du_last [n] (= d u[:, tw-1]),
 * gpart [min(1024, ceil(n/4))][8*k1+8+8*k2+1] partial gradients laid out [dw1 | db1 | dw2 | db2] */
int magnet_mpnn_readout_bwd_f32(const float* h, const float* dout, int n, int tw, const float* w1, const float* b1,
                                int k1, int s1, const float* w2, int k2, int use_swish, const float* dt, float* dh,
                                float* du_last, float* gpart, void* stream);

/* One Adam step on a flat fp32 parameter buffer (torch.optim.Adam semantics: L2 weight decay folded into the gradient,
 * bias-corrected moments, eps after the square root); `step` counts from 1.  Replaces the optimizer of
 * configure_optimizers (models/mpnn.py:206-214); the StepLR schedule stays host-side scalar logic. */
int magnet_adam_f32(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, void* stream);

/* Fused backward of the edge block (adjoint of magnet_mpnn_edge_*: mpnn.py:73-79,46 under autograd).  Recomputes
 * p = A[i]+Bm[j], s1 = Swish(p), q = W2 s1 + b2; dq = dagg[i]/deg(i) * Swish'(q); dp = (W2^T dq) * Swish'(p);
 * dA[i] = sum_e dp (all n_nodes rows written).  s1, dq, dp [n_edges][128] are written for the follow-up kernels
 * (dW2 = dq^T s1 via magnet_wgrad_f32; dBm = by-source segment sum of dp).  w2_f16 / w2t_f16: fp16x2 planes of W2
 * and W2^T (packing.py:pack_linear_fp16x2); partials: ceil(n_edges/32)*2*128 floats. */
int magnet_mpnn_edge_bwd_f32(const float* A, const float* Bm, const float* dagg, const int32_t* src, const int32_t* tgt,
                             const int32_t* rowptr, int n_nodes, int n_edges, const void* w2_f16, const void* w2t_f16,
                             const float* b2, float* s1, float* dq, float* dp, float* dA, float* partials, void* stream);

/* One hidden 128 -> 128 Linear of an MLP chain walked back in one kernel (torch autograd through models/backbones/mlp.py:9-28):
 * dx [M][lddx] = (mask ? x > 0 : 1) * (dz W), dW [128][128] (+)= dz^T x, db [128] (+)= column sums of dz.  wt_f16: fp16x3 planes of W^T
 * packed as a Linear (|w| < 31.98: the kernel keeps 2^11 times the high plane).  amax_in: n_amax_in block maxima of |dz| - the amax_out
 * of the call that produced dz - or NULL (one more pass over dz; needs ldz == 128); amax_out: NULL or magnet_linear_bwd128_grid(M)
 * floats.  workspace: magnet_linear_bwd128_workspace() floats. */
int magnet_linear_bwd128_workspace(void);
/* Which kernel serves magnet_linear_bwd128_f32 / magnet_linear_bwd64_f32: 1 (default) = k_linear_bwd_r (32-row steps, two waves per SIMD,
 * W^T in registers, dz / x split once into fp16 planes in LDS; the 128-wide one with four waves x two feature blocks and two workgroups
 * per CU from 65 536 rows, eight waves x one block below), 0 = k_linear_bwd<NT> (128-row steps, one wave per SIMD; kept for same-process
 * A/B runs), 2 / 3 = the two 128-wide shapes of k_linear_bwd_r at every size.  Returns the previous value; any other argument only
 * queries.  magnet_linear_bwd128_grid / 64_grid follow the choice: do not change it between a call that writes amax_out and the call
 * that reads it as amax_in. */
int magnet_linear_bwd128_generation(int generation);
int magnet_linear_bwd128_grid(long M);
int magnet_linear_bwd128_f32(const float* dz, int ldz, const float* x, int ldx, const void* wt_f16, long M, int mask, float* dx,
                             int lddx, float* dW, float* db, int accumulate, const float* amax_in, int n_amax_in, float* amax_out,
                             float* workspace, void* stream);
/* TWO consecutive hidden 128 -> 128 Linears of a chain walked back in one kernel (models/backbones/mlp.py:9-28, Linear -> ReLU -> Linear ->
 * ReLU): layer A (gradient dz at its output, its input x1, wtA = planes of W_A^T) and the layer below it, B (its input x2, wtB):
 *   dx [M][lddx] = (x2 > 0) * (((x1 > 0) * (dz W_A)) W_B),  dW_A (+)= dz^T x1,  db_A (+)= colsum(dz),  dW_B (+)= dzB^T x2,  db_B (+)= colsum(dzB)
 * - the gradient between the two layers never leaves the chip (four [M, 128] array passes instead of six).  Arguments as
 * magnet_linear_bwd128_f32 (same workspace function); amax_out: NULL or room for magnet_linear_bwd128_pair_grid(M) floats. */
int magnet_linear_bwd128_pair_grid(long M);
int magnet_linear_bwd64_pair_grid(long M);      /* ... and the 64 -> 64 twin (dW [64][64], db [64]; the hidden width of MAgNet[CNN]'s MLPs) */
int magnet_linear_bwd64_pair_f32(const float* dz, int ldz, const float* x1, int ldx1, const float* x2, int ldx2, const void* wtA_f16,
                                 const void* wtB_f16, long M, float* dx, int lddx, float* dWA, float* dbA, float* dWB, float* dbB,
                                 int accumulate, const float* amax_in, int n_amax_in, float* amax_out, float* workspace, void* stream);
int magnet_linear_bwd128_pair_f32(const float* dz, int ldz, const float* x1, int ldx1, const float* x2, int ldx2, const void* wtA_f16,
                                  const void* wtB_f16, long M, float* dx, int lddx, float* dWA, float* dbA, float* dWB, float* dbB,
                                  int accumulate, const float* amax_in, int n_amax_in, float* amax_out, float* workspace, void* stream);
/* the same for a 64 -> 64 Linear (dW [64][64], db [64]; the hidden width of MAgNet[CNN]'s MLPs); same workspace function, amax_out:
 * magnet_linear_bwd64_grid(M) floats */
int magnet_linear_bwd64_grid(long M);
int magnet_linear_bwd64_f32(const float* dz, int ldz, const float* x, int ldx, const void* wt_f16, long M, int mask, float* dx,
                            int lddx, float* dW, float* db, int accumulate, const float* amax_in, int n_amax_in, float* amax_out,
                            float* workspace, void* stream);

/* The same backward with the weight gradient of message_net_2 formed inside the kernel (what torch autograd returns for
 * message_net_2.weight / .bias, models/mpnn.py:32-35): dW2 [128][128] (+)= dq^T s1, db2 [128] (+)= column sums of dq - s1 and dq are
 * never written to memory; dp [n_edges][128] and dA [n_nodes][128] as above.  workspace: magnet_mpnn_edge_bwd_w_workspace() floats
 * (per-workgroup partial sums, added in workgroup order: deterministic). */
int magnet_mpnn_edge_bwd_w_workspace(void);
/* Which kernel serves the entry points below: 2 (default) = k_mpnn_edge_bwd_co (csrc/mpnn_edge_bwd_co.hip: the workgroup's eight waves
 * co-operate on one 32-edge tile, a wave owns 16 features of all three GEMMs; graphs of 8 M nodes and more take generation 1),
 * 1 = k_mpnn_edge_bwd_wp (csrc/mpnn_edge_bwd_wp.hip: every wave forms the weight gradient of its own tiles in the accumulator half of
 * the register file), 0 = k_mpnn_edge_bwd_w (the shared-slab kernel) - both kept for same-process A/B runs.  Returns the previous
 * value; any other argument only queries.  The results differ in the order of the fp32 additions only. */
int magnet_mpnn_edge_bwd_w_generation(int generation);
int magnet_mpnn_edge_bwd_w_f32(const float* A, const float* Bm, const float* dagg, const int32_t* src, const int32_t* tgt,
                               const int32_t* rowptr, int n_nodes, int n_edges, const void* w2_f16, const void* w2t_f16,
                               const float* b2, float* dp, float* dA, float* partials, float* dW2, float* db2, int accumulate,
                               float* workspace, void* stream);
/* the same with the block maxima of |dagg| handed in (amax_in [n_amax_in]: the amax_out of the magnet_linear_bwd128_f32 call that
 * produced dagg; NULL: computed here with a pass over dagg) */
int magnet_mpnn_edge_bwd_w_amax_f32(const float* A, const float* Bm, const float* dagg, const int32_t* src, const int32_t* tgt,
                                    const int32_t* rowptr, int n_nodes, int n_edges, const void* w2_f16, const void* w2t_f16,
                                    const float* b2, float* dp, float* dA, float* partials, float* dW2, float* db2, int accumulate,
                                    float* workspace, const float* amax_in, int n_amax_in, void* stream);
/* the same with `flags`: MAGNET_EDGE_BWD_AB_LOG2E = A / Bm are the training forward's log2(e)-scaled halves (what
 * magnet_mpnn_edge_stream_fp16x3_f32 takes with MAGNET_STREAM_L2E: the backward reads the forward's own arrays); dp / dA / dW2 / db2
 * are the gradients of the unscaled block (p = ln 2 (A' + Bm')) all the same.  Generation 2 only: -2 where
 * magnet_mpnn_edge_bwd_w_takes_flags(n_nodes) is 0; unknown bits: -2. */
#define MAGNET_EDGE_BWD_AB_LOG2E 1
int magnet_mpnn_edge_bwd_w_takes_flags(int n_nodes);
int magnet_mpnn_edge_bwd_w_flags_f32(const float* A, const float* Bm, const float* dagg, const int32_t* src, const int32_t* tgt,
                                     const int32_t* rowptr, int n_nodes, int n_edges, const void* w2_f16, const void* w2t_f16,
                                     const float* b2, float* dp, float* dA, float* partials, float* dW2, float* db2, int accumulate,
                                     float* workspace, const float* amax_in, int n_amax_in, int flags, void* stream);

/* Weight packing on the device (magnet_amd/packing.py: the same layouts as torch index shuffles; no reference counterpart - the
 * reference hands its nn.Linear weights to cuBLAS as they are).  Padded position (r, c) of the [np x kp] matrix takes
 * w[r' stride_r + c' stride_c] with r' = row_src ? row_src[r] : r (r < n) and c' = col_src ? col_src[c] : c (c < k); -1 / out of
 * range = 0.  mode 0: fp32 fragments [kt][nt][4][64][4]; 1: fp16 planes (h, 2^11 (w - h)) [kt][nt][2][2][64][8]; 2: three
 * truncated bf16 pieces [kt][nt][2][3][64][8]; 3: one RNE bf16 plane [kt][nt][2][1][64][8].  bias_out [np] (may be null) = bias
 * padded with zeros.  mode 1 sets range_bit in *status (device int, may be null) when some |w| >= 65504 or is not finite. */
int magnet_pack_linear_f32(const float* w, long stride_r, long stride_c, int n, int k, const int32_t* row_src,
                           const int32_t* col_src, const float* bias, int n_bias, int np, int kp, int mode, void* out,
                           float* bias_out, int* status, int range_bit, void* stream);

/* Measurement utility for bench.py (no reference counterpart): one wave on `stream` samples the shader clock for ~wall_us
 * microseconds while other streams run the workload.  out[0] = shader-clock ticks, out[1] = constant-rate ticks of the
 * wall clock whose rate (kHz) is returned through *wall_khz (host pointer): MHz = out[0] / out[1] * wall_khz / 1000. */
int magnet_clock_probe(long long* out, int wall_us, int* wall_khz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGNET_HIP_H */
