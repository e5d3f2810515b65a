// Generic register-chained MLP kernel (see common.cuh) and its C entry point.
//
//   rows M (nodes / edges / INR samples), one 32-row tile per wave, 4 waves per workgroup.
//   layer 0   : K streamed in 32-wide tiles straight from (gathered, concatenated, differenced)
//               global sources -> HT*32 hidden (or OT*32 outputs when NL == 1)
//   layers 1..NL-2 : HT*32 -> HT*32 in registers
//   layer NL-1: HT*32 -> OT*32
//   epilogue  : optional activation, optional LayerNorm, then store / store+residual /
//               CSR segment mean.
// Replaces the reference's per-edge / per-node torch.cat + nn.Linear + activation chains:
// models/mpnn.py:77-78,85-88,191-192; models/backbones/mlp.py:23-28 as used by
// models/magnet_gnn.py:22-36,55-68,129-132,193-197.
#include "common.cuh"
#include "edge_tile.cuh"

// layer 0: K streamed tile by tile from the gathered sources (hipcc schedules the loads; a hand
// pipelined version with sched_barriers measured 20-30 % slower, see DESIGN.md "what did not work")
template <int NT>
__device__ __forceinline__ void stream_layer(const MagnetMlpArgs& a, const MagnetLayer l0, f32x16 (&out)[NT], int row, int lane, bool row_ok = false) {
    const int half = lane >> 5;
    init_bias<NT>(out, l0.b, half);
    for (int kt = 0; kt < a.nkt; ++kt) {
        const f32x16 x = load_ktile(a.kt[kt], row, half);
        mma_ktile<NT>(x, out, l0.w, kt, lane);
        if (a.x0_out && row_ok && !((a.x0_skip >> kt) & 1)) store_ktile(x, a.x0_out, a.nkt * 32, kt * 32, row, true, half);
    }
}

template <int NL, int HT, int OT, int ACT, bool LAST_ACT, bool LN, int EPI>
__global__ void __launch_bounds__(256) k_mlp(const MagnetMlpArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const int tile = blockIdx.x * 4 + wave;
    const int row_raw = tile * 32 + (lane & 31);
    const bool row_ok = row_raw < a.M;
    const int row = row_ok ? row_raw : (a.M - 1);
    TileMeta md;                                        // CSR-mean epilogue: requested now, hidden behind the GEMMs
    if (EPI == EPI_SEGMEAN) md = tile_meta(tile, a.M, a.tgt, a.rowptr, lane & 31);

    const int cg = (NL == 1) ? blockIdx.y : 0;          // column group (single-Linear launches may split their outputs)
    f32x16 out[OT];
    if (NL == 1) {
        stream_layer<OT>(a, a.layer[cg], out, row, lane);
    } else {
        f32x16 h0[HT], h1[HT];
        stream_layer<HT>(a, a.layer[0], h0, row, lane, row_ok);
        if (a.pre[0].p) add_rows<HT>(h0, a.pre[0], row, half);       // per-node products of a factored first layer (magnet_hip.h: pre)
        if (a.pre[1].p) add_rows<HT>(h0, a.pre[1], row, half);
#pragma unroll
        for (int t = 0; t < HT; ++t) apply_act<ACT>(h0[t]);
        if (a.hid[0]) store_rows<HT, 0>(h0, a.hid[0], HT * 32, HT * 32, nullptr, 0, row, row_ok, half);
        // middle layers, ping-pong h0 <-> h1 (fully unrolled so register arrays stay static)
#pragma unroll
        for (int l = 1; l < NL - 1; ++l) {
            if (l & 1) {
                dense_layer<HT, HT>(h0, h1, a.layer[l], lane);
#pragma unroll
                for (int t = 0; t < HT; ++t) apply_act<ACT>(h1[t]);
                if (a.hid[l]) store_rows<HT, 0>(h1, a.hid[l], HT * 32, HT * 32, nullptr, 0, row, row_ok, half);
            } else {
                dense_layer<HT, HT>(h1, h0, a.layer[l], lane);
#pragma unroll
                for (int t = 0; t < HT; ++t) apply_act<ACT>(h0[t]);
                if (a.hid[l]) store_rows<HT, 0>(h0, a.hid[l], HT * 32, HT * 32, nullptr, 0, row, row_ok, half);
            }
        }
        if ((NL - 1) & 1) dense_layer<HT, OT>(h0, out, a.layer[NL - 1], lane);
        else dense_layer<HT, OT>(h1, out, a.layer[NL - 1], lane);
        if (a.hid[NL - 1]) store_rows<OT, 0>(out, a.hid[NL - 1], OT * 32, OT * 32, nullptr, 0, row, row_ok, half);
    }
    if (LAST_ACT) {
#pragma unroll
        for (int t = 0; t < OT; ++t) apply_act<ACT>(out[t]);
    }
    if (LN) layer_norm_rows<OT>(out, a.ln_g, a.ln_b, a.ln_eps, half);

    if (EPI == EPI_SEGMEAN) {
        segmean_epilogue_dpp<OT>(out, md, a.M, a.y, a.partials, lane);
    } else {
        float* yg = a.y + (a.y_group_stride ? (size_t)cg * a.y_group_stride : (size_t)cg * OT * 32);   // groups_in_block: always gridDim.y here
        store_rows<OT, EPI == EPI_STORE_MASK ? 2 : (EPI == EPI_STORE_RES ? 1 : 0)>(out, yg, a.ldy, a.ncols, a.res + cg * OT * 32, a.ldres, row, row_ok, half);
    }
}

// Combine the tile-boundary partial sums of segmean_epilogue and zero the empty rows.
__global__ void __launch_bounds__(256) k_segmean_fixup(const int32_t* __restrict__ rowptr, int N, int D,
                                                       const float* __restrict__ partials, float* __restrict__ agg) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= N) return;
    const int s = rowptr[r], e = rowptr[r + 1];
    if (e == s) {
        for (int f = lane; f < D; f += 64) agg[(size_t)r * D + f] = 0.f;
        return;
    }
    const int ta = s >> 5, tb = (e - 1) >> 5;
    if (ta == tb) return;
    for (int f = lane; f < D; f += 64) {
        float acc = partials[((size_t)ta * 2 + 1) * D + f];
        for (int t = ta + 1; t <= tb; ++t) acc += partials[((size_t)t * 2 + 0) * D + f];
        agg[(size_t)r * D + f] = acc / (float)(e - s);
    }
}

// ------------------------------------------------------------------------------------ dispatch
typedef void (*mlp_fn)(const MagnetMlpArgs);

#define MLP_CASE(NL_, HT_, OT_, ACT_, LA_, LN_, EPI_)                                              \
    if (nl == NL_ && ht == HT_ && ot == OT_ && act == ACT_ && last_act == LA_ && ln == LN_ && epi == EPI_) \
        fn = k_mlp<NL_, HT_, OT_, ACT_, LA_, LN_, EPI_>;

static mlp_fn pick(int nl, int ht, int ot, int act, int last_act, int ln, int epi) {
    mlp_fn fn = nullptr;
    // MPNN: embedding / node update (Swish after every Linear), factorised first edge layer
    MLP_CASE(2, 4, 4, ACT_SWISH, true, false, EPI_STORE)
    MLP_CASE(2, 4, 4, ACT_SWISH, true, false, EPI_STORE_RES)
    MLP_CASE(1, 4, 4, ACT_NONE, false, false, EPI_STORE)
    // EDSR convolutions as gathered-tap GEMMs (conv + ReLU, conv + skip)
    MLP_CASE(1, 4, 4, ACT_RELU, true, false, EPI_STORE)
    MLP_CASE(1, 4, 4, ACT_NONE, false, false, EPI_STORE_RES)
    MLP_CASE(1, 2, 2, ACT_RELU, true, false, EPI_STORE)
    MLP_CASE(1, 2, 2, ACT_NONE, false, false, EPI_STORE_RES)
    MLP_CASE(1, 2, 2, ACT_NONE, false, false, EPI_STORE)
    MLP_CASE(1, 1, 1, ACT_NONE, false, false, EPI_STORE)
    MLP_CASE(1, 1, 1, ACT_RELU, true, false, EPI_STORE)
    MLP_CASE(1, 1, 1, ACT_NONE, false, false, EPI_STORE_RES)
    // backward-data GEMMs with the ReLU mask of the layer below as the epilogue (magnet_amd/autograd.py)
    MLP_CASE(1, 4, 4, ACT_NONE, false, false, EPI_STORE_MASK)
    MLP_CASE(1, 2, 2, ACT_NONE, false, false, EPI_STORE_MASK)
    MLP_CASE(1, 1, 1, ACT_NONE, false, false, EPI_STORE_MASK)
    // MAgNet 5-Linear ReLU MLPs: hidden 128 (HT 4) or 64 (HT 2)
    MLP_CASE(5, 4, 4, ACT_RELU, false, true, EPI_STORE)
    MLP_CASE(5, 4, 4, ACT_RELU, false, true, EPI_STORE_RES)
    MLP_CASE(5, 4, 4, ACT_RELU, false, true, EPI_SEGMEAN)
    MLP_CASE(5, 4, 1, ACT_RELU, false, false, EPI_STORE)
    MLP_CASE(5, 2, 1, ACT_RELU, false, true, EPI_STORE)
    MLP_CASE(5, 2, 1, ACT_RELU, false, true, EPI_STORE_RES)
    MLP_CASE(5, 2, 1, ACT_RELU, false, true, EPI_SEGMEAN)
    MLP_CASE(5, 2, 1, ACT_RELU, false, false, EPI_STORE)
    MLP_CASE(5, 2, 2, ACT_RELU, false, true, EPI_STORE)
    MLP_CASE(5, 2, 4, ACT_RELU, false, true, EPI_STORE)
    // single Linear + LayerNorm (+ residual / CSR mean / quad blend): the LAST layer of a chain with mlp_layers != 4, whose hidden layers run as
    // one-Linear launches (magnet_amd/ops.py:_mlp_by_layers; models/backbones/mlp.py:9-28 is generic in its depth)
    MLP_CASE(1, 4, 4, ACT_NONE, false, true, EPI_STORE)
    MLP_CASE(1, 4, 4, ACT_NONE, false, true, EPI_STORE_RES)
    MLP_CASE(1, 4, 4, ACT_NONE, false, true, EPI_SEGMEAN)
    MLP_CASE(1, 1, 1, ACT_NONE, false, true, EPI_STORE)
    MLP_CASE(1, 1, 1, ACT_NONE, false, true, EPI_STORE_RES)
    MLP_CASE(1, 1, 1, ACT_NONE, false, true, EPI_SEGMEAN)
    MLP_CASE(1, 2, 2, ACT_NONE, false, true, EPI_STORE)
    return fn;
}

extern "C" int magnet_mlp_supported(int nl, int ht, int ot, int act, int last_act, int ln, int epi) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    return pick(nl, ht, ot, act, last_act, ln, epi) != nullptr;
}

extern "C" int magnet_mlp_f32(const MagnetMlpArgs* args, int nl, int ht, int ot, int act, int last_act, int ln,
                              int epi, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (!args || args->M <= 0) return args && args->M == 0 ? 0 : -1;
    if (args->nkt < 1 || args->nkt > MAGNET_MAX_KT) return -2;
    if (args->ncolgroups > 1 && (nl != 1 || args->ncolgroups > MAGNET_MAX_LAYERS)) return -2;
    mlp_fn fn = pick(nl, ht, ot, act, last_act, ln, epi);
    if (!fn) return -3;
    const int tiles = (args->M + 31) / 32;
    hipLaunchKernelGGL(fn, dim3((tiles + 3) / 4, args->ncolgroups > 1 ? args->ncolgroups : 1), dim3(256), 0, (hipStream_t)stream, *args);
    return (int)hipGetLastError();
}

extern "C" int magnet_segmean_fixup_f32(const int32_t* rowptr, int n_rows, int d, const float* partials, float* agg,
                                        void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(k_segmean_fixup, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rowptr, n_rows, d,
                       partials, agg);
    return (int)hipGetLastError();
}
