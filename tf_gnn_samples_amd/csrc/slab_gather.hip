// The gather + segment-sum of the aggregate-first RGCN layer with the gathered table tiled through LDS (north_star: "messages
// tiled through LDS").
//
// Replaces the same TF ops as seg_reduce.hip (tf.nn.embedding_lookup gnns/rgcn.py:87-89, the 1/(num_incoming + 1e-7) multiply
// :100-104, tf.concat :108, tf.unsorted_segment_sum :109-112) for batches whose disjoint-union structure is known: a batch is a
// union of graphs, no edge crosses graphs (tasks/ppi_task.py:220-233), so the messages into the buckets of graph g read rows of
// graph g's slab of the state table only.
//
// seg_reduce_wave_kernel gathers one 1 KiB row per message through L1 / L2: at the C2 size it is bound by the L2 -> CU path (93 us
// warm, 117-127 us inside a training step, 0.46-0.63 of the aggregate L2 rate).  Here a workgroup owns (graph, 8-column slice):
//   * it stages the slice of the graph's slab — n_g rows x 8 floats = 32 B per node, <= 150 KB — into LDS ONCE,
//   * every LANE owns one (node, type) bucket and folds its messages SEQUENTIALLY, in bucket order, product and sum rounded
//     separately: the same floating-point operations in the same order as seg_reduce_wave_kernel, hence the same bits
//     (tests/test_gpu_slab_gather.py) — the source rows come from LDS by two ds_read_b128 per message,
//   * the buckets of a graph are taken in order of decreasing length, 64 at a time (a wave's lanes then run nearly the same number
//     of messages), and their message lists are stored for exactly that access: sliced ELLPACK, entry k of lane i of slice q at
//     ell[slice_off[q] + 64 k + i] — one coalesced 128-byte load per wave and step for the (graph-local, 16-bit) row ids, one 256-byte
//     load for the weights.  That layout is a property of the GRAPH (ids are graph-local): built once per data fold
//     (tasks/slab.py), shared by every batch the graph appears in; a batch adds a K-entry table (graph, node offset, nodes).
// Measured upper bound of this inner loop (no index stream, no imbalance: scripts/micro/lds_gather_rate.hip): 51 us per C2 layer.
//
// Bound: LDS bandwidth / VALU issue (22 VALU + 3 memory instructions per message and lane).  HBM / L2 side per launch: the table
// once (37 MB at C2, 128-byte lines shared by four column slices), the ELL lists D/8 times from L2 (6 B per entry), the output once.
#include "common.h"

using namespace relgnn;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WAVES = 8;

struct SlabArgs {
  const float* X; int64_t ldx;          // gathered table [rows, D] (row = node of the batch)
  float* out; int64_t ldo;              // [buckets of the batch, D]: row = bucket_stride * node0 + local bucket id
  float* rowmax;                        // nullable: [buckets * NS] largest finite magnitude of the 8 floats written per (bucket, slice)
  const int64_t* desc;                  // [K][3]: fold graph index, first node of the graph in the batch, nodes — heaviest graph first
  const int32_t* slice_base;            // [G+1] first ELL slice of every graph of the fold
  const int32_t* slice_len;             // [S] steps of a slice = its longest bucket
  const int64_t* slice_off;             // [S] first entry of a slice
  const int32_t* slice_bucket;          // [S*64] graph-local bucket id of every lane (-1: none)
  const int32_t* slice_blen;            // [S*64] messages of that bucket
  const uint16_t* ell_id;               // graph-local row id per entry
  const float* ell_w;                   // nullable: weight per entry
  int32_t bucket_stride;                // buckets per node (L)
  int32_t NS;                           // column slices = D / 8
  int32_t units;                        // K * NS
};

__device__ __forceinline__ uint32_t finite_mag_bits(float x) {
  const uint32_t u = __float_as_uint(x) & 0x7FFFFFFFu;
  return u < 0x7F800000u ? u : 0u;
}

template <bool HAS_W>
__global__ __launch_bounds__(64 * WAVES) void slab_gather_kernel(const SlabArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // Workgroup b runs on XCD b % 8 (round-robin dispatch): ALL column slices of a graph go to ONE XCD, so that its L2 (4 MiB) serves
  // the D/8 re-reads of the graph's message lists (0.7 MB for a PPI-sized graph) and the 128-byte lines of the slab that four
  // neighbouring slices share.  Graphs are ranked by work, heaviest first (desc); round j of an XCD takes rank 8 j + x on even j
  // and 8 j + 7 - x on odd j (snake order: the heaviest graph is paired with the lightest of the next eight).
  const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
  const int j = i / a.NS, c = i - j * a.NS;
  const int r = 8 * j + ((j & 1) ? 7 - x : x);
  if (r >= a.units / a.NS) return;
  const int64_t f = a.desc[3 * r], node0 = a.desc[3 * r + 1];
  const int n = (int)a.desc[3 * r + 2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- the slice of the graph's slab: n rows x 32 bytes -------------------------------------------------------------------
  {
    const float* src = a.X + node0 * a.ldx + 8 * c;
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    for (int i = tid; i < 2 * n; i += 64 * WAVES)
      dst[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)(i >> 1) * a.ldx + 4 * (i & 1));
  }
  __syncthreads();

  const int q0 = a.slice_base[f], q1 = a.slice_base[f + 1];
  const int64_t row0 = node0 * a.bucket_stride;
  for (int q = q0 + wave; q < q1; q += WAVES) {
    const int steps = __builtin_amdgcn_readfirstlane(a.slice_len[q]);
    const int64_t off = a.slice_off[q] + lane;
    const int b = a.slice_bucket[(int64_t)q * 64 + lane];
    const int mine = a.slice_blen[(int64_t)q * 64 + lane];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    int k = 0;
    for (; k + U <= steps; k += U) {
      uint32_t id[U];
      float w[U];
      f32x4 x0[U], x1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {                   // (entries past a bucket's end inside its slice exist: zeros, never added)
        id[u] = a.ell_id[off + (int64_t)(k + u) * 64];
        w[u] = HAS_W ? a.ell_w[off + (int64_t)(k + u) * 64] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        x0[u] = *reinterpret_cast<const f32x4*>(lds + id[u] * 32u);
        x1[u] = *reinterpret_cast<const f32x4*>(lds + id[u] * 32u + 16u);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (k + u < mine) {                           // product and add rounded separately (the file is built with -ffp-contract=off)
          const f32x4 m0 = x0[u] * w[u], m1 = x1[u] * w[u];
          acc0 += m0;
          acc1 += m1;
        }
    }
    for (; k < steps; ++k) {
      const uint32_t id = a.ell_id[off + (int64_t)k * 64];
      const float w = HAS_W ? a.ell_w[off + (int64_t)k * 64] : 1.f;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(lds + id * 32u);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(lds + id * 32u + 16u);
      if (k < mine) {
        const f32x4 m0 = x0 * w, m1 = x1 * w;
        acc0 += m0;
        acc1 += m1;
      }
    }
    if (b >= 0) {
      float* o = a.out + (row0 + b) * a.ldo + 8 * c;
      *reinterpret_cast<f32x4*>(o) = acc0;
      *reinterpret_cast<f32x4*>(o + 4) = acc1;
      if (a.rowmax) {
        uint32_t m = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) m = max(m, max(finite_mag_bits(acc0[e]), finite_mag_bits(acc1[e])));
        a.rowmax[(row0 + b) * a.NS + c] = __uint_as_float(m);
      }
    }
  }
}

}  // namespace

extern "C" {

// rows of one graph's slab slice that fit the LDS of a workgroup (32 bytes each)
int32_t relgnn_slab_gather_max_nodes(void) { return 4800; }

// out[(bucket_stride * node0_j + b), :] = sum over the messages of bucket b of graph j (in bucket order) of w * X[node0_j + id, :]
// for every graph j of the batch (desc) and every local bucket b of it; D % 8 == 0, rows of X and out 16-byte aligned.
// The sliced-ELL arrays describe the graphs of the FOLD (graph-local ids), desc maps batch slots to them.  Same floating-point
// operations in the same order as relgnn_seg_reduce_fwd(RELGNN_AGG_SUM) on the batch's bucketed CSR: bit-identical output.
// rowmax (nullable): [buckets * D / 8] largest finite magnitude of every 8-float piece written.
int relgnn_slab_gather_f32(const float* X, int64_t ldx, int32_t D, const int64_t* desc, int32_t num_graphs, int32_t max_nodes,
                           const int32_t* slice_base, const int32_t* slice_len, const int64_t* slice_off,
                           const int32_t* slice_bucket, const int32_t* slice_blen, const uint16_t* ell_id, const float* ell_w,
                           int32_t bucket_stride, float* out, int64_t ldo, float* rowmax, void* stream) {
  if (D < 0 || num_graphs < 0 || max_nodes < 0 || bucket_stride < 1) return RELGNN_EINVAL;
  if (D == 0 || num_graphs == 0) return RELGNN_OK;
  if (!X || !desc || !slice_base || !slice_len || !slice_off || !slice_bucket || !slice_blen || !ell_id || !out) return RELGNN_EINVAL;
  if (D % 8 != 0 || ldx % 4 != 0 || ldo % 4 != 0 || ldx < D || ldo < D || !aligned16(X) || !aligned16(out)) return RELGNN_EUNSUPPORTED;
  if (max_nodes > relgnn_slab_gather_max_nodes()) return RELGNN_EUNSUPPORTED;
  SlabArgs a{};
  a.X = X; a.ldx = ldx; a.out = out; a.ldo = ldo; a.rowmax = rowmax; a.desc = desc; a.slice_base = slice_base; a.slice_len = slice_len;
  a.slice_off = slice_off; a.slice_bucket = slice_bucket; a.slice_blen = slice_blen; a.ell_id = ell_id; a.ell_w = ell_w;
  a.bucket_stride = bucket_stride; a.NS = D / 8; a.units = num_graphs * a.NS;
  const size_t lds_bytes = (size_t)(max_nodes > 0 ? max_nodes : 1) * 32;
  // (more than 64 KB of dynamic LDS has to be asked for; a per-function attribute of the current device, set on every call: no
  //  state of this library's own)
  const int cap = relgnn_slab_gather_max_nodes() * 32;
  const void* fn = ell_w ? reinterpret_cast<const void*>(&slab_gather_kernel<true>) : reinterpret_cast<const void*>(&slab_gather_kernel<false>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess) return RELGNN_EHIP;
  hipStream_t st = as_stream(stream);
  const unsigned grid = 8u * (unsigned)((num_graphs + 7) / 8) * (unsigned)a.NS;
  if (ell_w) slab_gather_kernel<true><<<grid, 64 * WAVES, lds_bytes, st>>>(a);
  else slab_gather_kernel<false><<<grid, 64 * WAVES, lds_bytes, st>>>(a);
  return launch_status();
}

}  // extern "C"
