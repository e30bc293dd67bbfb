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
//     of messages), and their message lists are stored for exactly that access: sliced ELLPACK in chunks of 8 steps, entry k of lane
//     i of slice q at ell[slice_off[q] + 512 (k / 8) + 8 i + k % 8] — per chunk one 16-byte load per lane for the 8 (graph-local,
//     16-bit) row ids and two for the 8 weights, 1 KB + 2 KB contiguous per wave.  That layout is a property of the GRAPH (ids are graph-local): built once per data fold
//     (tasks/slab.py), shared by every batch the graph appears in; a batch adds a K-entry table (graph, node offset, nodes).
// Measured upper bound of this inner loop (no index stream, no imbalance: scripts/micro/lds_gather_rate.hip): 51 us per C2 layer.
//
// Bound: LDS bandwidth / VALU issue (22 VALU + 3 memory instructions per message and lane).  HBM / L2 side per launch: the table
// once (37 MB at C2, 128-byte lines shared by four column slices), the ELL lists D/8 times from L2 (6 B per entry), the output once.
#include "common.h"
#include <cstdlib>

using namespace relgnn;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int WAVES = 16;               // one workgroup per CU (the slab slice takes most of its LDS): 4 waves per SIMD
constexpr int CH = 8;                   // steps per chunk: slice lengths are padded to a multiple (tasks/slab.py)

struct SlabArgs {
  const float* X; int64_t ldx;          // gathered table [rows, D] (row = node of the batch)
  float* out; int64_t ldo;              // [buckets of the batch, D]: row = bucket_stride * node0 + local bucket id
  float* rowmax;                        // nullable: [buckets * NS] largest finite magnitude of the 8 floats written per (bucket, slice)
  const int64_t* desc;                  // [K][3]: fold graph index, first node of the graph in the batch, nodes — heaviest graph first
  const int32_t* slice_base;            // [G+1] first ELL slice of every graph of the fold
  const int32_t* slice_len;             // [S] steps of a slice = its longest bucket (stored in whole chunks of CH steps)
  const int64_t* slice_off;             // [S] first entry of a slice
  const int32_t* slice_bucket;          // [S*64] graph-local bucket id of every lane (-1: none)
  const uint16_t* ell_id;               // graph-local row id per entry; entries past a bucket's end: the graph's node count
  const float* ell_w;                   // nullable: weight per entry
  int32_t* ticket;                      // [9] zero before the first launch: next unit per XCD, workgroups done (the last one re-zeroes)
  int32_t num_graphs;
  int32_t bucket_stride;                // buckets per node (L)
  int32_t NS;                           // column slices = D / 8
  int32_t units;                        // K * NS
  int32_t debug;
  int32_t lds_rows;                     // rows of the LDS slab (largest graph + the row of zeros); two ints behind them
};

__device__ __forceinline__ uint32_t finite_mag_bits(float x) {
  const uint32_t u = __float_as_uint(x) & 0x7FFFFFFFu;
  return u < 0x7F800000u ? u : 0u;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int N> struct Int { static constexpr int value = N; };

// The list entries of one chunk (CH = 8 steps) of one lane: 8 row ids in 16 bytes, 8 weights in 32 — one dwordx4 and two dwordx4
// loads per lane (the lists are stored lane-major inside a chunk for exactly this).
struct Chunk {
  u32x4 id;
  f32x4 w0, w1;
};

template <bool HAS_W>
__device__ __forceinline__ void load_chunk(const uint16_t* pid, const float* pw, int64_t k, Chunk& c) {
  c.id = *reinterpret_cast<const u32x4*>(pid + k * 64);
  if (HAS_W) {
    c.w0 = *reinterpret_cast<const f32x4*>(pw + k * 64);
    c.w1 = *reinterpret_cast<const f32x4*>(pw + k * 64 + 4);
  }
}

// One unit of work = (graph of the batch, 8-column slice), described for the whole workgroup by the thread that drew it.
struct Unit {
  int32_t c;            // column slice; -1: the queue is empty
  int32_t n;            // nodes of the graph (0: nothing to do — a slice index past D/8 in the last column group)
  int32_t q0, q1;       // ELL slices of the graph
  int64_t node0;        // first node of the graph in the batch
};

// Column slices are handed out in groups that share 128-byte lines of X and of out (4 slices = 32 columns, fewer for narrow tables),
// every group always to the same XCD: its L2 then serves the slab loads of the group's other slices and merges their 32-byte
// output pieces into whole lines before they leave for HBM.  Workgroup b runs on XCD b % 8 (round-robin dispatch); each XCD draws
// from its own ticket, heaviest graph first (longest-processing-time-first inside the XCD; all XCDs get the same work).
__device__ __forceinline__ void draw_unit(const SlabArgs& a, int x, Unit* u) {
  const int gs = a.NS >= 32 ? 4 : (a.NS >= 16 ? 2 : 1);            // slices per column group
  const int groups = (a.NS + gs - 1) / gs;
  const int mine = (groups - x + 7) / 8;                            // groups x, x + 8, ... belong to this XCD
  const int t = atomicAdd(a.ticket + x, 1);
  const int per_graph = mine * gs;
  if (per_graph == 0 || t >= a.num_graphs * per_graph) { u->c = -1; return; }
  const int r = t / per_graph, rem = t - r * per_graph;
  const int c = ((rem / gs) * 8 + x) * gs + rem % gs;
  const int64_t f = a.desc[3 * r];
  u->c = c;
  u->n = c < a.NS ? (int)a.desc[3 * r + 2] : 0;
  u->node0 = a.desc[3 * r + 1];
  u->q0 = a.slice_base[f];
  u->q1 = a.slice_base[f + 1];
}

template <bool HAS_W>
__global__ __launch_bounds__(64 * WAVES) void slab_gather_kernel(const SlabArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // the slab slice at offset 0: a row id is its address / 32
  Unit* units = reinterpret_cast<Unit*>(lds + a.lds_rows * 32);            // [2]: the unit being worked on, the next one
  int& s_slice = *reinterpret_cast<int*>(lds + a.lds_rows * 32 + 2 * sizeof(Unit));
  const int tid = threadIdx.x, lane = tid & 63;
  const int x = blockIdx.x & 7;

  if (tid == 0) draw_unit(a, x, &units[0]);
  __syncthreads();
  for (int turn = 0;; turn ^= 1) {
    const Unit u = units[turn];
    if (u.c < 0) break;
    const int c = u.c, n = u.n, q1 = u.q1;
    // ---- the slice of the graph's slab: n rows x 32 bytes, and one row of zeros behind it (what padded entries read) ---------
    {
      const float* src = a.X + u.node0 * a.ldx + 8 * c;
      f32x4* dst = reinterpret_cast<f32x4*>(lds);
      if (!(a.debug & 1))
      for (int i = tid; i < 2 * n; i += 64 * WAVES)
        dst[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)(i >> 1) * a.ldx + 4 * (i & 1));
      if (tid < 2) dst[2 * n + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (tid == 0) s_slice = n > 0 ? u.q0 : q1;
    }
    __syncthreads();
    if (tid == 64 * WAVES - 1) draw_unit(a, x, &units[turn ^ 1]);         // (its latency — an atomic, three loads — under the unit's work)

    const int64_t row0 = u.node0 * a.bucket_stride;
    // Slices of a graph are stored longest first; a wave draws the next one when it is done with its last.  Three slices in flight
    // per wave: the one being folded, the next (its descriptor read, its first chunk of list entries on the way) and the one after
    // (drawn, its descriptor on the way) — a slice is 25 steps on average and its descriptor -> first chunk -> LDS chain is two
    // memory latencies, which at four waves per SIMD nothing else would cover.
    const auto draw = [&]() {
      int q = 0;
      if (lane == 0) q = atomicAdd(&s_slice, 1);
      return __builtin_amdgcn_readfirstlane(q);
    };
    struct Slice { int q; int steps; int64_t off; int b; };
    const auto describe = [&](int q) {                                      // (vector loads: in-order with the list loads)
      Slice s;
      const int qq = q < q1 ? q : q1 - 1;
      s.q = q;
      s.steps = __builtin_nontemporal_load(a.slice_len + qq + 0 * lane);
      s.off = a.slice_off[qq + 0 * lane];
      s.b = a.slice_bucket[(int64_t)qq * 64 + lane];
      return s;
    };
    Slice s0 = describe(draw());
    Slice s1 = describe(draw());
    Chunk first;
    load_chunk<HAS_W>(a.ell_id + s0.off + lane * CH, HAS_W ? a.ell_w + s0.off + lane * CH : nullptr, 0, first);
    while (s0.q < q1) {
      const int steps = __builtin_amdgcn_readfirstlane(s0.steps);
      const int64_t off = __builtin_amdgcn_readfirstlane((int)(s0.off >> 32)) * (int64_t(1) << 32) +
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)s0.off) + lane * CH;
      const int b = s0.b;
      const uint16_t* pid = a.ell_id + off;
      const float* pw = HAS_W ? a.ell_w + off : nullptr;
      Chunk ca = first, cb;
      // the slice after this one: its first chunk (its descriptor came in during the last slice); the one after that: drawn now
      load_chunk<HAS_W>(a.ell_id + s1.off + lane * CH, HAS_W ? a.ell_w + s1.off + lane * CH : nullptr, 0, first);
      Slice s2 = describe(draw());
      __builtin_amdgcn_sched_barrier(0);

      f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      const auto fold = [&](const Chunk& ch, auto from, auto count) {
        constexpr int U0 = decltype(from)::value, N = decltype(count)::value;
        f32x4 x0[N], x1[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const int u = U0 + i;
          const uint32_t id = (u & 1) ? ch.id[u / 2] >> 16 : ch.id[u / 2] & 0xFFFFu;
          x0[i] = *reinterpret_cast<const f32x4*>(lds + id * 32u);
          x1[i] = *reinterpret_cast<const f32x4*>(lds + id * 32u + 16u);
        }
        // Product and add rounded separately (-ffp-contract=off), bucket order: the operations of seg_reduce_wave_kernel.  A padded
        // entry adds the zero row: acc + (+0) is acc bit for bit — acc is never -0 (it starts at +0, and a round-to-nearest sum
        // is -0 only when both terms are).
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const int u = U0 + i;
          if (HAS_W) {
            const float w = u < 4 ? ch.w0[u & 3] : ch.w1[u & 3];
            const f32x2 ww = {w, w};
            const f32x2 m0 = x0[i].xy * ww, m1 = x0[i].zw * ww, m2 = x1[i].xy * ww, m3 = x1[i].zw * ww;
            acc[0] += m0; acc[1] += m1; acc[2] += m2; acc[3] += m3;
          } else {
            acc[0] += x0[i].xy; acc[1] += x0[i].zw; acc[2] += x1[i].xy; acc[3] += x1[i].zw;
          }
        }
      };
      const auto whole = [&](const Chunk& ch) {
        fold(ch, Int<0>{}, Int<4>{});
        fold(ch, Int<4>{}, Int<4>{});
      };
      const auto tail = [&](const Chunk& ch, int r) {          // the last chunk of a slice: its first r (1..8) steps
        if (r >= 4) {
          fold(ch, Int<0>{}, Int<4>{});
          switch (r) {
            case 4: break;
            case 5: fold(ch, Int<4>{}, Int<1>{}); break;
            case 6: fold(ch, Int<4>{}, Int<2>{}); break;
            case 7: fold(ch, Int<4>{}, Int<3>{}); break;
            default: fold(ch, Int<4>{}, Int<4>{}); break;
          }
        } else {
          switch (r) {
            case 1: fold(ch, Int<0>{}, Int<1>{}); break;
            case 2: fold(ch, Int<0>{}, Int<2>{}); break;
            case 3: fold(ch, Int<0>{}, Int<3>{}); break;
            default: break;
          }
        }
      };
      if (steps > 0 && !(a.debug & 4)) {
        // chunks 0 .. last-1 are whole; `steps` (the slice's longest bucket) ends inside chunk `last`
        const int last = (steps - 1) / CH, rem = steps - last * CH;
        const int kst = (a.debug & 2) ? 0 : CH;
        int ci = 0;
        // Two chunks of list entries in flight, in two register sets taken in turn; every load is unconditional (the lists end in
        // two chunks of padding, tasks/slab.py) so that the loop body is straight-line code: with a branch around a prefetch the
        // compiler's wait-count pass joins the two paths and waits for the prefetch itself.
        for (; ci + 2 <= last; ci += 2) {
          load_chunk<HAS_W>(pid, pw, (ci + 1) * kst, cb);
          __builtin_amdgcn_sched_barrier(0);
          whole(ca);
          __builtin_amdgcn_sched_barrier(0);
          load_chunk<HAS_W>(pid, pw, (ci + 2) * kst, ca);
          __builtin_amdgcn_sched_barrier(0);
          whole(cb);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ci < last) {
          load_chunk<HAS_W>(pid, pw, (ci + 1) * kst, cb);
          whole(ca);
          tail(cb, rem);
        } else {
          tail(ca, rem);
        }
      }
      if (b >= 0 && !(a.debug & 8)) {
        float* o = a.out + (row0 + b) * a.ldo + 8 * c;
        const f32x4 o0 = {acc[0].x, acc[0].y, acc[1].x, acc[1].y}, o1 = {acc[2].x, acc[2].y, acc[3].x, acc[3].y};
        *reinterpret_cast<f32x4*>(o) = o0;
        *reinterpret_cast<f32x4*>(o + 4) = o1;
        if (a.rowmax) {
          uint32_t m = 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) m = max(m, max(finite_mag_bits(o0[e]), finite_mag_bits(o1[e])));
          a.rowmax[(row0 + b) * a.NS + c] = __uint_as_float(m);
        }
      }
      s0 = s1;
      s1 = s2;
    }
    __syncthreads();                                  // every wave is done with this slab; units[turn ^ 1] is written
  }
  // the last workgroup to leave puts the tickets back for the next launch on this stream
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(a.ticket + 8, 1) == (int)gridDim.x - 1) {
      for (int i = 0; i < 9; ++i) a.ticket[i] = 0;
      __threadfence();
    }
  }
}

}  // namespace

extern "C" {

// rows of one graph's slab slice that fit the LDS of a workgroup (32 bytes each, one row of zeros behind them)
int32_t relgnn_slab_gather_max_nodes(void) { return 4800; }
// steps of a slice are padded to a multiple of this (entries past a bucket's end name row `nodes of the graph`)
int32_t relgnn_slab_gather_chunk(void) { return CH; }
int32_t relgnn_slab_gather_ticket_ints(void) { return 9; }

// out[(bucket_stride * node0_j + b), :] = sum over the messages of bucket b of graph j (in bucket order) of w * X[node0_j + id, :]
// for every graph j of the batch (desc) and every local bucket b of it; D % 8 == 0, rows of X and out 16-byte aligned.
// The sliced-ELL arrays describe the graphs of the FOLD (graph-local ids), desc maps batch slots to them.  Same floating-point
// operations in the same order as relgnn_seg_reduce_fwd(RELGNN_AGG_SUM) on the batch's bucketed CSR: bit-identical output.
// rowmax (nullable): [buckets * D / 8] largest finite magnitude of every 8-float piece written.
// ticket: relgnn_slab_gather_ticket_ints() int32 on the device, zero before the first launch, private to the stream (the kernel
// leaves them zero).
int relgnn_slab_gather_f32(const float* X, int64_t ldx, int32_t D, const int64_t* desc, int32_t num_graphs, int32_t max_nodes,
                           const int32_t* slice_base, const int32_t* slice_len, const int64_t* slice_off,
                           const int32_t* slice_bucket, const uint16_t* ell_id, const float* ell_w,
                           int32_t bucket_stride, float* out, int64_t ldo, float* rowmax, int32_t* ticket, void* stream) {
  if (D < 0 || num_graphs < 0 || max_nodes < 0 || bucket_stride < 1) return RELGNN_EINVAL;
  if (D == 0 || num_graphs == 0) return RELGNN_OK;
  if (!X || !desc || !slice_base || !slice_len || !slice_off || !slice_bucket || !ell_id || !out || !ticket) return RELGNN_EINVAL;
  if (D % 8 != 0 || ldx % 4 != 0 || ldo % 4 != 0 || ldx < D || ldo < D || !aligned16(X) || !aligned16(out)) return RELGNN_EUNSUPPORTED;
  if (max_nodes > relgnn_slab_gather_max_nodes()) return RELGNN_EUNSUPPORTED;
  SlabArgs a{};
  a.X = X; a.ldx = ldx; a.out = out; a.ldo = ldo; a.rowmax = rowmax; a.desc = desc; a.slice_base = slice_base; a.slice_len = slice_len;
  a.slice_off = slice_off; a.slice_bucket = slice_bucket; a.ell_id = ell_id; a.ell_w = ell_w; a.ticket = ticket;
  a.bucket_stride = bucket_stride; a.NS = D / 8; a.units = num_graphs * a.NS; a.num_graphs = num_graphs;
  a.lds_rows = max_nodes + 1;
  { const char* e = getenv("RELGNN_SLAB_DEBUG"); a.debug = e ? atoi(e) : 0; if (a.debug & 16) a.units /= 2; }
  const size_t lds_bytes = (size_t)a.lds_rows * 32 + 2 * sizeof(Unit) + 16;
  // (more than 64 KB of dynamic LDS has to be asked for; a per-function attribute of the current device, set on every call: no
  //  state of this library's own)
  const int cap = (relgnn_slab_gather_max_nodes() + 1) * 32 + 2 * (int)sizeof(Unit) + 16;
  const void* fn = ell_w ? reinterpret_cast<const void*>(&slab_gather_kernel<true>) : reinterpret_cast<const void*>(&slab_gather_kernel<false>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess) return RELGNN_EHIP;
  int device = 0, cus = 0;
  if (hipGetDevice(&device) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess)
    return RELGNN_EHIP;
  hipStream_t st = as_stream(stream);
  if (cus < 8) return RELGNN_EUNSUPPORTED;
  const unsigned grid = (unsigned)(cus / 8 * 8);            // every XCD's queue is served: workgroup b -> XCD b % 8
  if (ell_w) slab_gather_kernel<true><<<grid, 64 * WAVES, lds_bytes, st>>>(a);
  else slab_gather_kernel<false><<<grid, 64 * WAVES, lds_bytes, st>>>(a);
  return launch_status();
}

}  // extern "C"
