/*
 * relgnn.h — C ABI of librelgnn.so: the MI355X (gfx950) sparse relational
 * message-passing hot path of microsoft/tf-gnn-samples.
 *
 * The reference has NO FFI boundary of its own for this path: below the Python
 * functions gnns.sparse_*_layer() it calls stock TensorFlow-1 ops.  Each entry
 * point below therefore replaces one TF op *call site family* of the reference
 * (cited as reference file:line, relative to the reference root) and is what a
 * maintainer would bind from Python (ctypes stub: INTEGRATION.md).
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller unless the
 *     parameter name starts with `h_` (host pointer).  Nothing is allocated,
 *     freed or retained by the library; workspaces are caller-allocated and
 *     sized by the matching *_workspace_bytes() query.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *     work is enqueued asynchronously on it; no entry point synchronises.
 *   - Every entry point returns an int status (RELGNN_OK == 0) and never
 *     throws.  Index-range errors found ON DEVICE are reported through the
 *     caller-provided `err_flag` word (see relgnn_relational_keys), because
 *     reporting them through the return value would need a device sync.
 *   - float == IEEE binary32, index == int32_t, row-major everywhere.
 *   - Functions are re-entrant w.r.t. distinct streams/devices; there is no
 *     global mutable state.
 */
#ifndef RELGNN_H_
#define RELGNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RELGNN_ABI_VERSION 1

/* ---- status codes ------------------------------------------------------- */
#define RELGNN_OK 0
#define RELGNN_EINVAL 1 /* bad argument (null pointer, negative size, bad enum, misalignment) */
#define RELGNN_ENOSPC 2 /* workspace too small */
#define RELGNN_EHIP 3   /* a HIP runtime call / kernel launch failed */
#define RELGNN_EUNSUPPORTED 4

/* bits OR-ed into *err_flag by device-side validation */
#define RELGNN_ERRFLAG_INDEX_OUT_OF_RANGE 1u /* TF: InvalidArgumentError for gather / segment ids */

/* ---- aggregation modes: utils/utils.py:23-33 (get_aggregation_function) -- */
#define RELGNN_AGG_SUM 0    /* tf.unsorted_segment_sum                      */
#define RELGNN_AGG_MEAN 1   /* tf.unsorted_segment_mean   : sum / max(n,1)   */
#define RELGNN_AGG_SQRT_N 2 /* tf.unsorted_segment_sqrt_n : sum / sqrt(max(n,1)) */
#define RELGNN_AGG_MAX 3    /* tf.unsorted_segment_max    : empty -> -FLT_MAX */

/* ---- activations: utils/utils.py:36-58 (get_activation) ------------------ */
#define RELGNN_ACT_LINEAR 0
#define RELGNN_ACT_TANH 1
#define RELGNN_ACT_RELU 2
#define RELGNN_ACT_LEAKY_RELU 3 /* tf.nn.leaky_relu default alpha = 0.2 */
#define RELGNN_ACT_ELU 4
#define RELGNN_ACT_SELU 5
#define RELGNN_ACT_GELU 6 /* erf form, utils/utils.py:53-55 */

int relgnn_abi_version(void);
const char* relgnn_status_string(int status);

/* ========================================================================== *
 * 1. Index bookkeeping (bit-exact integer work)
 * ========================================================================== */

/*
 * relgnn_relational_keys — flatten one adjacency list into the type-major
 * message list the reference builds with tf.concat.
 *
 * Replaces: the `adjacency_list_for_edge_type[:, 0] / [:, 1]` slicing and
 * `tf.concat(edge_type_to_message_targets, axis=0)` at gnns/rgcn.py:68-78
 * (identically ggnn.py:59-69, rgat.py:68-80, rgin.py:90-101,
 * gnn_film.py:68-83, gnn_edge_mlp.py:72-82).
 *
 * adj        : [num_edges, 2] int32, adj[e] = {source, target}
 *              (messages flow column 0 -> column 1, gnns/rgcn.py:85-86)
 * msg_base   : offset of this edge type's first message in the type-major
 *              message list (= sum of num_edges of lower edge types)
 * For message m = msg_base + e it writes
 *   key_by_target[m] = target * num_edge_types + edge_type
 *   key_by_source[m] = source * num_edge_types + edge_type
 * and ORs RELGNN_ERRFLAG_INDEX_OUT_OF_RANGE into *err_flag (if non-null) when
 * source or target is outside [0, num_nodes); the keys of an invalid edge are
 * clamped into range so that later stages stay in bounds (the caller is expected
 * to check the flag and raise, as TF-CPU does with InvalidArgumentError).
 */
int relgnn_relational_keys(const int32_t* adj, int64_t num_edges, int32_t edge_type,
                           int32_t num_edge_types, int32_t num_nodes, int64_t msg_base,
                           int32_t* key_by_target, int32_t* key_by_source,
                           uint32_t* err_flag, void* stream);

/* As relgnn_relational_keys, additionally writing the plain node ids target_node[m], source_node[m] (nullable):
 * the sort keys of relgnn_relational_plan. */
int relgnn_relational_keys2(const int32_t* adj, int64_t num_edges, int32_t edge_type,
                            int32_t num_edge_types, int32_t num_nodes, int64_t msg_base,
                            int32_t* key_by_target, int32_t* key_by_source, int32_t* target_node,
                            int32_t* source_node, uint32_t* err_flag, void* stream);

/* relgnn_relational_keys2 for ALL edge types of a batch in one launch per 32 types: h_adj / h_num_edges are HOST arrays
 * of num_edge_types device pointers / edge counts; messages are numbered type-major (type 0 first), as in the reference's
 * tf.concat over the per-type lists (gnns/rgcn.py:78). */
int relgnn_relational_keys_all(const int32_t* const* h_adj, const int64_t* h_num_edges, int32_t num_edge_types,
                               int32_t num_nodes, int32_t* key_by_target, int32_t* key_by_source, int32_t* target_node,
                               int32_t* source_node, uint32_t* err_flag, void* stream);

/*
 * relgnn_relational_plan — the (node, type)-bucketed CSR of the type-major message list in one call.
 * Because the message list is type-major (gnns/rgcn.py:78), a STABLE sort by node id alone already orders the
 * messages by (node, type, original edge order): only ceil(log2 V) key bits go through the radix passes.
 *   sort_node : [M] node id to bucket by (target_node for the forward plan, source_node for the transposed plan)
 *   full_key  : [M] node*L + type of the same side (defines the V*L buckets of rowptr)
 *   other_key : [M] node*L + type of the OTHER side
 * Outputs (sorted position p, original message m = perm[p]):
 *   rowptr [V*L+1], perm [M], col[p] = other_key[m], col_div[p] = other_key[m] / L (nullable),
 *   inv_out[m] = p (nullable: the inverse permutation), pos_out[p] = inv_in[m] (nullable pair: position of the
 *   same message in another plan's order).
 */
size_t relgnn_relational_plan_workspace_bytes(int64_t num_messages, int32_t num_nodes);
int relgnn_relational_plan(const int32_t* sort_node, const int32_t* full_key, const int32_t* other_key,
                           int64_t num_messages, int32_t num_nodes, int32_t num_edge_types,
                           int32_t* rowptr, int32_t* perm, int32_t* col, int32_t* col_div,
                           int32_t* inv_out, const int32_t* inv_in, int32_t* pos_out, void* workspace,
                           size_t workspace_bytes, void* stream);

/*
 * relgnn_segment_plan — stable bucketing of messages by segment id.
 *
 * Replaces: the implicit scatter order of tf.unsorted_segment_* (utils/utils.py:23-33):
 * a STABLE sort by segment id makes every segment's messages contiguous while
 * keeping the reference's message order inside the segment, so a sequential
 * in-register accumulation reproduces the sequential CPU kernel's summation
 * order without atomics.
 *
 * keys    : [num_messages] int32 in [0, num_segments) (negative ids are dropped by
 *           TF; here they are NOT allowed: validate first)
 * rowptr  : [num_segments + 1] out; messages of segment s are sorted positions
 *           rowptr[s] .. rowptr[s+1]-1
 * perm    : [num_messages] out; perm[p] = original message index at sorted position p
 *           (ascending inside a segment: stability)
 * sorted_keys : [num_messages] out (may be NULL if the caller does not need it)
 */
size_t relgnn_segment_plan_workspace_bytes(int64_t num_messages, int64_t num_segments);
int relgnn_segment_plan(const int32_t* keys, int64_t num_messages, int64_t num_segments,
                        int32_t* rowptr, int32_t* perm, int32_t* sorted_keys,
                        void* workspace, size_t workspace_bytes, void* stream);

/* out[i] = table[index[i]]  (int32).  Index plumbing between the two CSR orders. */
int relgnn_gather_i32(const int32_t* table, const int32_t* index, int64_t n, int32_t* out,
                      void* stream);
/* out[i] = table[index[i]] / divisor  (int32; recovers node id from node*L+type keys). */
int relgnn_gather_div_i32(const int32_t* table, const int32_t* index, int64_t n,
                          int32_t divisor, int32_t* out, void* stream);
/* out[i] = table[index[i]]  (float32). */
int relgnn_gather_f32(const float* table, const int32_t* index, int64_t n, float* out,
                      void* stream);
/* inv[perm[p]] = p. */
int relgnn_invert_perm(const int32_t* perm, int64_t n, int32_t* inv, void* stream);

/*
 * relgnn_degree_scale — per-message 1/(c + eps) normalisation weights.
 *
 * Replaces: gnns/rgcn.py:100-104 (identically gnn_film.py:96-100,
 * gnn_edge_mlp.py:104-108): embedding_lookup of type_to_num_incoming_edges[l, :]
 * by edge target, `1.0 / (x + SMALL_NUMBER)` in float32.
 *
 * degree_table : [num_edge_types, num_nodes] float32 exactly as fed by the task
 *                (tasks/sparse_graph_task.py:144-145)
 * rowptr       : [num_nodes*num_edge_types + 1] by-(target,type) CSR row pointer
 * scale        : [num_messages] out, in by-target sorted order:
 *                scale[p] = 1.0f / (degree_table[l, v] + eps) for p in sub-segment (v, l)
 */
int relgnn_degree_scale(const float* degree_table, const int32_t* rowptr, int32_t num_edge_types,
                        int32_t num_nodes, float eps, float* scale, void* stream);

/*
 * relgnn_segment_counts_scale — backward helper for mean / sqrt_n:
 * scale[p] = (w ? w[p] : 1) * f(n_seg(p)),  f = 1/max(n,1) (MEAN) or 1/sqrt(max(n,1)) (SQRT_N),
 * where n_seg is the length of the (merged) segment that holds sorted position p.
 */
int relgnn_segment_counts_scale(const int32_t* rowptr, int64_t num_segments, int32_t seg_stride,
                                int32_t mode, const float* w, float* scale, void* stream);

/* ========================================================================== *
 * 2. The hot kernel: fused gather + (scale) + segment reduce
 * ========================================================================== */

/*
 * relgnn_seg_reduce_fwd
 *
 *   out[s, :] = finalize_mode( REDUCE_{p = rowptr[s*seg_stride]}^{rowptr[(s+1)*seg_stride]-1}
 *                                  (w ? w[p] : 1) * X[col[p], :] )          s in [0, num_segments)
 *
 * accumulated SEQUENTIALLY in p (one rounding for the product, one for the add:
 * no FMA contraction), which is the order of TF-CPU's UnsortedSegmentSum on the
 * reference's concatenated message tensor.
 *
 * Replaces, fused into one pass: tf.nn.embedding_lookup (gnns/rgcn.py:87-89), the
 * 1/in-degree multiply (rgcn.py:100-104), tf.concat (rgcn.py:108) and
 * tf.unsorted_segment_{sum,mean,sqrt_n,max} (rgcn.py:109-112 via utils/utils.py:23-33);
 * same call-site family in ggnn.py:76-89, rgin.py:110-133, gnn_film.py:92-116,
 * gnn_edge_mlp.py:91-116 and the task head tasks/qm9_task.py:185-187.
 *
 * X       : [num_rows_x, ldx] float32, only the first D columns of a row are read
 * rowptr  : [num_segments*seg_stride + 1]; seg_stride > 1 merges that many consecutive
 *           sub-segments (the per-edge-type buckets of one target node) into one output row
 * col     : [num_messages] row of X gathered by each sorted message
 * w       : [num_messages] or NULL
 * out     : [num_segments, ldo] float32, first D columns written
 * act     : RELGNN_ACT_* applied to the finalized row before the store (epilogue
 *           fusion of rgcn.py:114); RELGNN_ACT_LINEAR for none
 * D, ldx, ldo in floats.  Fast path needs D%4==0, ldx%4==0, ldo%4==0 and 16-byte
 * aligned X/out; anything else takes the scalar-lane kernel.
 */
int relgnn_seg_reduce_fwd(int32_t mode, const float* X, int64_t num_rows_x, int64_t ldx, int32_t D,
                          const int32_t* rowptr, int64_t num_segments, int32_t seg_stride,
                          const int32_t* col, const float* w, int32_t act, float* out,
                          int64_t ldo, void* stream);

/*
 * relgnn_seg_reduce_acc64_fwd — the same gather + scale + segment SUM (sum / mean / sqrt_n; not max), with each
 * bucket accumulated in float64 (w * x is exact there) and rounded to float32 once.
 *
 * For bucket sums that FEED A GEMM — the aggregate-then-transform evaluation of gnns/rgcn.py:84-114
 * (out = sum_l A_l W_l with A_l[v] = sum_{(u,v) in A_l} 1/(c_{l,v}+1e-7) h_u): the reference never forms A_l, so there
 * is no reference summation order to reproduce; what counts against the 1e-5 budget is how much rounding the
 * aggregation adds in front of the K = L*D dot products.  relgnn_seg_reduce_fwd (sequential float32, the order of
 * tf.unsorted_segment_sum on the reference's message tensor) stays the kernel wherever the reduced values ARE the
 * reference's segment sums.  Rows of 132 .. 1024 floats, 16-byte aligned (RELGNN_EUNSUPPORTED otherwise).
 */
int relgnn_seg_reduce_acc64_fwd(int32_t mode, const float* X, int64_t num_rows_x, int64_t ldx, int32_t D,
                                const int32_t* rowptr, int64_t num_segments, int32_t seg_stride,
                                const int32_t* col, const float* w, int32_t act, float* out,
                                int64_t ldo, void* stream);

/*
 * relgnn_seg_reduce_fwd_rowmax — relgnn_seg_reduce_fwd (sum / mean / sqrt_n) that also writes rowmax[s] = the largest magnitude of
 * output row s: what the consumer of these rows needs when it evaluates its product from two fp16 limbs per value
 * (relgnn_limb16_gemm_xf32: the row's power-of-two scale).  The wave that reduces a bucket holds the whole row, so the maximum costs
 * a wave reduction and one store.  Rows of 132 .. 1024 floats, 16-byte aligned (RELGNN_EUNSUPPORTED otherwise).
 */
int relgnn_seg_reduce_fwd_rowmax(int32_t mode, const float* X, int64_t num_rows_x, int64_t ldx, int32_t D,
                                 const int32_t* rowptr, int64_t num_segments, int32_t seg_stride,
                                 const int32_t* col, const float* w, int32_t act, float* out,
                                 int64_t ldo, float* rowmax, void* stream);

/*
 * Per-MESSAGE activation before the reduction (gnns/gnn_edge_mlp.py:104-116, rgin.py:127-133: scale, activation,
 * segment reduce on the materialised message tensor) without an elementwise pass over [M, D]:
 *   relgnn_seg_reduce_msgact_fwd : out[s,:] = finalize_mode( REDUCE_p msg_act( w[p] * X[col[p],:] ) )
 *   relgnn_msg_act_bwd           : gX[m,:] = w[m] * msg_act'(w[m]*X[m,:]) * gagg[tgt[m],:]   (X [M, D] contiguous,
 *                                  w / tgt in X's row order, gagg [V, D] = gradient of the un-finalised aggregate)
 */
int relgnn_seg_reduce_msgact_fwd(int32_t mode, int32_t msg_act, const float* X, int64_t num_rows_x, int64_t ldx,
                                 int32_t D, const int32_t* rowptr, int64_t num_segments,
                                 int32_t seg_stride, const int32_t* col, const float* w, float* out,
                                 int64_t ldo, void* stream);
int relgnn_msg_act_bwd(int32_t act, const float* X, int32_t D, const float* w, const int32_t* tgt,
                       const float* gagg, int64_t num_messages, float* gX, void* stream);

/*
 * unsorted_segment_max gradient w.r.t. the gathered table, TF semantics
 * (math_grad.py _UnsortedSegmentMinOrMaxGrad [TF-internal]): the gradient of out[s,d] is
 * split EQUALLY among all messages p of segment s with w[p]*X[col[p],d] == out[s,d].
 *
 * relgnn_seg_max_count (forward plan):
 *   gsel[s,d] = gout[s,d] / #{p in s : w[p]*X[col[p],d] == out[s,d]}      (0 for empty s)
 * relgnn_seg_max_bwd (TRANSPOSED plan: segments = rows r of X, entries q = the output
 * segment seg_b[q] each message of r went to):
 *   gX[r,d] = sum_q [ w_b[q]*X[r,d] == out[seg_b[q],d] ] * w_b[q] * gsel[seg_b[q],d]
 */
int relgnn_seg_max_count(const float* X, int64_t ldx, int32_t D, const int32_t* rowptr,
                         int64_t num_segments, int32_t seg_stride, const int32_t* col,
                         const float* w, const float* out, const float* gout, int64_t ldo,
                         float* gsel, void* stream);
int relgnn_seg_max_bwd(const float* X, int64_t ldx, int32_t D, const int32_t* rowptr_b,
                       int64_t num_rows_x, int32_t seg_stride_b, const int32_t* seg_b,
                       const float* w_b, const float* out, const float* gsel, int64_t ldo,
                       float* gX, int64_t ldgx, void* stream);

/* elementwise epilogue gradient: gin = gout * act'(.) evaluated from the activation OUTPUT y
 * (valid for LINEAR/TANH/RELU/LEAKY_RELU/ELU/SELU).  n = number of floats. */
int relgnn_act_bwd_from_output(int32_t act, const float* y, const float* gout, int64_t n,
                               float* gin, void* stream);

/* ========================================================================== *
 * 3. GNN-FiLM fused message kernels  (gnns/gnn_film.py:86-116)
 * ========================================================================== */

/*
 *   out[v,:] = AGG_l AGG_{p in (v,l)}  act( gamma[v,l,:] * (w[p] * T[col[p],:]) + beta[v,l,:] )
 * with [gamma | beta] = film[v*L + l, 0:2D]  (film = H @ F_l on nodes, gnn_film.py:102-106).
 * Replaces gnn_film.py:92-116: gather (:92), degree scale (:96-100), FiLM-weight gather (:103-106),
 * modulate (:108), concat (:111), activation (:112), segment reduce (:113-116).  The two Dense
 * layers (:94, :102) run node-side in the caller (one GEMM each).
 *   T    : [num_nodes*L, ldt]    per-node per-type messages h_u W_l, row = src*L + l
 *   film : [num_nodes*L, ldf >= 2D]
 *   rowptr/col/w : by-(target,type) plan (relgnn_segment_plan over key_by_target)
 * D % 4 == 0, D <= 1024, 16-byte aligned rows (RELGNN_EUNSUPPORTED otherwise).
 */
int relgnn_film_fwd(int32_t mode, int32_t act, const float* T, int64_t ldt, const float* film,
                    int64_t ldf, int32_t D, const int32_t* rowptr, int32_t num_nodes,
                    int32_t num_edge_types, const int32_t* col, const float* w, float* out,
                    int64_t ldo, const int32_t* bucket_row, void* stream);
/* COMPACT ROW TABLES (bucket_row / bucket_row_b, nullable): graphs with many edge types leave most (node,type)
 * buckets empty (VarMisuse-shaped: 23 types, 2/3 empty), so the caller may keep T / film / gfilm / gT rows only for
 * non-empty buckets, in any numbering.  bucket_row[v*L+l] is then the film (gfilm) row of bucket (v,l) — read only
 * for non-empty buckets, and only those gfilm rows are written; `col` already holds arbitrary T row ids;
 * bucket_row_b[r] is the T / gT row of by-source bucket r (rowptr_b then spans all num_nodes*L buckets).
 * NULL = the dense numbering row = node*L + type.
 *
 * backward, pass A (by-target plan): with g_p = gagg[v,:] * act'(pre_p),
 *   gfilm[v*L+l, 0:D]  = sum_p g_p * (w[p] T[col[p]])      (d gamma)
 *   gfilm[v*L+l, D:2D] = sum_p g_p                          (d beta)
 * every (v,l) row is written.  gagg = d loss / d (un-finalised aggregate), i.e. the caller has
 * already folded the mean / sqrt_n factor in.  Sum-like modes only (max: use the unfused path). */
int relgnn_film_bwd_film(int32_t act, const float* T, int64_t ldt, const float* film, int64_t ldf,
                         int32_t D, const int32_t* rowptr, int32_t num_nodes,
                         int32_t num_edge_types, const int32_t* col, const float* w,
                         const float* gagg, int64_t ldg, float* gfilm, int64_t ldgf,
                         const int32_t* bucket_row, float* dmsg, void* sign_mask, void* stream);
/* dmsg (nullable, [num_messages, D] contiguous, by-target order): pass A also writes every message's gradient
 * w.r.t. its gathered row, dmsg[p] = w[p] * gamma * g_p (pair kernels: w[p] * g_p).  gT is then ONE plain
 * gather-reduce of dmsg over the by-source buckets (relgnn_seg_reduce_fwd with col = by-target position of each
 * by-source message) and pass B below, which re-gathers the 8*D-byte film row per MESSAGE, is not needed. */
/* backward, pass B (by-(source,type) plan; row r of T owns its outgoing messages q):
 *   gT[r,:] = sum_q w_b[q] * gamma[frow_b[q],:] * g_q,
 *   g_q = gagg[tgt_b[q],:] * act'(gamma[frow_b[q]] * (w_b[q] * T[r,:]) + beta[frow_b[q]]) */
int relgnn_film_bwd_msg(int32_t act, const float* T, int64_t ldt, const float* film, int64_t ldf,
                        int32_t D, const int32_t* rowptr_b, int64_t num_rows_t,
                        const int32_t* tgt_b, const int32_t* frow_b, const float* w_b,
                        const float* gagg, int64_t ldg, float* gT, int64_t ldgt,
                        const int32_t* bucket_row_b, void* stream);
/* sign_mask (nullable; [num_messages, 4] 64-bit words, by-target order; dense tables, 128 < D <= 256 only, else
 * RELGNN_EUNSUPPORTED): pass A also writes the sign of every message's pre-activation, one bit per feature (word j, bit i =
 * feature 4*i + j).  For the piecewise-linear activations (linear, ReLU, leaky ReLU) that is all pass B needs of a message:
 * relgnn_film_bwd_msg_masked = pass B without re-gathering beta and without the pre-activation,
 *   gT[r,:] = sum_q w_b[q] * gamma[frow_b[q],:] * gagg[tgt_b[q],:] * (bit ? 1 : negative slope),  bit from
 *   sign_mask[pos_b[q]] (pos_b[q] = by-target position of by-source message q).  2 KiB instead of 3 KiB gathered per
 * message at D = 256. */
int relgnn_film_bwd_msg_masked(int32_t act, const float* film, int64_t ldf, int32_t D, const int32_t* rowptr_b,
                               int64_t num_rows_t, const int32_t* tgt_b, const int32_t* frow_b, const float* w_b,
                               const int32_t* pos_b, const void* sign_mask, const float* gagg, int64_t ldg, float* gT,
                               int64_t ldgt, void* stream);

/* ========================================================================== *
 * 4. RGAT segmented-softmax attention  (gnns/rgat.py:86-138)
 * ========================================================================== */

/*
 * For target v, head k (K = num_heads, Dh = D/K, Dh % 4 == 0), messages p of v over ALL edge types:
 *   z[p,k]   = s_src[col[p], k] + s_tgt[v*L + l(p), k]
 *   e[p,k]   = leaky_relu_{slope}(z[p,k])                                (rgat.py:112-115)
 *   a[p,k]   = exp( (e - max_p e) - log(sum_p exp(e - max_p e)) )        (rgat.py:126-130,
 *              dpu_utils.tfutils.unsorted_segment_log_softmax + tf.exp)
 *   out[v, k*Dh:(k+1)*Dh] = sum_p a[p,k] * T[col[p], k*Dh:(k+1)*Dh]     (rgat.py:131-136)
 * s_src[r,k] = <T[r, head k], a_l[k, 0:Dh]> and s_tgt[r,k] = <T[r, head k], a_l[k, Dh:2Dh]> for
 * r = node*L + l are [num_nodes*L, K] tables computed node-side by the caller (they replace the
 * [E, K, 2Dh] concat + einsum of rgat.py:106-115).  alpha (nullable, [num_messages, K], by-target
 * order) is saved for the backward.
 */
int relgnn_rgat_fwd(const float* T, int64_t ldt, int32_t D, int32_t num_heads, const float* s_src,
                    const float* s_tgt, const int32_t* rowptr, int32_t num_nodes,
                    int32_t num_edge_types, const int32_t* col, float slope, float* out,
                    int64_t ldo, float* alpha, void* stream);
/* backward pass A (by-target plan):
 *   dz[p,k] = a[p,k] * (<gout_vk, T[col[p]]_k> - <gout_vk, out_vk>) * lrelu'(z[p,k])    ([M,K], by-target order)
 *   gs_tgt[v*L+l, k] = sum_{p in (v,l)} dz[p,k]                                            (every row written) */
int relgnn_rgat_bwd_logits(const float* T, int64_t ldt, int32_t D, int32_t num_heads,
                           const float* s_src, const float* s_tgt, const int32_t* rowptr,
                           int32_t num_nodes, int32_t num_edge_types, const int32_t* col,
                           float slope, const float* alpha, const float* out, const float* gout,
                           int64_t ldo, float* dz, float* gs_tgt, void* stream);
/* backward pass B (by-(source,type) plan): pos_b[q] = by-target position of message q
 *   gT[r, head k] = sum_q alpha[pos_b[q],k] * gout[tgt_b[q], head k];   gs_src[r,k] = sum_q dz[pos_b[q],k] */
int relgnn_rgat_bwd_msg(int32_t D, int32_t num_heads, const int32_t* rowptr_b, int64_t num_rows_t,
                        const int32_t* tgt_b, const int32_t* pos_b, const float* alpha,
                        const float* dz, const float* gout, int64_t ldo, float* gT, int64_t ldgt,
                        float* gs_src, void* stream);

/*
 * Fast path of the same computation (rgat_fast.hip), used when num_heads in {1,2,4,8} and (D/num_heads) % 4 == 0;
 * RELGNN_EUNSUPPORTED otherwise (callers fall back to relgnn_rgat_fwd / _bwd_logits / _bwd_msg):
 *   relgnn_rgat_alpha    alpha[p,k] (the segmented softmax, rgat.py:112-130) with lanes across MESSAGES: the two
 *                        softmax passes read 4K bytes per message and never touch the gathered rows
 *   relgnn_headw_reduce  out[s, head h] = sum_{p in segment s} W[(wpos ? wpos[p] : p), h] * X[col[p], head h]
 *                        = rgat.py:131-136 with W = alpha (forward), and the gradient w.r.t. T on the transposed plan
 *                        (X = gout, col = tgt_b, wpos = pos_b).  Z / zsum (both or neither): a second [*, K] table indexed
 *                        like W; zsum[s, k] = sum over the segment of Z[(wpos ? wpos[p] : p), k] — the by-source sum of dz
 *                        (gradient of the per-source score table) rides along in the backward's gather
 *   relgnn_rgat_dz       dz[p,k] as in relgnn_rgat_bwd_logits (needs D <= 256 and D/num_heads/4 a power of two);
 *                        gs_tgt (optional, num_edge_types * num_heads <= 64, else RELGNN_EUNSUPPORTED): [V*L, K] sums of dz
 *                        over the (target, type) buckets, written by the same pass; without it gs_tgt / gs_src are plain
 *                        relgnn_seg_reduce_fwd calls over dz [M, K]
 */
int relgnn_rgat_alpha(const float* s_src, const float* s_tgt, int32_t num_heads, const int32_t* rowptr,
                      int32_t num_nodes, int32_t num_edge_types, const int32_t* col, float slope,
                      float* alpha, void* stream);
int relgnn_headw_reduce(const float* X, int64_t num_rows_x, int64_t ldx, int32_t D, int32_t num_heads,
                        const int32_t* rowptr, int64_t num_segments, int32_t seg_stride,
                        const int32_t* col, const float* W, const int32_t* wpos, float* out, int64_t ldo,
                        const float* Z, float* zsum, void* stream);
int relgnn_rgat_dz(const float* T, int64_t num_rows_t, int64_t ldt, int32_t D, int32_t num_heads,
                   const float* s_src, const float* s_tgt, const int32_t* rowptr, int32_t num_nodes,
                   int32_t num_edge_types, const int32_t* col, float slope, const float* alpha,
                   const float* out, const float* gout, int64_t ldo, float* dz, float* gs_tgt, void* stream);

/*
 * Attention-logit tables and their gradients (gnns/rgat.py:103-115: the [E, K, 2*Dh] concat + einsum with
 * `Edge_%i_Attention_Parameters` reshaped (K, 2*Dh), :110-111, restated on (node, type) rows):
 *   fwd : s_src[r,k] = <T[r, head k], att[l, k, 0:Dh]>,  s_tgt[r,k] = <T[r, head k], att[l, k, Dh:2Dh]>,  r = v*L + l
 *   bwd : gT[r, head k] += gs_src[r,k] * att[l,k,0:Dh] + gs_tgt[r,k] * att[l,k,Dh:2Dh]        (in place; gT nullable)
 *         att_partial[g, l, :] = sum over the nodes of group g of gs_*[r,k] * T[r, head k]  in att's (K, 2*Dh) layout;
 *         the caller column-sums the [num_groups, L*2*D] partials (relgnn_column_sum) into d att [L, 2*D].
 *         num_groups must equal relgnn_rgat_scores_groups(num_nodes).  att_partial nullable.
 * att: [L, 2*D] contiguous.  Supported geometry: D/4 in {8,16,32,64} and (D/4)/K a power of two
 * (RELGNN_EUNSUPPORTED otherwise: the caller falls back to library reductions).
 */
int64_t relgnn_rgat_scores_groups(int64_t num_nodes);
int relgnn_rgat_scores_fwd(const float* T, int64_t ldt, int32_t D, int32_t num_heads, const float* att,
                           int32_t num_edge_types, int64_t num_nodes, float* s_src, float* s_tgt, void* stream);
int relgnn_rgat_scores_bwd(const float* T, int64_t ldt, int32_t D, int32_t num_heads, const float* att,
                           int32_t num_edge_types, int64_t num_nodes, const float* gs_src, const float* gs_tgt,
                           float* gT, int64_t ldg, float* att_partial, int64_t num_groups, void* stream);

/* ========================================================================== *
 * 5. Messages from BOTH endpoint states  (gnns/gnn_edge_mlp.py:91-116, rgin.py:110-129,
 *    rgcn.py:91-104 use_both_source_and_target)
 * ========================================================================== */

/*
 * The first Dense layer on [h_u || h_v] splits into node-side P = H @ W[:D_in] (row src*L+l) and
 * Q = H @ W[D_in:] (row tgt*L+l).
 *   single Dense (no hidden layer), fused:  out[v,:] = AGG_p act( w[p] * (P[col[p],:] + Q[v*L + l(p),:]) )
 *   backward: relgnn_pair_bwd_q (by-target) gQ[v*L+l] = sum_p w[p]*g_p ; relgnn_pair_bwd_p (by-source)
 *   gP[r] = sum_q w_b[q]*g_q, g = gagg[target] * act'(w*(P+Q)).
 */
int relgnn_pair_fwd(int32_t mode, int32_t act, const float* P, int64_t ldp, const float* Q,
                    int64_t ldq, int32_t D, const int32_t* rowptr, int32_t num_nodes,
                    int32_t num_edge_types, const int32_t* col, const float* w, float* out,
                    int64_t ldo, void* stream);
int relgnn_pair_bwd_q(int32_t act, const float* P, int64_t ldp, const float* Q, int64_t ldq,
                      int32_t D, const int32_t* rowptr, int32_t num_nodes, int32_t num_edge_types,
                      const int32_t* col, const float* w, const float* gagg, int64_t ldg, float* gQ,
                      int64_t ldgq, float* dmsg /* nullable, see relgnn_film_bwd_film */, void* stream);
int relgnn_pair_bwd_p(int32_t act, const float* P, int64_t ldp, const float* Q, int64_t ldq,
                      int32_t D, const int32_t* rowptr_b, int64_t num_rows_p, const int32_t* tgt_b,
                      const int32_t* frow_b, const float* w_b, const float* gagg, int64_t ldg,
                      float* gP, int64_t ldgp, void* stream);
/*
 * Deeper edge MLPs (utils/utils.py:120-126): materialise the first layer's activations in the
 * ORIGINAL type-major message order so the caller can run the remaining per-type Dense layers on
 * contiguous [E_l, D] blocks (genuinely per-edge GEMMs):
 *   ghidden == NULL : out[m,:] = act( P[row_src[m],:] + (Q ? Q[row_tgt[m],:] : 0) )
 *   ghidden != NULL : out[m,:] = ghidden[m,:] * act'( P[row_src[m],:] + (Q ? Q[row_tgt[m],:] : 0) )
 * row_src = key_by_source, row_tgt = key_by_target of relgnn_relational_keys.
 */
int relgnn_pair_materialize(int32_t act, const float* P, int64_t ldp, const float* Q, int64_t ldq,
                            int32_t D, const int32_t* row_src, const int32_t* row_tgt,
                            int64_t num_messages, const float* ghidden, float* out, int64_t ldo,
                            void* stream);

/* ========================================================================== *
 * 6. Dense-layer helper (task-head plumbing around the path)
 * ========================================================================== */

/* X[rows[i], 0 .. cols) = value for the num_rows listed rows (int64 ids on the device; ld floats between rows): the few padding rows
 * of a compact pair table that the edge kernels leave unwritten and the typed weight-gradient product reads (gnns/gnn_film.py:92-106
 * over graph.PairTables) — what torch's index_fill_ did in 30 us per call. */
int relgnn_fill_rows_f32(float* X, int64_t ld, int32_t cols, const int64_t* rows, int64_t num_rows, float value, void* stream);
/*
 * out[c] = sum_r X[r, c]: the bias gradient of a Keras Dense over V node rows (tasks/ppi_task.py:176-179
 * backward), deterministic two-stage reduction.  workspace: relgnn_column_sum_workspace_bytes().
 */
size_t relgnn_column_sum_workspace_bytes(int64_t rows, int32_t cols);
int relgnn_column_sum(const float* X, int64_t rows, int32_t cols, int64_t ld, float* out,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ========================================================================== *
 * 7. Training-step plumbing around the path
 * ========================================================================== */

#define RELGNN_MT_MAX 48 /* tensors per multi-tensor launch */

/*
 * Per-variable gradient norms and the fused update of models/sparse_graph_model.py:227-260:
 *   g' = g * clip / max(||g||_2, clip)            (tf.clip_by_norm per variable, :253-260; clip <= 0: no clipping)
 *   m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2;  p -= lr_t * m / (sqrt(v) + eps)      (tf.train.AdamOptimizer,
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller) [TF-internal update rule]
 * h_* are HOST arrays of n <= RELGNN_MT_MAX device pointers / element counts; norms is a device [n] array.
 */
size_t relgnn_mt_l2norm_workspace_bytes(void);
int relgnn_mt_l2norm(const float* const* h_grads, const int64_t* h_sizes, int32_t n, float* norms,
                     void* workspace, size_t workspace_bytes, void* stream);
int relgnn_mt_adam_clip(float* const* h_params, const float* const* h_grads, float* const* h_m,
                        float* const* h_v, const int64_t* h_sizes, int32_t n, const float* norms,
                        float clip, float lr_t, float beta1, float beta2, float eps, void* stream);
/* The same update with the step size read from DEVICE memory, for a training step captured in a hipGraph (a host
 * scalar would be frozen into the graph): relgnn_adam_step_size advances d_state[0] (steps taken) by one and writes
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) to d_state[1]; relgnn_mt_adam_clip_devlr reads *d_lr_t (= d_state + 1). */
int relgnn_adam_step_size(float* d_state, float lr, float beta1, float beta2, void* stream);
int relgnn_mt_adam_clip_devlr(float* const* h_params, const float* const* h_grads, float* const* h_m,
                              float* const* h_v, const int64_t* h_sizes, int32_t n, const float* norms,
                              float clip, const float* d_lr_t, float beta1, float beta2, float eps, void* stream);

/*
 * PPI output head in one pass (tasks/ppi_task.py:181-191 + utils/utils.py:61-74):
 *   stats[0] = sum_i sigmoid_cross_entropy_with_logits(logits_i, labels_i)
 *   stats[1..3] = true_pos, false_pos, false_neg of round(sigmoid(logits)) vs int(labels);  stats[4] = micro-F1
 *   stats[5] = mean_scale * stats[0]   (the task's loss: mean_scale = 1 / num_nodes, ppi_task.py:189)
 * and the loss gradient  glogits = (g_mean[0] * mean_scale + g_total[0]) * (sigmoid(logits) - labels), g_mean / g_total =
 * device scalars holding the incoming gradients of stats[5] / stats[0] (either may be NULL = 0, not both).
 * n = number of elements; stats has 6 floats.
 */
size_t relgnn_sigmoid_ce_stats_workspace_bytes(void);
int relgnn_sigmoid_ce_stats(const float* logits, const float* labels, int64_t n, float mean_scale, float* stats,
                            void* workspace, size_t workspace_bytes, void* stream);
int relgnn_sigmoid_ce_bwd(const float* logits, const float* labels, int64_t n, const float* g_mean, float mean_scale,
                          const float* g_total, float* glogits, void* stream);
/* The same gradient written into rows of ldg >= cols floats (logits / labels [rows, cols] contiguous), the ldg - cols columns behind
 * every row set to zero: the gradient of tasks/ppi_task.py:183-189 laid out as the left operand of the head's input-gradient
 * MatMul with its reduction length (121 labels) filled up to a multiple of 16 — relgnn_limb_gemm_xf32 over ldg columns against
 * the weight limbs of relgnn_limb_split_multi_f32, which fills the last k-tile of a matrix with zeros. */
int relgnn_sigmoid_ce_bwd_padded(const float* logits, const float* labels, int64_t rows, int32_t cols, const float* g_mean,
                                 float mean_scale, const float* g_total, float* glogits, int32_t ldg, void* stream);

/* ========================================================================== *
 * 8. GRU cell elementwise halves (node-side; gnns/ggnn.py:92 via utils/utils.py:15-16)
 * ========================================================================== */

/*
 * Keras GRUCell, TF 1.13 [TF-internal]: reset_after=False, recurrent_activation=hard_sigmoid, gate order z, r, h.
 * With xk = x @ kernel + bias [V, 3u], rec = h @ recurrent_kernel[:, :2u] [V, 2u] (caller GEMMs):
 *   gates_fwd : z = hs(xk[:, :u] + rec[:, :u]), r = hs(xk[:, u:2u] + rec[:, u:]), rh = r * h
 *   out_fwd   : hh = act(xk[:, 2u:] + q) with q = rh @ recurrent_kernel[:, 2u:] (caller GEMM); out = z*h + (1-z)*hh
 *   out_bwd   : gxk[:, 2u:] = gq = gout*(1-z)*act'(hh); gz = gout*(h-hh); gh = gout*z
 *   gates_bwd : gxk[:, :u] = gz*hs'(z); gxk[:, u:2u] = (grh*h)*hs'(r); gh += grh*r       (grh = gq @ U_h^T, caller)
 * act in {LINEAR, TANH, RELU, LEAKY_RELU, ELU, SELU} (derivative from the output); all tensors contiguous.
 */
int relgnn_gru_gates_fwd(const float* xk, const float* rec, const float* h, int64_t num_nodes, int32_t units,
                         float* z, float* r, float* rh, void* stream);
int relgnn_gru_out_fwd(const float* xk, const float* q, const float* z, const float* h, int64_t num_nodes,
                       int32_t units, int32_t act, float* hh, float* out, void* stream);
int relgnn_gru_out_bwd(const float* gout, const float* z, const float* h, const float* hh, int64_t num_nodes,
                       int32_t units, int32_t act, float* gxk, float* gq, float* gz, float* gh, void* stream);
int relgnn_gru_gates_bwd(const float* grh, const float* gz, const float* z, const float* r, const float* h,
                         int64_t num_nodes, int32_t units, float* gxk, float* gh, void* stream);
/*
 * The whole cell forward as ONE kernel (csrc/gru_cell.hip), three-bf16-limb products on the matrix cores (exact-fp32 operands,
 * fp32 accumulation: section 12's limb route):
 *   [z | r] = hs([x | h] @ [[K_z K_r]; [U_z U_r]] + b[:2u]);  hh = act([x | r*h] @ [K_h; U_h] + b[2u:]);  out = z*h + (1-z)*hh
 * — x @ kernel + h @ recurrent_kernel as ONE accumulation over k = [x | h] (the composition above rounds the two sums separately:
 * results agree to the last few bits, not bit for bit).  A persistent workgroup per CU owns 64-row panels: producer waves split the
 * rows of x and h into limbs in LDS, matrix waves compute z and r, r*h goes to the candidate's k-loop through LDS, and nothing but
 * the results leaves the CU.
 *   w_zr_limbs: limb image (relgnn_limb_split_multi_f32) of the [2u, in_dim + u] right operand [K[:, :2u]; U[:, :2u]]^T
 *   w_h_limbs : limb image of the [u, in_dim + u] right operand [K[:, 2u:]; U[:, 2u:]]^T;   bias [3u]
 *   z, r, rh, hh: [num_nodes, u] dense outputs for the backward — all four or all NULL (an inference pass); out [num_nodes, u]
 *   status: the caller's hand-over block (RELGNN_HANDOVER_PC_*: section 12), may be NULL.
 * units == in_dim == 128, act in {LINEAR, TANH, RELU, LEAKY_RELU} (relgnn_gru_cell_fwd_supported), 16-byte aligned rows:
 * RELGNN_EUNSUPPORTED otherwise (the caller composes the cell from the entries above).
 */
/*
 * relgnn_gru_cell_bwd_xf32: the data path of the cell's backward as ONE kernel (what gru_out_bwd / gru_gates_bwd above and three
 * limb products did in seven launches), from the tensors the forward kept:
 *   gpre = g (1-z) act'(hh);  gzp = g (h-hh) hs'(z);  [gx_h | grh] = gpre @ [K_h; U_h]^T;  grp = grh h hs'(r);
 *   [gx | gh] = [gx_h | g z + grh r] + [gzp | grp] @ [[K_z K_r]; [U_z U_r]]^T;   gxk = [gzp | grp | gpre]
 * gxk [num_nodes, 3u] is the right operand of the cell's weight gradients (x^T gxk, h^T gxk[:, :2u], (r*h)^T gxk[:, 2u:], bias =
 * its column sums: relgnn_gemm_tn_stream_group_f32).  Producer waves do every elementwise step and hand limbs to the matrix waves
 * through LDS; grh and g z + grh r cross between the two roles in LDS as well.
 *   w_h_nt_limbs : limb image of [K[:, 2u:]; U[:, 2u:]] as the [in_dim + u, u] right operand (rows = x | h columns)
 *   w_zr_nt_limbs: limb image of [K[:, :2u]; U[:, :2u]] as the [in_dim + u, 2u] right operand
 *   z, r, hh, gxk, gx, gh dense; gout, h with row strides.  Same shapes, activations and status block as the forward entry.
 */
int relgnn_gru_cell_fwd_supported(int32_t act, int32_t units, int32_t in_dim);
int relgnn_gru_cell_fwd_xf32(const float* x, int64_t ldx, const float* h, int64_t ldh, const uint16_t* w_zr_limbs,
                             const uint16_t* w_h_limbs, const float* bias, int32_t act, float* z, float* r, float* rh, float* hh,
                             float* out, int64_t num_nodes, int32_t units, int32_t in_dim, int32_t* status, void* stream);
int relgnn_gru_cell_bwd_xf32(const float* gout, int64_t ldg, const float* z, const float* r, const float* h, int64_t ldh,
                             const float* hh, const uint16_t* w_h_nt_limbs, const uint16_t* w_zr_nt_limbs, int32_t act, float* gxk,
                             float* gx, float* gh, int64_t num_nodes, int32_t units, int32_t in_dim, int32_t* status, void* stream);

/* ---- layer normalisation of node states ------------------------------------------------------------------
 * Replaces: tf.contrib.layers.layer_norm at gnns/gnn_film.py:120, gnns/rgin.py:139, gnns/gnn_edge_mlp.py:120 and
 * models/sparse_graph_model.py:192-193: moments over the last axis (biased variance), y = (x-mean)*rsqrt(var+eps)*gamma+beta.
 *   fwd : Y [rows, D]; mean, rstd [rows] are saved for the backward.
 *   bwd : dX = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat));
 *         partial [num_groups, 2*D]: per lane-group sums of g*xhat (d gamma) and g (d beta); the caller column-sums
 *         them (relgnn_column_sum).  num_groups = relgnn_layer_norm_groups(rows, D).
 * D % 4 == 0, D <= 1024, 16-byte aligned rows (RELGNN_EUNSUPPORTED otherwise). */
int64_t relgnn_layer_norm_groups(int64_t rows, int32_t D);
int relgnn_layer_norm_fwd(const float* X, int64_t ldx, int64_t rows, int32_t D, const float* gamma, const float* beta,
                          float eps, float* Y, int64_t ldy, float* mean, float* rstd, void* stream);
int relgnn_layer_norm_bwd(const float* X, int64_t ldx, const float* gY, int64_t ldg, int64_t rows, int32_t D,
                          const float* gamma, const float* mean, const float* rstd, float* dX, int64_t ldd,
                          float* partial, int64_t num_groups, void* stream);

/* ========================================================================== *
 * 9. Host-side disjoint-union batch builder (every pointer here is a HOST pointer)
 * ========================================================================== */

/*
 * Replaces: the numpy batching loops of tasks/ppi_task.py:209-256 and
 * tasks/qm9_task.py:212-261 (input contract tasks/sparse_graph_task.py:139-149).
 *
 * The dataset is flattened once by the caller into a STORE of G graphs:
 *   h_node_off     [G+1] int64   node range of graph g in the flat per-node arrays
 *   h_edge_off[l]  [G+1] int64   edge range of graph g in h_adj[l]
 *   h_adj[l]       [E_l_total, 2] int32, {source, target}, node ids LOCAL to their graph
 *   h_deg[l]       [N_total] float32  per-graph in-degree tables (type_to_node_to_num_incoming_edges)
 *   h_payload[p]   [N_total, row_bytes_p] any per-node rows (features, labels, ...)
 * A batch is the list h_graph_ids[0..n_graphs) taken in that order.
 *
 * relgnn_batch_count : how many ids from `first` on fit one batch under the reference's rule
 *                      `node_offset + |V_g| < max_nodes` (strict, ppi_task.py:220).  0 = the first graph
 *                      never fits (the reference would spin forever); -1 = bad argument.
 * relgnn_batch_layout: fills h_layout[relgnn_batch_layout_len(L, n_payloads)] with
 *                      {V, M, arena_bytes, off_deg, off_node_to_graph, off_payload[p]..., off_adj[l]..., E_l...};
 *                      offsets are bytes into the arena, 256-byte aligned.
 * relgnn_batch_pack  : writes the arena (pinned host memory for an async copy):
 *                        payload p      [V, row_bytes_p]
 *                        deg            [L, V] float32           (concatenated on axis 1, ppi_task.py:237)
 *                        node_to_graph  [V] int32                (graph_nodes_list, qm9_task.py:238)
 *                        adj l          [E_l, 2] int32 = local ids + node offset of the graph (ppi_task.py:228);
 *                                       E_l = 0 gives the empty list of ppi_task.py:248-249
 *                      with `num_threads` host threads (>= 1).  RELGNN_EUNSUPPORTED if V or M >= 2^31-1.
 */
int64_t relgnn_batch_layout_len(int32_t num_types, int32_t n_payloads);
int64_t relgnn_batch_count(const int64_t* h_node_off, const int64_t* h_graph_ids, int64_t n_ids, int64_t first,
                           int64_t max_nodes);
int relgnn_batch_layout(int32_t num_types, int64_t n_graphs, const int64_t* h_graph_ids, const int64_t* h_node_off,
                        const int64_t* const* h_edge_off, int32_t n_payloads, const int64_t* h_payload_row_bytes,
                        int64_t* h_layout);
int relgnn_batch_pack(int32_t num_types, int64_t n_graphs, const int64_t* h_graph_ids, const int64_t* h_node_off,
                      const int64_t* const* h_edge_off, const int32_t* const* h_adj, const float* const* h_deg,
                      int32_t n_payloads, const void* const* h_payload, const int64_t* h_payload_row_bytes,
                      const int64_t* h_layout, void* h_arena, size_t arena_bytes, int32_t num_threads);

/* ========================================================================== *
 * 10. Batch bucketing from dataset-resident plans
 * ========================================================================== */

/*
 * The (node,type)-bucketed orders of a disjoint-union batch are graph-major, so they are the per-graph slices of the
 * orders of ANY other union that holds the same graphs.  Bucket the whole data fold once as one union
 * (relgnn_relational_keys_all + relgnn_relational_plan, arrays suffixed _d, kept in HBM), then a batch = the graph
 * list ids[0..K) needs no sort: this call re-bases the slices (node ids, positions, type-major message numbering of
 * the batch) with three streaming kernels and produces exactly the arrays relgnn_relational_plan would
 * (bit-identical: same stable order).  All pointers are DEVICE pointers.
 *   ids          [K]        dataset graph id per batch slot (graphs may repeat)
 *   node_off_b   [K+1]      node offset of slot k in the batch            (tasks/ppi_task.py:228)
 *   msg_off_b    [K+1]      messages of the slots before k (all types)
 *   edge_off_b   [L][K+1]   messages of type l in the slots before k
 *   type_off_b   [L+1]      start of type l in the batch's type-major message list (gnns/rgcn.py:78)
 *   *_d                     the same four tables for the dataset union (G = num_dataset_graphs)
 * Optional (NULL to skip), fused into the same passes: src_t [M] = source NODE per by-target position; w_t / w_s [M] =
 * the per-message scales of the batch in by-target / by-source order, copied from the dataset-level arrays w_t_d / w_s_d
 * (1/(in-degree + 1e-7), gnns/rgcn.py:100-104, is a property of the graph, not of the batch).
 * LEAN call: perm_t, col_t, inv_perm_t, perm_s, frow_s, pos_t_of_s all NULL (all six or none).  Then only the row
 * pointers, src_t, tgt_s and the scales are produced — everything the fused gather kernels of the sum / mean / sqrt_n
 * layers read — in two passes that move 16 bytes per message; the six arrays (permutations to and from the reference's
 * type-major message order, the scattered inverse, (node, type) rows: what the pair / attention / materialised-message
 * paths read) can be produced later by a second, full call with the same tables.
 */
/*
 * The batch's TENSORS from the fold's flat device arrays, same packing rules as relgnn_batch_pack (replaces the numpy
 * batching loops of tasks/ppi_task.py:209-256 when the fold lives in HBM): node payload rows concatenated on the node
 * axis (:222-225,240-243), degree tables on axis 1 (:237), adjacency lists + node offset of the slot (:228), graph
 * slot per node (:226).  Tables as for relgnn_plan_assemble (msg_off_* not needed).
 *   h_payload_d / h_payload_b   HOST arrays of n_payloads DEVICE pointers: fold rows [N, cols] -> batch rows [V, cols]
 *                               of 4-byte elements (h_payload_cols: HOST array)
 *   deg_d [L, N] -> deg_b [L, V];  adj_d [M_fold, 2] type-major, graph-local ids -> adj_b [M, 2] type-major
 *   node_to_graph [V] int32
 * adj_b may be NULL: the batch's adjacency lists are then not produced (a caller that hands the bucketing, not the lists,
 * to the layers asks for them only when something reads them).
 */
int relgnn_batch_gather(const int64_t* ids, int32_t num_batch_graphs, int32_t num_edge_types, int64_t num_dataset_graphs,
                        const int64_t* node_off_b, const int64_t* edge_off_b, const int64_t* type_off_b,
                        const int64_t* node_off_d, const int64_t* edge_off_d, const int64_t* type_off_d, int64_t num_nodes,
                        int64_t num_messages, int64_t num_dataset_nodes, int32_t n_payloads, const void* const* h_payload_d,
                        const int32_t* h_payload_cols, void* const* h_payload_b, const float* deg_d, float* deg_b,
                        const int32_t* adj_d, int32_t* adj_b, int32_t* node_to_graph, void* stream);
int relgnn_plan_assemble(const int64_t* ids, int32_t num_batch_graphs, int32_t num_edge_types, int64_t num_dataset_graphs,
                         const int64_t* node_off_b, const int64_t* msg_off_b, const int64_t* edge_off_b,
                         const int64_t* type_off_b, const int64_t* node_off_d, const int64_t* msg_off_d,
                         const int64_t* edge_off_d, const int64_t* type_off_d, int64_t num_nodes, int64_t num_messages,
                         const int32_t* rowptr_t_d, const int32_t* perm_t_d, const int32_t* col_t_d,
                         const int32_t* rowptr_s_d, const int32_t* perm_s_d, const int32_t* frow_s_d,
                         const int32_t* pos_t_of_s_d, int32_t* rowptr_t, int32_t* perm_t, int32_t* col_t,
                         int32_t* inv_perm_t, int32_t* rowptr_s, int32_t* perm_s, int32_t* frow_s, int32_t* tgt_s,
                         int32_t* pos_t_of_s, const float* w_t_d, const float* w_s_d, int32_t* src_t, float* w_t, float* w_s,
                         void* stream);

/* ========================================================================== *
 * 12. Node-side Dense layers on the matrix cores, exact fp32
 * ========================================================================== */

/*
 * Replaces the MatMul of every bias-free / biased Keras Dense on the path when it is evaluated node-side: the
 * per-edge-type transforms (gnns/rgcn.py:70-74,98; ggnn.py:63-67,80-81; rgat.py:95-96; gnn_film.py:92-94,102),
 * the inter-layer Dense (models/sparse_graph_model.py:194-200) and the MatMul gradients TF derives for them.
 * The products below are exact fp32 (v_mfma_f32_32x32x2_f32 in relgnn_panel_gemm_f32 / relgnn_gemm_tn_stream_f32, fp32
 * solutions of the library in relgnn_blaslt_gemm_f32): f32 operands, f32 accumulation, no reduced precision.  Layouts:
 *   RELGNN_GEMM_NN  C[M,N] = act(A[M,K] @ B[K,N] + bias)      A, B row-major
 *   RELGNN_GEMM_NT  C[M,N] = A[M,K] @ Bt[N,K]^T               Bt row-major (dX = G @ W^T with Bt = W)
 *   RELGNN_GEMM_TN  C[M,N] = At[K,M]^T @ B[K,N]               At row-major (dW = X^T @ G, K = node dimension)
 */
#define RELGNN_GEMM_NN 0
#define RELGNN_GEMM_NT 1
#define RELGNN_GEMM_TN 2
/*
 * The same three products as PLAIN LIBRARY GEMMs (hipBLASLt), for the layers that are not fused: what torch.mm runs, minus
 * the per-call solution lookup.  Every batch of a shuffled epoch has its own node count, so every call is a shape the
 * library has not seen, and its heuristic costs ~70 us of host time per new shape (21 calls per C2 training step: the
 * host, not the GPU, bounded the step).  The lookup is done once per (layout, N, K, batch, M / 4096, bias, accumulate) and
 * the solution is reused for every M of that bucket.
 *   batch > 1: strided batched product, operand / result z at A + z*stride_a, B + z*stride_b, C + z*stride_c (elements)
 *   bias (NN-style epilogue, length N, batch == 1 only) is added to every row;  accumulate != 0: C += product
 *   act: RELGNN_ACT_LINEAR or RELGNN_ACT_RELU (the library's ReLU epilogue, after the bias; batch == 1, no accumulate)
 *   workspace: caller-allocated device scratch handed to the library (may be NULL with workspace_bytes 0)
 * Not re-entrant across streams for ONE process-wide handle: calls are serialised by a mutex (one rank per process).
 */
int relgnn_blaslt_gemm_f32(int32_t layout, int32_t act, const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                           float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t batch, int64_t stride_a,
                           int64_t stride_b, int64_t stride_c, int32_t accumulate, void* workspace, int64_t workspace_bytes,
                           void* stream);

/*
 * Weight gradient of a node-side Dense layer: C[M, N] (+)= A[K, M]^T @ B[K, N], K = the node dimension (3e4 .. 1e6), M, N
 * small (the MatMul gradient TF derives for models/sparse_graph_model.py:165-172,194-200, tasks/ppi_task.py:176-179,
 * gnns/rgcn.py:70-74).  Streaming kernel: every wave owns one 64 x 64 output tile for one chunk of rows and feeds
 * v_mfma_f32_32x32x2_f32 straight from global memory (the reduction index is the row of both row-major operands, which is
 * the MFMA operand layout): no LDS, no barriers; partial products per chunk go to `workspace`
 * (relgnn_gemm_tn_stream_workspace_bytes) and are summed in chunk order (deterministic).  Exact fp32.  Any M, N, lda >= M,
 * ldb >= N; 8-byte loads when a row stride is even and its base 8-byte aligned, 4-byte loads otherwise.
 */
int64_t relgnn_gemm_tn_stream_workspace_bytes(int32_t M, int32_t N, int64_t K);
int relgnn_gemm_tn_stream_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N,
                              int64_t K, int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream);
/*
 * The same product with the output as N / block_cols matrices of their own: column block j of A^T @ B goes to the [M, block_cols]
 * matrix at C + j * block_stride (row stride ldc >= block_cols).  The gradients of the L per-edge-type kernels a layer applies to
 * the SAME node states (gnns/ggnn.py:63-67,80-81: dW_l = H^T @ G[:, l * D : (l + 1) * D]) are ONE product whose operand A is read
 * once, each gradient a dense tensor of its own (block_stride = M * block_cols: a contiguous [L, M, block_cols] array).  Same
 * workspace size, chunking and summation order as relgnn_gemm_tn_stream_f32 for (M, N, K).  N % block_cols != 0: RELGNN_EINVAL.
 */
int relgnn_gemm_tn_stream_blocks_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                                     int64_t block_stride, int32_t M, int32_t N, int32_t block_cols, int64_t K, int32_t accumulate,
                                     void* workspace, int64_t workspace_bytes, void* stream);
/*
 * Up to four such products over the SAME K rows in one launch pair (one list of 64 x 64 tiles cut into the same row chunks, one
 * reduction launch): C[i][M[i], N[i]] = A[i][K, M[i]]^T @ B[i][K, N[i]] — the three weight gradients of a GRU cell (gnns/ggnn.py:92
 * through utils/utils.py:15-16: x^T gxk, h^T gxk[:, :2u], (r*h)^T gxk[:, 2u:]).  colsum0 (nullable, [N[0]]): the column sums of
 * B[0] — the bias gradient of the layer whose kernel gradient product 0 is — from the registers the matrix instructions read.
 * The pointer and size arrays are HOST arrays of `num` entries.  8-byte aligned operands with even row strides and M, N multiples
 * of 64 (RELGNN_EUNSUPPORTED otherwise: launch such products one by one); workspace: relgnn_gemm_tn_stream_group_workspace_bytes.
 */
int64_t relgnn_gemm_tn_stream_group_workspace_bytes(int32_t num, const int32_t* M, const int32_t* N, int64_t K, int32_t with_colsum);
int relgnn_gemm_tn_stream_group_f32(int32_t num, const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
                                    float* const* C, const int64_t* ldc, const int32_t* M, const int32_t* N, int64_t K,
                                    float* colsum0, void* workspace, int64_t workspace_bytes, void* stream);
/*
 * The closing pass of a split-K weight gradient: C[M, N] = sum over `num_slabs` partial products slabs[z] (each [M, N],
 * contiguous, summed in slab order) + At[R, M]^T @ Bt[R, N] for the R rows (R < one chunk, typically < 64) that the equal
 * chunks left over.  One launch instead of a sum, a second product and an accumulate.
 */
int relgnn_sum_slabs_tail_f32(const float* slabs, int32_t num_slabs, int32_t M, int32_t N, const float* At, int64_t lda,
                              const float* Bt, int64_t ldb, int32_t R, float* C, void* stream);

/*
 * relgnn_panel_gemm_f32 — the same three products on v_mfma_f32_32x32x2_f32 (+ 16x16x4 for a 16-row strip; exact fp32) with direct-to-LDS operand
 * staging, sized per call so that every CU gets the same number of equal row panels, plus what the node-side Dense
 * layers of MANY-TYPE graphs need and a library GEMM cannot express without copies:
 *   a_rows (nullable)  NN / NT: row r of the left operand is A[a_rows[r], :] (a_rows[r] < 0: a row of zeros) — the
 *                      tf.nn.embedding_lookup of gnns/gnn_film.py:92-93,105 / rgcn.py:87-89 fused into the product;
 *                      TN: the REDUCTION row k of both operands is (A[a_rows[k], :], B[k, :]) (K <= 1024 per product;
 *                      with independent batches product z uses a_rows[z*K + k])
 *   b_select (nullable) rows [p*rows_per_select, (p+1)*rows_per_select) of the output use the right operand
 *                      B + b_select[p]*b_select_stride: one Edge_%i_Weight / Edge_%i_FiLM_Computations kernel per
 *                      512-row tile of a compact (node, type) table (gnn_film.py:74-78,92-106); rows_per_select % 128 == 0
 *   batch > 1          split_k_rows == 0: independent products z at A + z*a_batch_stride, B + z*b_batch_stride,
 *                      C + z*c_batch_stride (per-tile weight-gradient partials);
 *                      split_k_rows  > 0: ONE product whose K range is cut into `batch` chunks of split_k_rows rows
 *                      (% 16 == 0), chunk z writes its partial product to C + z*c_batch_stride (the caller sums the slabs
 *                      in order: deterministic)
 *   bias / act         NN-style epilogue (bias of length N, any RELGNN_ACT_*), batch == 1 semantics per product
 *   zeros              256 zero floats in device memory (the source of padding rows and of the K tail)
 * Requirements (RELGNN_EUNSUPPORTED otherwise): N % 64 == 0, 16-byte aligned operands and row strides, NN / NT: K % 4 == 0,
 * TN: M % 4 == 0.  C rows [0, M) x columns [0, N) are written, nothing else.
 */
int relgnn_panel_gemm_zeros_floats(void);
int relgnn_panel_gemm_f32(int32_t layout, int32_t act, const float* A, int64_t lda, const int32_t* a_rows, const float* B,
                          int64_t ldb, const int32_t* b_select, int32_t rows_per_select, int64_t b_select_stride,
                          const float* bias, const float* zeros, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                          int32_t batch, int64_t a_batch_stride, int64_t b_batch_stride, int64_t c_batch_stride,
                          int32_t split_k_rows, void* stream);

/*
 * relgnn_limb_gemm_f32 / relgnn_limb_split_f32 — the same Dense products (tf.layers.dense / tf.matmul of gnns/rgcn.py:96-98 and
 * their input gradient) at fp32 accuracy on the bf16 matrix pipe: every fp32 operand is carried as THREE bf16 limbs
 * (x = hi + mid + lo exactly: 3 x 8 significant bits) and a product keeps the six limb products of weight >= 2^-16, each
 * exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the dropped ones are < 2^-23 of |x w| (csrc/limb_gemm.hip).
 * gfx950 has no xf32 / TF32 form; its fp32-input MFMA runs at 1/16 of the bf16 rate.
 *   limb format   "limb tiles": an [R, C] matrix (C % 16 == 0) as ceil(R / 32) x C / 16 tiles of 32 rows x 16 columns in
 *                 row-major tile order; a tile = three 1 KiB blocks (hi, mid, lo); a block holds element (i, k) at bf16 index
 *                 (k / 8) * 256 + i * 8 + k % 8 (the LDS image and MFMA operand layout: one DMA instruction per block); rows
 *                 past R are zeros.  relgnn_limb_elements(R, C) = bf16 elements of the whole thing
 *   split         X fp32 [rows, cols] (ldx) -> limb tiles of X (transpose == 0) or of X^T (transpose != 0)
 *   gemm          C[m][n] = act(bias[n] + sum_k A[m][k] * B[n][k]): A = limb tiles of an [M, K] matrix, B = limb tiles of an
 *                 [N, K] matrix, C fp32 [M, N] (ldc); bias nullable; zeros = 1 KiB of zero bytes in device memory
 *                 (relgnn_panel_gemm_zeros_floats floats)
 *   gemm_xf32     the same product with the LEFT operand given as plain fp32 [M, K] (lda) and split inside the kernel on its way
 *                 into LDS — the form the path uses: activations / gradients stay fp32 in HBM (what their producers write and
 *                 what the weight-gradient product reads), only the weights (a few hundred KB, split once per step) exist as limbs
 * Requirements (RELGNN_EUNSUPPORTED otherwise): K % 16 == 0, N % 256 == 0, 16-byte aligned bases.
 */
int64_t relgnn_limb_elements(int64_t rows, int64_t cols);
int relgnn_limb_split_f32(const float* X, int64_t ldx, int32_t rows, int32_t cols, int32_t transpose, uint16_t* out, void* stream);
/* `batch` matrices at X + i * x_batch_stride -> limb tiles at out + i * relgnn_limb_elements(R, C) (one launch) */
int relgnn_limb_split_batch_f32(const float* X, int64_t ldx, int64_t x_batch_stride, int32_t rows, int32_t cols, int32_t transpose,
                                int32_t batch, uint16_t* out, void* stream);
/* n matrices of different shapes in ONE launch (host arrays of n entries each): item d writes X[d] (or its transpose) as k-tiles
 * kt_offset[d] .. kt_offset[d] + C/16 - 1 of a limb matrix with kt_total[d] k-tiles per 32-row block at out[d] (kt_offset 0 and
 * kt_total = C/16: a whole matrix, as relgnn_limb_split_f32; several items with one `out` lay matrices side by side along k, e.g. the
 * stacked right operand [W_0 | W_1 | ..] of the input gradient of gnns/rgcn.py:96-98 without forming it in fp32).  The weight operands
 * of a training step change once per step (the optimizer's update): the host side splits them all behind it with this entry.
 * C (the k extent of an item) need not be a multiple of 16: the item then takes ceil(C / 16) k-tiles and the last one is filled up
 * with zeros (the 121-label head: K = 128 against a left operand whose rows are zero-padded alike). */
int relgnn_limb_split_multi_f32(int32_t n, const float* const* X, const int64_t* ldx, const int32_t* rows, const int32_t* cols,
                                const int32_t* transpose, uint16_t* const* out, const int32_t* kt_offset, const int32_t* kt_total,
                                void* stream);
int relgnn_limb_gemm_f32(int32_t act, const uint16_t* A, const uint16_t* B, const float* bias, const void* zeros, float* C,
                         int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream);
int relgnn_limb_gemm_xf32(int32_t act, const float* A, int64_t lda, const uint16_t* B, const float* bias, const void* zeros,
                          float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream);
/* The same products from TWO fp16 limbs per operand behind exact power-of-two scales: x * 2^j = hi + lo (+ a remainder <= 2^-22 of
 * it), three MFMA products (hi hi, hi lo, lo hi: each exact in fp32) instead of the six of the bf16 triple.  fp16's exponent range is
 * what the scales are for:
 *   weights   relgnn_limb16_split_multi_f32 writes n matrices into n_images images (image[d]: the image matrix d belongs to — the
 *             matrices of one product; laid out like relgnn_limb_split_multi_f32's, two 1 KiB blocks per tile:
 *             relgnn_limb16_elements), each image scaled by the power of two that puts ITS largest magnitude into [2^14, 2^15);
 *             the magnitudes are left in wmax[0 .. n_images) (device memory; the product is handed wmax + its image's index)
 *   left rows relgnn_limb16_gemm_xf32 scales row m by the power of two of max_g xmax[m * xgroups + g] — per-row magnitudes the
 *             PRODUCER of A wrote (relgnn_seg_reduce_fwd_rowmax: one value per (node, type) bucket = xgroups per row of the
 *             [V, L * D] operand of gnns/rgcn.py:96-98 in the aggregate-first order); a row's scale factors out of its output row,
 *             the epilogue removes it together with the weights' (both exact)
 * Same shape requirements as relgnn_limb_gemm_xf32.  Not an exact split like the bf16 triple: the representation error is
 * <= 2^-22 of an element, of the order of one fp32 rounding (tests: against float64 next to the exact-fp32 product). */
int64_t relgnn_limb16_elements(int64_t rows, int64_t cols);
int relgnn_limb16_split_multi_f32(int32_t n, const float* const* X, const int64_t* ldx, const int32_t* rows, const int32_t* cols,
                                  const int32_t* transpose, uint16_t* const* out, const int32_t* kt_offset, const int32_t* kt_total,
                                  const int32_t* image, int32_t n_images, float* wmax, void* stream);
int relgnn_limb16_gemm_xf32(int32_t act, const float* A, int64_t lda, const float* xmax, int32_t xgroups, const uint16_t* B,
                            const float* wmax, const float* bias, const void* zeros, float* C, int64_t ldc, int32_t M, int32_t N,
                            int32_t K, void* stream);
/* The weight gradient from two fp16 limbs: relgnn_limb_gemm_tn_f32's partial products behind exact power-of-two scales derived from
 * magnitudes in device memory.  Column j of A is scaled from amax[j / a_cols_per_scale], column c of G from
 * gmax[c / g_cols_per_scale]: 1 = one magnitude per COLUMN (relgnn_col_absmax_f32) — the reduction runs over the rows of both
 * operands, so a row's scale would not factor out, a column's does (out[j][c] carries sa[j] * sg[c]) and a column of small
 * gradients keeps its relative precision next to a column of large ones; J (resp. C) = one magnitude for the operand
 * (relgnn_absmax_f32): normwise accuracy only; anything between = per group of consecutive columns.  Magnitudes must bound the
 * finite elements they cover; relgnn_absmax_f32 / relgnn_col_absmax_f32 skip inf / NaN elements, which then spoil exactly the
 * sums they take part in, as in fp32.
 * relgnn_absmax_f32: out[0] = max |x[i]| over the finite x[i] (x 16-byte aligned).
 * relgnn_col_absmax_f32: out[c] = max_r |X[r][c]| over the finite elements.  cols <= 16: any layout (the [V, L] bucket magnitudes
 * of the gather -> one magnitude per edge type), no workspace.  cols > 16: cols % 4 == 0, ldx % 4 == 0, X 16-byte aligned, and a
 * 16-byte aligned workspace of relgnn_col_absmax_workspace_bytes(rows, cols) bytes (per-workgroup partial maxima: two stages, no
 * atomics). */
int relgnn_limb16_gemm_tn_f32(const float* A, int64_t lda, const float* G, int64_t ldg, const float* amax, int32_t a_cols_per_scale,
                              const float* gmax, int32_t g_cols_per_scale, float* P, int32_t V, int32_t J, int32_t C, void* stream);
int64_t relgnn_col_absmax_workspace_bytes(int32_t rows, int32_t cols);
int relgnn_col_absmax_f32(const float* X, int64_t ldx, int32_t rows, int32_t cols, float* out, void* workspace,
                          int64_t workspace_bytes, void* stream);
int relgnn_absmax_f32(const float* x, int64_t n, float* out, void* stream);
/* Input-gradient products with the activation gradient of the layer below in the epilogue:
 *     C = act(bias + A @ B^T) * dact'(Y)
 * dact'(Y) = the derivative of activation `dact` (RELGNN_ACT_TANH / RELU / LEAKY_RELU / ELU / SELU; GELU: RELGNN_EUNSUPPORTED, it
 * needs the pre-activation) evaluated from its OUTPUT Y [M, N] (row stride ldy floats, 16-byte aligned rows) — TF's ReluGrad /
 * TanhGrad behind the input-gradient MatMul: the backward of `activation_fn(...)` at gnns/rgcn.py:113-114 and of the Dense between
 * layers, models/sparse_graph_model.py:194-200, meeting the MatMul gradient of the layer above.  Bit-identical to
 * relgnn_limb_gemm_xf32 / relgnn_limb16_gemm_xf32 followed by relgnn_act_bwd_from_output; saves that pass (read 2, write 1 per
 * element).  Y = NULL: plain product.  Other arguments as in relgnn_limb_gemm_xf32 / relgnn_limb16_gemm_xf32. */
int relgnn_limb_gemm_xf32_dact(int32_t act, const float* A, int64_t lda, const uint16_t* B, const float* bias, const void* zeros,
                               int32_t dact, const float* Y, int64_t ldy, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                               void* stream);
int relgnn_limb16_gemm_xf32_dact(int32_t act, const float* A, int64_t lda, const float* xmax, int32_t xgroups, const uint16_t* B,
                                 const float* wmax, const float* bias, const void* zeros, int32_t dact, const float* Y, int64_t ldy,
                                 float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream);
/* relgnn_limb_gemm_xf32 (_dact) with the work divided between wave ROLES (csrc/limb_gemm_pc.hip): eight producer waves stream the fp32
 * rows of A, split them and lay them out as limb sub-slabs in LDS; eight matrix waves multiply them with W fragments read straight
 * from L2, without a workgroup barrier (hand-over by counters in LDS, as in relgnn_rgcn_fused_fwd).  The same MatMuls (gnns/rgcn.py:
 * 96-98 forward and input gradient; the Dense layers of models/sparse_graph_model.py:194-200), the same bits as
 * relgnn_limb_gemm_xf32_dact.  K % 128 == 0, N % 256 == 0 and N == 256 or K <= 256; act: linear, ReLU, or tanh with K in {128, 256, 512}; Y (nullable) / dact as in
 * relgnn_limb_gemm_xf32_dact.  RELGNN_EUNSUPPORTED otherwise: callers keep relgnn_limb_gemm_xf32 for those.
 *
 * status (nullable): the caller's hand-over status block, device int32[2].  The wave roles hand sub-slabs over through counters in
 * LDS; every poll of a counter is bounded so that a protocol bug cannot hang the device.  A wave whose poll runs out stops waiting,
 * finishes WITH WRONG NUMBERS and ORs RELGNN_HANDOVER_PC_MATRIX / RELGNN_HANDOVER_PC_PRODUCER into status[0] (the err_flag
 * convention: reporting through the return value would need a device sync).  status[1] is the poll bound, 0 = the default 2^22
 * rounds (tests write 1 to provoke a give-up).  The caller zeroes the block once and reads status[0] where it synchronises anyway
 * (the package: with every step's metrics copy, models/sparse_graph_model.py MetricsReadback -> RuntimeError).  NULL: nothing is
 * reported.
 * relgnn_limb_gemm_xf32_pc_supported: 1 iff (act, has_dact, M, N, K) is a shape this entry takes (alignment of the pointers aside) —
 * the one place the list lives, so that callers choose between this entry and relgnn_limb_gemm_xf32 without repeating it. */
#define RELGNN_HANDOVER_FUSED_MATRIX 1
#define RELGNN_HANDOVER_FUSED_GATHER 2
#define RELGNN_HANDOVER_PC_MATRIX 4
#define RELGNN_HANDOVER_PC_PRODUCER 8
int relgnn_limb_gemm_xf32_pc(int32_t act, const float* A, int64_t lda, const uint16_t* B, const float* bias, int32_t dact,
                             const float* Y, int64_t ldy, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t* status,
                             void* stream);
int relgnn_limb_gemm_xf32_pc_supported(int32_t act, int32_t M, int32_t N, int32_t K);
/*
 * relgnn_rgcn_fused_fwd — the aggregate-first RGCN layer in ONE kernel (north_star "per-edge-type linear transform
 * fused as an MFMA GEMM"):
 *
 *     S[v, l, :] = sum_{p in bucket (v, l)} (w ? w[p] : 1) * H[col[p], :]      bucket (v, l) = rowptr[v*L + l] .. rowptr[v*L + l + 1]
 *     out[v, :]  = act(bias + sum_l S[v, l, :] @ W_l)
 *
 * = gnns/rgcn.py:84-114 for the sum-like aggregations in the aggregate-first order (embedding_lookup :87-89, 1/in-degree scale
 * :100-104, the per-type Dense :96-98 summed over the types, unsorted_segment_sum :109-112, activation :114) — what
 * relgnn_seg_reduce_fwd (mode SUM, seg_stride 1) followed by relgnn_limb_gemm_xf32 compute, bit for bit (same sequential fp32
 * fold per bucket, same three-bf16-limb split, same k-tile and limb-product order), without the [V, L*256] fp32 round trip and
 * with the gather and the matrix-pipe work of a row panel overlapped on one CU (csrc/rgcn_fused.hip: gather waves hand the
 * bucket sums, already split into limbs, to the MFMA waves through LDS).
 *
 * H [num_rows_h, ldh] fp32 (256 columns read; `col` holds row ids of H as relgnn_plan_* / relgnn_relational_keys bucketed them —
 * their range was checked there, through err_flag: like relgnn_seg_reduce_fwd this entry does not look at them again, num_rows_h
 * documents the table's height and is checked for sign only); w_limbs: the limb tiles of the stacked W^T [256, L*256]
 * (relgnn_limb_split_multi_f32 with transpose: the operand relgnn_limb_gemm_xf32 takes as B); bias nullable;
 * bucket_sums (nullable, [V, lds >= L*256]): S as fp32, for the weight gradient of a training step (dW_l = S_l^T dOut);
 * out [V, ldo].  d_in = d_out = 256 only (RELGNN_EUNSUPPORTED otherwise); 16-byte aligned pointers, strides % 4 == 0.
 * Buckets of any length are walked by ONE wave: callers with hub buckets (ops.SplitPlan) keep the two-kernel route.
 *
 * status (nullable): the hand-over status block of relgnn_limb_gemm_xf32_pc (device int32[2]); a gather / matrix wave whose poll
 * runs out ORs RELGNN_HANDOVER_FUSED_GATHER / RELGNN_HANDOVER_FUSED_MATRIX into status[0] and the results are wrong.
 */
int relgnn_rgcn_fused_fwd(const float* H, int64_t num_rows_h, int64_t ldh, const int32_t* rowptr, int32_t num_nodes,
                          int32_t num_edge_types, const int32_t* col, const float* w, const uint16_t* w_limbs, const float* bias,
                          int32_t act, float* bucket_sums, int64_t lds, float* out, int64_t ldo, int32_t d_in, int32_t d_out,
                          int32_t* status, void* stream);
/* The same product in 128 x 128 panels, two workgroups per CU, with what the per-(node, type) transforms of many-type graphs need
 * (gnns/gnn_film.py:92-106; the limb counterpart of relgnn_panel_gemm_f32's a_rows / b_select for the forward product and the input
 * gradient; K = 128 there):
 *   a_rows (nullable)   output row r reads A[a_rows[r], :] (a_rows[r] < 0: a row of zeros)
 *   B, num_b, b_batch_stride, b_select (nullable), rows_per_select
 *                       num_b weight matrices at B + i * b_batch_stride floats (RELGNN_GEMM_NN: [K, N] each, RELGNN_GEMM_NT: [N, K]
 *                       each); rows [p * rows_per_select, (p+1) * rows_per_select) of the output use matrix b_select[p]
 *                       (rows_per_select % 128 == 0); without b_select num_b must be 1
 *   limb_ws             >= num_b * relgnn_limb_elements(N, K) bf16 elements of device scratch (all matrices are split first)
 * Requirements (RELGNN_EUNSUPPORTED otherwise): K % 16 == 0, 16-byte aligned rows of A; N % 128 == 0 and 16-byte aligned rows of C,
 * or (num_b == 1, no b_select) any N and ldc >= N: the last 128-column chunk is cut at N when it is stored (the [V, 121] logits of the
 * PPI head, tasks/ppi_task.py:165-178); limb_ws then holds relgnn_limb_elements(N rounded up to 128, K) elements.
 * K = 128 and >= 32 k rows: persistent workgroups with the weights resident in LDS (see csrc/limb_gemm.hip). */
int relgnn_limb_dense_sel_f32(int32_t layout, int32_t act, const float* A, int64_t lda, const int32_t* a_rows, const float* B,
                              int64_t ldb, int32_t num_b, int64_t b_batch_stride, const int32_t* b_select, int32_t rows_per_select,
                              const float* bias, const void* zeros, uint16_t* limb_ws, int64_t limb_ws_elements, float* C,
                              int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream);
/* relgnn_limb_dense_sel_f32 with the weights ALREADY split (the same MatMul call sites: gnns/gnn_film.py:92-106, gnns/ggnn.py:76-81,
 * the D = 128 Dense layers of models/sparse_graph_model.py:194-200): B_limbs holds num_b limb images of [N, K] operands one behind
 * the other, relgnn_limb_elements(N, K) elements each, as relgnn_limb_split_multi_f32 / relgnn_limb_split_batch_f32 write them (an
 * NN weight [K, N] is split transposed).  N % 128 == 0, K % 16 == 0, ldc % 4 == 0; a_rows / b_select / rows_per_select / bias / act
 * as above.  The package splits the images of a step's weights once per optimizer step and calls this entry for every product. */
int relgnn_limb_gemm_sel_xf32(int32_t act, const float* A, int64_t lda, const int32_t* a_rows, const uint16_t* B_limbs, int32_t num_b,
                              const int32_t* b_select, int32_t rows_per_select, const float* bias, const void* zeros, float* C,
                              int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream);
/* The same typed product (no bias, no activation: gnns/gnn_film.py:92-106 and its input gradients) on the wave-role form of
 * relgnn_limb_gemm_xf32_pc (csrc/limb_gemm_pc_typed.hip): eight producer waves gather the rows a_rows[r] of A (< 0: zeros), split
 * them and hand 32-row x 128-k limb sub-slabs through LDS to eight matrix waves that read their W fragments straight from L2 — an
 * N = 256 product gathers every row once (the LDS-resident-weights kernel behind relgnn_limb_gemm_sel_xf32: once per 128-column
 * chunk).  Bit-identical to relgnn_limb_gemm_sel_xf32.  M % 64 == 0, N and K in {128, 256}, rows_per_select % 64 == 0
 * (relgnn_limb_gemm_sel_pc_supported; RELGNN_EUNSUPPORTED otherwise); zeros: >= 128 zero floats; status: the hand-over status block
 * of relgnn_limb_gemm_xf32_pc (nullable). */
int relgnn_limb_gemm_sel_pc_supported(int32_t M, int32_t N, int32_t K, int32_t rows_per_select);
int relgnn_limb_gemm_sel_pc_xf32(const float* A, int64_t lda, const int32_t* a_rows, const uint16_t* B_limbs, int32_t num_b,
                                 const int32_t* b_select, int32_t rows_per_select, const void* zeros, float* C, int64_t ldc,
                                 int32_t M, int32_t N, int32_t K, int32_t* status, void* stream);
/* Weight gradients dW = A^T @ G (A [V, J] = the saved layer input, G [V, C] = the output gradient; tf.gradients of the Dense
 * products above) on the same limb arithmetic: both operands fp32 row-major, split AND transposed in flight (the reduction index is
 * the row of both).  The kernel takes the first V - V % 32 rows, cut into relgnn_limb_gemm_tn_chunks(V, J, C) chunks; chunk z writes
 * its partial product to P + z*J*C; the caller sums the slabs in chunk order and adds the product of the last V % 32 rows
 * (relgnn_sum_slabs_tail_f32 does both in one pass: deterministic).
 * Requirements (RELGNN_EUNSUPPORTED otherwise): V >= 32, J % 32 == 0, C % 256 == 0, 16-byte aligned rows. */
int64_t relgnn_limb_gemm_tn_chunks(int32_t V, int32_t J, int32_t C);
int relgnn_limb_gemm_tn_f32(const float* A, int64_t lda, const float* G, int64_t ldg, float* P, int32_t V, int32_t J, int32_t C,
                            void* stream);
/* Typed weight-gradient partials of many-type graphs — tf.gradients of the per-edge-type Dense kernels Edge_%i_Weight
 * (gnns/gnn_film.py:94, gnns/rgcn.py:96-98) and Edge_%i_FiLM_Computations (gnns/gnn_film.py:102-106) where the transforms run over a
 * compact table of the non-empty (node, type) buckets (relgnn_limb_dense_sel_f32's a_rows / b_select form).  The table's P rows come
 * in tiles of rows_per_tile rows of ONE edge type each; tile z's partial gradient is
 *     part[z] = A[a_rows[z * rows_per_tile .. ]]^T @ G[z * rows_per_tile .. ]        ([J, C] floats at part + z*J*C)
 * with a_rows[r] the row of the [*, J] node table A that table row r was computed from (< 0: padding, contributes zeros), G [P, C]
 * the gradient of the table.  Arithmetic: the exact three-bf16-limb split of both operands, gathered, transposed and split in
 * flight (relgnn_limb_gemm_tn_f32's kernel).  The caller sums the tiles of a type in tile order (deterministic).
 * Requirements (RELGNN_EUNSUPPORTED otherwise): rows_per_tile % 32 == 0, P % rows_per_tile == 0, J % 64 == 0, C % 128 == 0, 16-byte
 * aligned rows and a_rows; zeros = the relgnn_panel_gemm_zeros_floats() block. */
int relgnn_limb_gemm_tn_tiles_f32(const float* A, int64_t lda, const int32_t* a_rows, const float* G, int64_t ldg,
                                  const void* zeros, float* part, int32_t P, int32_t rows_per_tile, int32_t J, int32_t C,
                                  void* stream);
/* The Dense product as the path calls it, fp32 in / fp32 out: splits the weights B (RELGNN_GEMM_NN: [K, N] as tf.layers.dense
 * stores its kernel; RELGNN_GEMM_NT: [N, K]) into limb_ws (>= relgnn_limb_elements(N, K) bf16 elements of device scratch, reusable
 * by the next call on the same stream), then runs relgnn_limb_gemm_xf32: C = act(bias + A @ B) resp. A @ B^T. */
int relgnn_limb_dense_f32(int32_t layout, int32_t act, const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                          const void* zeros, uint16_t* limb_ws, int64_t limb_ws_elements, float* C, int64_t ldc, int32_t M,
                          int32_t N, int32_t K, void* stream);

/* ========================================================================== *
 * 11. Dynamic per-target convolution kernels  (gnns/rgdcn.py:126-160)
 * ========================================================================== */

/*
 * Replaces, for sum / mean / sqrt_n aggregation: the per-channel, per-type loop of gnns/rgdcn.py:121-160 —
 * tf.reshape of the computed weights to [V, K, K] (:139), their tf.nn.embedding_lookup per edge (:140-141), the
 * per-edge tf.einsum('vi,vij->vj') (:146), tf.concat over types (:155), tf.unsorted_segment_* (:156-159) and the
 * activation (:160).  The K x K kernel depends on the target node only, so it is applied once per
 * (target, type, channel) to A = the source states summed into the (target, type) buckets by relgnn_seg_reduce_fwd
 * (seg_stride 1, num_segments V*L):
 *     out[v, c*K + j] = out_act( f_mode( sum_l sum_i A[(v*L + l)*C*K + c*K + i] * weight_act(P[v,l,c][i*K + j]) ) )
 * P (pre-activation output of the weight-computation Dense, :134-138) is addressed in place as
 *     P + v*p_node_stride + l*p_type_stride + c*p_channel_stride  (K*K contiguous floats).
 * rowptr_t: the [V*L+1] (target, type) bucket pointers (message count of a target for mean / sqrt_n; may be NULL for
 * sum).  Backward: G = d loss / d f_mode(sum) [V, C*K] -> gA (layout of A) and gP (layout of P).
 * channel_dim must be a power of two <= 64 for the backward (RELGNN_EUNSUPPORTED otherwise).
 */
int relgnn_rgdcn_apply_fwd(int32_t mode, int32_t weight_act, int32_t out_act, const float* A, const float* P,
                           int64_t p_node_stride, int64_t p_type_stride, int64_t p_channel_stride, int32_t num_nodes,
                           int32_t num_edge_types, int32_t num_channels, int32_t channel_dim, const int32_t* rowptr_t,
                           float* out, void* stream);
int relgnn_rgdcn_apply_bwd(int32_t weight_act, const float* A, const float* P, int64_t p_node_stride, int64_t p_type_stride,
                           int64_t p_channel_stride, int32_t num_nodes, int32_t num_edge_types, int32_t num_channels,
                           int32_t channel_dim, const float* G, float* gA, float* gP, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RELGNN_H_ */
