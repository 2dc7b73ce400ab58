/*
 * lecone.h -- C ABI of liblecone.so: the MI355X (gfx950) kernels + host sampler behind the reference's
 * joint image+label hyperbolic entailment-cone training step (ankitdhall/learning_embeddings).
 *
 * The reference has NO FFI: the path sits behind duck-typed Python objects.  These entry points are
 * therefore the layer a maintainer would bind with ctypes from inside the reference's own classes; each one names the
 * reference code it replaces (file:line relative to the reference root).  INTEGRATION.md shows the binding stubs.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no torch types, no exceptions across the boundary, no ownership transfer.
 *   - every function returns 0 on success, a negative LEC_E_* code on failure; lec_last_error() gives the message
 *     (thread-local).  Python raises on nonzero.
 *   - all device pointers are caller-owned HBM buffers; every GPU entry takes the hipStream_t to enqueue on (pass
 *     torch's current stream) and is asynchronous with respect to the host.
 *   - float tensors are fp32 row-major with an explicit leading dimension (ld, in elements).
 *   - "node code" (int32): c >= 0 is a row of the label table; c < 0 is row (-1 - c) of the per-step image-feature
 *     buffer (raw CNN outputs).
 */
#ifndef LECONE_H
#define LECONE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lec_stream_t;              /* a hipStream_t, passed as an opaque pointer (0 = default stream) */

#define LEC_OK            0
#define LEC_E_ARG        -1              /* bad argument (null pointer, size/ld out of range, unsupported D)     */
#define LEC_E_HIP        -2              /* a HIP runtime call or kernel launch failed                            */
#define LEC_E_EMPTY      -3              /* sampler: empty candidate list (python's random.choice raises there)   */
#define LEC_E_STATE      -4              /* bad handle / not initialised                                          */

/* which energy E(x, y) a kernel evaluates */
#define LEC_ENERGY_HYP_CONE  0           /* network/oe_h.py:811-833  (Poincare-ball entailment cone, parameter K) */
#define LEC_ENERGY_ORDER     1           /* network/order_embeddings.py:818-824  sum_d max(0, x_d - y_d)^2        */
#define LEC_ENERGY_EUC_CONE  2           /* network/oe.py:721-739  Euclidean entailment cone in cosine space:
                                            max(0, -<x/|x|, (y-x)/|y-x|> + sqrt(1 - K^2/|x|^2)), parameter K        */

/* how a label-table row becomes a point (the Embedder.forward of the trainer in use) */
#define LEC_LABEL_RAW        0           /* order_embeddings.py:188-193 with K=None: the row itself               */
#define LEC_LABEL_HYP        1           /* oe_h.py:77-104: +1e-15, tanh(clamp(atanh(r_in)+|e|))*e/|e|, then the
                                            no-grad clip of rows into [r_in, 1-1e-5] (straight-through)           */
#define LEC_LABEL_SOFTCLIP_K 2           /* oe.py:65-80 Embedder.forward with K: x/|x| * (|x| + K)                 */
/* how a raw CNN output row becomes a point */
#define LEC_IMAGE_RAW        0
#define LEC_IMAGE_SOFTCLIP   1           /* oe_h.py:323-328 FeatCNN18.soft_clip: x/|x| * (|x| + r_in)             */
#define LEC_IMAGE_SOFTCLIP_K 2           /* oe.py:225-240   FeatCNN18.soft_clip: x/|x| * (|x| + K)                 */

const char* lec_last_error(void);
int         lec_abi_version(void);       /* bumped on any signature change; checked by the Python loader           */
/* Bytes of device workspace the fused loss needs for a launch of (B, K): per-block loss partials + arrival counter. */
int64_t     lec_loss_workspace_bytes(int B, int K, int D);

/* ---------------------------------------------------------------------------------------------------------------
 * (1) Pair energies on dense rows.  Replaces E_operator (oe_h.py:811-833 / order_embeddings.py:818-824) and its
 *     autograd.  x, y: [P, D] (ldx, ldy);  E, gE: [P];  gx, gy: [P, D] (ldg).  gx/gy are overwritten.
 * ------------------------------------------------------------------------------------------------------------- */
int lec_pair_energy_fwd(int energy, const float* x, int64_t ldx, const float* y, int64_t ldy, int64_t P, int D,
                        float K_cone, float* E, lec_stream_t stream);
int lec_pair_energy_bwd(int energy, const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gE,
                        int64_t P, int D, float K_cone, float* gx, float* gy, int64_t ldg, lec_stream_t stream);
/* All-pairs scoring used by calculate_classification_metrics (oe_h.py:2018-2036): E[i, j] = E(x_j, y_i) for
 * x: [N, D] (apexes, e.g. every label), y: [M, D] (e.g. every image).  E: [M, N] (ldE). */
int lec_pair_energy_matrix(int energy, const float* x, int64_t ldx, int64_t N, const float* y, int64_t ldy, int64_t M,
                           int D, float K_cone, float* E, int64_t ldE, lec_stream_t stream);

/* Fused all-pairs scoring + per-level top-k for calculate_classification_metrics (oe_h.py:2018-2036: E_operator of one
 * image against every label, then torch.topk(k, largest=False) per level), without writing the M x N matrix.
 * level_start: HOST int32 [L+1] (L <= 32), level l = apex rows [level_start[l], level_start[l+1]); level_start[L] <= N.
 * out_idx, out_val: DEVICE [M, L, k]; ascending energy, ties by lowest index, NaN energies never selected; when a level
 * has fewer than k rows the tail is (-1, +inf).  1 <= k <= 8, D <= 224. */
int lec_level_topk(int energy, const float* x, int64_t ldx, int64_t N, const float* y, int64_t ldy, int64_t M, int D,
                   const int32_t* level_start, int L, int k, float K_cone, int32_t* out_idx, float* out_val,
                   lec_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (2) Fused joint loss, forward + backward in one launch.  Replaces criterion.forward's train branch AFTER negative
 *     sampling (oe_h.py:929-967: calculate_from_and_to_emb for positives and negatives, positive_pair, negative_pair,
 *     get_image_label_loss), Embedder.forward (oe_h.py:77-104) and FeatCNN18.soft_clip (oe_h.py:323-328) for the rows
 *     involved, and loss.backward() down to the label table and the raw CNN outputs.
 *
 *       loss = sum_b w_b E(u_b, v_b) + sum_b w_b sum_{k<2K} max(0, alpha - E_neg[b,k])         (oe_h.py:846)
 *       negative slot k <  K : (u_b, neg[b,k])        ("to" end corrupted,   oe_h.py:951-952)
 *       negative slot k >= K : (neg[b,k], v_b)        ("from" end corrupted, oe_h.py:955-957)
 *
 *     table    [n_labels, D] (ld_table)   label parameters (nn.Embedding weight)
 *     feat     [n_feat,   D] (ld_feat)    raw CNN outputs of this step's distinct images (may be NULL if n_feat == 0)
 *     pos_from, pos_to [B] node codes;  neg [B, 2K] node codes;  weights [B] or NULL (all 1.0)
 *     e_pos [B], e_neg [B, 2K], loss [1]: outputs (overwritten)
 *     grad_table [n_labels, D] (ld_table), grad_feat [n_feat, D] (ld_feat): d loss / d table, d loss / d feat are
 *       ADDED into these buffers (float atomics) -- zero them first for a plain gradient.  Pass NULL for both to run
 *       forward only.
 *     workspace: >= lec_loss_workspace_bytes(B, K, D) bytes of device memory, caller-owned, ZEROED ONCE by the caller
 *       when allocated and then reused across calls; launches sharing a workspace must be stream-ordered.
 * ------------------------------------------------------------------------------------------------------------- */
int lec_joint_loss_fwd_bwd(int energy, int label_proj, int image_proj,
                           const float* table, int64_t ld_table, int n_labels,
                           const float* feat, int64_t ld_feat, int n_feat,
                           const int32_t* pos_from, const int32_t* pos_to, const int32_t* neg, const float* weights,
                           int B, int K, int D, float K_cone, float alpha,
                           float* e_pos, float* e_neg, float* loss,
                           float* grad_table, float* grad_feat,
                           void* workspace, int64_t workspace_bytes, lec_stream_t stream);

/* The same launch with the label table READ from a 2-byte fp16 shadow [n_labels, D] (ld_table in elements): BASELINE.json config 5
 * ("fp16+fp32-master": 50 000 labels, 256 negatives per positive -- the gather of 2(1+K) table rows per positive is the kernel's
 * traffic, SURVEY.md 8(d) s = 2).  Gradients still go to the fp32 grad_table; the fp32 master is updated by lec_table_step_adam_f16,
 * which refreshes the shadow in the same pass. */
int lec_joint_loss_fwd_bwd_f16(int energy, int label_proj, int image_proj,
                               const void* table_f16, int64_t ld_table, int n_labels,
                               const float* feat, int64_t ld_feat, int n_feat,
                               const int32_t* pos_from, const int32_t* pos_to, const int32_t* neg, const float* weights,
                               int B, int K, int D, float K_cone, float alpha,
                               float* e_pos, float* e_neg, float* loss,
                               float* grad_table, float* grad_feat,
                               void* workspace, int64_t workspace_bytes, lec_stream_t stream);

/* The same loss for a step whose image embeddings come into being CHUNK BY CHUNK (the reference embeds the four groups of a step in separate
 * forwards, oe_h.py:980-1009, and keeps every activation; config 5's 7 424 CNN rows per step do not fit one pass): one launch per chunk, right
 * behind that chunk's CNN forward, evaluates exactly the pairs whose image row (the feature row -1-code of its image end point) lies in
 * [row_lo, row_hi) -- plus, when labels_too != 0, the pairs between two labels (pass it with one chunk of the step).  Over a step's launches every
 * pair of criterion.forward is evaluated once: e_pos / e_neg fill up entry by entry, each launch writes the sum of ITS terms to loss[0] (the caller
 * adds them), gradients add into grad_table / grad_feat.  Rows of `feat` outside the window are never read into a result.  table_f16 != NULL:
 * the label rows are read from the fp16 shadow (as lec_joint_loss_fwd_bwd_f16), else from `table`.
 * window_dev != NULL (device int32[4] = {row_lo, row_hi, labels_too, feat_base}): the window is read from device memory at kernel start (the three
 * arguments are ignored) and `feat` / `grad_feat` are chunk-sized buffers whose row 0 is feature row feat_base (n_feat = their row count) -- one launch,
 * captured once into a hipGraph, then serves every chunk of every step; the host rewrites four integers per chunk.
 * CONTRACT: a pair has at most ONE image end point, or both of its image rows lie in the same window (the reference's pick_per_level sampler
 * draws a label or an image AGAINST a label: oe_h.py:880-898; without pick_per_level two images can meet).  A pair whose two image rows fall into
 * different windows cannot be evaluated by either launch: the launch owning the larger row writes NaN to that pair's energy and to loss[0] -- loud,
 * never a finite number computed from a row that is not there.  Callers with such pairs use the whole-batch entry (lec_joint_loss_fwd_bwd). */
int lec_joint_loss_fwd_bwd_window(int energy, int label_proj, int image_proj,
                                  const float* table, const void* table_f16, int64_t ld_table, int n_labels,
                                  const float* feat, int64_t ld_feat, int n_feat,
                                  const int32_t* pos_from, const int32_t* pos_to, const int32_t* neg, const float* weights,
                                  int B, int K, int D, float K_cone, float alpha,
                                  int row_lo, int row_hi, int labels_too, const int32_t* window_dev,
                                  float* e_pos, float* e_neg, float* loss,
                                  float* grad_table, float* grad_feat,
                                  void* workspace, int64_t workspace_bytes, lec_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (3) Stand-alone projections (used outside the fused loss: evaluation, metrics, FeatCNN18.forward itself).
 *     lec_label_project_fwd  = Embedder.forward (oe_h.py:77-104): out[i] = project(table[idx[i]]).
 *     lec_label_project_bwd  adds d/d table into grad_table (dense, sparse=False semantics; float atomics).
 *     lec_image_softclip_fwd/bwd = FeatCNN18.soft_clip (oe_h.py:323-328) and its autograd.
 *     label_proj is LEC_LABEL_HYP or LEC_LABEL_SOFTCLIP_K, image_proj LEC_IMAGE_SOFTCLIP or LEC_IMAGE_SOFTCLIP_K
 *     (the oe.py forms, whose additive constant is K itself).
 * ------------------------------------------------------------------------------------------------------------- */
int lec_label_project_fwd(int label_proj, const float* table, int64_t ld_table, int n_labels, const int64_t* idx,
                          int64_t n, int D, float K_cone, float* out, int64_t ld_out, lec_stream_t stream);
int lec_label_project_bwd(int label_proj, const float* table, int64_t ld_table, int n_labels, const int64_t* idx,
                          int64_t n, int D, float K_cone, const float* gout, int64_t ld_gout, float* grad_table,
                          lec_stream_t stream);
int lec_image_softclip_fwd(int image_proj, const float* raw, int64_t ld_raw, int64_t n, int D, float K_cone, float* out,
                           int64_t ld_out, lec_stream_t stream);
int lec_image_softclip_bwd(int image_proj, const float* raw, int64_t ld_raw, const float* gout, int64_t ld_gout,
                           int64_t n, int D, float K_cone, float* graw, int64_t ld_graw, lec_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (4) Label-table maintenance, one pass over [n_labels, D].
 *     lec_table_step_adam replaces oe_h.py:1768-1771: grad *= ((1-|w|)/2)^2 (lambda_x :1632-1636) -> torch.optim.Adam
 *       step `step` (1-based; betas/eps as given, no weight decay/amsgrad) -> soft_clip rows into [r_in, 1-1e-5]
 *       (:1604-1617).  `riemannian` = 0 skips the rescale and `clip` = 0 skips the clip (Euclidean trainers).
 *     lec_table_step_rsgd replaces oe_h.py:1761-1762 / order_embeddings_h.py:764-775: exp_map_x(w, -lr * rescaled grad)
 *       with Mobius addition (:1619-1644), then the same clip.
 * ------------------------------------------------------------------------------------------------------------- */
int lec_table_step_adam(float* table, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t ld, int n_labels,
                        int D, float lr, float beta1, float beta2, float eps, int step, float K_cone, int riemannian,
                        int clip, lec_stream_t stream);
int lec_table_step_adam_f16(float* table, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t ld, int n_labels,
                            int D, float lr, float beta1, float beta2, float eps, int step, float K_cone, int riemannian,
                            int clip, void* table_f16, lec_stream_t stream);   /* + fp16 shadow of the updated rows (config 5) */
int lec_table_step_rsgd(float* table, const float* grad, int64_t ld, int n_labels, int D, float lr, float K_cone,
                        lec_stream_t stream);
/* Adam over a flat fp32 parameter arena (the CNN's parameters live in one buffer so that the data-parallel gradient
 * all-reduce is ONE collective and the optimizer ONE launch).  Replaces optimizer_labels.step() for the CNN half of
 * oe_h.py:1523,1769.  grad_scale multiplies the gradient first (1.0 for the reference's SUM semantics).  param_bf16
 * (optional, n bf16 values): a low-precision shadow of the updated parameters written in the same pass. */
int lec_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, float grad_scale, void* param_bf16, lec_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (5) Negative sampler (HOST, bit-exact).  Replaces set_negative_graph + sample_negative_edge (oe_h.py:799-809,
 *     849-902; labels-only variant order_embeddings.py:797-816) WITHOUT the dense (N+M)^2 bool matrix: the candidate
 *     list "ascending indices j in the level window with A[u,j] == 1" is the window minus the node's sorted
 *     transitive-closure neighbours, and random.choice's pick is the r-th survivor with r drawn from CPython's
 *     MT19937 `_randbelow` stream.
 *
 *     Graph: nodes 0..n_labels-1 are labels in level order (level_sizes[L]); nodes n_labels..n_labels+n_images-1 are
 *     images.  label_edges: [n_label_edges, 2] (parent, child) pairs of the label DAG (not closed);
 *     image_parents: CSR (image_ptr [n_images+1], image_adj) listing, per image, the labels it hangs under directly
 *     (the reference adds one edge per level, oe_h.py:524-531).  The transitive closure is taken here.
 *     mode: 0 = joint trainer rules (level_id % (L+1), slot L = "labels only if an endpoint is an image, else images
 *     only", oe_h.py:880-898); 1 = labels-only trainer rules (level_id % L, order_embeddings.py:799).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct lec_sampler lec_sampler;
int  lec_sampler_create(lec_sampler** out, const int32_t* level_sizes, int n_levels,
                        const int32_t* label_edges, int64_t n_label_edges,
                        const int64_t* image_ptr, const int32_t* image_adj, int64_t n_images,
                        int pick_per_level, int mode, uint64_t seed);
void lec_sampler_destroy(lec_sampler* s);
int  lec_sampler_seed(lec_sampler* s, uint64_t seed);                    /* random.seed(seed)                       */
int  lec_sampler_set_levels_to_hide(lec_sampler* s, const int32_t* levels, int n);   /* oe_h.py:764-765, 850-854   */
/* The slot ids left after hiding, in the order oe_h.py:854 indexes them: `list(set(range(L+1)) - set(hidden))` is in CPython's set
 * iteration order, which is NOT ascending for L + 1 > 8 slots with fewer than 5 left (restated from setobject.c).  out: L + 1 entries. */
int  lec_sampler_visible_slots(const lec_sampler* s, int32_t* out, int* n);
/* One call of sample_negative_edge: side 0 = `u` fixed (corrupt the "to" end), side 1 = `v` fixed. */
int  lec_sampler_draw(lec_sampler* s, int side, int32_t node, int32_t level_id, int32_t* out);
/* The criterion's host loop (oe_h.py:940-957): for b < B: for p < K: draw(0, from[b], p) -> neg[b,p];
 * draw(1, to[b], p) -> neg[b,K+p].  Consumes the RNG in exactly that order. */
int  lec_sampler_draw_batch(lec_sampler* s, const int32_t* pos_from, const int32_t* pos_to, int B, int K, int32_t* neg);
int  lec_sampler_next_u32(lec_sampler* s, uint32_t* out);                /* raw MT19937 word (known-answer tests)   */
int64_t lec_sampler_tc_edges(const lec_sampler* s);                      /* |TC| (label-label + label-image)        */
/* The closure as CSR (replaces nx.transitive_closure at oe_h.py:539 for the host's G_train_tc): descendants of node u,
 * ascending, are adj[ptr[u] .. ptr[u+1]).  ptr: n_nodes + 1 entries; adj: lec_sampler_tc_edges() entries or NULL. */
int  lec_sampler_tc_export(const lec_sampler* s, int64_t* ptr, int32_t* adj);

/* ---------------------------------------------------------------------------------------------------------------
 * (5b) Data-parallel gradient exchange: a thin layer over RCCL (xGMI inside a node).  Replaces the reference's single-process
 *     nn.DataParallel (oe_h.py:301,1434,1439; ethec_experiments.py:240: parameter broadcast every forward + gather + reduce-add on
 *     device 0) by one process per GPU and a SUM all-reduce of the flat gradient arena (the loss is a plain sum over pairs,
 *     oe_h.py:843-846).  lec_dp_unique_id: rank 0 draws the 128-byte id and hands it to the other ranks (any side channel: a file,
 *     a TCP store); lec_dp_init: every rank joins, bound to GPU `device` (the caller's current HIP device is left as it was);
 *     lec_dp_allreduce_sum: in place over `count` elements of
 *     buf (dtype 0 = fp32, 1 = bf16), asynchronous on `stream`; call it per bucket of the arena as backward produces it.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct lec_dp lec_dp;
int  lec_dp_unique_id(void* id128);
int  lec_dp_init(lec_dp** out, int rank, int world, const void* unique_id, int device);
int  lec_dp_allreduce_sum(lec_dp* comm, void* buf, int64_t count, int dtype, lec_stream_t stream);
void lec_dp_destroy(lec_dp* comm);

/* ---------------------------------------------------------------------------------------------------------------
 * (6) Multi-level cross-entropy, forward + backward in one launch.  Replaces MultiLevelCELoss.forward
 *     (network/loss.py:29-38) and its autograd:  loss = mean_b sum_l w_l CE(logits[b, s_l:e_l], labels[b, l]).
 *     logits [B, C] (ld), level_labels [B, L] int64, level_sizes [L] (sum = C), level_weights [L] or NULL.
 *     glogits [B, C] (ld) is overwritten with d loss / d logits (pass NULL for forward only).
 *     level_sizes / level_weights are HOST arrays (L <= 16); class_weights: DEVICE array [C] of per-class weights (loss.py:16-25:
 *     the sample's term is scaled by its target class's weight) or NULL; workspace: >= 256 + 4*2048 bytes, zeroed once (as above).
 * ------------------------------------------------------------------------------------------------------------- */
int lec_multilevel_ce_fwd_bwd(const float* logits, int64_t ld, const int64_t* level_labels, int B, int C,
                              const int32_t* level_sizes, const float* level_weights, const float* class_weights, int L,
                              float* loss, float* glogits, void* workspace, int64_t workspace_bytes,
                              lec_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (7) Fused BatchNorm (+ residual add) (+ ReLU) on NHWC bf16 activations -- the HBM-bound half of the ResNet backbone
 *     behind FeatCNN18 / FeatCNN (oe_h.py:311,317 -> torchvision BasicBlock/Bottleneck: `relu(bn(conv(x)))`,
 *     `relu(bn(conv(x)) + identity)`, downsample `bn(conv(x))`) and their autograd.
 *     x, residual, y, dy, dx, dresidual: bf16 [M, C], C innermost (M = N*H*W), C % 8 == 0, C <= 2048.
 *     gamma, beta, running_mean/var, save_mean/invstd: fp32 [C].  training != 0: batch statistics (biased variance for normalisation,
 *     unbiased for the running estimate, running = (1-momentum)*running + momentum*batch; running stats may be NULL);
 *     training == 0: running statistics.  y = [relu](x*scale + shift [+ residual]).
 *     Backward: g = dy * [y > 0] (relu) ; dbeta = sum g ; dgamma = sum g*xhat ; dx = gamma*invstd*(g - mean(g) -
 *     xhat*mean(g*xhat)) ; dresidual = g (pass NULL when the layer had no residual).
 *     dy2 (optional): a second gradient stream for the same activation (an activation consumed by two branches -- the
 *     next block's conv path and its identity / downsample path); the kernels add the two on the fly instead of the
 *     framework materialising their sum.
 *     relu_mask (optional, [M, C/8] bytes): forward writes bit j of byte (row, c/8) = [y > 0]; backward given the mask
 *     reads it instead of y (1/16 of the bytes).  Pass NULL to mask from y.
 *     workspace: >= lec_bn_workspace_bytes(C) bytes, reusable across layers on one stream.
 * ------------------------------------------------------------------------------------------------------------- */
int64_t lec_bn_workspace_bytes(int C);
int lec_bn_fwd(const void* x, const void* residual, int64_t M, int C, const float* gamma, const float* beta, float eps,
               float momentum, float* running_mean, float* running_var, int training, float* save_mean,
               float* save_invstd, void* y, int relu, uint8_t* relu_mask, void* workspace, int64_t workspace_bytes,
               lec_stream_t stream);
int lec_bn_bwd(const void* dy, const void* dy2, const void* y, const uint8_t* relu_mask, const void* x, int64_t M, int C,
               const float* gamma, const float* save_mean, const float* save_invstd, void* dx, void* dresidual,
               float* dgamma, float* dbeta, int relu, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream);
/* The same kernels on fp32 activations -- the reference's own precision (oe_h.py:281-328 runs torchvision's ResNet in fp32,
 * no AMP anywhere): x, residual, y, dy, dy2, dx, dresidual are fp32 [M, C]; everything else as above.  The `_f32` twins of the
 * staged entry points further down (lec_bn_fwd_prestat, lec_bn_bwd_pass1 / _apply / _prereduced) and of the max pooling follow
 * the same rule: identical arguments, fp32 activations. */
int lec_bn_fwd_f32(const void* x, const void* residual, int64_t M, int C, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, int training, float* save_mean,
                   float* save_invstd, void* y, int relu, uint8_t* relu_mask, void* workspace, int64_t workspace_bytes,
                   lec_stream_t stream);
int lec_bn_bwd_f32(const void* dy, const void* dy2, const void* y, const uint8_t* relu_mask, const void* x, int64_t M, int C,
                   const float* gamma, const float* save_mean, const float* save_invstd, void* dx, void* dresidual,
                   float* dgamma, float* dbeta, int relu, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream);

/* 1x1 stride-1 convolution on NHWC bf16 as an HBM-bound MFMA GEMM, y[M, Cout] = x[M, Cin] * w[Cout, Cin]^T, optionally with
 * the BatchNorm statistics of its output in the epilogue: replaces the library convolution AND the statistics pass of
 * torchvision Bottleneck's bn(conv1x1(.)) pairs (oe_h.py:311,317) on the wide layers; with w := W^T it is also those
 * layers' data gradient.  x: [M, Cin] bf16 (M = N*H*W), w: [Cout, Cin] bf16, y: [M, Cout] bf16.
 * partials (optional): >= 512*2*Cout floats, receives n_partials (HOST int, <= 512) rows of [sum | sum of squares] per
 * channel of the bf16-rounded y -- the layout lec_bn_fwd_prestat consumes; pass NULL, NULL for a plain product.
 * w_transposed = 1: w is stored [Cin][Cout] -- i.e. it is the FORWARD weight of the layer whose data gradient this call
 * computes (x := dy) -- and is transposed while it is loaded; no transpose kernel is needed.
 * Shapes with a kernel instance: lec_conv1x1_supported(Cin, Cout, M) != 0. */
int lec_conv1x1_supported(int Cin, int Cout, int64_t M);
int lec_conv1x1_fwd(const void* x, const void* w, int w_transposed, int64_t M, int Cin, int Cout, void* y, float* partials,
                    int64_t partials_bytes, int* n_partials, lec_stream_t stream);
/* conv3 -> bn3 (+ identity, ReLU) of a bottleneck block (torchvision Bottleneck.forward, reached from oe_h.py:311,317) without the
 * BatchNorm apply pass: the 1x1 convolution reads a quarter of what it writes, so it runs twice.  lec_conv1x1_stats forms the
 * product and leaves only the statistics partials of its (bf16-rounded) output; lec_bn_fwd_finalize turns them into mean / invstd /
 * running statistics and the scale / shift pair at lec_bn_workspace_coeff_offset(C) bytes into the workspace (scale[C], shift[C]);
 * lec_conv1x1_fwd_bnapply forms the product again and writes y (the BatchNorm input, kept for backward), z = relu(y * scale + shift
 * + residual) and the bitmask of z > 0 -- the same values lec_conv1x1_fwd + lec_bn_fwd_prestat produce, one read of y and of the
 * residual less.  Shapes: lec_conv1x1_bnapply_supported ((64, 256), (128, 512); M % 32 == 0). */
int lec_conv1x1_bnapply_supported(int Cin, int Cout, int64_t M);
int lec_conv1x1_stats(const void* x, const void* w, int64_t M, int Cin, int Cout, float* partials, int64_t partials_bytes, int* n_partials,
                      lec_stream_t stream);
int64_t lec_bn_workspace_coeff_offset(int C);
int lec_bn_fwd_finalize(int64_t M, int C, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                        float* running_var, int n_partials, float* save_mean, float* save_invstd, void* workspace, int64_t workspace_bytes,
                        lec_stream_t stream);
int lec_conv1x1_fwd_bnapply(const void* x, const void* w, int64_t M, int Cin, int Cout, const float* scale, const float* shift,
                            const void* residual, void* y, void* z, uint8_t* relu_mask, lec_stream_t stream);
/* Data gradient of a 1x1 layer whose INPUT is the forked output z = relu(bn(x_bn) + residual) of a bottleneck block
 * (torchvision Bottleneck.forward as reached from oe_h.py:311,317: z feeds the next block's conv1 and its identity branch), with
 * pass 1 of that BatchNorm's backward folded into the epilogue: g = relu_mask * (dy W + dy2) (bf16), where dy2 is the
 * identity branch's gradient, written instead of the raw product, plus per-workgroup partials of (sum g, sum g * xhat) in the
 * layout lec_bn_bwd_prereduced consumes.  Replaces lec_conv1x1_fwd(w_transposed) + the reduce pass of lec_bn_bwd: one write and
 * one read of the [M, Cout] gradient less.  dy [M, Cin], w as for lec_conv1x1_fwd, dy2 / bn_x / g [M, Cout] bf16, relu_mask
 * [M, Cout / 8].  Shapes: lec_conv1x1_dgrad_bnfold_supported ((64, 256), (128, 256), (128, 512); M % 32 == 0). */
int lec_conv1x1_dgrad_bnfold_supported(int Cin, int Cout, int64_t M);
int lec_conv1x1_dgrad_bnfold(const void* dy, const void* w, int w_transposed, int64_t M, int Cin, int Cout, const void* dy2,
                             const void* bn_x, const uint8_t* relu_mask, const float* save_mean, const float* save_invstd, void* g,
                             float* partials, int64_t partials_bytes, int* n_partials, lec_stream_t stream);
/* The rest of that BatchNorm backward: finalize (d gamma, d beta, coefficients) from n_partials partial rows at the start
 * of `workspace`, then dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)).  The gradient of the residual branch is
 * g itself. */
int lec_bn_bwd_prereduced(const void* g, const void* x, int64_t M, int C, const float* gamma, const float* save_mean,
                          const float* save_invstd, int n_partials, void* dx, float* dgamma, float* dbeta, void* workspace,
                          int64_t workspace_bytes, int accumulate, lec_stream_t stream);
int lec_bn_bwd_prereduced_f32(const void* g, const void* x, int64_t M, int C, const float* gamma, const float* save_mean,
                              const float* save_invstd, int n_partials, void* dx, float* dgamma, float* dbeta, void* workspace,
                              int64_t workspace_bytes, int accumulate, lec_stream_t stream);
/* Weight gradient of the 3x3 / stride 1 / pad 1 / 64 -> 64 convolution (torchvision Bottleneck.conv2 of layer1, reached from
 * oe_h.py:311,317): dw[co][ky][kx][ci] (fp32, the channels_last weight layout) += sum over pixels of dy[.., co] * x[shifted by tap, ci],
 * accumulated with float atomics into the caller's gradient buffer.  dy, x: [N, H, W, 64] bf16; H % 8 == 0 and W % 8 == 0. */
int lec_conv3x3_c64_wgrad_supported(int N, int H, int W);
int lec_conv3x3_c64_wgrad(const void* dy, const void* x, int N, int H, int W, float* dw, lec_stream_t stream);
/* lec_bn_bwd in stages, for callers that run pass 2 elsewhere (lec_conv1x1_wgrad_bnapply).  lec_bn_bwd_pass1: g = mask * (dy [+ dy2])
 * written, sums reduced, d gamma / d beta and the coefficients c1, c2 finalized (c1[C], c2[C] at lec_bn_workspace_coeff_offset(C) bytes
 * into the workspace).  lec_bn_bwd_finalize: the finalize alone, from n_partials partial rows a convolution epilogue left.
 * lec_bn_bwd_apply: pass 2 alone (dx from g, x and the c1, c2 in the workspace). */
int lec_bn_bwd_pass1(const void* dy, const void* dy2, const uint8_t* relu_mask, const void* x, int64_t M, int C, const float* save_mean,
                     const float* save_invstd, void* g, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream);
int lec_bn_bwd_finalize(int64_t M, int C, int n_partials, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                        int accumulate, lec_stream_t stream);
int lec_bn_bwd_apply(const void* g, const void* x, int64_t M, int C, const float* gamma, const float* save_mean, const float* save_invstd,
                     void* dx, void* workspace, int64_t workspace_bytes, lec_stream_t stream);
int lec_bn_bwd_pass1_f32(const void* dy, const void* dy2, const uint8_t* relu_mask, const void* x, int64_t M, int C, const float* save_mean,
                         const float* save_invstd, void* g, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream);
int lec_bn_bwd_apply_f32(const void* g, const void* x, int64_t M, int C, const float* gamma, const float* save_mean, const float* save_invstd,
                         void* dx, void* workspace, int64_t workspace_bytes, lec_stream_t stream);
/* conv3 behind bn3, backward: the weight gradient of the 1x1 layer AND pass 2 of the BatchNorm backward in one kernel.  g [M, Cout] is
 * the masked gradient (pass 1's output: lec_bn_bwd's d residual, or lec_conv1x1_dgrad_bnfold's g), bn_x [M, Cout] the BatchNorm's
 * input, c1 / c2 [Cout] the per-channel means the finalize kernel leaves at lec_bn_workspace_coeff_offset(Cout) bytes into the
 * workspace.  Writes dx = gamma invstd (g - c1 - xhat c2) [M, Cout] (bf16, what lec_bn_bwd's pass 2 writes, bit for bit) for the data
 * gradient kernel and accumulates dw[Cout][Cin] (fp32) += dx^T x.  Replaces pass 2 + the weight-gradient kernel: dx is read once
 * less.  Shapes: lec_conv1x1_wgrad_bnapply_supported ((64, 256), (128, 512); M % 64 == 0). */
int lec_conv1x1_wgrad_bnapply_supported(int Cin, int Cout, int64_t M);
int lec_conv1x1_wgrad_bnapply(const void* g, const void* bn_x, const void* x, int64_t M, int Cin, int Cout, const float* gamma,
                              const float* save_mean, const float* save_invstd, const float* c1, const float* c2, void* dx, float* dw,
                              lec_stream_t stream);
/* Weight gradient of the same 1x1 layers: dw[Cout][Cin] (fp32) += dy[M, Cout]^T x[M, Cin], accumulated with float atomics
 * straight into the caller's gradient buffer (which must hold the running sum, e.g. zero at the start of a step):
 * replaces the library's weight-gradient kernel together with its zero-fill, its fp32 -> bf16 cast and the copy into the
 * optimizer's gradient slot.  M % 64 == 0; shapes: lec_conv1x1_wgrad_supported. */
int lec_conv1x1_wgrad_supported(int Cin, int Cout, int64_t M);
int lec_conv1x1_wgrad(const void* dy, const void* x, int64_t M, int Cin, int Cout, float* dw, lec_stream_t stream);
/* 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels on NHWC bf16 (ResNet-50 layer1's conv2, the
 * one 3x3 layer near the HBM ridge), same MFMA wave-strip scheme and optional statistics epilogue as lec_conv1x1_fwd.
 * x: [N, H, W, 64], w: [64 out][3][3][64 in] (a channels_last conv weight), y: [N, H, W, 64]; N*H*W % 32 == 0.
 * The layer's data gradient is the same call on dy with w'[ci][r][s][co] = w[co][2-r][2-s][ci]; w_transposed = 1 takes the
 * forward weight itself and applies that flip + transpose while loading it. */
int lec_conv3x3_c64_fwd(const void* x, const void* w, int w_transposed, int N, int H, int W, void* y, float* partials,
                        int64_t partials_bytes, int* n_partials, lec_stream_t stream);
/* The same for 128 -> 128 channels (layer2's conv2 at 28x28): weights streamed through LDS one tap at a time, any H, W. */
int lec_conv3x3_c128_fwd(const void* x, const void* w, int N, int H, int W, void* y, float* partials,
                         int64_t partials_bytes, int* n_partials, lec_stream_t stream);
/* lec_bn_fwd in training mode with the statistics pass already done: the first n_partials rows of the workspace hold
 * per-channel [sum | sum of squares] partials (written by lec_conv1x1_fwd into the SAME workspace). */
int lec_bn_fwd_prestat(const void* x, const void* residual, int64_t M, int C, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var, int n_partials,
                       float* save_mean, float* save_invstd, void* y, int relu, uint8_t* relu_mask, void* workspace,
                       int64_t workspace_bytes, lec_stream_t stream);
int lec_bn_fwd_prestat_f32(const void* x, const void* residual, int64_t M, int C, const float* gamma, const float* beta,
                           float eps, float momentum, float* running_mean, float* running_var, int n_partials,
                           float* save_mean, float* save_invstd, void* y, int relu, uint8_t* relu_mask, void* workspace,
                           int64_t workspace_bytes, lec_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (7b) Convolutions at the reference's precision: fp32 NHWC activations, fp32 weights, exact fp32 arithmetic on the f32-input
 *     matrix instruction (v_mfma_f32_32x32x2_f32).  Replaces torchvision's conv2d inside FeatCNN18 / FeatCNN (oe_h.py:311,317,
 *     :331-378; every 1x1 / 3x3 / 7x7, stride 1 / 2 layer of ResNet-18 / -50) and its autograd, as implicit GEMMs.
 *     x: [N, H, W, Cin], w / dw: [Cout][R][S][Cin] (a channels_last conv weight), y / dy: [N, Ho, Wo, Cout] with
 *     Ho = (H + 2 pad - R) / stride + 1.  Cin and Cout powers of two >= 4 (the stem's 3 input channels are padded to 4 by the
 *     caller), stride 1 or 2.
 *     lec_conv_f32_fwd: partials (optional, >= 512 * 2 * Cout floats) receives n_partials (HOST int, <= 512) rows of per-channel
 *       [sum | sum of squares] of y -- the layout lec_bn_fwd_prestat_f32 consumes: the BatchNorm statistics pass over y disappears.
 *     lec_conv_f32_dgrad: dx is overwritten (a strided layer runs one launch per parity class of the input pixels).
 *     lec_conv_f32_wgrad: dw += (float atomics over the split reduction; zero it first for a plain gradient).
 * ------------------------------------------------------------------------------------------------------------- */
int lec_conv_f32_fwd(const float* x, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                     float* y, float* partials, int64_t partials_bytes, int* n_partials, int schedule, lec_stream_t stream);
/*     lec_conv_f32_stem_fwd (csrc/conv_stem_f32.hip): torchvision's stem convolution (conv1 of the resnet FeatCNN18 / FeatCNN wrap, oe_h.py:281-378: 7x7 /
 *       stride 2 / pad 3, 3 -> 64 channels) on its own kernel, exact fp32.  x [N, H, W, 4] and w [64][7][7][4] fp32 as lec_conv_f32_fwd takes them for the stem;
 *       channels 0..2 enter the product, channel 3 is never multiplied; y [N, H/2, W/2, 64]; partials as lec_conv_f32_fwd (one row per workgroup).
 *       lec_conv_f32_stem_supported: even H, W in {64, 128, 224}, tensors below 2 GiB; anything else goes through lec_conv_f32_fwd. */
int lec_conv_f32_stem_supported(int N, int H, int W);
int lec_conv_f32_stem_fwd(const float* x, const float* w, int N, int H, int W, float* y, float* partials, int64_t partials_bytes, int* n_partials,
                          lec_stream_t stream);
int lec_conv_f32_dgrad(const float* dy, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                       float* dx, int schedule, lec_stream_t stream);
/*     lec_conv_f32_fwd_affine: the forward of an EVAL-mode network (FeatCNN.eval(): calculate_classification_metrics' image embedding in
 *       the 'val' / 'test' phases, oe_h.py:1989-2011): y = [relu](conv(x, w) * scale[c] + shift[c] [+ res]) with scale = gamma / sqrt(running_var
 *       + eps), shift = beta - running_mean * scale -- F.batch_norm(training=False) (+ the block's residual add and ReLU) in the convolution's
 *       epilogue, in the arithmetic of lec_bn_fwd_f32's apply pass (bit-equal to the two-kernel form); the raw convolution output is never
 *       written and inference runs without a BatchNorm pass.  res: null or [N, Ho, Wo, Cout]. */
/*     lec_bn_eval_coeffs_f32: the scale / shift vectors of an eval-mode BatchNorm, computed on the stream (so that they follow the parameters and
 *       running statistics earlier launches wrote; a captured inference graph recomputes them on every replay). */
int lec_bn_eval_coeffs_f32(int C, const float* gamma, const float* beta, float eps, const float* running_mean, const float* running_var,
                           float* scale, float* shift, lec_stream_t stream);
int lec_conv_f32_fwd_affine(const float* x, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                            float* y, const float* scale, const float* shift, const float* res, int relu, int schedule, lec_stream_t stream);
/*     Fused BatchNorm pieces (the fp32 convolutions are bound by the matrix pipe and leave HBM idle; BatchNorm passes are the reverse:
 *     what moves from a BatchNorm pass into a convolution's loader or epilogue is hidden).  torchvision Bottleneck's
 *     relu(bn(conv(.))) chain reached from oe_h.py:311,317, backward:
 *     lec_conv_f32_dgrad_fused (stride-1 layers):
 *       xsrc, coef (both or neither; 1x1 / pad 0): `dy` holds g, the masked gradient of the BatchNorm output behind this layer, xsrc
 *         that BatchNorm's input; the kernel forms dy = coef[0][c] g + coef[1][c] xsrc + coef[2][c] while loading (coef [3][Cout]
 *         from lec_bn_bwd_coeffs_f32): pass 2 of that BatchNorm's backward never runs as a kernel;
 *       xbn, mean, invstd, partials, n_partials (all or none; dres, mask optional): the kernel writes g = mask * (dx + dres) instead of
 *         dx -- pass 1 of the backward of the BatchNorm whose OUTPUT this layer consumed (xbn: its input [N, H, W, Cin]; mask: the ReLU
 *         bitmask lec_bn_fwd_f32 wrote; dres: the gradient from the output's other consumer) -- and leaves n_partials (HOST int,
 *         <= 512) rows of per-channel partial sums [2][Cin] (sum g, sum g * xhat) in `partials` for lec_bn_bwd_coeffs_f32 /
 *         lec_bn_bwd_finalize.
 *     lec_conv_f32_wgrad_fused: the same on-load form for the weight gradient of a 1x1 / stride 1 / pad 0 layer.
 *     lec_bn_bwd_coeffs_f32: sums the partials -> dgamma, dbeta (overwritten) and coef[3][C] = (gamma invstd, -gamma invstd^2 c2,
 *         gamma invstd (invstd c2 mean - c1)) with c1 = dbeta / M, c2 = dgamma / M, so that dx = coef0 g + coef1 x + coef2. */
int lec_conv_f32_dgrad_fused(const float* dy, const float* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                             float* dx, const float* xsrc, const float* coef, const float* dres, const float* xbn, const uint8_t* mask,
                             const float* mean, const float* invstd, float* partials, int64_t partials_bytes, int* n_partials,
                             int schedule, lec_stream_t stream);
int lec_conv_f32_wgrad_fused(const float* dy, const float* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                             float* dw, const float* xsrc, const float* coef, lec_stream_t stream);
/*     lec_conv_f32_wgrad_c3: the 3-channel stem.  x4 [N, H, W, 4] carries a zero 4th channel (the kernels want >= 4), dw3 is the layer's
 *       own [Cout][R][S][3] gradient: dw3 += with float atomics (safe under concurrent backward passes, unlike a separate add). */
int lec_conv_f32_wgrad_c3(const float* dy, const float* x4, int N, int H, int W, int Cout, int R, int S, int stride, int pad,
                          float* dw3, lec_stream_t stream);
int lec_bn_bwd_pass1_coeffs_f32(const void* dy, const void* dy2, const uint8_t* relu_mask, const void* x, int64_t M, int C, const float* gamma,
                                const float* save_mean, const float* save_invstd, void* g, float* dgamma, float* dbeta, float* coef,
                                void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream);   /* pass 1 as a kernel (writes g) + the same coefficients */
/*     `accumulate` (every BatchNorm backward entry point that produces d gamma / d beta: lec_bn_bwd, _pass1, _finalize, _prereduced,
 *     _coeffs_f32, _pass1_coeffs_f32 and their _f32 twins): nonzero = ADD d gamma / d beta into the output slots (float atomics) instead of
 *     overwriting them -- for a step that runs several backward passes over the same parameters, possibly on concurrent streams, and zeroes
 *     the slots once per step.  A per-call argument: there is no process-wide switch, two trainers in one process do not see each other. */
/*     lec_conv_f32_scratch: scratch for the BALANCED form of lec_conv_f32_fwd / the stride-1 data gradients on `stream`.  A launch whose 128 x 128
 *     output tiles would leave the last round of the chip's 512 workgroup slots mostly empty (ResNet-50 at the bench batch: 392 x 2^k tiles from
 *     layer2 on) is cut into 512 equal runs of (tile, K chunk) iterations instead; tiles whose chunks fall to several workgroups are summed through
 *     this scratch in a fixed order (same bits every run).  `buf`: lec_conv_f32_scratch_bytes() bytes, 16-byte aligned, ZEROED by the caller, alive
 *     until unregistered (buf = null); one per stream that launches convolutions (launches of one stream are ordered, so they share it).  Streams
 *     without a scratch use the tile-walk kernel.  With the balanced form the statistics / fold partials are one row per m-tile: n_partials can
 *     reach 2048 (the row count lec_bn_workspace_bytes provides for). */
int64_t lec_conv_f32_scratch_bytes(void);
int lec_conv_f32_scratch(lec_stream_t stream, void* buf, int64_t bytes);
/*     `schedule` (lec_conv_f32_fwd, _fwd_affine, _dgrad, _dgrad_fused): which of the two forms the call takes, per call (no process-wide switch):
 *     LEC_SCHEDULE_DEFAULT what LEC_CF_SK in the environment says (default: AUTO), LEC_SCHEDULE_TILE_WALK never the balanced form (what a step
 *     running concurrent passes wants: they fill each other's tails), LEC_SCHEDULE_AUTO the balanced form where it pays, LEC_SCHEDULE_BALANCED
 *     wherever it applies (tests).  Both forms give the same values to the last few bits (the K sum is split differently). */
#define LEC_SCHEDULE_DEFAULT   (-1)
#define LEC_SCHEDULE_TILE_WALK   0
#define LEC_SCHEDULE_AUTO        1
#define LEC_SCHEDULE_BALANCED    2
int lec_bn_bwd_coeffs_f32(int64_t M, int C, int n_partials, const float* gamma, const float* save_mean, const float* save_invstd,
                          float* dgamma, float* dbeta, float* coef, void* workspace, int64_t workspace_bytes, int accumulate, lec_stream_t stream);
int lec_conv_f32_wgrad(const float* dy, const float* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                       float* dw, lec_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (7c) The same fp32 convolutions on the bf16 matrix cores (csrc/conv_f32x3.hip): every fp32 operand is cut into three bf16
 *     pieces (x = h + m + l exactly) and every product into the six bf16 products above 2^-24 of it, each exact in the MFMA's
 *     fp32 accumulation: fp32-grade results (an fp32 dot product's error; tests measure it against fp64 next to (7b)) at 2.67x
 *     the matrix rate of the f32-input instruction.  Same tensors, shapes and semantics as (7b); replaces the same reference
 *     lines (oe_h.py:311,317,:331-378).  The weights arrive pre-split:
 *     lec_conv_f32x3_split_weights: w [Cout][R*S][Cin] fp32 -> planes_fwd (the forward GEMM's operand) and / or planes_t (the data
 *       gradient's), bf16 bit patterns in the kernels' tile-major order [column tile][k chunk][h | m | l][rows][16 k];
 *       lec_conv_f32x3_planes_elems(Cout, RS, Cin, transposed) elements each; either pointer may be NULL.  Run it after every
 *       optimizer step.
 *     lec_conv_f32x3_fwd takes planes_fwd, lec_conv_f32x3_dgrad takes planes_t (Cout % 32 == 0), lec_conv_f32x3_wgrad takes the
 *       fp32 tensors themselves (dw +=, float atomics, as in (7b)).
 * ------------------------------------------------------------------------------------------------------------- */
int64_t lec_conv_f32x3_planes_elems(int Cout, int RS, int Cin, int transposed);
int lec_conv_f32x3_split_weights(const float* w, int Cout, int RS, int Cin, uint16_t* planes_fwd, uint16_t* planes_t, lec_stream_t stream);
int lec_conv_f32x3_fwd(const float* x, const uint16_t* w_planes, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                       float* y, float* partials, int64_t partials_bytes, int* n_partials, lec_stream_t stream);
int lec_conv_f32x3_dgrad(const float* dy, const uint16_t* w_planes_t, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                         float* dx, lec_stream_t stream);
/*     lec_conv_f32x3_wgrad: layers with Cin and Cout powers of two >= 64 and at most 32 taps (lec_conv_f32x3_wgrad_supported; others, i.e. the
 *       stem: lec_conv_f32_wgrad). */
int lec_conv_f32x3_wgrad_supported(int Cin, int Cout, int R, int S);
int lec_conv_f32x3_wgrad(const float* dy, const float* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                         float* dw, lec_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (7d) The ResNet convolutions on bf16 activations (csrc/conv_bf16.hip): BASELINE config 5's 16-bit conv stack -- torchvision's resnet50
 *     inside FeatCNN (oe_h.py:331-351; finetuner.py:122 for the classifier experiments) -- as ONE implicit-GEMM family on
 *     v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16, fp32 accumulation: every layer shape, stride (1, 2) and direction; no library convolution is left on the
 *     16-bit path.  Tensors NHWC bf16: x [N, H, W, Cin], y [N, Ho, Wo, Cout]; weights w [Cout][R*S][Cin] bf16 (a channels_last
 *     [Cout, Cin, R, S] tensor); Cin, Cout powers of two >= 8 (the 3-channel stem: x and w carry zero channels 3..7); tensors < 2 GiB.
 *     lec_conv_bf16_fwd: `partials` (optional, >= 512 * 2 * Cout floats) receives n_partials (HOST int) rows of per-channel
 *       [sum y, sum y^2] over the bf16-ROUNDED output, lec_bn_fwd_prestat's layout: the BatchNorm statistics pass disappears.
 *     lec_conv_bf16_wt_transpose: w [Cout][RS][Cin] -> wt [Cin][RS][Cout], the data gradient's k-contiguous operand (after every
 *       optimizer step; tens of KB .. 4.7 MB per layer).
 *     lec_conv_bf16_dgrad: dx from dy [N, Ho, Wo, Cout] and wt (Cout % 64 == 0); a strided layer's parity classes run as one launch.
 *       With xbn / mean / invstd / partials / n_partials (all or none; stride 1) the result is not dx but g = mask * (dx + dres), rounded to
 *       bf16 -- pass 1 of the backward of the BatchNorm whose output this layer consumed (dres: that output's other consumer's gradient or
 *       NULL; mask: lec_bn_fwd's ReLU bitmask or NULL; xbn: that BatchNorm's input) -- and `partials` receives n_partials rows of
 *       [sum g, sum g * xhat] per channel (lec_bn_bwd_prereduced's input).
 *     lec_conv_bf16_wgrad: dw [Cout][R*S][dw_cin] fp32 += (float atomics over the K split; dw_cin = Cin, or the real channel count of a
 *       zero-padded stem input).
 * ------------------------------------------------------------------------------------------------------------- */
int lec_conv_bf16_supported(int Cin, int Cout, int R, int S, int stride, int pad);
int lec_conv_bf16_wt_transpose(const void* w, void* wt, int Cout, int RS, int Cin, lec_stream_t stream);
/*     ... and every convolution weight of a flat bf16 arena in ONE launch (after the Adam kernel has refreshed the arena): base / base_t = the arena and its
 *       transposed twin (same element offsets), table = DEVICE int32 [n_layers][5] {element offset, Cout, RS, Cin, first tile}, a layer owning
 *       RS * ceil(Cout / 64) * ceil(Cin / 64) consecutive tiles. */
int lec_conv_bf16_wt_transpose_flat(const void* base, void* base_t, const int32_t* table, int n_layers, int total_tiles, lec_stream_t stream);
int lec_conv_bf16_fwd(const void* x, const void* w, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                      void* y, float* partials, int64_t partials_bytes, int* n_partials, lec_stream_t stream);
/*     lec_conv_bf16_stem_fwd: torchvision's stem convolution (resnet.py conv1: 7x7 / stride 2 / pad 3, 3 -> 64 channels; FeatCNN, oe_h.py:331-351) on its own
 *       kernel.  x [N, H, W, 8] and w [64][7][7][8] bf16 as lec_conv_bf16_fwd takes them; channels 0..3 enter the product (the caller keeps channel 3 zero),
 *       channels 4..7 are never read; y [N, H/2, W/2, 64]; partials as lec_conv_bf16_fwd (up to 768 rows when the buffer holds them).
 *       lec_conv_bf16_stem_supported: even H and W in {64, 128, 224}; other sizes go through lec_conv_bf16_fwd. */
int lec_conv_bf16_stem_supported(int H, int W);
int lec_conv_bf16_stem_fwd(const void* x, const void* w, int N, int H, int W, void* y, float* partials, int64_t partials_bytes, int* n_partials,
                           lec_stream_t stream);
int lec_conv_bf16_dgrad(const void* dy, const void* wt, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                        void* dx, const void* dres, const void* xbn, const uint8_t* mask, const float* mean, const float* invstd,
                        float* partials, int64_t partials_bytes, int* n_partials, lec_stream_t stream);
int lec_conv_bf16_wgrad(const void* dy, const void* x, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                        float* dw, int dw_cin, lec_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (8) 3x3 / stride 2 / pad 1 max pooling on NHWC bf16 (the ResNet stem's `maxpool`, oe_h.py:311 -> torchvision).
 *     x: [N, H, W, C] bf16 (H, W even, C % 8 == 0); y: [N, H/2, W/2, C]; argmax: one byte per pooled element (window
 *     position kh*3+kw; first maximum wins, NaN propagates, like the framework op).  Backward is a gather over the <= 4
 *     windows covering an input position: dx is overwritten, no atomics.
 * ------------------------------------------------------------------------------------------------------------- */
int lec_maxpool3x3s2_fwd(const void* x, int N, int H, int W, int C, void* y, uint8_t* argmax, lec_stream_t stream);
int lec_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, int N, int H, int W, int C, void* dx, lec_stream_t stream);
int lec_maxpool3x3s2_fwd_f32(const void* x, int N, int H, int W, int C, void* y, uint8_t* argmax, lec_stream_t stream);   /* fp32 x, y */
int lec_maxpool3x3s2_bwd_f32(const void* dy, const uint8_t* argmax, int N, int H, int W, int C, void* dx, lec_stream_t stream);
/* The stem's tail as ONE op (fp32; csrc/bn.hip): p = maxpool3x3s2(relu(x * scale + shift)) -- torchvision ResNet `maxpool(relu(bn1(conv1(x))))`, reached from
 * FeatCNN18 / FeatCNN's backbone (oe_h.py:311,317; the train-mode BatchNorm of oe_h.py's `img_feat_net.train()` phases).  scale / shift: the vectors
 * lec_bn_fwd_finalize left in the BatchNorm workspace (lec_bn_workspace_coeff_offset).  p and argmax are bit for bit what lec_bn_fwd_prestat_f32(relu = 1) followed by
 * lec_maxpool3x3s2_fwd_f32 produce; the normalised activation is never written.  Backward: from dp, argmax and x both passes of the BatchNorm backward
 * (d gamma, d beta as lec_bn_bwd_f32 leaves them -- `accumulate` as there -- and dx, the gradient of the convolution output x); the pooling's input gradient
 * is never written either.  C / 8 must divide 256 (C <= 512); H, W even. */
int lec_bn_relu_maxpool_fwd_f32(const void* x, int N, int H, int W, int C, const float* scale, const float* shift, void* p, uint8_t* argmax, lec_stream_t stream);
int lec_bn_relu_maxpool_bwd_f32(const void* dp, const void* dp2 /* or NULL: a second gradient of p (it fed two branches), added on load */, const uint8_t* argmax, const void* x, int N, int H, int W, int C, const float* gamma, const float* beta,
                                const float* save_mean, const float* save_invstd, void* dx, float* dgamma, float* dbeta, void* workspace,
                                int64_t workspace_bytes, int accumulate, lec_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (9) HBM-resident image store (csrc/image_store.hip): the input side of the step.  Replaces, for every image row of a step's CNN
 *     batch, the reference's host-side ToTensor (+ RandomHorizontalFlip for positives) + torch.stack + .to(device): oe_h.py:700-712 and
 *     1463-1471 (positives, DataLoader workers), oe_h.py:668-677 called from 980-983 / 1003-1007 (every image drawn as a negative,
 *     synchronously in the training thread).  The caller keeps the RESIZED uint8 images (what cv2.imread -> ToPILImage -> Resize gives,
 *     [H, W, 3], channel order as decoded) in one device buffer and asks for the float batch:
 *     store: DEVICE uint8 [n_slots, H, W, 3];  slots: DEVICE int32 [n];  flip: DEVICE uint8 [n] or NULL (nonzero = mirrored along W);
 *     out:   DEVICE fp32 [n, H, W, c_out] (NHWC = a channels_last [n, c_out, H, W] tensor), c_out = 3, or 4 with a zero 4th channel
 *            (what the f32 stem convolution consumes).  out[i, h, w, c] = float(store[slots[i], h, flip ? W-1-w : w, c]) / 255 with an
 *            IEEE fp32 division: bit-identical to ToTensor.  A slot outside [0, n_slots) gives a black image.
 * ------------------------------------------------------------------------------------------------------------- */
int lec_image_gather_u8(const uint8_t* store, int64_t n_slots, const int32_t* slots, const uint8_t* flip, int n, int H, int W,
                        int c_out, float* out, lec_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LECONE_H */
