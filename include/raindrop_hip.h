/*
 * raindrop_hip.h -- C-ABI of libraindrop_hip.so, the MI355X (gfx950) implementation of the
 * Raindrop hot path: Raindrop_v2.forward and its backward.
 *
 * The reference (/root/reference, pure Python on PyTorch + PyTorch-Geometric) has NO native
 * interface for this path; its "FFI" is the set of torch / PyG / torch_scatter op call sites in
 *   code/models_rd.py:278-387      (Raindrop_v2.forward)
 *   code/Ob_propagation.py:94-228  (Observation_progation.forward / message / aggregate)
 *   code/transformer_conv.py:139-207
 * Each entry point below names the call sites it replaces.  Host binding: ctypes
 * (raindrop_amd/_lib.py); see INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every tensor pointer is DEVICE memory owned by the caller
 *     (the library never allocates, frees or retains tensor memory); fp32 unless stated;
 *     "int64" tensors are torch.int64; masks are 1 byte per element (torch.bool);
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no host
 *     synchronisation, no internal streams, safe to capture into a hipGraph;
 *   - scratch memory comes from the caller: query rd_*_workspace_bytes(), pass a buffer that
 *     stays alive until the stream has run the call;
 *   - return 0 on success, RD_EINVAL / RD_EUNSUPPORTED (negative) for argument errors detected
 *     before any launch, or a positive hipError_t from the launch; rd_last_error() returns a
 *     thread-local message for the last non-zero return;
 *   - re-entrant; results are deterministic (no floating-point atomics).  The library keeps no per-call state.  Two
 *     settings exist outside the call arguments: the arithmetic mode (rd_set_precision, process-wide, meant to be chosen
 *     once at start-up) and the dropout seed cell (rd_set_seed_cell, per host thread, read at enqueue time).
 */
#ifndef RAINDROP_HIP_H
#define RAINDROP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RD_OK 0
#define RD_EINVAL (-1)
#define RD_EUNSUPPORTED (-2)

#define RD_ABI_VERSION 1

/* Arithmetic of the dense contractions (inputs, outputs and accumulation are always fp32):
 *   RD_PREC_FP32   v_mfma_f32_16x16x4_f32, bitwise an fp32 fmaf chain (157 TF/s peak)
 *   RD_PREC_BF16X3 each fp32 operand split into bf16 hi+lo, products hi*hi + hi*lo + lo*hi on
 *                  v_mfma_f32_16x16x32_bf16 with fp32 accumulate: ~2^-16 relative per product
 *                  (fp32-class after accumulation; logits agree with the fp32 path to ~1e-6),
 *                  5.3x the MFMA throughput.  Default; env RD_PRECISION=fp32 selects the other. */
#define RD_PREC_FP32 0
#define RD_PREC_BF16X3 1
/*   RD_PREC_BF16   operands rounded to bf16 (hi part only), ONE product per MFMA step, fp32 accumulate: the arithmetic of
 *                  BASELINE.json's "P12 ... bf16" configuration (~2^-9 per product; logits within 3e-2 of the fp32
 *                  reference).  Dense layers of the encoder / head / generic message passing only; the fused
 *                  K1 path, attention, softmax, LayerNorm stay as in the other modes.  env RD_PRECISION=bf16. */
#define RD_PREC_BF16 2

/* Problem shape shared by the model-level entry points.
 * K = T*d_ob (channels per sensor node), Dm = F*d_ob, D = Dm + d_pe (transformer width). */
typedef struct rd_shape {
  int32_t B;         /* samples in this batch (columns of src)                                  */
  int32_t T;         /* time steps == max_len (code/models_rd.py:243: Linear(max_len*d_ob, .))  */
  int32_t F;         /* sensors, d_inp                                                          */
  int32_t d_ob;      /* observation-embedding width per sensor (code/Raindrop.py:126 -> 4)      */
  int32_t d_pe;      /* positional-encoding width (code/models_rd.py:216 -> 16)                 */
  int32_t nhead;     /* attention heads (code/Raindrop.py:130 -> 2)                             */
  int32_t nhid;      /* FFN width (code/Raindrop.py:128 -> 2*F*d_ob)                            */
  int32_t d_static;  /* static feature width, 0 when the model has no static branch            */
  int32_t n_classes;
  int32_t max_len;   /* PE timescale base (== T for the shipped configs)                        */
} rd_shape;

/* ---- library identity -------------------------------------------------------------------- */
int rd_version(void);             /* RD_ABI_VERSION                                           */
const char* rd_arch(void);        /* "gfx950"                                                  */
const char* rd_last_error(void);  /* thread-local; "" if none                                  */
int rd_set_precision(int32_t mode); /* process-wide; RD_PREC_*                                 */
int rd_get_precision(void);
/* Dropout seeds are passed by value, which a captured hipGraph freezes.  If a device cell is
 * registered, every dropout kernel adds its content to the seed it was launched with; bumping the
 * cell (a 1-thread kernel, itself capturable) gives each graph replay fresh masks while forward and
 * backward of one step still agree.  Pass NULL to unregister.  The registration is per HOST THREAD and is consumed
 * when a call is ENQUEUED (the pointer is passed to the kernels as an argument): register it around the enqueue or the
 * stream capture of a step and unregister afterwards -- a captured graph keeps using the cell it was captured with, and
 * calls made while no cell is registered are unaffected by it.  The cell must outlive every launch that received it. */
int rd_set_seed_cell(const uint64_t* device_cell);
int rd_seed_cell_advance(uint64_t* device_cell, uint64_t delta, void* stream);

/* Side branch for trailing launches.  rd_set_side_stream registers (per host thread, consumed when a call is enqueued; NULL
 * clears) a second stream: launches that only produce parameter gradients nothing later in the backward pass reads -- the slice
 * reduces of rd_encoder_layer_bwd's weight-gradient stream, rd_head_train's weight-gradient tiles and loss mean -- are then
 * ordered behind the caller's stream and enqueued THERE, beside whatever the caller enqueues next.  The caller owns the join:
 * rd_side_join(stream) makes `stream` wait for the side branch (call it before reading those gradients / the loss on `stream`,
 * and before the end of a stream capture that forked).  With no side stream registered nothing changes.  Results are identical
 * either way (the same kernels on the same data; only their overlap changes). */
int rd_set_side_stream(void* stream);
int rd_side_join(void* stream);

/* Deferred trailing launches (round 4; what replaced the side branch as the default of raindrop_amd.step.TrainStep).  With
 * rd_set_defer_trailing(1) (per host thread, consumed at enqueue) the same launches are not enqueued at all but PARKED -- one at a
 * time -- and the next rd_encoder_layer_bwd that runs its fused backward chain appends them to that launch as extra workgroups, which
 * run on the CUs the chain leaves idle.  A parked launch nobody picks up runs stand-alone when the next one is parked or when
 * rd_flush_trailing(stream) is called: call it before reading those gradients / the loss, and before the end of a stream capture.
 * Results are identical in every mode (same arithmetic on the same data).
 * Two rules for a C-ABI caller: (1) a parked slice reduce of layer i -- including the LayerNorm dgamma / dbeta column sums over the
 * layer's partial matrices -- runs INSIDE layer i-1's backward launch, beside workgroups that rewrite the partials and row tiles of
 * the workspace THEY were given: while the mode is on, every layer needs its OWN rd_encoder_layer workspace until
 * rd_flush_trailing (raindrop_amd.step.TrainStep: `enc_wss`); sharing one workspace across layers is only legal with the mode off.
 * (2) The slot holds raw device pointers.  rd_set_defer_trailing(0) and rd_set_side_stream(NULL) DISCARD whatever is still parked
 * (so does rd_drop_trailing): leave the mode through them on every error path, flush before leaving it on the normal one. */
int rd_set_defer_trailing(int32_t on);
int rd_flush_trailing(void* stream);
int rd_drop_trailing(void);

/* ---- token plan: the padding mask as a layout --------------------------------------------------------------------------
 * code/models_rd.py:298-299 builds mask[b,t] = (t >= lengths[b]) and uses it twice: as src_key_padding_mask of the encoder
 * (:358) and in the masked mean (:366-367,379).  Between them the two uses remove every padded step from the logits AND from
 * every gradient (its rows of the encoder's activation gradients are exactly zero).  rd_token_plan turns `lengths` into a
 * layout: samples ordered by descending length (ties by index), sample b's live steps t < min(lengths[b], T) at rows
 * off[rank[b]] + t of every [tokens, *] tensor.  While a plan is registered (rd_set_token_plan: per host thread, consumed when a
 * call is ENQUEUED, like the seed cell), rd_sensor_stage_fwd, rd_encoder_layer_fwd/bwd, rd_head_train and rd_msgpass_bwd read and
 * write ONLY those rows -- buffers keep their padded sizes, the first plan[0] rows are used -- and skip the arithmetic whose
 * operand is an exactly-zero padded step.  Logits, loss and all parameter gradients are the same function of the inputs as
 * without a plan (up to the order of the fp32 sums over tokens); z / x / dx at padded steps, which nothing reads, are not
 * produced.  Supported where the step runs on plan-aware kernels: a bf16 arithmetic mode, d_ob = 4, the fused row-local encoder
 * chains (ceil(D / 32) == 5, ceil(nhid / 32) == 9: P19 and P12), head_dim <= 96, the fused head -- at any T (T <= 64: in_proj + attention
 * as one launch per direction, rd_attnfuse.hip; beyond: the multi-tile attention kernels on plan rows) and with either message-passing
 * form (fused LDS-resident, or the panel products whose last scatter follows the plan; rd_pe_mask writes the PE rows the same way).
 * Other shapes / modes return RD_EUNSUPPORTED while a plan is registered (raindrop_amd.step.plan_supported is the host-side test).
 * plan_out: rd_token_plan_bytes(s) bytes of int32 (layout: raindrop_amd/csrc/rd_plan.h; [0] = live rows, [1] = their 32-row chunks,
 * [5] = chunks of the per-sample chunk space the fused attention exports its weight-gradient row tiles in; per-rank off / len,
 * per-sample rank / first row / length).  If seed_cell is not NULL the launch also adds `delta` to it (rd_seed_cell_advance folded
 * in: one launch fewer per step). */
size_t rd_token_plan_bytes(const rd_shape* s);
int rd_token_plan(const rd_shape* s, const int64_t* lengths, int32_t* plan_out, uint64_t* seed_cell, uint64_t delta,
                  void* stream);
int rd_set_token_plan(const int32_t* device_plan);

/* ---- a5/a6: integer work (bit-exact contracts) -------------------------------------------- */

/* code/models_rd.py:307-311: adj = global_structure; adj[diag] = 1; edge_index =
 * nonzero(adj).T (row-major; row 0 = source j, row 1 = target i); edge_weights = adj[r, c].
 * adj_out [F,F] receives the diagonal-patched adjacency (the reference patches its input in
 * place); edge_index is int64 [2, F*F] capacity with row stride F*F, edge_weights [F*F],
 * n_edges int32[1].  Entries beyond n_edges are left untouched. */
int rd_graph_build(int32_t F, const float* global_structure, float* adj_out, int64_t* edge_index,
                   float* edge_weights, int32_t* n_edges, void* stream);

/* code/models_rd.py:298-299: mask[b,t] = (t >= lengths[b]) as bytes [B,T];
 * code/models_rd.py:28-38,292,354: pe[t,b,:] = [sin(times/ts), cos(times/ts)] written into
 * columns [F*d_ob, F*d_ob + d_pe) of the concat buffer z [T,B,D]; timescales[d_pe/2] are
 * computed by the host in float64 and passed as fp32, exactly as the reference casts them. */
int rd_pe_mask(const rd_shape* s, const float* times, const int64_t* lengths,
               const float* timescales, float* z, uint8_t* mask, void* stream);

/* ---- a7-a13: inter-sensor message passing (kernel K1) ------------------------------------- */

/* PyG `softmax(edge_weights, index=target)` + per-target coefficient sum on the dense sensor
 * graph (code/Ob_propagation.py:187,195 via models_rd.py:307-311): for the patched adjacency
 * adj [F,F] (edge j->i exists iff adj[j,i] != 0) writes gamma[j,i] = softmax over sources j of
 * adj[:,i] (0 where no edge) and ssum[i] = sum_j gamma[j,i].  One wavefront per target node,
 * wave-shuffle reductions; F <= 1024. */
int rd_edge_softmax(int32_t F, const float* adj, float* gamma, float* ssum, void* stream);

/* The same softmax over an explicit edge list (duplicate edges allowed): edge_index int64 with
 * rows [source; target] `row_stride` elements apart, normalised by row `norm_row`
 * (1 = target: code/Ob_propagation.py:195, code/transformer_conv.py:201; 0 = source: the
 * use_beta branch, code/Ob_propagation.py:184).  gamma_e [E], ssum [N]. */
int rd_edge_softmax_list(int32_t N, int32_t E, const int64_t* edge_index, int64_t row_stride,
                         int32_t norm_row, const float* edge_weights, float* gamma_e, float* ssum,
                         void* stream);
/* The same followed by F.dropout(gamma, p) (code/Ob_propagation.py:196, code/transformer_conv.py:203; training mode of an operator
 * constructed with dropout > 0 -- the shipped model uses 0): gamma_e[e] is 0 or gamma / (1 - p), ssum the sum of THOSE; the mask is
 * a function of (seed + the registered seed cell, edge index). */
int rd_edge_softmax_list_dropout(int32_t N, int32_t E, const int64_t* edge_index, int64_t row_stride, int32_t norm_row,
                                 const float* edge_weights, float p_drop, uint64_t seed, float* gamma_e, float* ssum, void* stream);

/* Source-valued aggregate of TransformerConv (code/transformer_conv.py:158,168-175,205-206) on a
 * dense coefficient matrix: out[i,c] = sum_j gamma[j,i] V[j,c] (+ skip[i,c] if skip != NULL);
 * backward w.r.t. V: dV[j,c] = sum_i gamma[j,i] dout[i,c]. */
int rd_aggregate_fwd(int32_t N, int32_t C, const float* gamma, const float* V, const float* skip,
                     float* out, void* stream);
int rd_aggregate_bwd(int32_t N, int32_t C, const float* gamma, const float* dout, float* dV,
                     void* stream);

/* The GENERAL message / aggregate of the reference's TransformerConv (code/transformer_conv.py:186-207; round 6): H heads of C
 * channels, scores from the projections instead of given edge weights --
 *   alpha[e,h] = softmax over the edges into tgt(e) of <q[tgt(e),h], k[src(e),h] + edge_feat[e,h]> / sqrt(C)   (edge_feat =
 *   lin_edge(edge_attr) [E,H*C] or NULL), F.dropout(alpha, p) in training mode (:203), out[i,h] = sum_e alpha[e,h] v[src(e),h].
 * q, k, v [N,H*C] row-major; edge_index rows [source; target] `row_stride` apart (duplicate edges allowed); alpha [E,H] receives
 * the POST-softmax coefficients (what the operator returns), alpha_drop [E,H] the dropped ones (kept for the backward), out
 * [N,H*C].  Backward: dout [N,H*C] -> dq, dk, dv [N,H*C], dedge_feat [E,H*C] (NULL iff edge_feat is); ds_ws: E*H floats.
 * Deterministic: sums over a node's edges in edge order, fixed-tree block reductions.  Endpoints must lie in [0,N): the host
 * mirror validates each edge list once (IndexError, like the reference's index_select); the kernels clamp what they index with, so
 * a bad list gives unspecified coefficients for that edge, not a wild read.  E = 0 is legal (out = 0). */
int rd_edge_attention_fwd(int32_t N, int32_t E, int32_t H, int32_t C, const float* q, const float* k, const float* v,
                          const float* edge_feat, const int64_t* edge_index, int64_t row_stride, float p_drop, uint64_t seed,
                          float* alpha, float* alpha_drop, float* out, void* stream);
int rd_edge_attention_bwd(int32_t N, int32_t E, int32_t H, int32_t C, const float* q, const float* k, const float* v,
                          const float* edge_feat, const int64_t* edge_index, int64_t row_stride, float p_drop, uint64_t seed,
                          const float* alpha, const float* alpha_drop, const float* dout, float* ds_ws, float* dq, float* dk,
                          float* dv, float* dedge_feat, void* stream);

/* Batched forms (replace the per-sample Python loop + torch index_put_ the round-2 TransformerConv surface used):
 *  - rd_edge_softmax_list_batched: B edge lists in one launch, edge_index [B][2,E] int64 `batch_stride` elements apart (0 = one
 *    shared list), rows `row_stride` apart; weights [B,E] `w_bstride` floats apart (0 = shared); gamma_e [B,E], ssum [B,N].
 *    Layer 2 of the use_beta model reads each sample's OWN pruned edge list through it (code/models_rd.py:331-335).
 *  - rd_edge_gamma_dense: dense coefficient matrix gamma[j*N + i] = sum of gamma_e over the edges j -> i, duplicate edges added in
 *    edge order (the scatter-add of PyG's aggregate, code/transformer_conv.py:139-160, deterministic: no float atomics).
 *  - rd_aggregate_batched_fwd/bwd: out[b,i,c] = sum_j gamma[j,i] V[b,j,c] (+ skip[b,i,c]) for B feature matrices that share the
 *    graph, as ONE batched product (the legacy `Raindrop` model loops the operator over the batch, code/models_rd.py:155-165). */
int rd_edge_softmax_list_batched(int32_t B, int32_t N, int32_t E, const int64_t* edge_index, int64_t batch_stride,
                                 int64_t row_stride, int32_t norm_row, const float* edge_weights, int64_t w_bstride,
                                 float* gamma_e, float* ssum, void* stream);
int rd_edge_gamma_dense(int32_t N, int32_t E, const int64_t* edge_index, int64_t row_stride, const float* gamma_e,
                        float* gamma_dense, void* stream);
int rd_aggregate_batched_fwd(int32_t B, int32_t N, int32_t C, const float* gamma, const float* V, const float* skip,
                             float* out, void* stream);
int rd_aggregate_batched_bwd(int32_t B, int32_t N, int32_t C, const float* gamma, const float* dout, float* dV, void* stream);

size_t rd_msgpass_workspace_bytes(const rd_shape* s);   /* scratch of rd_msgpass_bwd            */
size_t rd_msgpass_saved_bytes(const rd_shape* s);       /* forward -> backward hand-over buffer */

/* Forward of both Observation_progation layers for the whole batch
 * (code/models_rd.py:285-296,313-343 + code/Ob_propagation.py:157-228, default branch):
 *   X[b,f,t*d+c]  = relu(src[t,b,f] * R_u[f*d+c])            (observation embedding)
 *   Y1 = relu(X  W1^T + b1) * ssum[f];   Y2 = relu(Y1 W2^T + b2) * ssum[f]
 *   z[t,b,f*d+c]  = Y2[b,f,t*d+c]                            (columns [0, F*d) of z, row stride ldz)
 * src [T,B,2F] (values in the first F columns).  `saved` (rd_msgpass_saved_bytes) receives what
 * the backward pass needs: X and Y1 as [B,F,K] and the weights as split-bf16 operand tiles (fused path: its
 * planes; other shapes, bf16 modes: W1, W2, W2^T, W1^T as native tiles, 4 x 4 ceil16(K) ceil32(K) bytes).
 * p_drop > 0 applies nn.Dropout to the embedding h (code/models_rd.py:296) with a counter-based mask
 * that is a pure function of (seed, element index); p_drop = 0 in eval mode.
 * Shapes with F <= 48, d_ob = 4, K = T*d_ob <= 240, K % 16 == 0 (P19) run as ONE fused
 * LDS-resident kernel per batch (one workgroup per sample); others as two panel products (rd_gemm.hip:
 * k_gemm_panel) behind the observation-embedding kernel. */
int rd_msgpass_fwd(const rd_shape* s, const float* src, const float* R_u, const float* W1,
                   const float* b1, const float* W2, const float* b2, const float* ssum,
                   float p_drop, uint64_t seed, float* z, int32_t ldz, void* saved, size_t saved_bytes,
                   void* stream);

/* rd_pe_mask + rd_msgpass_fwd in one call for the model's layout (z [T,B,D], D = F*d_ob + d_pe,
 * contiguous; mask [B,T]).  On the fused path the positional encoding and the padding mask of a
 * sample are produced by the same workgroup that owns its message passing. */
int rd_sensor_stage_fwd(const rd_shape* s, const float* src, const float* times, const int64_t* lengths,
                        const float* timescales, const float* R_u, const float* W1, const float* b1,
                        const float* W2, const float* b2, const float* ssum, float p_drop, uint64_t seed,
                        float* z, uint8_t* mask, void* saved, size_t saved_bytes, void* stream);

/* rd_sensor_stage_fwd after rd_step_prepare has written this step's operand tiles of W1, W2 into `saved`: same arguments, same
 * results, no weight-split launch of its own. */
int rd_sensor_stage_fwd_prepared(const rd_shape* s, const float* src, const float* times, const int64_t* lengths,
                                 const float* timescales, const float* R_u, const float* W1, const float* b1,
                                 const float* W2, const float* b2, const float* ssum, float p_drop, uint64_t seed,
                                 float* z, uint8_t* mask, void* saved, size_t saved_bytes, void* stream);

/* Backward of rd_msgpass_fwd.  dz is the gradient w.r.t. z (row stride ldz; only the first
 * F*d columns are read).  Writes dW1,db1,dW2,db2 and dR_u [F*d] (overwrite, not accumulate). */
int rd_msgpass_bwd(const rd_shape* s, const float* src, const float* R_u, const float* W1,
                   const float* W2, const float* ssum, float p_drop, const void* saved,
                   size_t saved_bytes, const float* z, const float* dz, int32_t ldz, float* dW1,
                   float* db1, float* dW2, float* db2, float* dR_u, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ---- a15/a16: temporal self-attention encoder (kernels K2/K3) and masked mean (K5) --------- */

/* Parameters of one nn.TransformerEncoderLayer(D, nhead, nhid) (code/models_rd.py:235-237),
 * names as in its state_dict: self_attn.in_proj_{weight[3D,D],bias}, self_attn.out_proj.*,
 * linear1.{weight[nhid,D],bias}, linear2.{weight[D,nhid],bias}, norm1.*, norm2.* */
typedef struct rd_encoder_weights {
  const float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b, *lin1_w, *lin1_b, *lin2_w, *lin2_b,
      *norm1_w, *norm1_b, *norm2_w, *norm2_b;
} rd_encoder_weights;
typedef struct rd_encoder_grads {
  float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b, *lin1_w, *lin1_b, *lin2_w, *lin2_b,
      *norm1_w, *norm1_b, *norm2_w, *norm2_b;
} rd_encoder_grads;

size_t rd_encoder_layer_saved_bytes(const rd_shape* s);      /* activations kept for backward */
size_t rd_encoder_layer_workspace_bytes(const rd_shape* s);  /* scratch, fwd and bwd */

/* One post-norm encoder layer, torch semantics (torch/nn/modules/transformer.py:799-983, used at
 * code/models_rd.py:358): x,y [T,B,D] with D = F*d_ob + d_pe; mask [B,T] bytes (1 = padded key);
 * y = LN2(x1 + drop(W2 drop(relu(W1 x1)))) with x1 = LN1(x + drop(out_proj(MHA(x)))).
 * The four dropout sites use Philox masks keyed by (seed, site, layer). */
int rd_encoder_layer_fwd(const rd_shape* s, int32_t layer, const float* x, const uint8_t* mask,
                         const rd_encoder_weights* w, float p_drop, uint64_t seed, float* y,
                         void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                         void* stream);
/* Optional: the weight-dependent part of the forward (splitting the layer's matrices into matrix-core operand tiles inside
 * `saved`) as its own call, for callers whose weights change less often than they run forward (inference: prepare once per
 * checkpoint); then pass `layer | RD_LAYER_WEIGHTS_PREPARED` to rd_encoder_layer_fwd, which skips that launch.  The caller
 * orders the two calls (same stream, or an event). */
#define RD_LAYER_WEIGHTS_PREPARED 0x10000
int rd_encoder_layer_prepare(const rd_shape* s, const rd_encoder_weights* w, void* saved, size_t saved_bytes,
                             void* stream);
/* Every weight matrix of a training step -> operand tiles in ONE launch (the weights change once per step, in the optimizer):
 * the `nlayers` (<= 2) encoder layers' into their `saved` buffers (what rd_encoder_layer_prepare does per layer) and the two
 * lin_value matrices of the message-passing stage into its `saved` buffer (what rd_sensor_stage_fwd does first; pass NULL to
 * skip).  Then call rd_encoder_layer_fwd(layer | RD_LAYER_WEIGHTS_PREPARED) and rd_sensor_stage_fwd_prepared.
 * rd_step_prepare_covers reports (1 / 0) whether the encoder layers / the sensor stage of this shape take prepared tiles in the
 * current arithmetic mode; where it says 0 the plain entry points must be used. */
int rd_step_prepare(const rd_shape* s, int32_t nlayers, const rd_encoder_weights* const* w, void* const* enc_saved,
                    const size_t* enc_saved_bytes, const float* W1, const float* W2, void* k1_saved, size_t k1_saved_bytes,
                    void* stream);
int rd_step_prepare_covers(const rd_shape* s, int32_t* encoder, int32_t* sensor_stage);
/* rd_token_plan and rd_step_prepare as ONE launch (the plan is built by one extra block row of the weight-split kernel: independent workgroups, a wave per sample): what a
 * hipGraph training step enqueues first.  seed_cell_dev / delta as in rd_token_plan (NULL: no bump). */
int rd_step_begin(const rd_shape* s, const int64_t* lengths, int32_t* plan_out, uint64_t* seed_cell_dev, uint64_t delta,
                  int32_t nlayers, const rd_encoder_weights* const* w, void* const* enc_saved, const size_t* enc_saved_bytes,
                  const float* W1, const float* W2, void* k1_saved, size_t k1_saved_bytes, void* stream);
/* Backward: dy -> dx and all 12 parameter gradients (overwritten). */
int rd_encoder_layer_bwd(const rd_shape* s, int32_t layer, const float* x, const uint8_t* mask,
                         const rd_encoder_weights* w, float p_drop, uint64_t seed, const void* saved,
                         size_t saved_bytes, const float* dy, float* dx, const rd_encoder_grads* g,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The attention core of that layer alone (between in_proj and out_proj; torch F.multi_head_attention_forward as used at
 * code/models_rd.py:235-237,358): qkv [T,B,3D] (q | k | v, heads side by side) -> out [T,B,D] and lse [B,H,T] (log-sum-exp of the
 * scaled masked scores, consumed by the backward); backward dout [T,B,D] -> dqkv [T,B,3D] (delta_ws: B*H*T floats of scratch).
 * Same kernels as inside rd_encoder_layer_fwd/bwd (head_dim <= 96).  Dropout on the probabilities with Philox site
 * (attention, layer). */
int rd_attention_fwd(const rd_shape* s, int32_t layer, const float* qkv, const uint8_t* mask, float p_drop, uint64_t seed,
                     float* out, float* lse, void* stream);
int rd_attention_bwd(const rd_shape* s, int32_t layer, const float* qkv, const uint8_t* mask, float p_drop, uint64_t seed,
                     const float* out, const float* lse, const float* dout, float* dqkv, float* delta_ws, void* stream);

/* code/models_rd.py:366-367,379: out[b, :D] = sum_t r[t,b,:] * (1 - mask[b,t]) / (lengths[b] + 1);
 * out has row stride ldo (so it can be the left block of the [agg | emb] head input). */
int rd_masked_mean_fwd(const rd_shape* s, int32_t D, const float* r, const uint8_t* mask,
                       const int64_t* lengths, float* out, int32_t ldo, void* stream);
int rd_masked_mean_bwd(const rd_shape* s, int32_t D, const float* dout, int32_t ldo,
                       const uint8_t* mask, const int64_t* lengths, float* dr, void* stream);

/* ---- caller step (a19): the optimizer of code/Raindrop.py:256 ------------------------------- */

/* torch.optim.Adam (no amsgrad; weight_decay added to the gradient) over one flat buffer of n
 * floats: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
 * `step` is t (1-based).  All four buffers 16-byte aligned. */
int rd_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream);
/* The same update with the optimizer's step state on the DEVICE -- ONE launch a hipGraph can replay.  `state`: 64 bytes, 16-byte
 * aligned, eight doubles {t, beta1^t, beta2^t, lr, 0, weight_decay, 0, 0} OF THE STEP BEING APPLIED; the launch only reads it (no
 * workgroup of a launch ever sees a state another workgroup of the same launch wrote).  The state is advanced once per step by an
 * EARLIER launch: rd_adam_state_advance (one thread: t += 1, beta^t *= beta), or -- no extra launch -- by the step's first launch:
 * rd_set_adam_state registers the cell on this host thread and rd_step_begin (enqueued while it is registered) advances it next
 * to the dropout seed bump.  lr / weight_decay are read from the cell so that a learning-rate schedule (code/Raindrop.py:257-259)
 * is an 8-byte copy into it, not a new capture; beta1, beta2, eps are launch constants.
 * Initialise {steps taken so far, beta1^t, beta2^t, lr, 0, weight_decay, 0, 0}.
 * raindrop_amd.optim.FlatAdam.step_captured / raindrop_amd.step.TrainStep.capture_full. */
int rd_adam_step_dev(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float beta1, float beta2,
                     float eps, const void* state, void* stream);
int rd_adam_state_advance(void* state, float beta1, float beta2, void* stream);
int rd_set_adam_state(void* state, float beta1, float beta2);      /* NULL: unregister */

/* ---- generic dense pieces (used by the temporal encoder, the head and the large-K path) ---- */

/* y[M,N] = act(x[M,K] W[N,K]^T + b)   (torch.nn.functional.linear; act: 0 none, 1 relu). */
int rd_linear_fwd(int32_t M, int32_t N, int32_t K, const float* x, int32_t ldx, const float* W,
                  const float* b, float* y, int32_t ldy, int32_t act, void* stream);
/* The same product on the exact-fp32 matrix instruction whatever the process's arithmetic mode (rd_set_precision): for values that
 * feed index work (code/Ob_propagation.py:161-185: edge scores -> top-K pruning).  Affects this call only, on this host thread. */
int rd_linear_fwd_fp32(int32_t M, int32_t N, int32_t K, const float* x, int32_t ldx, const float* W,
                       const float* b, float* y, int32_t ldy, int32_t act, void* stream);
/* dx[M,K] = dy[M,N] W[N,K]   (optionally masked by relu_src > 0 when relu_src != NULL). */
int rd_linear_bwd_input(int32_t M, int32_t N, int32_t K, const float* dy, int32_t lddy,
                        const float* W, float* dx, int32_t lddx, void* stream);
/* dx = (dy W) where gate[m,k] > 0, else 0: the ReLU of nn.Sequential(Linear, ReLU, Linear)
 * (mlp_static, code/models_rd.py:254-258) folded into the input-gradient product. */
int rd_linear_bwd_input_gated(int32_t M, int32_t N, int32_t K, const float* dy, int32_t lddy,
                              const float* W, const float* gate, int32_t ldgate, float* dx, int32_t lddx,
                              void* stream);
/* torch.nn.CrossEntropyLoss() (mean) forward + backward in one launch (code/Raindrop.py:255,322):
 * loss[0] = mean_b(logsumexp(logits[b]) - logits[b, y[b]]); dlogits = (softmax - onehot) / B. */
int rd_softmax_xent(int32_t B, int32_t C, const float* logits, const int64_t* y, float* loss,
                    float* dlogits, void* stream);
/* Tuning knob of the encoder's row-block products (process-global, read when a product is enqueued): bit mask of the kernel
 * variants that run on 32-row instead of 64-row workgroups (1: K <= 160, 2: K <= 288, 4: LayerNorm epilogue, 8: LayerNorm-
 * backward prologue); -1 restores the default (environment RD_RG_ROWS32, else 15).  Results do not depend on it beyond
 * the grouping of the LayerNorm parameter-gradient partial sums.  raindrop_amd.step.TrainStep times both settings when
 * it captures its graph and keeps the faster one. */
int rd_set_rowgemm_rows32(int32_t mask);
/* Same bit assignment: kernel variants that run on 16-wave (1024-thread) workgroups, one column tile per wave and round, instead
 * of 8-wave ones; -1 restores the default (environment RD_RG_WAVES16, else 12). */
int rd_set_rowgemm_waves16(int32_t mask);
/* Classifier head + loss, forward and backward, in two launches (training step; rd_head.hip).
 *   agg = masked mean over time of r [T,B,D] (code/models_rd.py:366-367,379); feat = [agg | static W_emb^T + b_emb]
 *   (:381-384; Fe = 0: no static branch); logits = W2 relu(W0 feat + b0) + b2 (mlp_static, :263-267,385);
 *   loss = mean cross entropy against y (code/Raindrop.py:255,322).
 * Outputs: loss[1], logits [B,C], the six parameter gradients (overwritten, not accumulated) and dr [T,B,D], the
 * gradient entering the encoder stack.  fp32 FMA arithmetic in every precision mode.  Replaces the sequence
 * rd_masked_mean_fwd, rd_linear_fwd x3, rd_softmax_xent, rd_linear_bwd_weight x3, rd_linear_bwd_input(_gated),
 * rd_masked_mean_bwd of the same quantities.  rd_head_train_supported: D % 4 == 0, D + Fe <= 256, C <= 16 (and the
 * environment switch RD_HEAD_FUSED=0 reports 0); workspace: rd_head_train_workspace_bytes(B, D + Fe, C). */
int rd_head_train_supported(int32_t D, int32_t Fe, int32_t C);
size_t rd_head_train_workspace_bytes(int32_t B, int32_t dh, int32_t C);
int rd_head_train(const rd_shape* s, int32_t D, int32_t d_static, int32_t Fe, int32_t C, const float* r,
                  const uint8_t* mask, const int64_t* lengths, const float* stat, const float* emb_w,
                  const float* emb_b, const float* w0, const float* b0, const float* w2, const float* b2,
                  const int64_t* y, float* loss, float* logits, float* g_emb_w, float* g_emb_b, float* g_w0,
                  float* g_b0, float* g_w2, float* g_b2, float* dr, void* workspace, size_t workspace_bytes,
                  void* stream);
/* The same head around a loss the CALLER evaluates -- the reference's loop: outputs = model(..); loss = criterion(outputs, y);
 * loss.backward() (code/Raindrop.py:317-323).  rd_head_forward stops at the logits; rd_head_backward recomputes the head's
 * forward phases (a masked mean and two small products) and continues from the caller's d loss / d logits [B,C].  Same kernel,
 * arithmetic, support test and workspace as rd_head_train.  raindrop_amd.graph_module (the captured module step) uses them. */
int rd_head_forward(const rd_shape* s, int32_t D, int32_t d_static, int32_t Fe, int32_t C, const float* r, const uint8_t* mask,
                    const int64_t* lengths, const float* stat, const float* emb_w, const float* emb_b, const float* w0,
                    const float* b0, const float* w2, const float* b2, float* logits, void* workspace, size_t workspace_bytes,
                    void* stream);
int rd_head_backward(const rd_shape* s, int32_t D, int32_t d_static, int32_t Fe, int32_t C, const float* r, const uint8_t* mask,
                     const int64_t* lengths, const float* stat, const float* emb_w, const float* emb_b, const float* w0,
                     const float* b0, const float* w2, const float* b2, const float* dlogits, float* g_emb_w, float* g_emb_b,
                     float* g_w0, float* g_b0, float* g_w2, float* g_b2, float* dr, void* workspace, size_t workspace_bytes,
                     void* stream);
size_t rd_linear_bwd_weight_workspace_bytes(int32_t M, int32_t N, int32_t K);
/* dW[N,K] = dy[M,N]^T x[M,K];  db[N] = sum_m dy[m,:]  (db may be NULL).  Deterministic split
 * over M with a fixed-order reduction. */
int rd_linear_bwd_weight(int32_t M, int32_t N, int32_t K, const float* dy, int32_t lddy,
                         const float* x, int32_t ldx, float* dW, float* db, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ---- batch feed: the caller's per-step slice, code/Raindrop.py:310-317 ---------------- */

/* `P = Ptrain_tensor[:, idx, :]`, `Ptime = Ptrain_time_tensor[:, idx]`, `Pstatic = Ptrain_static_tensor[idx]`,
 * `y = ytrain_tensor[idx]` (host fancy-index + H2D in the reference, :310-315) and
 * `lengths = torch.sum(Ptime > 0, dim=0)` (:317) in ONE launch over a dataset that is resident in HBM:
 * P_all [T,N,W] (W = 2F), time_all [T,N], static_all [N,d_static] or NULL, y_all [N] or NULL, idx [B] int64
 * (device) -> src [T,B,W], times [T,B], static_out [B,d_static], y_out [B], lengths [B] int64.  Pure copies:
 * bit-exact.  An index outside [0,N) is clamped and ADDED to *bad_index_count (device int32, may be NULL; the
 * caller zeroes it -- the library enqueues nothing but the one kernel). */
int rd_batch_gather(int32_t T, int32_t B, int32_t W, int32_t d_static, int64_t N, const float* P_all,
                    const float* time_all, const float* static_all, const int64_t* y_all, const int64_t* idx,
                    float* src, float* times, float* static_out, int64_t* y_out, int64_t* lengths,
                    int32_t* bad_index_count, void* stream);

/* ---- paper-faithful graph operator: the use_beta branch of Observation_progation.message,
 * code/Ob_propagation.py:161-185,190-191,195,200,207-208,227, batched over B sample graphs that share an edge list.
 *   V [B,N,K] = relu(lin_value(x)),  H [B,N,T*32] = increase_dim(x)   (both per NODE: rd_linear_fwd)
 *   beta[i,t] = mean_c H[i,t,c] * cat(map_weights[i], p_t[t])[c];  gamma[e,t] = beta[tgt(e),t] * w[e]
 *   keep the int(E*0.5) edges with the largest mean score, in descending order (ties: lower edge id first);
 *   softmax of gamma per channel over the kept edges of one SOURCE node;  out[n] = sum_{kept e: src(e)=n} softmax[e] (.) V[tgt(e)]
 * edge_index int64, rows [source; target] `row_stride` apart; edge_weights [E] per sample (stride w_bstride floats, 0 =
 * shared); p_t [T,16] per sample (stride pt_bstride floats, 0 = shared).  Outputs: out [B,N,K]; edge_index_out
 * [B][2,Kk] int64 and alpha_out [B][Kk] (the pruned edges in pruning order and their mean scores: what the reference
 * returns as (self.edge_index, self._alpha)), Kk = rd_graph_beta_kept(E); beta_save [B,N,T] and kept int32 [B,Kk] are
 * handed to the backward.  d_ob must be 4.  Two forms behind the same entry points: graphs with N <= 64 nodes and E <= 4096 edges
 * whose per-step scores fit one workgroup's LDS run as ONE launch per direction (graph staged in LDS); anything larger, up to
 * N <= 1024 nodes and 2^28 edges (round 4: SYN256's 256 sensors x 65 536 edges x 512 steps), runs as a sequence of
 * element-parallel launches with the per-sample state -- sort keys, kept-edge lists by source and by target, softmax statistics
 * -- in `workspace` (256-byte aligned, rd_graph_beta_workspace_bytes(..) bytes, 0 where the LDS form applies: then the two
 * workspace arguments are ignored).  Same pruning order (descending score, ties by edge id) and the same order of every sum in
 * both forms.  The reference has no limit (code/Ob_propagation.py:161-185). */
int32_t rd_graph_beta_kept(int32_t E);
size_t rd_graph_beta_workspace_bytes(int32_t B, int32_t N, int32_t K, int32_t T, int32_t E);
int rd_graph_beta_fwd(int32_t B, int32_t N, int32_t K, int32_t T, int32_t d_ob, int32_t E, const float* V, const float* H,
                      const float* map_weights, const float* p_t, int64_t pt_bstride, const int64_t* edge_index,
                      int64_t row_stride, const float* edge_weights, int64_t w_bstride, float* out, int64_t* edge_index_out,
                      float* alpha_out, float* beta_save, int32_t* kept, void* workspace, size_t workspace_bytes, void* stream);
/* Backward: dout [B,N,K] -> dV [B,N,K], dH [B,N,T*32], dmap_part [B,N,16] (sum over B = d map_weights), dw [B,E] or NULL. */
int rd_graph_beta_bwd(int32_t B, int32_t N, int32_t K, int32_t T, int32_t d_ob, int32_t E, const float* V, const float* H,
                      const float* map_weights, const float* p_t, int64_t pt_bstride, const int64_t* edge_index,
                      int64_t row_stride, const float* edge_weights, int64_t w_bstride, const float* beta_save,
                      const int32_t* kept, const float* dout, float* dV, float* dH, float* dmap_part, float* dw, void* workspace,
                      size_t workspace_bytes, void* stream);
/* code/models_rd.py:345-346: distance = mean(cdist(alpha_all.T, alpha_all.T, p=2)) for alpha_all [E,B] (one column of edge
 * scores per sample); workspace B floats.  Identically 0 on the shipped path (equal columns); evaluated here in general. */
int rd_structure_distance(int32_t E, int32_t B, const float* alpha_all, float* workspace, float* distance, void* stream);

/* ---- building blocks of the paper-faithful sensor stage: Raindrop_v2(use_beta=True), i.e. code/models_rd.py:313-343 with the
 * literal at :317 flipped.  Layer 1 prunes a different edge set per sample, so the stage is composed instead of fused:
 *   rd_obs_embed_fwd -> rd_linear_fwd (lin_value + ReLU, increase_dim) -> rd_graph_beta_fwd -> rd_edge_softmax_list_batched ->
 *   rd_linear_fwd (layer 2's lin_value + ReLU) -> rd_rows_to_tokens_fwd; rd_pe_mask writes the PE columns and the mask.
 *  - rd_obs_embed_fwd: X[b,f,t*d+c] = dropout(relu(src[t,b,f] * R_u[f*d+c]))  ([B,F,T*d]; code/models_rd.py:290-296,326-327; same
 *    dropout site and element numbering as the fused stage).  rd_obs_embed_bwd: dR_u from dX (gate X > 0 = ReLU open AND kept).
 *  - rd_rows_to_tokens_fwd: z[t,b,f*d+c] = Y[b,f,t*d+c] * rowscale[b,f] (rowscale NULL = 1): the aggregate coefficient of layer 2
 *    (sum of the per-target softmax over a sample's surviving edges) and the layout change of code/models_rd.py:338-342;
 *    _bwd: dY = dz * rowscale in Y's layout. */
/* Scalar scale + nn.Dropout as a pure function of (seed + seed cell, site, element index): out = x * scale * keep / (1 - p)
 * (p = 0: the plain scale).  The backward is the same call on the gradient.  Sites < 16 are free for callers (the library's own sites start at 16; the observation embedding is 1).
 * The legacy `Raindrop` model's `encoder(src) * sqrt(d_model)` and input dropout (code/models_rd.py:131-134) run through it. */
int rd_scale_dropout(int64_t n, const float* x, float scale, float p_drop, uint64_t seed, uint32_t site, float* out, void* stream);

int rd_obs_embed_fwd(const rd_shape* s, const float* src, const float* R_u, float p_drop, uint64_t seed, float* X, void* stream);
size_t rd_obs_embed_bwd_workspace_bytes(const rd_shape* s);
int rd_obs_embed_bwd(const rd_shape* s, const float* src, const float* X, const float* dX, float p_drop, float* dR_u,
                     void* workspace, size_t workspace_bytes, void* stream);
int rd_rows_to_tokens_fwd(const rd_shape* s, const float* Y, const float* rowscale, float* z, int32_t ldz, void* stream);
int rd_rows_to_tokens_bwd(const rd_shape* s, const float* dz, int32_t ldz, const float* rowscale, float* dY, void* stream);

/* ---- host preprocessing on the device: code/utils_rd.py:149-257, code/Raindrop.py:215-231 ------------
 * Inputs are float64 (the reference's numpy arrays), outputs float32 (its torch.Tensor casts).  Every result is
 * bit-identical to the reference: elementwise steps use the same IEEE float64 operations in the same order, and the
 * statistics reproduce numpy's pairwise summation tree (raindrop_amd/csrc/rd_preprocess.hip). */

/* utils_rd.getStats (:149-161): P [NT, F] (any [N,T,F] flattened) -> mf [F], stdf [F] = mean / max(population std, 1e-7)
 * of the values > 0 of every sensor.  NT < 2^31. */
size_t rd_prep_stats_workspace_bytes(int64_t NT, int32_t F);
int rd_prep_stats(int64_t NT, int32_t F, const double* P, double* mf, double* stdf, void* workspace,
                  size_t workspace_bytes, void* stream);
/* utils_rd.mask_normalize (:164-175) + the float32 cast of tensorize_normalize (:232): P [N,T,F] f64 -> out f32 with the
 * normalised values in channels [0,F) and the observation mask in [F,2F).  layout 0: out [N,T,2F]; layout 1: out
 * [T,N,2F], i.e. the permute of code/Raindrop.py:232-238 fused into the store. */
int rd_prep_mask_normalize(int64_t N, int32_t T, int32_t F, const double* P, const double* mf, const double* stdf,
                           float* out, int32_t layout, void* stream);
/* utils_rd.mask_normalize_static (:206-219) + float32 cast: S [N,D] f64 -> out [N,D] f32. */
int rd_prep_static(int64_t N, int32_t D, const double* S, const double* ms, const double* ss, float* out, void* stream);
/* `torch.Tensor(P_time) / 60.0` (:235,253): minutes [N,T] f64 -> hours f32, [N,T] (layout 0) or [T,N] (layout 1). */
int rd_prep_time(int64_t N, int32_t T, const double* minutes, float* hours, int32_t layout, void* stream);
/* Setting 2 / 3 (code/Raindrop.py:215-231): zero k value channels of every sample of P f32 ([N,T,2F] layout 0 or
 * [T,N,2F] layout 1) in place; idx int32 [N,k] (per_sample = 1: the per-patient np.random.choice draws, made by the
 * host in the reference's order) or [k] (per_sample = 0: the same top-ranked set for every sample). */
int rd_prep_remove_features(int64_t N, int32_t T, int32_t F, float* P, const int32_t* idx, int32_t k, int32_t per_sample,
                            int32_t layout, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAINDROP_HIP_H */
