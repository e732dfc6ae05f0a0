/* segx.h -- C ABI of libsegx.so: the MI355X (gfx950) hot-path kernels of the Segtran `--net segtran`
 * train step.  This is the drop-in boundary: every entry point replaces an ATen
 * dispatch the reference performs (file:line cited per function, relative to /root/reference/code).
 *
 * Conventions
 *  - plain device pointers + sizes; fp32 everywhere (the reference computes in fp32).
 *  - the library never allocates / frees / retains device memory: caller passes outputs + workspace.
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*) and is re-entrant across streams and threads (the
 *    main thread and autograd's backward thread call in concurrently).  The library's only process-wide state is CONFIGURATION a caller
 *    sets through segx_tune / segx_set_rng_base -- atomics that no compute call ever writes (defaults such as the tile engine, tuning
 *    knobs whose every setting gives identical results, one launch counter) -- plus the per-thread error text; what a call should do
 *    differently from the defaults travels in its arguments (segx_gemm_desc.engine / .tile / .splitk).
 *  - return 0 = ok, <0 = invalid argument (see segx_last_error), >0 = HIP error code.
 */
#ifndef SEGX_H
#define SEGX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int segx_version(void);
/* copies the calling thread's last error message; returns its length */
int segx_last_error(char* buf, int buflen);

/* ---------------------------------------------------------------------------------------------
 * Batched strided fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, H1):
 *      C[z][m][n] = epilogue( alpha * sum_k A[z][m][k] * B[z][n][k] + bias )
 * z = z0*nb1 + z1 walks two batch dims with independent strides (0 = broadcast), so per-mode /
 * per-sample views of [B,N,M*d] tensors need no transposed copies (the reference bounces
 * [B,M*F,U] <-> [B,M,U,F], networks/segtran_shared.py:416-419,449).  One of (a_m,a_k) and one of
 * (b_n,b_k) must be 1.  Replaces: nn.Linear query/key :559-560, first_linear :414,
 * shared_linear :243, Conv1d group_linear :267, torch.matmul :566 and :447, every 1x1(x1) conv
 * (segtran2d.py:245,287,304,427; segtran3d.py:300,348,367,490; efficientnet/model.py:96,113,280;
 * aj_i3d.py:92 for 1x1x1 kernels) and all of their backward GEMMs.
 * ------------------------------------------------------------------------------------------- */
enum { SEGX_EPI_NONE = 0, SEGX_EPI_GELU = 1 /* aux = pre-activation, C = dropout(gelu(.)), :244-245 */ };
enum { SEGX_BIAS_NONE = 0, SEGX_BIAS_N = 1 /* bias[n] */, SEGX_BIAS_M = 2 /* bias[m] */ };
/* tile engines (segx_tune knob 4): SEGX_ENGINE_F32 = v_mfma_f32_32x32x2_f32 on fp32 operands (bit-for-bit a k-ordered fmaf chain);
 * SEGX_ENGINE_BF16X6 = fp32 operands split in registers into three bf16 planes, six v_mfma_f32_32x32x16_bf16 per block (fp32-equivalent:
 * error vs fp64 1.2e-6 against 1.0e-6), used for float4-legal operands with more than 48 rows on both sides; everything else stays on F32 */
enum { SEGX_ENGINE_F32 = 0, SEGX_ENGINE_BF16X6 = 1 };
/* per-call engine selector of segx_gemm_desc.engine: 0 (a zero-initialised desc) = the process default set by segx_tune knob 4 */
enum { SEGX_ENGINE_SEL_DEFAULT = 0, SEGX_ENGINE_SEL_F32 = 1, SEGX_ENGINE_SEL_BF16X6 = 2 };
enum { SEGX_TILE_AUTO = 0, SEGX_TILE_128x128 = 1, SEGX_TILE_64x64 = 2, SEGX_TILE_128x32 = 3, SEGX_TILE_32x128 = 4, SEGX_TILE_64x128 = 5,
       SEGX_TILE_256x128 = 6, SEGX_TILE_WS128x128 = 7, SEGX_TILE_WS128x256 = 8, SEGX_TILE_WS64x256 = 9,
       SEGX_TILE_WS96x256 = 10, SEGX_TILE_WS256x96 = 11,  /* 96-row side: k-contiguous operand only (A for 96x256, B for 256x96); other layouts quietly take 128x128 */
       /* 6..11: bf16x6 engine only -- the wave-specialised persistent kernels of gemm_x6ws.h (M x N of the workgroup tile) */
       SEGX_TILE_SKINNY_NT = 12  /* what segx_gemm_plan returns for a batch_reduce product of two k-contiguous operands, one of <= 32 rows and one of <= 192, over a long K
                                    (the weight gradients of the backbone's first pointwise convolutions, efficientnet/model.py:96, 113): one streaming pass, the batch summed
                                    inside the kernel, splitk x nbatch slabs (gemm_skinny.hip); any other call that names it quietly takes the planner's tile */ };
typedef struct {
    int32_t M, N, K, nb0, nb1;
    int64_t a_b0, a_b1, a_m, a_k;
    int64_t b_b0, b_b1, b_n, b_k;
    int64_t c_b0, c_b1, c_m;          /* C[m][n]: n contiguous */
    float alpha;
    int32_t epilogue, bias_mode;
    int64_t bias_b1;                  /* bias stride over z1 (grouped / per-mode biases) */
    const float* bias;
    float* aux;                       /* SEGX_EPI_GELU: same layout as C */
    float* gmax;                      /* optional: *gmax = max(*gmax, max(0, C)) (score clip flag, :570-580) */
    float dropout_p;                  /* SEGX_EPI_GELU only */
    uint64_t seed, offset;            /* Philox stream for the dropout mask */
    int32_t splitk;                   /* >1: K split over `splitk` slabs in `workspace`, then reduced */
    float* workspace;                 /* splitk*nb0*nb1*M*N floats when splitk>1 */
    int32_t tile;                     /* workgroup tile: SEGX_TILE_AUTO or one of SEGX_TILE_* (a tuning knob; results are identical) */
    int64_t bias_b0;                  /* bias stride over z0 (a bias vector per (z0, z1): the key-side term of re-associated scores) */
    int32_t batch_reduce;             /* 1: C [M][N] = alpha * sum over ALL nb0*nb1 batch members (and split-K slabs) of A_z B_z^T (+ bias): the gradient of an
                                       * operand shared by the batch (conv weights) in ONE deterministic reduction; needs workspace =
                                       * max(1, splitk)*nb0*nb1*M*N floats, EPI_NONE, no gmax; c_b0 / c_b1 are ignored */
    int32_t engine;                   /* SEGX_ENGINE_SEL_*: the tile engine of THIS call (0 = process default) */
    const void* b_planes;             /* optional: the B operand split ahead of time by segx_x6_presplit(B, N, K, b_n, b_k, nb0, nb1, b_b0, b_b1, ...) -- a
                                       * speed hint for a B that many row tiles share (a weight matrix: otherwise every workgroup that stages it splits
                                       * it again).  Used by the wave-specialised 256x128 / 128x256 bf16x6 kernels with a plain epilogue; every other
                                       * path reads B itself, which must stay valid.  Results are bit-identical either way. */
    int64_t bp_b0, bp_b1;             /* batch strides of b_planes in bf16 elements (3*N*K per matrix; 0 = shared) */
    const float* resid;               /* optional: C = alpha * A B^T (+ bias) + resid, resid laid out exactly like C (c_b0, c_b1, c_m).  Plain epilogue only
                                       * (no GELU, no gmax, no batch_reduce); with split-K it is added by the slab reduction.  Used for the gradient of a
                                       * tensor with two consumers -- the input of an MBConv block feeds the expansion convolution AND the skip connection
                                       * (efficientnet/model.py:96,118-122): dX = W^T dY + d(skip) in ONE launch instead of a GEMM and an add */
} segx_gemm_desc;
int segx_gemm_f32(const float* A, const float* B, float* C, const segx_gemm_desc* d, void* stream);
/* The library's own choice of workgroup tile and split-K factor for this problem (desc fields tile / splitk / workspace are
 * ignored): callers that want split-K size the workspace from *splitk and pass both back through the desc. */
int segx_gemm_plan(const float* A, const float* B, const segx_gemm_desc* d, int* tile, int* splitk);
/* The cost model's pick alone, without the table of measured choices segx_gemm_plan consults first (csrc/gemm_tuned.h): what tools/tune_gemm.py compares
 * its device sweep with when it decides which shapes need a table entry.  Not used on the step. */
int segx_gemm_plan_model(const float* A, const float* B, const segx_gemm_desc* d, int* tile, int* splitk);
/* The three-plane bf16 image of an fp32 operand for segx_gemm_desc.b_planes (bf16x6 engine): W is nb0 x nb1 matrices of rows x K floats
 * with element strides (s_b0, s_b1, s_row, s_k), one of s_row / s_k equal to 1, K % 8 == 0; planes receives, per matrix, [3][rows][K] bf16
 * (x = hi + mid + lo, each step rounded to nearest even: the split the kernels otherwise do in registers), matrices in (z0, z1) order:
 * segx_x6_presplit_elems(rows, K, nb0, nb1) 2-byte elements, 16-byte aligned.  The reference has no counterpart (its GEMMs are torch.matmul,
 * segtran_shared.py:414-449, 559-567); this removes the per-workgroup re-split of shared weights. */
int64_t segx_x6_presplit_elems(int rows, int K, int nb0, int nb1);
int segx_x6_presplit(const float* W, int rows, int K, int64_t s_row, int64_t s_k, int nb0, int nb1, int64_t s_b0, int64_t s_b1, void* planes, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Row kernels of the Squeeze-and-Expansion transformer (tokens.hip).  All tensors fp32, row-major,
 * row width a multiple of 4 (<= 4096).  Dropout masks come from a Philox4x32 counter stream
 * (seed, offset) with offset % 4 == 0 and are regenerated in backward (never stored).
 * ------------------------------------------------------------------------------------------- */
/* softmax over the last axis of S[rows, L]; clamps S to +-clip first iff gmax && *gmax > clip
 * (conditional clip N5, networks/segtran_shared.py:578-580), softmax :601, attention dropout :605.
 * P = probabilities (kept for backward); Pdrop = dropout(P) when p > 0 (else unused / may be NULL). */
int segx_softmax_fwd(const float* S, float* P, float* Pdrop, int64_t rows, int L, float clip, const float* gmax,
                     float p, uint64_t seed, uint64_t offset, void* stream);
int segx_softmax_bwd(const float* P, const float* dPdrop, const float* S /* for the clamp mask, may be NULL */, float* dS,
                     int64_t rows, int L, float clip, const float* gmax, float p, uint64_t seed, uint64_t offset, void* stream);
/* Sliding positional biases (SlidingPosBiases2D/3D, :1002-1175) as a relative-offset lookup that is never materialised:
 * out = clamp_if(*gmax > clip)(S) + weight * table[pos(j) - pos(i) + R] for the nmat score matrices S [nmat, N, N]
 * (:578-580 then :590-592).  geom (int32[5]) = {D, H, W, R, nd}: token grid (D = 1 in 2-D), radius, 2 or 3 position dims.
 * bwd: dtable [(2R+1)^nd]; dS = clamp-masked dOut (pass S = dS = NULL when the clamp was inactive: dS == dOut). */
int segx_posbias_fwd(const float* S, float* out, const float* table, int64_t nmat, int N, const int* geom, float weight, float clip,
                     const float* gmax, void* stream);
int segx_posbias_bwd(const float* dOut, const float* S, float* dS, float* dtable, int64_t nmat, int N, const int* geom, float weight,
                     float clip, void* stream);
/* nn.LayerNorm(eps=1e-12) (:262,:361 affine; :889 non-affine: w=b=NULL).  mean/rstd [rows] saved for backward. */
int segx_layernorm_fwd(const float* X, const float* w, const float* b, float* Y, float* mean, float* rstd,
                       int64_t rows, int C, float eps, void* stream);
int segx_layernorm_bwd(const float* dY, const float* X, const float* w, const float* mean, const float* rstd, float* dX,
                       int64_t rows, int C, void* stream);
/* deterministic two-stage column reductions; ws needs segx_colreduce_ws_floats(rows, C, nout) floats */
int64_t segx_colreduce_ws_floats(int64_t rows, int64_t C, int nout);
int segx_colsum(const float* X, float* out, float* ws, int64_t rows, int64_t C, void* stream);            /* nout = 1 */
int segx_ln_param_grad(const float* dY, const float* X, const float* mean, const float* rstd, float* dw, float* db,
                       float* ws, int64_t rows, int C, void* stream);                                      /* nout = 2 */
int segx_sum(const float* x, int64_t n, float* out, float* ws /* >= 1024 floats */, float scale, void* stream);
/* out[r] = sum_s X[r][s]: conv-style (per output channel) bias gradients on NC[D]HW tensors */
int segx_rowsum(const float* X, float* out, int64_t R, int64_t S, void* stream);
/* SegtranFusionEncoder.forward per-layer prologue (:916-946):
 *   Y = mask * dropout( LN_noaffine( LN_affine(X; w1,b1) + pos_weight * pos[n, :C] ) ),  X [B,N,C], pos [N,pos_ld], mask [B*N]
 * pos == NULL ('bias' positional codes, :937-940): Y = mask * dropout(LN_affine(X)), no second norm.
 * stats = 4*B*N floats.  Backward returns dX and dU (grad wrt LN_affine's output): dpos = pos_weight * sum_b dU,
 * (dw1, db1) = segx_ln_param_grad(dU, X, stats[0], stats[1]). */
int segx_prenorm_fwd(const float* X, const float* w1, const float* b1, const float* pos, int64_t pos_ld, float pos_weight,
                     const float* mask, float* Y, float* stats, int64_t B, int N, int C, float eps,
                     float p, uint64_t seed, uint64_t offset, void* stream);
int segx_prenorm_bwd(const float* dY, const float* X, const float* w1, const float* b1, const float* pos, int64_t pos_ld,
                     float pos_weight, const float* mask, const float* stats, float* dX, float* dU, int64_t B, int N, int C,
                     float p, uint64_t seed, uint64_t offset, void* stream);
/* The same backward with everything the host formed FROM dU computed in the kernel, so that dU is never written (round 6): dw / db = the LayerNorm-1 parameter
 * gradients (sum dU * xhat1, sum dU over all B N rows; autograd over :916-946), dsum [N, C] = sum_b dU[b] (the positional code's gradient before the pos_weight scale;
 * NULL allowed when pos == NULL).  A wave owns a token and walks the batch; ws: segx_prenorm_bwd_all_ws_floats(N, C) floats; C <= 2048. */
int64_t segx_prenorm_bwd_all_ws_floats(int N, int C);
int segx_prenorm_bwd_all(const float* dY, const float* X, const float* w1, const float* b1, const float* pos, int64_t pos_ld, float pos_weight,
                         const float* mask, const float* stats, float* dX, float* dsum, float* dw, float* db, float* ws, int B, int N, int C,
                         float p, uint64_t seed, uint64_t offset, void* stream);
/* LearnedSinuPosEmbedder.forward (:989-998) for the batch-invariant [N, pd] normalised coordinates:
 *   out = LN_noaffine(interleave(sin(z_even), cos(z_odd))), z = posn Wp^T + bp.  stats = 2*N floats.
 * Backward gives dZ [N,C]; dWp = dZ^T posn (segx_gemm_f32), dbp = segx_colsum(dZ). */
int segx_posembed_fwd(const float* posn, const float* Wp, const float* bp, float* out, float* stats, int64_t N, int C, int pd,
                      float eps, void* stream);
int segx_posembed_bwd(const float* dOut, const float* posn, const float* Wp, const float* bp, const float* stats, float* dZ,
                      int64_t N, int C, int pd, void* stream);
/* Expansion tail: MMPrivateOutput dropout + LayerNorm (:273-274) and LearnedSoftAggregate (:318-325) fused.
 *   Z [Mo, R, F] (mode-major) -> Y [R, F];  stats = 3*Mo*R floats (mean, rstd, mode probability).
 * Backward: dZ + dscore [Mo*R]; parameter grads via segx_modes_aggr_param_grad (ws: nout = 3), dba = sum(dscore).
 * lnw == lnb == NULL: no LayerNorm -- the soft aggregate runs on the raw mode features (ExpandedFeatTrans without FFN, :452-457). */
int segx_modes_aggr_fwd(const float* Z, const float* lnw, const float* lnb, const float* wa, const float* ba, float* Y, float* stats,
                        int Mo, int64_t R, int F, float eps, float p, uint64_t seed, uint64_t offset, void* stream);
int segx_modes_aggr_bwd(const float* dY, const float* Z, const float* lnw, const float* lnb, const float* wa, const float* stats,
                        float* dZ, float* dscore, int Mo, int64_t R, int F, float p, uint64_t seed, uint64_t offset, void* stream);
int segx_modes_aggr_param_grad(const float* dY, const float* Z, const float* lnw, const float* lnb, const float* wa,
                               const float* stats, const float* dscore, float* dlnw, float* dlnb, float* dwa, float* ws,
                               int Mo, int64_t R, int F, float p, uint64_t seed, uint64_t offset, void* stream);
/* The whole backward of the expansion tail in ONE pass over Z and dY (round 6): dZ, dscore AND the three parameter gradients (dlnw, dlnb, dwa; the reference
 * gets them from autograd over LayerNorm :273-274 and LearnedSoftAggregate :318-325).  Four modes and F <= 2048: every wave adds its tokens' column terms into
 * LDS vectors of its own, a workgroup writes one chunk of ws, a second launch adds the chunks in a fixed order (deterministic; no second read of Z, no regenerated
 * dropout mask).  Anything else runs segx_modes_aggr_bwd + segx_modes_aggr_param_grad.  ws: segx_modes_aggr_bwd_all_ws_floats(Mo, R, F) floats. */
int64_t segx_modes_aggr_bwd_all_ws_floats(int Mo, int64_t R, int F);
int segx_modes_aggr_bwd_all(const float* dY, const float* Z, const float* lnw, const float* lnb, const float* wa, const float* stats,
                            float* dZ, float* dscore, float* dlnw, float* dlnb, float* dwa, float* ws, int Mo, int64_t R, int F,
                            float p, uint64_t seed, uint64_t offset, void* stream);
/* backward of the GELU(+dropout) epilogue of segx_gemm_f32 (MMSharedMid :244-245): dT = dH * keep * gelu'(T) */
int segx_gelu_bwd(const float* dH, const float* T, float* dT, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);
/* the same for dH, T, dT [rows, N] with colsum[n] = sum_r dT[r][n] from the same pass -- the bias gradient of the nn.Linear in front of the GELU (MMSharedMid :244), which
 * autograd would take from a second pass over dT.  N % 4 == 0, N <= 2048, offset % 4 == 0, 16-byte aligned; ws: segx_gelu_bwd_colsum_ws_floats(rows, N) floats. */
int64_t segx_gelu_bwd_colsum_ws_floats(int64_t rows, int N);
int segx_gelu_bwd_colsum(const float* dH, const float* T, float* dT, float* colsum, float* ws, int64_t rows, int N, float p, uint64_t seed, uint64_t offset,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Train-step glue (train.hip)
 * ------------------------------------------------------------------------------------------- */
/* loss = (1-dice_w) * BCEWithLogits(pos_weight)(logits, mask) + dice_w * sum_c class_w[c] * mean_b dice_loss_indiv
 * (train2d.py:1233-1242,1314-1318; train3d.py:738-756; utils/losses.py:47-60).  logits/mask [B, C, S] fp32 at mask
 * resolution.  out[0]=loss, out[1]=ce, out[2]=dice_total, out[3+c]=dice_c.  ws: segx_loss_ws_floats(B, C) floats,
 * must be kept untouched between fwd and bwd.  bwd: dlogits = grad_out[0] * dloss/dlogits. */
int64_t segx_loss_ws_floats(int B, int C);
int segx_seg_loss_fwd(const float* logits, const float* mask, const float* pos_weight, const float* class_w, float* out,
                      float* ws, int B, int C, int64_t S, float dice_w, void* stream);
int segx_seg_loss_bwd(const float* logits, const float* mask, const float* pos_weight, const float* class_w, const float* ws,
                      const float* grad_out, float* dlogits, int B, int C, int64_t S, float dice_w, void* stream);
/* One optimizer step for ALL parameter tensors in three launches: nn.utils.clip_grad_norm_(all, max_global_norm)
 * (train2d.py:1324-1325) followed by BertAdam.step (optimization.py:90-164: per-tensor clip max_tensor_norm, Adam
 * moments without bias correction, decoupled weight decay, lr * sched).  Device tables (built once by the caller):
 * params/grads/m/v = arrays of device pointers [ntensors]; sizes [ntensors]; tensors cut into `chunk`-element chunks:
 * chunk_tensor/chunk_off [nchunks], chunk_first [ntensors+1] (chunks of a tensor are consecutive); active[t] = 0 for
 * parameters that never receive a gradient (N3: skipped, no weight decay).  ws: nchunks + 2*ntensors + 2 floats;
 * afterwards ws[nchunks + 2*ntensors] = global grad norm. */
int segx_mt_bertadam_step(void* const* params, const void* const* grads, void* const* m, void* const* v, const int64_t* sizes,
                          const int* chunk_tensor, const int64_t* chunk_off, const int* chunk_first, const int* active,
                          const float* lr, const float* wd, int ntensors, int nchunks, int chunk,
                          float max_global_norm, float max_tensor_norm, float sched, float b1, float b2, float eps,
                          float* ws, void* stream);

/* Multi-tensor gather (data parallel): copies chunks [chunk_begin, chunk_begin + nchunks) of the tensors src[t] (this step's
 * gradient tensors, owned by autograd) into the flat slices dst[t] (their all-reduce bucket) in ONE launch -- replaces the
 * per-parameter `grad += g` kernels of a view-based bucket.  Tables as in segx_mt_bertadam_step; src[t] == NULL leaves the slice. */
int segx_mt_gather(const void* const* src, void* const* dst, const int64_t* sizes, const int* chunk_tensor, const int64_t* chunk_off,
                   int chunk_begin, int nchunks, int chunk, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backbone kernels (backbone.hip): BatchNorm(+activation), depthwise convolution, squeeze-excite plane ops.
 * Tensors are NC[D]HW fp32; S = product of the spatial dims; a (sample, channel) plane is contiguous.
 * act: 0 none, 1 swish (efficientnet/utils.py:64-79), 2 ReLU (aj_i3d.py:95-96), 3 LeakyReLU(0.2) (networks/discriminator.py:14-21).
 * ------------------------------------------------------------------------------------------- */
/* scratch of the BatchNorm backward reductions: segx_bn_ws_floats(B, C, S) floats (S = 0: the two-launch forms only -- what segx_bn_act_bwd_reduce / _apply need) */
int64_t segx_bn_ws_floats(int B, int C, int64_t S);
/* Backward of BatchNorm (+ activation): dX, dw[C], db[C]; training != 0 differentiates through the batch statistics.  One process: segx_bn_act_bwd2 (below).
 * gate / dpool ([B*C]; gate needs dpool, dpool alone = gate of one): a squeeze-excite gate multiplies the BatchNorm output (Z = Y * gate[b][c],
 * efficientnet/model.py:110) and dY is the gradient w.r.t. Z: the kernels then use dY * gate[b][c] + dpool[b][c] * inv_S in place of dY (dpool = gradient
 * w.r.t. the pooled sums' mean), which replaces a pass that would write that tensor. */
/* the two halves of the backward pass for synchronised BatchNorm (nn.SyncBatchNorm, train2d.py:1109): reduce gives the LOCAL sums dw = sum du*xhat,
 * db = sum du; after an all-reduce of both, apply uses the GLOBAL sums and inv_n = 1 / (global element count); dc_p / seed / offset: the
 * drop_connect scale of segx_bn_act_fwd2 (0: none) */
int segx_bn_act_bwd_reduce(const float* dY, const float* X, const float* mean, const float* var, const float* w, const float* b,
                           float* dw, float* db, float* ws, int B, int C, int64_t S, float eps, int act,
                           const float* gate, const float* dpool, float inv_S, float dc_p, uint64_t seed, uint64_t offset, void* stream);
int segx_bn_act_bwd_apply(const float* dY, const float* X, const float* mean, const float* var, const float* w, const float* b,
                          const float* sum_dw, const float* sum_db, float* dX, int B, int C, int64_t S, float eps, int act,
                          float inv_n, const float* gate, const float* dpool, float inv_S, float dc_p, uint64_t seed, uint64_t offset, void* stream);
/* r04 -- training-mode BatchNorm in two launches (statistics partials, then ONE pass that finishes the statistics, normalises, activates and
 * optionally pools / scales / adds the skip input).  Replaces the nn.BatchNorm2d/3d forward of efficientnet/model.py:96-116 and aj_i3d.py:65-97
 * together with the ops around it in an MBConv block (swish :98,:102; squeeze-excite pooling :106; drop_connect + skip add :118-122).
 *   parts: [C][nparts] float4 records (n, mean, M2, -) of disjoint runs covering the B*S elements of each channel; Chan et al.'s merge makes the
 *   result independent of how a producer cut the data (nparts > 0: a producer's own partials [C][nparts]; the buffer then needs C*4 more
 *   floats behind them when nparts > 256).  The buffer holds segx_bn_parts_floats(B, C, S) floats, 16-byte aligned.
 *   nparts == 0 with parts given = AUTO: the library computes the statistics itself, using `parts` as scratch -- in ONE launch for the whole
 *   layer when a channel's B planes fit one team's registers (S <= 4096 floats, B <= 8: "channel-resident", 66 of EfficientNet-B4's 96 layers at
 *   512 x 512 with the stride-1 stem); in ONE launch by a TEAM of B x chunks workgroups per channel for larger planes with S % 4 == 0 (each keeps its
 *   chunk in registers while the partials are exchanged: 1 read + 1 write; forward teams up to 64 workgroups, backward up to 128; segx_tune knob 3);
 *   else a statistics-partials launch + the folding apply pass.  segx_bn_pool_chunks(B, S, auto) = chunks per plane written to psum.
 *   parts == NULL: mean / var are INPUTS (running statistics: eval mode / synchronised BatchNorm after the merge); otherwise they are OUTPUTS
 *   (saved for the backward pass) and run_mean / run_var (optional) are updated with momentum and the unbiased variance.
 *   psum (optional): [B*C][segx_plane_chunks(S)] partial sums of Y per plane (the squeeze-excite pooling; segx_se_fwd2 adds the chunks up).
 *   resid (optional): Y = act(bn(X)) * dcs[sample] + resid, dcs = drop_connect keep scale of the sample (0 or 1/(1-dc_p), Philox element
 *   `sample` of stream (seed, offset): efficientnet/utils.py:129-154); dc_p = 0: plain skip add. */
int64_t segx_plane_chunks(int64_t S);
int64_t segx_bn_pool_chunks(int B, int64_t S, int auto_stats);
int64_t segx_bn_parts_floats(int B, int C, int64_t S);
/* synchronised BatchNorm (nn.SyncBatchNorm, train2d.py:1109), local half: ONE partial (n, mean, M2) per channel of this process's batch into part [C]
 * float4 (ws: segx_bn_parts_floats(B, C, S) floats of scratch).  The ranks all-gather their partials into [ranks][C] float4 and hand them to
 * segx_bn_act_fwd2 with nparts = -ranks: the apply pass merges them (Chan) itself and updates the running statistics -- no merge launch. */
int segx_bn_stats_local(const float* X, float* part, float* ws, int B, int C, int64_t S, void* stream);
int segx_bn_act_fwd2(const float* X, const float* parts, int nparts, float* mean, float* var, float* run_mean, float* run_var, float momentum,
                     const float* w, const float* b, float* Y, float* psum, const float* resid, float dc_p, uint64_t seed, uint64_t offset,
                     int B, int C, int64_t S, float eps, int act, int64_t parts_floats, int64_t y_bs, void* stream);
/* y_bs (r05): 0 = Y is dense [B][C][S]; else Y is a channel slice of a wider [B][Ctot][S] tensor and y_bs its batch stride in floats (>= C * S, a multiple of 4;
 * S % 4 == 0, Y 16-byte aligned): the branch of a channel concatenation (aj_i3d.py:118, the Inception module's torch.cat) written where it belongs.
 * parts_floats / ws_floats (r05): the floats the caller's `parts` / `ws` buffer holds.  Both calls re-derive what they need under the knob settings in force
 * AT THE CALL and refuse a smaller buffer (a knob-3 change between sizing and launch used to write past it).
 * TEAM FORM, failure behaviour (r05): a team is at most segx_team_cap() workgroups (half the compute units the runtime reports, at most 128) and its kernels
 * are checked for >= 2 workgroups per compute unit at first use; the inter-workgroup polls are bounded (segx_tune knob 12), and a poll that expires adds one to
 * a process-wide error word in pinned host memory AND turns the exchanged statistics into NaN.  segx_team_status(clear) returns the number of expired polls
 * since the last clear without synchronising the device; the Python host raises RuntimeError on a non-zero count at every optimizer step
 * (segtran_amd/segx.py: team_check).  Replaces nn.BatchNorm2d of efficientnet/model.py:43, 98, 102, 115, which is never silently wrong. */
int segx_team_status(int clear);
int segx_team_cap(void);
/* diagnostics (tests of the team form under contention): `wgs` workgroups of 256 threads holding 80 KB (heavy != 0) or 64 bytes of LDS each for `ms` milliseconds on `stream` */
int segx_occupy(int wgs, int heavy, float ms, float* sink, void* stream);
/* backward of segx_bn_act_fwd2 in two launches (the apply pass sums the reduction partials itself): as segx_bn_act_bwd, plus the drop_connect scale of
 * the forward (same dc_p / seed / offset), which multiplies dY; the gradient w.r.t. resid is dY itself.  ws: segx_bn_ws_floats(B, C, S) floats.
 * training != 0 and a channel-resident or team shape (see segx_bn_act_fwd2): ONE launch (x and dy read once).  dy_bs: batch stride of dY in floats (0 = dense, C * S): the gradient of
 * one operand of a channel concatenation (an Inception module's branches, aj_i3d.py:139-141) is read in place from the concatenation's gradient. */
int segx_bn_act_bwd2(const float* dY, const float* X, const float* mean, const float* var, const float* w, const float* b,
                     float* dX, float* dw, float* db, float* ws, int B, int C, int64_t S, float eps, int act, int training,
                     const float* gate, const float* dpool, float inv_S, float dc_p, uint64_t seed, uint64_t offset, int64_t dy_bs, int64_t ws_floats, void* stream);
/* r04 -- the squeeze-excite excitation of an MBConv block (efficientnet/model.py:105-113) in 2 + 3 launches.
 * fwd: p = (sum of the nch pooling chunks psum[B*C][nch]) * inv_S; hpre = W1 p + b1; gate = sigmoid(W2 swish(hpre) + b2); and, when Wproj [M][C] is
 *      given, the gate folded into per-sample projection weights Wb[b][m][k] = Wproj[m][k] * gate[b][k] (exact re-association of
 *      efficientnet/model.py:110-113: project_conv(y * gate) == pointwise convolution of y with per-sample weights).
 * bwd: from dWb [B][M][C] (the per-sample weight gradient of the projection GEMM) -- or from dgate [B][C] when Wproj / dWb are NULL --:
 *      dpool (= dL/d pooled sum, already times inv_S), dW1, db1, dW2, db2 and dWproj[m][k] = sum_b dWb * gate.  ws: segx_se_ws2_floats(B, C, Cs) floats */
int segx_se_fwd2(const float* psum, int nch, float inv_S, const float* W1, const float* b1, const float* W2, const float* b2, const float* Wproj,
                 float* p, float* hpre, float* gate, float* Wb, int B, int C, int Cs, int M, void* stream);
int64_t segx_se_ws2_floats(int B, int C, int Cs);
int segx_se_bwd2(const float* dWb, const float* Wproj, const float* dgate, const float* gate, const float* hpre, const float* p, const float* W1,
                 const float* W2, float inv_S, float* dpool, float* dW1, float* db1, float* dW2, float* db2, float* dWproj, float* ws,
                 int B, int C, int Cs, int M, void* stream);
/* depthwise k x k convolution (k in {3,5}, stride in {1,2}) with explicit top/left zero padding (static 'same'
 * padding N6, efficientnet/utils.py:248-275): Y[b,c,oy,ox] = sum w[c,ky,kx] X[b,c,oy*s+ky-pad_t, ox*s+kx-pad_l] */
int segx_dwconv2d_fwd(const float* X, const float* W, float* Y, int B, int C, int H, int Wd, int OH, int OW, int k, int stride,
                      int pad_t, int pad_l, void* stream);
int segx_dwconv2d_bwd_data(const float* dY, const float* W, float* dX, int B, int C, int H, int Wd, int OH, int OW, int k,
                           int stride, int pad_t, int pad_l, void* stream);
/* partial weight gradients part[B * rows][C][k*k], rows = segx_dwconv2d_wgrad_rows(OH, OW) row strips per sample;
 * dW = segx_colsum over the B*rows rows (deterministic two-stage sum, no atomics) */
int64_t segx_dwconv2d_wgrad_rows(int OH, int OW);
/* Stride-1 'same' depthwise convolution (k = 3 / 5, pads (k - 1) / 2, rows of float4 multiples): the data gradient AND the weight-gradient partials from ONE pass over dY
 * (efficientnet/model.py:100-104 in backward; segx_dwconv2d_bwd_data + segx_dwconv2d_bwd_weight read dY twice).  part: [B * rows][C][k * k] with rows = segx_dwconv2d_bwd_fused_rows(...)
 * (0: shape not served -- use segx_dwconv2d_bwd_data + segx_dwconv2d_bwd_weight[_direct]); dW = segx_colsum(part) over its B * rows rows. */
int64_t segx_dwconv2d_bwd_fused_rows(int H, int Wd, int OH, int OW, int k, int stride, int pad_t, int pad_l);
int segx_dwconv2d_bwd_fused(const float* dY, const float* X, const float* W, float* dX, float* part, int B, int C, int H, int Wd, int OH, int OW,
                            int k, int stride, int pad_t, int pad_l, void* stream);
/* r04: the weight gradient dW [C][k*k] in ONE launch where a channel's B planes are one workgroup's work (B <= 8, <= 8192 outputs per plane, the float4
 * layout): returns 1 when done, 0 when the shape needs the two-stage form above (nothing launched), < 0 on error */
int segx_dwconv2d_bwd_weight_direct(const float* dY, const float* X, float* dW, int B, int C, int H, int Wd, int OH, int OW, int k,
                                    int stride, int pad_t, int pad_l, void* stream);
int segx_dwconv2d_bwd_weight(const float* dY, const float* X, float* part, int B, int C, int H, int Wd, int OH, int OW, int k,
                             int stride, int pad_t, int pad_l, void* stream);
/* squeeze-excite plane ops (efficientnet/model.py:105-110) on [planes = B*C, S]:
 * Y = X * gate[plane];  out[plane] = sum_s A*B   (the reference operation order of an MBConv block: MBConvBlock.gate_in_weights = False) */
int segx_plane_scale(const float* X, const float* gate, float* Y, int64_t planes, int64_t S, void* stream);
/* Y[p][s] = X[p][s] + bias[p % C]: the bias of a dense k x k convolution run on the implicit-GEMM engine (nn.Conv2d(..., 3, padding=1) of the
 * U-Net host, unet2d/unet_parts.py:16-20); in place allowed */
int segx_plane_bias_add(const float* X, const float* bias, float* Y, int64_t planes, int C, int64_t S, void* stream);
int segx_plane_dot(const float* A, const float* Bm, float* out, int64_t planes, int64_t S, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Feature-pyramid kernels (fpn.hip)
 * ------------------------------------------------------------------------------------------- */
/* nn.GroupNorm(G, C) on [B, C, S] (segtran2d.py:148-149,190-192; segtran3d.py:182-183,225-227).  mean/rstd: [B*G].
 * ws: segx_gn_ws_floats(B, C, G) floats. */
int64_t segx_gn_ws_floats(int B, int C, int G);
int segx_groupnorm_fwd(const float* X, const float* w, const float* b, float* Y, float* mean, float* rstd, float* ws,
                       int B, int C, int G, int64_t S, float eps, void* stream);
/* plane_dx_sums (r05; may be NULL): [B * C] floats = the sum over every (sample, channel) plane of the dX this call writes, in closed form from the plane
 * sums the backward computes anyway.  Where the normalised tensor was `conv1x1(...) + up(...)` (segtran3d.py:336-360, segtran2d.py:263-291) these are the
 * convolution's bias gradient: the row-sum pass over the full-resolution gradient need not run. */
int segx_groupnorm_bwd(const float* dY, const float* X, const float* w, const float* mean, const float* rstd, float* dX,
                       float* dw, float* db, float* ws, int B, int C, int G, int64_t S, float* plane_dx_sums, void* stream);
/* r05: GroupNorm statistics from partials left by the pass that WROTE the tensor.  segx_interp_linear_fwd_axis2_gn = segx_interp_linear_fwd_axis2 that also
 * reduces what it writes to (count, mean, M2) partials per run of a (sample, group): parts [outer / cpg][nparts] float4, nparts =
 * segx_interp_gn_nparts(n1_out * n2_out * inner / 4, cpg) (0 = shape not served: float4 per plane not a multiple of 256); segx_groupnorm_fwd_parts merges
 * them (Chan) and applies the normalisation.  Replaces the statistics pass of nn.GroupNorm over the up-sampled pyramid levels (segtran3d.py:338-360). */
int64_t segx_interp_gn_nparts(int64_t float4_per_plane, int cpg);
int segx_interp_linear_fwd_axis2_gn(const float* in, const float* base, float* out, int64_t outer, int n1_in, int n1_out, int n2_in, int n2_out,
                                    int64_t inner, int cpg, float* parts, int nparts, void* stream);
int segx_groupnorm_fwd_parts(const float* X, const float* parts, int nparts, const float* w, const float* b, float* Y, float* mean, float* rstd,
                             int B, int C, int G, int64_t S, float eps, void* stream);
/* r05: GroupNorm folded into a pointwise-convolution consumer (no apply pass forward, no plane-sums pass backward; the host composes per-sample weights
 * W * sc_b and biases W sh_b + bias from mean / rstd -- segtran_amd/functional.py: up_group_norm_conv).  segx_groupnorm_stats_parts: mean / rstd [BG] from
 * the partials of segx_interp_linear_fwd_axis2_gn.  segx_gn_fold_bwd: dX = Gd + A[b, g] + Bc[b, g] * xhat in one pass (Gd = the consumer's data gradient,
 * A / Bc [B * G] = the statistics' share of the chain rule); plane_sums [B * C] = sum of dX over every plane (the lateral's bias gradient); ws: B * C * 64 floats.
 * Replaces out_gn2b / out_gn3b + their consumers of segtran3d.py:336-367 on the re-associated path. */
int segx_groupnorm_stats_parts(const float* parts, int nparts, float* mean, float* rstd, int BG, float eps, void* stream);
/* segx_gn_fold_bwd_proj: the same pass where the consumer projects onto NC <= 8 channels (the class projection): Gd is not materialised, the kernel forms
 * sum_o Wb[b][o][c] * dOut[b][o][s] itself (dOut [B][NC][S], Wb [B][NC][C]) */
int segx_gn_fold_bwd_proj(const float* dOut, const float* Wb, int NC, const float* X, const float* mean, const float* rstd, const float* A, const float* Bc,
                          float* dX, float* plane_sums, float* ws, int B, int C, int G, int64_t S, void* stream);
int segx_gn_fold_bwd(const float* Gd, const float* X, const float* mean, const float* rstd, const float* A, const float* Bc, float* dX, float* plane_sums,
                     float* ws, int B, int C, int G, int64_t S, void* stream);
/* F.interpolate(mode='bilinear'|'trilinear', align_corners=False) from [planes, d, h, w] to [planes, D, H, W] (2-D: d = D = 1);
 * out = interp(in) (+ base, the FPN lateral, when base != NULL).  bwd is the exact adjoint, computed as a gather. */
/* PolyformerLayer glue (networks/polyformer.py:36-55): nn.AvgPool2d(2) on [planes, H, W] (+ adjoint) and the batched transpose
 * [batch, R, C] -> [batch, C, R] between channel-major feature maps and token-major rows */
int segx_avgpool2_fwd(const float* X, float* Y, int64_t planes, int H, int W, void* stream);
int segx_avgpool2_bwd(const float* dY, float* dX, int64_t planes, int H, int W, void* stream);
int segx_transpose(const float* X, float* Y, int64_t batch, int R, int C, void* stream);
/* nn.Dropout as its own pass (out-FPN output under --outdrop, segtran2d.py:308-310 / segtran3d.py:392-394):
 * y[i] = x[i] * keep(seed, offset + i) / (1 - p); the backward pass is the same call on dy.  offset % 4 == 0, 16-B aligned. */
int segx_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);
/* Device-side base of every dropout stream (library state, NULL = none): every kernel that takes (seed, offset) adds *base to offset.  It exists for
 * hipGraph capture of the train step -- a replayed graph re-issues the captured offsets; a captured segx_rng_advance(base, span) at the end of the step
 * moves the base by the number of stream positions the step reserved, so each replay draws fresh masks (forward and backward of ONE replay still agree). */
int segx_set_rng_base(const uint64_t* base);
int segx_rng_advance(uint64_t* base, uint64_t span, void* stream);

/* tuning / bisecting knobs (results are identical for every setting): knob 1 = interp_linear_fwd kernel (0 auto, 1 scalar, 2 float4 rows);
 * knob 4 = tile engine of segx_gemm_f32 and the
 * implicit-GEMM convolutions (SEGX_ENGINE_*: same results to fp32 rounding, see above); returns the previous value of knob 4;
 * knob 7: weight gradients of the packed 3-D convolutions on the bf16x6 engine -- 0 (default): where the loader reads whole rows (OW % 8 == 0, or OW % 4 == 0
 * at unit W stride) and the tile is not the strided 64-row case; 1: every one (also the per-position gather); 2: every whole-row case (also the strided 64-row tile);
 * knob 8 = outputs per strip of the depthwise weight gradient (default 8192; >= 256);
 * knob 6 = schedule variant of the bf16x6 kernels with IDENTICAL results (0 = product; 1 = raised wave priority in the MFMA phase / of the consumer
 * waves; 6 = split-early schedule, 7 = product schedule at two waves per SIMD); the ablation variants 2..5, whose results are NOT the GEMM, exist only
 * in -DSEGX_BENCH builds (tools/build_variant.py) and are rejected by the product library; knob 9 = workgroups of a persistent launch of the wave-specialised bf16x6 kernels (default 256 = one per CU; a multiple of 8);
 * knob 5 = number of launches that ran on the bf16x6 engine since the last query (resets the count);
 * knob 3 = form of training BatchNorm when the library computes the statistics itself (segx_bn_act_fwd2 with nparts == 0, segx_bn_act_bwd2): 0 (default) =
 * channel-resident where a channel fits one workgroup, else a TEAM of workgroups per channel (one launch, the slabs stay in registers across a team barrier),
 * else two launches; 1 = never teams; 2 = teams for every shape with S % 4 == 0 (tests).  Same results to fp32 summation order;
 * knob 12 = poll bound of a team exchange (default 2^20 polls, ~1 s; 32 .. 2^24); knob 13 = FAULT INJECTION for the tests of the team form's failure
 * path: the last `value` workgroups of every team launch are not launched, so their team mates time out (0 = off, the default);
 * knob 14 = slab-in-LDS form of the stride-1 3 x 3 x 3 'same' max-pools: 0 (default) where the four-cells-per-thread form does not apply (row length not a
 * multiple of 4), 1 wherever a slab fits the LDS, 2 never; knob 15 = the same pools with rows of a multiple of 4 floats: 1 (default) a thread keeps its four
 * cells for up to eight slices and slides along the depth (a third of the row loads), 0 = one slice per thread;
 * knob 19 = order in which the tiles of a GEMM are walked inside an XCD's run: 1 (default) M fastest where the A operand of a member fits the XCD's L2 (<= 2 MiB) and B is
 * the big operand -- pointwise convolutions over 10^5 positions: B crosses the fabric once instead of once per row of tiles --, 0 = N fastest always (rounds 1-5).
 * Identical results for every setting. */
int segx_tune(int knob, int value);
/* r06: the value a knob holds now (knobs 1-4, 6-9, 12-19; -1 for an unknown knob and for the counter knob 5): a caller that changes a process-wide default for its own
 * lifetime (dist.GradReducer: knob 3) reads it first and puts it back instead of assuming the default */
int segx_tune_get(int knob);
/* r04: TWO adjacent outer axes of a linear resampling in one streaming pass over [outer, n1, n2, inner] (inner % 4 == 0: the contiguous extent, read
 * and written as float4; align_corners = False): the y and z axes of the 3-D feature pyramid's trilinear up-sampling (segtran3d.py:304,319,351,364,384)
 * after the x pass, with the lateral added in the same pass; and the adjoint (z and y before the x pass).  Same blends in the same order as the
 * one-axis passes chained -> the same numbers, 21 instead of 29 coarse-tensor sizes of traffic forward, 13 instead of 21 backward. */
int segx_interp_linear_fwd_axis2(const float* in, const float* base, float* out, int64_t outer, int n1_in, int n1_out, int n2_in, int n2_out,
                                 int64_t inner, void* stream);
int segx_interp_linear_bwd_axis2(const float* dout, float* din, int64_t outer, int n1_out, int n1_in, int n2_out, int n2_in, int64_t inner, void* stream);
int segx_interp_linear_fwd(const float* in, const float* base, float* out, int64_t planes, int d, int h, int w, int D, int H, int W,
                           void* stream);
/* forward along ONE axis of a tensor viewed as [outer, n_in, inner] -> [outer, n_out, inner] (+ base).  Chained x -> y -> z it is
 * bit-identical to segx_interp_linear_fwd (same blends in the same order) and streams at HBM rate */
/* src_scale: source step per destination index; 0 -> n_in / n_out (F.interpolate(size=...)); explicit s > 0 reproduces
 * F.interpolate(scale_factor=1/s) on sizes s does not divide (Mince transformer, reference segtran_shared.py:47-66); s < 0 selects
 * align_corners=True with |s| = (n_in - 1) / (n_out - 1) (nn.Upsample(..., align_corners=True) of the U-Net host, unet2d/unet_parts.py:48) */
int segx_interp_linear_fwd_axis(const float* in, const float* base, float* out, int64_t outer, int n_in, int n_out, int64_t inner,
                                float src_scale, void* stream);
/* RandomResizedCrop of the 3-D trainer (dataloaders/datasets3d.py:611-665, train3d.py:713-715): resample X [planes, d, h, w] to (D, H, W)
 * (trilinear, align_corners=False), zero-pad, crop (od, oh, ow) voxels -- as one gather pass that only computes the cropped window.
 * geom (int32[12]) = {d, h, w, D, H, W, od, oh, ow, oz, oy, ox}; (oz, oy, ox) = crop start minus front pad, in the resampled grid */
int segx_resized_crop3d(const float* X, float* Y, int64_t planes, const int* geom, void* stream);
int segx_interp_linear_bwd(const float* dout, float* din, int64_t planes, int d, int h, int w, int D, int H, int W, void* stream);
/* ---------------------------------------------------------------------------------------------
 * Data augmentation on the device (augment.hip).  The reference transforms each sample on the CPU in DataLoader workers (numpy / imgaug /
 * torchvision); here a batch resident in HBM is transformed by one-pass kernels, the random parameters drawn on the host.
 * ------------------------------------------------------------------------------------------- */
/* Axis-permuting gather with zero padding: RandomRotFlip + RandomCrop of the 3-D trainer (dataloaders/datasets3d.py:547-579, 491-545) composed
 * into ONE pass; Fliplr / Flipud / Rot90 / CropAndPad / PadToFixedSize / CropToFixedSize of the 2-D pipeline (train_util.py:34-53) with I0 = O0 = 1.
 * X [planes, I0, I1, I2] -> Y [planes, O0, O1, O2]: Y[p][o] = X[p][i], i[src[a]] = sgn[a] > 0 ? o_a + off[a] : off[a] - o_a, 0 where i falls
 * outside.  geom (int32[15]) = {I0, I1, I2, O0, O1, O2, src0, src1, src2, sgn0, sgn1, sgn2, off0, off1, off2}; src is a permutation of (0, 1, 2) */
int segx_axis_gather(const float* X, float* Y, int64_t planes, const int* geom, void* stream);
/* nn.ConvTranspose2d(k = 2, s = 2) of the U-Net's bilinear=False decoder (networks/unet2d/unet_parts.py:53) = pointwise convolution onto
 * 4 Cout channels (segx_gemm_f32) + this re-arrangement: X [4 planes, h, w] -> Y [planes, 2h, 2w], Y[p][2i+a][2j+c] = X[4p+2a+c][i][j];
 * inverse = 1: the adjoint, X [planes, 2h, 2w] -> Y [4 planes, h, w] */
int segx_pixel_shuffle2(const float* X, float* Y, int64_t planes, int h, int w, int inverse, void* stream);
/* RandomNoise (datasets3d.py:581-597): Y = X + (clip(sigma z, -2 sigma, 2 sigma) + mu) * (nonzero_only ? X != 0 : 1), z ~ N(0, 1) -- `noise` [n]
 * when given (parity tests inject the reference's field), else Box-Muller on the Philox stream (seed, offset); offset % 4 == 0 */
int segx_add_noise(const float* X, const float* noise, float* Y, int64_t n, float mu, float sigma, int nonzero_only, uint64_t seed, uint64_t offset,
                   void* stream);
/* iaa.Resize / the keep_size resize of iaa.CropAndPad (train_util.py:31-37), cv2.resize conventions: X [planes, h, w] -> Y [planes, H, W];
 * mode 0 nearest (segmentation maps), 1 bilinear, 2 bicubic (A = -0.75, replicated border: imgaug's default for images); quantize = 1 rounds and
 * clamps to [0, 255] (the uint8 image the reference carries between augmenters) */
int segx_resize2d(const float* X, float* Y, int64_t planes, int h, int w, int H, int W, int mode, int quantize, void* stream);
/* transforms.ColorJitter / iaa.Grayscale(alpha) on channels-first RGB X [B, 3, HW] (train_util.py:53-60): Y = f X + (1 - f) D with a per-sample
 * factor[B]; mode 0 brightness (D = 0), 1 contrast (D = pivot[b] = mean luma, segx_gray_mean), 2 saturation (D = luma), 3 grayscale-alpha
 * (f = 1 - alpha, D = luma); luma = 0.299 R + 0.587 G + 0.114 B; quantize = 1: uint8 rounding as PIL / imgaug apply it */
int segx_color_blend(const float* X, float* Y, int B, int64_t HW, int mode, const float* factor, const float* pivot, int quantize, void* stream);
int64_t segx_gray_mean_ws_floats(int B, int64_t HW);
int segx_gray_mean(const float* X, float* mean, float* ws, int B, int64_t HW, int quantize, void* stream);
/* transforms.ToTensor + Normalize (train_util.py:101-105): Y[b][c] = (X[b][c] * scale - mean[c]) / std[c] */
int segx_normalize(const float* X, float* Y, int B, int C, int64_t HW, float scale, const float* mean, const float* std, void* stream);

/* separable form: adjoint along ONE axis of a tensor viewed as [outer, n_out, inner] -> [outer, n_in, inner] */
int segx_interp_linear_bwd_axis(const float* dout, float* din, int64_t outer, int n_out, int n_in, int64_t inner, float src_scale,
                                void* stream);

/* ---------------------------------------------------------------------------------------------
 * Inception-I3D spatial convolutions as implicit GEMM on the fp32 MFMA engine + TF-'same' max-pool (conv3d.hip).
 * NCDHW fp32.  geom (int32[16]) = {Cin, ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw}, p* = FRONT zero
 * pads of the dynamic 'same' padding (aj_i3d.py:68-90: front = pad // 2).  nn.Conv3d at aj_i3d.py:57-62,92.
 * ------------------------------------------------------------------------------------------- */
/* splitk > 1: the contraction (Cin*KV) is split over slabs in `workspace` (splitk*B*Cout*P floats) and reduced deterministically;
 * segx_conv3d_splitk returns the library's choice for the forward (wgrad = 0) or weight-gradient (wgrad = 1) GEMM */
int64_t segx_conv3d_splitk(int B, int Cout, const int* geom, int wgrad);
int segx_conv3d_fwd(const float* X, const float* W /* [Cout][Cin][KD][KH][KW] */, float* Y, int B, int Cout, const int* geom, int splitk,
                    float* workspace, void* stream);
/* Packed contraction order for Cin % 8 == 0: k runs (channel block of 8, tap, channel in block), so eight consecutive k share one tap and
 * the im2col loader decodes / masks once per eight gathers.  segx_conv3d_pack_weights writes Wp[o][c/8][t][c%8] from the layer's
 * W [Cout][Cin][KV]: mode 0 forward filters (o = Cout, c = Cin), mode 1 backward-data filters (o = Cin, c = Cout, transposed + flipped,
 * for segx_conv3d_fwd_packed(dY, Wp) with pads K-1-p); C = the contracted channel count. */
int segx_conv3d_pack_weights(const float* W, float* Wp, int O, int C, int KV, int mode, void* stream);
int segx_conv3d_fwd_packed(const float* X, const float* Wp, float* Y, int B, int Cout, const int* geom, int splitk, float* workspace,
                           void* stream);
/* Wt[ci][co][t] = W[co][ci][KV-1-t]: backward-data of a stride-1 conv = segx_conv3d_fwd(dY, Wt) with pads K-1-p */
int segx_conv3d_flip_weights(const float* W, float* Wt, int Cout, int Cin, int KV, void* stream);
/* per-sample weight gradients dWb[B][Cout][Cin*KV] (sum over B with segx_colsum); split-K over output positions:
 * workspace = splitk*B*Cout*Cin*KV floats when splitk > 1 */
int segx_conv3d_bwd_weight(const float* dY, const float* X, float* dWb, int B, int Cout, const int* geom, int splitk,
                           float* workspace, void* stream);
/* packed-row variant (Cin % 8 == 0): gradient rows ordered [Cout][Cin/8][KV][8] (one tap lookup per eight gathers in the loader);
 * segx_conv3d_unpack_wgrad(dWp, dW, Cout, Cin, KV) restores the layer's [Cout][Cin][KV] layout (after the sum over the batch) */
int segx_conv3d_bwd_weight_packed(const float* dY, const float* X, float* dWb, int B, int Cout, const int* geom, int splitk,
                                  float* workspace, void* stream);
int segx_conv3d_unpack_wgrad(const float* dWp, float* dW, int Cout, int Cin, int KV, void* stream);
/* Channel-slice forms of the packed convolutions: the two 3x3x3 convolutions of an Inception module (aj_i3d.py:112-141: b1b(b1a(x)), b2b(b2a(x)))
 * read their inputs as channel slices of ONE tensor produced by the fused b1a | b2a pointwise convolution + BatchNorm, and their backward-data
 * passes (segx_conv3d_fwd_packed_bs on dY with the transposed filters) write slices of one gradient tensor.  X / Y / dY point at the first channel of
 * the slice; *_bstride = distance between samples in floats (0 = dense); pointers 16-byte aligned */
int segx_conv3d_fwd_packed_bs(const float* X, const float* Wp, float* Y, int B, int Cout, const int* geom, int splitk, float* workspace,
                              int64_t x_bstride, int64_t y_bstride, void* stream);
int segx_conv3d_bwd_weight_packed_bs(const float* dY, const float* X, float* dWb, int B, int Cout, const int* geom, int splitk, float* workspace,
                                     int64_t dy_bstride, int64_t x_bstride, void* stream);
/* r06 -- the 3 x 3 x 3, stride-1, 'same' convolutions (Unit3D, aj_i3d.py:75-97; every spatial convolution of the Inception modules, :198-273) with an LDS-RESIDENT
 * INPUT HALO on the bf16x6 engine (conv3d_halo.hip): a workgroup owns 128 outputs (4 x 4 x 8 or 8 x 4 x 4) x 64 / 128 / 192 output channels, stages the halo of 8 input
 * channels once (split into the three bf16 planes once per voxel) and takes the 27 taps as shifted fragment reads; the filters arrive pre-split from
 * segx_conv3d_halo_pack.  Same products and fp32 accumulation as the im2col form of segx_conv3d_fwd_packed, other summation order.
 *   segx_conv3d_halo_ok      1 when the geometry (geom as above) is served: KD = KH = KW = 3, strides 1, pads 1, extents equal, Cin % 8 == 0, the process default
 *                            engine is bf16x6, knob 16 is on, the 4 x 4 x 8 / 8 x 4 x 4 tiling pads the extent by at most half and has >= knob 17 tiles over the batch;
 *   segx_conv3d_halo_wq_floats  size of the packed filter bank Wq in 4-byte units (the caller allocates a float tensor): ceil(C / 8) * 7 * 3 * O * 16;
 *   segx_conv3d_halo_pack    W [Cout][Cin][27] -> Wq; mode 0: forward (O = Cout rows, C = Cin contracted), mode 1: backward-data (O = Cin rows, C = Cout contracted,
 *                            taps flipped) -- the data gradient is segx_conv3d_halo_fwd on dY with that bank;
 *   segx_conv3d_halo_fwd     Y[b][Cout][D][H][W]; x_bstride / y_bstride: distance between samples in floats (0 = dense; channel slices of wider tensors as in
 *                            segx_conv3d_fwd_packed_bs); mtile: 0 = chosen by the library, 64 / 128 / 192 = output channels per workgroup (measurements). */
int segx_conv3d_halo_ok(int B, int Cout, const int* geom);
int64_t segx_conv3d_halo_wq_floats(int O, int C);
int segx_conv3d_halo_pack(const float* W, void* Wq, int O, int C, int mode, void* stream);
int segx_conv3d_halo_fwd(const float* X, const void* Wq, float* Y, int B, int Cout, const int* geom, int64_t x_bstride, int64_t y_bstride, int mtile, void* stream);
/* weight gradient of the same convolutions with the halo resident (three x-shifted windows per halo row, so that a lane's eight consecutive output positions are 16 aligned
 * bytes for every tap): dW [Cout][Cin][27], summed over the batch, from dY [B][Cout][D][H][W] and X [B][Cin][D][H][W].  A workgroup owns 128 / 192 output channels x
 * (8 input channels x 27 taps) and streams over its share of the 4 x 4 x 8 spatial blocks; the K-split slabs (ws: segx_conv3d_halo_wgrad_ws_floats floats) are reduced
 * in split order by a second launch (deterministic).  segx_conv3d_halo_wgrad_ok: the conditions of segx_conv3d_halo_ok for rows of eight outputs. */
int segx_conv3d_halo_wgrad_ok(int B, int Cout, const int* geom);
int64_t segx_conv3d_halo_wgrad_ws_floats(int B, int Cout, const int* geom);
int segx_conv3d_halo_wgrad(const float* dY, const float* X, float* dW, float* ws, int B, int Cout, const int* geom, int64_t dy_bstride, int64_t x_bstride, void* stream);
/* backward-data of a STRIDED convolution by direct gather (the stride-2 7x7x7 stem onto 3 channels); geom as above */
int segx_conv3d_bwd_data_direct(const float* dY, const float* W, float* dX, float* wt_ws /* Cout*Cin*KV floats of scratch */, int B, int Cout,
                                const int* geom, void* stream);
/* Input bridge composed into the I3D stem (segtran3d.py:420-423 feeding aj_i3d.py Conv3d_1a_7x7; exact re-association of two linear maps):
 * Wc [O][Cc][T] = stem filters Ws [O][C3][T] contracted with the bridge Wb [C3][Cb]; channel Cb of Wc carries the bridge bias bb [C3] (NULL:
 * none) for a constant-one input channel, channels above it are zero.  bwd: dWs, dWb, dbb from dWc by the chain rule. */
int segx_stem_compose_fwd(const float* Ws, const float* Wb, const float* bb, float* Wc, int O, int C3, int Cb, int Cc, int T, void* stream);
int segx_stem_compose_bwd(const float* dWc, const float* Ws, const float* Wb, const float* bb, float* dWs, float* dWb, float* dbb,
                          int O, int C3, int Cb, int Cc, int T, void* stream);
/* x [B][Cb][H][W][D] -> y [B][Cc][D][H][W]: depth moved in front (segtran3d.py:422), channel Cb = 1, channels above it = 0 */
int segx_bridge_input(const float* X, float* Y, int B, int Cb, int Cc, int H, int W, int D, void* stream);
/* X [B, Cb, H, W, D] -> Y [B, 2 Cb, D, H, U]: depth in front and space-to-depth along W, Y[b][2 c + j][d][h][u] = X[b][c][h][2 u + j - 2][d] (0 outside [0, W)) --
 * the input of the stride-(2, 2, 1) form of the I3D stem (segtran3d.py:420-423 + aj_i3d.py:75-97; U = W / 2 + 3) in one pass */
int segx_stem_s2d_input(const float* X, float* Y, int B, int Cb, int H, int W, int D, int U, void* stream);
/* The dense 3 x 3 stem of EfficientNet (efficientnet/model.py:128, 163: Conv2dStaticSamePadding(3, c0, 3, stride) on the up-sized image) as a direct convolution:
 * X [B, 3, H, W], W [Cout, 3, 3, 3], Y [B, Cout, OH, OW], zero padding pt rows on top / pl columns on the left (the rest of the window falls off the far edges).
 * _im2col: Xcol [B, rows, OH * OW] (rows >= 27; rows beyond 27 are zero) -- the window matrix the weight gradient contracts with dY as a batch-reduced skinny GEMM
 * (segx_gemm_f32 with batch_reduce; SEGX_TILE_SKINNY_NT).  Cin = 3, K = 3, stride 1 or 2 only. */
int segx_conv2d_stem_fwd(const float* X, const float* W, float* Y, int B, int Cin, int Cout, int H, int Wd, int OH, int OW, int K, int stride, int pt, int pl,
                         void* stream);
int segx_conv2d_stem_im2col(const float* X, float* Xcol, int B, int Cin, int H, int Wd, int OH, int OW, int K, int stride, int pt, int pl, int rows, void* stream);
/* foreground-token mask (get_mask, segtran2d.py:229-233 / segtran3d.py:266-270): out[b][cell] = (sum_c avgpool_{kd,kh,kw}(|x|) > 0) as 0/1 floats */
/* r05: get_mask(in_bridge_to3(batch)) of segtran3d.py:420-425 without materialising the bridged image: X = the raw batch [B][Cb][H][W][D], Wb [C3][Cb] / bb [C3]
 * (NULL: no bias) = the 1x1x1 bridge convolution; out [B][D/kd][H/kh][W/kw] (the permuted (D, H, W) order the network works in) */
int segx_bridge_mask(const float* X, const float* Wb, const float* bb, float* out, int B, int Cb, int C3, int H, int W, int D, int kd, int kh, int kw, void* stream);
int segx_nonzero_mask(const float* X, float* out, int B, int C, int D, int H, int W, int kd, int kh, int kw, void* stream);
/* in-step label -> n-hot maps (datasets2d.py:90-139,200-223; datasets3d.py:16-40): mode 0 fundus uint8 [B,Cin,S] -> [B,3,S];
 * 1 polyp uint8 -> [B,2,S]; 2 brats int32 [B,S] -> [B,4,S]; 3 fundus with exclusive=True (--exclusive: disc = ch0 without the cup) */
int segx_label_nhot(const void* labels, float* out, int B, int Cin, int64_t S, int mode, void* stream);
/* MaxPool3dSamePadding (aj_i3d.py:6-30): zero 'same' padding then max-pool.  geom (int32[15]) =
 * {ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw}; arg = arg-max index per output (-1 = a padded zero won) */
int segx_maxpool3d_fwd(const float* X, float* Y, int* arg, int64_t planes, const int* geom, void* stream);
/* addend (r05; may be NULL; strided pools only): a tensor of dX's shape added to the result in the same pass -- the gradient of the pooled tensor's other
 * consumers where it is also a feature-pyramid endpoint (segtran3d.py:436-441 feats[1..3]): autograd's accumulation kernel (read + read + write) need not run. */
int segx_maxpool3d_bwd(const float* dY, const int* arg, float* dX, int64_t planes, const int* geom, const float* addend, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sliding-window evaluation path (infer.hip)
 * ------------------------------------------------------------------------------------------- */
/* One window of test_single_batch (test_util2d.py:189-214) / test_single_case (test_util3d.py:151-166) after the network call:
 *   acc[b][c][window] += sigmoid(F.interpolate(scores[b][c], size=window, align_corners=False));  cnt[b][window] += 1
 * scores [B, C, d, h, w]; acc [B, C, CD, CH, CW]; cnt [B, CD, CH, CW];
 * geom (int32[12]) = {d, h, w, D, H, W (window), CD, CH, CW (canvas), oz, oy, ox (window origin)}; 2-D: d = D = CD = 1, oz = 0 */
int segx_window_accum(const float* scores, float* acc, float* cnt, int B, int C, const int* geom, void* stream);
/* soft = acc / cnt (cnt NULL: soft = acc), then
 *   mode 0: harden_segmap2d/3d (datasets2d.py:178-196, datasets3d.py:92-111): hard[c>=1] = soft[c] >= T, hard[0] = no other class on
 *   mode 1: make_brats_pred_consistent(is_conservative=False) (datasets3d.py:43-63) first, C == 4 (test_util3d.py:169-174)
 * acc/soft/hard [B, C, S] (hard as 0/1 floats), cnt [B, S]; soft may be NULL */
int segx_harden_segmap(const float* acc, const float* cnt, float* soft, float* hard, int B, int C, int64_t S, int mode, float T,
                       void* stream);
/* calc_dice sums (test_util2d.py:233-240) per plane: part [chunks][planes][3] = {sum pred*gt, sum pred^2, sum gt^2},
 * chunks*planes*3 = segx_dice_ws_floats(planes, S); reduce over chunks with segx_colsum */
int64_t segx_dice_ws_floats(int64_t planes, int64_t S);
int segx_dice_sums(const float* pred, const float* gt, float* part, int64_t planes, int64_t S, void* stream);

#ifdef __cplusplus
}
#endif
#endif
