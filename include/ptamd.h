/*
 * ptamd.h - C ABI of libptamd.so: the MI355X (gfx950) implementation of the
 * protein-transformer training hot path.
 *
 * The upstream reference has no FFI/plugin layer: its hot path sits behind plain
 * Python call signatures.  Each entry point below names the
 * reference function(s) whose arithmetic it replaces (paths relative to
 * /root/reference/protein_transformer/), and `protein_transformer_amd/` keeps
 * those Python signatures on top of this ABI via ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - row-major, innermost dimension contiguous, fp32 unless stated otherwise;
 *   - no allocation inside the library: the caller owns outputs and workspaces
 *     (query sizes with the *_workspace_bytes functions);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing
 *     synchronises;
 *   - return value: PTAMD_OK or a negative PTAMD_ERR_* code for host-detectable
 *     problems; device-detected anomalies are OR-ed into a caller-provided
 *     int32 status word (PTAMD_ST_* bits) and never trap.
 *
 * Shapes: B proteins, L padded residues, 12 angles / 14 atom slots per residue,
 * T = B*L tokens, D model width, F feed-forward width, H heads, dk = D/H.
 */
#ifndef PTAMD_H
#define PTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTAMD_OK 0
#define PTAMD_ERR_BAD_SHAPE (-1)  /* non-positive or unsupported size                      */
#define PTAMD_ERR_TOO_LONG (-2)   /* L exceeds what the kernel's LDS staging supports      */
#define PTAMD_ERR_WORKSPACE (-3)  /* workspace pointer NULL or too small                   */
#define PTAMD_ERR_HIP (-4)        /* a HIP runtime call failed (see ptamd_last_hip_error)  */
#define PTAMD_ERR_ALIGN (-5)      /* pointer / leading dimension not 16-byte aligned       */

/* device status bits (int32 word, OR-ed by kernels) */
#define PTAMD_ST_BAD_RESIDUE 1 /* id outside 0..19 before the padding (Sequence.py:50-51 KeyError)   */
#define PTAMD_ST_TOO_SHORT 2   /* fewer than 2 residues (StructureBuilder.py:58-59 StopIteration)    */
#define PTAMD_ST_BAD_THETA 4   /* bond angle outside [-pi, pi] (Structure.py:42 AssertionError)      */
#define PTAMD_ST_NONFINITE 8   /* NaN/inf loss (log.py:182-185 exits the process)                    */

#define PTAMD_PAD_ID 20
#define PTAMD_NUM_ANGLES 12
#define PTAMD_NUM_SLOTS 14

const char *ptamd_version(void);
const char *ptamd_last_hip_error(void);
/* number of side-chain atoms of residue type r (0..19); -1 if r is out of range */
int ptamd_sidechain_atoms(int residue);

/* ------------------------------------------------------------------ angles
 * inverse_trig_transform (losses.py:26-36): sincos[n,2] = (cos, sin) -> ang[n] = atan2(sin, cos). */
int ptamd_angles_fwd(const float *sincos, float *ang, int64_t n, void *stream);
/* d(ang)/d(cos, sin): dsincos[n,2] = dang[n] * (-sin, cos) / (cos^2 + sin^2)   */
int ptamd_angles_bwd(const float *sincos, const float *dang, float *dsincos, int64_t n, void *stream);

/* ------------------------------------------------------------------ NeRF
 * generate_coords / StructureBuilder.build / ResidueBuilder.build_bb,build_sc / nerf
 * (protein/Structure.py:12-65, protein/StructureBuilder.py:55-92,147-236).
 *   ang [B,L,12] radians; seq [B,L] int64 residue ids with trailing PTAMD_PAD_ID;
 *   crd [B,L*14,3] out (unused slots and padded residues = 0); status: int32[1], OR-ed. */
size_t ptamd_nerf_workspace_bytes(int B, int L);
/* nerf (protein/Structure.py:23-65), n independent placements: a, b, c, d [n,3]; l, theta, chi [n] */
int ptamd_nerf_place(const float *a, const float *b, const float *c, const float *l, const float *theta,
                     const float *chi, int64_t n, float *d, int32_t *status, void *stream);
/* pairwise_internal_dist (losses.py:233-253): x [n,dim] -> out [n,n] = sqrt(max(|x_i - x_j|^2, 1e-30)); the
 * training path never materialises this matrix (ptamd_drmsd_fwd_bwd), it exists for API parity */
int ptamd_pairwise_dist(const float *x, int n, int dim, float *out, void *stream);
int ptamd_nerf_fwd(const float *ang, const int64_t *seq, int B, int L, float *crd, int32_t *status,
                   void *stream);
/* adjoint of the build: dcrd [B,L*14,3] -> dang [B,L,12] (overwritten). Reproduces the
 * reference's graph: first residue's C is detached (StructureBuilder.py:185-187). */
int ptamd_nerf_bwd(const float *ang, const int64_t *seq, const float *crd, const float *dcrd, int B, int L,
                   float *dang, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ dRMSD
 * drmsd_work's loss part for a whole batch (losses.py:63-92,233-278; structure_utils.py:19-32).
 *   pred_crd, true_crd [B,L*14,3]; NaN in true_crd = atom absent.
 *   stats [B,8] out: {drmsd, drmsd/n, bb_drmsd, bb_drmsd/n_bb, n, n_bb, 0, 0}
 *   dcrd [B,L*14,3] out or NULL: d(drmsd/n)/d(pred_crd)  (the reference always
 *   differentiates the length-normalised loss, losses.py:80,91-92).
 * Workspace: 52 B per atom slot (compacted copies: predicted coordinates, (predicted, true) interleaved per axis, index) plus
 * the fixed-order partial sums of the upper-triangle sweep - one 1 KB column partial per (4-row-tile strip, 64-atom column
 * tile) and one 4 KB row partial per (strip, chunk of 8 column tiles): O(n^2 / 256) per protein (n = 14 L atom slots) -
 * CAPPED at 200 MB: beyond that the strips are swept in passes over groups of strips with the same buffers (+ 32 B per atom slot
 * of running sums), same bits as one launch.  B = 32: 159 MB at L = 512, 242 MB at L = 1500 (1.39 GB in round 4).  The size is
 * a function of (B, L) only - nothing is read from the process environment.  The *_budget forms take the cap on the partial
 * sums as an argument (bytes; 0 = the built-in 200 MB): for callers short of memory and for the test that forces passes on a
 * small batch; the same budget must be given to both calls. */
size_t ptamd_drmsd_workspace_bytes(int B, int L);
int ptamd_drmsd_fwd_bwd(const float *pred_crd, const float *true_crd, const int64_t *seq, int B, int L,
                        float *stats, float *dcrd, void *workspace, size_t workspace_bytes, void *stream);
size_t ptamd_drmsd_workspace_bytes_budget(int B, int L, size_t partial_budget_bytes);
int ptamd_drmsd_fwd_bwd_budget(const float *pred_crd, const float *true_crd, const int64_t *seq, int B, int L,
                               float *stats, float *dcrd, void *workspace, size_t workspace_bytes,
                               size_t partial_budget_bytes, void *stream);

/* rmsd (losses.py:281-286, ProDy calcTransformation + calcRMSD) for a whole batch: RMSD of the predicted atoms after optimal
 * rigid superposition (Kabsch) onto the true ones, over the atoms whose truth is present (NaN = absent), residues with
 * PTAMD_PAD_ID skipped.  rmsd [B] out (NaN for a protein without atoms).  One workgroup per protein, fp64 moments. */
int ptamd_kabsch_rmsd(const float *pred_crd, const float *true_crd, const int64_t *seq, int B, int L, float *rmsd,
                      void *stream);

/* mse_over_angles x3 (losses.py:175-214; train.py:64-66) in one pass.
 *   pred, truth [T,24]; out[6] = {sum_full, cnt_full, sum_bb, cnt_bb, sum_sc, cnt_sc} (fp32); the workspace holds
 *   the fp64 partial sums of the first pass */
size_t ptamd_mse_angles_workspace_bytes(void);
int ptamd_mse_angles_fwd(const float *pred, const float *truth, int64_t T, float *out, void *workspace,
                         size_t workspace_bytes, void *stream);
/* dpred[T,24] (+)= coef * 2*(pred-truth)/cnt_full on the selected elements; accumulate != 0 adds */
int ptamd_mse_angles_bwd(const float *pred, const float *truth, int64_t T, const float *sums, float coef,
                         int accumulate, float *dpred, void *stream);

/* ------------------------------------------------------------------ encoder building blocks
 * Row-major fp32 GEMM on the matrix cores (the `arith` field selects the arithmetic, see PTAMD_GEMM_* below):
 *     C[M,N] = epilogue( A (*) B )          with reduction length K
 *   a_kmajor = 0: A is [M,K] (K contiguous, lda);  1: A is stored [K,M] (M contiguous, lda)
 *   b_kmajor = 0: B is [N,K] (K contiguous, ldb) - the torch.nn.Linear weight layout;
 *              1: B is stored [K,N] (N contiguous, ldb)
 * epilogue, in order: + bias[N] (may be NULL); ReLU if (flags & PTAMD_EPI_RELU);
 *   dropout(p, seed, stream_id) if p > 0;  + residual[M,N] (ldr; may be NULL), or the ReLU/dropout gate of
 *   PTAMD_EPI_GATE;  tanh if (flags & PTAMD_EPI_TANH).
 * Replaces torch.nn.Linear (Attention.py:49,69; Sublayers.py:34; encoder_only.py:39)
 * and their autograd backward GEMMs. */
#define PTAMD_EPI_RELU 1
#define PTAMD_EPI_TANH 2
#define PTAMD_EPI_ACCUM 4 /* C += result (used for split reductions) */
#define PTAMD_EPI_SLABS 16 /* split_k > 1 only, plain epilogue only (no bias / residual / activation / dropout / colsum): the K
                            * slices' partial products stay in `workspace` as [splits][M][N] fp32 and NO reduction is launched -
                            * the caller sums them in slab order in the kernel that reads the product next
                            * (ptamd_layernorm_bwd_dropout: dy_slabs; ptamd_layernorm_fwd_sum).  C is not written; with an
                            * effective split of 1 (K of one or two 32-blocks) the flag is ignored and C is. */
#define PTAMD_EPI_GATE 8  /* result = residual[m,n] > 0 ? result * gate_scale : 0 - the backward of ReLU + dropout through
                            the saved activation (Sublayers.py:34), instead of adding `residual` */
typedef struct {
  int M, N, K;
  const float *A; int lda; int a_kmajor;
  const float *B; int ldb; int b_kmajor;
  float *C; int ldc;
  const float *bias;
  const float *residual; int ldr;
  int flags;
  float dropout_p; uint64_t seed; uint32_t stream_id;
  int split_k;            /* >1: partials go to workspace and are reduced deterministically */
  void *workspace; size_t workspace_bytes;
  float *colsum;          /* optional, a_kmajor only: colsum[m] += sum_k A[k][m] (the bias gradient of a dW product) */
  float gate_scale;       /* PTAMD_EPI_GATE: 1 / (1 - p) of the dropout that followed the ReLU */
  int arith;              /* PTAMD_GEMM_* below: the arithmetic of THIS call (the library keeps no mode of its own) */
  int reserved_cus;       /* the persistent kernels leave this many CUs free, e.g. for an RCCL all-reduce of the layer above
                             that runs beside the backward GEMMs under data parallelism; 0 = take every CU */
  /* F16X2 arithmetic only, optional: the row scales of the operands (uint32 bits of a power of two s with
   * max |x_row| * s < 2^15, see PTAMD_GEMM_F16X2) when the caller already has them - written by the kernel that
   * produced the operand (ptamd_layernorm_fwd, ptamd_layernorm_bwd_dropout) or derived from a bound of the row maxima
   * (ptamd_weight_scales, ptamd_bound_scales).  NULL: ptamd_gemm finds them with a pass over the operand (needs the workspace).
   * a_scale_stride / b_scale_stride: 1 = one scale per operand row (A: per m, B: per n); 0 = ONE scale for every row of
   * the operand - the array then holds FOUR copies of it (row-contiguous operands load the scales of four rows at once).
   * With uniform scales on both sides the weight-gradient products (k-major A and B) can run in F16X2 without any pass:
   * the error is then relative to the largest row of the operand (norm-wise over the whole product). */
  const uint32_t *a_scale; int a_scale_stride;
  const uint32_t *b_scale; int b_scale_stride;
  /* PTAMD_EPI_GATE from ONE BIT per element instead of the [M, N] fp32 activation (exactly one of `residual` and
   * `gate_mask` with that flag): what ptamd_gemm_hp wrote through `gate_mask_out` when it produced the activation
   * (ptamd_gate_mask_bytes(M, N) bytes).  F16X2 arithmetic (AUTO resolves to it for K-contiguous A), float4 epilogue (N, ldc % 4 == 0, C 16-byte
   * aligned), split_k <= 1 and dropout_p == 0 only - PTAMD_ERR_BAD_SHAPE otherwise.  Same result, bit for bit, as gating by the activation. */
  const uint64_t *gate_mask;
} ptamd_gemm_args;
/* Layout of the 1-bit gate of an [M, N] activation: entry ((cb * ceil(M / 32) + rb) * 16 + r), one uint64 = the 64 lane
 * decisions of accumulator register r of the 32 x 32 block (rb, cb) of v_mfma_f32_32x32x*: bit l set <=> element
 * (row 32 rb + (r & 3) + 8 (r >> 2) + 4 (l >> 5), column 32 cb + (l & 31)) is > 0.  The producing epilogue stores the
 * ballots of its compares, the gated epilogue reads them as scalar loads and selects with them: 4 MB instead of 134 MB per
 * layer for the hidden layer of the feed-forward block at 32 x 512 tokens (Sublayers.py:28-34, ReLU + dropout). */
size_t ptamd_gate_mask_bytes(int M, int N);
size_t ptamd_gemm_workspace_bytes(int M, int N, int split_k);
int ptamd_gemm(const ptamd_gemm_args *args, void *stream);

/* Arithmetic of a ptamd_gemm call (`arith` field of its arguments; the library holds no process-wide mode - two
 * models, or two streams, may use different arithmetics side by side).  The reference computes its Linear layers in fp32 (torch.nn.Linear on fp32 tensors); all modes take and
 * return fp32 and accumulate in fp32:
 *   PTAMD_GEMM_F32          v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain over k (157 TF/s peak).
 *   PTAMD_GEMM_BF16X3       every f32 operand is split EXACTLY into three bf16 terms x = x1 + x2 + x3 (round to nearest
 *                           at each level: 3 x 8 significand bits = the 24 of an f32) while it is staged into LDS, and
 *                           x*y is evaluated on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, f32 accumulate) as the
 *                           six products x1y1 + x1y2 + x2y1 + x1y3 + x2y2 + x3y1.  The three dropped terms are below
 *                           2^-23 |x y| and unbiased, so the result is at least as close to the exact dot product as the
 *                           fp32 fma chain is (asserted against fp64 in tests/test_gpu_kernels.py), at 6/16 of the
 *                           matrix-pipe cost.
 *   PTAMD_GEMM_BF16X3_FULL  all nine products.
 *   PTAMD_GEMM_F16X2        every operand row (A: per m, B: per n) is scaled by the power of two that takes its largest
 *                           |x| into [2^14, 2^15) - found by a pass over the operands in front of the product, kept in
 *                           the workspace - and split into TWO f16 terms (2 x 11 significand bits); x*y is the three
 *                           products x1y1 + x1y2 + x2y1 on v_mfma_f32_32x32x16_f16 and the scales are taken out of the
 *                           f32 accumulators before the epilogue.  Half the matrix-pipe work of BF16X3.  The error is
 *                           fp32-grade in the NORM-WISE sense: an element keeps 22 bits down to 2^-18 of its row's
 *                           maximum and an absolute 2^-39 of that maximum below, so
 *                               |error| <= 2^-20 sum|x||y| + 2^-36 K max|x_row| max|y_col|     (worst case)
 *                           - on the tensors of the training step (rows spanning a few binades) it measures slightly
 *                           BELOW the fp32 fma chain (4e-7 vs 6e-7 max, 4e-8 vs 7e-8 rms in units of sum|x||y|), for
 *                           rows spanning 40 binades it does not (tests/test_gpu_kernels.py shows both).  A row of
 *                           subnormals only is flushed to zero.  Needs the workspace also when split_k <= 1.
 *   PTAMD_GEMM_AUTO         per call: F16X2 when A is K-contiguous (activations / per-token gradients x weights: the
 *                           pass over the operands is cheap next to the product), BF16X3 when A is k-major (the
 *                           weight-gradient reductions over all tokens, where that pass would read both big operands
 *                           once more).  Measured on the benchmark step: 15.0 ms against 15.8 (all BF16X3) and 15.25
 *                           (all F16X2); gradients of a whole step against fp64: same level in every mode
 *                           (tests/test_gpu_model.py).
 * Whatever the mode, products with K < 16 or an operand of 4 GiB or more run in PTAMD_GEMM_F32. */
#define PTAMD_GEMM_F32 0
#define PTAMD_GEMM_BF16X3 1
#define PTAMD_GEMM_BF16X3_FULL 2
#define PTAMD_GEMM_F16X2 3
#define PTAMD_GEMM_AUTO 4
/* matrix-pipe products per fp32 product that ptamd_gemm would use for these arguments (shapes, layouts, `arith`):
 * 1 (F32), 3 (F16X2), 6 (BF16X3), 9 (BF16X3_FULL); host only, nothing is launched */
int ptamd_gemm_products(const ptamd_gemm_args *args);
/* Up to four INDEPENDENT weight-gradient products in ONE launch (+ one launch for all their split-K reductions): the four
 * dW = dy^T x of an encoder layer (autograd of the nn.Linear layers of Attention.py:49,69 and Sublayers.py:28-34), whose
 * output matrices are too few tiles to fill the chip one by one.  The work items (256 x 128 tile, K split) of the members
 * are concatenated and dealt out in contiguous ranges, so the launch is balanced when the members' K per split agree (four
 * members with 96 tiles and split_k = 8: three items per CU).  Same results, bit for bit, as ptamd_gemm called on each member
 * with the same split_k.  Every member must be: a_kmajor and b_kmajor, PTAMD_GEMM_F16X2 with a_scale and b_scale given,
 * split_k >= 2 with its own workspace (ptamd_gemm_workspace_bytes), flags == PTAMD_EPI_ACCUM and no other epilogue, N and ldc
 * multiples of 4, C 16-byte aligned, colsum in all members or in none - PTAMD_ERR_BAD_SHAPE otherwise (nothing is launched). */
int ptamd_gemm_group(const ptamd_gemm_args *args, int n, void *stream);

/* ------------------------------------------------------------------ pre-split ("half-pair", hp) operands
 * An fp32 matrix [rows, K] stored as TWO f16 planes + one power-of-two scale per row, x * scale = hi + lo to 22 bits (the
 * PTAMD_GEMM_F16X2 arithmetic with the splitting done once by the WRITER of the operand instead of by every GEMM that
 * reads it).  4 bytes per element; 32 x 16 blocks that are byte-for-byte the LDS image of an MFMA operand, so that the
 * GEMM fills its stages by LDS-DMA without touching the vector ALU (layout: csrc/hp_format.h).  rows / K are zero padded
 * to multiples of 32.
 *   ptamd_hp_bytes        size of the planes buffer;  ptamd_hp_padded_rows  length of the scale array
 *   ptamd_hp_split        x [rows, K] (ld; transposed != 0: x is stored [K, rows] and the operand is its transpose)
 *                         -> planes, scale.  Row maximum pass + write pass.
 *   ptamd_gemm_hp         C[M,N] = epilogue(A B^T), A = hp [M,K], B = hp [N,K]; same epilogue, flags, dropout masks and
 *                         split-K convention as ptamd_gemm.  Replaces the same torch.nn.Linear forward / dX products
 *                         (Attention.py:49,69; Sublayers.py:34; encoder_only.py:39). */
size_t ptamd_hp_bytes(int rows, int K);
int ptamd_hp_padded_rows(int rows);
int ptamd_hp_split(const float *x, int ld, int rows, int K, int transposed, void *planes, float *scale, void *stream);
/* ptamd_hp_split_rows: up to 16 K-contiguous matrices [rows, K] (K % 4 == 0, K <= 2048) -> planes + scales in ONE launch
 * (one wavefront per row: maximum, scale, split): the weight matrices of a model once per step. */
typedef struct {
  const float *x; int ld, rows, K;
  void *planes; float *scale;
} ptamd_hp_split_job;
int ptamd_hp_split_rows(const ptamd_hp_split_job *jobs_host, int njobs, void *stream);
/* ptamd_hp_split_cols: up to 16 ROW-contiguous matrices x [K, rows] (row stride ld) -> the planes of their TRANSPOSES (operand
 * rows = the columns of x: the weights as B operands of the dX products dy W, Sublayers.py:28-34 backward) in ONE launch,
 * with the operand rows' scales GIVEN in `scale` (an INPUT here: floats that are powers of two, e.g. the column scales
 * ptamd_weight_scales wrote, reinterpreted). */
int ptamd_hp_split_cols(const ptamd_hp_split_job *jobs_host, int njobs, void *stream);
typedef struct {
  int M, N, K;
  const void *A; const float *A_scale;
  const void *B; const float *B_scale;
  float *C; int ldc;
  const float *bias;
  const float *residual; int ldr;
  int flags;
  float dropout_p; uint64_t seed; uint32_t stream_id;
  int split_k;
  void *workspace; size_t workspace_bytes;
  float gate_scale;
  int reserved_cus;
  const uint64_t *gate_mask;   /* PTAMD_EPI_GATE from the 1-bit gate (see ptamd_gemm_args.gate_mask) */
  uint64_t *gate_mask_out;     /* optional: the 1-bit gate "result > 0" of THIS product's output (after bias / ReLU /
                                  dropout), ptamd_gate_mask_bytes(M, N) bytes; float4 epilogue and split_k <= 1 only */
  /* optional, the QKV product (Attention.py:49): columns kv_col0 .. N - 1 = K then V, kv_heads heads of 64 columns each
   * (N == kv_col0 + 128 kv_heads), leave the epilogue as the pre-split planes the f16x2 attention kernels read
   * (ptamd_attention_kv_bytes / _kv_inv_floats; csrc/kv_format.h) INSTEAD of fp32 - C[:, kv_col0:] is not written; columns in
   * front (Q) are stored as usual.  M a multiple of 32, bias only (flags 0, no dropout / residual / gate), split_k <= 1. */
  void *kv_planes; float *kv_inv; int kv_col0, kv_heads;
} ptamd_gemm_hp_args;
size_t ptamd_gemm_hp_workspace_bytes(int M, int N, int split_k);
int ptamd_gemm_hp(const ptamd_gemm_hp_args *args, void *stream);

/* ------------------------------------------------------------------ f16x2 bookkeeping without passes over activations
 * (csrc/scales.hip).  Scales are uint32 bit patterns of powers of two, as ptamd_gemm_args.a_scale / b_scale take them.
 * ptamd_weight_scales: for each job (a weight matrix or a sub-matrix / vector of the flat parameter buffer,
 *   w [rows, cols], row stride ld): row_scale[rows] (B operand of x W^T), col_scale[cols] (B operand of dy W),
 *   stats[4] = {largest row L2 norm, largest column L2 norm, largest |w|, 0}; any output may be NULL.  Two launches for
 *   the whole list (at most 40 jobs per call).
 * ptamd_bound_scales: out = ((max|gamma| sqrt_d + |beta|_2) if a LayerNorm feeds the product else 1) * w_stats[w_stat_index]
 *   [+ max|bias|], times post_scale - an upper bound of |x W^T + b|_inf for every row x = LN(.) (Cauchy-Schwarz), or of
 *   |dy W|_inf / |dy|_2 when no LayerNorm is given; out_scale[0..3] receive the f16x2 scale of a row with that maximum (four
 *   copies, to be used with a_scale_stride / b_scale_stride = 0), out_value the bound itself (the `bound_factor` of
 *   ptamd_layernorm_bwd_dropout).
 *   ln_*_stats / w_stats / bias_stats point at stats[4] records written by ptamd_weight_scales earlier on the stream. */
typedef struct {
  const float *w; int rows, cols, ld;
  uint32_t *row_scale; uint32_t *col_scale; float *stats;
  int rows_only;          /* != 0: no column pass (stats[1] stays 0): for a big activation whose row scales / largest |x| are wanted */
} ptamd_wscale_job;
int ptamd_weight_scales(const ptamd_wscale_job *jobs_host, int njobs, void *stream);
typedef struct {
  const float *ln_gamma_stats; const float *ln_beta_stats;
  const float *w_stats; int w_stat_index;
  const float *bias_stats;
  float sqrt_d, post_scale;
  uint32_t *out_scale; float *out_value;
} ptamd_bound_job;
int ptamd_bound_scales(const ptamd_bound_job *jobs_host, int njobs, void *stream);
/* Presets of up to 8 small device buffers in ONE launch (dst[0..n) = value): the atomicMin targets of a backward pass (the
 * uniform scales of dy2 / dz1 / dyo / dqkv and the row scales of dqkv start at 0x7F000000, the largest finite power of two). */
typedef struct { uint32_t *dst; int64_t n; uint32_t value; } ptamd_fill_job;
int ptamd_fill_u32(const ptamd_fill_job *jobs_host, int njobs, void *stream);

/* ------------------------------------------------------------------ the weights of a step in ONE pass (csrc/wprep.hip)
 * ptamd_weights_prep: what ptamd_weight_scales + ptamd_bound_scales + ptamd_hp_split_rows + ptamd_hp_split_cols compute for
 *   the weight matrices of a model (row / column f16x2 scales, statistics, weight-derived bounds, W as hp planes, W^T as hp
 *   planes) in two launches and one read of the weights; same bits.
 * ptamd_sgd_step_prep / ptamd_adam_step_prep: ptamd_sgd_step / ptamd_adam_step (clip + optimizer.step(), train.py:41-46,
 *   371-381) that leave all of that behind for the NEXT forward pass, in the pass that writes the weights anyway.
 * The description of the model is a PLAN whose tables live in DEVICE memory (built once by the caller):
 *   segs        the listed matrices (or vectors, rows = 1) of the flat parameter buffer, K-contiguous, row stride = cols,
 *               cols % 4 == 0 and cols <= 512 (wider ones as column panels, see rowmax_index); pairwise disjoint;
 *   blocks_a    [nblocks_a][2] int32: (segment, block of 32 rows inside it) for the first nblocks_a_matrices entries, then
 *               (-(range + 1), block of ptamd_wprep_plain_floats_per_block() floats inside plain range `range`): the plain
 *               ranges [nplain][2] int64 (first element, elements) cover everything that is not a segment - segments + plain
 *               ranges partition [0, numel);
 *   blocks_b    [nblocks_b][4] int32: (0, segment, block of 2048 columns, 0) column scales; (1, segment, block of 256 16-byte
 *               chunks of its col_planes, 0); (2, bounds group, 0, 0); (3, panelled matrix, block of 256 rows, 0) row scales;
 *   bounds / groups [ngroups][4] int32 (first bound job, jobs, first entry of colnorm_segs, entries): a group = an encoder layer;
 *               a bound job is ptamd_bound_job with indices instead of pointers (statistics record, entry of `scales`, entry of
 *               `values`; -1 = none);
 *   scales      uint32 row / column / bound scales; values: floats (bound values); colmax [2][ncolmax], stats [2][nstats][4]:
 *               zeroed ONCE by the caller, `parity` (0 / 1) must alternate from call to call on a plan; colsq: column
 *               sum-of-squares partials in fp64, [(rows + 31) / 32][cols] doubles per segment that asks for them (16-byte
 *               aligned; colsq_index counts doubles). */
typedef struct {
  int64_t offset;              /* first element in the flat parameter buffer */
  int32_t rows, cols;
  int32_t stats_row0;          /* rows >= stats_row0 enter the statistics (a sub-matrix: W_v inside W_qkv) */
  int32_t stats_index;         /* record of `stats` {largest row L2 norm, largest column L2 norm, largest |w|, -}, or -1 */
  int32_t row_scale_index;     /* first entry of `scales` for the rows' scales, or -1 */
  int32_t col_scale_index;     /* ... for the columns' scales, or -1 (then no column work at all) */
  int32_t colmax_index;        /* first entry of this matrix's columns in a copy of `colmax` */
  int32_t colsq_index;         /* first double of its partials in `colsq` (the column NORM is wanted: statistics [1]), or -1 */
  uint64_t row_planes;         /* device pointer: W as hp planes (ptamd_hp_bytes(rows, cols)) split with the row scales; rows and
                                  cols multiples of 32; 0 = none */
  uint64_t col_planes;         /* device pointer: W^T as hp planes (ptamd_hp_bytes(cols, rows)) split with the column scales; 0 = none */
  int32_t ld;                  /* row stride in floats (= cols for a whole matrix) */
  int32_t rowmax_index;        /* >= 0: this segment is a COLUMN PANEL (<= 512 columns) of a wider matrix - kernel A keeps whole rows
                                  in registers, 512 columns at most: its row maxima go by atomicMax to colmax[rowmax_index + r] (the
                                  pool of maxima is shared), the row scales come from a block (3, full matrix, .) of kernel B, where
                                  the full matrix is a segment of its own that blocks_a does not list (row_scale_index, rowmax_index,
                                  rows; col_planes with colmax_index = its first panel's); a panel has stats_row0 = -1: its
                                  statistics are the maximum only.  -1: not a panel */
} ptamd_wprep_seg;
typedef struct {
  int32_t ln_gamma_stats, ln_beta_stats, w_stats, w_stat_index, bias_stats;   /* w_stats = -1: factor 1 (the bound of the input row) */
  float sqrt_d, post_scale;
  int32_t out_scale, out_value;
} ptamd_wprep_bound;
typedef struct {
  const ptamd_wprep_seg *segs; int32_t nsegs;
  const int32_t *blocks_a; int32_t nblocks_a, nblocks_a_matrices;
  const int64_t *plain; int32_t nplain;
  const int32_t *blocks_b; int32_t nblocks_b;
  const ptamd_wprep_bound *bounds; int32_t nbounds;
  const int32_t *groups; int32_t ngroups;
  const int32_t *colnorm_segs;
  uint32_t *scales; float *values;
  uint32_t *colmax; int32_t ncolmax;
  void *colsq;
  float *stats; int32_t nstats;
  int64_t numel;               /* floats in the flat parameter buffer */
  int32_t with_planes;         /* this call writes the row_planes / col_planes of the segments (0: scales and bounds only -
                                  what a pass needs whose products do not run on ptamd_gemm_hp) */
} ptamd_wprep_plan;
int ptamd_wprep_rows_per_block(void);
int ptamd_wprep_plain_floats_per_block(void);
int ptamd_weights_prep(const ptamd_wprep_plan *plan_host, const float *w, int parity, void *stream);
/* zero_grad != 0 (here and in ptamd_sgd_step / ptamd_adam_step): g is zeroed behind its last read - the `optimizer.zero_grad()`
 * of the NEXT step (train.py:37) in the pass that streams g anyway instead of a 76 MB fill of its own. */
int ptamd_sgd_step_prep(const ptamd_wprep_plan *plan_host, int parity, float *w, float *g, int64_t n, const float *sqnorm,
                        float max_norm, float lr, float weight_decay, int zero_grad, void *stream);
int ptamd_adam_step_prep(const ptamd_wprep_plan *plan_host, int parity, float *w, float *g, float *m, float *v, int64_t n,
                         const float *sqnorm, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                         int step, int zero_grad, void *stream);

/* torch.nn.LayerNorm(D, eps=1e-5) (Sublayers.py:13,17): y = (x-mean)*rstd*gamma+beta; saves mean,rstd [T];
 * row_scale [T] (may be NULL): the f16x2 scale of every row of y, for the GEMM that reads y as its A operand;
 * planes (may be NULL; needs row_scale and D % 32 == 0): y once more in the pre-split hp format (ptamd_hp_bytes(T, D)
 * bytes), scaled by row_scale - the A operand of ptamd_gemm_hp for the product behind the LayerNorm (QKV, FFN layer 1) */
int ptamd_layernorm_fwd(const float *x, const float *gamma, const float *beta, int64_t T, int D, float *y,
                        float *mean, float *rstd, uint32_t *row_scale, void *planes, void *stream);
/* Round 6: the same LayerNorm behind a split product whose K slices were left unreduced (PTAMD_EPI_SLABS; FFN layer 2 at few
 * tokens): the rows are first made here - x_out = residual + dropout(sum of the n_slabs (2 ... 4) [T, D] slabs, slab_stride
 * floats apart, in slab order + bias) with the decisions of the product's own epilogue (dropout_p, seed, stream_id as
 * ptamd_gemm_args; Sublayers.py:16-17) - and normalised from registers: the bits of the reduction launch + ptamd_layernorm_fwd,
 * one launch fewer.  residual / x_out [T, D] dense, D <= 512. */
int ptamd_layernorm_fwd_sum(const float *slabs, int n_slabs, int64_t slab_stride, const float *bias, const float *residual,
                            float dropout_p, uint64_t seed, uint32_t stream_id, float *x_out, const float *gamma,
                            const float *beta, int64_t T, int D, float *y, float *mean, float *rstd, uint32_t *row_scale,
                            void *planes, void *stream);
/* dx [T,D] = LN'(dy) + dres (dres: gradient of the residual branch, may be NULL; dx may alias dres);
 * dgamma, dbeta [D] accumulated (+=) through fixed-order partials in workspace */
size_t ptamd_layernorm_bwd_workspace_bytes(int D);
/* dgamma == dbeta == NULL in the two calls below: the kernel leaves its fixed-order partial sums in `workspace` and the caller
 * finishes up to 16 such sites with ONE launch of ptamd_layernorm_bwd_reduce (dgamma / dbeta += their column sums) - e.g.
 * both LayerNorms of an encoder layer, or all of a backward pass; every pending site needs a workspace of its own. */
typedef struct { const void *partials; int D; float *dgamma; float *dbeta; } ptamd_ln_reduce_job;
int ptamd_layernorm_bwd_reduce(const ptamd_ln_reduce_job *jobs_host, int njobs, void *stream);
int ptamd_layernorm_bwd(const float *dy, const float *x, const float *gamma, const float *mean, const float *rstd,
                        const float *dres, int64_t T, int D, float *dx, float *dgamma, float *dbeta, void *workspace,
                        size_t workspace_bytes, void *stream);

/* ptamd_layernorm_bwd fused with the dropout backward that follows it in the encoder (SublayerConnection: x + drop(f(LN(x))),
 * Sublayers.py:17): dx as above, and `dropped` [T,D] = dx * mask / (1 - p) with the mask the GEMM epilogue drew for
 * (seed, stream_id) - what ptamd_dropout_bwd would make of dx (dropout_p = 0: dropped is not written, it equals dx).
 * Also, for the GEMMs that read `dropped` as their A operand: row_scale [T] = its f16x2 row scales, and bound_scale [T] =
 * the scale of a row bounded by |dropped[t]|_2 * *bound_factor (device scalar, e.g. from ptamd_bound_scales): the scale of
 * row t of dropped W.  row_scale_min[0..3] / bound_scale_min[0..3] (may be NULL): the smallest of those scales over all
 * rows, i.e. the scale of the largest row - atomicMin into four copies that the caller preset to 0x7F000000; the uniform
 * scale of `dropped` / `dropped W` as an operand of a weight-gradient product.  dropped_planes (may be NULL; needs row_scale,
 * D % 32 == 0): `dropped` a second time in the pre-split hp format (ptamd_hp_bytes(T, D) bytes) with the scales of
 * row_scale - the A operand of ptamd_gemm_hp for the dX product behind it.  Any of the outputs may be NULL.  D <= 1024.
 * dy_slabs (1 ... 4; round 6): dy is the SUM of that many [T, D] slabs, dy_slab_stride floats apart - the K slices of a split
 * product left unreduced (PTAMD_EPI_SLABS) - added in slab order while the rows are read: the bits of the reduction launch it
 * replaces, at a fraction of that launch's cost for the few tokens at which products are split (D <= 512 for dy_slabs > 1). */
int ptamd_layernorm_bwd_dropout(const float *dy, const float *x, const float *gamma, const float *mean, const float *rstd,
                                const float *dres, int64_t T, int D, float dropout_p, uint64_t seed, uint32_t stream_id,
                                float *dx, float *dropped, uint32_t *row_scale, const float *bound_factor,
                                uint32_t *bound_scale, uint32_t *row_scale_min, uint32_t *bound_scale_min,
                                void *dropped_planes, float *dgamma,
                                float *dbeta, int dy_slabs, int64_t dy_slab_stride, void *workspace, size_t workspace_bytes,
                                void *stream);

/* Embeddings * sqrt(D) and the doubled positional add of Encoder.py:30 + Sublayers.py:59-62,72:
 *   x0 = emb[seq]*sqrt(D); out = drop2(x0 + drop1(x0 + pe[pos]))      (eval: 2*x0 + pe) */
int ptamd_embed_fwd(const int64_t *seq, const float *emb, const float *pe, int B, int L, int D, float dropout_p,
                    uint64_t seed, float *out, void *stream);
/* demb [22,D] += scatter of dout with the same dropout masks (fixed-order partial tables in workspace) */
size_t ptamd_embed_bwd_workspace_bytes(int D);
int ptamd_embed_bwd(const int64_t *seq, const float *dout, int B, int L, int D, float dropout_p, uint64_t seed,
                    float *demb, void *workspace, size_t workspace_bytes, void *stream);

/* Fused masked multi-head attention (Attention.py:14-22,55-69), scores never materialised.  `arith` (head sizes 64
 * and 32; other head sizes always run the exact-f32 kernels):
 *   PTAMD_GEMM_F32                 the exact-f32 MFMA kernels;
 *   PTAMD_GEMM_BF16X3 / _FULL      operands split exactly into three bf16 terms, six products on the bf16 matrix pipe;
 *   PTAMD_GEMM_F16X2 / _AUTO       operands scaled by powers of two and split into two f16 terms, three products on the
 *                                  f16 matrix pipe (the f16x2 error model of ptamd_gemm: relative to the largest element of
 *                                  a scaling group - one Q / dO / K / V row, or four staged rows - instead of to every
 *                                  element); the scales are found inside the kernels, nothing is added to the interface.
 *   qkv [T,3D]: Q | K | V column blocks, head h at columns h*dk..; key-padding mask from seq != 20;
 *   softmax(QK^T/sqrt(dk)) with dropout p on the probabilities; out [T,D] heads merged.
 *   lse [B,H,L] saves log-sum-exp per query row for the backward. dk must be 32 or 64.
 *   ptamd_attention_bwd, row_scale [T] / row_scale_min [4] (optional, both or only the first; f16x2 arithmetic with
 *   dk 32 / 64 only, PTAMD_ERR_BAD_SHAPE otherwise): the f16x2 row scales of dqkv (ptamd_gemm a_scale) and the smallest
 *   of them (4 copies: a uniform scale, stride 0), accumulated with atomicMin - preset both to 0x7F000000.
 *   keep_bits (optional, f16x2 arithmetic with dk 32 / 64 only, PTAMD_ERR_BAD_SHAPE otherwise; ptamd_attention_keep_bits_bytes
 *   bytes): the dropout decisions of the probabilities, written by ptamd_attention_fwd and handed to ptamd_attention_bwd by
 *   the caller (the library keeps nothing between calls), so that the backward kernel reads one word per key and 32
 *   queries instead of drawing the counter hash again: word [(b, h)][q / 32][key] (keys padded to a multiple of 32),
 *   bit q % 32 = 1 where query q keeps key.  The decisions are the generator's (csrc/attn_dropout.h) either way: a
 *   backward call without them, or in another arithmetic, regenerates exactly the same mask.
 *   kv_planes / kv_inv (optional, both or none; f16x2 arithmetic and shapes for which ptamd_attention_reads_kv_planes says 1,
 *   PTAMD_ERR_BAD_SHAPE otherwise): K and V PRE-SPLIT, as the epilogue of the QKV product wrote them (ptamd_gemm_hp_args.kv_planes;
 *   ptamd_attention_kv_bytes(T, H) bytes and ptamd_attention_kv_inv_floats(T, H) floats, T = B L; layout: csrc/kv_format.h).
 *   The K | V columns of `qkv` are then NOT read (they need not have been written); Q is.  The forward result is bit for bit
 *   that of the call without planes on the fp32 K / V the planes were made from; the backward one differs by rounding (a key
 *   row is scaled with its group of four there instead of on its own). */
int ptamd_attention_fwd(const float *qkv, const int64_t *seq, int B, int L, int H, int dk, float dropout_p,
                        uint64_t seed, uint32_t stream_id, int arith, float *out, float *lse, uint32_t *keep_bits,
                        const void *kv_planes, const float *kv_inv, void *stream);
int ptamd_attention_bwd(const float *qkv, const int64_t *seq, const float *out, const float *dout, const float *lse,
                        int B, int L, int H, int dk, float dropout_p, uint64_t seed, uint32_t stream_id, int arith,
                        float *dqkv, uint32_t *row_scale, uint32_t *row_scale_min, const uint32_t *keep_bits,
                        const void *kv_planes, const float *kv_inv, void *workspace, size_t workspace_bytes, void *stream);
size_t ptamd_attention_kv_bytes(int T, int H);
size_t ptamd_attention_kv_inv_floats(int T, int H);
/* 1 when ptamd_attention_fwd / _bwd of this shape and arithmetic read pre-split K / V (head size 64, L a multiple of 32, a batch
 * large enough for the 256-query forward kernel and the one-sweep backward kernel) */
int ptamd_attention_reads_kv_planes(int B, int L, int H, int dk, int arith);
/* Workspace of ptamd_attention_bwd: delta [B, H, L] and, behind it, the slabs of the SPLIT one-sweep backward kernel where this
 * shape takes it (round 6; head size 64 and at most half as many (protein, head) pairs as CUs - the per-GPU share of a strongly
 * scaled batch): the sweep runs as one workgroup per (pair, 256-key block) and, while that leaves CUs without one, per range of
 * query tiles as well; a key block's contribution to dQ goes to slab [key block][B L][D], a query range's dK | dV to slab
 * [range][B L][2 D], one more launch sums them in a fixed order (dQ: the bits of the unsplit sweep) and finishes the row
 * scales.  32 x 512 x 8 heads: 0.5 MB (delta alone); 16 / 8 / 4 proteins: + 33.5 / 50.3 / 41.9 MB.
 * The size is a function of (B, L, H, dk), the CU count of the current device and - for head size 64 - of the measurement knob
 * PTAMD_ATTN_FUSED in the process environment (read at every call: 0 = never a one-sweep kernel, 1 = the unsplit one, 2 = the split
 * one, unset = by the number of pairs): the size query and the call must see the same value; a call that finds its workspace too
 * small for the kernel it would take returns PTAMD_ERR_WORKSPACE and launches nothing. */
size_t ptamd_attention_workspace_bytes(int B, int L, int H, int dk);
size_t ptamd_attention_keep_bits_bytes(int B, int L, int H);
/* 1 when ptamd_attention_bwd of this shape and arithmetic would read keep_bits (f16x2 arithmetic, dk 32 / 64: the one-sweep
 * kernel or the dK / dV kernel of the two-kernel path), 0 when it would draw every decision again - a caller saves the buffer then */
int ptamd_attention_bwd_reads_keep_bits(int B, int L, int H, int dk, int arith);

/* column sums: out[N] (+)= sum_t x[t,N]   (bias gradients) */
size_t ptamd_colsum_workspace_bytes(int N);
int ptamd_colsum(const float *x, int64_t T, int N, int ldx, int accumulate, float *out, void *workspace,
                 size_t workspace_bytes, void *stream);
/* elementwise backward helpers.
 *   relu_dropout_bwd: y = dropout(relu(.)) -> dx = dy * (y > 0) / (1-p)   (y > 0 iff active AND kept)
 *   tanh_bwd:         dx = dy * (1 - y*y)
 *   dropout_bwd:      dx = dy * mask / (1-p), mask regenerated from (seed, stream_id) exactly as the GEMM
 *                     epilogue drew it: 16-bit field drop_field(row) of the counter hash pt_rand4(seed, drop_call_index(row, col,
 *                     cols), stream_id) of csrc/common.h */
int ptamd_relu_dropout_bwd(const float *dy, const float *y, int64_t n, float dropout_p, float *dx, void *stream);
int ptamd_tanh_bwd(const float *dy, const float *y, int64_t n, float *dx, void *stream);
int ptamd_dropout_bwd(const float *dy, int64_t rows, int cols, float dropout_p, uint64_t seed, uint32_t stream_id,
                      float *dx, void *stream);

/* ------------------------------------------------------------------ conv-enc front end
 * torch.nn.Conv1d(C, Co, k, padding=(k-1)/2) of models/convolutional_encoder.py:92-129 as im2col + ptamd_gemm:
 *   col [T, k*Cp] = im2col(x [T, C], row stride ldx)   with Cp = C rounded up to 4 (zero padded), k odd
 *   y   [T, Co]   = col @ W2^T + bias                  W2 [Co, k*Cp] = pack(W [Co, C, k])
 *   dx = col2im(dy @ W2),  dW += unpack(dy^T @ col),  db += colsum(dy)
 * Windows never cross a protein boundary (rows are grouped in B proteins of L residues). */
int ptamd_im2col1d(const float *x, int ldx, int B, int L, int C, int k, float *col, void *stream);
int ptamd_col2im1d(const float *dcol, int B, int L, int C, int k, float *dx, int lddx, void *stream);
int ptamd_conv_weight_pack(const float *w, int Co, int C, int k, float *w2, void *stream);
int ptamd_conv_weight_unpack_add(const float *dw2, int Co, int C, int k, float *dw, void *stream);
/* x [T, Cp] = one_hot(seq, C) (convolutional_encoder.py:110-111, use_embedding=False) */
int ptamd_onehot(const int64_t *seq, int64_t T, int C, float *x, void *stream);
/* y = x + dropout(x + pe[pos]) and its adjoint (convolutional_encoder.py:118-119 with Sublayers.py:59-62) */
int ptamd_posenc_add_fwd(const float *x, const float *pe, int B, int L, int D, float dropout_p, uint64_t seed, float *y,
                         void *stream);
int ptamd_posenc_add_bwd(const float *dy, int64_t n, float dropout_p, uint64_t seed, float *dx, void *stream);

/* ------------------------------------------------------------------ optimizer (train.py:41-46,371-381)
 * clip_grad_norm_(params, max_norm) + SGD(lr, weight_decay) / Adam(betas, eps, weight_decay) over ONE flat
 * fp32 parameter buffer.  sqnorm: out[0] = sum g^2 (zeroed inside, deterministic two-stage reduction). */
size_t ptamd_grad_sqnorm_workspace_bytes(void);
int ptamd_grad_sqnorm(const float *g, int64_t n, float *out, void *workspace, size_t workspace_bytes, void *stream);
/* clip coefficient = min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) computed on device from sqnorm[0];
 * max_norm <= 0 disables clipping */
int ptamd_sgd_step(float *w, float *g, int64_t n, const float *sqnorm, float max_norm, float lr,
                   float weight_decay, int zero_grad, void *stream);
int ptamd_adam_step(float *w, float *g, float *m, float *v, int64_t n, const float *sqnorm, float max_norm,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step, int zero_grad, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PTAMD_H */
