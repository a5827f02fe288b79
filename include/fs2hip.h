/* fs2hip -- C ABI of the MI355X (gfx950) kernels behind the FastSpeech2
 * feature-prediction path.
 *
 * The reference (EveryVoiceTTS/FastSpeech2_lightning) has no FFI: its device work
 * is ATen ops called from Python modules.  Each entry point below names the
 * reference call site(s) (file:line, relative to the reference checkout) whose
 * ATen ops it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Dropout: element i keeps iff its 16-bit field of hash(seed + *drop_step * c, i >> 1) >= p * 2^16, regenerated
 * (never stored) by the backward kernels; `drop_step` is a device-resident counter so that a
 * captured hipGraph draws a fresh mask on every replay.
 *
 * Conventions (all entry points):
 *   - raw device pointers into caller-owned dense row-major buffers; activations
 *     are (B, T, C) = matrices of B*T rows and C contiguous columns, fp32;
 *   - no allocation, no host synchronisation, no global state: everything is
 *     enqueued on `stream` (a hipStream_t passed as void*);
 *   - workspaces are caller-provided;
 *   - return value: 0 on success, otherwise the hipError_t of the launch, or
 *     FS2HIP_EINVAL (-22) when the arguments fail the host-side shape checks.
 */
#ifndef FS2HIP_H
#define FS2HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS2HIP_EINVAL (-22)

/* activation codes */
enum { FS2_ACT_NONE = 0, FS2_ACT_RELU = 1, FS2_ACT_SILU = 2, FS2_ACT_TANH = 3 };

/* GEMM epilogues (v = alpha * acc + bias[n]) */
enum {
  FS2_EPI_STORE = 0, /* C = v                                                         */
  FS2_EPI_ACT = 1,   /* out_pre = v (optional); C = dropout(act(v))                   */
  FS2_EPI_RESID = 2, /* C = resid + res_scale * dropout(v)                            */
  FS2_EPI_DACT = 3   /* C = v * act'(aux) * dropmask   (backward through ACT)         */
};

int fs2hip_version(void);

/* ------------------------------------------------------------------------------------
 * GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32; bf16 operands on request):  C[Mc][Nc] = epi( sum_r A(m,r) * B(r,n) )
 *
 * Replaces every nn.Linear / pointwise nn.Conv1d / k-tap nn.Conv1d contraction on the
 * path, forward and backward:
 *   torchaudio Conformer FFN / in_proj / out_proj / pointwise convs (call sites
 *   fs2/model.py:193, :241), fs2/blocks.py:14-16 (pointwise), fs2/layers.py:30-38 (k-tap
 *   conv of the variance predictors), fs2/layers.py:204-212 (PostNet convs),
 *   fs2/attn/attention.py:220-227 (aligner projections), fs2/model.py:244 (mel_linear).
 *
 *   a_kcontig = 1: A stored [Mc][R] (lda)      0: A stored [R][Mc] (lda)
 *   b_kcontig = 1: B stored [Nc][R] (ldb)      0: B stored [R][Nc] (ldb)
 *   Conv taps (rows are (b, t), t = row % T):
 *     shift_operand = 0: R = taps * Rper; reduction step (tap, k) reads A row m + s(tap)
 *                        (zero when t + s(tap) is outside [0, T)) and B + tap*b_tap_stride;
 *                        A must be a_kcontig with Rper columns.
 *     shift_operand = 1: (weight gradient) one launch per tap j = blockIdx.z / splitk:
 *                        B row r + s(j) (zero outside), C + j*c_tap_stride.
 *     s(tap) = tap * tap_mul + tap_add.
 *   splitk > 1 (only with a_kcontig = b_kcontig = 0): R is cut in `splitk` chunks, partial
 *     tiles go to workspace[split][taps][Mc*Nc].  With `counters` == NULL fs2hip_reduce_slabs finishes; with
 *     `counters` (FS2_SPLITK_COUNTERS zero-initialised ints, one set per stream that runs split GEMMs) the workgroup
 *     that delivers the LAST slab of an output tile sums the tile's slabs in slab order (the same sum, bit for bit) and
 *     writes C itself, then re-arms the tile's counter: no second launch.
 * ------------------------------------------------------------------------------------ */
typedef struct {
  const float* A;
  const float* B;
  float* C;
  int Mc, Nc, R;
  int lda, ldb, ldc;
  int a_kcontig, b_kcontig;
  int taps, T, tap_mul, tap_add, shift_operand;
  long long b_tap_stride, c_tap_stride;
  const float* bias;
  int epi, act;
  float alpha;
  const float* resid;
  int ldr;
  float res_scale;
  const float* aux;
  int ldaux;
  float* out_pre;
  int ldpre;
  float drop_p;
  unsigned long long drop_seed;
  const unsigned long long* drop_step; /* device step counter mixed into the seed, or NULL */
  int splitk;
  float* workspace;
  int tile; /* 0 = heuristic; workgroup tile chosen by the host autotuner: 1/2/3 = 128x128, 128x64, 64x64 on the
               register-staged BK=16 core, 4..9 = direct-to-LDS BK=32 core (128x128, 128x64, 64x64 with a 3/3/4-stage LDS
               ring; 7/8/9 = 64x64, 128x64, 128x128 with 2 stages and more workgroups per CU), 10/11/12 = persistent
               64x64, 128x64, 128x128, 13/14/15 = persistent 128x128, 128x64, 64x64 whose last partial round of tiles
               is cut along the reduction and finished (sum + the same epilogue) by a second pass (needs `workspace`) */
  long long workspace_floats; /* capacity of `workspace` */
  int* counters; /* splitk > 1: per-output-tile arrival counters (see above), or NULL */
  int operand_bf16; /* 0: fp32 MFMA (the default, the parity path).  1: "bf16-mixed" -- A and B stay fp32 in memory and
                       in LDS, are rounded to bf16 (RNE) in registers and multiplied with v_mfma_f32_32x32x16_bf16;
                       accumulation, bias/epilogue and C stay fp32.  2: "32-split" -- fp32 accuracy on the bf16 pipe:
                       every operand value is cut exactly into three bf16 planes in registers (x = x0 + x1 + x2) and a
                       product is the six partial products a_i b_j with i + j <= 2, accumulated in fp32, smallest first
                       (what is dropped is below the rounding of the fp32 product): the error bound of mode 0 at
                       2.7x its matrix-pipe rate.  Only the direct-to-LDS cores (tile >= 4) carry modes 1 and 2;
                       shapes those cores refuse run in fp32.  3: bf16 operand STORAGE -- A and B point to bf16
                       (k-contiguous rows: a_kcontig = b_kcontig = 1; lda, ldb, R, b_tap_stride in bf16 elements, all
                       multiples of 8, tap widths multiples of 64; no split-K), C, bias and the epilogue tensors stay
                       fp32; tiles 4..9.  4: the bf16-storage core (gemm_bf16.hip, gemm_bf16p.hip; tiles 20, 22..25) -- A and B point to bf16 in
                       ANY of the three orientations (forward a_kcontig = b_kcontig = 1; data gradient a_kcontig = 1 with
                       the weight as stored, b_kcontig = 0; weight gradient both 0, split-K allowed), lda / ldb / R /
                       b_tap_stride / c_tap_stride in elements; R, lda, ldb multiples of 8, the row count of a
                       reduction-major operand a multiple of 8, Nc and ldc multiples of 4; tap widths multiples of 64
                       (shift_operand = 0) / T >= 64 (shift_operand = 1); bias, resid and split-K slabs fp32 */
  int io_bf16;   /* operand_bf16 == 4 only.  bit 0: C and out_pre are bf16 (ldc / ldpre in elements; the activation of
                    FS2_EPI_ACT is applied to the ROUNDED pre-activation, i.e. to what the backward pass reads back);
                    bit 1: aux is bf16 */
  float* colsum; /* weight gradient (a_kcontig = b_kcontig = 0) only, or NULL: receives the column sums of A over the
                    reduction -- the bias gradient dY^T . 1 of the layer whose weight gradient this launch computes -- as
                    colsum[split][Mc] partial sums (splitk rows; the caller adds them).  Direct-to-LDS cores (tiles 4..12)
                    and the bf16-storage core */
} Fs2GemmArgs;
#define FS2_SPLITK_COUNTERS 4096

int fs2hip_gemm(const Fs2GemmArgs* args, void* stream);

/* Up to FS2_GEMM_GROUP_MAX independent GEMMs in ONE launch: args[0 .. n) as for fs2hip_gemm, each with its own shapes,
 * pointers, split-K and epilogue.  Replaces runs of small launches that leave most of the chip idle one at a time: the
 * three variance predictors' pointwise convolutions of one layer (independent in a training step -- their inputs add
 * TARGET embeddings, fs2/variance_adaptor.py:309-352 -- forward, data and weight gradients; fs2/blocks.py:14-16) and the
 * eight weight gradients of one encoder Conformer layer (fs2/model.py:95-107: B x Ts rows, a quarter of a CU round each).
 * Every member computes exactly what fs2hip_gemm would (the member's workgroups run the same tile code on the member's own
 * arguments: same bits).  Members must share one kernel instance: the same operand orientations (a_kcontig, b_kcontig)
 * and operand_bf16 in {0, 4}, taps == 1, counters == NULL, drop_p == 0; `tile` (args[0].tile) is 0 or one of 7 / 8 (fp32)
 * and 22 / 23 / 26 (bf16 storage), the other members' `tile` is ignored.  n == 1 is fs2hip_gemm.  FS2HIP_EINVAL: the
 * members cannot share a launch (the caller launches them one by one); nothing has been enqueued then. */
#define FS2_GEMM_GROUP_MAX 8
int fs2hip_gemm_grouped(const Fs2GemmArgs* args, int n, void* stream);

/* Several row reductions in one launch: out0[c] (c < n0) / out1[c - n0] = sum over `rows` rows of src[r * stride + c],
 * c < n.  Finishes the partial sums of fs2hip_colsum (out == NULL) and fs2hip_layernorm_bwd (dgamma == dbeta == NULL),
 * i.e. the bias and LayerNorm parameter gradients (torch autograd's sum-to-size in the reference), whose only
 * consumer is the optimizer: the host batches them.  njobs <= FS2_REDUCE_MAX_JOBS. */
#define FS2_REDUCE_MAX_JOBS 48
typedef struct {
  const float* src;
  float* out0;
  float* out1;
  long long stride;
  int rows, n, n0, pad_;
} Fs2ReduceJob;
int fs2hip_reduce_rows_multi(const Fs2ReduceJob* jobs, int njobs, void* stream);

/* out[i] = sum_s slabs[s*slab_stride + i], i < n  (split-K / partial-sum finish) */
int fs2hip_reduce_slabs(const float* slabs, float* out, long long n, int nslabs,
                        long long slab_stride, void* stream);

/* The same for up to FS2_REDUCE_MAX_JOBS independent slab sets in one launch (the split-K finishes of
 * a whole backward pass: weight gradients are only read by the optimizer, fs2/model.py:530-549 via
 * Lightning's optimizer step).  `vec` is set by the library. */
typedef struct {
  const float* slabs;
  float* out;
  long long n, stride;
  int nslabs, vec;
} Fs2SlabJob;
int fs2hip_reduce_slabs_multi(const Fs2SlabJob* jobs, int njobs, void* stream);

/* column sums of a [M][N] matrix (bias gradients): partial[gy][N] then reduce_slabs.
 * partial must hold fs2hip_colsum_rows(M) * N floats. */
int fs2hip_colsum_rows(int M);
int fs2hip_colsum(const float* x, int ldx, int M, int N, float* partial, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the last dim (C <= 1024, C % 4 == 0), one wavefront per row.
 * nn.LayerNorm sites: Conformer (ffn/self_attn/conv/final norms), fs2/layers.py:43.
 * fwd: y = (x-mean)*rstd*gamma+beta, saves mean/rstd [M].
 * bwd: dx (+= dx_add if given), partial dgamma/dbeta -> [nblk][2][C] then reduce.
 * ------------------------------------------------------------------------------------ */
int fs2hip_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                         float* mean, float* rstd, int M, int C, float eps, void* stream);
int fs2hip_layernorm_bwd_blocks(int M);
int fs2hip_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                         const float* rstd, const float* dx_add, float* dx, float* partial,
                         float* dgamma, float* dbeta, int M, int C, void* stream);
/* The same with a second output for the sub-module below in the backward order (every Conformer sub-module is
 * x + scale * Dropout(f(LayerNorm(x))), torchaudio conformer.py; call sites fs2/model.py:95-119): dz = dz_scale *
 * dropmask * dx with the mask of that sub-module's output dropout (element index row * C + c, as fs2hip_axpby draws
 * it), and its column sums -- the gradient of f's last bias.  partial is [fs2hip_layernorm_bwd_blocks(M)][3][C]:
 * rows of dgamma | dbeta | colsum(dz) partial sums, finished by fs2hip_reduce_rows_multi. */
int fs2hip_layernorm_bwd_dz(const float* dy, const float* x, const float* gamma, const float* mean,
                            const float* rstd, const float* dx_add, float* dx, float* dz, float dz_scale,
                            float drop_p, unsigned long long drop_seed, const unsigned long long* drop_step,
                            float* partial, int M, int C, void* stream);
/* The variance predictors' Conv -> ReLU -> LayerNorm -> Dropout layers (fs2/layers.py:30-48, five per predictor, three
 * predictors: fs2/variance_adaptor.py:18-62) in two launches instead of five:
 *   fwd_drop: y = dropout(LayerNorm(x)) (mask at element index row * C + c, as fs2hip_axpby draws it over y);
 *   bwd_pred: dx = relu'(x) . LayerNormBackward(dropmask . dy) with x = the ReLU output the LayerNorm normalised
 *             (relu' = (x > 0)); partial is [fs2hip_layernorm_bwd_blocks(M)][2][C] (dgamma | dbeta partial sums). */
int fs2hip_layernorm_fwd_drop(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                              float* rstd, int M, int C, float eps, float drop_p, unsigned long long drop_seed,
                              const unsigned long long* drop_step, void* stream);
/* dx_bf16 != 0: dx is written as bf16 (bf16 operand storage: the layer's weight- and data-gradient GEMMs read it) */
int fs2hip_layernorm_bwd_pred(const float* dy, const float* x, const float* gamma, const float* mean,
                              const float* rstd, void* dx, int dx_bf16, float* partial, int M, int C, float drop_p,
                              unsigned long long drop_seed, const unsigned long long* drop_step, void* stream);
/* bf16 on either side of a LayerNorm ("bf16-mixed" with bf16 activation storage: the normalised activations and the
 * gradients between GEMMs exist only as the bf16 operands those GEMMs read).
 *   fwd_b: y is bf16.
 *   bwd_x: flags bit 0: dy is bf16 (the result of a data-gradient GEMM); bit 1: dz is bf16.  dz == NULL: no second
 *          output and partial is [nblk][2][C]; else [nblk][3][C] (the column sums are those of the rounded dz).
 *          dx_add, dx stay fp32 (the residual stream).  The caller finishes the partial sums. */
int fs2hip_layernorm_fwd_b(const float* x, const float* gamma, const float* beta, void* y_bf16,
                           float* mean, float* rstd, int M, int C, float eps, void* stream);
int fs2hip_layernorm_bwd_x(const void* dy, const float* x, const float* gamma, const float* mean,
                           const float* rstd, const float* dx_add, float* dx, void* dz, float dz_scale,
                           float drop_p, unsigned long long drop_seed, const unsigned long long* drop_step,
                           float* partial, int M, int C, int flags, void* stream);

/* ------------------------------------------------------------------------------------
 * Multi-head self-attention with key-padding mask (flash style; fp32 MFMA: v_mfma_f32_32x32x2_f32 for
 * head dims 64 / 128 -- attention2.hip -- and 16x16x4 for head dims 16 / 32).
 * Replaces nn.MultiheadAttention's scaled-dot-product core inside torchaudio's
 * ConformerLayer (call sites fs2/model.py:193, :241).
 *   qkv  [B*T][3*H*HD]  in_proj output (q | k | v);  lens [B] int32 (keys >= lens[b] masked)
 *   o    [B*T][H*HD];   lse [B][H][T] (log-sum-exp per query, saved for the backward)
 *   dropout acts on the normalised probabilities (attention dropout), mask regenerated
 *   from (seed, b, h, q, k) in the backward.  HD in {16, 32, 64, 128}.
 * bwd: delta = scratch of 2*B*H*T + 4 floats ({lse', delta'} pairs per row and head); dqkv [B*T][3*H*HD]
 *      fully written.
 * operand_bf16 = 1 ("bf16-mixed"): the operands of all five products (Q, K, V, dO, P, dS) are rounded to bf16 for
 *   v_mfma_f32_16x16x32_bf16; scores, softmax statistics, accumulators and outputs stay fp32.
 * ------------------------------------------------------------------------------------ */
int fs2hip_attention_fwd(const float* qkv, const int* lens, float* o, float* lse, int B, int T, int H,
                         int HD, float drop_p, unsigned long long drop_seed,
                         const unsigned long long* drop_step, int operand_bf16, void* stream);
int fs2hip_attention_bwd(const float* qkv, const int* lens, const float* o, const float* dout,
                         const float* lse, float* delta, float* dqkv, int B, int T, int H, int HD,
                         float drop_p, unsigned long long drop_seed,
                         const unsigned long long* drop_step, int operand_bf16, void* stream);

/* The fp32 backward pass (operand_bf16 = 0) with dS written out by the dK/dV kernel and dQ = scale * dS . K as a product
 * of its own -- 5 products per block instead of the 9 of two kernels that each recompute S and dP.
 * Head dims for which fs2hip_attention_bwd_spill_supported() returns 1 (64, 128).
 *   aux: 2*B*H*T + 4 floats; ds: scratch of at least B*H*T*(T rounded up to 32) floats (ds_floats = its size).
 * Same results as fs2hip_attention_bwd up to the rounding of S (the scale is folded into K there, into Q in the
 * recomputing dQ kernel). */
int fs2hip_attention_bwd_spill_supported(int HD);
int fs2hip_attention_bwd_spill(const float* qkv, const int* lens, const float* o, const float* dout,
                               const float* lse, float* aux, float* ds, long long ds_floats, float* dqkv,
                               int B, int T, int H, int HD, float drop_p, unsigned long long drop_seed,
                               const unsigned long long* drop_step, void* stream);

/* ... and with the forward pass's masked scores kept for it: fs2hip_attention_fwd_s writes them (log2 units, scale folded
 * in, -inf at masked keys) into `scores` (at least B*H*T*(T rounded up to 32) floats, kept until the backward pass), and
 * fs2hip_attention_bwd_spill_s's dK/dV kernel reads them instead of recomputing K.Q^T: 3 + 1 products per block in the
 * backward pass, and the backward's probabilities are the forward's to the bit.  operand_bf16: 0 (exact fp32) or 2
 * ("32-split": the scores come from the three-plane products; the backward pass is the fp32 one either way). */
int fs2hip_attention_fwd_s(const float* qkv, const int* lens, float* o, float* lse, float* scores,
                           long long score_floats, int B, int T, int H, int HD, float drop_p,
                           unsigned long long drop_seed, const unsigned long long* drop_step, int operand_bf16,
                           void* stream);
int fs2hip_attention_bwd_spill_s(const float* qkv, const int* lens, const float* o, const float* dout,
                                 const float* lse, const float* scores, float* aux, float* ds, long long ds_floats,
                                 float* dqkv, int B, int T, int H, int HD, float drop_p,
                                 unsigned long long drop_seed, const unsigned long long* drop_step, void* stream);

/* The same attention on tensors that ARE bf16 in memory (precision "bf16-mixed" with bf16 activation
 * storage; torch.autocast(bfloat16) around nn.MultiheadAttention, call sites fs2/model.py:193, :241):
 * qkv, o, dout, dqkv are bf16 with the shapes above, lse and the scratch `aux` (2*B*H*T + 4 floats)
 * are fp32.  Head dims for which fs2hip_attention_b_supported() returns 1 (128).  The dropout keep
 * mask is the one fs2hip_attention_fwd generates for the same (drop_p, drop_seed, drop_step). */
int fs2hip_attention_b_supported(int HD);
int fs2hip_attention_fwd_b(const void* qkv, const int* lens, void* o, float* lse, int B, int T, int H,
                           int HD, float drop_p, unsigned long long drop_seed,
                           const unsigned long long* drop_step, void* stream);
int fs2hip_attention_bwd_b(const void* qkv, const int* lens, const void* o, const void* dout,
                           const float* lse, float* aux, void* dqkv, int B, int T, int H, int HD,
                           float drop_p, unsigned long long drop_seed,
                           const unsigned long long* drop_step, void* stream);

/* fs2hip_attention_bwd_b with dS -- masked and rounded to bf16: the operand the dQ product consumes -- written out by the
 * dK/dV kernel and dQ = scale * dS . K as a product of its own: S, dP and the softmax / dropout arithmetic are computed once
 * per backward pass instead of twice.  ds: scratch of at least B*H*T*(T rounded up to 32) bf16 elements (ds_elems). */
int fs2hip_attention_bwd_b_spill(const void* qkv, const int* lens, const void* o, const void* dout,
                                 const float* lse, float* aux, void* ds, long long ds_elems, void* dqkv,
                                 int B, int T, int H, int HD, float drop_p, unsigned long long drop_seed,
                                 const unsigned long long* drop_step, void* stream);

/* ------------------------------------------------------------------------------------
 * Depthwise Conv1d over time on (B, T, C), 'same' padding, K in {3,5,7,9,15,31}; w is [K][C].
 * glu = 1: the input has 2C columns (value | gate) and a = value * sigmoid(gate) is formed on
 * the fly (torchaudio conv module: Conv1d(D,2D,1) -> GLU -> depthwise Conv1d).
 * stats = 1: also writes per-workgroup (mean, sum of squared deviations) per channel for BatchNorm:
 *            partial[fs2hip_dwconv_blocks(B,T)][2][C]; a workgroup covers fs2hip_dwconv_part_rows()
 *            time steps of one utterance (bn_finalize: part_rows = that, group_rows = T).
 * Also the depthwise half of fs2/blocks.py:8-13.
 * bwd: dx has the layout of x; partial [blocks][K+1][C]; dw [K][C] and dbias [C] are finished.
 * ------------------------------------------------------------------------------------ */
int fs2hip_dwconv_blocks(int B, int T);
int fs2hip_dwconv_part_rows(void);
int fs2hip_dwconv_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, float* partial,
                      int B, int T, int C, int K, int glu, int stats, void* stream);
int fs2hip_dwconv_bwd(const float* dy, const float* x, int ldx, const float* w, float* dx, float* partial,
                      float* dw, float* dbias, int B, int T, int C, int K, int glu, void* stream);
/* bf16 tensors (precision "bf16-mixed" with bf16 activation storage; what torch.autocast hands the convolution
 * module of torchaudio's ConformerLayer, call sites fs2/model.py:193, :241), GLU form only:
 *   fwd_b, io_bf16 = 1: x and y are bf16 (the BatchNorm statistics are those of the rounded y);
 *          io_bf16 = 2: x fp32, y bf16, plain (non-GLU) form without statistics -- the variance predictors' depthwise
 *          layer (fs2/blocks.py:4-19) feeding its pointwise GEMM's bf16 operand;
 *   bwd_b, dx_bf16 bit 0: dx is written as bf16 (layout and leading dimension of x); bit 1: dy and x are bf16. */
int fs2hip_dwconv_fwd_b(const void* x, int ldx, const float* w, const float* bias, void* y, float* partial,
                        int B, int T, int C, int K, int glu, int stats, int io_bf16, void* stream);
int fs2hip_dwconv_bwd_b(const void* dy, const void* x, int ldx, const float* w, void* dx, int dx_bf16, float* partial,
                        float* dw, float* dbias, int B, int T, int C, int K, int glu, void* stream);

/* ------------------------------------------------------------------------------------
 * BatchNorm1d over the channels of [M][C] (+ activation + dropout), fs2/layers.py:204-212 and
 * the Conformer conv module.  stats is [4][C]: scale, shift, mean, invstd.
 *   colstats   : partial[fs2hip_colstats_parts(M)][2][C] = per-stripe (mean, sum of squared
 *                deviations) over stripes of fs2hip_colstats_part_rows(M) rows, accumulated on
 *                pivot-shifted values (torch's BatchNorm is Welford: a plain E[x^2]-E[x]^2 in fp32
 *                loses the variance of a channel whose |mean| >> std)
 *   finalize   : training: Chan merge of the parts in fp64 (the parts tile `count` rows in groups
 *                of group_rows rows, each in stripes of part_rows) + running-stat update
 *                (momentum, unbiased variance); eval: running statistics
 *   bn_act_fwd : out = dropout(act(y*scale + shift))
 *   bn_act_bwd : dy (grad of y), dgamma, dbeta; partial as colstats, coef [2][C] scratch
 * ------------------------------------------------------------------------------------ */
int fs2hip_colstats_parts(int M);
int fs2hip_colstats_part_rows(int M);
int fs2hip_colstats(const float* y, int M, int C, float* partial, void* stream);
/* the same over a bf16 tensor when in_bf16 != 0 (a convolution's bf16 result under bf16 activation storage) */
int fs2hip_colstats_b(const void* y, int M, int C, float* partial, int in_bf16, void* stream);
int fs2hip_bn_finalize(const float* partial, int nparts, long long count, int part_rows, int group_rows,
                       const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float momentum, float eps, int training, float* stats, int C, void* stream);
int fs2hip_bn_act_fwd(const float* y, const float* stats, float* out, int M, int C, int act, float drop_p,
                      unsigned long long drop_seed, const unsigned long long* drop_step, void* stream);
int fs2hip_bn_act_bwd(const float* dout, const float* y, const float* stats, float* partial, float* coef,
                      float* dgamma, float* dbeta, float* dy, int M, int C, int act, float drop_p,
                      unsigned long long drop_seed, const unsigned long long* drop_step, int training,
                      void* stream);
/* The same two with a bf16 form of the output (out_bf16 / dy_bf16, [M][C] bf16) for a consuming GEMM that reads bf16
 * operands from memory (Fs2GemmArgs.operand_bf16 == 4).  Either output pointer may be NULL (not both): with out / dy
 * NULL the tensor exists only in bf16. */
/* in_bf16: the inputs are bf16 tensors as well -- fwd: y (any non-zero value); bwd: bit 0 y, bit 1 dout. */
int fs2hip_bn_act_fwd_b(const void* y, const float* stats, float* out, void* out_bf16, int M, int C, int act,
                        float drop_p, unsigned long long drop_seed, const unsigned long long* drop_step, int in_bf16,
                        void* stream);
int fs2hip_bn_act_bwd_b(const void* dout, const void* y, const float* stats, float* partial, float* coef,
                        float* dgamma, float* dbeta, float* dy, void* dy_bf16, int M, int C, int act, float drop_p,
                        unsigned long long drop_seed, const unsigned long long* drop_step, int training, int in_bf16,
                        void* stream);

/* ------------------------------------------------------------------------------------
 * Positional table / embeddings / bucketize  (fs2/layers.py:123-140, fs2/model.py:183-193,
 * :233-241, fs2/variance_adaptor.py:197-205)
 * ------------------------------------------------------------------------------------ */
int fs2hip_posenc_table(const float* inv_freq, float* table, int T, int D, void* stream);
int fs2hip_add_posenc(const float* x, const float* table, const int* lens, float* out, int B, int T, int D,
                      void* stream);
int fs2hip_embedding_fwd(const int* idx, const float* W, float* out, int M, int V, int D, void* stream);
/* embedding backward = fs2hip_onehot + fs2hip_gemm (dW = onehot^T @ dy; split-K weight-gradient mode):
 * out[m][v] = (idx[m] == v && v != padding_idx), row length Vp (multiple of 4, >= vocabulary) */
int fs2hip_onehot(const int* idx, float* out, int M, int Vp, int padding_idx, void* stream);
/* out = x + W[lower_bound(bins, val*control)]; idx_out (int32, bit-exact vs torch.bucketize) optional */
int fs2hip_bucket_embed_add(const float* val, float control, const float* bins, int NB, const float* W,
                            const float* x, float* out, int* idx_out, int M, int D, void* stream);

/* ------------------------------------------------------------------------------------
 * LengthRegulator (fs2/variance_adaptor.py:65-81): out[b, t] = x[b, j] for the token j whose
 * duration segment contains frame t (zero rows past the total), bit-exact indexing.
 *   cum [B][Ts] inclusive cumulative durations (written), out_lens [B] = min(total, Tm),
 *   src_idx [B][Tm] source token or -1 (optional), posenc_table [>=Tm][D] added on valid frames
 *   (optional: fuses fs2/model.py:233-241).
 * bwd: dx[b, j] = sum of dy over the token's frame segment (contiguous, no atomics).
 * ------------------------------------------------------------------------------------ */
int fs2hip_length_regulate_fwd(const float* x, const int* dur, const float* posenc_table, float* out,
                               int* cum, int* out_lens, int* src_idx, int B, int Ts, int Tm, int D,
                               void* stream);
int fs2hip_length_regulate_bwd(const float* dy, const int* cum, float* dx, int B, int Ts, int Tm, int D,
                               void* stream);
/* cum [B][Ts] inclusive cumulative durations, out_lens [B] = min(total, Tm) (the first half of the forward).
 * expect_lens / mismatch (both or neither) + optional bad_count: the consistency check of the aligner's durations,
 * fs2/variance_adaptor.py:289-305 -- mismatch[b] = (total of utterance b != expect_lens[b]); each mismatch also adds 1
 * to *bad_count, a persistent device word the host polls instead of the per-utterance flags (no sync in a step). */
int fs2hip_duration_cumsum(const int* dur, int* cum, int* out_lens, const int* expect_lens, int* mismatch,
                           int* bad_count, int B, int Ts, int Tm, void* stream);

/* predictor head (fs2/variance_adaptor.py:53-62): out[m] = (x[m,:].w + b) * (t < lens[b]) */
int fs2hip_rowdot_fwd(const float* x, const float* w, const float* bias, const int* lens, float* out, int M,
                      int T, int C, void* stream);
int fs2hip_rowdot_blocks(int M);
int fs2hip_rowdot_bwd(const float* dout, const float* x, const float* w, const int* lens, float* dx,
                      float* partial, float* dw, float* dbias, int M, int T, int C, void* stream);

/* masked MSE (kind 0) / MAE (kind 1) of fs2/loss.py:44-106 with its padded-mean denominator;
 * tgt_int != NULL: target = log(tgt_int + 1) (duration loss).  Writes loss_out[0] and, if
 * dpred != NULL, d(loss)/d(pred).  partial: >= 1024 floats. */
int fs2hip_masked_loss(const float* pred, const float* tgt, const int* tgt_int, const int* lens, int B, int T,
                       int C, int kind, float weight, float* dpred, float* partial, float* loss_out,
                       void* stream);

/* ------------------------------------------------------------------------------------
 * Optimizer (fs2/model.py:530-549 AdamW + fs2/noam.py:20-26 + clip of fs2/cli/train.py:38).
 * `state` is a 32-byte device record { uint64 step; float lr, bc1, bc2, clip_coef, grad_norm, pad }
 * advanced on the device so that a captured hipGraph needs no new arguments per step; its first
 * 8 bytes double as the `drop_step` counter of the dropout kernels.
 * ------------------------------------------------------------------------------------ */
int fs2hip_step_advance(void* state, float base_lr, float warmup, float beta1, float beta2, void* stream);
int fs2hip_grad_clip_coef(const float* grad, long long n, float max_norm, float grad_scale, float* partial,
                          void* state, void* stream);
int fs2hip_adamw_step(float* p, const float* g, float* m, float* v, long long n, const void* state,
                      float beta1, float beta2, float eps, float weight_decay, void* stream);

/* bf16 operand storage (Fs2GemmArgs.operand_bf16 == 3; Lightning's `precision="bf16-mixed"` autocast copies in the
 * reference): dst = bf16(src), round to nearest even, n % 8 == 0; and dst[c][r] = bf16(src[r][c]) for a [rows][cols]
 * matrix with rows ld_src apart, dst rows ld_dst >= rows elements apart (pad columns zeroed), for `batch` matrices
 * stored back to back (the taps of a convolution weight). */
int fs2hip_cast_bf16(const float* src, void* dst, long long n, void* stream);
/* out[(b, t)][tap * C + c] = bf16(x[(b, t + dir * (tap - (taps - 1) / 2))][c]), zero outside [0, T): a k-tap 'same'
 * convolution's input rows side by side (x fp32 or bf16, C % 8 == 0, taps odd).  The PostNet's 80-mel-bin convolutions
 * (fs2/layers.py:143-212: first layer forward, last layer data gradient) then run as ONE plain K = taps * C GEMM on the
 * bf16-storage core; dir = +1 pairs with the weight as [taps * Cin][Cout], dir = -1 (data gradient) with the weight as
 * stored. */
int fs2hip_im2col_taps(const void* x, int x_bf16, void* out_bf16, int B, int T, int C, int taps, int dir, void* stream);
int fs2hip_transpose_cast_bf16(const float* src, int rows, int cols, int ld_src, void* dst, int ld_dst, int batch,
                               void* stream);
/* The same for up to FS2_TRANSPOSE_MAX_JOBS matrices in one launch: dst[c][r] = bf16(src[r][c]), src fp32 [rows][cols]
 * (dense), dst bf16 [cols][rows] (dense).  Keeps the transposed bf16 mirrors of the weights whose DATA gradient runs in
 * the forward orientation (dz . W with W [256][K'] -> W^T [K'][256], the out-projection / second pointwise convolution /
 * second feed-forward weights of a Conformer layer: torchaudio ConformerLayer, call sites fs2/model.py:193, :241), once
 * per optimizer step, beside the bf16 mirror of the flat parameter buffer. */
#define FS2_TRANSPOSE_MAX_JOBS 64
typedef struct {
  const float* src;
  void* dst;
  int rows, cols;
  int fp32_out; /* 0: dst is bf16; 1: dst is fp32 (the "32-true" mirrors of the same weights, tile 32 of fs2hip_gemm) */
  int pad_;
} Fs2TransposeJob;
int fs2hip_transpose_cast_bf16_multi(const Fs2TransposeJob* jobs, int njobs, void* stream);

/* out = a * x * dropmask + b * y (y may be NULL);  out[b,t,:] = x[b,t,:] + e[b,:] */
int fs2hip_axpby(const float* x, const float* y, float* out, long long n, float a, float b, float drop_p,
                 unsigned long long drop_seed, const unsigned long long* drop_step, void* stream);
int fs2hip_add_rowvec(const float* x, const float* e, float* out, int B, int T, int D, void* stream);
/* x *= *scalar (scalar in device memory; a no-op pass when it is exactly 1): the upstream gradient that autograd hands
 * to the node behind FastSpeech2.training_step's loss (fs2/model.py:384-390 returns the loss Lightning differentiates) */
int fs2hip_scale_dev(float* x, long long n, const float* scalar, void* stream);
/* out = dy * act'(aux) (aux = activation output for ReLU, input otherwise); bool mask t < lens[b]
 * (fs2/utils/heavy.py:11-15); out[0] = sum of n loss slots (fs2/loss.py:125) */
int fs2hip_dact_mul(const float* dy, const float* aux, float* out, long long n, int act, void* stream);
/* inference durations (fs2/variance_adaptor.py:360-366): out = int(max(rint(exp(logd) - 1) * control, 0)) */
int fs2hip_duration_round(const float* logd, float control, int* out, int n, void* stream);
int fs2hip_mask_from_lens(const int* lens, unsigned char* mask, int B, int T, void* stream);
int fs2hip_sum_slots(const float* x, int n, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Learned alignment (fs2/attn/attention.py:195-251, fs2/attn/alignment.py:48-74,
 * fs2/attn/attention_loss.py:22-73, fs2/variance_adaptor.py:160-222, :267-268).
 * All attention maps are dense [B][T1 = frames][T2 = tokens] fp32.
 *   attn_dist       logits = -0.0005 * sum_c (q - k)^2      q [B][T1][C], k [B][T2][C]
 *   attn_softmax    logprob = log_softmax(logits) + log(prior + 1e-8); soft = softmax over keys < key_lens[b]
 *   mas             hard 0/1 map, hard_idx [B][T1] (token of each frame, -1 on padding), dur [B][T2] int32;
 *                   `in` = attn_soft (is_log 0) or log-probabilities (is_log 1); DP in fp32 adds/max only:
 *                   bit-exact against mas_width1 on the same input.  dirs_ws: B*T1*ceil(T2/32) uint32.
 *   avg_variance    phone-level mean of a frame-level track over each token's frames (non-zero count)
 *   attn_ctc_loss   AttentionCTCLoss value (loss_out[0], weight applied) and d/d logprob (may be NULL)
 *                   alpha_ws B*T1*(2*T2+1), lse_ws B*T1, nll_ws B floats
 *   attn_bin_loss   AttentionBinarizationLoss value and the coefficient the backward needs
 *   attn_softmax_bwd  d logits from d logprob (CTC) and the binarisation term (hard_idx, bin_coef)
 *   attn_dist_bwd   dq, dk from d logits (either may be NULL)
 * ------------------------------------------------------------------------------------ */
int fs2hip_attn_dist(const float* q, const float* k, float* logits, int B, int T1, int T2, int C, void* stream);
int fs2hip_attn_softmax(const float* logits, const float* prior, const int* key_lens, float* logprob,
                        float* soft, int B, int T1, int T2, void* stream);
int fs2hip_mas(const float* in, int is_log, const int* in_lens, const int* out_lens, float* hard,
               int* hard_idx, int* dur, unsigned* dirs_ws, int B, int Tm, int Ts, void* stream);
int fs2hip_avg_variance(const float* var, const int* cum, float* out, int B, int Tm, int Ts, void* stream);
int fs2hip_attn_ctc_loss(const float* logprob, const int* key_lens, const int* query_lens, float* alpha_ws,
                         float* lse_ws, float* nll_ws, float* dlogprob, float weight, float* loss_out, int B,
                         int Tm, int Ts, void* stream);
int fs2hip_attn_bin_loss(const float* soft, const int* hard_idx, float* partial, float weight, float* loss_out,
                         float* bin_coef, int B, int Tm, int Ts, void* stream);
int fs2hip_attn_softmax_bwd(const float* logits, const float* soft, const float* dlogprob, const int* hard_idx,
                            const float* bin_coef, float* dlogits, int B, int T1, int T2, void* stream);
int fs2hip_attn_dist_bwd(const float* dlogits, const float* q, const float* k, float* dq, float* dk, int B,
                         int T1, int T2, int C, void* stream);

/* ------------------------------------------------------------------------------------
 * GST style encoder (fs2/gst/model.py:103-257, fs2/gst/attn.py:48-194; BASELINE config 5).
 *   conv2d_s2_*     3x3 / stride 2 / pad 1 / no bias Conv2d on channels-last [B][H][W][C];
 *                   weights [kh][kw][Cin][Cout]; output extent (n - 1) / 2 + 1 per spatial dim;
 *                   weight gradient through partial[parts][9*Cin*Cout]
 *   gru_gate_*      one nn.GRU time step (gate order r, z, n): gi rows (stride gi_stride) and gh
 *                   [B][3U] come from the GEMM; gates [B][4U] keeps r, z, n, gh_n for the backward
 *   gst_attn_*      softmax(q k^T / 8) v for one query per utterance against NT <= 32 style tokens,
 *                   `heads` <= 4 heads of 64 dims; dk_part / dv_part are per-utterance [B][NT][F]
 *   act_apply       out = act(x)  (tanh of the style tokens)
 * ------------------------------------------------------------------------------------ */
int fs2hip_conv2d_s2_fwd(const float* x, const float* w, float* y, int B, int H, int W, int Cin, int Cout,
                         void* stream);
int fs2hip_conv2d_s2_bwd_data(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Cout,
                              void* stream);
/* The same convolution as GEMM operands (Cin % 4 == 0): col[B*Ho*Wo][9*Cin] gathers the 3x3 stride-2 windows (zero
 * outside the image), col2im sums dcol back into dx[B][H][W][Cin].  Weight seen as [9*Cin][Cout].
 * im2col with Cin == 1 writes rows of 12 floats (9 taps + 3 zeros) for the first layer's weight gradient. */
int fs2hip_im2col_s2(const float* x, float* col, int B, int H, int W, int Cin, void* stream);
int fs2hip_col2im_s2(const float* dcol, float* dx, int B, int H, int W, int Cin, void* stream);
int fs2hip_conv2d_s2_wgrad_parts(int B, int H, int W);
int fs2hip_conv2d_s2_bwd_weight(const float* x, const float* dy, float* partial, float* dw, int B, int H, int W,
                                int Cin, int Cout, void* stream);
int fs2hip_gru_gate_fwd(const float* gi, long long gi_stride, const float* gh, const float* hprev, float* hnew,
                        float* gates, int B, int U, void* stream);
int fs2hip_gru_gate_bwd(const float* dh, const float* gates, const float* hprev, float* dgi, long long dgi_stride,
                        float* dgh, float* dhprev, int B, int U, void* stream);
int fs2hip_gst_attn_fwd(const float* q, const float* k, const float* v, float* p, float* ctx, int B, int NT,
                        int heads, void* stream);
int fs2hip_gst_attn_bwd(const float* dctx, const float* q, const float* k, const float* v, const float* p,
                        float* dq, float* dk_part, float* dv_part, int B, int NT, int heads, void* stream);
int fs2hip_act_apply(const float* x, float* out, long long n, int act, void* stream);

/* dst[0 .. nbytes) = byte, enqueued on the stream (what torch.zeros / Tensor.zero_() did inside a step: the loss slot
 * vector of fs2/model.py:387-389's terms, the GRU's initial state fs2/gst/model.py:140-147).  A launch-plan command like
 * every other entry point, which an ATen fill is not. */
int fs2hip_memset(void* dst, int byte, long long nbytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Launch plans: a training step's whole launch sequence enqueued by ONE call.
 *
 * The reference's step is Python all the way down (Lightning -> nn.Module.forward -> ATen dispatch,
 * fs2/model.py:153-268, :384-390); this build's eager step is Python down to the entry points above (~590 launches,
 * ~18 us of interpreter time each: the host needs 10 ms to enqueue an 11 ms bf16 step).  A plan is that sequence
 * recorded once per batch geometry -- every entry-point call with its arguments, on which of the two streams it
 * ran, and the fork / join events between them -- and replayed by fs2hip_plan_replay: the host cost of a step
 * becomes the launches themselves.  The recorded step's temporaries live in a memory pool of the plan's own (the
 * caller's business: fastspeech2_lightning_amd/plan.py), so the recorded addresses stay valid; inputs are copied into
 * the recorded input buffers; dropout masks, the learning-rate schedule and BatchNorm statistics advance through
 * device memory exactly as in the eager step, which is why a replay is bit-identical to it (tests/test_plan_gpu.py).
 *
 * op >= 0: index of an entry point (fs2hip_plan_op_id(name); the table is generated from this header by
 *          tools/gen_plan_thunks.py -> csrc/plan_thunks.inc); a[i] = its i-th argument, the stream excluded:
 *          pointers and integers as they are (int sign-extended), a float as its bit pattern in the low 32 bits;
 *          struct arguments (Fs2GemmArgs, job arrays) point to HOST copies the plan's owner keeps alive.
 * FS2_PLAN_SYNC: record events[a[0]] on stream a[1], make stream a[2] wait for it (stream indices into the replay call's
 *          stream table: 0 = the main stream, 1.. = the step's side streams).
 * ------------------------------------------------------------------------------------ */
#define FS2_PLAN_SLOTS 20
#define FS2_PLAN_SYNC (-1)
typedef struct {
  int op;      /* entry-point id or FS2_PLAN_SYNC */
  int stream;  /* index into the stream table of fs2hip_plan_replay: 0 main, 1.. side streams */
  unsigned long long a[FS2_PLAN_SLOTS];
} Fs2PlanCmd;

int fs2hip_plan_op_count(void);
/* id of an entry point by name, -1 when it is not a plan op (no stream parameter, unknown) */
int fs2hip_plan_op_id(const char* name);
/* n events for FS2_PLAN_SYNC commands (created without timing) into out[0..n); destroy frees them */
int fs2hip_plan_events_create(void** out, int n);
int fs2hip_plan_events_destroy(void* const* events, int n);
/* enqueues cmds[first .. last); on failure returns the failing command's code and writes its index to *failed_at */
int fs2hip_plan_replay(const Fs2PlanCmd* cmds, int first, int last, void* const* streams, int n_streams,
                       void* const* events, int n_events, int* failed_at);

#ifdef __cplusplus
}
#endif
#endif /* FS2HIP_H */
