/* fs2hip -- C ABI of the MI355X (gfx950) kernels behind the FastSpeech2
 * feature-prediction path.
 *
 * The reference (EveryVoiceTTS/FastSpeech2_lightning) has no FFI: its device work
 * is ATen ops called from Python modules.  Each entry point below names the
 * reference call site(s) (file:line, relative to the reference checkout) whose
 * ATen ops it replaces.  INTEGRATION.md shows the ctypes binding.
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
 * GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32):  C[Mc][Nc] = epi( sum_r A(m,r) * B(r,n) )
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
 *     tiles go to workspace[split][taps][Mc*Nc] and fs2hip_reduce_slabs finishes.
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
  int splitk;
  float* workspace;
} Fs2GemmArgs;

int fs2hip_gemm(const Fs2GemmArgs* args, void* stream);

/* out[i] = sum_s slabs[s*slab_stride + i], i < n  (split-K / partial-sum finish) */
int fs2hip_reduce_slabs(const float* slabs, float* out, long long n, int nslabs,
                        long long slab_stride, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* FS2HIP_H */
