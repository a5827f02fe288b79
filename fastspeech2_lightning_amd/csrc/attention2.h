// Second-generation fp32 attention kernels for head dims 64 / 128 (the Conformer's 2 x 128): see attention2.hip.
#pragma once
#include "common.h"

struct Attn2Args {
  const float* qkv;   // [B*T][3*D]: q | k | v
  const int* lens;    // [B] valid keys per utterance
  int B, T, H, HD;
  float scale;        // 1/sqrt(HD)
  Fs2Drop drop;       // dropout on the attention probabilities, element index ((b*H + h)*T + q)*Tp + key, Tp = T rounded up to even
  int planes;         // 0: fp32 MFMA.  1: operands rounded to bf16 ("bf16-mixed").  3: three exact bf16 planes ("32-split")
  long long* stamps;  // diagnostic builds only (-DFS2_ATTN_STAMPS, tools/probes/attn2_probe.hip): per-phase cycle sums
};

// true when the second-generation kernels take the shape (HD in {64, 128}; operand_bf16: 0 fp32, 1 bf16, 2 split)
bool fs2_attn2_supported(int HD, int operand_bf16);
// s_out (fp32 MFMA path, may be null): the masked scores for fs2_attn2_bwd_spill's s_in, B * H * T * (T rounded up to 32) floats
int fs2_attn2_fwd(const Attn2Args& a, float* o, float* lse, hipStream_t s, float* s_out = nullptr);
int fs2_attn2_bwd(const Attn2Args& a, const float* o, const float* dout, const float* lse, float* delta, float* dqkv,
                  hipStream_t s);

// the same backward pass with dS spilled by the dK/dV kernel and dQ = scale * dS . K as a product of its own (fp32 MFMA
// path); `ds` holds fs2_attn2_bwd_spill_elems(a) floats (0: this shape / operand mode does not take the path)
long long fs2_attn2_bwd_spill_elems(const Attn2Args& a);
int fs2_attn2_bwd_spill(const Attn2Args& a, const float* o, const float* dout, const float* lse, float* aux, float* ds,
                        float* dqkv, hipStream_t s, const float* s_in = nullptr);

// bf16-storage family (attention_bf16.hip): qkv / o / dout / dqkv are bf16 tensors of the same shapes; `a.qkv` is unused
bool fs2_attnb_supported(int HD);
int fs2_attnb_fwd(const Attn2Args& a, const void* qkv, void* o, float* lse, hipStream_t s);
int fs2_attnb_bwd(const Attn2Args& a, const void* qkv, const void* o, const void* dout, const float* lse, float* aux,
                  void* dqkv, hipStream_t s, void* ds = nullptr);
