// Step tail on flat f32 buffers (trainer.py:67-87, train_setup.py:42-53, utils.py:169-225):
//   global grad-norm (deterministic two-stage) -> clip coefficient + device-side skip predicate -> fused
//   AdamW + EMA.  All schedule arithmetic (LambdaLR warm-up, Adam bias correction, EMA warm-up decay) is derived on
//   the device from the count of successful optimiser steps, so the host never has to read the skip decision.
#include "common.h"

namespace cgen {

// state_dev: [0] sum_sq  [1] grad_norm  [2] clip_coef  [3] skip_flag  [4] n_skipped  [5] opt_steps (successful steps so far)

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, int64_t count, float* partial) {
  __shared__ float sm[4];
  float a = 0.f;
  // fixed assignment of elements to (block, thread, iteration): bitwise reproducible run to run
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) a += g[i] * g[i];
  const float t = block_sum_256(a, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void clip_decide_kernel(const float* partial, int nblk, const float* out3, float max_norm,
                                                          float skip_norm, float* state) {
  __shared__ float sm[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) a += partial[i];
  const float t = block_sum_256(a, sm);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(t);
    state[0] = t;
    state[1] = norm;
    const float c = max_norm / (norm + 1e-6f);
    state[2] = c < 1.f ? c : 1.f;
    bool skip = !(norm < skip_norm);
    if (out3) skip = skip || isnan(out3[1]) || isnan(out3[2]);
    state[3] = skip ? 1.f : 0.f;
    if (skip) state[4] += 1.f;
    if (skip && !(norm <= 3.402823466e38f)) state[6] += 1.f;  // ... of which for a NON-FINITE norm (an f16 overflow: what the loss-scale back-off counts)
  }
}

struct AdamP {
  float* p; const float* g; float* m; float* v; float* ema; int64_t count;
  float lr, beta1, beta2, eps, wd, ema_beta;
  int warmup_steps, ema_update_after;
  const float* state;
};

__global__ __launch_bounds__(256) void adamw_ema_kernel(AdamP a) {
  if (a.state[3] != 0.f) return;  // skipped update: parameters, moments and EMA all untouched
  const float clip = a.state[2];
  const int t0 = (int)a.state[5];  // successful steps before this one
  const float t = (float)(t0 + 1);
  const float lr = a.lr * (t0 > a.warmup_steps ? 1.f : (float)t0 / (float)a.warmup_steps);  // LambdaLR(linear_warmup)
  const float bc1 = 1.f - powf(a.beta1, t);
  const float bc2s = sqrtf(1.f - powf(a.beta2, t));
  const float step = lr / bc1;
  const float decay_w = 1.f - lr * a.wd;
  // EMA (utils.py:169-193): calls that see step <= update_after (+1: first 'initted' call) are straight copies
  float ema_decay = -1.f;
  if (t0 > a.ema_update_after + 1) {
    const float epoch = (float)(t0 - a.ema_update_after);
    const float d = 1.f - 1.f / (1.f + epoch);
    ema_decay = d < a.ema_beta ? d : a.ema_beta;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.count; i += (int64_t)gridDim.x * 256) {
    const float g = a.g[i] * clip;
    float p = a.p[i] * decay_w;
    float m = a.m[i];
    m = m + (g - m) * (1.f - a.beta1);  // lerp, as torch.optim's exp_avg.lerp_
    const float v = a.v[i] * a.beta2 + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / bc2s + a.eps;
    p = p - step * (m / denom);
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
    if (a.ema) {
      if (ema_decay < 0.f) a.ema[i] = p;
      else { const float e = a.ema[i]; a.ema[i] = e - (e - p) * (1.f - ema_decay); }
    }
  }
}

__global__ void step_commit_kernel(float* state) {
  if (state[3] == 0.f) state[5] += 1.f;
}

}  // namespace cgen

using namespace cgen;

extern "C" int cgen_sumsq_partial(const float* g, int64_t count, float* partial, int32_t nblk, cgen_stream_t stream) {
  CGEN_REQUIRE(g && partial && nblk > 0 && count >= 0, "cgen_sumsq_partial: bad args");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, count, partial);
  return check_launch("cgen_sumsq_partial");
}

extern "C" int cgen_clip_decide(const float* partial, int32_t nblk, const float* out3, float max_norm, float skip_norm,
                                float* state_dev, cgen_stream_t stream) {
  CGEN_REQUIRE(partial && state_dev && nblk > 0, "cgen_clip_decide: bad args");
  hipLaunchKernelGGL(clip_decide_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, out3, max_norm, skip_norm, state_dev);
  return check_launch("cgen_clip_decide");
}

extern "C" int cgen_adamw_ema(const cgen_adamw_args* a, cgen_stream_t stream) {
  CGEN_REQUIRE(a && a->p && a->g && a->m && a->v && a->state_dev && a->count >= 0, "cgen_adamw_ema: bad args");
  if (a->count == 0) return CGEN_OK;
  AdamP p;
  p.p = a->p; p.g = a->g; p.m = a->m; p.v = a->v; p.ema = a->ema; p.count = a->count;
  p.lr = a->lr; p.beta1 = a->beta1; p.beta2 = a->beta2; p.eps = a->eps; p.wd = a->wd; p.ema_beta = a->ema_beta;
  p.warmup_steps = a->warmup_steps; p.ema_update_after = a->ema_update_after; p.state = a->state_dev;
  int64_t b = (a->count + 255) / 256;
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("cgen_adamw_ema");
}

extern "C" int cgen_step_commit(float* state_dev, cgen_stream_t stream) {
  CGEN_REQUIRE(state_dev, "cgen_step_commit: null");
  hipLaunchKernelGGL(step_commit_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state_dev);
  return check_launch("cgen_step_commit");
}
