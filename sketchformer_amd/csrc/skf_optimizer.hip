// Fused Keras-Adam sweep over the flat parameter buffer + the per-step scalar
// prologue (WarmupDecay learning rate, Adam bias correction, dropout key).
//
// Replaces tf.keras.optimizers.Adam(lr_schedule, beta_1=0.9, beta_2=0.98, epsilon=1e-9)
// .apply_gradients (models/sketchformer.py:122-124,348) and builders/schedulers.py:13-46.
// Keras semantics kept: the schedule is evaluated on `iterations` BEFORE the
// increment (first update has lr = 0); t = iterations + 1;
// alpha = lr * sqrt(1 - b2^t) / (1 - b1^t); m += (g - m)(1 - b1); v += (g*g - v)(1 - b2);
// w -= alpha * m / (sqrt(v) + eps)   (epsilon outside the sqrt, not bias corrected).
//
// Everything step dependent lives in device memory (SkfStepState) so the whole
// train step can be replayed from one hipGraph.
#include "skf_common.h"

namespace {

// schedule: 0 = WarmupDecay(d_model, warmup), 1 = StepDecay(init_lr, rate, steps, min_ratio), 2 = constant
__global__ void step_prologue_kernel(SkfStepState* st, int schedule, float p0, float p1, float p2, float p3,
                                     float beta1, float beta2, uint32_t seed) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long it = st->iterations;
  const float step = (float)it;
  float lr;
  if (schedule == 0) {
    // rsqrt(d_model) * min(rsqrt(step), step * warmup^-1.5); step 0 -> min(inf, 0) = 0
    const float arg1 = 1.0f / sqrtf(step);
    const float arg2 = step * p1;          // p1 = (float)(warmup ** -1.5), computed in double on the host
    lr = (1.0f / sqrtf(p0)) * fminf(arg1, arg2);
  } else if (schedule == 1) {
    lr = fmaxf(p0 * powf(p1, floorf(step / p2)), p0 * p3);
  } else {
    lr = p0;
  }
  const double t = (double)(it + 1);
  const double b1p = pow((double)beta1, t), b2p = pow((double)beta2, t);
  st->lr = lr;
  st->alpha = lr * (float)(sqrt(1.0 - b2p) / (1.0 - b1p));
  st->drop_key = skf_hash32(seed ^ skf_hash32((uint32_t)it + 0x632be5abU));
}

__global__ void step_epilogue_kernel(SkfStepState* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) st->iterations += 1;
}

// ADVANCE: the sweep also closes the step (optimizer.iterations += 1, what step_epilogue_kernel does as a launch of its own):
// one thread adds to `iterations`, a field no thread of this kernel reads (they read alpha)
template <bool ADVANCE>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   SkfStepState* st, float grad_scale,
                                                   float one_minus_b1, float one_minus_b2, float eps) {
  const float alpha = st->alpha;
  if (ADVANCE && blockIdx.x == 0 && threadIdx.x == 0) st->iterations += 1;
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 wv = reinterpret_cast<float4*>(w)[i];
    float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define SKF_ADAM1(c)                                   \
    {                                                  \
      const float gg = gv.c * grad_scale;              \
      mv.c += (gg - mv.c) * one_minus_b1;              \
      vv.c += (gg * gg - vv.c) * one_minus_b2;         \
      wv.c -= alpha * mv.c / (sqrtf(vv.c) + eps);      \
    }
    SKF_ADAM1(x) SKF_ADAM1(y) SKF_ADAM1(z) SKF_ADAM1(w)
#undef SKF_ADAM1
    reinterpret_cast<float4*>(w)[i] = wv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gg = g[i] * grad_scale;
    float mm = m[i], vv = v[i];
    mm += (gg - mm) * one_minus_b1;
    vv += (gg * gg - vv) * one_minus_b2;
    m[i] = mm; v[i] = vv;
    w[i] -= alpha * mm / (sqrtf(vv) + eps);
  }
}

// tf.keras.optimizers.SGD(lr_schedule, momentum=0.9), nesterov=False (models/sketchformer.py:124-126):
//   velocity = momentum * velocity - lr * g ;  w += velocity       (lr from the schedule, pre-increment step)
__global__ __launch_bounds__(256) void sgd_momentum_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                           float* __restrict__ vel, size_t n,
                                                           const SkfStepState* __restrict__ st, float grad_scale,
                                                           float momentum) {
  const float lr = st->lr;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float v = momentum * vel[i] - lr * (g[i] * grad_scale);
    vel[i] = v;
    w[i] += v;
  }
}

}  // namespace

extern "C" int skf_sgd_momentum_step(float* w, const float* g, float* velocity, size_t n, const void* step_state,
                                     float grad_scale, float momentum, skf_stream_t stream) {
  SKF_CHECK_ARG(w && g && velocity && step_state, "null operand");
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  SkfProfScope ps((hipStream_t)stream, "sgd_momentum", 0.0, 20.0 * n);
  hipLaunchKernelGGL(sgd_momentum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, g, velocity, n,
                     (const SkfStepState*)step_state, grad_scale, momentum);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" size_t skf_step_state_bytes(void) { return sizeof(SkfStepState); }

extern "C" int skf_step_prologue(void* step_state, int schedule, float p0, float p1, float p2, float p3, float beta1,
                                 float beta2, unsigned seed, skf_stream_t stream) {
  SKF_CHECK_ARG(step_state, "null step state");
  SKF_CHECK_ARG(schedule >= 0 && schedule <= 2, "bad schedule");
  hipLaunchKernelGGL(step_prologue_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (SkfStepState*)step_state, schedule,
                     p0, p1, p2, p3, beta1, beta2, seed);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_step_epilogue(void* step_state, skf_stream_t stream) {
  SKF_CHECK_ARG(step_state, "null step state");
  hipLaunchKernelGGL(step_epilogue_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (SkfStepState*)step_state);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

// internal (skf_model.hip): advance != 0 = the sweep of a whole step, which also performs skf_step_epilogue's increment
int skf_adam_step_launch(float* w, const float* g, float* m, float* v, size_t n, void* step_state, float grad_scale, float beta1, float beta2,
                         float eps, int advance, hipStream_t stream) {
  SKF_CHECK_ARG(w && g && m && v && step_state, "null operand");
  SKF_CHECK_ARG((((uintptr_t)w | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "buffers must be 16-byte aligned");
  size_t blocks = ((n >> 2) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  SkfProfScope ps(stream, "adam", 0.0, 28.0 * n);
  if (advance)
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, w, g, m, v, n, (SkfStepState*)step_state, grad_scale,
                       1.0f - beta1, 1.0f - beta2, eps);
  else
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, w, g, m, v, n, (SkfStepState*)step_state, grad_scale,
                       1.0f - beta1, 1.0f - beta2, eps);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
extern "C" int skf_adam_step(float* w, const float* g, float* m, float* v, size_t n, const void* step_state,
                             float grad_scale, float beta1, float beta2, float eps, skf_stream_t stream) {
  return skf_adam_step_launch(w, g, m, v, n, const_cast<void*>(step_state), grad_scale, beta1, beta2, eps, 0, (hipStream_t)stream);
}
