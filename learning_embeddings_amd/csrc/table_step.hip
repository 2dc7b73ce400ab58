// Label-table maintenance and stand-alone projections.
//   lec_table_step_adam : oe_h.py:1768-1771  grad *= ((1-|w|)/2)^2 -> Adam -> clip rows into [r_in, 1-1e-5]
//   lec_table_step_rsgd : oe_h.py:1761-1762, :1619-1644  exp_map_x / Mobius addition, then the same clip
//   lec_adam_flat       : torch.optim.Adam over one flat fp32 arena (the CNN's parameters)
//   lec_label_project_* : Embedder.forward (oe_h.py:77-104) and its autograd (dense table gradient)
//   lec_image_softclip_*: FeatCNN18.soft_clip (oe_h.py:323-328) and its autograd
// All are one pass over [rows, D]; T lanes per row, xor-butterfly row norms.  HBM-bound: the Adam step moves
// 7*N*D*4 bytes (read w, g, m, v; write w, m, v).
#include <hip/hip_bf16.h>
#include "lec_common.h"

namespace lec {

static int pick_T(int D) { return D <= 4 ? 1 : (D <= 16 ? 4 : (D <= 64 ? 16 : 64)); }

struct AdamConsts { float lr_step, bc2_sqrt, beta1, beta2, one_m_beta1, one_m_beta2, eps; };

static AdamConsts adam_consts(float lr, float beta1, float beta2, float eps, int step) {
  AdamConsts c;
  double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  c.lr_step = (float)((double)lr / bc1); c.bc2_sqrt = (float)sqrt(bc2);
  c.beta1 = beta1; c.beta2 = beta2; c.one_m_beta1 = (float)(1.0 - (double)beta1); c.one_m_beta2 = (float)(1.0 - (double)beta2);
  c.eps = eps;
  return c;
}

__device__ __forceinline__ void adam_elem(float& w, float g, float& m, float& v, const AdamConsts& c) {
  m = m + (g - m) * c.one_m_beta1;                                   // exp_avg.lerp_(grad, 1-beta1)
  v = v * c.beta2 + c.one_m_beta2 * g * g;                           // mul_(beta2).addcmul_(g, g, 1-beta2)
  float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  w = w + (-c.lr_step) * (m / denom);                                // addcdiv_(exp_avg, denom, -step_size)
}

template <int T>
__global__ __launch_bounds__(256) void table_adam_kernel(float* __restrict__ W, const float* __restrict__ G,
                                                         float* __restrict__ Mo, float* __restrict__ Vo, int64_t ld,
                                                         int N, int D, AdamConsts c, float r_in, int riemannian,
                                                         int clip, _Float16* __restrict__ Wh) {
  constexpr int RPW = kWave / T;
  const int lane = threadIdx.x & 63, t = lane % T, slot = lane / T;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwave = gridDim.x * (blockDim.x >> 6);
  for (int base = wave * RPW; base < N; base += nwave * RPW) {
    const int row = base + slot;
    const bool valid = row < N;
    const int64_t off = (int64_t)(valid ? row : 0) * ld;
    float nn = 0.0f;
    for (int d = t; d < D; d += T) { float w = W[off + d]; nn += w * w; }
    nn = group_sum<T>(nn);
    float scale = 1.0f;
    if (riemannian) {                                                // (1/lambda_x)^2, lambda_x = 2/(1-|w|)  (oe_h.py:1636,1768)
      float lam = 2.0f / (1.0f - sqrtf(nn));
      float inv = 1.0f / lam;
      scale = inv * inv;
    }
    float n2 = 0.0f;
    if (valid) {
      for (int d = t; d < D; d += T) {
        float w = W[off + d], g = G[off + d] * scale, m = Mo[off + d], v = Vo[off + d];
        adam_elem(w, g, m, v, c);
        W[off + d] = w; Mo[off + d] = m; Vo[off + d] = v;
        n2 += w * w;
      }
    }
    n2 = group_sum<T>(n2);
    if (clip && valid) {                                             // oe_h.py:1611-1616
      float no = sqrtf(n2);
      if (no <= r_in) { for (int d = t; d < D; d += T) W[off + d] = W[off + d] / no * r_in; }
      else if (no >= 1.0f) { for (int d = t; d < D; d += T) W[off + d] = W[off + d] / no * (float)(1.0 - 1e-5); }
    }
    if (Wh && valid) {                                               // config 5: refresh the fp16 shadow the loss kernel reads
      for (int d = t; d < D; d += T) Wh[off + d] = (_Float16)W[off + d];
    }
  }
}

template <int T>
__global__ __launch_bounds__(256) void table_rsgd_kernel(float* __restrict__ W, const float* __restrict__ G, int64_t ld,
                                                         int N, int D, float lr, float r_in) {
  constexpr int RPW = kWave / T;
  const int lane = threadIdx.x & 63, t = lane % T, slot = lane / T;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwave = gridDim.x * (blockDim.x >> 6);
  for (int base = wave * RPW; base < N; base += nwave * RPW) {
    const int row = base + slot;
    const bool valid = row < N;
    const int64_t off = (int64_t)(valid ? row : 0) * ld;
    float xx = 0.0f;
    for (int d = t; d < D; d += T) { float w = W[off + d]; xx += w * w; }
    xx = group_sum<T>(xx);
    const float xn = sqrtf(xx);
    const float lam = 2.0f / (1.0f - xn);                            // oe_h.py:1636
    const float inv = 1.0f / lam, scale = inv * inv;                 // :1761
    float vv = 0.0f;
    for (int d = t; d < D; d += T) { float v = -lr * (G[off + d] * scale) + 1e-15f; vv += v * v; }   // :1639
    vv = group_sum<T>(vv);
    const float nv = sqrtf(vv);
    float arg = lam * nv / 2.0f;
    arg = arg < -15.0f ? -15.0f : (arg > 15.0f ? 15.0f : arg);
    const float th = tanhf(arg);                                     // :1641
    float dot = 0.0f, tt = 0.0f;
    for (int d = t; d < D; d += T) {
      float v = -lr * (G[off + d] * scale) + 1e-15f;
      float sec = th * v / nv + 1e-6f;                               // second_term, then mob_add's v + 1e-6 (:1620)
      dot += W[off + d] * sec; tt += sec * sec;
    }
    dot = 2.0f * group_sum<T>(dot); tt = group_sum<T>(tt);
    const float den = 1.0f + dot + tt * xx;                          // :1624
    const float cu = (1.0f + dot + tt) / den, cv = (1.0f - xx) / den;
    float n2 = 0.0f;
    if (valid) {
      for (int d = t; d < D; d += T) {
        float v = -lr * (G[off + d] * scale) + 1e-15f;
        float sec = th * v / nv + 1e-6f;
        float r = cu * W[off + d] + cv * sec;                        // :1629
        W[off + d] = r; n2 += r * r;
      }
    }
    n2 = group_sum<T>(n2);
    if (valid) {
      float no = sqrtf(n2);
      if (no <= r_in) { for (int d = t; d < D; d += T) W[off + d] = W[off + d] / no * r_in; }
      else if (no >= 1.0f) { for (int d = t; d < D; d += T) W[off + d] = W[off + d] / no * (float)(1.0 - 1e-5); }
    }
  }
}

// `lp` (optional): a bf16 shadow of the updated parameters, written in the same pass -- the conv layers read it directly,
// so no per-layer fp32 -> bf16 cast kernels run in the forward pass.
__device__ __forceinline__ unsigned short f2bf_rn(float f) {
  __hip_bfloat16 h = __float2bfloat16(f);
  return *reinterpret_cast<unsigned short*>(&h);
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        AdamConsts c, float grad_scale, unsigned short* __restrict__ lp) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float4* p4 = (float4*)p; const float4* g4 = (const float4*)g; float4* m4 = (float4*)m; float4* v4 = (float4*)v;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 P = p4[i], G = g4[i], M = m4[i], V = v4[i];
    adam_elem(P.x, G.x * grad_scale, M.x, V.x, c); adam_elem(P.y, G.y * grad_scale, M.y, V.y, c);
    adam_elem(P.z, G.z * grad_scale, M.z, V.z, c); adam_elem(P.w, G.w * grad_scale, M.w, V.w, c);
    p4[i] = P; m4[i] = M; v4[i] = V;
    if (lp) {
      ushort4 o; o.x = f2bf_rn(P.x); o.y = f2bf_rn(P.y); o.z = f2bf_rn(P.z); o.w = f2bf_rn(P.w);
      ((ushort4*)lp)[i] = o;
    }
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float P = p[i], M = m[i], V = v[i];
    adam_elem(P, g[i] * grad_scale, M, V, c);
    p[i] = P; m[i] = M; v[i] = V;
    if (lp) lp[i] = f2bf_rn(P);
  }
}

// ---- stand-alone projections --------------------------------------------------------------------------------------
// MODE 0: label (Embedder.forward of oe_h.py), rows gathered by idx;  MODE 1: soft_clip x/|x|(|x| + add), rows dense;
// MODE 2: the same soft_clip on rows gathered by idx (Embedder.forward of oe.py).  `r_in` carries `add` for 1 and 2.
template <int T, int MODE, bool BWD>
__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ src, int64_t ld_src,
                                                      const int64_t* __restrict__ idx, int64_t n, int D, float r_in,
                                                      float r_in_h, float* __restrict__ out, int64_t ld_out,
                                                      const float* __restrict__ gout, int64_t ld_gout,
                                                      float* __restrict__ gdst, int64_t ld_gdst) {
  constexpr int RPW = kWave / T;
  const int lane = threadIdx.x & 63, t = lane % T, slot = lane / T;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t base = wave * RPW; base < n; base += nwave * RPW) {
    const int64_t i = base + slot;
    const bool valid = i < n;
    const int64_t srow = valid ? (MODE != 1 ? idx[i] : i) : 0;
    const float* e = src + srow * ld_src;
    const float add = MODE == 0 ? 1e-15f : 0.0f;
    float nn = 0.0f;
    for (int d = t; d < D; d += T) { float v = e[d] + add; nn += v * v; }
    nn = group_sum<T>(nn);
    const float nrm = sqrtf(nn), den = fmaxf(nrm, 1e-12f), denp = nrm >= 1e-12f ? 1.0f : 0.0f;
    float mul, A, Bc;
    if (MODE == 0) {
      float arg = r_in_h + nrm;
      float argc = arg < -15.0f ? -15.0f : (arg > 15.0f ? 15.0f : arg);
      float th = tanhf(argc);
      float tp = (arg >= -15.0f && arg <= 15.0f) ? 1.0f - th * th : 0.0f;
      mul = th; A = th / den; Bc = nrm > 0.0f ? (tp / den - th * denp / (den * den)) / nrm : 0.0f;
    } else {
      float sc = nrm + r_in;
      mul = sc; A = sc / den; Bc = nrm > 0.0f ? (1.0f / den - sc * denp / (den * den)) / nrm : 0.0f;
    }
    if (!BWD) {
      float pp = 0.0f;
      for (int d = t; d < D; d += T) { float p = mul * ((e[d] + add) / den); pp += p * p; }
      pp = group_sum<T>(pp);
      float post_div = 1.0f, post_mul = 1.0f; bool clipped = false;
      if (MODE == 0) {
        float no = sqrtf(pp);
        if (no <= r_in) { clipped = true; post_div = no; post_mul = r_in; }
        else if (no >= 1.0f) { clipped = true; post_div = no; post_mul = (float)(1.0 - 1e-5); }
      }
      if (valid) {
        for (int d = t; d < D; d += T) {
          float p = mul * ((e[d] + add) / den);
          out[i * ld_out + d] = clipped ? p / post_div * post_mul : p;
        }
      }
    } else {
      float dot = 0.0f;
      if (valid) for (int d = t; d < D; d += T) dot += (e[d] + add) * gout[i * ld_gout + d];
      dot = group_sum<T>(dot);
      if (valid) {
        const float c = Bc * dot;
        for (int d = t; d < D; d += T) {
          float gval = A * gout[i * ld_gout + d] + c * (e[d] + add);
          if (MODE != 1) atomicAdd(gdst + srow * ld_gdst + d, gval);      // dense table gradient, duplicates add up
          else gdst[i * ld_gdst + d] = gval;
        }
      }
    }
  }
}

template <int MODE, bool BWD>
static int launch_project(const float* src, int64_t ld_src, const int64_t* idx, int64_t n, int D, float K,
                          float* out, int64_t ld_out, const float* gout, int64_t ld_gout, float* gdst, int64_t ld_gdst,
                          hipStream_t st, bool add_is_K = false) {
  const int T = pick_T(D);
  int64_t waves = (n + (64 / T) - 1) / (64 / T);
  int nblocks = (int)((waves + 3) / 4 > 4096 ? 4096 : (waves + 3) / 4);
  const float r_in = add_is_K ? K : inner_radius_f(K), r_in_h = inner_radius_h_f(K);
#define L(T_) hipLaunchKernelGGL((project_kernel<T_, MODE, BWD>), dim3(nblocks), dim3(256), 0, st, src, ld_src, idx, n, D, r_in, r_in_h, out, ld_out, gout, ld_gout, gdst, ld_gdst)
  if (T == 1) L(1); else if (T == 4) L(4); else if (T == 16) L(16); else L(64);
#undef L
  LEC_CHECK_LAUNCH("project_kernel");
  return LEC_OK;
}

}  // namespace lec

static int table_step_adam_impl(float* table, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t ld,
                                   int n_labels, int D, float lr, float beta1, float beta2, float eps, int step,
                                   float K_cone, int riemannian, int clip, void* table_f16, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(table && grad && exp_avg && exp_avg_sq, "table_step_adam: null pointer");
  LEC_CHECK_ARG(n_labels > 0 && D > 0 && ld >= D && step >= 1, "table_step_adam: bad sizes N=%d D=%d step=%d", n_labels, D, step);
  const int T = pick_T(D);
  int waves = (n_labels + (64 / T) - 1) / (64 / T);
  int nblocks = (waves + 3) / 4 > 2048 ? 2048 : (waves + 3) / 4;
  AdamConsts c = adam_consts(lr, beta1, beta2, eps, step);
  const float r_in = inner_radius_f(K_cone);
  hipStream_t st = (hipStream_t)stream;
#define L(T_) hipLaunchKernelGGL((table_adam_kernel<T_>), dim3(nblocks), dim3(256), 0, st, table, grad, exp_avg, exp_avg_sq, ld, n_labels, D, c, r_in, riemannian, clip, (_Float16*)table_f16)
  if (T == 1) L(1); else if (T == 4) L(4); else if (T == 16) L(16); else L(64);
#undef L
  LEC_CHECK_LAUNCH("table_adam_kernel");
  return LEC_OK;
}

extern "C" int lec_table_step_adam(float* table, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t ld, int n_labels, int D, float lr, float beta1, float beta2, float eps, int step, float K_cone, int riemannian, int clip, lec_stream_t stream) {
  return table_step_adam_impl(table, grad, exp_avg, exp_avg_sq, ld, n_labels, D, lr, beta1, beta2, eps, step, K_cone, riemannian, clip, nullptr, stream);
}

// The same step with the fp16 shadow of the table (BASELINE.json config 5, "fp16+fp32-master") refreshed in the same pass:
// table_f16 [n_labels, D] (same ld, in elements) receives the updated, clipped rows rounded to fp16.
extern "C" int lec_table_step_adam_f16(float* table, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t ld, int n_labels, int D, float lr, float beta1, float beta2, float eps, int step, float K_cone, int riemannian, int clip, void* table_f16, lec_stream_t stream) {
  LEC_CHECK_ARG(table_f16, "table_step_adam_f16: null shadow");
  return table_step_adam_impl(table, grad, exp_avg, exp_avg_sq, ld, n_labels, D, lr, beta1, beta2, eps, step, K_cone, riemannian, clip, table_f16, stream);
}

extern "C" int lec_table_step_rsgd(float* table, const float* grad, int64_t ld, int n_labels, int D, float lr,
                                   float K_cone, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(table && grad, "table_step_rsgd: null pointer");
  LEC_CHECK_ARG(n_labels > 0 && D > 0 && ld >= D, "table_step_rsgd: bad sizes");
  const int T = pick_T(D);
  int waves = (n_labels + (64 / T) - 1) / (64 / T);
  int nblocks = (waves + 3) / 4 > 2048 ? 2048 : (waves + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
#define L(T_) hipLaunchKernelGGL((table_rsgd_kernel<T_>), dim3(nblocks), dim3(256), 0, st, table, grad, ld, n_labels, D, lr, inner_radius_f(K_cone))
  if (T == 1) L(1); else if (T == 4) L(4); else if (T == 16) L(16); else L(64);
#undef L
  LEC_CHECK_LAUNCH("table_rsgd_kernel");
  return LEC_OK;
}

extern "C" int lec_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, int step, float grad_scale, void* param_bf16,
                             lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(n >= 0 && step >= 1, "adam_flat: bad n/step");
  if (n == 0) return LEC_OK;
  LEC_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "adam_flat: null pointer");
  LEC_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "adam_flat: buffers must be 16-byte aligned");
  AdamConsts c = adam_consts(lr, beta1, beta2, eps, step);
  int64_t blocks = ((n >> 2) + 255) / 256;
  int nblocks = (int)(blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks));
  hipLaunchKernelGGL(adam_flat_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, c, grad_scale, (unsigned short*)param_bf16);
  LEC_CHECK_LAUNCH("adam_flat_kernel");
  return LEC_OK;
}

extern "C" int lec_label_project_fwd(int label_proj, const float* table, int64_t ld_table, int n_labels,
                                     const int64_t* idx, int64_t n, int D, float K_cone, float* out, int64_t ld_out,
                                     lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(label_proj == LEC_LABEL_HYP || label_proj == LEC_LABEL_SOFTCLIP_K, "label_project_fwd: unknown label_proj %d", label_proj);
  LEC_CHECK_ARG(n >= 0 && D > 0 && ld_table >= D && ld_out >= D && n_labels > 0, "label_project_fwd: bad sizes");
  if (n == 0) return LEC_OK;
  LEC_CHECK_ARG(table && idx && out, "label_project_fwd: null pointer");
  if (label_proj == LEC_LABEL_SOFTCLIP_K)
    return launch_project<2, false>(table, ld_table, idx, n, D, K_cone, out, ld_out, nullptr, 0, nullptr, 0, (hipStream_t)stream, true);
  return launch_project<0, false>(table, ld_table, idx, n, D, K_cone, out, ld_out, nullptr, 0, nullptr, 0, (hipStream_t)stream);
}

extern "C" int lec_label_project_bwd(int label_proj, const float* table, int64_t ld_table, int n_labels,
                                     const int64_t* idx, int64_t n, int D, float K_cone, const float* gout,
                                     int64_t ld_gout, float* grad_table, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(label_proj == LEC_LABEL_HYP || label_proj == LEC_LABEL_SOFTCLIP_K, "label_project_bwd: unknown label_proj %d", label_proj);
  LEC_CHECK_ARG(n >= 0 && D > 0 && ld_table >= D && ld_gout >= D && n_labels > 0, "label_project_bwd: bad sizes");
  if (n == 0) return LEC_OK;
  LEC_CHECK_ARG(table && idx && gout && grad_table, "label_project_bwd: null pointer");
  if (label_proj == LEC_LABEL_SOFTCLIP_K)
    return launch_project<2, true>(table, ld_table, idx, n, D, K_cone, nullptr, 0, gout, ld_gout, grad_table, ld_table, (hipStream_t)stream, true);
  return launch_project<0, true>(table, ld_table, idx, n, D, K_cone, nullptr, 0, gout, ld_gout, grad_table, ld_table, (hipStream_t)stream);
}

extern "C" int lec_image_softclip_fwd(int image_proj, const float* raw, int64_t ld_raw, int64_t n, int D, float K_cone,
                                      float* out, int64_t ld_out, lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(image_proj == LEC_IMAGE_SOFTCLIP || image_proj == LEC_IMAGE_SOFTCLIP_K, "image_softclip_fwd: unknown image_proj %d", image_proj);
  LEC_CHECK_ARG(n >= 0 && D > 0 && ld_raw >= D && ld_out >= D, "image_softclip_fwd: bad sizes");
  if (n == 0) return LEC_OK;
  LEC_CHECK_ARG(raw && out, "image_softclip_fwd: null pointer");
  return launch_project<1, false>(raw, ld_raw, nullptr, n, D, K_cone, out, ld_out, nullptr, 0, nullptr, 0, (hipStream_t)stream,
                                  image_proj == LEC_IMAGE_SOFTCLIP_K);
}

extern "C" int lec_image_softclip_bwd(int image_proj, const float* raw, int64_t ld_raw, const float* gout,
                                      int64_t ld_gout, int64_t n, int D, float K_cone, float* graw, int64_t ld_graw,
                                      lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(image_proj == LEC_IMAGE_SOFTCLIP || image_proj == LEC_IMAGE_SOFTCLIP_K, "image_softclip_bwd: unknown image_proj %d", image_proj);
  LEC_CHECK_ARG(n >= 0 && D > 0 && ld_raw >= D && ld_gout >= D && ld_graw >= D, "image_softclip_bwd: bad sizes");
  if (n == 0) return LEC_OK;
  LEC_CHECK_ARG(raw && gout && graw, "image_softclip_bwd: null pointer");
  return launch_project<1, true>(raw, ld_raw, nullptr, n, D, K_cone, nullptr, 0, gout, ld_gout, graw, ld_graw, (hipStream_t)stream,
                                 image_proj == LEC_IMAGE_SOFTCLIP_K);
}
