// Multi-level cross-entropy, forward + backward fused (network/loss.py:29-38 MultiLevelCELoss + autograd):
//   loss = mean_b sum_l w_l * c[s_l + y_bl] * CE(logits[b, s_l:e_l], y_bl = level_labels[b, l]),
// c = the optional per-class weights (loss.py:16-25: nn.CrossEntropyLoss(weight=weight[s_l:e_l], reduction='none'), i.e. the sample's
// term is scaled by its target class's weight and nothing is renormalised); c = NULL: all ones.
// One wave per (sample b, level l): max -> sum-exp -> loss term -> gradient slice, butterflies for the reductions.
// HBM-bound: reads B*C*4 (twice, second pass from L1/L2), writes B*C*4.
#include "lec_common.h"

namespace lec {

struct LevelTable { int start[16]; int size[16]; float weight[16]; int L; };

__global__ __launch_bounds__(256) void mlce_kernel(const float* __restrict__ logits, int64_t ld,
                                                   const int64_t* __restrict__ labels, int B, LevelTable lt,
                                                   const float* __restrict__ cw, float* __restrict__ glogits, float* partials,
                                                   unsigned int* counter, float* loss) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
  const float invB = 1.0f / (float)B;
  float lsum = 0.0f;
  for (int64_t task = wave; task < (int64_t)B * lt.L; task += nwave) {
    const int b = (int)(task / lt.L), l = (int)(task - (int64_t)b * lt.L);
    const float* z = logits + (int64_t)b * ld + lt.start[l];
    const int n = lt.size[l];
    const int lab = (int)labels[(int64_t)b * lt.L + l];
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, z[i]);
    mx = group_max<64>(mx);
    float se = 0.0f;
    for (int i = lane; i < n; i += 64) se += expf(z[i] - mx);
    se = group_sum<64>(se);
    const float lse = logf(se);
    const float w = cw ? lt.weight[l] * cw[lt.start[l] + lab] : lt.weight[l];
    if (lane == 0) lsum += w * (lse - (z[lab] - mx));
    if (glogits) {
      float* g = glogits + (int64_t)b * ld + lt.start[l];
      const float sc = w * invB;
      for (int i = lane; i < n; i += 64) {
        float p = expf(z[i] - mx - lse);
        g[i] = sc * (p - (i == lab ? 1.0f : 0.0f));
      }
    }
  }
  lsum = group_sum<64>(lsum);
  block_publish_and_finalize(lsum, partials, counter, loss, invB);
}

}  // namespace lec

extern "C" int lec_multilevel_ce_fwd_bwd(const float* logits, int64_t ld, const int64_t* level_labels, int B, int C,
                                         const int32_t* level_sizes, const float* level_weights, const float* class_weights, int L, float* loss,
                                         float* glogits, void* workspace, int64_t workspace_bytes,
                                         lec_stream_t stream) {
  using namespace lec;
  LEC_CHECK_ARG(logits && level_labels && level_sizes && loss, "multilevel_ce: null pointer");
  LEC_CHECK_ARG(B > 0 && C > 0 && ld >= C && L > 0 && L <= 16, "multilevel_ce: bad sizes B=%d C=%d L=%d", B, C, L);
  LevelTable lt; lt.L = L; int s = 0;
  for (int l = 0; l < L; ++l) {
    LEC_CHECK_ARG(level_sizes[l] > 0, "multilevel_ce: empty level %d", l);
    lt.start[l] = s; lt.size[l] = level_sizes[l]; lt.weight[l] = level_weights ? level_weights[l] : 1.0f; s += level_sizes[l];
  }
  LEC_CHECK_ARG(s == C, "multilevel_ce: level sizes sum to %d, expected C=%d", s, C);
  int64_t tasks = (int64_t)B * L;
  int nblocks = (int)((tasks + 3) / 4 > 2048 ? 2048 : (tasks + 3) / 4);
  const int64_t need = 256 + (int64_t)nblocks * sizeof(float);
  LEC_CHECK_ARG(workspace && workspace_bytes >= need, "multilevel_ce: workspace needs %lld bytes", (long long)need);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mlce_kernel, dim3(nblocks), dim3(256), 0, st, logits, ld, level_labels, B, lt, class_weights, glogits,
                     (float*)((char*)workspace + 256), (unsigned int*)workspace, loss);
  LEC_CHECK_LAUNCH("mlce_kernel");
  return LEC_OK;
}
