// Error plumbing + version of the C ABI (include/lecone.h).
#include <cstdarg>
#include <cstdio>
#include <hip/hip_runtime_api.h>
#include <cstdlib>
#include "../../include/lecone.h"
#include "tuning.h"

namespace lec {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return LEC_E_HIP;
}

// The ONLY place liblecone.so reads the environment (see tuning.h): called once, by the first launch that asks.
static Tuning load_tuning() {
  auto env = [](const char* name) -> const char* { return getenv(name); };
  auto i = [&](const char* name, int dflt) { const char* e = env(name); return e ? atoi(e) : dflt; };
  auto pos = [&](const char* name, int dflt) { const int v = i(name, dflt); return v > 0 ? v : dflt; };
  Tuning t;
  t.bn_apply_blocks = pos("LEC_BN_BLOCKS", 1536);
  t.cf_sk = i("LEC_CF_SK", 1);
  { const char* e = env("LEC_CF_SK_FILL"); t.cf_sk_fill = e ? atof(e) : 0.92; }
  t.cf_sk_min_chunks = i("LEC_CF_SK_MIN_CHUNKS", 8);
  t.cf_sk_wgs = pos("LEC_CF_SK_WGS", 512);
  t.cf_stem = i("LEC_CF_STEM", 1);
  t.cf_xcd = i("LEC_CF_XCD", 0);
  t.cf_lds_pad = i("LEC_CF_LDS_PAD", 0);
  t.dgrad_classes = i("LEC_DGRAD_CLASSES", 1);
  t.wg_dense_tile = i("LEC_WGRAD_DENSE_TILE", 1);
  t.wg_bm128 = i("LEC_WGRAD_BM128", 1);
  t.wg_shift = i("LEC_WGRAD_SHIFT", 1);
  t.wg_shift64 = i("LEC_WGRAD_SHIFT64", 1);
  t.wg_items = pos("LEC_WGRAD_ITEMS", 1024);
  t.wg_split_floor = i("LEC_WGRAD_SPLIT_FLOOR", 1);
  t.wg_lds_pad = i("LEC_WGRAD_LDS_PAD", 0);
  t.wg_wgs = pos("LEC_WGRAD_WGS", 1 << 30);
  t.wg_smask = i("LEC_WGRAD_SMASK", 0);
  t.bf_dma = i("LEC_BF16_DMA", 1);
  t.bf_tile = i("LEC_BF16_TILE", 1);
  t.bf_stem = i("LEC_BF16_STEM", 1);
  t.x3_wgs = pos("LEC_X3_WGS", 256);
  t.x3_force_narrow = env("LEC_X3_FORCE_NARROW") != nullptr;
  t.x3_chain = pos("LEC_X3_CHAIN", 512);
  t.c3_strip = env("LEC_C3_STRIP") != nullptr;
  t.jl_T = t.jl_EPL = t.jl_iters = 0;
  if (const char* e = env("LEC_JOINT_GEOM")) { if (sscanf(e, "%d,%d,%d", &t.jl_T, &t.jl_EPL, &t.jl_iters) < 2) t.jl_T = t.jl_EPL = t.jl_iters = 0; }
  t.jl_stage = i("LEC_JOINT_STAGE", 0);
  t.jl_fixed_point = i("LEC_JOINT_FIXED_POINT", 1);
  t.jl_wpb = i("LEC_JOINT_WPB", 0); if (t.jl_wpb < 0 || t.jl_wpb > 8) t.jl_wpb = 0;
  return t;
}
const Tuning& tuning() { static const Tuning t = load_tuning(); return t; }
}  // namespace lec

extern "C" const char* lec_last_error(void) { return lec::g_err; }
extern "C" int lec_abi_version(void) { return 33; }
