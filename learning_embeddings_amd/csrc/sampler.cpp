// Bit-exact negative sampler on the host, without the dense (N+M)^2 matrix.
//
// Replaces (reference file:line):
//   oe_h.py:799-809   set_negative_graph (stores dense bool A = 1 - TC - I over labels+images, built at :554-561)
//   oe_h.py:849-902   sample_negative_edge: candidates = np.where(A[u,:]) or np.where(A[:,v]) (ascending), optional
//                     level window, `random.choice` on CPython's MT19937
//   oe_h.py:940-957   the criterion's per-batch host loop (call order b, pass_ix, u-side then v-side)
//   order_embeddings.py:797-816  labels-only variant (level_id % L, no image slot)
//
// Representation: per node the SORTED list {node} U TC-neighbours (descendants for the row query, ancestors for the
// column query).  The candidate list of the reference is exactly "window [lo,hi) minus that sorted list", so
// len(candidates) = (hi-lo) - |list in window| and the r-th candidate is found with one binary search over the
// monotone map i -> list[i] - lo - i ("how many survivors precede the i-th excluded node").  O(log) per draw instead
// of O(N+M); memory O(|TC|) instead of O((N+M)^2).
//
// RNG: CPython's random.Random core restated from its published algorithm (Modules/_randommodule.c: init_by_array
// seeding from the 32-bit limbs of abs(seed), genrand_uint32; Lib/random.py: choice -> _randbelow_with_getrandbits:
// k = n.bit_length(), r = getrandbits(k) = top k bits of one 32-bit word, reject while r >= n).  Pinned by the
// known-answer tests in tests/ (first words of seed 0, choice(range(2000)) stream) and fixture F4.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>
#include "../../include/lecone.h"

namespace lec {
void set_error(const char* fmt, ...);

struct MT19937 {
  static constexpr int N = 624, M = 397;
  uint32_t mt[N]; int idx;
  void init_genrand(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = N;
  }
  void seed(uint64_t seed) {
    uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
    const int klen = key[1] ? 2 : 1;
    init_genrand(19650218u);
    int i = 1, j = 0;
    for (int k = (N > klen ? N : klen); k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      ++i; ++j;
      if (i >= N) { mt[0] = mt[N - 1]; i = 1; }
      if (j >= klen) j = 0;
    }
    for (int k = N - 1; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      ++i;
      if (i >= N) { mt[0] = mt[N - 1]; i = 1; }
    }
    mt[0] = 0x80000000u;
  }
  uint32_t u32() {
    if (idx >= N) {
      int k = 0;
      for (; k < N - M; ++k) { uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu); mt[k] = mt[k + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
      for (; k < N - 1; ++k) { uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu); mt[k] = mt[k + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
      uint32_t y = (mt[N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu); mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
  uint32_t randbelow(uint32_t n) {                      // n >= 1
    int k = 32 - __builtin_clz(n);                      // n.bit_length()
    uint32_t r = u32() >> (32 - k);
    while (r >= n) r = u32() >> (32 - k);
    return r;
  }
};
}  // namespace lec

struct lec_sampler {
  int L = 0;
  std::vector<int32_t> level_start, level_stop;
  int32_t n_labels = 0; int64_t n_images = 0, n_nodes = 0;
  std::vector<int64_t> desc_ptr, anc_ptr;
  std::vector<int32_t> desc, anc;                       // sorted, each list includes the node itself
  int pick_per_level = 0, mode = 0;
  std::vector<int32_t> hidden, visible;
  int64_t tc_edges = 0;
  lec::MT19937 rng;
};

extern "C" int lec_sampler_create(lec_sampler** out, const int32_t* level_sizes, int n_levels,
                                  const int32_t* label_edges, int64_t n_label_edges, const int64_t* image_ptr,
                                  const int32_t* image_adj, int64_t n_images, int pick_per_level, int mode,
                                  uint64_t seed) {
  using lec::set_error;
  if (!out || !level_sizes || n_levels <= 0 || n_label_edges < 0 || n_images < 0 || (n_label_edges && !label_edges) ||
      (n_images && (!image_ptr || !image_adj)) || (mode != 0 && mode != 1)) {
    set_error("sampler_create: bad arguments"); return LEC_E_ARG;
  }
  lec_sampler* s = new (std::nothrow) lec_sampler();
  if (!s) { set_error("sampler_create: out of memory"); return LEC_E_STATE; }
  s->L = n_levels;
  int64_t acc = 0;
  for (int l = 0; l < n_levels; ++l) {
    if (level_sizes[l] <= 0) { delete s; set_error("sampler_create: level %d is empty", l); return LEC_E_ARG; }
    s->level_start.push_back((int32_t)acc); acc += level_sizes[l]; s->level_stop.push_back((int32_t)acc);
  }
  if (acc + n_images >= (int64_t)1 << 31) { delete s; set_error("sampler_create: too many nodes"); return LEC_E_ARG; }
  s->n_labels = (int32_t)acc; s->n_images = n_images; s->n_nodes = acc + n_images;
  s->pick_per_level = pick_per_level ? 1 : 0; s->mode = mode;
  const int32_t N = s->n_labels;

  // label DAG: parents per label, then ancestors by memoised DFS (iterative; the DAG need not be level-adjacent)
  std::vector<std::vector<int32_t>> parents(N);
  for (int64_t e = 0; e < n_label_edges; ++e) {
    int32_t u = label_edges[2 * e], v = label_edges[2 * e + 1];
    if (u < 0 || u >= N || v < 0 || v >= N || u == v) { delete s; set_error("sampler_create: label edge %lld out of range", (long long)e); return LEC_E_ARG; }
    parents[v].push_back(u);
  }
  std::vector<std::vector<int32_t>> anc(N);
  std::vector<int8_t> state(N, 0);                     // 0 new, 1 on stack, 2 done
  std::vector<int32_t> stack;
  for (int32_t root = 0; root < N; ++root) {
    if (state[root]) continue;
    stack.push_back(root);
    while (!stack.empty()) {
      int32_t v = stack.back();
      if (state[v] == 0) {
        state[v] = 1;
        for (int32_t p : parents[v]) {
          if (state[p] == 1) { delete s; set_error("sampler_create: label graph has a cycle through %d", p); return LEC_E_ARG; }
          if (state[p] == 0) stack.push_back(p);
        }
      } else {
        stack.pop_back();
        if (state[v] == 2) continue;
        std::vector<int32_t>& a = anc[v];
        for (int32_t p : parents[v]) { a.push_back(p); a.insert(a.end(), anc[p].begin(), anc[p].end()); }
        std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end());
        state[v] = 2;
      }
    }
  }
  // ancestor lists (incl. self) for every node; descendant lists by inversion
  s->anc_ptr.assign(s->n_nodes + 1, 0);
  std::vector<int64_t> desc_cnt(s->n_nodes, 1);        // self
  std::vector<int32_t> tmp;
  auto image_anc = [&](int64_t j, std::vector<int32_t>& a) -> bool {
    a.clear();
    for (int64_t k = image_ptr[j]; k < image_ptr[j + 1]; ++k) {
      int32_t p = image_adj[k];
      if (p < 0 || p >= N) return false;
      a.push_back(p); a.insert(a.end(), anc[p].begin(), anc[p].end());
    }
    std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end());
    return true;
  };
  for (int32_t v = 0; v < N; ++v) { s->anc_ptr[v + 1] = s->anc_ptr[v] + (int64_t)anc[v].size() + 1; for (int32_t a : anc[v]) ++desc_cnt[a]; }
  for (int64_t j = 0; j < n_images; ++j) {
    if (!image_anc(j, tmp)) { delete s; set_error("sampler_create: image %lld has a parent out of range", (long long)j); return LEC_E_ARG; }
    s->anc_ptr[N + j + 1] = s->anc_ptr[N + j] + (int64_t)tmp.size() + 1;
    for (int32_t a : tmp) ++desc_cnt[a];
  }
  s->anc.resize(s->anc_ptr[s->n_nodes]);
  s->desc_ptr.assign(s->n_nodes + 1, 0);
  for (int64_t v = 0; v < s->n_nodes; ++v) s->desc_ptr[v + 1] = s->desc_ptr[v] + desc_cnt[v];
  s->desc.resize(s->desc_ptr[s->n_nodes]);
  std::vector<int64_t> fill(s->desc_ptr.begin(), s->desc_ptr.end() - 1);
  // visiting nodes in ascending order keeps every descendant list sorted; self goes in at its own turn
  for (int64_t v = 0; v < s->n_nodes; ++v) {
    const std::vector<int32_t>* a;
    if (v < N) a = &anc[v]; else { image_anc(v - N, tmp); a = &tmp; }
    int64_t o = s->anc_ptr[v];
    for (int32_t x : *a) { s->anc[o++] = x; s->desc[fill[x]++] = (int32_t)v; }   // ancestors < v ... not guaranteed: fixed below
    s->anc[o] = (int32_t)v;
    s->desc[fill[v]++] = (int32_t)v;
    s->tc_edges += (int64_t)a->size();
  }
  // general DAGs may number an ancestor above its descendant: sort every list to be safe (no-op for level-ordered ids)
  for (int64_t v = 0; v < s->n_nodes; ++v) {
    std::sort(s->anc.begin() + s->anc_ptr[v], s->anc.begin() + s->anc_ptr[v + 1]);
    std::sort(s->desc.begin() + s->desc_ptr[v], s->desc.begin() + s->desc_ptr[v + 1]);
  }
  s->rng.seed(seed);
  *out = s;
  return LEC_OK;
}

extern "C" void lec_sampler_destroy(lec_sampler* s) { delete s; }

extern "C" int lec_sampler_seed(lec_sampler* s, uint64_t seed) {
  if (!s) { lec::set_error("sampler_seed: null handle"); return LEC_E_STATE; }
  s->rng.seed(seed); return LEC_OK;
}

// oe_h.py:854 indexes `list(set(list(range(L+1))) - set(self.levels_to_hide))`: the ORDER of that list is CPython's set iteration
// order, which is ascending only while no two keys share a slot of the hash table.  With L + 1 > 8 slot ids and few of them left
// (the result set still has its initial 8-slot table) key 8 lands in slot 0: hidden {0,4,5,6,7} of 8 levels leaves [8, 1, 2, 3], not
// [1, 2, 3, 8] (fixture F4b).  Restated from CPython's Objects/setobject.c (3.7 - 3.12: same table policy): open addressing,
// hash(int) = int, LINEAR_PROBES = 9 following slots when they fit below the mask, then i = 5 i + 1 + (perturb >>= 5); growth to
// the first power of two above 4 x used once fill * 5 >= mask * 3; set_difference builds a NEW set by walking `so` in table order unless
// len(so) >> 2 > len(other), where it copies `so` and discards.  Pinned in tests against this interpreter's own sets.
namespace lec {
struct PySet {
  std::vector<int64_t> table; size_t mask = 7, fill = 0;
  PySet() : table(8, -1) {}
  static size_t find_free(const std::vector<int64_t>& t, size_t mask, int64_t key) {
    size_t perturb = (size_t)key, i = (size_t)key & mask;
    for (;;) {
      if (t[i] < 0) return i;
      if (i + 9 <= mask) for (size_t j = 1; j <= 9; ++j) if (t[i + j] < 0) return i + j;
      perturb >>= 5; i = (i * 5 + 1 + perturb) & mask;
    }
  }
  void resize(size_t minused) {
    size_t n = 8; while (n <= minused) n <<= 1;
    std::vector<int64_t> t(n, -1);
    for (int64_t k : table) if (k >= 0) t[find_free(t, n - 1, k)] = k;         // set_insert_clean in old table order
    table.swap(t); mask = n - 1;
  }
  void add(int64_t key) {                                                       // distinct keys only (set_add_entry without the compare)
    table[find_free(table, mask, key)] = key; ++fill;
    if (fill * 5 >= mask * 3) resize(fill > 50000 ? fill * 2 : fill * 4);
  }
};
}  // namespace lec

extern "C" int lec_sampler_set_levels_to_hide(lec_sampler* s, const int32_t* levels, int n) {
  if (!s || n < 0 || (n && !levels)) { lec::set_error("sampler_set_levels_to_hide: bad arguments"); return LEC_E_ARG; }
  s->hidden.assign(levels, levels + n);
  s->visible.clear();
  std::vector<int32_t> uniq(s->hidden); std::sort(uniq.begin(), uniq.end()); uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  auto is_hidden = [&](int64_t k) { return std::binary_search(uniq.begin(), uniq.end(), (int32_t)k); };
  lec::PySet so;
  for (int32_t l = 0; l <= s->L; ++l) so.add(l);                                // set(list(range(L + 1)))
  if ((so.fill >> 2) > uniq.size()) {                                           // set_copy_and_difference: a copy of `so`, the hidden ids discarded
    lec::PySet cp;
    if ((cp.fill + so.fill) * 5 >= cp.mask * 3) cp.resize(so.fill * 2);        // set_merge into an empty set
    if (cp.mask == so.mask) cp.table = so.table;
    else for (int64_t k : so.table) if (k >= 0) cp.table[lec::PySet::find_free(cp.table, cp.mask, k)] = k;
    for (int64_t k : cp.table) if (k >= 0 && !is_hidden(k)) s->visible.push_back((int32_t)k);
  } else {                                                                      // a new set, filled in `so`'s iteration order
    lec::PySet res;
    for (int64_t k : so.table) if (k >= 0 && !is_hidden(k)) res.add(k);
    for (int64_t k : res.table) if (k >= 0) s->visible.push_back((int32_t)k);
  }
  return LEC_OK;
}

// The slot ids left after hiding, in the order oe_h.py:854 indexes them (see above).  out: up to L + 1 entries; returns their count in *n.
extern "C" int lec_sampler_visible_slots(const lec_sampler* s, int32_t* out, int* n) {
  if (!s || !out || !n) { lec::set_error("sampler_visible_slots: bad arguments"); return LEC_E_ARG; }
  if (s->hidden.empty()) { for (int32_t l = 0; l <= s->L; ++l) out[l] = l; *n = s->L + 1; return LEC_OK; }
  std::copy(s->visible.begin(), s->visible.end(), out); *n = (int)s->visible.size();
  return LEC_OK;
}

static inline int draw_one(lec_sampler* s, int side, int32_t node, int32_t level_id, int32_t* out) {
  if (node < 0 || node >= s->n_nodes) { lec::set_error("sampler_draw: node %d out of range", node); return LEC_E_ARG; }
  const int L = s->L;
  int32_t lvl;
  if (level_id < 0) { lec::set_error("sampler_draw: negative level_id"); return LEC_E_ARG; }
  if (s->mode == 1) lvl = level_id % L;                                            // order_embeddings.py:799
  else if (!s->hidden.empty()) {                                                    // oe_h.py:850-854
    int32_t m = L - (int32_t)s->hidden.size() + 1;
    if (m <= 0) { lec::set_error("sampler_draw: all levels hidden"); return LEC_E_ARG; }
    int32_t i = level_id % m;
    if (i >= (int32_t)s->visible.size()) { lec::set_error("sampler_draw: hidden-level remap out of range"); return LEC_E_ARG; }
    lvl = s->visible[i];
  } else lvl = level_id % (L + 1);                                                  // oe_h.py:881
  int64_t lo = 0, hi = s->n_nodes;
  if (s->pick_per_level) {
    if (lvl < L) { lo = s->level_start[lvl]; hi = s->level_stop[lvl]; }             // oe_h.py:890-892
    else if (s->mode == 0) {                                                        // :893-898
      if (node >= s->n_labels) { lo = 0; hi = s->n_labels; } else { lo = s->n_labels; hi = s->n_nodes; }
    }
  }
  const int32_t* ex = side == 0 ? s->desc.data() + s->desc_ptr[node] : s->anc.data() + s->anc_ptr[node];
  const int64_t len = side == 0 ? s->desc_ptr[node + 1] - s->desc_ptr[node] : s->anc_ptr[node + 1] - s->anc_ptr[node];
  const int32_t* e0 = std::lower_bound(ex, ex + len, (int32_t)lo);
  const int32_t* e1 = std::lower_bound(e0, ex + len, (int32_t)hi);
  const int64_t m = e1 - e0;
  const int64_t count = (hi - lo) - m;
  if (count <= 0) { lec::set_error("sampler_draw: empty candidate list (node %d, level slot %d)", node, lvl); return LEC_E_EMPTY; }
  const int64_t r = s->rng.randbelow((uint32_t)count);
  // j = #{i < m : e0[i] - lo - i <= r}   (excluded nodes that precede the r-th survivor)
  int64_t a = 0, b = m;
  while (a < b) { int64_t mid = (a + b) >> 1; if ((int64_t)e0[mid] - lo - mid <= r) a = mid + 1; else b = mid; }
  *out = (int32_t)(lo + r + a);
  return LEC_OK;
}

extern "C" int lec_sampler_draw(lec_sampler* s, int side, int32_t node, int32_t level_id, int32_t* out) {
  if (!s || !out || (side != 0 && side != 1)) { lec::set_error("sampler_draw: bad arguments"); return LEC_E_ARG; }
  return draw_one(s, side, node, level_id, out);
}

extern "C" int lec_sampler_draw_batch(lec_sampler* s, const int32_t* pos_from, const int32_t* pos_to, int B, int K,
                                      int32_t* neg) {
  if (!s || B < 0 || K < 0 || (B && (!pos_from || !pos_to)) || (B && K && !neg)) { lec::set_error("sampler_draw_batch: bad arguments"); return LEC_E_ARG; }
  for (int b = 0; b < B; ++b)
    for (int p = 0; p < K; ++p) {                                                   // oe_h.py:948-957
      int rc = draw_one(s, 0, pos_from[b], p, &neg[(int64_t)b * 2 * K + p]);
      if (rc) return rc;
      rc = draw_one(s, 1, pos_to[b], p, &neg[(int64_t)b * 2 * K + K + p]);
      if (rc) return rc;
    }
  return LEC_OK;
}

extern "C" int lec_sampler_next_u32(lec_sampler* s, uint32_t* out) {
  if (!s || !out) { lec::set_error("sampler_next_u32: bad arguments"); return LEC_E_ARG; }
  *out = s->rng.u32(); return LEC_OK;
}

extern "C" int64_t lec_sampler_tc_edges(const lec_sampler* s) { return s ? s->tc_edges : -1; }

// The transitive closure itself, as CSR over all nodes: node u's descendants (ascending, u excluded) are
// adj[ptr[u] .. ptr[u + 1]).  ptr: n_nodes + 1 entries, adj: lec_sampler_tc_edges() entries (pass adj = NULL to get ptr only).
// Lets the host build G_train_tc (oe_h.py:539 nx.transitive_closure) from the closure the sampler already holds.
extern "C" int lec_sampler_tc_export(const lec_sampler* s, int64_t* ptr, int32_t* adj) {
  if (!s || !ptr) { lec::set_error("sampler_tc_export: bad arguments"); return LEC_E_ARG; }
  int64_t k = 0;
  for (int64_t u = 0; u < s->n_nodes; ++u) {
    ptr[u] = k;
    for (int64_t i = s->desc_ptr[u]; i < s->desc_ptr[u + 1]; ++i) {
      if (s->desc[i] == (int32_t)u) continue;
      if (adj) adj[k] = s->desc[i];
      ++k;
    }
  }
  ptr[s->n_nodes] = k;
  return LEC_OK;
}
