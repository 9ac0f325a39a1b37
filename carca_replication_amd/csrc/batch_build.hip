// f1: batch construction on the device -- the work of data.py:53-192 for a whole batch in one
// launch, from the interaction log kept in HBM (CSR: `offs` per user into `hist` item ids and `hctx` context rows).
//   leave-one-out window      pad_profile              data.py:53-74     (held_out / floor chosen by the host per split)
//   negatives                 sample_negatives         data.py:77-87     distinct, in [1, n_items-1], not in the
//                                                                       user's WHOLE history (the reference passes
//                                                                       the full profile, data.py:103,153)
//   evaluation sample         get_test_sequences       data.py:140-192   candidate 0 = held-out item, then N
//                                                                       negatives; every candidate carries the
//                                                                       held-out interaction's context (:185)
//   training sample           get_train_sequences      data.py:90-137    slot t: item, its successor (positive) and
//                                                                       one negative with the positive's context (:130)
// Only ids and context rows are produced: attribute rows are gathered from the device table inside the feature GEMM
// (AllEmbedding.register_attr_table).  The negatives come from a counter-based hash (seed, user, attempt), not from
// python's `random`: same distribution (uniform without replacement by rejection), reproducible per seed.
// One wave per user; HBM/latency-bound integer work: no tiling, no LDS beyond the accepted list.
#include "carca_common.h"
#include "../../include/carca_hip.h"

namespace {

__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// candidate k of user u: uniform in [1, n_items - 1]
__device__ __forceinline__ int candidate(unsigned long long seed, unsigned user, unsigned k, int n_items) {
  unsigned h = mix32((unsigned)seed ^ (user * 0x9E3779B1u));
  h = mix32(h ^ (unsigned)(seed >> 32) ^ (k * 0x85EBCA77u));
  return 1 + (int)(((unsigned long long)h * (unsigned)(n_items - 1)) >> 32);
}

constexpr int NEG_MAX = 2048;  // negatives per user kept in LDS

// Fill acc[0..want) with distinct ids in [1, n_items-1] that are not in hist[0..n): whole wave, lane-parallel rejection.
// Every wave leaves after at most MAX_ROUNDS rounds; if the hash has not delivered by then (only when nearly every item
// is in the history) the remaining slots are filled by a linear scan over the ids, which always terminates.
__device__ void sample_negatives(const int32_t* __restrict__ hist, int n, int n_items, int want, unsigned long long seed,
                                 unsigned user, int* acc /*LDS*/, int lane) {
  constexpr int MAX_ROUNDS = 256;
  int have = 0;
  for (int round = 0; round < MAX_ROUNDS && have < want; ++round) {
    const int c = candidate(seed, user, (unsigned)(round * 64 + lane), n_items);
    bool ok = true;
    for (int j = 0; j < n; ++j) ok = ok && hist[j] != c;         // (uniform address: one broadcast load per step)
    for (int j = 0; j < have; ++j) ok = ok && acc[j] != c;
    for (int j = 0; j < 63; ++j) {                                 // duplicates inside this round: the lower lane wins
      const int cj = __shfl(c, j);
      const bool okj = __shfl((int)ok, j) != 0;
      if (j < lane && okj && cj == c) ok = false;
    }
    const unsigned long long m = __ballot(ok);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (ok && have + rank < want) acc[have + rank] = c;
    have = min(want, have + __popcll(m));
    __builtin_amdgcn_wave_barrier();
  }
  for (int id = 1; have < want && id < n_items; ++id) {           // fallback, practically never entered
    bool ok = true;
    for (int j = 0; j < n; ++j) ok = ok && hist[j] != id;
    for (int j = 0; j < have; ++j) ok = ok && acc[j] != id;
    if (ok) {
      if (lane == 0) acc[have] = id;
      ++have;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

struct BuildArgs {
  const int32_t* hist;     // [total] item ids, users' histories back to back, interaction order
  const int64_t* offs;     // [n_users + 1]
  const float* hctx;       // [total, n_ctx] context row of every interaction
  const int32_t* users;    // [B] user indices of this batch (into offs)
  int B, L, N, n_ctx, n_items, held_out, floor_, n_users;
  unsigned long long seed;
  int32_t* p_x;   // [B, L]
  float* p_c;     // [B, L, n_ctx]
  int32_t* o_x;   // eval [B, 1 + N]; train [B, 2L]
  float* o_c;     // eval [B, 1 + N, n_ctx]; train [B, 2L, n_ctx]
  int32_t* y_true;
};

// window of data.py:53-74: indices [start, stop) of the user's history, the last one is the entry to predict
__device__ __forceinline__ bool window(int n, int L, int held_out, int floor_, int& start, int& stop) {
  if (n <= floor_) return false;
  stop = max(floor_, n - held_out);
  start = max(0, n - held_out - L - 1);
  return true;
}

__global__ __launch_bounds__(256) void build_eval_kernel(const BuildArgs a) {
  __shared__ int acc_s[4][NEG_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= a.B) return;
  int* acc = acc_s[wave];
  const int u_raw = a.users[b];
  const bool u_ok = u_raw >= 0 && u_raw < a.n_users;  // an index outside the log reads as a user without history: an all-pad row
  const int u = u_ok ? u_raw : 0;
  const long o0 = a.offs[u];
  const int n = u_ok ? (int)(a.offs[u + 1] - o0) : 0;
  const int32_t* h = a.hist + o0;
  const int T = 1 + a.N, L = a.L, nc = a.n_ctx;
  int start = 0, stop = 0;
  const bool valid = window(n, L, a.held_out, a.floor_, start, stop);
  const int nh = valid ? stop - 1 - start : 0;  // history slots, left-padded into L
  for (int t = lane; t < L; t += 64) {
    const int src = t - (L - nh);
    a.p_x[(size_t)b * L + t] = src >= 0 ? h[start + src] : 0;
  }
  for (int i = lane; i < L * nc; i += 64) {
    const int t = i / nc, c = i - t * nc, src = t - (L - nh);
    a.p_c[(size_t)b * L * nc + i] = src >= 0 ? a.hctx[(size_t)(o0 + start + src) * nc + c] : 0.f;
  }
  if (!valid) {  // a user without a window in this split scores nothing (callers filter them out; stay defined anyway)
    for (int t = lane; t < T; t += 64) {
      a.o_x[(size_t)b * T + t] = 0;
      a.y_true[(size_t)b * T + t] = 0;
    }
    for (int i = lane; i < T * nc; i += 64) a.o_c[(size_t)b * T * nc + i] = 0.f;
    return;
  }
  sample_negatives(h, n, a.n_items, a.N, a.seed, (unsigned)u, acc, lane);
  const int held = h[stop - 1];
  for (int t = lane; t < T; t += 64) {
    a.o_x[(size_t)b * T + t] = t == 0 ? held : acc[t - 1];
    a.y_true[(size_t)b * T + t] = t == 0 ? 1 : 0;
  }
  for (int i = lane; i < T * nc; i += 64)  // every candidate carries the held-out interaction's context (data.py:185)
    a.o_c[(size_t)b * T * nc + i] = a.hctx[(size_t)(o0 + stop - 1) * nc + (i % nc)];
}

__global__ __launch_bounds__(256) void build_train_kernel(const BuildArgs a) {
  __shared__ int acc_s[4][NEG_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= a.B) return;
  int* acc = acc_s[wave];
  const int u_raw = a.users[b];
  const bool u_ok = u_raw >= 0 && u_raw < a.n_users;  // an index outside the log reads as a user without history: an all-pad row
  const int u = u_ok ? u_raw : 0;
  const long o0 = a.offs[u];
  const int n = u_ok ? (int)(a.offs[u + 1] - o0) : 0;
  const int32_t* h = a.hist + o0;
  const int L = a.L, nc = a.n_ctx;
  int start = 0, stop = 0;
  const bool valid = window(n, L, a.held_out, a.floor_, start, stop);
  const int ns = valid ? stop - 1 - start : 0;  // slots that have a successor (data.py:106)
  if (ns > 0) sample_negatives(h, n, a.n_items, ns, a.seed, (unsigned)u, acc, lane);
  const int lo = L - ns;
  for (int t = lane; t < L; t += 64) {
    const int s = t - lo;
    const bool on = s >= 0;
    a.p_x[(size_t)b * L + t] = on ? h[start + s] : 0;
    a.o_x[(size_t)b * 2 * L + t] = on ? h[start + s + 1] : 0;          // the item that followed slot t's item
    a.o_x[(size_t)b * 2 * L + L + t] = on ? acc[s] : 0;
    a.y_true[(size_t)b * 2 * L + t] = on ? 1 : 0;                      // p_x > 0 (item ids are >= 1)
    a.y_true[(size_t)b * 2 * L + L + t] = 0;
  }
  for (int i = lane; i < L * nc; i += 64) {
    const int t = i / nc, c = i - t * nc, s = t - lo;
    const bool on = s >= 0;
    a.p_c[(size_t)b * L * nc + i] = on ? a.hctx[(size_t)(o0 + start + s) * nc + c] : 0.f;
    const float pc = on ? a.hctx[(size_t)(o0 + start + s + 1) * nc + c] : 0.f;  // the positive's context ...
    a.o_c[(size_t)b * 2 * L * nc + i] = pc;
    a.o_c[(size_t)b * 2 * L * nc + (size_t)L * nc + i] = pc;                     // ... also for its negative (data.py:130)
  }
}

int check(const BuildArgs& a, const char* who, int n_out) {
  CARCA_CHECK_ARG(a.hist && a.offs && a.users && a.p_x && a.o_x && a.y_true && (a.n_ctx == 0 || (a.hctx && a.p_c && a.o_c)),
                  "%s: null pointer", who);
  CARCA_CHECK_ARG(a.B >= 1 && a.L >= 1 && a.n_ctx >= 0 && a.n_items >= 2 && a.held_out >= 0 && a.floor_ >= 1 && a.n_users >= 1,
                  "%s: bad dimensions", who);
  CARCA_CHECK_SUPPORTED(n_out >= 0 && n_out <= NEG_MAX, "%s: %d negatives per user > %d", who, n_out, NEG_MAX);
  return CARCA_OK;
}

}  // namespace

extern "C" int carca_build_eval_batch(const int32_t* hist, const int64_t* offs, const float* hctx, const int32_t* users,
                                      int n_users, int B, int L, int N, int n_ctx, int n_items, int held_out, int floor_,
                                      uint64_t seed, int32_t* p_x, float* p_c, int32_t* o_x, float* o_c, int32_t* y_true,
                                      void* stream_) {
  const BuildArgs a{hist, offs, hctx, users, B, L, N, n_ctx, n_items, held_out, floor_, n_users, seed, p_x, p_c, o_x, o_c, y_true};
  if (int rc = check(a, "build_eval_batch", N)) return rc;
  hipLaunchKernelGGL(build_eval_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream_, a);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_build_train_batch(const int32_t* hist, const int64_t* offs, const float* hctx, const int32_t* users,
                                       int n_users, int B, int L, int n_ctx, int n_items, int held_out, int floor_,
                                       uint64_t seed, int32_t* p_x, float* p_c, int32_t* o_x, float* o_c, int32_t* y_true,
                                       void* stream_) {
  const BuildArgs a{hist, offs, hctx, users, B, L, 0, n_ctx, n_items, held_out, floor_, n_users, seed, p_x, p_c, o_x, o_c, y_true};
  if (int rc = check(a, "build_train_batch", L)) return rc;
  hipLaunchKernelGGL(build_train_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream_, a);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}
