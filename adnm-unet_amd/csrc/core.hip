// Error reporting, ABI version and the opt-in per-kernel HIP-event profiler of libadnm_hip.
#include "adnm_common.h"
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

static thread_local char g_err[512] = "";

void adnm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* adnm_last_error(void) { return g_err; }
extern "C" int adnm_abi_version(void) { return 10; }   // 10: narrow weight shadows (b_dtype / b_scale of adnm_skgemm, shadow args + hyper of adnm_adamw_step, adnm_shadow_refresh, narrow of adnm_adnprep_fwd_multi), counters / slabs_uc of adnm_skgemm; 9: the split-K workspace is the caller's (adnm_uncached_alloc / _free; the per-device rings are gone); 8: adnm_mixnorm_*, adnm_bridge_pool_* (adnm_tokmean_* gone), up1 / up2 of adnm_haar_idwt; 7: precision ladder (`q` of the GEMM-shaped entry points, adnm_quant_update, fp8); 6: tap_ld of adnm_adnprep_*; 5: `prec` of adnm_tsgemm_nt; 4: workspace of adnm_colsum

// ---- profiler: OFF by default (one relaxed atomic load per launch).  When bench.py enables it, every kernel
// launch of the library is bracketed by hipEventRecord on the stream it is launched on; adnm_prof_collect()
// synchronises the events and reports, per kernel name, launches / total ms / algorithmic bytes.
namespace {
struct Rec {
  const char* name;
  hipEvent_t a, b;
  double bytes;
};
std::atomic<int> g_on{0};
std::mutex g_mu;
std::vector<Rec> g_recs;
}  // namespace

AdnmProfScope::AdnmProfScope(const char* name, hipStream_t st, double bytes) : st_(st), idx_(-1) {
  if (!g_on.load(std::memory_order_relaxed)) return;
  Rec r{name, nullptr, nullptr, bytes};
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  hipEventRecord(r.a, st);
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.push_back(r);
  idx_ = (long)g_recs.size() - 1;
}

AdnmProfScope::~AdnmProfScope() {
  if (idx_ < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if ((size_t)idx_ < g_recs.size()) hipEventRecord(g_recs[idx_].b, st_);
}

extern "C" int adnm_prof_enable(int on) {
  g_on.store(on ? 1 : 0);
  return ADNM_OK;
}

extern "C" int64_t adnm_prof_collect(char* buf, int64_t buflen) {
  std::lock_guard<std::mutex> lk(g_mu);
  struct Agg {
    long n = 0;
    double ms = 0, bytes = 0;
  };
  std::map<std::string, Agg> agg;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      char key[160];
      snprintf(key, sizeof(key), "%s@%.0f", r.name, r.bytes);  // one entry per (kernel, algorithmic bytes per launch) = per shape
      Agg& a = agg[key];
      a.n += 1;
      a.ms += ms;
      a.bytes += r.bytes;
    }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  g_recs.clear();
  std::string out;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s\t%ld\t%.6f\t%.0f\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.bytes);
    out += line;
  }
  if (buf && buflen > 0) {
    const size_t n = out.size() < (size_t)buflen - 1 ? out.size() : (size_t)buflen - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (int64_t)out.size();
}

// ---- shared cross-block fold of fp32 partial rows (second stage of every deterministic reduction in the library)
// One launch folds up to kMaxFolds independent partial sets ("descriptors"): immediately (one descriptor) or, when the calling
// thread has bound a caller-owned fold queue (adnm_foldq_*), later and together with the other queued sets — the parameter
// gradients of a backward pass are not read before the optimiser, so their ~200 second-stage launches per step collapse
// into a few.  The arithmetic per descriptor (slice / unroll order) is identical either way: results do not depend on batching.
namespace {
// A workgroup of 1024 threads folds `cols` columns with 1024 / cols row slices: 64 x 16 by default; partial sets with many rows and
// few columns (per-channel scalars, bias sums: 1024 x 4, 1024 x 131 ...) would leave one or two workgroups walking hundreds of rows
// each, so they take 16 x 64 or 4 x 256; sets with few rows take 128 .. 1024 columns.  The geometry is a function of (rows, n) only: a set is summed in the same order whether it
// is folded at once or from the queue.
constexpr int kFoldThreads = 1024, kMaxFolds = 48;   // (48 descriptors = 3.7 KB of the 4 KB kernel-argument segment)
struct FoldSegs {
  float* ptr[4];
  int end[4];
};
struct FoldDesc {
  const float* part;
  int rows, n;
  int lc;   // log2(columns per workgroup): 2 .. 10
  int acc;  // bit q: segment q ADDS the folded value to its destination instead of overwriting it (adnm_foldq_accumulate_next)
  FoldSegs segs;
};
struct MultiFold {
  int count;
  int blk_end[kMaxFolds];   // exclusive prefix of column-block counts
  FoldDesc d[kMaxFolds];
};
int fold_lc(int rows, int n) {
  if (rows >= 1024 && n <= 64) return 2;
  if (rows >= 256 && n <= 2048) return 4;
  // few partial rows (the split-K slabs of a weight gradient: 2 .. 16 rows x 100 K+ columns): as many columns per workgroup as rows
  // allow, or the launch is ~10 K workgroups of 16 waves of which 12 never load anything — wave launch rate, not bandwidth
  if (rows <= 4) return 10;
  if (rows <= 8) return 9;
  if (rows <= 16) return 8;
  if (rows <= 32) return 7;
  return 6;
}
constexpr int kFoldVec = 16;   // FoldDesc::acc bit 4: the set is folded four columns per lane (16-byte loads)
__global__ __launch_bounds__(kFoldThreads) void fold_rows_kernel(MultiFold mf_by_value) {
  __shared__ __attribute__((aligned(16))) float sm[4 * (kFoldThreads + kFoldThreads / 4)];
  // The descriptor table is indexed with a run-time (workgroup-uniform) k: read it through the kernarg segment pointer (constant
  // memory / scalar loads).  The explicit arguments of a HIP kernel start at offset 0 of that segment.
  (void)mf_by_value;
  const __attribute__((address_space(4))) MultiFold& mf = *(const __attribute__((address_space(4))) MultiFold*)__builtin_amdgcn_kernarg_segment_ptr();
  // which descriptor: count the prefix entries at or below blockIdx with CONSTANT indices — three s_load_dwordx16 and 48 scalar compares; a
  // `while (blockIdx >= blk_end[k]) ++k` walk is a chain of up to 48 dependent scalar loads in front of every workgroup of the late sets
  // (unused entries repeat the total: launch_folds)
  int k = 0;
#pragma unroll
  for (int i = 0; i < kMaxFolds - 1; ++i) k += (int)blockIdx.x >= mf.blk_end[i] ? 1 : 0;
  const __attribute__((address_space(4))) FoldDesc& fd = mf.d[k];
  const float* __restrict__ part = fd.part;
  const int rows = fd.rows, n = fd.n, lc = fd.lc;
  const int cols = 1 << lc, slices = kFoldThreads >> lc;
  const int cl = threadIdx.x & (cols - 1), sl = threadIdx.x >> lc;
  const int blk = (int)blockIdx.x - (k ? mf.blk_end[k - 1] : 0);
  if (fd.acc & kFoldVec) {
    // large sets (a weight gradient's split-K slabs: a few rows x 10^5 .. 10^6 columns): four adjacent columns per lane.  Row slices, unroll
    // and tree are those of the scalar path, so every column is summed in the same order — bit for bit the scalar result, a quarter of the
    // load instructions (the scalar walk ran at 1.2 TB/s on the 50 MB sets)
    const int c = ((blk << lc) + cl) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < n) {
      float4 acc2 = make_float4(0.f, 0.f, 0.f, 0.f);
      int r = sl;
      for (; r + 7 * slices < rows; r += 8 * slices) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(part + (int64_t)(r + u * slices) * n + c);
        acc.x += (v[0].x + v[1].x) + (v[2].x + v[3].x), acc.y += (v[0].y + v[1].y) + (v[2].y + v[3].y);
        acc.z += (v[0].z + v[1].z) + (v[2].z + v[3].z), acc.w += (v[0].w + v[1].w) + (v[2].w + v[3].w);
        acc2.x += (v[4].x + v[5].x) + (v[6].x + v[7].x), acc2.y += (v[4].y + v[5].y) + (v[6].y + v[7].y);
        acc2.z += (v[4].z + v[5].z) + (v[6].z + v[7].z), acc2.w += (v[4].w + v[5].w) + (v[6].w + v[7].w);
      }
      for (; r < rows; r += slices) {
        const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)r * n + c);
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
      }
      acc.x += acc2.x, acc.y += acc2.y, acc.z += acc2.z, acc.w += acc2.w;
    }
    float4* const mine = reinterpret_cast<float4*>(sm) + sl * (cols + 1) + cl;
    *mine = acc;
    __syncthreads();
    for (int h = slices >> 1; h >= 1; h >>= 1) {   // fixed tree over the row slices
      if (sl < h) {
        const float4 o = mine[h * (cols + 1)];
        float4 m = *mine;
        m.x += o.x, m.y += o.y, m.z += o.z, m.w += o.w;
        *mine = m;
      }
      __syncthreads();
    }
    if (sl == 0 && c < n) {
      const float4 t = *mine;
      int begin = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (c < fd.segs.end[q]) {   // (segment boundaries are multiples of 4 in this mode: the four columns share a segment)
          if (fd.segs.ptr[q]) {
            float4* dst = reinterpret_cast<float4*>(fd.segs.ptr[q] + (c - begin));
            if (fd.acc & (1 << q)) {
              float4 d = *dst;
              d.x += t.x, d.y += t.y, d.z += t.z, d.w += t.w;
              *dst = d;
            } else {
              *dst = t;
            }
          }
          break;
        }
        begin = fd.segs.end[q];
      }
    }
    return;
  }
  const int c = (blk << lc) + cl;
  float acc = 0.f;
  if (c < n) {
    // independent loads: 8 in flight per lane, two accumulators (fixed order -> still deterministic)
    float acc2 = 0.f;
    int r = sl;
    for (; r + 7 * slices < rows; r += 8 * slices) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(r + u * slices) * n + c];
      acc += (v[0] + v[1]) + (v[2] + v[3]);
      acc2 += (v[4] + v[5]) + (v[6] + v[7]);
    }
    for (; r < rows; r += slices) acc += part[(int64_t)r * n + c];
    acc += acc2;
  }
  float* const mine = sm + sl * (cols + 1) + cl;
  *mine = acc;
  __syncthreads();
  for (int h = slices >> 1; h >= 1; h >>= 1) {   // fixed tree over the row slices
    if (sl < h) *mine += mine[h * (cols + 1)];
    __syncthreads();
  }
  if (sl == 0 && c < n) {
    const float t = *mine;
    int begin = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (c < fd.segs.end[q]) {
        if (fd.segs.ptr[q]) {
          if (fd.acc & (1 << q)) fd.segs.ptr[q][c - begin] += t;
          else fd.segs.ptr[q][c - begin] = t;
        }
        break;
      }
      begin = fd.segs.end[q];
    }
  }
}

inline int fold_blocks(const FoldDesc& d) { return (int)adnm_cdiv(d.n, (int64_t)(1 << d.lc) * ((d.acc & kFoldVec) ? 4 : 1)); }

struct FoldQueue {
  std::vector<FoldDesc> pending;
  std::vector<const char*> names;
};
thread_local FoldQueue* tls_foldq = nullptr;
thread_local int tls_fold_acc_next = 0;   // accumulate mask of the next fold queued on this thread (cleared when used)

// measurement aid: ADNM_FOLD_BATCH=1 launches every queued fold on its own (and under its own profiler name)
int fold_batch_limit() {
  static const int lim = [] {
    const char* e = getenv("ADNM_FOLD_BATCH");
    const int v = e ? atoi(e) : kMaxFolds;
    return v < 1 ? 1 : (v > kMaxFolds ? kMaxFolds : v);
  }();
  return lim;
}

bool fold_debug() {
  static const bool v = [] {
    const char* e = getenv("ADNM_FOLD_DEBUG");
    return e && e[0] == '1';
  }();
  return v;
}

void launch_folds(const FoldDesc* d, const char* const* names, int count, hipStream_t st) {
  const int lim = fold_batch_limit();
  for (int i0 = 0; i0 < count; i0 += lim) {
    MultiFold mf;
    const int m = count - i0 < lim ? count - i0 : lim;
    mf.count = m;
    int blocks = 0;
    double bytes = 0;
    for (int k = 0; k < m; ++k) {
      mf.d[k] = d[i0 + k];
      blocks += fold_blocks(d[i0 + k]);
      mf.blk_end[k] = blocks;
      bytes += 4.0 * ((double)d[i0 + k].rows + 1) * d[i0 + k].n;
    }
    for (int k = m; k < kMaxFolds; ++k) mf.blk_end[k] = blocks;
    if (fold_debug()) {   // measurement aid: ADNM_FOLD_DEBUG=1 lists every fold launch (name: partial rows x columns)
      fprintf(stderr, "[adnm fold] %d blocks:", blocks);
      for (int k = 0; k < m; ++k) fprintf(stderr, " %s:%dx%d", names[i0 + k], d[i0 + k].rows, d[i0 + k].n);
      fprintf(stderr, "\n");
    }
    ADNM_PROF(m == 1 ? names[i0] : "fold_batch", st, bytes);
    fold_rows_kernel<<<(unsigned)blocks, kFoldThreads, 0, st>>>(mf);
  }
}
}  // namespace

void adnm_launch_fold(const char* prof_name, const float* part, int rows, int n, AdnmFoldSeg s0, AdnmFoldSeg s1, AdnmFoldSeg s2,
                      AdnmFoldSeg s3, hipStream_t st) {
  FoldDesc fd;
  fd.part = part, fd.rows = rows, fd.n = n, fd.lc = fold_lc(rows, n);
  fd.acc = tls_foldq ? tls_fold_acc_next : 0;   // accumulation exists for queued folds only: the flush orders it behind the overwriting ones
  tls_fold_acc_next = 0;
  const AdnmFoldSeg in[4] = {s0, s1, s2, s3};
  int end = 0;
  bool vec = n >= 2048 && n % 4 == 0 && ((uintptr_t)part & 15) == 0;   // four columns per lane: every segment starts and ends on a 16-byte boundary
  for (int k = 0; k < 4; ++k) {
    end += in[k].len;
    fd.segs.ptr[k] = in[k].ptr;
    fd.segs.end[k] = end;
    vec = vec && in[k].len % 4 == 0 && ((uintptr_t)in[k].ptr & 15) == 0;
  }
  if (vec) fd.acc |= kFoldVec;
  if (tls_foldq) {   // deferred: the caller keeps `part` and the destinations alive until adnm_foldq_flush
    tls_foldq->pending.push_back(fd);
    tls_foldq->names.push_back(prof_name);
    return;
  }
  launch_folds(&fd, &prof_name, 1, st);
}

extern "C" void* adnm_foldq_create(void) { return new FoldQueue(); }
extern "C" int adnm_foldq_destroy(void* q) {
  if (tls_foldq == q) tls_foldq = nullptr;
  delete (FoldQueue*)q;
  return ADNM_OK;
}
extern "C" int adnm_foldq_bind(void* q) {
  tls_foldq = (FoldQueue*)q;
  return ADNM_OK;
}
extern "C" int64_t adnm_foldq_pending(void* q) { return q ? (int64_t)((FoldQueue*)q)->pending.size() : 0; }
extern "C" int adnm_foldq_clear(void* q) {   // forget what is queued without launching it (the caller's error path)
  ADNM_REQUIRE(q, "foldq_clear: null queue");
  ((FoldQueue*)q)->pending.clear();
  ((FoldQueue*)q)->names.clear();
  return ADNM_OK;
}
extern "C" int adnm_foldq_accumulate_next(int mask) {
  tls_fold_acc_next = mask & 15;
  return ADNM_OK;
}
extern "C" int adnm_foldq_flush(void* q, adnm_stream_t stream) {
  ADNM_REQUIRE(q, "foldq_flush: null queue");
  FoldQueue* fq = (FoldQueue*)q;
  if (fq->pending.empty()) return ADNM_OK;
  // the overwriting folds first, then — in later launches, i.e. ordered behind them on the stream — the accumulating ones; two
  // accumulating folds that share a destination never share a launch (the workgroups of one launch run concurrently)
  std::vector<FoldDesc> plain, accs;
  std::vector<const char*> pn, an;
  for (size_t i = 0; i < fq->pending.size(); ++i) {
    if (fq->pending[i].acc & 15) accs.push_back(fq->pending[i]), an.push_back(fq->names[i]);
    else plain.push_back(fq->pending[i]), pn.push_back(fq->names[i]);
  }
  if (!plain.empty()) launch_folds(plain.data(), pn.data(), (int)plain.size(), (hipStream_t)stream);
  while (!accs.empty()) {
    std::vector<FoldDesc> round, rest;
    std::vector<const char*> rn, restn;
    for (size_t i = 0; i < accs.size(); ++i) {
      bool clash = false;
      for (const FoldDesc& r : round)
        for (int a = 0; a < 4 && !clash; ++a)
          for (int b = 0; b < 4 && !clash; ++b)
            clash = r.segs.ptr[a] && r.segs.ptr[a] == accs[i].segs.ptr[b] && ((r.acc >> a) & 1) && ((accs[i].acc >> b) & 1);
      if (clash) rest.push_back(accs[i]), restn.push_back(an[i]);
      else round.push_back(accs[i]), rn.push_back(an[i]);
    }
    launch_folds(round.data(), rn.data(), (int)round.size(), (hipStream_t)stream);
    accs.swap(rest), an.swap(restn);
  }
  fq->pending.clear();
  fq->names.clear();
  ADNM_CHECK_LAUNCH("foldq_flush");
  return ADNM_OK;
}

// ---- deferred leaf launches (adnm_common.h: AdnmLeaf)
namespace {
struct LeafQueue {
  std::vector<AdnmLeaf> items;
};
thread_local LeafQueue* tls_leafq = nullptr;
}  // namespace
bool adnm_leafq_active() { return tls_leafq && tls_foldq; }
bool adnm_leafq_push(const AdnmLeaf& leaf) {
  if (!tls_leafq || !tls_foldq) return false;   // (a leaf's partials are folded by a QUEUED fold: without the fold queue the order would break)
  tls_leafq->items.push_back(leaf);
  return true;
}
extern "C" void* adnm_leafq_create(void) { return new LeafQueue(); }
extern "C" int adnm_leafq_destroy(void* q) {
  if (tls_leafq == q) tls_leafq = nullptr;
  delete (LeafQueue*)q;
  return ADNM_OK;
}
extern "C" int adnm_leafq_bind(void* q) {
  tls_leafq = (LeafQueue*)q;
  return ADNM_OK;
}
extern "C" int64_t adnm_leafq_pending(void* q) { return q ? (int64_t)((LeafQueue*)q)->items.size() : 0; }
extern "C" int adnm_leafq_clear(void* q) {
  ADNM_REQUIRE(q, "leafq_clear: null queue");
  ((LeafQueue*)q)->items.clear();
  return ADNM_OK;
}
extern "C" int adnm_leafq_flush(void* q, adnm_stream_t stream) {
  ADNM_REQUIRE(q, "leafq_flush: null queue");
  LeafQueue* lq = (LeafQueue*)q;
  if (lq->items.empty()) return ADNM_OK;
  int rc = ADNM_OK;
  for (int kind = 0; kind < ADNM_LEAF_KINDS && rc == ADNM_OK; ++kind) {
    std::vector<const AdnmLeaf*> sel;
    for (const AdnmLeaf& l : lq->items)
      if (l.kind == kind) sel.push_back(&l);
    if (sel.empty()) continue;
    if (kind == ADNM_LEAF_SKGEMM_TN) rc = adnm_skgemm_tn_launch_multi(sel.data(), (int)sel.size(), (hipStream_t)stream);
    else if (kind == ADNM_LEAF_TSGEMM_TN) rc = adnm_tsgemm_tn_launch_multi(sel.data(), (int)sel.size(), (hipStream_t)stream);
    else rc = adnm_dwconv_wgrad_launch_multi(sel.data(), (int)sel.size(), kind == ADNM_LEAF_DWCONV_WGRAD_K3 ? 3 : 5, (hipStream_t)stream);
  }
  lq->items.clear();
  return rc;
}

// Column sums of a contiguous (rows, n) fp32 matrix — a bias gradient whose weight gradient is computed elsewhere (the transposed
// conv's, ops.ConvT2xFn).  Tall matrices (65 536 x 32 at config 2) first go through colsum_partial_kernel: each workgroup adds a
// contiguous band of rows (lanes along the columns, 256 / ncol row lanes, LDS tree), then the shared fold adds the bands.
namespace {
constexpr int kColsumThreads = 256, kColsumMinRows = 2048;
__global__ __launch_bounds__(kColsumThreads) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part, int rows, int n, int band) {
  __shared__ float sm[kColsumThreads];
  int ncol = 1;
  while (ncol < n && ncol < kColsumThreads) ncol <<= 1;   // columns a workgroup pass covers (power of two >= n, or 256)
  const int lanes = kColsumThreads / ncol, cl = threadIdx.x & (ncol - 1), rl = threadIdx.x / ncol;
  const int r0 = blockIdx.x * band, r1 = r0 + band < rows ? r0 + band : rows;
  for (int c0 = 0; c0 < n; c0 += ncol) {
    const int c = c0 + cl;
    float a0 = 0.f, a1 = 0.f;
    if (c < n) {
      int r = r0 + rl;
      for (; r + 3 * lanes < r1; r += 4 * lanes) {
        const float v0 = x[(int64_t)r * n + c], v1 = x[(int64_t)(r + lanes) * n + c], v2 = x[(int64_t)(r + 2 * lanes) * n + c],
                    v3 = x[(int64_t)(r + 3 * lanes) * n + c];
        a0 += v0 + v1, a1 += v2 + v3;
      }
      for (; r < r1; r += lanes) a0 += x[(int64_t)r * n + c];
    }
    sm[threadIdx.x] = a0 + a1;
    __syncthreads();
    for (int h = lanes >> 1; h >= 1; h >>= 1) {
      if (rl < h) sm[threadIdx.x] += sm[threadIdx.x + h * ncol];
      __syncthreads();
    }
    if (rl == 0 && c < n) part[(int64_t)blockIdx.x * n + c] = sm[threadIdx.x];
    __syncthreads();
  }
}
int colsum_bands(int64_t rows) { return rows < kColsumMinRows ? 0 : (int)(rows / 256 < 256 ? rows / 256 : 256); }
}  // namespace

extern "C" int64_t adnm_colsum_ws_bytes(int64_t rows, int64_t n) {
  const int bands = colsum_bands(rows);
  return bands ? (int64_t)bands * n * (int64_t)sizeof(float) : 16;
}
extern "C" int adnm_colsum(const float* x, float* out, int64_t rows, int64_t n, void* ws, int64_t ws_bytes, adnm_stream_t stream) {
  ADNM_REQUIRE(x && out, "colsum: null pointer");
  ADNM_REQUIRE(rows > 0 && n > 0 && rows < (1ll << 31) && n < (1ll << 31), "colsum: bad shape rows=%lld n=%lld", (long long)rows, (long long)n);
  hipStream_t st = (hipStream_t)stream;
  const int bands = colsum_bands(rows);
  if (!bands) {
    adnm_launch_fold("colsum", x, (int)rows, (int)n, {out, (int)n}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
    ADNM_CHECK_LAUNCH("colsum");
    return ADNM_OK;
  }
  if (!ws || ws_bytes < adnm_colsum_ws_bytes(rows, n)) {
    adnm_set_error("colsum: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_colsum_ws_bytes(rows, n));
    return ADNM_EWORKSPACE;
  }
  {
    ADNM_PROF("colsum_partial", st, 4.0 * (double)rows * n);
    colsum_partial_kernel<<<bands, kColsumThreads, 0, st>>>(x, (float*)ws, (int)rows, (int)n, (int)adnm_cdiv(rows, bands));
  }
  ADNM_CHECK_LAUNCH("colsum_partial");
  adnm_launch_fold("colsum", (const float*)ws, bands, (int)n, {out, (int)n}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("colsum");
  return ADNM_OK;
}


// ---- delayed per-tensor scaling of the fp8 configuration (include/adnm_hip.h: adnm_quant_update).  One workgroup: a model has a
// few hundred GEMM call sites, and a single workgroup can advance the step counter behind its own barrier.
namespace {
__global__ __launch_bounds__(256) void quant_update_kernel(AdnmQuant* __restrict__ tab, int n, float* __restrict__ state, float headroom) {
  const float c = state[0], period = state[1] < 1.f ? 1.f : state[1];
  const bool was_recording = fmodf(c, period) == 0.f;        // the step that just ended collected amax
  const bool will_record = fmodf(c + 1.f, period) == 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    AdnmQuant r = tab[i];
    if (was_recording) {
      if (r.amax_a > 0.f && r.fmax_a > 0.f) r.scale_a = r.fmax_a / (r.amax_a * headroom);
      if (r.amax_b > 0.f && r.fmax_b > 0.f) r.scale_b = r.fmax_b / (r.amax_b * headroom);
      r.amax_a = r.amax_b = 0.f;
    }
    r.record = will_record ? 1.f : 0.f;
    tab[i] = r;
  }
  __syncthreads();   // every thread has read the counter
  if (threadIdx.x == 0) state[0] = c + 1.f;
}
}  // namespace

extern "C" int adnm_quant_update(float* table, int64_t n, float* state, float headroom, adnm_stream_t stream) {
  ADNM_REQUIRE(table && state && n >= 0 && n < (1 << 20), "quant_update: bad arguments");
  ADNM_REQUIRE(headroom >= 1.f, "quant_update: headroom %f must be >= 1", headroom);
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("quant_update", st, 64.0 * (double)n);
  quant_update_kernel<<<1, 256, 0, st>>>(reinterpret_cast<AdnmQuant*>(table), (int)n, state, headroom);
  ADNM_CHECK_LAUNCH("quant_update");
  return ADNM_OK;
}


// Host-side allocation helpers for the workspace of split GEMM launches (include/adnm_hip.h).  Setup-time calls (they synchronise the
// device): never made by a compute entry point, never under stream capture.
extern "C" void* adnm_uncached_alloc(int64_t bytes) {
  if (bytes <= 0) return nullptr;
  void* ptr = nullptr;
  if (hipExtMallocWithFlags(&ptr, (size_t)bytes, hipDeviceMallocUncached) != hipSuccess || !ptr) {
    (void)hipGetLastError();
    adnm_set_error("adnm_uncached_alloc: hipExtMallocWithFlags(%lld bytes, hipDeviceMallocUncached) failed", (long long)bytes);
    return nullptr;
  }
  if (hipMemset(ptr, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(ptr);
    adnm_set_error("adnm_uncached_alloc: zero fill failed");
    return nullptr;
  }
  return ptr;
}
extern "C" int adnm_uncached_free(void* ptr) {
  if (!ptr) return ADNM_OK;
  if (hipFree(ptr) != hipSuccess) {
    (void)hipGetLastError();
    adnm_set_error("adnm_uncached_free: hipFree failed");
    return ADNM_EINVAL;
  }
  return ADNM_OK;
}
