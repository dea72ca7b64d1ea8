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
extern "C" int adnm_abi_version(void) { return 3; }   // 3: fused LayerNorm epilogue arguments of ssd_reduce_fwd, cast_* entry points

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
constexpr int kFoldCols = 64, kFoldSlices = 16, kMaxFolds = 16;
struct FoldSegs {
  float* ptr[4];
  int end[4];
};
struct FoldDesc {
  const float* part;
  int rows, n;
  FoldSegs segs;
};
struct MultiFold {
  int count;
  int blk_end[kMaxFolds];   // exclusive prefix of column-block counts
  FoldDesc d[kMaxFolds];
};
__global__ __launch_bounds__(kFoldCols* kFoldSlices) void fold_rows_kernel(MultiFold mf_by_value) {
  __shared__ float sm[kFoldSlices][kFoldCols + 1];
  // The descriptor table is indexed with a run-time (workgroup-uniform) k.  Indexing the by-value argument makes the compiler copy the
  // whole 1.1 KB struct into per-thread scratch first (measured: a 3.7 MB fold took 167 us); reading it through the kernarg segment
  // pointer keeps it in constant memory / scalar loads.  The explicit arguments of a HIP kernel start at offset 0 of that segment.
  (void)mf_by_value;
  const __attribute__((address_space(4))) MultiFold& mf = *(const __attribute__((address_space(4))) MultiFold*)__builtin_amdgcn_kernarg_segment_ptr();
  int k = 0;
  while (k + 1 < mf.count && (int)blockIdx.x >= mf.blk_end[k]) ++k;
  const __attribute__((address_space(4))) FoldDesc& fd = mf.d[k];
  const float* __restrict__ part = fd.part;
  const int rows = fd.rows, n = fd.n;
  const int cl = threadIdx.x & (kFoldCols - 1), sl = threadIdx.x / kFoldCols;
  const int c = ((int)blockIdx.x - (k ? mf.blk_end[k - 1] : 0)) * kFoldCols + cl;
  float acc = 0.f;
  if (c < n) {
    // independent loads: 8 in flight per lane, two accumulators (fixed order -> still deterministic)
    float acc2 = 0.f;
    int r = sl;
    for (; r + 7 * kFoldSlices < rows; r += 8 * kFoldSlices) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(r + u * kFoldSlices) * n + c];
      acc += (v[0] + v[1]) + (v[2] + v[3]);
      acc2 += (v[4] + v[5]) + (v[6] + v[7]);
    }
    for (; r < rows; r += kFoldSlices) acc += part[(int64_t)r * n + c];
    acc += acc2;
  }
  sm[sl][cl] = acc;
  __syncthreads();
  if (sl == 0 && c < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < kFoldSlices; ++q) t += sm[q][cl];
    int begin = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (c < fd.segs.end[q]) {
        if (fd.segs.ptr[q]) fd.segs.ptr[q][c - begin] = t;
        break;
      }
      begin = fd.segs.end[q];
    }
  }
}

struct FoldQueue {
  std::vector<FoldDesc> pending;
  std::vector<const char*> names;
};
thread_local FoldQueue* tls_foldq = nullptr;

// measurement aid: ADNM_FOLD_BATCH=1 launches every queued fold on its own (and under its own profiler name)
int fold_batch_limit() {
  static const int lim = [] {
    const char* e = getenv("ADNM_FOLD_BATCH");
    const int v = e ? atoi(e) : kMaxFolds;
    return v < 1 ? 1 : (v > kMaxFolds ? kMaxFolds : v);
  }();
  return lim;
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
      blocks += (int)adnm_cdiv(d[i0 + k].n, kFoldCols);
      mf.blk_end[k] = blocks;
      bytes += 4.0 * ((double)d[i0 + k].rows + 1) * d[i0 + k].n;
    }
    for (int k = m; k < kMaxFolds; ++k) mf.blk_end[k] = blocks;
    ADNM_PROF(m == 1 ? names[i0] : "fold_batch", st, bytes);
    fold_rows_kernel<<<(unsigned)blocks, kFoldCols * kFoldSlices, 0, st>>>(mf);
  }
}
}  // namespace

void adnm_launch_fold(const char* prof_name, const float* part, int rows, int n, AdnmFoldSeg s0, AdnmFoldSeg s1, AdnmFoldSeg s2,
                      AdnmFoldSeg s3, hipStream_t st) {
  FoldDesc fd;
  fd.part = part, fd.rows = rows, fd.n = n;
  const AdnmFoldSeg in[4] = {s0, s1, s2, s3};
  int end = 0;
  for (int k = 0; k < 4; ++k) {
    end += in[k].len;
    fd.segs.ptr[k] = in[k].ptr;
    fd.segs.end[k] = end;
  }
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
extern "C" int adnm_foldq_flush(void* q, adnm_stream_t stream) {
  ADNM_REQUIRE(q, "foldq_flush: null queue");
  FoldQueue* fq = (FoldQueue*)q;
  if (fq->pending.empty()) return ADNM_OK;
  launch_folds(fq->pending.data(), fq->names.data(), (int)fq->pending.size(), (hipStream_t)stream);
  fq->pending.clear();
  fq->names.clear();
  ADNM_CHECK_LAUNCH("foldq_flush");
  return ADNM_OK;
}

// Column sums of a contiguous (rows, n) fp32 matrix with the deterministic fold kernel: the bias gradient of a Linear whose
// weight gradient went to the library GEMM (torch's own sum(0) costs 7-11 us on these 64 .. 1024-row matrices).
extern "C" int adnm_colsum(const float* x, float* out, int64_t rows, int64_t n, adnm_stream_t stream) {
  ADNM_REQUIRE(x && out, "colsum: null pointer");
  ADNM_REQUIRE(rows > 0 && n > 0 && rows < (1ll << 31) && n < (1ll << 31), "colsum: bad shape rows=%lld n=%lld", (long long)rows, (long long)n);
  adnm_launch_fold("colsum", x, (int)rows, (int)n, {out, (int)n}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, (hipStream_t)stream);
  ADNM_CHECK_LAUNCH("colsum");
  return ADNM_OK;
}
