// Context management and error plumbing of the C ABI (include/metafem_mi355x.h).
#include <stdarg.h>
#include <atomic>

#include <exception>
#include <mutex>
#include <new>
#include <set>

#include "common.h"
#include <string>

// contexts that exist: a pattern handle destroyed after its context (host-language finalisers run in any order) must not touch it
static std::mutex g_ctx_mutex;
static std::set<mfem_context_s*> g_live_contexts;
bool mfem_context_alive(mfem_context_s* ctx) {
  std::lock_guard<std::mutex> lk(g_ctx_mutex);
  return g_live_contexts.count(ctx) != 0;
}

static thread_local char g_err[1024] = "";

void mfem_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mfem_api_exception(const char* entry) noexcept {
  try {
    throw;  // the exception the entry point's handler caught
  } catch (const std::bad_alloc&) {
    mfem_set_error("%s: host memory exhausted (std::bad_alloc)", entry);
    return MFEM_ERR_ALLOC;
  } catch (const std::exception& e) {
    mfem_set_error("%s: internal error: %s", entry, e.what());
    return MFEM_ERR_INTERNAL;
  } catch (...) {
    mfem_set_error("%s: internal error (unknown C++ exception)", entry);
    return MFEM_ERR_INTERNAL;
  }
}

// countdown to an injected host allocation failure: 0 = off, k = the k-th probed allocation from now throws (once)
static std::atomic<int> g_fail_host_alloc{0};
extern "C" int mfem_debug_fail_host_alloc(int nth) try {
  g_fail_host_alloc.store(nth > 0 ? nth : 0);
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_fail_host_alloc")
void mfem_host_alloc_probe() {
  int v = g_fail_host_alloc.load();
  while (v > 0) {
    if (g_fail_host_alloc.compare_exchange_weak(v, v - 1)) {
      if (v == 1) throw std::bad_alloc();
      return;
    }
  }
}

std::atomic<int> mfem_debug_epoch{0};

extern "C" int mfem_abi_version(void) { return MFEM_ABI_VERSION; }
extern "C" const char* mfem_last_error(void) { return g_err; }

int mfem_ws_release(void* raw);  // (defined with the workspace allocator below)
static int context_allocate(mfem_context_s* c) {
  MFEM_CHECK_HIP(hipMalloc(&c->d_partials, sizeof(double) * MFEM_MAX_PARTIALS * 8));
  MFEM_CHECK_HIP(hipMalloc(&c->d_scalars, sizeof(double) * MFEM_NSCALARS));
  MFEM_CHECK_HIP(hipHostMalloc(&c->h_scalars, sizeof(double) * MFEM_NSCALARS));
  MFEM_CHECK_HIP(hipMalloc(&c->d_flags, sizeof(int32_t) * 24));
  MFEM_CHECK_HIP(hipHostMalloc(&c->h_flags, sizeof(int32_t) * 24));
  MFEM_CHECK_HIP(hipMemsetAsync(c->d_scalars, 0, sizeof(double) * MFEM_NSCALARS, c->stream));
  MFEM_CHECK_HIP(hipMemsetAsync(c->d_flags, 0, sizeof(int32_t) * 24, c->stream));
  MFEM_CHECK_HIP(hipEventCreate(&c->ev0));
  MFEM_CHECK_HIP(hipEventCreate(&c->ev1));
  return MFEM_OK;
}

static void context_release(mfem_context_s* ctx) {
  if (ctx->d_partials) hipFree(ctx->d_partials);
  if (ctx->d_scalars) hipFree(ctx->d_scalars);
  if (ctx->h_scalars) hipHostFree(ctx->h_scalars);
  if (ctx->d_flags) hipFree(ctx->d_flags);
  if (ctx->h_flags) hipHostFree(ctx->h_flags);
  (void)mfem_ws_release(ctx->ws_raw);
  (void)mfem_ws_release(ctx->ws_alt_raw);
  if (ctx->prof_ev) {
    for (int i = 0; i < 2 * MFEM_PROF_PAIRS; ++i)
      if (ctx->prof_ev[i]) hipEventDestroy(ctx->prof_ev[i]);
    delete[] ctx->prof_ev;
  }
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  for (int i = 0; i < MFEM_GRAPH_SLOTS; ++i)
    if (ctx->graph_exec[i]) hipGraphExecDestroy(ctx->graph_exec[i]);
  if (ctx->graph_ev) hipEventDestroy(ctx->graph_ev);
  if (ctx->graph_stream) hipStreamDestroy(ctx->graph_stream);
}

extern "C" int mfem_context_create(int device, void* stream, mfem_context* out) try {
  MFEM_REQUIRE(out != nullptr, "out is null");
  int ndev = 0;
  MFEM_CHECK_HIP(hipGetDeviceCount(&ndev));
  MFEM_REQUIRE(device >= 0 && device < ndev, "no such HIP device");
  MFEM_CHECK_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  MFEM_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    mfem_set_error("libmetafem_mi355x is built for gfx950 only, device %d is %s", device, prop.gcnArchName);
    return MFEM_ERR_UNSUPPORTED;
  }
  mfem_host_alloc_probe();
  mfem_context_s* c = new mfem_context_s();
  memset(c, 0, sizeof(*c));
  c->device = device;
  c->stream = (hipStream_t)stream;
  c->num_cus = prop.multiProcessorCount;
  int rc = context_allocate(c);
  if (rc != MFEM_OK) {  // release what the failed creation had got (hipFree / hipHostFree / hipEventDestroy of null are no-ops here)
    context_release(c);
    delete c;
    return rc;
  }
  {  // registered only once it is complete: a failed creation leaves nothing behind in the registry
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    g_live_contexts.insert(c);
  }
  *out = c;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_context_create")

extern "C" int mfem_context_set_stream(mfem_context ctx, void* stream) try {
  MFEM_REQUIRE(ctx != nullptr, "ctx is null");
  ctx->stream = (hipStream_t)stream;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_context_set_stream")

extern "C" int mfem_context_sync(mfem_context ctx) try {
  MFEM_REQUIRE(ctx != nullptr, "ctx is null");
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return MFEM_OK;
} MFEM_API_CATCH("mfem_context_sync")

extern "C" int mfem_context_destroy(mfem_context ctx) try {
  if (!ctx) return MFEM_OK;
  {
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    g_live_contexts.erase(ctx);
  }
  hipStreamSynchronize(ctx->stream);
  context_release(ctx);
  delete ctx;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_context_destroy")

int mfem_prof_flush(mfem_context_s* ctx) {
  if (ctx->prof_used == 0) return MFEM_OK;
  MFEM_CHECK_HIP(hipEventSynchronize(ctx->prof_ev[2 * ctx->prof_used - 1]));
  for (int k = 0; k < ctx->prof_used; ++k) {
    float ms = 0.f;
    MFEM_CHECK_HIP(hipEventElapsedTime(&ms, ctx->prof_ev[2 * k], ctx->prof_ev[2 * k + 1]));
    ctx->prof_ms += ms;
  }
  ctx->prof_count += ctx->prof_used;
  ctx->prof_used = 0;
  return MFEM_OK;
}

extern "C" int mfem_prof_spmv_enable(mfem_context ctx, int on) try {
  MFEM_REQUIRE(ctx, "null ctx");
  if (on && !ctx->prof_ev) {
    ctx->prof_ev = new hipEvent_t[2 * MFEM_PROF_PAIRS]();
    for (int i = 0; i < 2 * MFEM_PROF_PAIRS; ++i) MFEM_CHECK_HIP(hipEventCreate(&ctx->prof_ev[i]));
  }
  if (!on) {
    int rc = mfem_prof_flush(ctx);
    if (rc) return rc;
  }
  ctx->prof_on = on ? 1 : 0;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_prof_spmv_enable")

extern "C" int mfem_prof_spmv_read(mfem_context ctx, double* total_ms, int64_t* launches, int reset) try {
  MFEM_REQUIRE(ctx && total_ms && launches, "null argument");
  int rc = mfem_prof_flush(ctx);
  if (rc) return rc;
  *total_ms = ctx->prof_ms;
  *launches = ctx->prof_count;
  if (reset) {
    ctx->prof_ms = 0.0;
    ctx->prof_count = 0;
  }
  return MFEM_OK;
} MFEM_API_CATCH("mfem_prof_spmv_read")

uint64_t mfem_next_csr_serial() {
  static std::atomic<uint64_t> next{1};
  return next.fetch_add(1);
}

void mfem_graphs_invalidate(mfem_context_s* ctx) {
  for (int i = 0; i < MFEM_GRAPH_SLOTS; ++i)
    if (ctx->graph_exec[i]) {
      hipGraphExecDestroy(ctx->graph_exec[i]);
      ctx->graph_exec[i] = nullptr;
      ctx->graph_key[i] = 0;
    }
}

static std::atomic<size_t> g_ws_align{0}, g_ws_offset{0};  // placement experiment (mfem_debug_set_ws_placement): base = align_up(raw, align) + offset
extern "C" int mfem_debug_set_ws_placement(long long align, long long offset) try {
  g_ws_align = align > 0 ? (size_t)align : 0;
  g_ws_offset = offset > 0 ? (size_t)offset : 0;
  return MFEM_OK;
} MFEM_API_CATCH("mfem_debug_set_ws_placement")
// (Round 4 tried the virtual-memory API for the workspace -- hipMemAddressReserve + hipMemCreate / hipMemMap in one, 1 GiB or 2 MiB physical chunks:
// the same two SpMV speeds, and one run faulted on a re-mapped range; profiles/r04_placement_counters.txt.  hipMalloc stays.)
static int ws_alloc(mfem_context_s* ctx, void** raw, size_t bytes) {
  (void)ctx;
  *raw = nullptr;
  MFEM_CHECK_HIP(hipMalloc(raw, bytes));
  return MFEM_OK;
}
static int ws_free(void* raw) {
  if (raw) MFEM_CHECK_HIP(hipFree(raw));
  return MFEM_OK;
}
int mfem_ws_release(void* raw) { return ws_free(raw); }
extern "C" unsigned long long mfem_debug_ws_address(mfem_context ctx) { return ctx ? (unsigned long long)(uintptr_t)ctx->ws : 0ull; }
// Next candidate for the workspace (same size, same placement rule).  First call (no alternative held): the current one moves to ws_alt*.  Later
// calls: the current one is freed (the alternative stays) -- at most two are alive.  ws_try counts the candidates allocated after the first; it is
// unchanged when memory did not allow another one.
#include <chrono>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int mfem_ws_next_candidate(mfem_context_s* ctx) {
  const bool verbose = getenv("MFEM_WS_TRIAL_VERBOSE") != nullptr;
  double t0 = now_ms();
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (verbose) fprintf(stderr, "ws trial: sync %.1f ms\n", now_ms() - t0);
  const size_t need = ctx->ws_bytes + g_ws_align + g_ws_offset;
  if (ctx->ws_alt_raw) {  // drop the current candidate first
    t0 = now_ms();
    {
      int rcf = ws_free(ctx->ws_raw);
      if (rcf) return rcf;
    }
    if (verbose) fprintf(stderr, "ws trial: free %.1f ms\n", now_ms() - t0);
    ctx->ws = ctx->ws_raw = nullptr;
  }
  size_t freeb = 0, totalb = 0;
  void* raw = nullptr;
  bool ok = hipMemGetInfo(&freeb, &totalb) == hipSuccess && freeb >= need + ((size_t)4 << 30);
  t0 = now_ms();
  if (ok && ws_alloc(ctx, &raw, need) != MFEM_OK) ok = false;
  if (verbose) fprintf(stderr, "ws trial: alloc %.1f ms\n", now_ms() - t0);
  if (!ok) {
    (void)hipGetLastError();
    if (!ctx->ws_raw) {  // the current one is gone: back to the alternative
      ctx->ws = ctx->ws_alt;
      ctx->ws_raw = ctx->ws_alt_raw;
      ctx->ws_alt = ctx->ws_alt_raw = nullptr;
    }
    return MFEM_OK;
  }
  if (ctx->ws_raw) {
    ctx->ws_alt = ctx->ws;
    ctx->ws_alt_raw = ctx->ws_raw;
  }
  ctx->ws_raw = raw;
  uintptr_t p = (uintptr_t)raw;
  if (g_ws_align) p = (p + g_ws_align - 1) / g_ws_align * g_ws_align;
  ctx->ws = (void*)(p + g_ws_offset);
  ctx->ws_try += 1;
  return MFEM_OK;
}
// end of the trial: keep_current = false goes back to the first allocation
int mfem_ws_decide(mfem_context_s* ctx, bool keep_current) {
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->ws_alt_raw) {
    if (keep_current) {
      int rcf = ws_free(ctx->ws_alt_raw);
      if (rcf) return rcf;
    } else {
      int rcf = ws_free(ctx->ws_raw);
      if (rcf) return rcf;
      ctx->ws = ctx->ws_alt;
      ctx->ws_raw = ctx->ws_alt_raw;
    }
  }
  ctx->ws_alt = ctx->ws_alt_raw = nullptr;
  ctx->ws_try = 99;  // decided
  return MFEM_OK;
}
int mfem_ws_reserve(mfem_context_s* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return MFEM_OK;
  MFEM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  int rcf = ws_free(ctx->ws_raw);
  if (!rcf) rcf = ws_free(ctx->ws_alt_raw);
  ctx->ws = ctx->ws_raw = ctx->ws_alt = ctx->ws_alt_raw = nullptr;
  ctx->ws_try = 0;  // (a new allocation: undecided again)
  ctx->ws_bytes = 0;  // (BEFORE the early return: a failed free must not leave a size behind that lets a later, smaller request succeed on a null workspace)
  if (rcf) return rcf;
  rcf = ws_alloc(ctx, &ctx->ws_raw, bytes + g_ws_align + g_ws_offset);
  if (rcf) return rcf;
  uintptr_t p = (uintptr_t)ctx->ws_raw;
  if (g_ws_align) p = (p + g_ws_align - 1) / g_ws_align * g_ws_align;
  ctx->ws = (void*)(p + g_ws_offset);
  ctx->ws_bytes = bytes;
  return MFEM_OK;
}

// ---- the ONE entry point of the tuning knobs (round 6: 22 mfem_debug_set_* prototypes left include/metafem_mi355x_debug.h; the functions stay in the
// library, undeclared, and are reached through this table -- keys and the meaning of (a, b): the header)
extern "C" {
int mfem_debug_set_spmv(int, int);
int mfem_debug_set_ell(int);
int mfem_debug_set_sell(int);
int mfem_debug_set_lat27(int);
int mfem_debug_set_lat8(int);
int mfem_debug_set_layout_min_rows(int64_t, int64_t);
int mfem_debug_set_graphs(int, int64_t);
int mfem_debug_set_idrs(int);
int mfem_debug_set_bicgstabl(int);
int mfem_debug_set_vec_grid(int);
int mfem_debug_set_halo_overlap(int);
int mfem_debug_set_hex27(int);
int mfem_debug_set_elasticity(int);
int mfem_debug_set_hex8_thermal(int);
int mfem_debug_set_ws_placement(long long, long long);
int mfem_debug_set_ws_trial(int);
int mfem_debug_set_cg_streaming(int);
int mfem_debug_set_cg_single_max_rows(int64_t);
int mfem_debug_set_csr_strips(int, int64_t);
int mfem_debug_set_remainder(int);
int mfem_debug_set_recheck_scale(double);
int mfem_debug_set_bsell(int);
}
extern int g_mesh_gather_rows, g_mesh_abl, g_mesh_stage_min_itp, g_mesh_term_matrix;  // assemble_mesh.hip
extern int g_op_wave_forms, g_op_wave_min_itp;  // ops.hip
extern "C" int mfem_debug_set(const char* key, int64_t a, int64_t b) try {
  MFEM_REQUIRE(key, "null key");
  const std::string k(key);
  if (k == "spmv") return mfem_debug_set_spmv((int)a, (int)b);
  if (k == "ell") return mfem_debug_set_ell((int)a);
  if (k == "sell") return mfem_debug_set_sell((int)a);
  if (k == "lat27") return mfem_debug_set_lat27((int)a);
  if (k == "lat8") return mfem_debug_set_lat8((int)a);
  if (k == "layout_min_rows") return mfem_debug_set_layout_min_rows(a, b);
  if (k == "graphs") return mfem_debug_set_graphs((int)a, b);
  if (k == "idrs") return mfem_debug_set_idrs((int)a);
  if (k == "bicgstabl") return mfem_debug_set_bicgstabl((int)a);
  if (k == "vec_grid") return mfem_debug_set_vec_grid((int)a);
  if (k == "halo_overlap") return mfem_debug_set_halo_overlap((int)a);
  if (k == "hex27") return mfem_debug_set_hex27((int)a);
  if (k == "elasticity") return mfem_debug_set_elasticity((int)a);
  if (k == "hex8_thermal") return mfem_debug_set_hex8_thermal((int)a);
  if (k == "ws_placement") return mfem_debug_set_ws_placement((long long)a, (long long)b);
  if (k == "ws_trial") return mfem_debug_set_ws_trial((int)a);
  if (k == "cg_streaming") return mfem_debug_set_cg_streaming((int)a);
  if (k == "cg_single_max_rows") return mfem_debug_set_cg_single_max_rows(a);
  if (k == "csr_strips") return mfem_debug_set_csr_strips((int)a, b);
  if (k == "remainder") return mfem_debug_set_remainder((int)a);
  if (k == "mesh_gather_rows") { g_mesh_gather_rows = (int)a; return MFEM_OK; }
  if (k == "bsell") return mfem_debug_set_bsell((int)a);
  if (k == "op_wave_forms") { g_op_wave_forms = (int)a; g_op_wave_min_itp = b > 0 ? (int)b : 10; return MFEM_OK; }
  if (k == "mesh_stage_min_itp") { g_mesh_stage_min_itp = a > 0 ? (int)a : 16; return MFEM_OK; }
  if (k == "mesh_term_matrix") { g_mesh_term_matrix = (int)a; return MFEM_OK; }
  if (k == "mesh_abl") { g_mesh_abl = (int)a; return MFEM_OK; }
  if (k == "recheck_scale_ppm") return mfem_debug_set_recheck_scale((double)a * 1e-6);
  mfem_set_error("mfem_debug_set: unknown key '%s'", key);
  return MFEM_ERR_INVALID;
} MFEM_API_CATCH("mfem_debug_set")
