// pdbeda_hip.hip -- host side of libpdbeda_hip.so: contexts, device arenas, job setup and the
// extern "C" entry points declared in include/pdbeda.h.  gfx950 only, no CPU fallback.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cerrno>

#include <algorithm>
#include <chrono>
#include <thread>
#include <atomic>
#include <random>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "pdbeda_kernels.h"
#include "pdbeda_tile.h"
#include "pdbeda_upload.h"

using namespace pdbeda;

// ------------------------------------------------------------------------------------
// Handles
// ------------------------------------------------------------------------------------
struct Arena {
    char *base = nullptr;
    size_t cap = 0;
};

struct pdbeda_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    std::multimap<size_t, Arena> pool;  // free device arenas by capacity (reused: no hipMalloc in steady state)
    size_t pool_bytes = 0;              // bytes parked in the pool ...
    size_t pool_cap = (size_t)24 << 30; // ... trimmed (largest first) beyond this (PDBEDA_POOL_CAP_MB): streams x live lists x multi-GB
                                        // whole-map arenas must not creep up on the 288 GB until hipMalloc fails
    double *partials = nullptr;         // reduction partials (N_PARTIAL doubles) + 4 result slots
    char *dev_stage = nullptr;          // device staging block of d2h_many (results packed by k_pack, one copy to the host); handed out until the next ctx_sync
    size_t dev_stage_cap = 0, dev_stage_used = 0;
    // device -> host results are staged through pinned memory: the copies are truly asynchronous (a copy into pageable
    // memory blocks inside the runtime, where no watchdog can see it) and land in the caller's buffers when ctx_sync()
    // has seen the stream drain
    char *pinned = nullptr;
    size_t pinned_cap = 0, pinned_used = 0;
    struct Pending { void *dst; size_t off, bytes; };
    std::vector<Pending> pending;
    // file -> device uploads (pdbeda_map_upload_file): the chunks of a map go through the process's upload engine (reader threads
    // with pinned chunks and copy streams of their own); what a context keeps is one event per reader, recorded
    // behind the reader's copies of the map and waited for by the context's stream
    static constexpr int MAX_READERS = 8;
    hipEvent_t reader_ev[MAX_READERS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int live_handles = 0;
    // optional per-kernel timing with HIP events on ctx->stream (bench.py's roofline leg)
    bool profiling = false;
    struct ProfRec { const char *name; hipEvent_t a, b; };
    std::vector<ProfRec> prof;
    // test hooks (environment, read once at creation): PDBEDA_DEBUG_POISON=1 fills every arena with 0xFF bytes when it
    // is handed out (a kernel that trusts recycled memory shows up at once); PDBEDA_DEBUG_EDGE_CAP=n shrinks the
    // cross-tile pair buffer so the shard-overflow path runs on small inputs.
    bool debug_poison = false;
    bool debug_shrink_totals = false;   // PDBEDA_DEBUG_SHRINK_TOTALS=1: the host sizes per-atom sphere batches for HALF their mask words -- what k_make_vols' check of the device's totals is there to catch (tests)
    int64_t debug_edge_cap = 0;
    bool debug_worst_case_arena = false;   // PDBEDA_DEBUG_WORST_CASE_ARENA=1: whole-map jobs are carved for the worst case at once (no second run)
    // per-entry watchdog (multipleStructures.py:359-377 wraps every entry in a SIGALRM timeout; threads cannot): when
    // timeout_s > 0 every wait on the stream is a timed hipStreamQuery loop; a wait that expires marks the context
    // abandoned: every later call fails at once with PDBEDA_ERR_TIMEOUT and destroy does not wait for the stream.
    double timeout_s = 0.0;
    std::chrono::steady_clock::time_point deadline;   // ONE deadline for every wait of the entry: set when the watchdog is (re-)armed
    bool timed_out = false;
    // the pool is touched by its own context's thread, and -- when another context of the process runs out of device memory --
    // by that context's thread, which trims its siblings' pools before it gives up
    std::mutex pool_mu;
    std::map<char *, size_t> lent;      // arenas handed out and not yet returned (maps, jobs): an abandoned context frees them too
};

// Contexts of this process: live ones (a context short of memory trims the arena pools of its siblings) and the ones the
// watchdog abandoned.  An abandoned context cannot be destroyed at once -- its stream may still run, hipFree would hang with
// it -- but leaking its arenas for the life of the process is not an option either: the entries that time out are the large
// ones.  So it is parked here and REAPED (normal destroy path, including the arenas its lost handles still hold) as soon as a
// query finds its stream drained: at every context creation and whenever an allocation fails.
static std::mutex g_ctx_mu;
static std::vector<pdbeda_ctx *> g_live, g_abandoned;

static void ctx_release_device(pdbeda_ctx *ctx, bool lent_too) {
    for (auto &kv : ctx->pool) (void)hipFree(kv.second.base);
    ctx->pool.clear();
    ctx->pool_bytes = 0;
    if (lent_too) {
        for (auto &kv : ctx->lent) (void)hipFree(kv.first);
        ctx->lent.clear();
    }
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->dev_stage) (void)hipFree(ctx->dev_stage);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (int k = 0; k < pdbeda_ctx::MAX_READERS; ++k)
        if (ctx->reader_ev[k]) (void)hipEventDestroy(ctx->reader_ev[k]);
    for (auto &r : ctx->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
}

// Destroy the abandoned contexts whose stream has drained meanwhile; returns how many are still parked.
static size_t reap_abandoned() {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    if (g_abandoned.empty()) return 0;
    int caller_device = -1;
    if (hipGetDevice(&caller_device) != hipSuccess) caller_device = -1;   // (the caller's current device is put back below)
    for (size_t i = 0; i < g_abandoned.size();) {
        pdbeda_ctx *ctx = g_abandoned[i];
        (void)hipSetDevice(ctx->device);
        // still running (or hung) on its stream, or the upload engine still copies into one of its maps: try again later
        bool busy = hipStreamQuery(ctx->stream) == hipErrorNotReady;
        for (int k = 0; k < pdbeda_ctx::MAX_READERS && !busy; ++k)      // (the readers' copies of its last map: behind these events)
            if (ctx->reader_ev[k] && hipEventQuery(ctx->reader_ev[k]) == hipErrorNotReady) busy = true;
        if (busy) { ++i; continue; }
        ctx_release_device(ctx, true);
        delete ctx;
        g_abandoned.erase(g_abandoned.begin() + i);
    }
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    return g_abandoned.size();
}

// Give the parked arenas of the OTHER live contexts of this device back to the driver (the caller is out of device memory).
static void trim_sibling_pools(pdbeda_ctx *self) {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    for (pdbeda_ctx *other : g_live) {
        if (other == self || other->device != self->device) continue;
        std::lock_guard<std::mutex> p(other->pool_mu);
        for (auto &kv : other->pool) (void)hipFree(kv.second.base);
        other->pool.clear();
        other->pool_bytes = 0;
    }
}

struct ProfScope {
    pdbeda_ctx *ctx;
    hipEvent_t b = nullptr;
    ProfScope(pdbeda_ctx *c, const char *name) : ctx(c) {
        if (!ctx->profiling) return;
        hipEvent_t a = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { b = nullptr; return; }
        (void)hipEventRecord(a, ctx->stream);
        ctx->prof.push_back({name, a, b});
    }
    ~ProfScope() { if (b) (void)hipEventRecord(b, ctx->stream); }
};
#define PROF(ctx, name) ProfScope prof_scope_##__LINE__(ctx, name)

struct pdbeda_map {
    pdbeda_ctx *ctx = nullptr;
    const float *dens = nullptr;
    bool own_dens = false;
    Arena arena;                      // geometry (+ the grid when the library owns it): from the context's arena pool, recycled in
                                      // stream order -- a stream pool analysing entry after entry never calls hipMalloc / hipFree
    Geom geom;
    Geom *geom_dev = nullptr;
    int64_t n_vox = 0;
    double fix_mul = 0.0;             // 2^S of the order-independent blob sums (FixSums); 0 = not yet derived from the map's range
    bool fix_refused = false;         // the range pass found a NaN / infinity: labelling calls refuse the map (map_fix_mul)
};

struct pdbeda_bloblist {
    std::vector<AtomBox> host_boxes;   // a per-atom sphere batch whose boxes the host made: box of atom a (else empty)
    pdbeda_ctx *ctx = nullptr;
    pdbeda_map *map = nullptr;
    Arena arena;
    Job job;
    int vol_lo = 0, vol_hi = 0;      // volumes of the job that belong to this list
    bool owns_arena = true;          // the "red" list of a fused call shares the green list's arena
    pdbeda_bloblist *sibling = nullptr;
    bool freed = false;
    // resolved lazily
    bool have_counts = false;
    int64_t rank_lo = 0, rank_hi = 0;  // blob rank range of this list inside the job's table
    int64_t job_blobs = -1;            // blobs of the whole job (known once the counters have been read)
    int64_t n_voxels = -1;
    int32_t *labels_dev = nullptr;     // inside arena when requested
    bool labels_done = false;
    int64_t *offsets_dev = nullptr;
    int32_t *crs_dev = nullptr;
    unsigned int *cursor_dev = nullptr;
    bool voxels_done = false;
    Arena vox_arena;
    bool whole_map = false;
    TileDims td;
    int sign = 1;
    // whole-map jobs: what it takes to run the job again (a typical-size arena that turned out too small: see whole_map_enqueue)
    float cut_pos = 0.0f, cut_neg = 0.0f;
    bool want_pos = false, want_neg = false;
    uint32_t flags = 0;
    int tier = 0, unit_form = 0, reruns = 0;
    size_t job_bytes = 0;              // bytes the job carved out of its arena (a recycled arena may be larger)
    // whole-map jobs, on the list that owns the arena: the first rows of the job's blob table, fetched WITH the counters (round 5: the count and the table of
    // each of the two lists of a fused call were four waits; now one serves both)
    int64_t spec_rows = 0;
    std::vector<int64_t> spec_n, spec_key;
    std::vector<double> spec_total, spec_centroid, spec_center, spec_volume;
    std::vector<int32_t> spec_group;
};

static const int N_PARTIAL = 2048;

static int fail(pdbeda_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    // results of a failed call are never delivered.  The staging regions are NOT handed out again here: copies queued by an
    // earlier call (a sphere batch's staged inputs) may still read them -- only ctx_sync(), which has drained the stream, rewinds
    if (ctx) ctx->pending.clear();
    if (ctx && ctx->timed_out) return PDBEDA_ERR_TIMEOUT;   // (the watchdog's message stays)
    if (ctx) ctx->err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, PDBEDA_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Wait for the context's stream: plain hipStreamSynchronize, or (watchdog armed) a timed query loop; then deliver the
// staged device -> host results.
// (round 6, measured and dropped: polling the stream for the first 300 us of a wait without a watchdog instead of hipStreamSynchronize at once --
//  the analysis entry 3.07 against 3.02 ms, the pools inside their noise: the runtime's own wait is not where an entry's waits lose time)
static hipError_t ctx_wait(pdbeda_ctx *ctx) {
    if (ctx->timed_out) return hipErrorNotReady;
    if (ctx->timeout_s <= 0.0) return hipStreamSynchronize(ctx->stream);
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q != hipErrorNotReady) return q;
        if (spins > 256) {   // first ~0.3 ms: busy poll (the usual wait is tens of microseconds), then back off
            if (std::chrono::steady_clock::now() > ctx->deadline) {   // the entry's ONE deadline: an entry makes dozens of waits
                ctx->timed_out = true;
                ctx->err = "watchdog: the stream did not drain within the per-entry time-out; context abandoned";
                return hipErrorNotReady;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(spins > 4096 ? 200 : 20));
        }
    }
}

// The small staged copies of a call (inputs through the pinned block, results into it) are KERNELS that read / write the pinned block over the
// link, not hipMemcpyAsync (round 5).  A copy of a few hundred bytes is a packet on the SDMA engine, where it waits behind every 4-8 MiB chunk the
// upload engine has queued (up to 0.46 ms), and each hop between the compute queue and the copy engine is a semaphore: the analysis of a resident
// entry 0.48 -> 0.44 ms alone, 0.68-0.80 -> 0.60 beside two uploading threads; four workers 41.6 k -> 44.5 k entries/min.  PDBEDA_COPY_KERNELS=0: the
// runtime's copies (A/B).
static bool copy_kernels() { static const bool v = [] { const char *e = getenv("PDBEDA_COPY_KERNELS"); return !(e && e[0] == '0'); }(); return v; }
static hipError_t copy_by_kernel(pdbeda_ctx *ctx, void *dst, const void *src, size_t bytes) {
    hipLaunchKernelGGL(k_copy_bytes, dim3((unsigned)std::min<size_t>((bytes >> 12) + 1, 64)), dim3(256), 0, ctx->stream, (const unsigned char *)src, (unsigned char *)dst,
                       (unsigned long long)bytes);
    return hipGetLastError();
}

static hipError_t ctx_sync(pdbeda_ctx *ctx) {
    const hipError_t e = ctx_wait(ctx);
    if (e == hipSuccess)
        for (const auto &p : ctx->pending) memcpy(p.dst, ctx->pinned + p.off, p.bytes);
    ctx->pending.clear();
    ctx->pinned_used = 0;
    ctx->dev_stage_used = 0;
    return e;
}

// Device -> host copy of a result, complete after the next ctx_sync(): staged through the pinned buffer when it fits,
// otherwise the stream is drained first (timed) so that the blocking copy into pageable memory has nothing to wait for.
static hipError_t d2h(pdbeda_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (ctx->pinned && ctx->pinned_used + need <= ctx->pinned_cap) {
        const hipError_t e = copy_kernels() ? copy_by_kernel(ctx, ctx->pinned + ctx->pinned_used, src, bytes)
                                            : hipMemcpyAsync(ctx->pinned + ctx->pinned_used, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) { ctx->pending.push_back({dst, ctx->pinned_used, bytes}); ctx->pinned_used += need; }
        return e;
    }
    hipError_t e = ctx_sync(ctx);
    if (e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    return e;
}

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

static int arena_get_raw(pdbeda_ctx *ctx, size_t bytes, Arena *out);

static int arena_get(pdbeda_ctx *ctx, size_t bytes, Arena *out) {
    int rc = arena_get_raw(ctx, bytes, out);
    if (rc == 0 && ctx->debug_poison) HIP_TRY(ctx, hipMemsetAsync(out->base, 0xFF, out->cap, ctx->stream));
    return rc;
}

static int arena_get_raw(pdbeda_ctx *ctx, size_t bytes, Arena *out) {
    if (ctx->timed_out) return PDBEDA_ERR_TIMEOUT;
    bytes = align_up(std::max<size_t>(bytes, 256));
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        auto it = ctx->pool.lower_bound(bytes);
        if (it != ctx->pool.end() && it->first <= bytes * 2 + (1u << 20)) {
            *out = it->second;
            ctx->pool_bytes -= it->second.cap;
            ctx->pool.erase(it);
            ctx->lent[out->base] = out->cap;
            return 0;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        {   // drop the cache and retry
            std::lock_guard<std::mutex> g(ctx->pool_mu);
            for (auto &kv : ctx->pool) (void)hipFree(kv.second.base);
            ctx->pool.clear();
            ctx->pool_bytes = 0;
        }
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {   // then what abandoned contexts still hold, then what the sibling contexts have parked
            reap_abandoned();
            trim_sibling_pools(ctx);
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) return fail(ctx, PDBEDA_ERR_MEMORY, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    out->base = (char *)p;
    out->cap = bytes;
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    ctx->lent[out->base] = out->cap;
    return 0;
}

static void arena_put(pdbeda_ctx *ctx, Arena &a) {
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    if (a.base) { ctx->lent.erase(a.base); ctx->pool.emplace(a.cap, a); ctx->pool_bytes += a.cap; }
    a.base = nullptr;
    a.cap = 0;
    while (ctx->pool_bytes > ctx->pool_cap && !ctx->pool.empty() && !ctx->timed_out) {   // trim: the largest parked arena goes back to the
        auto last = std::prev(ctx->pool.end());                                            // driver (hipFree waits for the device: rare by design)
        ctx->pool_bytes -= last->second.cap;
        (void)hipFree(last->second.base);
        ctx->pool.erase(last);
    }
}

struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(char *b) : base(b) {}
    template <typename T> T *take(size_t n) {
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += align_up(n * sizeof(T));
        return p;
    }
};

static inline unsigned grid_for(int64_t n, int block, int64_t cap = 1 << 20) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// Several small inputs that sit in a row in device scratch (consecutive Carver takes), in ONE copy: laid out in the pinned
// staging block as on the device; one by one from the caller's memory when the block has no room (the caller waits before it
// returns either way).
struct H2DItem { void *dst; const void *src; size_t bytes; };
static hipError_t h2d_row(pdbeda_ctx *ctx, const H2DItem *items, int n) {
    char *lo = nullptr, *hi = nullptr;
    for (int k = 0; k < n; ++k) {
        if (items[k].bytes == 0) continue;
        char *d = (char *)items[k].dst;
        if (!lo || d < lo) lo = d;
        if (!hi || d + items[k].bytes > hi) hi = d + items[k].bytes;
    }
    if (!lo) return hipSuccess;
    const size_t span = (size_t)(hi - lo), need = (span + 63) & ~(size_t)63;
    if (ctx->pinned && span <= (256u << 10) && ctx->pinned_used + need <= ctx->pinned_cap) {
        char *stage = ctx->pinned + ctx->pinned_used;
        for (int k = 0; k < n; ++k)
            if (items[k].bytes) memcpy(stage + ((char *)items[k].dst - lo), items[k].src, items[k].bytes);
        const hipError_t e = copy_kernels() ? copy_by_kernel(ctx, lo, stage, span) : hipMemcpyAsync(lo, stage, span, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) ctx->pinned_used += need;
        return e;
    }
    for (int k = 0; k < n; ++k) {
        if (items[k].bytes == 0) continue;
        const hipError_t e = hipMemcpyAsync(items[k].dst, items[k].src, items[k].bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static hipError_t h2d_one(pdbeda_ctx *ctx, void *dst, const void *src, size_t bytes) {
    const H2DItem in[1] = {{dst, src, bytes}};
    return h2d_row(ctx, in, 1);
}

// Several results of one call, complete after the next ctx_sync(): packed on the device (k_pack) and brought over in ONE copy
// when they fit the staging blocks, else one by one (d2h).  Items with a null destination or no bytes are skipped.
struct D2HItem { void *dst; const void *src; size_t bytes; };
static hipError_t d2h_many(pdbeda_ctx *ctx, const D2HItem *items, int n_items) {
    int live[8], n = 0;
    size_t total = 0;
    bool packable = (ctx->dev_stage != nullptr || copy_kernels()) && ctx->pinned != nullptr;
    for (int k = 0; k < n_items; ++k) {
        if (!items[k].dst || items[k].bytes == 0) continue;
        if (n == 8 || (items[k].bytes & 3u) || ((uintptr_t)items[k].src & 3u)) { packable = false; break; }
        live[n++] = k;
        total += (items[k].bytes + 63) & ~(size_t)63;
    }
    const bool via_dev_stage = !copy_kernels();      // (with copy kernels k_pack writes straight into the pinned block: nothing is staged on the device)
    if (packable && n >= 2 && (!via_dev_stage || ctx->dev_stage_used + total <= ctx->dev_stage_cap) && ctx->pinned_used + total <= ctx->pinned_cap) {
        PackArgs a;
        memset(&a, 0, sizeof a);
        a.n = n;
        size_t off = 0, words = 0;
        for (int j = 0; j < n; ++j) {
            const D2HItem &it = items[live[j]];
            a.seg[j].src = reinterpret_cast<const uint32_t *>(it.src);
            a.seg[j].words = it.bytes / 4;
            a.seg[j].dst_word = off / 4;
            ctx->pending.push_back({it.dst, ctx->pinned_used + off, it.bytes});
            off += (it.bytes + 63) & ~(size_t)63;
            words = std::max<size_t>(words, it.bytes / 4);
        }
        char *block = copy_kernels() ? ctx->pinned + ctx->pinned_used : ctx->dev_stage + ctx->dev_stage_used;   // (packed straight into the pinned block)
        hipLaunchKernelGGL(k_pack, dim3(grid_for((int64_t)words, 256, 256)), dim3(256), 0, ctx->stream, a, reinterpret_cast<uint32_t *>(block));
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && !copy_kernels()) e = hipMemcpyAsync(ctx->pinned + ctx->pinned_used, block, total, hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) { for (int j = 0; j < n; ++j) ctx->pending.pop_back(); return e; }
        if (via_dev_stage) ctx->dev_stage_used += total;
        ctx->pinned_used += total;
        return hipSuccess;
    }
    for (int k = 0; k < n_items; ++k) {
        if (!items[k].dst || items[k].bytes == 0) continue;
        const hipError_t e = d2h(ctx, items[k].dst, items[k].src, items[k].bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------------
// Small batched helpers: stage host arrays through a scratch arena
// ------------------------------------------------------------------------------------
// Round 6: small results and single-touch inputs without copy launches.  pinned_out: a result array that a kernel writes STRAIGHT into the context's pinned
// block (delivered to `dst` by the next ctx_sync); pinned_in: an input that every thread reads once, staged in the pinned block and read by the kernel over the
// link.  nullptr: no room, or PDBEDA_COPY_KERNELS=0 (kernels do not touch host memory then) -- the caller stages on the device and copies as before.
// (An input that threads read many times -- the atoms of a nearest-atom search -- belongs in device memory: a copy launch is cheaper than the link.)
template <typename T>
static T *pinned_out(pdbeda_ctx *ctx, T *dst, size_t count) {
    const size_t bytes = sizeof(T) * count, need = (bytes + 63) & ~(size_t)63;
    if (!copy_kernels() || !ctx->pinned || !dst || bytes == 0 || ctx->pinned_used + need > ctx->pinned_cap) return nullptr;
    T *p = reinterpret_cast<T *>(ctx->pinned + ctx->pinned_used);
    ctx->pending.push_back({dst, ctx->pinned_used, bytes});
    ctx->pinned_used += need;
    return p;
}
template <typename T>
static const T *pinned_in(pdbeda_ctx *ctx, const T *src, size_t count) {
    const size_t bytes = sizeof(T) * count, need = (bytes + 63) & ~(size_t)63;
    if (!copy_kernels() || !ctx->pinned || !src || bytes == 0 || ctx->pinned_used + need > ctx->pinned_cap) return nullptr;
    T *p = reinterpret_cast<T *>(ctx->pinned + ctx->pinned_used);
    memcpy(p, src, bytes);
    ctx->pinned_used += need;
    return p;
}

template <typename Fn>
static int with_scratch(pdbeda_ctx *ctx, size_t bytes, Fn fn) {
    Arena a;
    int rc = arena_get(ctx, bytes, &a);
    if (rc) return rc;
    rc = fn(a.base);
    hipError_t e = ctx_sync(ctx);
    arena_put(ctx, a);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, PDBEDA_ERR_DEVICE, "stream sync: %s", hipGetErrorString(e));
    return 0;
}

// ------------------------------------------------------------------------------------
// Library / context
// ------------------------------------------------------------------------------------
extern "C" const char *pdbeda_version(void) { return "pdbeda-hip 0.1 (gfx950)"; }

extern "C" int pdbeda_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int pdbeda_device_pci_address(int device_id, char *out, int out_len) {
    if (!out || out_len < 16) return PDBEDA_ERR_ARGUMENT;
    out[0] = 0;
    if (hipDeviceGetPCIBusId(out, out_len, device_id) != hipSuccess) { (void)hipGetLastError(); return PDBEDA_ERR_DEVICE; }
    return PDBEDA_OK;
}

extern "C" int pdbeda_ctx_create_on_stream(int device_id, void *hip_stream, pdbeda_ctx **out) {
    if (!out) return PDBEDA_ERR_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return PDBEDA_ERR_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return PDBEDA_ERR_DEVICE;
    pdbeda_ctx *ctx = new pdbeda_ctx();
    ctx->device = device_id;
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        // (plain priority, like the upload engine's copy streams: r05 tried the contexts' streams above the copy streams -- with
        //  several PROCESSES on one GPU any mix of priorities took an entry from 1.4 to 9-15 ms)
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return PDBEDA_ERR_DEVICE; }
        ctx->own_stream = true;
    }
    if (hipMalloc((void **)&ctx->partials, sizeof(double) * (2 * N_PARTIAL + 8)) != hipSuccess) {
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return PDBEDA_ERR_MEMORY;
    }
    // (coherent + mapped, explicitly: with PDBEDA_COPY_KERNELS the staged results are WRITTEN BY KERNELS into this block and read by the host after the
    //  stream wait -- or after a hipStreamQuery poll under the watchdog --, which is only right for coherent host memory)
    if (hipHostMalloc((void **)&ctx->pinned, 4 << 20, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) ctx->pinned_cap = 4 << 20;   // (without it results are copied directly)
    else (void)hipGetLastError();
    if (ctx->pinned_cap && hipMalloc((void **)&ctx->dev_stage, 2 << 20) == hipSuccess) ctx->dev_stage_cap = 2 << 20; else { ctx->dev_stage = nullptr; (void)hipGetLastError(); }
    if (const char *v = getenv("PDBEDA_DEBUG_POISON")) ctx->debug_poison = v[0] && v[0] != '0';
    if (const char *v = getenv("PDBEDA_DEBUG_SHRINK_TOTALS")) ctx->debug_shrink_totals = v[0] && v[0] != '0';
    if (const char *v = getenv("PDBEDA_DEBUG_EDGE_CAP")) ctx->debug_edge_cap = atoll(v);
    if (const char *v = getenv("PDBEDA_DEBUG_WORST_CASE_ARENA")) ctx->debug_worst_case_arena = v[0] && v[0] != '0';
    if (const char *v = getenv("PDBEDA_POOL_CAP_MB")) ctx->pool_cap = (size_t)std::max<long long>(atoll(v), 0) << 20;
    reap_abandoned();   // (what the watchdog left behind earlier may have drained by now)
    {
        std::lock_guard<std::mutex> g(g_ctx_mu);
        g_live.push_back(ctx);
    }
    *out = ctx;
    return PDBEDA_OK;
}

extern "C" int pdbeda_ctx_create(int device_id, pdbeda_ctx **out) { return pdbeda_ctx_create_on_stream(device_id, nullptr, out); }

extern "C" int pdbeda_ctx_destroy(pdbeda_ctx *ctx) {
    if (!ctx) return PDBEDA_ERR_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    {
        std::lock_guard<std::mutex> g(g_ctx_mu);
        g_live.erase(std::remove(g_live.begin(), g_live.end(), ctx), g_live.end());
        // abandoned by the watchdog: its stream may never drain, and hipFree would hang with it -- parked, and reaped (with the
        // arenas of the maps / jobs its entry left behind) once a query finds the stream empty; leaked only if it never drains
        if (ctx->timed_out) g_abandoned.push_back(ctx);
    }
    if (ctx->timed_out) {
        reap_abandoned();
        return PDBEDA_OK;
    }
    (void)ctx_sync(ctx);
    ctx_release_device(ctx, false);   // (arenas still lent belong to live handles of the caller)
    delete ctx;
    return PDBEDA_OK;
}

extern "C" int pdbeda_ctx_synchronize(pdbeda_ctx *ctx) {
    if (!ctx) return PDBEDA_ERR_ARGUMENT;
    HIP_TRY(ctx, ctx_sync(ctx));
    return PDBEDA_OK;
}

extern "C" int pdbeda_ctx_profile_begin(pdbeda_ctx *ctx) {
    if (!ctx) return PDBEDA_ERR_ARGUMENT;
    for (auto &r : ctx->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    ctx->prof.clear();
    ctx->profiling = true;
    return PDBEDA_OK;
}

// Stops profiling, synchronises, and writes "name calls total_ms\n" lines into buf.
extern "C" int pdbeda_ctx_profile_end(pdbeda_ctx *ctx, char *buf, int64_t cap) {
    if (!ctx || !buf || cap <= 0) return PDBEDA_ERR_ARGUMENT;
    ctx->profiling = false;
    HIP_TRY(ctx, ctx_sync(ctx));
    std::map<std::string, std::pair<int64_t, double>> agg;
    for (auto &r : ctx->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { auto &e = agg[r.name]; e.first++; e.second += ms; }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    ctx->prof.clear();
    std::string out;
    char line[160];
    for (auto &kv : agg) {
        snprintf(line, sizeof line, "%s %lld %.6f\n", kv.first.c_str(), (long long)kv.second.first, kv.second.second);
        out += line;
    }
    if ((int64_t)out.size() + 1 > cap) return fail(ctx, PDBEDA_ERR_CAPACITY, "profile buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    return PDBEDA_OK;
}

extern "C" int pdbeda_ctx_set_timeout(pdbeda_ctx *ctx, double seconds) {
    if (!ctx || !(seconds >= 0.0)) return PDBEDA_ERR_ARGUMENT;
    ctx->timeout_s = seconds;
    ctx->deadline = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(seconds));
    return PDBEDA_OK;
}

extern "C" int64_t pdbeda_reap_abandoned(void) { return (int64_t)reap_abandoned(); }

extern "C" void *pdbeda_ctx_stream(pdbeda_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" const char *pdbeda_last_error(pdbeda_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// ------------------------------------------------------------------------------------
// Maps
// ------------------------------------------------------------------------------------
static int fill_geom(pdbeda_ctx *ctx, const pdbeda_geometry *in, Geom *g) {
    for (int k = 0; k < 3; ++k) {
        g->ncrs[k] = in->ncrs[k];
        g->crs_start[k] = in->crs_start[k];
        g->xyz_interval[k] = in->xyz_interval[k];
        g->map2xyz[k] = in->map2xyz[k];
        g->map2crs[k] = in->map2crs[k];
        g->origin[k] = in->origin[k];
        g->grid_len[k] = in->grid_len[k];
        if (in->ncrs[k] <= 0) return fail(ctx, PDBEDA_ERR_ARGUMENT, "ncrs[%d] = %d", k, in->ncrs[k]);
        if (in->xyz_interval[k] <= 0) return fail(ctx, PDBEDA_ERR_ARGUMENT, "xyz_interval[%d] = %d", k, in->xyz_interval[k]);
        if (in->map2xyz[k] < 0 || in->map2xyz[k] > 2 || in->map2crs[k] < 0 || in->map2crs[k] > 2)
            return fail(ctx, PDBEDA_ERR_ARGUMENT, "axis map out of range");
    }
    for (int k = 0; k < 9; ++k) { g->ortho[k] = in->ortho[k]; g->deortho[k] = in->deortho[k]; }
    g->orthogonal = in->orthogonal;
    g->unit_volume = in->unit_volume;
    for (int k = 0; k < 3; ++k) {
        g->crs_interval[k] = g->xyz_interval[g->map2crs[k]];                  // ccp4.py:237
        g->unique_ncrs[k] = std::min(g->ncrs[k], g->crs_interval[k]);         // ccp4.py:262-269
    }
    return 0;
}

static int stats_enqueue(pdbeda_map *m, double *chunk_sums, double host[2], double range[2], bool want_range, bool with_geom = false);
static bool stats_carries_geom(const pdbeda_map *m) { return m->n_vox / NP_CHUNK > 0; }      // (the chain's first launch exists: it brings the geometry along when asked to)
static void range_apply(pdbeda_map *m, const double range[2]);

static int map_create(pdbeda_ctx *ctx, const float *host, const float *dev, const pdbeda_geometry *geom, pdbeda_map **out, double *mean = nullptr, double *std = nullptr) {
    if (!ctx || !geom || !out || (!host && !dev)) return PDBEDA_ERR_ARGUMENT;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pdbeda_map *m = new pdbeda_map();
    m->ctx = ctx;
    int rc = fill_geom(ctx, geom, &m->geom);
    if (rc) { delete m; return rc; }
    m->n_vox = (int64_t)geom->ncrs[0] * geom->ncrs[1] * geom->ncrs[2];
    if (m->n_vox >= (1ll << 32)) { delete m; return fail(ctx, PDBEDA_ERR_ARGUMENT, "grids of 2^32 voxels or more are not supported"); }
    if (dev && ((uintptr_t)dev & 15u) != 0) { delete m; return fail(ctx, PDBEDA_ERR_ARGUMENT, "device density pointer must be 16-byte aligned"); }
    rc = arena_get(ctx, align_up(sizeof(Geom)) + (host ? align_up(sizeof(float) * (size_t)m->n_vox) : 0), &m->arena);
    if (rc) { delete m; return rc; }
    Carver cv(m->arena.base);
    m->geom_dev = cv.take<Geom>(1);
    hipError_t e = hipSuccess;
    if (host) {
        float *d = cv.take<float>((size_t)m->n_vox);
        m->dens = d;
        m->own_dens = true;
        // (measured, round 5: the same grid through the upload engine -- readers memcpy chunks into their pinned slots -- takes twice as long as the runtime's
        //  own copy from pageable memory, which pins the caller's pages and lets the engine read them in place: 0.6 against 0.3 ms for 8 MB)
        e = hipMemcpyAsync(d, host, sizeof(float) * (size_t)m->n_vox, hipMemcpyHostToDevice, ctx->stream);
    } else {
        m->dens = dev;
    }
    // (an uploaded map with its statistics from the same wait: see upload_file_impl; the geometry rides in the statistics' first launch then)
    const bool with_stats = (mean || std) && host && e == hipSuccess;
    const bool geom_in_stats = with_stats && stats_carries_geom(m);
    if (e == hipSuccess && !geom_in_stats) e = h2d_one(ctx, m->geom_dev, &m->geom, sizeof(Geom));
    double st_host[2] = {0.0, 0.0}, range[2] = {0.0, 0.0};
    Arena scratch;
    bool have_scratch = false;
    if (with_stats) {
        if (arena_get(ctx, 24 * (size_t)std::max<int64_t>(m->n_vox / NP_CHUNK, 1), &scratch) == 0) {
            have_scratch = true;
            if (stats_enqueue(m, reinterpret_cast<double *>(scratch.base), st_host, range, true, geom_in_stats) != 0) e = hipErrorUnknown;
        } else {
            e = hipErrorOutOfMemory;
        }
    }
    {
        const hipError_t e2 = ctx_sync(ctx);  // host buffers may be released on return
        if (e == hipSuccess) e = e2;
    }
    if (have_scratch) arena_put(ctx, scratch);
    if (e != hipSuccess) {
        arena_put(ctx, m->arena);
        delete m;
        return fail(ctx, PDBEDA_ERR_DEVICE, "map upload: %s", hipGetErrorString(e));
    }
    if (with_stats) {
        range_apply(m, range);
        if (mean) *mean = st_host[0];
        if (std) *std = st_host[1];
    }
    ctx->live_handles++;
    *out = m;
    return PDBEDA_OK;
}

extern "C" int pdbeda_map_upload(pdbeda_ctx *ctx, const float *density, const pdbeda_geometry *geom, pdbeda_map **out) {
    return map_create(ctx, density, nullptr, geom, out);
}
extern "C" int pdbeda_map_upload_stats(pdbeda_ctx *ctx, const float *density, const pdbeda_geometry *geom, pdbeda_map **out, double *mean, double *std) {
    return map_create(ctx, density, nullptr, geom, out, mean, std);
}
extern "C" int pdbeda_map_from_device(pdbeda_ctx *ctx, const float *density_dev, const pdbeda_geometry *geom, pdbeda_map **out) {
    return map_create(ctx, nullptr, density_dev, geom, out);
}

// Wait for an event of the context's stream with the watchdog's rules (see ctx_wait).
static hipError_t event_wait(pdbeda_ctx *ctx, hipEvent_t ev) {
    if (ctx->timed_out) return hipErrorNotReady;
    if (ctx->timeout_s <= 0.0) return hipEventSynchronize(ev);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        if (spins > 256) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ctx->timeout_s) {
                ctx->timed_out = true;
                ctx->err = "watchdog: the stream did not drain within the per-entry time-out; context abandoned";
                return hipErrorNotReady;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
}

// The grid of a CCP4 file straight into HBM: pread() fills pinned chunks while earlier ones are on their way over PCIe, and
// no host copy of the map is ever made (ccp4.py:77-127 unpacks it into a Python tuple).  The readers are the PROCESS's (UploadEngine,
// pdbeda_upload.h: three threads, two pinned chunks and a copy stream each); rounds 3-4 gave every context a ring and two readers of its own.
// A thread copies out of the page cache at 20-40 GB/s (by host), one copy engine moves 54 GB/s: one 32 MB map alone arrives in 0.76-0.85 ms,
// four loads in flight move 50-53 GB/s (`tools/exp/single_load.py`, `trace_load.sh`; a pageable copy out of an mmap of the file runs at link
// speed only while its pages stay mapped -- a fresh mapping per file pays 0.6 ms of page faults per 32 MB before the first byte moves).
// byteswap: the file has the other endianness.
//
// Round 6: the engine itself lives in pdbeda_upload.h and reaches the runtime through a table of function pointers -- the HIP calls below
// here, a host stand-in in tests/upload_harness.cpp, where the engine runs under ThreadSanitizer and AddressSanitizer (tools/sanitize_cpu.sh).
using pdbeda_upload::UploadEngine;
using pdbeda_upload::UploadLoad;
using pdbeda_upload::now_s;
using pdbeda_upload::upload_trace;
static_assert(pdbeda_ctx::MAX_READERS == pdbeda_upload::MAX_READERS, "a context keeps one event per reader");
static int up_code(hipError_t e) { return e == hipSuccess ? pdbeda_upload::UP_OK : (e == hipErrorNotReady ? pdbeda_upload::UP_NOT_READY : (int)e); }
static hipError_t up_error(int code) { return code == pdbeda_upload::UP_OK ? hipSuccess : (code < 0 ? hipErrorUnknown : (hipError_t)code); }
static const pdbeda_upload::Backend g_hip_backend = {
    [](int device) { return up_code(hipSetDevice(device)); },
    [](pdbeda_upload::Stream *out) { hipStream_t s = nullptr; const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking); if (e != hipSuccess) (void)hipGetLastError(); *out = (void *)s; return up_code(e); },
    [](void **out, size_t bytes) { const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault); if (e != hipSuccess) (void)hipGetLastError(); return up_code(e); },
    [](pdbeda_upload::Event *out) { hipEvent_t ev = nullptr; const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); if (e != hipSuccess) (void)hipGetLastError(); *out = (void *)ev; return up_code(e); },
    [](pdbeda_upload::Event ev) { return up_code(hipEventQuery((hipEvent_t)ev)); },
    [](void *dst, const void *src, size_t bytes, pdbeda_upload::Stream s) { return up_code(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s)); },
    [](pdbeda_upload::Event ev, pdbeda_upload::Stream s) { return up_code(hipEventRecord((hipEvent_t)ev, (hipStream_t)s)); },
};

static std::mutex g_engine_mu;
static std::map<int, UploadEngine *> g_engines;   // (never destroyed: the readers are parked on the queue when the process ends)

static UploadEngine *upload_engine(int device) {
    std::lock_guard<std::mutex> g(g_engine_mu);
    auto it = g_engines.find(device);
    if (it != g_engines.end()) return it->second;
    UploadEngine *en = UploadEngine::create(&g_hip_backend, device);
    g_engines[device] = en;
    return en;
}

static bool ensure_reader_events(pdbeda_ctx *ctx, UploadEngine *engine) {
    for (int r = 0; r < engine->n_readers; ++r)
        if (!ctx->reader_ev[r] && hipEventCreateWithFlags(&ctx->reader_ev[r], hipEventDisableTiming) != hipSuccess) {
            ctx->reader_ev[r] = nullptr;
            (void)hipGetLastError();
            return false;
        }
    return true;
}

// One map's bytes -- from a file (fd >= 0: pread at offset + position) or from the caller's memory (src) -- into HBM at dst through the process's
// upload engine; on return every chunk is queued on a reader's stream (or given up) and ctx->stream waits for them.  *why: a file-level reason.
static hipError_t engine_copy(pdbeda_ctx *ctx, UploadEngine *engine, int fd, int64_t offset, const char *src, char *dst, size_t need, const char **why) {
    hipError_t e = hipSuccess;
    // The arena may be a recycled one: the pool protects a recycled arena by STREAM ORDER on ctx->stream (maps and lists are
    // freed without a host sync while their kernels are still queued; the debug poison fill above is queued there too).  The
    // readers' copies go through their own streams, which know nothing of that order -- so the first chunk is handed out only
    // when everything queued on ctx->stream up to here has run (normally: nothing is pending, the query says so at once).
    if (e == hipSuccess) {
        hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipErrorNotReady) q = ctx_wait(ctx);     // (timed, under the watchdog)
        if (q != hipSuccess) e = q;
    }
    if (e == hipSuccess && !ctx->timed_out) {
        UploadLoad ld;
        ld.fd = fd; ld.offset = offset; ld.src = src; ld.dst = dst; ld.need = need;
        ld.timeout_s = ctx->timeout_s;
        ld.deadline = ctx->deadline;
        engine->submit(ld);      // (chunked, queued on the readers' streams or given up: pdbeda_upload.h)
        // the context's stream waits for the readers' copies of THIS map: an event behind whatever each reader has queued so far
        // (a little more than needed when it has gone on to another load's chunk -- never less)
        for (int r = 0; r < engine->n_readers; ++r) {
            if (!ld.used[r]) continue;
            hipError_t je = hipEventRecord(ctx->reader_ev[r], (hipStream_t)ld.stream_used[r]);      // (the stream this load's chunks went to: the reader may have taken a new one since)
            if (je == hipSuccess) je = hipStreamWaitEvent(ctx->stream, ctx->reader_ev[r], 0);
            if (je != hipSuccess && e == hipSuccess) e = je;
        }
        if (ld.e != pdbeda_upload::UP_OK && e == hipSuccess) e = up_error(ld.e);
        if (ld.stalled && e == hipSuccess) { e = hipErrorNotReady; if (!ld.why) ld.why = "a copy stream of the upload engine stalled behind another copy and was given up; the map is incomplete"; }
        *why = ld.why;
        if (ld.timed_out) {
            ctx->timed_out = true;
            ctx->err = "watchdog: the stream did not drain within the per-entry time-out; context abandoned";
            if (e == hipSuccess) e = hipErrorNotReady;
        }
    }
    return e;
}

static int upload_file_impl(pdbeda_ctx *ctx, const char *path, int64_t offset, int byteswap, const pdbeda_geometry *geom, pdbeda_map **out, double *mean, double *std) {
    if (!ctx || !path || !geom || !out || offset < 0) return PDBEDA_ERR_ARGUMENT;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n_vox = (int64_t)geom->ncrs[0] * geom->ncrs[1] * geom->ncrs[2];
    if (n_vox <= 0 || n_vox >= (1ll << 32)) return fail(ctx, PDBEDA_ERR_ARGUMENT, "grid size out of range");
    const size_t need = sizeof(float) * (size_t)n_vox;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return fail(ctx, PDBEDA_ERR_ARGUMENT, "cannot open %s: %s", path, strerror(errno));
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (int64_t)sb.st_size < offset + (int64_t)need) {
        close(fd);
        return fail(ctx, PDBEDA_ERR_ARGUMENT, "%s holds fewer than %lld grid bytes after offset %lld", path, (long long)need, (long long)offset);
    }
    UploadEngine *engine = upload_engine(ctx->device);
    if (engine->n_readers < 1) { close(fd); return fail(ctx, PDBEDA_ERR_MEMORY, "no reader (pinned chunks, copy stream, thread) for the file upload"); }
    if (!ensure_reader_events(ctx, engine)) { close(fd); return fail(ctx, PDBEDA_ERR_MEMORY, "no event for the file upload"); }
    pdbeda_map *m = new pdbeda_map();
    m->ctx = ctx;
    int rc = fill_geom(ctx, geom, &m->geom);
    if (rc == 0) rc = arena_get(ctx, align_up(sizeof(Geom)) + align_up(need), &m->arena);
    if (rc) { close(fd); delete m; return rc; }
    m->n_vox = n_vox;
    Carver cv(m->arena.base);
    m->geom_dev = cv.take<Geom>(1);
    float *d = cv.take<float>((size_t)n_vox);
    m->dens = d;
    m->own_dens = true;
    const char *why = nullptr;
    hipError_t e = engine_copy(ctx, engine, fd, offset, nullptr, (char *)d, need, &why);   // (first: the stream is idle, its query says so at once)
    const bool geom_in_stats = (mean || std) && stats_carries_geom(m);      // (the statistics' first launch brings the geometry along: a copy launch less)
    if (e == hipSuccess && !geom_in_stats) e = h2d_one(ctx, m->geom_dev, &m->geom, sizeof(Geom));
    close(fd);
    if (e == hipSuccess && !why && byteswap) {
        hipLaunchKernelGGL(k_byteswap32, dim3(grid_for(n_vox, 256, 8192)), dim3(256), 0, ctx->stream, (uint32_t *)d, n_vox);
        e = hipGetLastError();
    }
    // mean / std (and the range of the blob sums' quantum) queued behind the copies: ONE wait for the map and its statistics -- every
    // caller asks for them next (cutoffs are mean + k std), and a wait of their own was one of a pool entry's host round trips
    const bool with_stats = (mean || std) && e == hipSuccess && !why;
    double host[2] = {0.0, 0.0}, range[2] = {0.0, 0.0};
    Arena scratch;
    bool have_scratch = false;
    if (with_stats) {
        if (arena_get(ctx, 24 * (size_t)std::max<int64_t>(n_vox / NP_CHUNK, 1), &scratch) == 0) {
            have_scratch = true;
            if (stats_enqueue(m, reinterpret_cast<double *>(scratch.base), host, range, true, geom_in_stats) != 0) e = hipErrorUnknown;
        } else {
            e = hipErrorOutOfMemory;
        }
    }
    const double t_s0 = now_s();
    const hipError_t e2 = ctx_sync(ctx);      // (&m->geom is free again)
    if (upload_trace()) fprintf(stderr, "upload: final wait %.3f ms\n", 1e3 * (now_s() - t_s0));
    if (have_scratch) arena_put(ctx, scratch);
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess || why) {
        arena_put(ctx, m->arena);
        delete m;
        return why ? fail(ctx, PDBEDA_ERR_ARGUMENT, "reading %s: %s", path, why) : fail(ctx, PDBEDA_ERR_DEVICE, "map upload from file: %s", hipGetErrorString(e));
    }
    if (with_stats) {
        range_apply(m, range);
        if (mean) *mean = host[0];
        if (std) *std = host[1];
    }
    ctx->live_handles++;
    *out = m;
    return PDBEDA_OK;
}

extern "C" int pdbeda_map_upload_file(pdbeda_ctx *ctx, const char *path, int64_t offset, int byteswap, const pdbeda_geometry *geom, pdbeda_map **out) {
    return upload_file_impl(ctx, path, offset, byteswap, geom, out, nullptr, nullptr);
}
extern "C" int pdbeda_map_upload_file_stats(pdbeda_ctx *ctx, const char *path, int64_t offset, int byteswap, const pdbeda_geometry *geom, pdbeda_map **out,
                                            double *mean, double *std) {
    return upload_file_impl(ctx, path, offset, byteswap, geom, out, mean, std);
}

extern "C" int pdbeda_map_combine(pdbeda_map *a, pdbeda_map *b, double alpha, pdbeda_map **out) {
    if (!a || !b || !out) return PDBEDA_ERR_ARGUMENT;
    *out = nullptr;
    pdbeda_ctx *ctx = a->ctx;
    if (b->ctx != ctx) return fail(ctx, PDBEDA_ERR_ARGUMENT, "maps of different contexts");
    if (a->n_vox != b->n_vox || memcmp(a->geom.ncrs, b->geom.ncrs, sizeof a->geom.ncrs) != 0)
        return fail(ctx, PDBEDA_ERR_ARGUMENT, "maps of different shapes");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pdbeda_map *m = new pdbeda_map();
    m->ctx = ctx;
    m->geom = a->geom;
    m->n_vox = a->n_vox;
    int rc = arena_get(ctx, align_up(sizeof(Geom)) + align_up(sizeof(float) * (size_t)m->n_vox), &m->arena);
    if (rc) { delete m; return rc; }
    Carver cv(m->arena.base);
    m->geom_dev = cv.take<Geom>(1);
    float *d = cv.take<float>((size_t)m->n_vox);
    hipError_t e = h2d_one(ctx, m->geom_dev, &m->geom, sizeof(Geom));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_map_combine, dim3(grid_for(m->n_vox / 4 + 1, 256, 4096)), dim3(256), 0, ctx->stream, a->dens, b->dens, alpha, m->n_vox, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = ctx_sync(ctx);   // (&m->geom is read by the copy above)
    if (e != hipSuccess) {
        arena_put(ctx, m->arena);
        delete m;
        return fail(ctx, PDBEDA_ERR_DEVICE, "map combine: %s", hipGetErrorString(e));
    }
    m->dens = d;
    m->own_dens = true;
    ctx->live_handles++;
    *out = m;
    return PDBEDA_OK;
}

extern "C" int pdbeda_map_download(pdbeda_map *m, float *density_out) {
    if (!m || !density_out) return PDBEDA_ERR_ARGUMENT;
    pdbeda_ctx *ctx = m->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, d2h(ctx, density_out, m->dens, sizeof(float) * (size_t)m->n_vox));
    HIP_TRY(ctx, ctx_sync(ctx));
    return PDBEDA_OK;
}

extern "C" int pdbeda_map_free(pdbeda_map *m) {
    if (!m) return PDBEDA_ERR_ARGUMENT;
    pdbeda_ctx *ctx = m->ctx;
    // the arena goes back to the context's pool: whoever gets it next is enqueued on this stream behind the last kernel that
    // reads the map -- no wait, no hipFree (which would synchronise the whole device under the other streams of a pool)
    arena_put(ctx, m->arena);
    ctx->live_handles--;
    delete m;
    return PDBEDA_OK;
}

// ------------------------------------------------------------------------------------
// Whole-map reductions
// ------------------------------------------------------------------------------------
static int reduce_launch(pdbeda_map *m, int mode, const double *mean_dev, double cutoff, double scale, int take_sqrt, double *out_dev) {
    pdbeda_ctx *ctx = m->ctx;
    hipLaunchKernelGGL(k_reduce_partials, dim3(N_PARTIAL), dim3(256), 0, ctx->stream, m->dens, m->n_vox, mode, mean_dev, cutoff, ctx->partials);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, ctx->partials, N_PARTIAL, scale, take_sqrt, out_dev);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// The quantum of the order-independent blob sums of this map (see FixSums in pdbeda_kernels.h): S such that neither
// sum |rho| over the whole map nor the first moment of one tile / one run (<= 2^22 max |rho|) can leave 62 bits.  Two
// deterministic reductions over the map, once per map (the result is cached; a borrowed device pointer is taken to hold the
// same grid for the life of the map).
// The quantum of a map's integer blob sums from its range (sum |x|, max |x|: fixed reduction order, so the quantum is the same in
// every run).  range_enqueue + range_apply: the pass rides in the same wait as the map's mean / std when those are asked for
// first (pdbeda_map_stats: every DensityMatrix asks at once) -- a wait of its own in front of the map's first labelling job was
// one of an entry's ~17 host round trips (round 4).
static int range_enqueue(pdbeda_map *m, double *range_host /* [2], filled at the next ctx_sync; nullptr: the caller copies the result slots itself */) {
    pdbeda_ctx *ctx = m->ctx;
    double *res = ctx->partials + 2 * N_PARTIAL + 4;
    { PROF(ctx, "k_range_partials"); hipLaunchKernelGGL(k_range_partials, dim3(N_PARTIAL), dim3(256), 0, ctx->stream, m->dens, m->n_vox, ctx->partials, ctx->partials + N_PARTIAL); }
    hipLaunchKernelGGL(k_range_final, dim3(1), dim3(256), 0, ctx->stream, ctx->partials, ctx->partials + N_PARTIAL, N_PARTIAL, res);
    HIP_TRY(ctx, hipGetLastError());
    if (range_host) HIP_TRY(ctx, d2h(ctx, range_host, res, 2 * sizeof(double)));
    return 0;
}
static void range_apply(pdbeda_map *m, const double range[2]) {
    // the integer sums cannot hold a NaN or an infinity (fix_of would saturate without a word): such a map is refused by the
    // labelling calls instead of producing wrong blob totals (the reference's sums would be NaN / inf for every blob that holds one)
    if (!std::isfinite(range[0]) || !std::isfinite(range[1])) { m->fix_refused = true; return; }
    const double bound = std::max(range[0], 4194304.0 * range[1]);
    int S = 40;
    if (bound > 0.0 && std::isfinite(bound)) {
        int e = 0;
        (void)frexp(bound, &e);        // bound < 2^e
        S = 61 - e;
    }
    S = std::max(-900, std::min(S, 900));
    m->fix_mul = ldexp(1.0, S);
}
static int map_fix_mul(pdbeda_map *m) {
    pdbeda_ctx *ctx = m->ctx;
    if (m->fix_mul == 0.0 && !m->fix_refused) {
        double range[2] = {0.0, 0.0};
        const int rc = range_enqueue(m, range);
        if (rc) return rc;
        HIP_TRY(ctx, ctx_sync(ctx));
        range_apply(m, range);
    }
    if (m->fix_refused) return fail(ctx, PDBEDA_ERR_ARGUMENT, "the map holds non-finite density values: blob sums are undefined");
    return 0;
}

// A map made by pdbeda_map_from_device borrows the caller's buffer; what the library caches about its CONTENTS -- the quantum of
// the order-independent blob sums, derived from sum |rho| and max |rho| -- must be dropped when the caller rewrites the buffer
// in place (larger values would overflow the 62-bit sums silently).
extern "C" int pdbeda_map_invalidate(pdbeda_map *m) {
    if (!m) return PDBEDA_ERR_ARGUMENT;
    m->fix_mul = 0.0;
    m->fix_refused = false;
    return PDBEDA_OK;
}

// mean and std in numpy's own summation tree (k_np_chunk_sums / k_np_final): == np.mean / np.std, not merely close -- and, while
// the map's range is not known yet, the range pass behind them (range_enqueue): everything queued, nothing waited for.
// chunk_sums: 8 * max(n_vox / NP_CHUNK, 1) bytes of device scratch; host[2] / range[2] are filled at the next ctx_sync.
// chunk_sums: scratch of 3 x max(n_vox / NP_CHUNK, 1) doubles (the chunks' numpy sums; with want_range their range partials behind them).
// Round 6: with want_range the range rides in the mean's pass (k_np_chunk_sums / k_np_final, mode 0) and the LAST launch of the chain writes mean,
// std and range straight into the pinned block: four launches where there were seven (two range kernels and a copy), and one pass over the map less.
// with_geom: the first launch of the chain also writes the map's geometry struct to the device (a kernel argument by value; callers check stats_carries_geom)
static int stats_enqueue(pdbeda_map *m, double *chunk_sums, double host[2], double range[2], bool want_range, bool with_geom) {
    pdbeda_ctx *ctx = m->ctx;
    // (mean / std in the first two of the eight result slots behind the partial sums, the range in slots 4 / 5)
    double *res = ctx->partials + 2 * N_PARTIAL;
    const int64_t n_full = m->n_vox / NP_CHUNK;
    const int64_t n_chunks = std::max<int64_t>(n_full, 1);
    hipStream_t st = ctx->stream;
    // the results' place in the pinned block (kernels write host memory only with PDBEDA_COPY_KERNELS, the default)
    double *host_out = nullptr;
    if (want_range && copy_kernels() && ctx->pinned && ctx->pinned_used + 64 <= ctx->pinned_cap) {
        host_out = reinterpret_cast<double *>(ctx->pinned + ctx->pinned_used);
        ctx->pending.push_back({host, ctx->pinned_used, 2 * sizeof(double)});
        ctx->pending.push_back({range, ctx->pinned_used + 4 * sizeof(double), 2 * sizeof(double)});
        ctx->pinned_used += 64;
    }
    double *r_sum = want_range ? chunk_sums + n_chunks : nullptr, *r_max = want_range ? chunk_sums + 2 * n_chunks : nullptr;
    for (int mode = 0; mode < 2; ++mode) {
        const bool with_range = want_range && mode == 0;
        if (n_full > 0) { PROF(ctx, "k_np_chunk_sums"); hipLaunchKernelGGL(k_np_chunk_sums, dim3((unsigned)std::min<int64_t>(n_full, 1 << 16)), dim3(256), 0, st, m->dens, n_full, mode, res, chunk_sums,
                                                                           with_range ? r_sum : (double *)nullptr, with_range ? r_max : (double *)nullptr,
                                                                           m->geom, with_geom && mode == 0 ? m->geom_dev : (Geom *)nullptr); }
        { PROF(ctx, "k_np_final"); hipLaunchKernelGGL(k_np_final, dim3(1), dim3(256), 0, st, m->dens, m->n_vox, n_full, mode, res, chunk_sums, mode, res + mode,
                                                      with_range && n_full > 0 ? r_sum : (const double *)nullptr, with_range && n_full > 0 ? r_max : (const double *)nullptr,
                                                      with_range ? res + 4 : (double *)nullptr, want_range && mode == 1 ? res + 4 : (const double *)nullptr, mode == 1 ? host_out : (double *)nullptr); }
    }
    HIP_TRY(ctx, hipGetLastError());
    if (host_out) return 0;
    if (!want_range) {
        HIP_TRY(ctx, d2h(ctx, host, res, 2 * sizeof(double)));
        return 0;
    }
    // (no room in the pinned block, or the runtime's copies asked for: one copy of the six doubles, or two)
    if (ctx->pinned && ctx->pinned_used + 64 <= ctx->pinned_cap) {
        HIP_TRY(ctx, copy_kernels() ? copy_by_kernel(ctx, ctx->pinned + ctx->pinned_used, res, 6 * sizeof(double))
                                    : hipMemcpyAsync(ctx->pinned + ctx->pinned_used, res, 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ctx->pending.push_back({host, ctx->pinned_used, 2 * sizeof(double)});
        ctx->pending.push_back({range, ctx->pinned_used + 4 * sizeof(double), 2 * sizeof(double)});
        ctx->pinned_used += 64;
        return 0;
    }
    HIP_TRY(ctx, d2h(ctx, host, res, 2 * sizeof(double)));
    HIP_TRY(ctx, d2h(ctx, range, res + 4, 2 * sizeof(double)));
    return 0;
}

extern "C" int pdbeda_map_stats(pdbeda_map *m, double *mean, double *std) {
    if (!m) return PDBEDA_ERR_ARGUMENT;
    pdbeda_ctx *ctx = m->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n_full = m->n_vox / NP_CHUNK;
    double host[2], range[2] = {0.0, 0.0};
    const bool want_range = m->fix_mul == 0.0 && !m->fix_refused;   // the quantum of the map's blob sums, in the same wait
    int rc = with_scratch(ctx, 24 * (size_t)std::max<int64_t>(n_full, 1), [&](char *base) -> int {
        return stats_enqueue(m, reinterpret_cast<double *>(base), host, range, want_range);
    });
    if (rc) return rc;
    if (want_range) range_apply(m, range);
    if (mean) *mean = host[0];
    if (std) *std = host[1];
    return PDBEDA_OK;
}

extern "C" int pdbeda_sum_of_abs(pdbeda_map *m, float cutoff, double *out) {
    if (!m || !out) return PDBEDA_ERR_ARGUMENT;
    pdbeda_ctx *ctx = m->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double *res = ctx->partials + N_PARTIAL + 2;
    int rc = reduce_launch(m, 2, nullptr, (double)cutoff, 1.0, 0, res);
    if (rc) return rc;
    HIP_TRY(ctx, d2h(ctx, out, res, sizeof(double)));
    HIP_TRY(ctx, ctx_sync(ctx));
    return PDBEDA_OK;
}

extern "C" int pdbeda_abs_select_hist(pdbeda_map *a, pdbeda_map *b, double alpha, double cut_a, double cut_b, int which, int shift,
                                      unsigned long long prefix, unsigned long long prefix_mask, uint32_t *hist) {
    if (!a || !hist || shift < 0 || shift > 48 || (which != 0 && which != 1) || (which == 1 && !b)) return PDBEDA_ERR_ARGUMENT;
    pdbeda_ctx *ctx = a->ctx;
    if (b && (b->ctx != ctx || b->n_vox != a->n_vox || memcmp(a->geom.ncrs, b->geom.ncrs, sizeof a->geom.ncrs) != 0))
        return fail(ctx, PDBEDA_ERR_ARGUMENT, "maps of different contexts or shapes");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n = (int64_t)a->geom.unique_ncrs[0] * a->geom.unique_ncrs[1] * a->geom.unique_ncrs[2];
    return with_scratch(ctx, 65536 * sizeof(uint32_t), [&](char *base) -> int {
        unsigned int *d_hist = reinterpret_cast<unsigned int *>(base);
        HIP_TRY(ctx, hipMemsetAsync(d_hist, 0, 65536 * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(k_abs_select_hist, dim3(grid_for(n, 256, 4096)), dim3(256), 0, ctx->stream, a->geom_dev, a->dens, b ? b->dens : nullptr,
                           alpha, cut_a, cut_b, which, shift, prefix, prefix_mask, d_hist);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, d2h(ctx, hist, d_hist, 65536 * sizeof(uint32_t)));
        return 0;
    });
}

extern "C" int pdbeda_point_density(pdbeda_map *m, const int32_t *crs, int64_t n, double *out) {
    if (!m || (n > 0 && (!crs || !out)) || n < 0) return PDBEDA_ERR_ARGUMENT;
    if (n == 0) return PDBEDA_OK;
    pdbeda_ctx *ctx = m->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return with_scratch(ctx, align_up(12 * n) + align_up(8 * n), [&](char *base) -> int {
        Carver cv(base);
        int32_t *d_crs = cv.take<int32_t>(3 * n);
        double *d_out = cv.take<double>(n);
        const int32_t *in = pinned_in(ctx, crs, (size_t)(3 * n));
        if (!in) { HIP_TRY(ctx, h2d_one(ctx, d_crs, crs, (size_t)(12 * n))); in = d_crs; }
        double *res = pinned_out(ctx, out, (size_t)n);
        hipLaunchKernelGGL(k_point_density, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, m->geom_dev, m->dens, in, n, res ? res : d_out, (uint8_t *)nullptr);
        if (!res) HIP_TRY(ctx, d2h(ctx, out, d_out, 8 * n));
        return 0;
    });
}

extern "C" int pdbeda_valid_crs(pdbeda_map *m, const int32_t *crs, int64_t n, uint8_t *out) {
    if (!m || (n > 0 && (!crs || !out)) || n < 0) return PDBEDA_ERR_ARGUMENT;
    if (n == 0) return PDBEDA_OK;
    pdbeda_ctx *ctx = m->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return with_scratch(ctx, align_up(12 * n) + align_up(n), [&](char *base) -> int {
        Carver cv(base);
        int32_t *d_crs = cv.take<int32_t>(3 * n);
        uint8_t *d_out = cv.take<uint8_t>(n);
        const int32_t *in = pinned_in(ctx, crs, (size_t)(3 * n));
        if (!in) { HIP_TRY(ctx, h2d_one(ctx, d_crs, crs, (size_t)(12 * n))); in = d_crs; }
        uint8_t *res = pinned_out(ctx, out, (size_t)n);
        hipLaunchKernelGGL(k_point_density, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, m->geom_dev, m->dens, in, n, (double *)nullptr, res ? res : d_out);
        if (!res) HIP_TRY(ctx, d2h(ctx, out, d_out, n));
        return 0;
    });
}

extern "C" int pdbeda_crs2xyz(pdbeda_map *m, const int32_t *crs, int64_t n, double *xyz) {
    if (!m || (n > 0 && (!crs || !xyz)) || n < 0) return PDBEDA_ERR_ARGUMENT;
    if (n == 0) return PDBEDA_OK;
    pdbeda_ctx *ctx = m->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return with_scratch(ctx, align_up(12 * n) + align_up(24 * n), [&](char *base) -> int {
        Carver cv(base);
        int32_t *d_crs = cv.take<int32_t>(3 * n);
        double *d_xyz = cv.take<double>(3 * n);
        const int32_t *in = pinned_in(ctx, crs, (size_t)(3 * n));
        if (!in) { HIP_TRY(ctx, h2d_one(ctx, d_crs, crs, (size_t)(12 * n))); in = d_crs; }
        double *res = pinned_out(ctx, xyz, (size_t)(3 * n));
        hipLaunchKernelGGL(k_crs2xyz, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, m->geom_dev, in, n, res ? res : d_xyz);
        if (!res) HIP_TRY(ctx, d2h(ctx, xyz, d_xyz, 24 * n));
        return 0;
    });
}

extern "C" int pdbeda_xyz2crs(pdbeda_map *m, const double *xyz, int64_t n, int32_t *crs) {
    if (!m || (n > 0 && (!crs || !xyz)) || n < 0) return PDBEDA_ERR_ARGUMENT;
    if (n == 0) return PDBEDA_OK;
    pdbeda_ctx *ctx = m->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return with_scratch(ctx, align_up(12 * n) + align_up(24 * n), [&](char *base) -> int {
        Carver cv(base);
        double *d_xyz = cv.take<double>(3 * n);
        int32_t *d_crs = cv.take<int32_t>(3 * n);
        const double *in = pinned_in(ctx, xyz, (size_t)(3 * n));
        if (!in) { HIP_TRY(ctx, h2d_one(ctx, d_xyz, xyz, (size_t)(24 * n))); in = d_xyz; }
        int32_t *res = pinned_out(ctx, crs, (size_t)(3 * n));
        hipLaunchKernelGGL(k_xyz2crs, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, m->geom_dev, in, n, res ? res : d_crs);
        if (!res) HIP_TRY(ctx, d2h(ctx, crs, d_crs, 12 * n));
        return 0;
    });
}

// ------------------------------------------------------------------------------------
// Labelling jobs
// ------------------------------------------------------------------------------------
// Carve a job out of an arena.  max_runs / max_blobs are worst-case bounds (a run needs a
// gap: <= bits/2 + 1 per word; a blob owns >= one 2x2x2 cell... we simply bound blobs by runs).
// max_runs: run ids (comp_of_run); max_comps: component ids (every per-component array; generic jobs: == max_runs, a run is a component)
static size_t job_carve(Job &job, char *base, int n_vols, int64_t total_words, int64_t total_keys, int64_t max_runs,
                        int64_t max_blobs, size_t extra_labels, int32_t **labels_out, int64_t n_tiles = 0, int64_t max_comps = -1) {
    if (max_comps < 0) max_comps = max_runs;
    job.run_cap = (uint32_t)max_runs; job.comp_cap = (uint32_t)max_comps; job.blob_cap = (uint32_t)std::min<int64_t>(max_blobs, 0xffffffffll);
    Carver cv(base);
    job.n_vols = n_vols;
    job.total_words = total_words;
    job.key_words = (total_keys + 63) / 64;
    job.n_fine = (int32_t)((job.key_words + KEY_FINE - 1) / KEY_FINE);
    job.fine_shift = 4;   // counters per table entry: a power of two >= 16 that covers the job with KEY_GROUPS entries
    while (((int64_t)KEY_GROUPS << job.fine_shift) < job.n_fine) job.fine_shift++;
    job.fine_per_group = 1 << job.fine_shift;
    job.n_groups = (job.n_fine + job.fine_per_group - 1) / job.fine_per_group;
    job.vols = cv.take<VolDesc>(std::max(n_vols, 1));
    // counters, masks, first-key bitmap and rank counters in a row: a sphere / list job clears them with ONE fill (grouped_job;
    // whole-map jobs clear theirs in kernels)
    job.ctr = cv.take<Counters>(1);
    job.mask = cv.take<uint64_t>(total_words);
    job.key_bits = cv.take<uint64_t>((size_t)(job.key_words + KEY_FINE - 1) / KEY_FINE * KEY_FINE);   // whole fine buckets (rank_of_key loads a bucket whole)
    // the rank counters sit right behind the bitmap (ONE clear covers both), padded to whole groups
    job.n_fine_alloc = (int32_t)((job.n_fine + job.fine_per_group - 1) / job.fine_per_group * job.fine_per_group);
    job.fine_count = cv.take<uint32_t>((size_t)std::max(job.n_fine_alloc / 2, 8));   // 16-bit counters, two per word
    job.mid_count = cv.take<uint32_t>((size_t)(job.key_words + KEY_FINE - 1) / KEY_FINE * (KEY_FINE / 16));   // a byte per 4 key words: 8 per bucket
    job.group_count = (n_tiles && job.fine_shift > 4) ? cv.take<uint32_t>(KEY_GROUPS) : nullptr;   // (groups of 32 counters or more: maps beyond 2^25 keys = 256^3 fused)
    job.run_base = cv.take<uint32_t>(total_words);
    job.comps_are_runs = n_tiles ? 0 : 1;
    job.comp_of_run = n_tiles ? cv.take<uint32_t>(max_runs) : nullptr;
    job.label_of_comp = n_tiles ? cv.take<int32_t>(max_comps) : nullptr;
    job.word_comps = n_tiles ? cv.take<uint8_t>((size_t)n_tiles * 2 * 256 * 8) : nullptr;
    job.unit_done = n_tiles ? cv.take<uint32_t>((size_t)n_tiles) : nullptr;
    job.unit_flag = n_tiles ? cv.take<uint32_t>(4) : nullptr;
    job.tile_mode = n_tiles ? cv.take<uint8_t>(n_tiles) : nullptr;
    job.tile_runs = n_tiles ? cv.take<uint32_t>(n_tiles) : nullptr;
    job.root_mask = n_tiles ? cv.take<uint64_t>((size_t)n_tiles * 4) : nullptr;
    job.n_tiles = (int32_t)n_tiles;
    if (n_tiles) {   // (what a tile's k_face_merge workgroup clears)
        const int64_t nt = n_tiles, nfc = job.n_fine_alloc / 2, nmid = (job.key_words + KEY_FINE - 1) / KEY_FINE * (KEY_FINE / 16);
        job.clear_bits = (int32_t)((job.key_words + nt - 1) / nt);
        job.clear_fine = (int32_t)((nfc + nt - 1) / nt);
        job.clear_mid = (int32_t)((nmid + nt - 1) / nt);
    }
    job.inbox = n_tiles ? cv.take<InboxEntry>((size_t)n_tiles * INBOX_CAP) : nullptr;
    job.inbox_count = n_tiles ? cv.take<uint32_t>((size_t)n_tiles * INBOX_STRIDE) : nullptr;
    job.vol_sign[0] = job.vol_sign[1] = 1;
    job.parent = cv.take<int32_t>(max_comps);
    job.kpar = n_tiles ? cv.take<unsigned long long>(max_comps) : nullptr;
    job.r_n = cv.take<uint32_t>(max_comps);
    job.r_sum = cv.take<long long>((size_t)7 * max_comps);
    job.r_sum_stride = max_comps;
    job.r_c = cv.take<long long>(max_comps);
    job.r_r = cv.take<long long>(max_comps);
    job.r_s = cv.take<long long>(max_comps);
    job.r_key = cv.take<unsigned long long>(max_comps);
    job.r_rank = cv.take<uint32_t>(max_comps);
    job.b_n = cv.take<int64_t>(max_blobs);
    job.b_key = cv.take<int64_t>(max_blobs);
    job.b_total = cv.take<double>(max_blobs);
    job.b_centroid = cv.take<double>(3 * max_blobs);
    job.b_center = cv.take<double>(3 * max_blobs);
    job.b_volume = cv.take<double>(max_blobs);
    job.b_group = cv.take<int32_t>(max_blobs);
    int32_t *lab = extra_labels ? cv.take<int32_t>(extra_labels) : nullptr;
    if (labels_out) *labels_out = lab;
    return cv.off;
}

// Enqueue the labelling engine on a job whose masks are already painted.
// labelled: k_atom_engine has done everything up to the painted keys.  unordered: the caller takes the roots' rows as they come (k_resolve hands them
// out) and writes its results itself (pdbeda_aggregate_cloud's union job: k_union_finish) -- no keys painted, no ranks, no blob table
static int engine_enqueue(pdbeda_ctx *ctx, pdbeda_map *m, Job &job, int64_t max_runs, bool labelled = false, bool unordered = false) {
    hipStream_t st = ctx->stream;
    // per-run kernels are grid-stride over the ACTUAL run count (read on the device)
    const unsigned run_grid = grid_for(max_runs, 256, 2048);
    if (job.total_words > 0 && !labelled) {
        {   // few words in large volumes: a quarter of the words a block, four times the blocks (the loop over a wave's words is
            // serial; the rows of many small volumes -- atom spheres -- are mostly narrow: a thread each, 256 to the block)
            PROF(ctx, "k_run_index");
            static const int force_wpb = [] { const char *e = getenv("PDBEDA_RUNINDEX_WPB"); return e ? atoi(e) : 0; }();   // (experiments: 64 / 256)
            if (force_wpb != 256 && (force_wpb == 64 || (job.total_words < 256ll * 2048 && job.total_words > 64ll * job.n_vols))) hipLaunchKernelGGL(k_run_index<64>, dim3(grid_for(job.total_words, 64, 1ll << 30)), dim3(256), 0, st, job, m->dens, m->geom_dev);
            else hipLaunchKernelGGL(k_run_index<256>, dim3(grid_for(job.total_words, 256, 1ll << 30)), dim3(256), 0, st, job, m->dens, m->geom_dev);
        }
        { PROF(ctx, "k_union"); hipLaunchKernelGGL(k_union, dim3(grid_for(job.total_words * 4, 256, 1ll << 30)), dim3(256), 0, st, job); }
        { PROF(ctx, "k_resolve"); hipLaunchKernelGGL(k_resolve, dim3(run_grid), dim3(256), 0, st, job, unordered ? 1 : 0); }
        if (!unordered) { PROF(ctx, "k_paint_keys"); hipLaunchKernelGGL(k_paint_keys, dim3(run_grid), dim3(256), 0, st, job); }
    }
    // (k_emit also runs for an empty job: its first block publishes the blob count)
    if (!unordered) { PROF(ctx, "k_emit"); hipLaunchKernelGGL(k_emit, dim3(job.total_words > 0 ? std::min(run_grid, 512u) : 1u), dim3(256), 0, st, job, m->geom_dev); }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

static pdbeda_bloblist *new_list(pdbeda_ctx *ctx, pdbeda_map *m) {
    pdbeda_bloblist *bl = new pdbeda_bloblist();
    bl->ctx = ctx;
    bl->map = m;
    ctx->live_handles++;
    return bl;
}

template <int CW>
static void launch_tile_label(pdbeda_ctx *ctx, unsigned n_tiles, const Job &job, const float *dens, const Geom *geom_dev, const TileDims &td,
                              const JobInit &init, int pair_slots) {
    { PROF(ctx, "k_tile_label"); hipLaunchKernelGGL((k_tile_label<CW>), dim3(td.ctiles, td.rtiles, td.stiles), dim3(512), 0, ctx->stream, job, dens, geom_dev, td, init); }
    // cross-tile unions (and, when a tile overflowed LDS, every pair that has such a tile on either side)
    {   // (grids of one tile column have no c faces: two waves fewer per workgroup to dispatch)
        PROF(ctx, "k_face_merge");
        if (td.ctiles > 1) hipLaunchKernelGGL((k_face_merge<CW, FM_THREADS_WIDE>), dim3(td.ctiles, td.rtiles, td.stiles), dim3(FM_THREADS_WIDE), 0, ctx->stream, job, dens, geom_dev, td, pair_slots);
        else hipLaunchKernelGGL((k_face_merge<CW, FM_THREADS>), dim3(td.ctiles, td.rtiles, td.stiles), dim3(FM_THREADS), 0, ctx->stream, job, dens, geom_dev, td, pair_slots);
    }
    if (job.unit_form) {   // the job is known to have unit tiles (its first run said so): their labelling, then their pairs
        { PROF(ctx, "k_unit_label"); hipLaunchKernelGGL((k_unit_label<CW>), dim3(td.ctiles, td.rtiles, td.stiles), dim3(512), 0, ctx->stream, job, dens, geom_dev, td); }
        { PROF(ctx, "k_unit_pairs"); hipLaunchKernelGGL((k_unit_pairs<CW>), dim3(td.ctiles, td.rtiles, td.stiles), dim3(512), 0, ctx->stream, job, td); }
    }
}

// fused: the launch also ranks the roots and writes the blob table (k_emit_tiles is then not launched) -- see k_labels_tiles
template <bool FUSED>
static void launch_labels(pdbeda_ctx *ctx, const Job &job, const TileDims &td, int32_t *labels_dev, const Geom *geom_dev) {
    switch (td.cw) {
        case 1: hipLaunchKernelGGL((k_labels_tiles<1, FUSED>), dim3(td.ctiles, td.rtiles, td.stiles), dim3(PDBEDA_LABELS_NT_THREADS), 0, ctx->stream, job, td, labels_dev, geom_dev); break;
        case 2: hipLaunchKernelGGL((k_labels_tiles<2, FUSED>), dim3(td.ctiles, td.rtiles, td.stiles), dim3(PDBEDA_LABELS_NT_THREADS), 0, ctx->stream, job, td, labels_dev, geom_dev); break;
        case 3: hipLaunchKernelGGL((k_labels_tiles<3, FUSED>), dim3(td.ctiles, td.rtiles, td.stiles), dim3(PDBEDA_LABELS_NT_THREADS), 0, ctx->stream, job, td, labels_dev, geom_dev); break;
        default: hipLaunchKernelGGL((k_labels_tiles<4, FUSED>), dim3(td.ctiles, td.rtiles, td.stiles), dim3(PDBEDA_LABELS_NT_THREADS), 0, ctx->stream, job, td, labels_dev, geom_dev); break;
    }
}

// Enqueue a whole-map job.  tier 0 carves the arena for what maps need in practice -- unit-tile run / component ids for a quarter of
// the tiles' own run-id range above the tiles' ranges, a blob table of one row per 32 keys -- instead of the worst case (every other
// voxel a run of a unit tile, a blob per 2x2x2 cell: 2.9 GB at 256^3, of which a job touches a few hundred MB).  A map that
// needs more raises Counters::overflow on the device, stays inside its arena, and is run again at tier 1 (the worst case)
// by the first accessor that reads the counters (list_resolve_counts): correct always, slower only where it was slow already.
struct WholeMapJob {
    Job job;
    Arena arena;
    TileDims td;
    int32_t *labels_dev = nullptr;
    bool labels = false;
    size_t bytes = 0;
};
static int whole_map_enqueue(pdbeda_map *m, float cut_pos, float cut_neg, bool want_pos, bool want_neg, uint32_t flags, int tier, int unit_form, WholeMapJob *out) {
    pdbeda_ctx *ctx = m->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const Geom &g = m->geom;
    const int n_planes = (want_pos ? 1 : 0) + (want_neg ? 1 : 0);
    const int uc = g.unique_ncrs[0], ur = g.unique_ncrs[1], us = g.unique_ncrs[2];
    const int row_words = (uc + 63) / 64;
    const int64_t words_pp = (int64_t)row_words * ur * us;
    const int64_t keys_pp = (int64_t)uc * ur * us;
    const int64_t total_words = words_pp * n_planes, total_keys = keys_pp * n_planes;
    // 26-connectivity: all voxels of an aligned 2x2x2 cell are mutually adjacent -> <= 1 blob per cell
    const int64_t worst_blobs = (int64_t)((uc + 1) / 2) * ((ur + 1) / 2) * ((us + 1) / 2) * n_planes + 1;
    const bool labels = (flags & PDBEDA_FLAG_LABELS) != 0;
    const size_t lab_elems = labels ? (size_t)keys_pp : 0;   // ONE signed volume, also for a fused call

    TileDims td;
    td.ctiles = (row_words + 3) / 4;                      // tiles of at most 4 mask words (256 voxels) along c ...
    td.cw = (row_words + td.ctiles - 1) / td.ctiles;      // ... of equal width: 5 words are 3 + 2, not 4 + 1
    td.rtiles = (ur + TILE_R - 1) / TILE_R;
    td.stiles = (us + TILE_S - 1) / TILE_S;
    td.n_planes = n_planes;
    td.uc = uc; td.ur = ur; td.us = us; td.row_words = row_words;
    td.nc = g.ncrs[0]; td.nr = g.ncrs[1];
    td.cut[0] = want_pos ? cut_pos : cut_neg;
    td.sign[0] = want_pos ? 1 : -1;
    td.cut[1] = cut_neg;
    td.sign[1] = -1;
    const int64_t tiles_pp = (int64_t)td.ctiles * td.rtiles * td.stiles;
    if (tiles_pp >= (1ll << 31) || keys_pp >= (1ll << 31)) return fail(ctx, PDBEDA_ERR_ARGUMENT, "grid too large");   // (first keys inside a plane are 31-bit)
    if (td.rtiles > 65535 || td.stiles > 65535) return fail(ctx, PDBEDA_ERR_ARGUMENT, "grid too large");              // (the tile kernels' grids are 3-D)
    // run / component ids: a fixed range per tile (no allocation atomics) + the ids of the unit tiles above them
    const int64_t tile_runs = tiles_pp * td.cw * 64 * 32, tile_comps = tiles_pp * CCAP;
    const int64_t worst_unit = (int64_t)((uc + 1) / 2) * ur * us * n_planes + 1;       // every other voxel a run
    if (tile_runs + worst_unit >= (1ll << 31))
        return fail(ctx, PDBEDA_ERR_ARGUMENT, "grid too large for whole-map labelling: %d x %d x %d voxels need %lld run ids (limit 2^31: about 1100^3 for a fused job)",
                    uc, ur, us, (long long)(tile_runs + worst_unit));
    // (tier 0: a quarter of the tiles' own run-id range -- every tile a unit tile of 2 048 word-runs, or a quarter of them at the
    //  most a tile can hold: a protein-like map at 0.5 sigma, every tile over its LDS capacities, still runs once)
    const int64_t unit_ids = tier == 0 ? std::min<int64_t>(worst_unit, std::max<int64_t>(65536, tile_runs / 4)) : worst_unit;
    const int64_t max_blobs = tier == 0 ? std::min<int64_t>(worst_blobs, std::max<int64_t>(4096, total_keys / 32)) : worst_blobs;
    const int64_t max_runs = tile_runs + unit_ids, max_comps = tile_comps + unit_ids;

    int rc_fix = map_fix_mul(m);
    if (rc_fix) return rc_fix;
    Job job;
    memset(&job, 0, sizeof job);
    job.fix_mul = m->fix_mul;
    size_t need = job_carve(job, nullptr, n_planes, total_words, total_keys, max_runs, max_blobs, lab_elems, nullptr, tiles_pp, max_comps);
    Arena arena;
    int rc = arena_get(ctx, need, &arena);
    if (rc) return rc;
    int32_t *labels_dev = nullptr;
    job_carve(job, arena.base, n_planes, total_words, total_keys, max_runs, max_blobs, lab_elems, &labels_dev, tiles_pp, max_comps);
    job.vol_sign[0] = td.sign[0];
    job.vol_sign[1] = td.sign[1];
    job.unit_form = unit_form;
    // the job's number: unique in the process (contexts recycle each other's memory through the driver), started at a random value
    // (so is another process's); stale flags of an earlier job in recycled memory never match it.  Never 0 / the poison pattern.
    static std::atomic<uint32_t> g_epoch{[] {
        uint32_t seed;
        try { seed = (uint32_t)std::random_device{}(); }
        catch (...) { seed = (uint32_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint32_t)getpid() * 2654435761u); }
        return seed | 1u;
    }()};
    do { job.epoch = g_epoch.fetch_add(1u); } while (job.epoch == 0u || job.epoch == 0xffffffffu);

    JobInit init;
    memset(&init, 0, sizeof init);
    for (int p = 0; p < n_planes; ++p) {
        VolDesc &v = init.v[p];
        v.dim[0] = uc; v.dim[1] = ur; v.dim[2] = us;
        v.org[0] = v.org[1] = v.org[2] = 0;
        v.row_words = row_words;
        v.group = p;
        v.word_base = words_pp * p;
        v.key_base = keys_pp * p;
    }
    init.runs0 = (unsigned)tile_runs;
    init.comps0 = (unsigned)tile_comps;
    hipStream_t st = ctx->stream;
    int pair_slots = PAIR_SLOTS;   // test hook: a tiny LDS pair set overflows on small inputs (PDBEDA_DEBUG_EDGE_CAP = slots, a power of two)
    if (ctx->debug_edge_cap > 0) { pair_slots = 1; while (pair_slots * 2 <= std::min<int64_t>(ctx->debug_edge_cap, PAIR_SLOTS)) pair_slots *= 2; }
    switch (td.cw) {
        case 1: launch_tile_label<1>(ctx, (unsigned)tiles_pp, job, m->dens, m->geom_dev, td, init, pair_slots); break;
        case 2: launch_tile_label<2>(ctx, (unsigned)tiles_pp, job, m->dens, m->geom_dev, td, init, pair_slots); break;
        case 3: launch_tile_label<3>(ctx, (unsigned)tiles_pp, job, m->dens, m->geom_dev, td, init, pair_slots); break;
        default: launch_tile_label<4>(ctx, (unsigned)tiles_pp, job, m->dens, m->geom_dev, td, init, pair_slots); break;
    }
    { PROF(ctx, "k_resolve_tiles"); hipLaunchKernelGGL(k_resolve_tiles, dim3((unsigned)tiles_pp + 64u), dim3(256), 0, st, job, (int)tiles_pp); }
    if (job.group_count) { PROF(ctx, "k_group_counts"); hipLaunchKernelGGL(k_group_counts, dim3((unsigned)(job.n_groups + 255) / 256u), dim3(256), 0, st, job); }
    if (labels) {   // the label writer ranks the roots and writes the blob table itself: one launch
        PROF(ctx, "k_labels_tiles");
        launch_labels<true>(ctx, job, td, labels_dev, m->geom_dev);
    } else {
        PROF(ctx, "k_emit_tiles");
        hipLaunchKernelGGL(k_emit_tiles, dim3(std::min<unsigned>(512u, ((unsigned)tiles_pp + 1u) / 2u)), dim3(256), 0, st, job, m->geom_dev);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { arena_put(ctx, arena); return fail(ctx, PDBEDA_ERR_DEVICE, "whole-map labelling launch: %s", hipGetErrorString(e)); }
    out->job = job; out->arena = arena; out->td = td; out->labels_dev = labels_dev; out->labels = labels; out->bytes = need;
    return PDBEDA_OK;
}

static int full_blobs_impl(pdbeda_map *m, float cut_pos, float cut_neg, bool want_pos, bool want_neg, uint32_t flags,
                           pdbeda_bloblist **out_pos, pdbeda_bloblist **out_neg) {
    pdbeda_ctx *ctx = m->ctx;
    WholeMapJob wj;
    const int tier = ctx->debug_worst_case_arena ? 1 : 0, form = tier;   // (the debug hook: the second run's shape at once)
    int rc = whole_map_enqueue(m, cut_pos, cut_neg, want_pos, want_neg, flags, tier, form, &wj);
    if (rc) return rc;
    const int n_planes = (want_pos ? 1 : 0) + (want_neg ? 1 : 0);
    pdbeda_bloblist *first = nullptr;
    for (int p = 0; p < n_planes; ++p) {
        pdbeda_bloblist *bl = new_list(ctx, m);
        bl->job = wj.job;
        bl->td = wj.td;
        bl->vol_lo = p;
        bl->vol_hi = p + 1;
        bl->whole_map = true;
        bl->sign = wj.td.sign[p];
        if (p == 0) { bl->arena = wj.arena; bl->owns_arena = true; first = bl; }
        else { bl->owns_arena = false; bl->sibling = first; first->sibling = bl; }
        bl->labels_dev = wj.labels_dev;
        bl->labels_done = wj.labels;
        bl->cut_pos = cut_pos; bl->cut_neg = cut_neg; bl->want_pos = want_pos; bl->want_neg = want_neg; bl->flags = flags; bl->tier = tier; bl->unit_form = form;
        bl->job_bytes = wj.bytes;
        if (bl->sign > 0) *out_pos = bl; else *out_neg = bl;
    }
    return PDBEDA_OK;
}

extern "C" int pdbeda_full_blobs(pdbeda_map *m, float cutoff, uint32_t flags, pdbeda_bloblist **out) {
    if (!m || !out) return PDBEDA_ERR_ARGUMENT;
    *out = nullptr;
    if (cutoff == 0.0f || cutoff != cutoff) return fail(m->ctx, PDBEDA_ERR_ARGUMENT, "cutoff must be non-zero (the reference returns None)");
    pdbeda_bloblist *dummy = nullptr;
    if (cutoff > 0) return full_blobs_impl(m, cutoff, 0.0f, true, false, flags, out, &dummy);
    return full_blobs_impl(m, 0.0f, cutoff, false, true, flags, &dummy, out);
}

extern "C" int pdbeda_full_blobs_pm(pdbeda_map *m, float cutoff_pos, float cutoff_neg, uint32_t flags, pdbeda_bloblist **green,
                                    pdbeda_bloblist **red) {
    if (!m || !green || !red) return PDBEDA_ERR_ARGUMENT;
    *green = *red = nullptr;
    if (!(cutoff_pos > 0.0f) || !(cutoff_neg < 0.0f)) return fail(m->ctx, PDBEDA_ERR_ARGUMENT, "need cutoff_pos > 0 > cutoff_neg");
    return full_blobs_impl(m, cutoff_pos, cutoff_neg, true, true, flags, green, red);
}

// ---- accessors ------------------------------------------------------------------------
static pdbeda_bloblist *owner_of(pdbeda_bloblist *bl) { return bl->owns_arena ? bl : bl->sibling; }
static const int64_t SPEC_ROWS = 2048;      // rows of a whole-map job's blob table fetched with its counters (84 B each: 172 KB; the analysis entry's 128^3 map has ~1 400 blobs)
static int list_resolve_counts(pdbeda_bloblist *bl) {
    if (bl->have_counts) return 0;
    pdbeda_ctx *ctx = bl->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // rank range of this list inside the job's blob table: a list is the whole job, or one plane of a fused whole-map job
    // (k_emit published the split)
    Counters ctr;
    // a whole-map job: the first rows of its blob table ride along with the counters, kept on the list that owns the job for both lists of a fused call
    // (rows beyond the blob count are stale bytes: never served)
    auto fetch = [&]() -> int {
        pdbeda_bloblist *ow = bl->whole_map ? owner_of(bl) : nullptr;
        const int64_t g = ow ? std::min<int64_t>((int64_t)bl->job.blob_cap, SPEC_ROWS) : 0;
        if (g > 0) {
            const Job &job = bl->job;
            ow->spec_rows = 0;
            ow->spec_n.resize((size_t)g); ow->spec_key.resize((size_t)g); ow->spec_total.resize((size_t)g); ow->spec_centroid.resize(3 * (size_t)g);
            ow->spec_center.resize(3 * (size_t)g); ow->spec_volume.resize((size_t)g); ow->spec_group.resize((size_t)g);
            const D2HItem parts[8] = {{&ctr, job.ctr, sizeof ctr}, {ow->spec_n.data(), job.b_n, (size_t)(8 * g)}, {ow->spec_total.data(), job.b_total, (size_t)(8 * g)},
                                      {ow->spec_centroid.data(), job.b_centroid, (size_t)(24 * g)}, {ow->spec_center.data(), job.b_center, (size_t)(24 * g)},
                                      {ow->spec_volume.data(), job.b_volume, (size_t)(8 * g)}, {ow->spec_key.data(), job.b_key, (size_t)(8 * g)},
                                      {ow->spec_group.data(), job.b_group, (size_t)(4 * g)}};
            HIP_TRY(ctx, d2h_many(ctx, parts, 8));
            HIP_TRY(ctx, ctx_sync(ctx));
            ow->spec_rows = g;
            return 0;
        }
        HIP_TRY(ctx, d2h(ctx, &ctr, bl->job.ctr, sizeof ctr));
        HIP_TRY(ctx, ctx_sync(ctx));
        return 0;
    };
    {
        const int rc_fetch = fetch();
        if (rc_fetch) return rc_fetch;
    }
    while (bl->whole_map && ctr.overflow != 0u) {
        // The job is run again, ONCE (both lists of a fused call move to the new job), when the typical-size arena was too small
        // for this map (bits 0 / 1), or when the map has unit tiles and the job was enqueued without their two launches (bit 2,
        // Job::unit_form).  The second run has both: the worst-case arena and the unit launches -- a first run without the unit
        // launches does not know how many ids its unit tiles would have asked for.
        pdbeda_bloblist *ow = owner_of(bl);
        if (!ow) return fail(ctx, PDBEDA_ERR_STATE, "whole-map labelling: a list without its job");
        const int tier = 1, form = 1;
        if (tier == ow->tier && form == ow->unit_form) return fail(ctx, PDBEDA_ERR_STATE, "whole-map labelling overflowed its worst-case arena");
        if (ow->voxels_done || bl->voxels_done) return fail(ctx, PDBEDA_ERR_STATE, "whole-map labelling: overflow noticed after the voxel lists were made");
        WholeMapJob wj;
        arena_put(ctx, ow->arena);                         // (stream order: the first run's kernels are done -- ctx_sync above)
        int rc = whole_map_enqueue(ow->map, ow->cut_pos, ow->cut_neg, ow->want_pos, ow->want_neg, ow->flags, tier, form, &wj);
        if (rc) { ow->arena.base = nullptr; ow->arena.cap = 0; return rc; }
        ow->arena = wj.arena;
        pdbeda_bloblist *both[2] = {ow, ow->sibling};
        for (pdbeda_bloblist *l : both) {
            if (!l) continue;
            l->job = wj.job; l->td = wj.td; l->labels_dev = wj.labels_dev; l->labels_done = wj.labels;
            l->tier = tier; l->unit_form = form; l->reruns += 1; l->have_counts = false; l->job_bytes = wj.bytes; l->spec_rows = 0;
        }
        const int rc_fetch = fetch();
        if (rc_fetch) return rc_fetch;
    }
    if (!bl->whole_map && ctr.unit_wait_failed) return fail(ctx, PDBEDA_ERR_DEVICE, "sphere batch: the device's volumes outgrew what the host sized the job for");
    if (bl->vol_lo == 0 && bl->vol_hi == bl->job.n_vols) { bl->rank_lo = 0; bl->rank_hi = ctr.n_blobs; }
    else if (bl->whole_map && bl->job.n_vols == 2 && bl->vol_hi == bl->vol_lo + 1) {
        bl->rank_lo = bl->vol_lo == 0 ? 0 : ctr.n_blobs_vol0;
        bl->rank_hi = bl->vol_lo == 0 ? ctr.n_blobs_vol0 : ctr.n_blobs;
    } else return fail(ctx, PDBEDA_ERR_STATE, "blob list covers an unexpected volume range");
    bl->job_blobs = ctr.n_blobs;
    bl->have_counts = true;
    // the other list of a fused call reads the same counters: its range is known now too
    if (bl->whole_map && bl->job.n_vols == 2) {
        pdbeda_bloblist *ow = owner_of(bl);
        pdbeda_bloblist *both[2] = {ow, ow ? ow->sibling : nullptr};
        for (pdbeda_bloblist *l : both) {
            if (!l || l == bl || l->freed || l->have_counts || l->job.ctr != bl->job.ctr || l->vol_hi != l->vol_lo + 1) continue;
            l->rank_lo = l->vol_lo == 0 ? 0 : ctr.n_blobs_vol0;
            l->rank_hi = l->vol_lo == 0 ? ctr.n_blobs_vol0 : ctr.n_blobs;
            l->job_blobs = ctr.n_blobs;
            l->have_counts = true;
        }
    }
    return 0;
}

// Diagnostic: device counters of the labelling job behind a list:
// out[0..7] = run ids, component ids, runs of the job (1 = the typical-size arena overflowed and the job ran again), blobs,
// unit tiles by cause (run slots, -, component table), bytes of the job's arena.
extern "C" int pdbeda_bloblist_counters(pdbeda_bloblist *bl, int64_t *out) {
    if (!bl || bl->freed || !out) return PDBEDA_ERR_ARGUMENT;
    pdbeda_ctx *ctx = bl->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Counters c;
    HIP_TRY(ctx, d2h(ctx, &c, bl->job.ctr, sizeof c));
    HIP_TRY(ctx, ctx_sync(ctx));
    out[0] = c.n_runs; out[1] = c.n_comps; out[2] = bl->reruns; out[3] = c.n_blobs;
    out[4] = out[5] = out[6] = 0;
    out[7] = (int64_t)(bl->job_bytes ? bl->job_bytes : owner_of(bl)->arena.cap);   // bytes of device memory the job needs (its arena, if recycled, may be up to twice that)
    if (bl->whole_map && bl->job.tile_mode) {   // tiles off the fast path by kind: tile_mode 1 = unit tile (run slots / values), 2 = wide tile (more than CCAP components, united in LDS all the same), 3 = unit tile (no ids left for a wide tile's components)
        const int64_t n_tiles = (int64_t)bl->td.ctiles * bl->td.rtiles * bl->td.stiles;
        std::vector<uint8_t> mode(n_tiles);
        HIP_TRY(ctx, d2h(ctx, mode.data(), bl->job.tile_mode, n_tiles));
        HIP_TRY(ctx, ctx_sync(ctx));
        for (uint8_t v : mode) { if (v == 1) ++out[4]; else if (v == 2) ++out[5]; else if (v == 3) ++out[6]; }
    }
    return PDBEDA_OK;
}


extern "C" int64_t pdbeda_bloblist_count(pdbeda_bloblist *bl) {
    if (!bl || bl->freed) return PDBEDA_ERR_ARGUMENT;
    int rc = list_resolve_counts(bl);
    if (rc) return rc;
    return bl->rank_hi - bl->rank_lo;
}

extern "C" int pdbeda_bloblist_stats(pdbeda_bloblist *bl, int64_t *n, double *total_density, double *centroid, double *coord_center,
                                     double *volume, int64_t *first_key, int32_t *group) {
    if (!bl || bl->freed) return PDBEDA_ERR_ARGUMENT;
    int rc = list_resolve_counts(bl);
    if (rc) return rc;
    pdbeda_ctx *ctx = bl->ctx;
    const Job &job = bl->job;
    const int64_t lo = bl->rank_lo, cnt = bl->rank_hi - bl->rank_lo;
    if (cnt == 0) return PDBEDA_OK;
    if (bl->whole_map) {      // the rows that came with the counters
        const pdbeda_bloblist *ow = owner_of(bl);
        if (ow && ow->spec_rows >= lo + cnt && ow->job.ctr == job.ctr) {
            if (n) memcpy(n, ow->spec_n.data() + lo, (size_t)(8 * cnt));
            if (total_density) memcpy(total_density, ow->spec_total.data() + lo, (size_t)(8 * cnt));
            if (centroid) memcpy(centroid, ow->spec_centroid.data() + 3 * lo, (size_t)(24 * cnt));
            if (coord_center) memcpy(coord_center, ow->spec_center.data() + 3 * lo, (size_t)(24 * cnt));
            if (volume) memcpy(volume, ow->spec_volume.data() + lo, (size_t)(8 * cnt));
            if (first_key) memcpy(first_key, ow->spec_key.data() + lo, (size_t)(8 * cnt));
            if (group) memcpy(group, ow->spec_group.data() + lo, (size_t)(4 * cnt));
            return PDBEDA_OK;
        }
    }
    const D2HItem columns[7] = {{n, job.b_n + lo, (size_t)(8 * cnt)}, {total_density, job.b_total + lo, (size_t)(8 * cnt)},
                                {centroid, job.b_centroid + 3 * lo, (size_t)(24 * cnt)}, {coord_center, job.b_center + 3 * lo, (size_t)(24 * cnt)},
                                {volume, job.b_volume + lo, (size_t)(8 * cnt)}, {first_key, job.b_key + lo, (size_t)(8 * cnt)},
                                {group, job.b_group + lo, (size_t)(4 * cnt)}};
    HIP_TRY(ctx, d2h_many(ctx, columns, 7));      // (packed on the device: one copy instead of seven)
    HIP_TRY(ctx, ctx_sync(ctx));
    return PDBEDA_OK;
}

// Counters AND the first `guess` rows of a batch job's blob table in ONE host round trip (the batches of pdbeda_aggregate_cloud:
// a count and then the table were two waits each; under four workers on one GPU a wait costs a quarter of a millisecond).
// Rows beyond the blob count are stale arena bytes, copied and dropped.  Falls back to the two-step path when the guess was
// too small (or the list is not a whole batch job).
static int list_stats_one_trip(pdbeda_bloblist *bl, int64_t guess, std::vector<int64_t> &n, std::vector<double> &tot, std::vector<double> &cen,
                               std::vector<int32_t> &grp, const D2HItem *extra = nullptr) {      // (extra: one more result of the caller's, in the same pack and wait)
    pdbeda_ctx *ctx = bl->ctx;
    const Job &job = bl->job;
    guess = std::min<int64_t>(guess, (int64_t)job.blob_cap);
    const bool whole_job = !bl->whole_map && bl->vol_lo == 0 && bl->vol_hi == job.n_vols;
    const int64_t extra_bytes = extra ? (int64_t)extra->bytes + 64 : 0;
    if (!bl->have_counts && whole_job && guess > 0 && 44 * guess + 4096 + extra_bytes < (int64_t)ctx->pinned_cap - (int64_t)ctx->pinned_used) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        n.resize((size_t)guess); tot.resize((size_t)guess); cen.resize(3 * (size_t)guess); grp.resize((size_t)guess);
        Counters ctr;
        const D2HItem parts[6] = {{&ctr, job.ctr, sizeof ctr}, {n.data(), job.b_n, (size_t)(8 * guess)}, {tot.data(), job.b_total, (size_t)(8 * guess)},
                                  {cen.data(), job.b_centroid, (size_t)(24 * guess)}, {grp.data(), job.b_group, (size_t)(4 * guess)},
                                  extra ? *extra : D2HItem{nullptr, nullptr, 0}};
        HIP_TRY(ctx, d2h_many(ctx, parts, 6));
        HIP_TRY(ctx, ctx_sync(ctx));
        if (ctr.unit_wait_failed) return fail(ctx, PDBEDA_ERR_DEVICE, "sphere batch: the device's volumes outgrew what the host sized the job for");
        bl->rank_lo = 0; bl->rank_hi = ctr.n_blobs; bl->job_blobs = ctr.n_blobs; bl->have_counts = true;
        if ((int64_t)ctr.n_blobs <= guess) {
            const size_t nb = ctr.n_blobs;
            n.resize(nb); tot.resize(nb); cen.resize(3 * nb); grp.resize(nb);
            return PDBEDA_OK;
        }
        extra = nullptr;      // (the guess was too small for the table; the caller's extra result has arrived with this first wait all the same)
    }
    if (extra && extra->dst && extra->bytes) HIP_TRY(ctx, d2h(ctx, extra->dst, extra->src, extra->bytes));      // (delivered by the waits below)
    const int64_t nb = pdbeda_bloblist_count(bl);
    if (nb < 0) return (int)nb;
    n.resize((size_t)nb); tot.resize((size_t)nb); cen.resize(3 * (size_t)nb); grp.resize((size_t)nb);
    return pdbeda_bloblist_stats(bl, n.data(), tot.data(), cen.data(), nullptr, nullptr, nullptr, grp.data());
}

// Voxel lists are materialised once per JOB (shared by the lists of a fused call through
// the owning list).

static int list_materialise_voxels(pdbeda_bloblist *bl) {
    pdbeda_bloblist *ow = owner_of(bl);
    if (ow->voxels_done) return 0;
    pdbeda_ctx *ctx = bl->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {   // the job's blob count: one read of the counters per list, not one more here (every caller has resolved its counts;
        // a whole-map job that overflowed its typical-size arena has been run again by then)
        const int rc_counts = list_resolve_counts(bl);
        if (rc_counts) return rc_counts;
        ow = owner_of(bl);
    }
    Job &job = ow->job;
    const int64_t nb = bl->job_blobs;
    // total voxels unknown until the offsets scan; bound by key bits
    const int64_t max_vox = job.key_words * 64;
    size_t need = align_up(8 * (nb + 1)) + align_up(4 * std::max<int64_t>(nb, 1)) + align_up(12 * std::max<int64_t>(max_vox, 1));
    int rc = arena_get(ctx, need, &ow->vox_arena);
    if (rc) return rc;
    Carver cv(ow->vox_arena.base);
    ow->offsets_dev = cv.take<int64_t>(nb + 1);
    ow->cursor_dev = cv.take<unsigned int>(std::max<int64_t>(nb, 1));
    ow->crs_dev = cv.take<int32_t>(3 * std::max<int64_t>(max_vox, 1));
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_blob_offsets, dim3(1), dim3(1024), 0, st, job, ow->offsets_dev, ow->cursor_dev);
    if (job.total_words > 0)
        hipLaunchKernelGGL(k_voxel_lists, dim3(grid_for(job.total_words * 64, 256, 8192)), dim3(256), 0, st, job, ow->offsets_dev,
                           ow->cursor_dev, ow->crs_dev);
    HIP_TRY(ctx, hipGetLastError());
    ow->voxels_done = true;      // (enqueued: whatever reads the lists is ordered behind them on the stream, and waits there)
    return 0;
}

extern "C" int64_t pdbeda_bloblist_num_voxels(pdbeda_bloblist *bl) {
    if (!bl || bl->freed) return PDBEDA_ERR_ARGUMENT;
    int rc = list_resolve_counts(bl);
    if (rc) return rc;
    if (bl->n_voxels >= 0) return bl->n_voxels;
    rc = list_materialise_voxels(bl);
    if (rc) return rc;
    pdbeda_ctx *ctx = bl->ctx;
    pdbeda_bloblist *ow = owner_of(bl);
    int64_t off[2] = {0, 0};
    HIP_TRY(ctx, d2h(ctx, &off[0], ow->offsets_dev + bl->rank_lo, 8));
    HIP_TRY(ctx, d2h(ctx, &off[1], ow->offsets_dev + bl->rank_hi, 8));
    HIP_TRY(ctx, ctx_sync(ctx));
    bl->n_voxels = off[1] - off[0];
    return bl->n_voxels;
}

extern "C" int pdbeda_bloblist_voxels(pdbeda_bloblist *bl, int32_t *crs, int64_t *blob_offsets) {
    if (!bl || bl->freed) return PDBEDA_ERR_ARGUMENT;
    int64_t nv = pdbeda_bloblist_num_voxels(bl);
    if (nv < 0) return (int)nv;
    pdbeda_ctx *ctx = bl->ctx;
    pdbeda_bloblist *ow = owner_of(bl);
    const int64_t cnt = bl->rank_hi - bl->rank_lo;
    std::vector<int64_t> off(cnt + 1);
    HIP_TRY(ctx, d2h(ctx, off.data(), ow->offsets_dev + bl->rank_lo, 8 * (cnt + 1))); HIP_TRY(ctx, ctx_sync(ctx));
    const int64_t base = off[0];
    if (blob_offsets)
        for (int64_t i = 0; i <= cnt; ++i) blob_offsets[i] = off[i] - base;
    if (crs && nv > 0) HIP_TRY(ctx, d2h(ctx, crs, ow->crs_dev + 3 * base, 12 * nv)); HIP_TRY(ctx, ctx_sync(ctx));
    return PDBEDA_OK;
}

extern "C" int pdbeda_bloblist_labels(pdbeda_bloblist *bl, int32_t *labels_host) {
    if (!bl || bl->freed || !labels_host) return PDBEDA_ERR_ARGUMENT;
    if (!bl->whole_map) return fail(bl->ctx, PDBEDA_ERR_STATE, "dense labels exist only for whole-map blob lists");
    pdbeda_ctx *ctx = bl->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {   // a job whose unit tiles waited in vain (k_face_merge) has no valid labels either: the same check the blob table makes
        const int rc_counts = list_resolve_counts(bl);
        if (rc_counts) return rc_counts;
    }
    const Geom &g = bl->map->geom;
    const int uc = g.unique_ncrs[0], ur = g.unique_ncrs[1], us = g.unique_ncrs[2];
    const int64_t nvox = (int64_t)uc * ur * us;
    // scratch: [signed volume if it was not requested at labelling time] + decoded volume
    const bool have = bl->labels_dev && bl->labels_done;
    Arena a;
    int rc = arena_get(ctx, align_up(4 * nvox) * (have ? 1 : 2), &a);
    if (rc) return rc;
    int32_t *decoded = (int32_t *)a.base;
    const int32_t *signed_vol = bl->labels_dev;
    if (!have) {
        int32_t *tmp = (int32_t *)(a.base + align_up(4 * nvox));
        launch_labels<false>(ctx, bl->job, bl->td, tmp, bl->map->geom_dev);
        signed_vol = tmp;
    }
    hipLaunchKernelGGL(k_labels_decode, dim3(grid_for(nvox, 256, 4096)), dim3(256), 0, ctx->stream, signed_vol, nvox, bl->sign, decoded);
    hipError_t e = d2h(ctx, labels_host, decoded, 4 * nvox);
    if (e == hipSuccess) e = ctx_sync(ctx);
    arena_put(ctx, a);
    if (e != hipSuccess) return fail(ctx, PDBEDA_ERR_DEVICE, "labels: %s", hipGetErrorString(e));
    return PDBEDA_OK;
}

extern "C" int pdbeda_bloblist_free(pdbeda_bloblist *bl) {
    if (!bl) return PDBEDA_ERR_ARGUMENT;
    if (bl->freed) return PDBEDA_ERR_STATE;
    pdbeda_ctx *ctx = bl->ctx;
    bl->freed = true;
    ctx->live_handles--;
    pdbeda_bloblist *ow = owner_of(bl);
    pdbeda_bloblist *other = bl->sibling;
    const bool other_alive = other && !other->freed;
    if (!other_alive) {
        // last list of the job.  Arenas are recycled only inside this context, i.e. by work that
        // is enqueued later on the SAME stream, so stream order protects them: no host sync.
        arena_put(ctx, ow->arena);
        arena_put(ctx, ow->vox_arena);
        if (other) delete other;
        delete bl;
    }
    return PDBEDA_OK;
}

// ------------------------------------------------------------------------------------
// Sphere / list batches
// ------------------------------------------------------------------------------------
struct GroupSetup {
    Arena in_arena;          // inputs + boxes + bounds (scratch, released after painting)
    double *d_xyz = nullptr;
    float *d_radii = nullptr;
    int32_t *d_item_group = nullptr;
    AtomBox *d_boxes = nullptr;
    int32_t *d_crs = nullptr;
    int32_t *g_lo = nullptr, *g_hi = nullptr;
    VolDesc *d_vols = nullptr;
    Counters *d_ctr = nullptr;
    int64_t total_words = 0, total_keys = 0;
    bool host_totals = false;   // the totals are the host's (per-atom spheres): nobody waited for the device's -- k_make_vols checks them
    std::vector<AtomBox> host_boxes;   // per-atom spheres: the boxes the host made (kept on the list: aggregateCloud's union volumes are boxes around them)
    bool atom_engine = false;   // per-atom spheres whose boxes the host made, every box one mask word a row and at most ATOM_WORDS rows: ONE launch labels the batch (k_atom_engine)
    // staged inputs whose copy into the device scratch has not been launched yet: grouped_job's k_job_init does it (one launch less), flush_pending for everybody else
    const char *pend_src = nullptr;
    char *pend_dst = nullptr;
    size_t pend_bytes = 0;
};

static hipError_t flush_pending(pdbeda_ctx *ctx, GroupSetup *gs) {
    if (!gs->pend_bytes) return hipSuccess;
    const hipError_t e = copy_by_kernel(ctx, gs->pend_dst, gs->pend_src, gs->pend_bytes);
    gs->pend_bytes = 0;
    return e;
}

static int expand_groups(const int64_t *group_offsets, int64_t n_groups, int64_t n_items, std::vector<int32_t> &item_group) {
    item_group.assign((size_t)n_items, 0);
    if (group_offsets[0] != 0 || group_offsets[n_groups] != n_items) return -1;
    for (int64_t g = 0; g < n_groups; ++g) {
        if (group_offsets[g + 1] < group_offsets[g]) return -1;
        for (int64_t i = group_offsets[g]; i < group_offsets[g + 1]; ++i) item_group[(size_t)i] = (int32_t)g;
    }
    return 0;
}

// Scratch of a sphere / list batch: inputs, boxes, group bounds, volume descriptors.
static int group_alloc(pdbeda_ctx *ctx, int64_t n_items, int64_t n_groups, GroupSetup *gs) {
    if (n_groups >= (1ll << 31) || n_items >= (1ll << 40)) return fail(ctx, PDBEDA_ERR_ARGUMENT, "batch too large");
    const int64_t ni = std::max<int64_t>(n_items, 1), ng = std::max<int64_t>(n_groups, 1);
    size_t need = align_up(24 * ni) + align_up(4 * ni) + align_up(4 * ni) + align_up(sizeof(AtomBox) * ni) + align_up(12 * ni) +
                  2 * align_up(12 * ng) + align_up(sizeof(VolDesc) * ng) + align_up(sizeof(Counters));
    int rc = arena_get(ctx, need, &gs->in_arena);
    if (rc) return rc;
    Carver cv(gs->in_arena.base);
    gs->d_xyz = cv.take<double>(3 * ni);
    gs->d_radii = cv.take<float>(ni);
    gs->d_item_group = cv.take<int32_t>(ni);
    gs->d_boxes = cv.take<AtomBox>(ni);
    gs->d_vols = cv.take<VolDesc>(ng);       // (boxes, volumes and counters follow the inputs: a per-atom sphere batch whose bounds the host makes sends all six in one row)
    gs->d_ctr = cv.take<Counters>(1);
    gs->d_crs = cv.take<int32_t>(3 * ni);
    gs->g_lo = cv.take<int32_t>(3 * ng);
    gs->g_hi = cv.take<int32_t>(3 * ng);
    return 0;
}

// Inputs are on the device (d_xyz + d_radii, or d_crs; d_item_group): group bounding volumes, volume descriptors, and the
// totals read back (the one host round trip of a sphere / list batch).
static int group_bounds(pdbeda_map *m, GroupSetup *gs, int64_t n_items, int64_t n_groups, bool spheres, const int64_t *host_totals = nullptr) {
    pdbeda_ctx *ctx = m->ctx;
    hipStream_t st = ctx->stream;
    const int64_t ng = std::max<int64_t>(n_groups, 1);
    { PROF(ctx, "k_init_bounds"); hipLaunchKernelGGL(k_init_bounds, dim3(grid_for(3 * ng, 256)), dim3(256), 0, st, gs->g_lo, gs->g_hi, 3 * ng, gs->d_ctr); }
    if (n_items > 0) {
        if (spheres)
            { PROF(ctx, "k_atom_boxes"); hipLaunchKernelGGL(k_atom_boxes, dim3(grid_for(n_items, 256)), dim3(256), 0, st, m->geom_dev, gs->d_xyz, gs->d_radii,
                               gs->d_item_group, n_items, gs->d_boxes, gs->g_lo, gs->g_hi); }
        else
            { PROF(ctx, "k_list_boxes"); hipLaunchKernelGGL(k_list_boxes, dim3(grid_for(n_items, 256 * 4, 256)), dim3(256), 0, st, gs->d_crs, gs->d_item_group, n_items,
                               gs->g_lo, gs->g_hi); }
    }
    { PROF(ctx, "k_make_vols"); hipLaunchKernelGGL(k_make_vols, dim3(1), dim3(1024), 0, st, gs->g_lo, gs->g_hi, (int)n_groups, gs->d_vols, gs->d_ctr,
                                                   host_totals ? (long long)host_totals[0] : LLONG_MAX, host_totals ? (long long)host_totals[1] : LLONG_MAX); }
    HIP_TRY(ctx, hipGetLastError());
    if (host_totals) {   // sized by the host: no round trip (the kernel holds the device's totals against these)
        gs->total_words = host_totals[0];
        gs->total_keys = host_totals[1];
        gs->host_totals = true;
        if (gs->total_words >= (1ll << 31) * 2) return fail(ctx, PDBEDA_ERR_ARGUMENT, "sphere batch too large (%lld mask words)", (long long)gs->total_words);
        return 0;
    }
    Counters ctr;
    HIP_TRY(ctx, d2h(ctx, &ctr, gs->d_ctr, sizeof ctr));
    HIP_TRY(ctx, ctx_sync(ctx));  // (host-side staging vectors of the caller are also safe to drop now)
    gs->total_words = ctr.total_words;
    gs->total_keys = ctr.total_keys;
    if (gs->total_words >= (1ll << 31) * 2) return fail(ctx, PDBEDA_ERR_ARGUMENT, "sphere batch too large (%lld mask words)", (long long)gs->total_words);
    return 0;
}

// Upload atoms (or explicit voxels), then group_bounds.
static int group_setup(pdbeda_map *m, const double *xyz, const float *radii, const int32_t *crs, int64_t n_items,
                       const int64_t *group_offsets, int64_t n_groups, GroupSetup *gs) {
    pdbeda_ctx *ctx = m->ctx;
    std::vector<int32_t> item_group;
    if (expand_groups(group_offsets, n_groups, n_items, item_group)) return fail(ctx, PDBEDA_ERR_ARGUMENT, "bad group_offsets");
    int rc = group_alloc(ctx, n_items, n_groups, gs);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    // Per-atom spheres (a group per atom: the clouds of aggregateCloud, the per-atom region tables): an atom's box is
    // [C - R - 1, C + R] with R = xyz2crs(origin + radius) -- its SIZE follows from the radius alone, so the host knows the
    // job's mask words and keys without asking the device (the round trip for the totals was one of an entry's host waits).
    // The inputs go through the pinned staging buffer, so nothing here needs the caller's arrays after the call returns.
    // Round 6: GROUPED spheres (a residue's atoms: the residue region tables, RSCC / RSR) get their boxes and volumes from the host too -- a group's
    // volume is the box around its atoms' boxes, which is what k_atom_boxes' atomic min / max and k_make_vols make of them on the device (the same
    // xyz2crs, the same int32 arithmetic): k_init_bounds, k_atom_boxes, k_make_vols, the copy of their totals and the WAIT for it go (four launches and
    // one of an entry's waits per grouped call).  PDBEDA_HOST_BOXES=0: the device makes them, as for the per-atom batches.
    if (xyz && n_items > 0 && n_groups > 0 && n_groups < n_items) {
        static const bool host_boxes_g = [] { const char *e = getenv("PDBEDA_HOST_BOXES"); return !(e && e[0] == '0'); }();
        bool sane = true;
        for (int64_t a = 0; a < n_items && sane; ++a) sane = radii[a] >= 0.0f && std::isfinite(radii[a]);
        const size_t row = (size_t)((char *)(gs->d_ctr + 1) - gs->in_arena.base), row_need = (row + 63) & ~(size_t)63;
        if (sane && host_boxes_g && copy_kernels() && !ctx->debug_shrink_totals && (char *)gs->d_xyz == gs->in_arena.base && ctx->pinned && row <= ((size_t)1 << 20) &&
            ctx->pinned_used + row_need <= ctx->pinned_cap) {
            char *stage = ctx->pinned + ctx->pinned_used;
            memcpy(stage + ((char *)gs->d_xyz - gs->in_arena.base), xyz, 24 * (size_t)n_items);
            memcpy(stage + ((char *)gs->d_radii - gs->in_arena.base), radii, 4 * (size_t)n_items);
            memcpy(stage + ((char *)gs->d_item_group - gs->in_arena.base), item_group.data(), 4 * (size_t)n_items);
            AtomBox *boxes = reinterpret_cast<AtomBox *>(stage + ((char *)gs->d_boxes - gs->in_arena.base));
            VolDesc *vols = reinterpret_cast<VolDesc *>(stage + ((char *)gs->d_vols - gs->in_arena.base));
            Counters *ctr0 = reinterpret_cast<Counters *>(stage + ((char *)gs->d_ctr - gs->in_arena.base));
            std::vector<int32_t> glo(3 * (size_t)n_groups, INT32_MAX), ghi(3 * (size_t)n_groups, INT32_MIN);
            float cached_rad = NAN;
            int32_t R[3] = {0, 0, 0};
            for (int64_t a = 0; a < n_items; ++a) {
                if (!(radii[a] == cached_rad)) {
                    const double rad = (double)radii[a];
                    const double o[3] = {m->geom.origin[0] + rad, m->geom.origin[1] + rad, m->geom.origin[2] + rad};
                    xyz2crs(m->geom, o, R);
                    cached_rad = radii[a];
                }
                int32_t C[3];
                xyz2crs(m->geom, xyz + 3 * a, C);
                AtomBox bx;
                bool empty = false;
                for (int k = 0; k < 3; ++k) {      // (int32 arithmetic as in k_atom_boxes)
                    bx.lo[k] = C[k] - R[k] - 1;
                    bx.hi[k] = C[k] + R[k];
                    empty = empty || bx.hi[k] < bx.lo[k];
                }
                if (empty) { for (int k = 0; k < 3; ++k) { bx.lo[k] = 0; bx.hi[k] = -1; } }
                boxes[a] = bx;
                if (!empty) {
                    const size_t g = (size_t)item_group[(size_t)a];
                    for (int k = 0; k < 3; ++k) { glo[3 * g + k] = std::min(glo[3 * g + k], bx.lo[k]); ghi[3 * g + k] = std::max(ghi[3 * g + k], bx.hi[k]); }
                }
            }
            int64_t words = 0, keys = 0;
            bool fits = true;
            for (int64_t g = 0; g < n_groups; ++g) {      // (k_make_vols)
                VolDesc vd;
                memset(&vd, 0, sizeof vd);
                bool empty = false;
                for (int k = 0; k < 3; ++k) empty = empty || ghi[3 * (size_t)g + k] < glo[3 * (size_t)g + k];
                for (int k = 0; k < 3 && !empty; ++k) {
                    const int64_t d = (int64_t)ghi[3 * (size_t)g + k] - glo[3 * (size_t)g + k] + 1;
                    fits = fits && d < (1ll << 30);
                    vd.org[k] = glo[3 * (size_t)g + k];
                    vd.dim[k] = (int32_t)d;
                }
                vd.row_words = (vd.dim[0] + 63) / 64;
                vd.group = (int32_t)g;
                vd.word_base = words;
                vd.key_base = keys;
                vols[g] = vd;
                words += (int64_t)vd.row_words * vd.dim[1] * vd.dim[2];
                keys += (int64_t)vd.dim[0] * vd.dim[1] * vd.dim[2];
                fits = fits && words < (1ll << 40) && keys < (1ll << 46);
            }
            if (fits) {
                memset(ctr0, 0, sizeof *ctr0);
                ctr0->total_words = words;
                ctr0->total_keys = keys;
                gs->pend_src = stage; gs->pend_dst = gs->in_arena.base; gs->pend_bytes = (row + 15) & ~(size_t)15;
                ctx->pinned_used += row_need;
                gs->total_words = words;
                gs->total_keys = keys;
                gs->host_totals = true;
                if (gs->total_words >= (1ll << 31) * 2) return fail(ctx, PDBEDA_ERR_ARGUMENT, "sphere batch too large (%lld mask words)", (long long)gs->total_words);
                return 0;
            }
        }
    }
    if (xyz && n_items > 0 && n_groups == n_items) {
        bool per_atom = true;
        for (int64_t g = 0; g <= n_groups && per_atom; ++g) per_atom = group_offsets[g] == g;
        int64_t totals[2] = {0, 0};
        float last_rad = NAN;
        int64_t last_words = 0, last_keys = 0;
        for (int64_t a = 0; a < n_items && per_atom; ++a) {
            const float rad = radii[a];
            if (!(rad >= 0.0f) || !std::isfinite(rad)) { per_atom = false; break; }
            if (!(rad == last_rad)) {   // (a handful of distinct radii: one per atom type)
                const double o[3] = {m->geom.origin[0] + (double)rad, m->geom.origin[1] + (double)rad, m->geom.origin[2] + (double)rad};
                int32_t R[3];
                xyz2crs(m->geom, o, R);
                int64_t dim[3];
                bool empty = false;
                for (int k = 0; k < 3; ++k) { dim[k] = 2 * (int64_t)R[k] + 2; empty = empty || dim[k] <= 0; }
                last_words = empty ? 0 : (dim[0] + 63) / 64 * dim[1] * dim[2];
                last_keys = empty ? 0 : dim[0] * dim[1] * dim[2];
                last_rad = rad;
            }
            totals[0] += last_words;
            totals[1] += last_keys;
            if (totals[0] >= (1ll << 40) || totals[1] >= (1ll << 46)) per_atom = false;   // (absurd: let the waiting path report it)
        }
        // Round 5: the host makes the boxes and the volume descriptors too (xyz2crs is one function for host and device: the same IEEE operations in
        // the same order) -- k_init_bounds, k_atom_boxes and k_make_vols were three launches in front of every per-atom batch (the clouds of
        // aggregateCloud, the region tables), and a launch costs several times its 5 us when other processes' uploads hold the link.
        // PDBEDA_HOST_BOXES=0: the device makes them (A/B, and what the test of the two paths' equality runs against).
        static const bool host_boxes = [] { const char *e = getenv("PDBEDA_HOST_BOXES"); return !(e && e[0] == '0'); }();
        const size_t row = (size_t)((char *)(gs->d_ctr + 1) - gs->in_arena.base), row_need = (row + 63) & ~(size_t)63;
        if (per_atom && host_boxes && !ctx->debug_shrink_totals && (char *)gs->d_xyz == gs->in_arena.base && ctx->pinned && row <= ((size_t)1 << 20) &&
            ctx->pinned_used + row_need <= ctx->pinned_cap) {
            char *stage = ctx->pinned + ctx->pinned_used;
            memcpy(stage + ((char *)gs->d_xyz - gs->in_arena.base), xyz, 24 * (size_t)n_items);
            memcpy(stage + ((char *)gs->d_radii - gs->in_arena.base), radii, 4 * (size_t)n_items);
            memcpy(stage + ((char *)gs->d_item_group - gs->in_arena.base), item_group.data(), 4 * (size_t)n_items);
            AtomBox *boxes = reinterpret_cast<AtomBox *>(stage + ((char *)gs->d_boxes - gs->in_arena.base));
            VolDesc *vols = reinterpret_cast<VolDesc *>(stage + ((char *)gs->d_vols - gs->in_arena.base));
            Counters *ctr0 = reinterpret_cast<Counters *>(stage + ((char *)gs->d_ctr - gs->in_arena.base));
            int64_t words = 0, keys = 0;
            float cached_rad = NAN;
            int32_t R[3] = {0, 0, 0};
            bool small_boxes = true;
            gs->host_boxes.resize((size_t)n_items);
            for (int64_t a = 0; a < n_items; ++a) {
                if (!(radii[a] == cached_rad)) {
                    const double rad = (double)radii[a];
                    const double o[3] = {m->geom.origin[0] + rad, m->geom.origin[1] + rad, m->geom.origin[2] + rad};
                    xyz2crs(m->geom, o, R);
                    cached_rad = radii[a];
                }
                int32_t C[3];
                xyz2crs(m->geom, xyz + 3 * a, C);
                AtomBox bx;
                bool empty = false;
                for (int k = 0; k < 3; ++k) {      // (int32 arithmetic as in k_atom_boxes)
                    bx.lo[k] = C[k] - R[k] - 1;
                    bx.hi[k] = C[k] + R[k];
                    empty = empty || bx.hi[k] < bx.lo[k];
                }
                if (empty) { for (int k = 0; k < 3; ++k) { bx.lo[k] = 0; bx.hi[k] = -1; } }
                boxes[a] = bx;
                gs->host_boxes[(size_t)a] = bx;
                VolDesc vd;
                memset(&vd, 0, sizeof vd);
                for (int k = 0; k < 3 && !empty; ++k) { vd.org[k] = bx.lo[k]; vd.dim[k] = bx.hi[k] - bx.lo[k] + 1; }
                vd.row_words = (vd.dim[0] + 63) / 64;
                vd.group = (int32_t)a;
                vd.word_base = words;
                vd.key_base = keys;
                vols[a] = vd;
                small_boxes = small_boxes && vd.row_words <= 1 && (int64_t)vd.dim[1] * vd.dim[2] <= ATOM_WORDS;
                words += (int64_t)vd.row_words * vd.dim[1] * vd.dim[2];
                keys += (int64_t)vd.dim[0] * vd.dim[1] * vd.dim[2];
            }
            if (!(words == totals[0] && keys == totals[1])) gs->host_boxes.clear();
            if (words == totals[0] && keys == totals[1]) {      // (they are: both follow from the radii; a mismatch takes the device's path below)
                memset(ctr0, 0, sizeof *ctr0);
                ctr0->total_words = words;
                ctr0->total_keys = keys;
                if (copy_kernels()) { gs->pend_src = stage; gs->pend_dst = gs->in_arena.base; gs->pend_bytes = (row + 15) & ~(size_t)15; }   // (copied by the job's first launch; whole 16-byte units: the stage and the carve are padded)
                else HIP_TRY(ctx, hipMemcpyAsync(gs->in_arena.base, stage, row, hipMemcpyHostToDevice, st));
                ctx->pinned_used += row_need;
                gs->total_words = words;
                gs->total_keys = keys;
                gs->host_totals = true;
                static const bool atom_engine_on = [] { const char *e = getenv("PDBEDA_ATOM_ENGINE"); return !(e && e[0] == '0'); }();   // (A/B switch)
                gs->atom_engine = small_boxes && atom_engine_on && n_items < (1ll << 31);
                if (gs->total_words >= (1ll << 31) * 2) return fail(ctx, PDBEDA_ERR_ARGUMENT, "sphere batch too large (%lld mask words)", (long long)gs->total_words);
                return 0;
            }
        }
        if (ctx->debug_shrink_totals) { totals[0] /= 2; totals[1] /= 2; }
        // (coordinates, radii and groups sit in a row at the head of the scratch arena: one staged block, one copy)
        const size_t block = (size_t)((char *)(gs->d_item_group + n_items) - gs->in_arena.base);
        const size_t need = (block + 63) & ~(size_t)63;
        if (per_atom && (char *)gs->d_xyz == gs->in_arena.base && ctx->pinned && ctx->pinned_used + need <= ctx->pinned_cap) {
            char *stage = ctx->pinned + ctx->pinned_used;
            memcpy(stage + ((char *)gs->d_xyz - gs->in_arena.base), xyz, 24 * (size_t)n_items);
            memcpy(stage + ((char *)gs->d_radii - gs->in_arena.base), radii, 4 * (size_t)n_items);
            memcpy(stage + ((char *)gs->d_item_group - gs->in_arena.base), item_group.data(), 4 * (size_t)n_items);
            HIP_TRY(ctx, copy_kernels() ? copy_by_kernel(ctx, gs->in_arena.base, stage, block) : hipMemcpyAsync(gs->in_arena.base, stage, block, hipMemcpyHostToDevice, st));
            ctx->pinned_used += need;
            return group_bounds(m, gs, n_items, n_groups, true, totals);
        }
    }
    if (n_items > 0) {
        if (xyz) {
            const H2DItem in[3] = {{gs->d_xyz, xyz, (size_t)(24 * n_items)}, {gs->d_radii, radii, (size_t)(4 * n_items)}, {gs->d_item_group, item_group.data(), (size_t)(4 * n_items)}};
            HIP_TRY(ctx, h2d_row(ctx, in, 3));       // (a row at the head of the scratch arena: one copy)
        } else {
            HIP_TRY(ctx, h2d_one(ctx, gs->d_crs, crs, (size_t)(12 * n_items)));
            HIP_TRY(ctx, h2d_one(ctx, gs->d_item_group, item_group.data(), (size_t)(4 * n_items)));
        }
    }
    return group_bounds(m, gs, n_items, n_groups, xyz != nullptr);   // (synchronises: item_group may go)
}

// Paint the group volumes and enqueue the labelling engine on them; the input scratch is recycled in stream order.
static int grouped_job(pdbeda_map *m, GroupSetup &gs, int64_t n_items, int64_t n_groups, bool spheres, float cutoff, pdbeda_bloblist **out, const PoolPaint *pool = nullptr,
                       bool unordered = false) {
    pdbeda_ctx *ctx = m->ctx;
    const int64_t max_runs = gs.total_keys / 2 + gs.total_words + 1;
    if (gs.pend_bytes && m->fix_mul == 0.0 && !m->fix_refused) {
        // map_fix_mul is about to WAIT (a map whose range is not known yet), and a wait hands the pinned block out afresh: inputs staged there must be on
        // their way before it, not left to k_job_init behind it
        const hipError_t ep = flush_pending(ctx, &gs);
        if (ep != hipSuccess) { arena_put(ctx, gs.in_arena); return fail(ctx, PDBEDA_ERR_DEVICE, "grouped blobs: %s", hipGetErrorString(ep)); }
    }
    int rc_fix = map_fix_mul(m);
    if (rc_fix) { arena_put(ctx, gs.in_arena); return rc_fix; }
    Job job;
    memset(&job, 0, sizeof job);
    job.fix_mul = m->fix_mul;
    size_t need = job_carve(job, nullptr, (int)n_groups, gs.total_words, gs.total_keys, max_runs, max_runs, 0, nullptr);
    Arena arena;
    int rc = arena_get(ctx, need, &arena);
    if (rc) { arena_put(ctx, gs.in_arena); return rc; }
    job_carve(job, arena.base, (int)n_groups, gs.total_words, gs.total_keys, max_runs, max_runs, 0, nullptr);
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    {   // volume descriptors into the job + zeroes over counters, masks, first-key bitmap and both levels of rank counters (adjacent in the arena: job_carve): one launch
        static_assert(sizeof(VolDesc) % 16 == 0, "VolDesc is copied in 16-byte units");
        const size_t zero_bytes = ((size_t)((char *)(job.mid_count + (job.key_words + KEY_FINE - 1) / KEY_FINE * (KEY_FINE / 16)) - (char *)job.ctr) + 15) & ~(size_t)15;   // (into the carve's own padding)
        const size_t vol16 = sizeof(VolDesc) * (size_t)n_groups / 16, in16 = gs.pend_bytes / 16;
        // (volume descriptors that are part of the inputs still to be copied come straight from the staged block)
        const char *vol_src = reinterpret_cast<const char *>(gs.d_vols);
        if (gs.pend_bytes && vol_src >= gs.pend_dst && vol_src < gs.pend_dst + gs.pend_bytes) vol_src = gs.pend_src + (vol_src - gs.pend_dst);
        hipLaunchKernelGGL(k_job_init, dim3((unsigned)std::min<size_t>((std::max(std::max(zero_bytes / 16, vol16), in16) + 255) / 256, 2048)), dim3(256), 0, st,
                           reinterpret_cast<const uint4 *>(gs.pend_src), reinterpret_cast<uint4 *>(gs.pend_dst), (unsigned long long)in16,
                           reinterpret_cast<const uint4 *>(vol_src), reinterpret_cast<uint4 *>(job.vols), (unsigned long long)vol16,
                           reinterpret_cast<uint4 *>(job.ctr), (unsigned long long)(zero_bytes / 16));
        e = hipGetLastError();
        gs.pend_bytes = 0;
    }
    const bool fused_atoms = spheres && gs.atom_engine && n_items > 0 && n_items == n_groups;
    if (e == hipSuccess && fused_atoms) {   // a per-atom batch of small boxes: paint, run index, unions, resolve and key painting in one launch
        PROF(ctx, "k_atom_engine");
        // (PDBEDA_DEBUG_ATOM_CAPS="runs,comps": tests shrink the LDS tables' use so that small inputs run the kernel's global-table path)
        static const std::pair<int, int> caps = [] {
            int r = ATOM_RUNS, c = ATOM_COMPS;
            if (const char *e = getenv("PDBEDA_DEBUG_ATOM_CAPS")) { int x = 0, y = 0; if (sscanf(e, "%d,%d", &x, &y) == 2) { r = std::max(0, std::min(x, ATOM_RUNS)); c = std::max(0, std::min(y, ATOM_COMPS)); } }
            return std::make_pair(r, c);
        }();
        hipLaunchKernelGGL(k_atom_engine, dim3((unsigned)n_items), dim3(256), 0, st, job, m->geom_dev, m->dens, gs.d_xyz, gs.d_radii, gs.d_boxes, cutoff, caps.first, caps.second);
        e = hipGetLastError();
    } else if (e == hipSuccess && n_items > 0) {
        if (spheres)
            { PROF(ctx, "k_sphere_paint"); hipLaunchKernelGGL(k_sphere_paint, dim3((unsigned)n_items), dim3(256), 0, st, m->geom_dev, m->dens, gs.d_xyz, gs.d_radii,
                               gs.d_item_group, gs.d_boxes, job.vols, job.mask, cutoff, gs.d_ctr, job.ctr); }
        else if (pool)      // (aggregateCloud's union job: the pooled voxels painted straight from the clouds' lists, the bonded pairs tested by the same launch)
            { PoolPaint pp = *pool; pp.ctr = job.ctr; PROF(ctx, "k_pool_paint"); hipLaunchKernelGGL(k_pool_paint, dim3(pp.paint_blocks + (unsigned)pp.n_pairs), dim3(256), 0, st, pp, job.vols, job.mask); }
        else
            { PROF(ctx, "k_list_paint"); hipLaunchKernelGGL(k_list_paint, dim3(grid_for(n_items, 256)), dim3(256), 0, st, gs.d_crs, gs.d_item_group, n_items, job.vols, job.mask); }
        e = hipGetLastError();
    }
    if (e == hipSuccess) rc = engine_enqueue(ctx, m, job, max_runs, fused_atoms, unordered);
    // (the inputs are consumed by the paint kernel: whoever gets their arena next is enqueued behind it on this stream)
    arena_put(ctx, gs.in_arena);
    if (e != hipSuccess || rc) {
        arena_put(ctx, arena);
        return rc ? rc : fail(ctx, PDBEDA_ERR_DEVICE, "grouped blobs: %s", hipGetErrorString(e));
    }
    pdbeda_bloblist *bl = new_list(ctx, m);
    bl->job = job;
    bl->arena = arena;
    bl->vol_lo = 0;
    bl->vol_hi = (int)n_groups;
    bl->host_boxes.swap(gs.host_boxes);
    *out = bl;
    return PDBEDA_OK;
}

static int grouped_blobs(pdbeda_map *m, const double *xyz, const float *radii, const int32_t *crs, int64_t n_items,
                         const int64_t *group_offsets, int64_t n_groups, float cutoff, pdbeda_bloblist **out) {
    pdbeda_ctx *ctx = m->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    GroupSetup gs;
    int rc = group_setup(m, xyz, radii, crs, n_items, group_offsets, n_groups, &gs);
    if (rc) { arena_put(ctx, gs.in_arena); return rc; }
    return grouped_job(m, gs, n_items, n_groups, xyz != nullptr, cutoff, out);
}

extern "C" int pdbeda_sphere_blobs(pdbeda_map *m, const double *xyz, const float *radii, int64_t n_atoms, const int64_t *group_offsets,
                                   int64_t n_groups, float density_cutoff, pdbeda_bloblist **out) {
    if (!m || !out || n_atoms < 0 || n_groups < 0 || !group_offsets || (n_atoms > 0 && (!xyz || !radii))) return PDBEDA_ERR_ARGUMENT;
    *out = nullptr;
    return grouped_blobs(m, xyz, radii, nullptr, n_atoms, group_offsets, n_groups, density_cutoff, out);
}

extern "C" int pdbeda_list_blobs(pdbeda_map *m, const int32_t *crs, int64_t n, const int64_t *group_offsets, int64_t n_groups,
                                 pdbeda_bloblist **out) {
    if (!m || !out || n < 0 || n_groups < 0 || !group_offsets || (n > 0 && !crs)) return PDBEDA_ERR_ARGUMENT;
    *out = nullptr;
    return grouped_blobs(m, nullptr, nullptr, crs, n, group_offsets, n_groups, 0.0f, out);
}

extern "C" int pdbeda_region_sums(pdbeda_map *m, const double *xyz, const float *radii, int64_t n_atoms, const int64_t *group_offsets,
                                  int64_t n_groups, float cutoff, double *pos, double *neg, int64_t *n_region, uint8_t *valid) {
    if (!m || n_atoms < 0 || n_groups < 0 || !group_offsets || (n_atoms > 0 && (!xyz || !radii))) return PDBEDA_ERR_ARGUMENT;
    pdbeda_ctx *ctx = m->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n_groups == 0) return PDBEDA_OK;
    GroupSetup gs;
    int rc = group_setup(m, xyz, radii, nullptr, n_atoms, group_offsets, n_groups, &gs);
    if (rc) { arena_put(ctx, gs.in_arena); return rc; }
    std::vector<unsigned long long> h_cnt(n_groups);
    std::vector<unsigned int> h_inv(n_groups);
    {   // a group per atom whose inputs the host staged (coordinates, radii, host-made volumes in the pinned block): ONE launch, no mask (k_atom_region)
        static const bool atom_region_on = [] { const char *e = getenv("PDBEDA_ATOM_REGION"); return !(e && e[0] == '0'); }();      // (A/B switch)
        const size_t out_bytes = 3 * align_up(8 * (size_t)n_groups, 64) + align_up(4 * (size_t)n_groups, 64);
        if (atom_region_on && n_groups == n_atoms && gs.pend_bytes && !gs.host_boxes.empty() && copy_kernels() && ctx->pinned && ctx->pinned_used + out_bytes <= ctx->pinned_cap) {
            char *blk = ctx->pinned + ctx->pinned_used;
            size_t off = 0;
            auto take = [&](void *dst, size_t bytes) { char *p = blk + off; if (dst && bytes) ctx->pending.push_back({dst, ctx->pinned_used + off, bytes}); off += align_up(std::max<size_t>(bytes, 1), 64); return p; };
            double *o_pos = reinterpret_cast<double *>(take(pos, 8 * (size_t)n_groups));
            double *o_neg = reinterpret_cast<double *>(take(neg, 8 * (size_t)n_groups));
            unsigned long long *o_cnt = reinterpret_cast<unsigned long long *>(take(h_cnt.data(), 8 * (size_t)n_groups));
            unsigned int *o_inv = reinterpret_cast<unsigned int *>(take(h_inv.data(), 4 * (size_t)n_groups));
            ctx->pinned_used += off;
            // (the staged row mirrors the scratch arena: the same offsets)
            const char *stage = gs.pend_src;
            const double *s_xyz = reinterpret_cast<const double *>(stage + ((char *)gs.d_xyz - gs.in_arena.base));
            const float *s_rad = reinterpret_cast<const float *>(stage + ((char *)gs.d_radii - gs.in_arena.base));
            const VolDesc *s_vols = reinterpret_cast<const VolDesc *>(stage + ((char *)gs.d_vols - gs.in_arena.base));
            gs.pend_bytes = 0;
            { PROF(ctx, "k_atom_region"); hipLaunchKernelGGL(k_atom_region, dim3((unsigned)std::min<int64_t>(n_groups, 65536)), dim3(256), 0, ctx->stream, m->geom_dev, m->dens, s_xyz, s_rad,
                                                            s_vols, (int)n_groups, cutoff, o_pos, o_neg, o_cnt, o_inv); }
            hipError_t e1 = hipGetLastError();
            if (e1 == hipSuccess) e1 = ctx_sync(ctx);
            arena_put(ctx, gs.in_arena);
            if (e1 != hipSuccess) return fail(ctx, PDBEDA_ERR_DEVICE, "region sums: %s", hipGetErrorString(e1));
            for (int64_t g = 0; g < n_groups; ++g) {
                if (n_region) n_region[g] = (int64_t)h_cnt[g];
                if (valid) valid[g] = h_inv[g] ? 0 : 1;
            }
            return PDBEDA_OK;
        }
    }
    const int64_t tw = std::max<int64_t>(gs.total_words, 1);
    Arena a;
    rc = arena_get(ctx, align_up(8 * tw) + 3 * align_up(8 * n_groups) + align_up(4 * n_groups), &a);
    if (rc) { arena_put(ctx, gs.in_arena); return rc; }
    Carver cv(a.base);
    uint64_t *mask = cv.take<uint64_t>(tw);
    double *d_pos = cv.take<double>(n_groups);
    double *d_neg = cv.take<double>(n_groups);
    unsigned long long *d_cnt = cv.take<unsigned long long>(n_groups);
    unsigned int *d_inv = cv.take<unsigned int>(n_groups);
    hipStream_t st = ctx->stream;
    hipError_t e;
    {   // the staged inputs into the scratch (when group_setup left their copy to us) and zeroes over masks and sums: one launch
        const size_t in16 = gs.pend_bytes / 16, zero16 = cv.off / 16;
        hipLaunchKernelGGL(k_job_init, dim3((unsigned)std::min<size_t>((std::max(zero16, in16) + 255) / 256, 2048)), dim3(256), 0, st,
                           reinterpret_cast<const uint4 *>(gs.pend_src), reinterpret_cast<uint4 *>(gs.pend_dst), (unsigned long long)in16,
                           (const uint4 *)nullptr, (uint4 *)nullptr, 0ull, reinterpret_cast<uint4 *>(a.base), (unsigned long long)zero16);
        e = hipGetLastError();
        gs.pend_bytes = 0;
    }
    if (e == hipSuccess && n_atoms > 0) {
        { PROF(ctx, "k_sphere_paint"); hipLaunchKernelGGL(k_sphere_paint, dim3((unsigned)n_atoms), dim3(256), 0, st, m->geom_dev, m->dens, gs.d_xyz, gs.d_radii,
                           gs.d_item_group, gs.d_boxes, gs.d_vols, mask, 0.0f, gs.d_ctr, (Counters *)nullptr); }
        { PROF(ctx, "k_region_reduce"); hipLaunchKernelGGL(k_region_reduce, dim3((unsigned)std::min<int64_t>(n_groups, 65536)), dim3(256), 0, st, m->geom_dev, m->dens, gs.d_vols,
                           (int)n_groups, mask, gs.total_words, cutoff, d_pos, d_neg, d_cnt, d_inv); }
        e = hipGetLastError();
    }
    Counters setup;
    memset(&setup, 0, sizeof setup);
    if (e == hipSuccess) {
        const D2HItem parts[5] = {{pos, d_pos, (size_t)(8 * n_groups)}, {neg, d_neg, (size_t)(8 * n_groups)}, {h_cnt.data(), d_cnt, (size_t)(8 * n_groups)},
                                  {h_inv.data(), d_inv, (size_t)(4 * n_groups)}, {gs.host_totals ? &setup : nullptr, gs.d_ctr, sizeof setup}};
        e = d2h_many(ctx, parts, 5);
    }
    if (e == hipSuccess) e = ctx_sync(ctx);
    arena_put(ctx, a);
    arena_put(ctx, gs.in_arena);
    if (e != hipSuccess) return fail(ctx, PDBEDA_ERR_DEVICE, "region sums: %s", hipGetErrorString(e));
    if (setup.overflow != 0u) return fail(ctx, PDBEDA_ERR_DEVICE, "region sums: the device's volumes outgrew what the host sized the batch for");
    for (int64_t g = 0; g < n_groups; ++g) {
        if (n_region) n_region[g] = (int64_t)h_cnt[g];
        if (valid) valid[g] = h_inv[g] ? 0 : 1;
    }
    return PDBEDA_OK;
}

// ------------------------------------------------------------------------------------
// Voxel-set adjacency, symmetry atoms, nearest atom
// ------------------------------------------------------------------------------------
extern "C" int pdbeda_test_overlap(pdbeda_ctx *ctx, const int32_t *crs, const int64_t *set_offsets, int64_t n_sets, const int32_t *a_idx,
                                   const int32_t *b_idx, int64_t n_pairs, uint8_t *out) {
    if (!ctx || n_sets < 0 || n_pairs < 0 || !set_offsets || (n_pairs > 0 && (!a_idx || !b_idx || !out))) return PDBEDA_ERR_ARGUMENT;
    if (n_pairs == 0) return PDBEDA_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t nv = set_offsets[n_sets];
    for (int64_t p = 0; p < n_pairs; ++p)
        if (a_idx[p] < 0 || a_idx[p] >= n_sets || b_idx[p] < 0 || b_idx[p] >= n_sets) return fail(ctx, PDBEDA_ERR_ARGUMENT, "pair index out of range");
    std::vector<unsigned int> h_out(n_pairs);
    int rc = with_scratch(ctx, align_up(12 * std::max<int64_t>(nv, 1)) + align_up(8 * (n_sets + 1)) + 3 * align_up(4 * n_pairs), [&](char *base) -> int {
        Carver cv(base);
        int32_t *d_crs = cv.take<int32_t>(3 * std::max<int64_t>(nv, 1));
        int64_t *d_off = cv.take<int64_t>(n_sets + 1);
        int32_t *d_a = cv.take<int32_t>(n_pairs);
        int32_t *d_b = cv.take<int32_t>(n_pairs);
        unsigned int *d_out = cv.take<unsigned int>(n_pairs);
        hipStream_t st = ctx->stream;
        if (nv > 0) HIP_TRY(ctx, h2d_one(ctx, d_crs, crs, (size_t)(12 * nv)));
        {
            const H2DItem in[3] = {{d_off, set_offsets, (size_t)(8 * (n_sets + 1))}, {d_a, a_idx, (size_t)(4 * n_pairs)}, {d_b, b_idx, (size_t)(4 * n_pairs)}};
            HIP_TRY(ctx, h2d_row(ctx, in, 3));      // (consecutive takes: one copy)
        }
        HIP_TRY(ctx, hipMemsetAsync(d_out, 0, 4 * n_pairs, st));
        hipLaunchKernelGGL(k_test_overlap, dim3((unsigned)n_pairs), dim3(256), 0, st, d_crs, d_off, d_a, d_b, d_out);
        HIP_TRY(ctx, d2h(ctx, h_out.data(), d_out, 4 * n_pairs));
        return 0;
    });
    if (rc) return rc;
    for (int64_t p = 0; p < n_pairs; ++p) out[p] = h_out[p] ? 1 : 0;
    return PDBEDA_OK;
}

extern "C" int pdbeda_symmetry_atoms(pdbeda_ctx *ctx, const double *xyz, int64_t n_atoms, const double *rot, int32_t n_ops,
                                     const double ortho[9], const double bbox_lo[3], const double bbox_hi[3], int32_t *atom_index,
                                     int32_t *symmetry, double *out_xyz, int64_t cap, int64_t *n_out) {
    if (!ctx || n_atoms < 0 || n_ops < 0 || (n_ops > 0 && !rot) || !ortho || !bbox_lo || !bbox_hi || !n_out || (n_atoms > 0 && !xyz)) return PDBEDA_ERR_ARGUMENT;
    *n_out = 0;
    // no operators (a file without REMARK 290): the reference's loop over them runs zero times and builds an empty list
    // (densityAnalysis.py:896-912) -- not an error
    if (n_atoms == 0 || n_ops == 0) return PDBEDA_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t total = 27ll * n_ops * n_atoms, n_words = (total + 63) / 64;
    std::vector<unsigned long long> h_keep((size_t)n_words);
    std::vector<int64_t> picked;
    std::vector<double> h_xyz;
    int rc = with_scratch(ctx, align_up(24 * n_atoms) + align_up(96 * n_ops) + align_up(72) + 2 * align_up(24) + align_up(8 * n_words) + align_up(8 * total) + align_up(24 * total),
                          [&](char *base) -> int {
        Carver cv(base);
        double *d_xyz = cv.take<double>(3 * n_atoms);
        double *d_rot = cv.take<double>(12 * n_ops);
        double *d_ortho = cv.take<double>(9);
        double *d_lo = cv.take<double>(3);
        double *d_hi = cv.take<double>(3);
        unsigned long long *d_keep = cv.take<unsigned long long>(n_words);
        int64_t *d_picked = cv.take<int64_t>(total);
        double *d_out = cv.take<double>(3 * total);
        hipStream_t st = ctx->stream;
        {
            const H2DItem in[5] = {{d_xyz, xyz, (size_t)(24 * n_atoms)}, {d_rot, rot, (size_t)(96 * n_ops)}, {d_ortho, ortho, 72}, {d_lo, bbox_lo, 24}, {d_hi, bbox_hi, 24}};
            HIP_TRY(ctx, h2d_row(ctx, in, 5));      // (one copy: the five sit in a row)
        }
        unsigned long long *r_keep = pinned_out(ctx, h_keep.data(), (size_t)n_words);
        { PROF(ctx, "k_symmetry_keep"); hipLaunchKernelGGL(k_symmetry_keep, dim3(grid_for(total, 256)), dim3(256), 0, st, d_xyz, n_atoms, d_rot, (int)n_ops, d_ortho, d_lo, d_hi, r_keep ? r_keep : d_keep); }
        if (!r_keep) HIP_TRY(ctx, d2h(ctx, h_keep.data(), d_keep, 8 * n_words));
        HIP_TRY(ctx, ctx_sync(ctx));
        for (int64_t w = 0; w < n_words; ++w)
            for (unsigned long long bits = h_keep[(size_t)w]; bits; bits &= bits - 1) picked.push_back(64 * w + __builtin_ctzll(bits));
        const int64_t n = (int64_t)picked.size();
        if (n == 0 || n > cap || !out_xyz) return 0;
        h_xyz.resize(3 * (size_t)n);
        const int64_t *in_picked = pinned_in(ctx, picked.data(), (size_t)n);      // (a survivor's number is read by its own thread, once)
        if (!in_picked) { HIP_TRY(ctx, h2d_one(ctx, d_picked, picked.data(), (size_t)(8 * n))); in_picked = d_picked; }
        double *r_out = pinned_out(ctx, h_xyz.data(), 3 * (size_t)n);
        { PROF(ctx, "k_symmetry_pick"); hipLaunchKernelGGL(k_symmetry_pick, dim3(grid_for(n, 256)), dim3(256), 0, st, d_xyz, n_atoms, d_rot, (int)n_ops, d_ortho, d_lo, d_hi, in_picked, n, r_out ? r_out : d_out); }
        if (!r_out) HIP_TRY(ctx, d2h(ctx, h_xyz.data(), d_out, 24 * n));
        return 0;
    });
    if (rc) return rc;
    const int64_t n = (int64_t)picked.size();
    *n_out = n;
    if (n > cap && (atom_index || symmetry || out_xyz)) return fail(ctx, PDBEDA_ERR_CAPACITY, "need capacity %lld", (long long)n);
    for (int64_t k = 0; k < n; ++k) {
        const int64_t t = picked[(size_t)k], a = t % n_atoms, cell_op = t / n_atoms;
        const int op = (int)(cell_op % n_ops), cell = (int)(cell_op / n_ops);
        if (atom_index) atom_index[k] = (int32_t)a;
        if (symmetry) { symmetry[4 * k] = cell / 9 - 1; symmetry[4 * k + 1] = (cell / 3) % 3 - 1; symmetry[4 * k + 2] = cell % 3 - 1; symmetry[4 * k + 3] = op; }
    }
    if (out_xyz && n > 0) memcpy(out_xyz, h_xyz.data(), 24 * (size_t)n);
    return PDBEDA_OK;
}

extern "C" int pdbeda_nearest_atom(pdbeda_ctx *ctx, const double *centroids, int64_t n_centroids, const double *atom_xyz, int64_t n_atoms,
                                   int64_t *index, double *distance) {
    if (!ctx || n_centroids < 0 || n_atoms <= 0 || !atom_xyz || (n_centroids > 0 && (!centroids || !index || !distance))) return PDBEDA_ERR_ARGUMENT;
    if (n_centroids == 0) return PDBEDA_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return with_scratch(ctx, align_up(24 * n_centroids) + align_up(24 * n_atoms) + 2 * align_up(8 * n_centroids), [&](char *base) -> int {
        Carver cv(base);
        double *d_c = cv.take<double>(3 * n_centroids);
        double *d_a = cv.take<double>(3 * n_atoms);
        int64_t *d_i = cv.take<int64_t>(n_centroids);
        double *d_d = cv.take<double>(n_centroids);
        hipStream_t st = ctx->stream;
        {
            const H2DItem in[2] = {{d_c, centroids, (size_t)(24 * n_centroids)}, {d_a, atom_xyz, (size_t)(24 * n_atoms)}};
            HIP_TRY(ctx, h2d_row(ctx, in, 2));
        }
        int64_t *r_i = pinned_out(ctx, index, (size_t)n_centroids);
        double *r_d = pinned_out(ctx, distance, (size_t)n_centroids);
        hipLaunchKernelGGL(k_nearest_atom, dim3((unsigned)n_centroids), dim3(256), 0, st, d_c, d_a, n_atoms, r_i ? r_i : d_i, r_d ? r_d : d_d);
        if (!r_i) HIP_TRY(ctx, d2h(ctx, index, d_i, 8 * n_centroids));
        if (!r_d) HIP_TRY(ctx, d2h(ctx, distance, d_d, 8 * n_centroids));
        return 0;
    });
}

// ------------------------------------------------------------------------------------
// aggregateCloud (densityAnalysis.py:571-731) behind one call
// ------------------------------------------------------------------------------------
struct pdbeda_cloud {
    pdbeda_ctx *ctx = nullptr;
    std::vector<int32_t> atom_idx;
    std::vector<double> atom_total, atom_centroid, atom_dist;
    std::vector<int64_t> atom_nvox;
    struct Row { int32_t residue; double total; int64_t n; double electrons; double cen[3]; };
    std::vector<Row> res_rows, dom_rows;
    std::vector<uint8_t> owner_state;
    double totals[4] = {0.0, 0.0, 0.0, 0.0};
};

// numpy's add.reduce over a contiguous float64 array (see k_np_chunk_sums): blocks of 8192 accumulated in order, each a
// pairwise sum.  mode 1: (x - shift)^2.  The centroid-distance cut-off (densityAnalysis.py:607) decides which atoms are
// pooled, so np.nanmedian + 2.5 * np.nanstd is reproduced operation by operation, not approximated.
static double np_elem_h(const double *a, int64_t i, int mode, double shift) {
    double v = a[i];
    if (mode == 1) { v = v - shift; v = v * v; }
    return v;
}
static double np_pairwise_h(const double *a, int64_t off, int64_t n, int mode, double shift) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += np_elem_h(a, off + i, mode, shift);
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = np_elem_h(a, off + j, mode, shift);
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += np_elem_h(a, off + i + j, mode, shift);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += np_elem_h(a, off + i, mode, shift);
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_h(a, off, n2, mode, shift) + np_pairwise_h(a, off + n2, n - n2, mode, shift);
}
static double np_sum_h(const double *a, int64_t n, int mode, double shift) {
    double out = 0.0;
    for (int64_t off = 0; off < n; off += 8192) out += np_pairwise_h(a, off, std::min<int64_t>(8192, n - off), mode, shift);
    return out;
}
static double np_median_h(std::vector<double> v) {   // np.median of a NaN-free 1-D array (the middle order statistics: a selection, not a sort)
    const size_t n = v.size();
    if (n == 0) return NAN;
    std::nth_element(v.begin(), v.begin() + n / 2, v.end());
    const double hi = v[n / 2];
    if (n & 1) return hi;
    const double lo = *std::max_element(v.begin(), v.begin() + n / 2);   // (the largest of what was put below the middle)
    return (lo + hi) / 2.0;
}
static double np_std_h(const std::vector<double> &v) {   // np.std (population) of a NaN-free 1-D array
    const int64_t n = (int64_t)v.size();
    if (n == 0) return NAN;
    const double mean = np_sum_h(v.data(), n, 0, 0.0) / (double)n;
    return std::sqrt(np_sum_h(v.data(), n, 1, mean) / (double)n);
}

static inline double norm3(const double *a, const double *b) {   // np.linalg.norm(a - b): sqrt((dx^2 + dy^2) + dz^2)
    const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return std::sqrt((dx * dx + dy * dy) + dz * dz);
}

extern "C" int pdbeda_aggregate_cloud(pdbeda_map *m, const pdbeda_cloud_atoms *at, float density_cutoff, double min_cloud_electrons,
                                      pdbeda_cloud **out) {
    if (!m || !at || !out || at->n < 0 || at->n_keys < 0 || at->n_owners < 0) return PDBEDA_ERR_ARGUMENT;
    *out = nullptr;
    pdbeda_ctx *ctx = m->ctx;
    const int64_t n = at->n;
    if (n > 0 && (!at->xyz || !at->radius || !at->weight || !at->residue || !at->alias || !at->key)) return PDBEDA_ERR_ARGUMENT;
    if (at->n_keys > 0 && !at->bonded_off) return PDBEDA_ERR_ARGUMENT;
    if (at->n_owners > 0 && !at->owner_key) return PDBEDA_ERR_ARGUMENT;
    if (n >= (1ll << 31)) return fail(ctx, PDBEDA_ERR_ARGUMENT, "too many atoms");
    for (int64_t i = 0; i < n; ++i) {
        if (at->alias[i] < 0 || at->alias[i] >= n || at->key[i] < 0 || at->key[i] >= at->n_keys || (i > 0 && at->residue[i] < at->residue[i - 1]))
            return fail(ctx, PDBEDA_ERR_ARGUMENT, "atom %lld: alias / key out of range or residues not in order", (long long)i);
    }
    for (int64_t o = 0; o < at->n_owners; ++o)
        if (at->owner_key[o] < 0 || at->owner_key[o] >= at->n_keys) return fail(ctx, PDBEDA_ERR_ARGUMENT, "owner key out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pdbeda_cloud *res = new pdbeda_cloud();
    res->ctx = ctx;
    res->owner_state.assign((size_t)at->n_owners, 0);
    res->totals[3] = NAN;
    *out = res;
    if (n == 0) return PDBEDA_OK;
    hipStream_t st = ctx->stream;
    // PDBEDA_CLOUD_TRACE=1 (experiments): the sizes of the two jobs and the host-side phases of this call on stderr
    static const bool cloud_trace = [] { const char *e = getenv("PDBEDA_CLOUD_TRACE"); return e && e[0] && e[0] != '0'; }();
    const double t_call = now_s();
    double t_wait1 = 0.0, t_pooled = 0.0, t_marks[6] = {0, 0, 0, 0, 0, 0};

    // ---- 1. the clouds of every atom: one sphere batch, a group per atom (findAberrantBlobs, 603) ----
    pdbeda_bloblist *clouds = nullptr;
    {
        std::vector<int64_t> goff((size_t)n + 1);
        for (int64_t i = 0; i <= n; ++i) goff[(size_t)i] = i;
        int rc = grouped_blobs(m, at->xyz, at->radius, nullptr, n, goff.data(), n, density_cutoff, &clouds);
        if (rc) { delete res; *out = nullptr; return rc; }
    }
    auto bail = [&](int rc, pdbeda_bloblist *a, pdbeda_bloblist *b) { if (a) pdbeda_bloblist_free(a); if (b) pdbeda_bloblist_free(b); delete res; *out = nullptr; return rc; };
    std::vector<int64_t> c_n;
    std::vector<double> c_tot, c_cen;
    std::vector<int32_t> c_grp;
    int rc = list_stats_one_trip(clouds, 4 * n + 64, c_n, c_tot, c_cen, c_grp);   // (an atom has one to three clouds: the count and the table in one wait)
    if (rc) return bail(rc, clouds, nullptr);
    const int64_t nb = (int64_t)c_n.size();
    t_wait1 = now_s();
    // the voxel lists of the clouds (what the pooled voxels and the overlap tests are gathered from) are made by the device
    // while the host decides what to pool
    rc = list_materialise_voxels(clouds);
    if (rc) return bail(rc, clouds, nullptr);

    t_marks[0] = now_s();
    // ---- 2. host: centroid-distance cut-off, best cloud, pool (604-642) ----
    std::vector<int64_t> first((size_t)n + 1, 0);       // clouds of atom a: [first[a], first[a + 1])  (sorted by group)
    for (int64_t c = 0; c < nb; ++c) first[(size_t)c_grp[(size_t)c] + 1]++;
    for (int64_t a = 0; a < n; ++a) first[(size_t)a + 1] += first[(size_t)a];
    std::vector<double> c_dist((size_t)nb), min_dist((size_t)n, NAN);
    std::vector<double> centroidDistances;
    centroidDistances.reserve((size_t)n);
    for (int64_t a = 0; a < n; ++a) {
        double mn = NAN;
        for (int64_t c = first[(size_t)a]; c < first[(size_t)a + 1]; ++c) {
            c_dist[(size_t)c] = norm3(at->xyz + 3 * a, c_cen.data() + 3 * c);
            if (c == first[(size_t)a] || c_dist[(size_t)c] < mn) mn = c_dist[(size_t)c];
        }
        min_dist[(size_t)a] = mn;
        if (first[(size_t)a + 1] > first[(size_t)a]) centroidDistances.push_back(mn);
    }
    const double cutoff = np_median_h(centroidDistances) + 2.5 * np_std_h(centroidDistances);
    res->totals[3] = cutoff;
    t_marks[1] = now_s();
    std::vector<int32_t> pool_cloud, pool_atom;
    pool_cloud.reserve((size_t)nb); pool_atom.reserve((size_t)nb);
    res->atom_idx.reserve((size_t)n); res->atom_total.reserve((size_t)n); res->atom_nvox.reserve((size_t)n);
    res->atom_centroid.reserve(3 * (size_t)n); res->atom_dist.reserve((size_t)n);
    std::vector<int32_t> pooled_of_key((size_t)at->n_keys, -1);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t s = at->alias[i], lo = first[(size_t)s], hi = first[(size_t)s + 1];
        if (hi == lo) continue;
        int64_t best = lo;
        if (hi - lo > 1) {
            if (min_dist[(size_t)s] > cutoff) continue;
            for (int64_t c = lo + 1; c < hi; ++c)
                if (c_dist[(size_t)c] < c_dist[(size_t)best]) best = c;      // the first minimum (distances.index(min))
        }
        for (int64_t c = lo; c < hi; ++c) { pool_cloud.push_back((int32_t)c); pool_atom.push_back((int32_t)i); }
        pooled_of_key[(size_t)at->key[i]] = (int32_t)i;                      // the last atom of a name wins (640)
        res->atom_idx.push_back((int32_t)i);
        res->atom_total.push_back(c_tot[(size_t)best]);
        res->atom_nvox.push_back(c_n[(size_t)best]);
        for (int k = 0; k < 3; ++k) res->atom_centroid.push_back(c_cen[3 * (size_t)best + k]);
        res->atom_dist.push_back(norm3(at->xyz + 3 * i, c_cen.data() + 3 * best));
    }
    const int64_t n_pool = (int64_t)pool_cloud.size();
    if (n_pool == 0) { pdbeda_bloblist_free(clouds); return PDBEDA_OK; }
    t_marks[2] = now_s();

    // ---- 3. device: pooled voxels -> union job (a group per residue + the domain group) ----
    pdbeda_bloblist *cow = owner_of(clouds);
    std::vector<int32_t> pool_group((size_t)n_pool), group_res;     // compact residue groups in increasing order
    for (int64_t p = 0; p < n_pool; ++p) {
        const int32_t r = at->residue[pool_atom[(size_t)p]];
        if (group_res.empty() || group_res.back() != r) group_res.push_back(r);
        pool_group[(size_t)p] = (int32_t)group_res.size() - 1;
    }
    const int n_rg = (int)group_res.size(), n_groups = n_rg + 1;
    std::vector<int64_t> pool_voff((size_t)n_pool + 1, 0), set_off((size_t)n + 1, 0);
    for (int64_t p = 0; p < n_pool; ++p) pool_voff[(size_t)p + 1] = pool_voff[(size_t)p] + c_n[(size_t)pool_cloud[(size_t)p]];
    const int64_t V = pool_voff[(size_t)n_pool];
    {   // voxel slice (all clouds) of every atom inside the sphere job's list, for the overlap tests
        std::vector<int64_t> coff((size_t)nb + 1, 0);
        for (int64_t c = 0; c < nb; ++c) coff[(size_t)c + 1] = coff[(size_t)c] + c_n[(size_t)c];
        for (int64_t a = 0; a <= n; ++a) set_off[(size_t)a] = coff[(size_t)first[(size_t)a]];
    }
    // bonded pairs between pooled names (656): owner o tests its name's atom against every bonded partner that is pooled
    std::vector<int32_t> pair_a, pair_b, pair_owner;
    for (int64_t o = 0; o < at->n_owners; ++o) {
        const int32_t k = at->owner_key[o], ia = pooled_of_key[(size_t)k];
        if (ia < 0) continue;
        res->owner_state[(size_t)o] = 1;
        for (int64_t q = at->bonded_off[k]; q < at->bonded_off[k + 1]; ++q) {
            const int32_t k2 = at->bonded[q];
            if (k2 < 0 || k2 >= at->n_keys) return bail(fail(ctx, PDBEDA_ERR_ARGUMENT, "bonded key out of range"), clouds, nullptr);
            const int32_t ib = pooled_of_key[(size_t)k2];
            if (ib < 0) continue;
            pair_a.push_back(at->alias[ia]); pair_b.push_back(at->alias[ib]); pair_owner.push_back((int32_t)o);
        }
    }
    const int64_t n_pairs = (int64_t)pair_a.size();
    t_marks[3] = now_s();

    GroupSetup gs;
    rc = group_alloc(ctx, 2 * V, n_groups, &gs);
    if (rc) return bail(rc, clouds, nullptr);
    Arena aux;
    rc = arena_get(ctx, align_up(sizeof(VolDesc) * (size_t)n_groups) + 2 * align_up(4 * n_pool) + align_up(8 * (n_pool + 1)) + align_up(8 * n_pool) + align_up(8 * (n + 1)) + 3 * align_up(4 * std::max<int64_t>(n_pairs, 1)), &aux);
    if (rc) { arena_put(ctx, gs.in_arena); return bail(rc, clouds, nullptr); }
    // what the device needs of the host's decisions, in ONE copy: pooled clouds, their groups, voxel offsets, the atoms' voxel
    // slices, the bonded pairs and the pairs' (zeroed) touch flags sit in a row in the scratch arena and in one host block
    // (six copies and a fill were seven launches on the stream, 6-8 us apart each: round 4, tools/exp/trace_cloud.sh)
    Carver cv(aux.base);
    const int64_t np1 = std::max<int64_t>(n_pairs, 1);
    VolDesc *d_union_vols = cv.take<VolDesc>(n_groups);      // (the union job's volume descriptors ride in the same block: round 5)
    int32_t *d_pool_cloud = cv.take<int32_t>(n_pool), *d_pool_group = cv.take<int32_t>(n_pool);
    int64_t *d_pool_voff = cv.take<int64_t>(n_pool + 1);
    int64_t *d_set_off = cv.take<int64_t>(n + 1);
    int32_t *d_pa = cv.take<int32_t>(np1), *d_pb = cv.take<int32_t>(np1);
    unsigned int *d_touch = cv.take<unsigned int>(np1);
    const size_t upload_bytes = cv.off;
    int32_t *d_comp = cv.take<int32_t>(2 * n_pool);
    auto fail_dev = [&](hipError_t e, pdbeda_bloblist *u) {
        arena_put(ctx, aux);
        return bail(fail(ctx, PDBEDA_ERR_DEVICE, "aggregate cloud: %s", hipGetErrorString(e)), clouds, u);
    };
    // The union job's volumes are made by the HOST (round 5; three launches -- k_init_bounds, k_list_boxes, k_make_vols -- and a
    // wait for their totals before): the voxels of a pooled cloud lie inside its atom's sphere box [C - R - 1, C + R], so the box
    // around the boxes of a residue's pooled atoms -- around all of them for the domain group -- holds the group's voxels.  A
    // volume a little larger than its voxels' own bounds changes nothing in the result: keys are lexicographic in (c, r, s)
    // whatever the box, so the blobs come out in the same order with the same sums.
    int64_t union_totals[2] = {0, 0};
    std::vector<VolDesc> union_vols((size_t)n_groups);
    {
        std::vector<int64_t> lo(3 * (size_t)n_groups, INT64_MAX), hi(3 * (size_t)n_groups, INT64_MIN);
        const bool have_boxes = (int64_t)cow->host_boxes.size() == n;      // (the sphere job's own boxes, made by the host with the same statements: two xyz2crs per pooled cloud were 0.06 ms here)
        int64_t last_a = -1;
        bool cur_empty = false;
        int64_t blo[3] = {0, 0, 0}, bhi[3] = {0, 0, 0};
        for (int64_t p = 0; p < n_pool; ++p) {
            const int64_t a = at->alias[pool_atom[(size_t)p]];          // (the clouds are those of the coordinate's last atom: its coordinate, its radius)
            if (a != last_a) {
                if (have_boxes) {
                    const AtomBox &bx = cow->host_boxes[(size_t)a];
                    for (int k = 0; k < 3; ++k) { blo[k] = bx.lo[k]; bhi[k] = bx.hi[k]; }
                    cur_empty = bx.hi[0] < bx.lo[0];      // (an empty box holds no voxel: it has no cloud, it is not pooled -- unreachable)
                } else {
                    cur_empty = false;
                    const double rad = (double)at->radius[a];
                    const double o[3] = {m->geom.origin[0] + rad, m->geom.origin[1] + rad, m->geom.origin[2] + rad};
                    int32_t C[3], R[3];
                    xyz2crs(m->geom, at->xyz + 3 * a, C);
                    xyz2crs(m->geom, o, R);
                    for (int k = 0; k < 3; ++k) { blo[k] = (int64_t)C[k] - R[k] - 1; bhi[k] = (int64_t)C[k] + R[k]; }
                }
                last_a = a;
            }
            if (cur_empty) continue;
            for (int g : {(int)pool_group[(size_t)p], n_rg})
                for (int k = 0; k < 3; ++k) {
                    lo[3 * (size_t)g + k] = std::min<int64_t>(lo[3 * (size_t)g + k], blo[k]);
                    hi[3 * (size_t)g + k] = std::max<int64_t>(hi[3 * (size_t)g + k], bhi[k]);
                }
        }
        for (int g = 0; g < n_groups; ++g) {
            VolDesc &vd = union_vols[(size_t)g];
            memset(&vd, 0, sizeof vd);
            vd.group = g;
            vd.word_base = union_totals[0];
            vd.key_base = union_totals[1];
            if (hi[3 * (size_t)g] < lo[3 * (size_t)g]) continue;
            bool fits = true;
            for (int k = 0; k < 3; ++k) {
                const int64_t d = hi[3 * (size_t)g + k] - lo[3 * (size_t)g + k] + 1;
                fits = fits && d < (1ll << 30) && lo[3 * (size_t)g + k] > INT32_MIN && hi[3 * (size_t)g + k] < INT32_MAX;
                vd.org[k] = (int32_t)lo[3 * (size_t)g + k];
                vd.dim[k] = (int32_t)d;
            }
            if (!fits) { union_totals[0] = INT64_MAX / 2; break; }
            vd.row_words = (vd.dim[0] + 63) / 64;
            union_totals[0] += (int64_t)vd.row_words * vd.dim[1] * vd.dim[2];
            union_totals[1] += (int64_t)vd.dim[0] * vd.dim[1] * vd.dim[2];
        }
    }
    t_marks[4] = now_s();
    const bool host_sized = !ctx->debug_shrink_totals && union_totals[0] < (1ll << 31) && union_totals[1] < (1ll << 40);   // (absurd boxes, or the debug hook that wants the device's own sizing: the waiting path sizes and reports)
    hipError_t e;
    std::vector<char> block(upload_bytes, 0);      // (staged: the copy below reads a pinned block or, too large for that, has finished reading when it returns)
    {
        auto put = [&](const void *dev, const void *src, size_t bytes) { if (bytes) memcpy(block.data() + ((const char *)dev - aux.base), src, bytes); };
        put(d_pool_cloud, pool_cloud.data(), 4 * (size_t)n_pool);
        put(d_pool_group, pool_group.data(), 4 * (size_t)n_pool);
        put(d_pool_voff, pool_voff.data(), 8 * (size_t)(n_pool + 1));
        put(d_set_off, set_off.data(), 8 * (size_t)(n + 1));
        put(d_pa, pair_a.data(), 4 * (size_t)n_pairs);
        put(d_pb, pair_b.data(), 4 * (size_t)n_pairs);
        if (host_sized) put(d_union_vols, union_vols.data(), sizeof(VolDesc) * (size_t)n_groups);
    }
    {
        const size_t span16 = (upload_bytes + 15) & ~(size_t)15, need = (span16 + 63) & ~(size_t)63;      // (upload_bytes is a carve boundary: a multiple of 256)
        if (copy_kernels() && ctx->pinned && span16 <= ((size_t)1 << 20) && ctx->pinned_used + need <= ctx->pinned_cap) {
            char *stage = ctx->pinned + ctx->pinned_used;      // staged now, copied by the union job's first launch (k_job_init)
            memcpy(stage, block.data(), upload_bytes);
            ctx->pinned_used += need;
            gs.pend_src = stage; gs.pend_dst = aux.base; gs.pend_bytes = span16;
            e = hipSuccess;
        } else {
            const H2DItem in[1] = {{aux.base, block.data(), upload_bytes}};
            e = h2d_row(ctx, in, 1);
        }
    }
    if (e != hipSuccess) { arena_put(ctx, gs.in_arena); return fail_dev(e, nullptr); }
    PoolPaint paint;
    memset(&paint, 0, sizeof paint);
    paint.src_crs = cow->crs_dev; paint.src_off = cow->offsets_dev;
    paint.pool_cloud = d_pool_cloud; paint.pool_voff = d_pool_voff; paint.pool_group = d_pool_group;
    paint.n_pool = (int)n_pool; paint.domain_group = n_rg; paint.V = V;
    paint.set_off = d_set_off; paint.pair_a = d_pa; paint.pair_b = d_pb; paint.touch = d_touch; paint.n_pairs = (int)n_pairs;
    paint.paint_blocks = grid_for(2 * V, 256, 1ll << 30);
    const bool fused_paint = host_sized && 2 * V <= (1ll << 30) * 256 && n_pairs < (1ll << 30);      // (the device-sized path needs the gathered list for its boxes)
    if (!fused_paint) {
        e = flush_pending(ctx, &gs);      // (these two read the aux block)
        if (e != hipSuccess) { arena_put(ctx, gs.in_arena); return fail_dev(e, nullptr); }
        { PROF(ctx, "k_pool_gather"); hipLaunchKernelGGL(k_pool_gather, dim3(grid_for(2 * V, 256)), dim3(256), 0, st, cow->crs_dev, cow->offsets_dev, d_pool_cloud, d_pool_voff,
                                                         d_pool_group, (int)n_pool, V, n_rg, gs.d_crs, gs.d_item_group); }
        if (n_pairs > 0) { PROF(ctx, "k_test_overlap"); hipLaunchKernelGGL(k_test_overlap, dim3((unsigned)n_pairs), dim3(256), 0, st, cow->crs_dev, d_set_off, d_pa, d_pb, d_touch); }
    }
    if (host_sized) {
        gs.d_vols = d_union_vols;          // (in the aux block, which outlives the job's enqueue)
        gs.total_words = union_totals[0];
        gs.total_keys = union_totals[1];
        gs.host_totals = true;
        rc = 0;
    } else {
        rc = group_bounds(m, &gs, 2 * V, n_groups, false);      // (synchronises)
    }
    if (rc) { arena_put(ctx, gs.in_arena); arena_put(ctx, aux); return bail(rc, clouds, nullptr); }
    pdbeda_bloblist *uni = nullptr;
    t_pooled = now_s();
    // The union job's results: rows of its blobs, the row of the component that holds each pooled cloud, the pairs' touch flags.  The
    // host orders the rows by the pooled clouds they hold (below), not by key: the job runs UNORDERED (no k_paint_keys, no k_emit) and its
    // last launch writes the results straight into the pinned block (k_union_finish; four launches -- keys, ranks, components, pack --
    // before).  Without room in the block (or with PDBEDA_COPY_KERNELS=0: kernels do not write host memory then) the ordered form serves.
    const int64_t u_cap = 2 * n_pool + 64;      // (a union component holds at least one pooled cloud, per kind)
    std::vector<int32_t> comp(2 * (size_t)n_pool);
    std::vector<unsigned int> touch((size_t)n_pairs);
    std::vector<int64_t> u_n;
    std::vector<double> u_tot, u_cen;
    std::vector<int32_t> u_grp;
    const size_t fin_bytes = align_up(sizeof(Counters), 64) + align_up(8 * (size_t)u_cap, 64) * 2 + align_up(24 * (size_t)u_cap, 64) + align_up(4 * (size_t)u_cap, 64) +
                             align_up(4 * (size_t)std::max<int64_t>(n_pairs, 1), 64) + align_up(8 * (size_t)n_pool, 64);
    static const bool unordered_on = [] { const char *e = getenv("PDBEDA_UNORDERED_UNION"); return !(e && e[0] == '0'); }();      // (A/B switch)
    const bool unordered = unordered_on && fused_paint && copy_kernels() && ctx->pinned && ctx->pinned_used + fin_bytes <= ctx->pinned_cap;
    rc = grouped_job(m, gs, 2 * V, n_groups, false, 0.0f, &uni, fused_paint ? &paint : nullptr, unordered);
    if (rc) { arena_put(ctx, aux); return bail(rc, clouds, nullptr); }
    if (unordered) {
        u_n.resize((size_t)u_cap); u_tot.resize((size_t)u_cap); u_cen.resize(3 * (size_t)u_cap); u_grp.resize((size_t)u_cap);
        Counters u_ctr;
        memset(&u_ctr, 0, sizeof u_ctr);
        char *blk = ctx->pinned + ctx->pinned_used;
        size_t off = 0;
        auto take = [&](void *dst, size_t bytes) { char *p = blk + off; if (bytes) ctx->pending.push_back({dst, ctx->pinned_used + off, bytes}); off += align_up(std::max<size_t>(bytes, 1), 64); return p; };
        UnionFinish fin;
        memset(&fin, 0, sizeof fin);
        fin.src_crs = cow->crs_dev; fin.src_off = cow->offsets_dev; fin.pool_cloud = d_pool_cloud; fin.pool_group = d_pool_group;
        fin.n_pool = (int)n_pool; fin.domain_group = n_rg; fin.touch = d_touch; fin.n_pairs = (int)n_pairs; fin.cap = (unsigned int)u_cap;
        fin.out_ctr = reinterpret_cast<Counters *>(take(&u_ctr, sizeof u_ctr));
        fin.out_n = reinterpret_cast<long long *>(take(u_n.data(), 8 * (size_t)u_cap));
        fin.out_total = reinterpret_cast<double *>(take(u_tot.data(), 8 * (size_t)u_cap));
        fin.out_centroid = reinterpret_cast<double *>(take(u_cen.data(), 24 * (size_t)u_cap));
        fin.out_group = reinterpret_cast<int32_t *>(take(u_grp.data(), 4 * (size_t)u_cap));
        fin.out_touch = reinterpret_cast<unsigned int *>(take(touch.data(), 4 * (size_t)n_pairs));
        fin.out_comp = reinterpret_cast<int32_t *>(take(comp.data(), 8 * (size_t)n_pool));
        ctx->pinned_used += off;
        const int64_t fin_threads = std::max<int64_t>(std::max<int64_t>(uni->job.run_cap, 2 * n_pool), n_pairs);
        { PROF(ctx, "k_union_finish"); hipLaunchKernelGGL(k_union_finish, dim3(grid_for(fin_threads, 256, 1024)), dim3(256), 0, st, uni->job, m->geom_dev, fin); }
        e = hipGetLastError();
        if (e != hipSuccess) return fail_dev(e, uni);
        e = ctx_sync(ctx);      // (the one wait of the union job: everything above lands in the host's arrays)
        arena_put(ctx, aux);
        if (e != hipSuccess) return bail(fail(ctx, ctx->timed_out ? PDBEDA_ERR_TIMEOUT : PDBEDA_ERR_DEVICE, "aggregate cloud: %s", ctx->timed_out ? ctx->err.c_str() : hipGetErrorString(e)), clouds, uni);
        if (u_ctr.unit_wait_failed) return bail(fail(ctx, PDBEDA_ERR_DEVICE, "aggregate cloud: a pooled voxel lies outside its group's volume"), clouds, uni);
        if ((int64_t)u_ctr.n_blobs > u_cap) return bail(fail(ctx, PDBEDA_ERR_STATE, "aggregate cloud: more union components than pooled clouds"), clouds, uni);
        const size_t nb_u = u_ctr.n_blobs;
        u_n.resize(nb_u); u_tot.resize(nb_u); u_cen.resize(3 * nb_u); u_grp.resize(nb_u);
    } else {
        { PROF(ctx, "k_pool_component"); hipLaunchKernelGGL(k_pool_component, dim3(grid_for(2 * n_pool, 256)), dim3(256), 0, st, uni->job, cow->crs_dev, cow->offsets_dev, d_pool_cloud,
                                                            d_pool_group, (int)n_pool, n_rg, d_comp); }
        e = hipGetLastError();
        if (e != hipSuccess) return fail_dev(e, uni);
        // (the pairs' touch flags and the pooled clouds' components are neighbours in the aux arena: one copy brings both)
        const size_t tail_bytes = (size_t)((char *)(d_comp + 2 * n_pool) - (char *)d_touch);
        std::vector<char> tail((tail_bytes + 3) & ~(size_t)3);
        const D2HItem tail_item = {tail.data(), d_touch, tail.size()};      // (whole words: the pack kernel copies words; the carve's padding covers the round-up)
        rc = list_stats_one_trip(uni, u_cap, u_n, u_tot, u_cen, u_grp, &tail_item);   // (synchronises: comp / touch land with the statistics)
        arena_put(ctx, aux);
        if (rc) return bail(rc, clouds, uni);
        if (n_pairs > 0) memcpy(touch.data(), tail.data(), 4 * (size_t)n_pairs);
        memcpy(comp.data(), tail.data() + ((char *)d_comp - (char *)d_touch), 8 * (size_t)n_pool);
    }
    const int64_t nu = (int64_t)u_n.size();
    if (cloud_trace)
        fprintf(stderr, "aggregate_cloud: %lld atoms, %lld clouds, %lld pooled clouds (%lld voxels), %d residue groups + the domain group: %lld mask words, %lld keys, %lld union blobs; "
                        "ms: clouds enqueued + waited %.3f, host pooling %.3f (voxel lists enqueued %.3f, distances + cut-off %.3f, pool loop %.3f, groups + pairs %.3f, arenas + volumes %.3f, block + staging %.3f), union job enqueued + waited %.3f\n",
                (long long)n, (long long)nb, (long long)n_pool, (long long)V, n_rg, (long long)union_totals[0], (long long)union_totals[1], (long long)nu,
                1e3 * (t_wait1 - t_call), 1e3 * (t_pooled - t_wait1), 1e3 * (t_marks[0] - t_wait1), 1e3 * (t_marks[1] - t_marks[0]), 1e3 * (t_marks[2] - t_marks[1]),
                1e3 * (t_marks[3] - t_marks[2]), 1e3 * (t_marks[4] - t_marks[3]), 1e3 * (t_pooled - t_marks[4]), 1e3 * (now_s() - t_pooled));
    pdbeda_bloblist_free(uni);
    pdbeda_bloblist_free(clouds);
    if (rc) { delete res; *out = nullptr; return rc; }

    // ---- 4. host: overlap completeness, electrons per union component, emission order, totals (652-731) ----
    for (int64_t q = 0; q < n_pairs; ++q)
        if (!touch[(size_t)q]) res->owner_state[(size_t)pair_owner[(size_t)q]] = 2;
    std::vector<double> electrons((size_t)nu, 0.0);
    std::vector<int64_t> first_pool((size_t)nu, n_pool);
    // Whose electrons a pooled cloud carries (cloud.atoms, 639 / 690 / 718).  The atoms of one coordinate share the cloud OBJECTS
    // of the last of them (605, 622), and the atom loop of a residue overwrites their .atoms (639): when two pooled atoms of ONE
    // residue share a coordinate, both pool entries are the same objects and name only the LATER atom -- the earlier one's
    // electrons are in no residue or domain cloud (golden analysis_alias; round 3 counted both).  Atoms of different residues
    // each get their own residue's snapshot (clone(), 675), so both count there and in the domain.
    std::vector<int32_t> named_by((size_t)n);
    for (int64_t i = 0; i < n; ++i) named_by[(size_t)i] = (int32_t)i;
    {
        std::vector<int32_t> last_of_alias((size_t)n, -1);             // per residue: the last pooled atom that uses the clouds of alias a
        size_t r0 = 0;
        const std::vector<int32_t> &pooled = res->atom_idx;            // (in atom order, hence residue by residue)
        while (r0 < pooled.size()) {
            size_t r1 = r0;
            while (r1 < pooled.size() && at->residue[pooled[r1]] == at->residue[pooled[r0]]) ++r1;
            for (size_t q = r0; q < r1; ++q) last_of_alias[(size_t)at->alias[pooled[q]]] = pooled[q];
            for (size_t q = r0; q < r1; ++q) named_by[(size_t)pooled[q]] = last_of_alias[(size_t)at->alias[pooled[q]]];
            for (size_t q = r0; q < r1; ++q) last_of_alias[(size_t)at->alias[pooled[q]]] = -1;
            r0 = r1;
        }
    }
    std::vector<std::pair<int32_t, int32_t>> members;                  // (union component, atom named by a pooled cloud in it): distinct pairs add electrons once
    members.reserve(2 * (size_t)n_pool);
    for (int kind = 0; kind < 2; ++kind) {
        for (int64_t p = 0; p < n_pool; ++p) {
            const int32_t k = comp[(size_t)(kind * n_pool + p)];
            if (k < 0 || k >= nu) { delete res; *out = nullptr; return fail(ctx, PDBEDA_ERR_STATE, "aggregate cloud: component rank out of range"); }
            members.emplace_back(k, named_by[(size_t)pool_atom[(size_t)p]]);
            if (p < first_pool[(size_t)k]) first_pool[(size_t)k] = p;
        }
    }
    {   // (summed in atom order per component, as before: pool order is atom order)
        std::stable_sort(members.begin(), members.end());
        for (size_t q = 0; q < members.size(); ++q)
            if (q == 0 || members[q] != members[q - 1]) electrons[(size_t)members[q].first] += at->weight[members[q].second];
    }
    std::vector<int64_t> order((size_t)nu);
    for (int64_t k = 0; k < nu; ++k) order[(size_t)k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        const bool da = u_grp[(size_t)a] == n_rg, db = u_grp[(size_t)b] == n_rg;
        if (da != db) return db;                                   // residue clouds first, then the domain clouds
        return first_pool[(size_t)a] < first_pool[(size_t)b];
    });
    double num_voxels = 0.0, total_electrons = 0.0, total_density = 0.0;
    for (int64_t idx = 0; idx < nu; ++idx) {
        const int64_t k = order[(size_t)idx];
        const bool dom = u_grp[(size_t)k] == n_rg;
        pdbeda_cloud::Row row;
        row.total = u_tot[(size_t)k]; row.n = u_n[(size_t)k]; row.electrons = electrons[(size_t)k];
        for (int q = 0; q < 3; ++q) row.cen[q] = u_cen[3 * (size_t)k + q];
        if (dom) {
            total_electrons += row.electrons;
            num_voxels += (double)row.n;
            total_density += row.total;
            row.residue = at->residue[pool_atom[(size_t)first_pool[(size_t)k]]];
            if (row.electrons >= min_cloud_electrons) res->dom_rows.push_back(row);
        } else {
            row.residue = group_res[(size_t)u_grp[(size_t)k]];
            if (row.electrons >= min_cloud_electrons) res->res_rows.push_back(row);
        }
    }
    res->totals[0] = num_voxels; res->totals[1] = total_electrons; res->totals[2] = total_density;
    return PDBEDA_OK;
}

extern "C" int pdbeda_cloud_counts(pdbeda_cloud *c, int64_t counts[4], double totals[4]) {
    if (!c) return PDBEDA_ERR_ARGUMENT;
    if (counts) { counts[0] = (int64_t)c->atom_idx.size(); counts[1] = (int64_t)c->res_rows.size(); counts[2] = (int64_t)c->dom_rows.size(); counts[3] = (int64_t)c->owner_state.size(); }
    if (totals) for (int k = 0; k < 4; ++k) totals[k] = c->totals[k];
    return PDBEDA_OK;
}

extern "C" int pdbeda_cloud_atom_rows(pdbeda_cloud *c, int32_t *atom, double *total_density, int64_t *n_voxels, double *centroid, double *distance) {
    if (!c) return PDBEDA_ERR_ARGUMENT;
    const size_t n = c->atom_idx.size();
    if (atom) memcpy(atom, c->atom_idx.data(), 4 * n);
    if (total_density) memcpy(total_density, c->atom_total.data(), 8 * n);
    if (n_voxels) memcpy(n_voxels, c->atom_nvox.data(), 8 * n);
    if (centroid) memcpy(centroid, c->atom_centroid.data(), 24 * n);
    if (distance) memcpy(distance, c->atom_dist.data(), 8 * n);
    return PDBEDA_OK;
}

static int cloud_rows_out(const std::vector<pdbeda_cloud::Row> &rows, int32_t *residue, double *total_density, int64_t *n_voxels, double *electrons, double *centroid) {
    for (size_t i = 0; i < rows.size(); ++i) {
        if (residue) residue[i] = rows[i].residue;
        if (total_density) total_density[i] = rows[i].total;
        if (n_voxels) n_voxels[i] = rows[i].n;
        if (electrons) electrons[i] = rows[i].electrons;
        if (centroid) for (int q = 0; q < 3; ++q) centroid[3 * i + q] = rows[i].cen[q];
    }
    return PDBEDA_OK;
}
extern "C" int pdbeda_cloud_residue_rows(pdbeda_cloud *c, int32_t *residue, double *total_density, int64_t *n_voxels, double *electrons, double *centroid) {
    if (!c) return PDBEDA_ERR_ARGUMENT;
    return cloud_rows_out(c->res_rows, residue, total_density, n_voxels, electrons, centroid);
}
extern "C" int pdbeda_cloud_domain_rows(pdbeda_cloud *c, int32_t *residue, double *total_density, int64_t *n_voxels, double *electrons, double *centroid) {
    if (!c) return PDBEDA_ERR_ARGUMENT;
    return cloud_rows_out(c->dom_rows, residue, total_density, n_voxels, electrons, centroid);
}
extern "C" int pdbeda_cloud_owner_states(pdbeda_cloud *c, uint8_t *state) {
    if (!c || (!state && !c->owner_state.empty())) return PDBEDA_ERR_ARGUMENT;
    if (!c->owner_state.empty()) memcpy(state, c->owner_state.data(), c->owner_state.size());
    return PDBEDA_OK;
}
extern "C" int pdbeda_cloud_free(pdbeda_cloud *c) {
    if (!c) return PDBEDA_ERR_ARGUMENT;
    delete c;
    return PDBEDA_OK;
}
