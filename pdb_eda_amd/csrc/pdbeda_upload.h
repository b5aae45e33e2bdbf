// pdbeda_upload.h -- the upload engine: ONE per process and device (round 5), behind a table of function pointers (round 6).
//
// A map read from a file goes to HBM in chunks: a reader thread pread()s a chunk out of the page cache into one of its two
// pinned slots and queues the slot's PCIe copy on its own stream.  The readers belong to the PROCESS: the chunks of every load in
// flight go through one FIFO, a lone load gets all readers, and the number of threads and pinned chunks does not grow with the
// number of contexts (rounds 3-4: two readers and a ring per context -- sixteen threads and 128 MiB of pinned rings a worker).
//
// This header has no HIP in it: everything the engine asks of the runtime -- a copy stream, pinned chunks, events, an
// asynchronous copy -- goes through `Backend`.  pdbeda_hip.hip fills the table with the HIP calls; tests/upload_harness.cpp fills
// it with a host stand-in (a worker thread per stream that executes the queued copies with memcpy) and is built with
// -fsanitize=thread and -fsanitize=address,undefined in the CPU container (tools/sanitize_cpu.sh): three reader threads, a FIFO,
// slot recycling, deadlines and the reader replacement below never ran under a sanitizer before (GPU sanitizers are not
// available on the pool).
//
// Stalled slots.  A reader waits for the copy that left a slot two chunks ago before it reads into the slot again.
// With per-process readers that copy may be ANOTHER context's -- e.g. into the arena of a context the watchdog has abandoned
// behind a kernel that never ends.  Rules:
//   * a load's own deadline still ends the load (the entry has run out of time whatever the reason);
//   * a slot that does not come free within `stall_s` (2 s; an 8 MiB copy takes 150 us), or whose event query fails, is given up:
//     the reader takes a NEW stream, two NEW pinned chunks and events (the old ones are leaked on purpose -- a copy may still be
//     reading them) and goes on, so one copy that never completes cannot hold every later upload of the process;
//   * a load that has chunks on a stream that was given up fails (its bytes may never arrive): `stalled`.
#pragma once
#include <sys/types.h>
#include <unistd.h>
#include <cerrno>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace pdbeda_upload {

constexpr int MAX_READERS = 8;
constexpr int FILE_READERS_DEFAULT = 3;
typedef void *Stream;
typedef void *Event;
enum { UP_OK = 0, UP_NOT_READY = -1 };   // any other value: the backend's own error code (returned to the caller as it is)

struct Backend {
    int (*set_device)(int device);
    int (*stream_create)(Stream *out);
    int (*host_alloc)(void **out, size_t bytes);        // pinned
    int (*event_create)(Event *out);
    int (*event_query)(Event ev);                       // UP_OK, UP_NOT_READY or an error
    int (*copy_async)(void *dst, const void *src, size_t bytes, Stream s);
    int (*event_record)(Event ev, Stream s);
};

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline long env_long(const char *name, long dflt) { const char *e = getenv(name); return e ? atol(e) : dflt; }

// the most a chunk may be; chunks (= a reader's pinned slots) are 8 MiB by default (four workers, both maps: 4 MiB 1.44-1.45 ms per
// entry, 8 MiB 1.40-1.42, 16 MiB 1.39-1.40).  PDBEDA_FILE_CHUNK_KB: experiments.
constexpr long FILE_CHUNK_MAX_KB = 16384;
inline size_t file_chunk_bytes() {
    static const size_t v = (size_t)std::min<long>(std::max<long>(env_long("PDBEDA_FILE_CHUNK_KB", 8192), 64), FILE_CHUNK_MAX_KB) << 10;
    return v;
}
// A load that finds the engine idle is cut into smaller chunks: with nobody else's copies to fill the link, the pread of a full
// chunk (0.2-0.4 ms at the 20-40 GB/s of one thread) is time the link idles at the start of the map; with other loads in flight the
// larger chunk's fewer copies win (one engine serves the copies one after another, 6 us apart).  PDBEDA_FILE_CHUNK_ALONE_KB
// (experiments; 0 = the full chunk always): one 32 MB map alone 0.85 ms with 8 MiB chunks, 0.77 with 4 or 2, 1.0 with 1.
inline size_t file_chunk_alone_bytes() {
    static const size_t v = [] {
        const long kb = env_long("PDBEDA_FILE_CHUNK_ALONE_KB", 4096);
        return kb <= 0 ? file_chunk_bytes() : std::min(file_chunk_bytes(), std::max<size_t>((size_t)kb << 10, (size_t)64 << 10));
    }();
    return v;
}
// the first two rounds of a load's chunks (PDBEDA_FILE_RAMP_KB="a,b": experiments)
inline size_t ramp_bytes(int round) {
    static const std::pair<size_t, size_t> v = [] {
        long a = 256, b = 1024;
        if (const char *e = getenv("PDBEDA_FILE_RAMP_KB")) { long x = 0, y = 0; if (sscanf(e, "%ld,%ld", &x, &y) == 2 && x >= 16 && y >= 16) { a = x; b = y; } }
        return std::make_pair((size_t)a << 10, (size_t)b << 10);
    }();
    return round == 0 ? v.first : v.second;
}
inline unsigned reader_spins() { static const unsigned v = (unsigned)env_long("PDBEDA_READER_SPINS", 4); return v; }
inline double stall_seconds() { static const double v = [] { const char *e = getenv("PDBEDA_SLOT_STALL_S"); return e ? atof(e) : 2.0; }(); return v; }
inline bool upload_trace() { static const bool v = [] { const char *e = getenv("PDBEDA_UPLOAD_TRACE"); return e && e[0] && e[0] != '0'; }(); return v; }

struct UploadLoad {
    int fd = -1;                                   // the source: a file (pread at offset + position) ...
    int64_t offset = 0;
    const char *src = nullptr;                     // ... or the caller's memory (fd < 0)
    char *dst = nullptr;
    size_t need = 0;
    int64_t n_chunks = 0;
    std::vector<std::pair<size_t, size_t>> pieces;   // (position, length) of chunk k: small ones first -- the link starts while the big ones are read
    double timeout_s = 0.0;
    std::chrono::steady_clock::time_point deadline;
    std::mutex mu;
    std::condition_variable cv;
    int64_t handled = 0;                          // chunks whose copy is queued, or that were given up
    int e = UP_OK;
    const char *why = nullptr;
    bool timed_out = false;
    bool stalled = false;                         // some chunk of this load sits on a stream that was given up
    bool used[MAX_READERS] = {false, false, false, false, false, false, false, false};
    Stream stream_used[MAX_READERS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // where reader r queued this load's chunks
    bool failed() { return e != UP_OK || why || timed_out || stalled; }   // (under mu)
    // PDBEDA_UPLOAD_TRACE=1 (experiments): seconds spent in pread / in the runtime's calls, summed over the readers
    double t_pread = 0.0, t_queue = 0.0, t_slot = 0.0;
    int chunks_by[MAX_READERS] = {0, 0, 0, 0, 0, 0, 0, 0};
};

struct UploadEngine {
    const Backend *be = nullptr;
    int device = 0, n_readers = 0;
    std::atomic<int> active{0};   // loads between their submission and their last chunk
    std::atomic<int> replaced{0}; // readers that gave a stalled stream up (counters for tests / traces)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<UploadLoad *, int64_t>> chunks;
    std::mutex dead_mu;
    std::vector<Stream> dead;     // streams that were given up
    struct Reader {
        Stream stream = nullptr;
        char *slot[2] = {nullptr, nullptr};
        Event done[2] = {nullptr, nullptr};
        int64_t count = 0;
        bool ok = false;
        int broken = UP_OK;       // the reader could not replace a stalled stream: every chunk it takes fails with this code
    } readers[MAX_READERS];

    bool make_reader(Reader &rd) {
        Reader fresh;
        bool ok = be->stream_create(&fresh.stream) == UP_OK;
        for (int k = 0; k < 2 && ok; ++k)
            ok = be->host_alloc((void **)&fresh.slot[k], file_chunk_bytes()) == UP_OK && be->event_create(&fresh.done[k]) == UP_OK;
        if (!ok) return false;
        fresh.ok = true;
        rd = fresh;
        return true;
    }

    void run(int r) {
        Reader &me = readers[r];
        (void)be->set_device(device);
        for (;;) {
            std::pair<UploadLoad *, int64_t> task;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !chunks.empty(); });
                task = chunks.front();
                chunks.pop_front();
            }
            UploadLoad *ld = task.first;
            const int64_t c = task.second;
            bool skip;
            Stream earlier = nullptr;
            {
                std::lock_guard<std::mutex> g(ld->mu);
                skip = ld->failed();
                earlier = ld->used[r] ? ld->stream_used[r] : nullptr;
            }
            int ce = me.broken;
            const char *why = nullptr;
            bool timed_out = false, queued = false, stalled = false;
            double t_a = 0, t_b = 0, t_c = 0, t_d = 0;
            if (!skip && ce == UP_OK) {
                int slot = (int)(me.count & 1);
                t_a = now_s();
                if (me.count >= 2) {   // the chunk sent from this slot two rounds ago must have left it
                    bool give_up = false;
                    for (unsigned spins = 0;; ++spins) {
                        const int q = be->event_query(me.done[slot]);
                        if (q == UP_OK) break;
                        if (q != UP_NOT_READY) { give_up = true; break; }   // (the stream is in an error state: nothing of it is used again)
                        if (spins >= reader_spins()) {   // (no long busy poll: six readers hammering the event query slowed every OTHER thread's runtime calls)
                            if (ld->timeout_s > 0.0 && std::chrono::steady_clock::now() > ld->deadline) { timed_out = true; break; }   // (the entry's one deadline, as ctx_wait)
                            if (now_s() - t_a > stall_seconds()) { give_up = true; break; }
                            std::this_thread::sleep_for(std::chrono::microseconds(30));
                        }
                    }
                    if (give_up) {
                        // the copy that holds this slot does not end (another context's, into an arena behind a hung kernel) or its stream failed:
                        // new stream, new chunks, new events; the old ones stay where they are (a copy may still read them)
                        if (earlier == me.stream) stalled = true;   // this load has chunks on the stream that is given up
                        { std::lock_guard<std::mutex> g(dead_mu); dead.push_back(me.stream); }
                        if (make_reader(me)) { me.count = 0; slot = 0; replaced.fetch_add(1); }
                        else { me.broken = ce = -2; }
                    }
                }
                t_b = now_s();
                if (ce == UP_OK && !timed_out && !stalled) {
                    const size_t pos = ld->pieces[(size_t)c].first, len = ld->pieces[(size_t)c].second;
                    if (ld->fd < 0) memcpy(me.slot[slot], ld->src + pos, len);
                    for (size_t got = 0; ld->fd >= 0 && got < len;) {
                        const ssize_t n = pread(ld->fd, me.slot[slot] + got, len - got, (off_t)(ld->offset + (int64_t)pos + (int64_t)got));
                        if (n < 0 && errno == EINTR) continue;
                        if (n <= 0) { why = n < 0 ? strerror(errno) : "unexpected end of file"; break; }
                        got += (size_t)n;
                    }
                    t_c = now_s();
                    if (!why) {
                        ce = be->copy_async(ld->dst + pos, me.slot[slot], len, me.stream);
                        if (ce == UP_OK) ce = be->event_record(me.done[slot], me.stream);
                        if (ce == UP_OK) { queued = true; ++me.count; }
                    }
                    t_d = now_s();
                }
            }
            {
                std::lock_guard<std::mutex> g(ld->mu);
                if (ce != UP_OK && ld->e == UP_OK) ld->e = ce;
                if (why && !ld->why) ld->why = why;
                if (timed_out) ld->timed_out = true;
                if (stalled) ld->stalled = true;
                if (queued) {
                    if (ld->used[r] && ld->stream_used[r] != me.stream) ld->stalled = true;   // (earlier chunks of the load went to a stream given up since)
                    ld->used[r] = true;
                    ld->stream_used[r] = me.stream;
                }
                ld->t_slot += t_b - t_a; ld->t_pread += t_c - t_b; ld->t_queue += t_d - t_c; ld->chunks_by[r]++;
                if (++ld->handled == ld->n_chunks) ld->cv.notify_all();   // (the load lives on its caller's stack: nothing of it is touched after this)
            }
        }
    }

    // Cut [0, need) into chunks, hand them to the readers and sleep until every one of them is queued on a reader's stream or given up.
    // Chunk sizes ramp up: a pread of a full chunk takes the better part of a millisecond before its copy can start -- with every reader
    // on such a chunk the link idled for the first 0.4 ms of every map; the first round is 256 KiB each, the second 1 MiB, then full chunks.
    void submit(UploadLoad &ld) {
        const size_t full = active.fetch_add(1) == 0 ? file_chunk_alone_bytes() : file_chunk_bytes();
        for (size_t pos = 0, k = 0; pos < ld.need; ++k) {
            const size_t round = k / (size_t)n_readers;
            const size_t len = std::min(ld.need - pos, round == 0 ? std::min(full, ramp_bytes(0)) : (round == 1 ? std::min(full, ramp_bytes(1)) : full));
            ld.pieces.emplace_back(pos, len);
            pos += len;
        }
        ld.n_chunks = (int64_t)ld.pieces.size();
        const double t_q = now_s();
        if (ld.n_chunks > 0) {
            {
                std::lock_guard<std::mutex> g(mu);
                for (int64_t c = 0; c < ld.n_chunks; ++c) chunks.emplace_back(&ld, c);
            }
            cv.notify_all();
            std::unique_lock<std::mutex> lk(ld.mu);
            ld.cv.wait(lk, [&] { return ld.handled == ld.n_chunks; });   // (a reader gives a chunk up at the entry's deadline: this wait ends)
        }
        active.fetch_sub(1);
        {   // (a chunk of this load on a stream that another load's reader gave up meanwhile: its bytes may never arrive)
            std::lock_guard<std::mutex> g(dead_mu);
            for (int r = 0; r < n_readers && !dead.empty(); ++r)
                if (ld.used[r] && std::find(dead.begin(), dead.end(), ld.stream_used[r]) != dead.end()) ld.stalled = true;
        }
        if (upload_trace())
            fprintf(stderr, "upload %.1f MB: chunks handled after %.3f ms; readers' sums: slot wait %.3f, pread %.3f, queueing %.3f ms; chunks by reader %d %d %d %d %d %d %d %d\n",
                    ld.need / 1e6, 1e3 * (now_s() - t_q), 1e3 * ld.t_slot, 1e3 * ld.t_pread, 1e3 * ld.t_queue, ld.chunks_by[0], ld.chunks_by[1], ld.chunks_by[2],
                    ld.chunks_by[3], ld.chunks_by[4], ld.chunks_by[5], ld.chunks_by[6], ld.chunks_by[7]);
    }

    // Readers, their chunks and threads (never destroyed: the readers are parked on the queue when the process ends).
    static UploadEngine *create(const Backend *be, int device) {
        UploadEngine *en = new UploadEngine();
        en->be = be;
        en->device = device;
        int want = (int)env_long("PDBEDA_FILE_READERS", FILE_READERS_DEFAULT);
        want = std::max(1, std::min(want, MAX_READERS));
        for (int r = 0; r < want; ++r) {
            if (!en->make_reader(en->readers[en->n_readers])) break;   // (fewer readers than asked for: what was made so far serves)
            ++en->n_readers;
        }
        for (int r = 0; r < en->n_readers; ++r) {
            try { std::thread(&UploadEngine::run, en, r).detach(); } catch (...) { en->n_readers = r; break; }
        }
        return en;
    }
};

}  // namespace pdbeda_upload
