// tests/upload_harness.cpp -- the upload engine (pdb_eda_amd/csrc/pdbeda_upload.h) on a host stand-in for the HIP runtime.
//
// Test infrastructure (CPU container): the engine's reader threads, FIFO, slot recycling, deadlines and stalled-stream replacement
// under ThreadSanitizer / AddressSanitizer (tools/sanitize_cpu.sh) and, unsanitized, in the CPU suite (tests/test_upload_engine.py).
// The stand-in: a "stream" is a worker thread that executes its queue in order (copies with memcpy, events by raising a flag), so
// copies are asynchronous to the readers exactly as on the device; "pinned" memory is malloc; "device" memory is malloc.
//
//   upload_harness many      six caller threads x files of 420 B ... 27 MB (and a load from memory), every byte compared
//   upload_harness deadline  slow copies: a load whose deadline passes in the middle of its upload beside two that keep uploading
//   upload_harness stall     a copy that never completes (a black-holed destination): the readers give their streams up and
//                            take new ones, the stuck load fails as `stalled`, later loads arrive whole
#include "pdbeda_upload.h"

#include <fcntl.h>
#include <sys/stat.h>

#include <functional>
#include <string>

using namespace pdbeda_upload;

namespace {

struct StubEvent { std::atomic<int> done{1}; };
struct Op { int kind; void *dst; const void *src; size_t bytes; StubEvent *ev; };   // 0 copy, 1 event
struct StubStream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Op> q;
    void run() {
        for (;;) {
            Op op;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !q.empty(); });
                op = q.front();
                q.pop_front();
            }
            if (op.kind == 0) {
                if (black_hole(op.dst)) for (;;) std::this_thread::sleep_for(std::chrono::seconds(3600));   // a copy that never ends
                if (copy_delay_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(copy_delay_us.load()));
                memcpy(op.dst, op.src, op.bytes);
            } else {
                op.ev->done.store(1, std::memory_order_release);
            }
        }
    }
    static std::atomic<long> copy_delay_us;
    static std::atomic<char *> hole_lo, hole_hi;
    static bool black_hole(void *p) { return (char *)p >= hole_lo.load() && (char *)p < hole_hi.load(); }
};
std::atomic<long> StubStream::copy_delay_us{0};
std::atomic<char *> StubStream::hole_lo{nullptr}, StubStream::hole_hi{nullptr};

const Backend g_stub = {
    [](int) { return (int)UP_OK; },
    [](Stream *out) { StubStream *s = new StubStream(); std::thread(&StubStream::run, s).detach(); *out = s; return (int)UP_OK; },
    [](void **out, size_t bytes) { *out = malloc(bytes); return *out ? (int)UP_OK : 2; },
    [](Event *out) { *out = new StubEvent(); return (int)UP_OK; },
    [](Event ev) { return ((StubEvent *)ev)->done.load(std::memory_order_acquire) ? (int)UP_OK : (int)UP_NOT_READY; },
    [](void *dst, const void *src, size_t bytes, Stream s) {
        StubStream *st = (StubStream *)s;
        { std::lock_guard<std::mutex> g(st->mu); st->q.push_back({0, dst, src, bytes, nullptr}); }
        st->cv.notify_one();
        return (int)UP_OK;
    },
    [](Event ev, Stream s) {
        StubStream *st = (StubStream *)s;
        ((StubEvent *)ev)->done.store(0, std::memory_order_release);
        { std::lock_guard<std::mutex> g(st->mu); st->q.push_back({1, nullptr, nullptr, 0, (StubEvent *)ev}); }
        st->cv.notify_one();
        return (int)UP_OK;
    },
};

std::string g_dir;
std::vector<char> pattern(size_t n, unsigned seed) {
    std::vector<char> v(n);
    unsigned x = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; v[i] = (char)(x >> 24); }
    return v;
}
std::string write_file(const char *name, const std::vector<char> &bytes, size_t header) {
    const std::string path = g_dir + "/" + name;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror("fopen"); exit(2); }
    std::vector<char> head(header, 'h');
    if (!head.empty()) fwrite(head.data(), 1, head.size(), f);
    if (!bytes.empty()) fwrite(bytes.data(), 1, bytes.size(), f);
    fclose(f);
    return path;
}

// what engine_copy does around submit(): the load, then a wait for an event behind every reader stream the load used
struct Result { bool ok, timed_out, stalled; };
Result upload(UploadEngine *en, int fd, int64_t offset, const char *src, char *dst, size_t need, double timeout_s, bool wait = true) {
    UploadLoad ld;
    ld.fd = fd; ld.offset = offset; ld.src = src; ld.dst = dst; ld.need = need;
    ld.timeout_s = timeout_s;
    ld.deadline = std::chrono::steady_clock::now() + std::chrono::microseconds((long)(1e6 * timeout_s));
    en->submit(ld);
    Result r = {ld.e == UP_OK && !ld.why && !ld.timed_out && !ld.stalled, ld.timed_out, ld.stalled};
    if (!wait || !r.ok) return r;
    for (int k = 0; k < en->n_readers; ++k) {
        if (!ld.used[k]) continue;
        StubEvent ev;
        g_stub.event_record(&ev, ld.stream_used[k]);
        while (g_stub.event_query(&ev) != UP_OK) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return r;
}

int fail(const char *what) { fprintf(stderr, "upload_harness: FAILED: %s\n", what); return 1; }

int test_many(UploadEngine *en) {
    const size_t sizes[] = {105 * 4, 4096, ((size_t)1 << 20) + 13, (size_t)5 << 20, (size_t)27 << 20, 0, ((size_t)8 << 20) + 1};
    const int n_sizes = (int)(sizeof sizes / sizeof sizes[0]);
    std::vector<std::vector<char>> data;
    std::vector<std::string> paths;
    for (int k = 0; k < n_sizes; ++k) {
        data.push_back(pattern(sizes[k], 7 + k));
        char name[64];
        snprintf(name, sizeof name, "f%d.bin", k);
        paths.push_back(write_file(name, data.back(), 1024));
    }
    std::atomic<int> bad{0};
    std::vector<std::thread> callers;
    for (int t = 0; t < 6; ++t)
        callers.emplace_back([&, t] {
            for (int rep = 0; rep < 4; ++rep)
                for (int k = 0; k < n_sizes; ++k) {
                    const int j = (k + t + rep) % n_sizes;
                    std::vector<char> dst(sizes[j] + 16, 'x');
                    Result r;
                    if ((t + rep) % 3 == 0) {      // from the caller's memory
                        r = upload(en, -1, 0, data[j].data(), dst.data(), sizes[j], 0.0);
                    } else {
                        const int fd = open(paths[j].c_str(), O_RDONLY);
                        r = upload(en, fd, 1024, nullptr, dst.data(), sizes[j], 0.0);
                        close(fd);
                    }
                    if (!r.ok || (sizes[j] && memcmp(dst.data(), data[j].data(), sizes[j]) != 0) || dst[sizes[j]] != 'x') bad.fetch_add(1);
                }
        });
    for (auto &c : callers) c.join();
    {   // a file shorter than the load says: a reason, not a hang
        const int fd = open(paths[1].c_str(), O_RDONLY);
        std::vector<char> dst(1 << 20);
        UploadLoad ld;
        ld.fd = fd; ld.offset = 1024; ld.dst = dst.data(); ld.need = dst.size();
        en->submit(ld);
        close(fd);
        if (!ld.why) return fail("a short file gave no reason");
    }
    if (bad.load()) return fail("bytes differ after concurrent uploads");
    printf("many: 6 callers x 4 rounds x %d sizes: all bytes equal\n", n_sizes);
    return 0;
}

int test_deadline(UploadEngine *en) {
    const size_t big = (size_t)27 << 20;
    const std::vector<char> a = pattern(big, 1), b = pattern(big, 2), c = pattern((size_t)3 << 20, 3);
    const std::string pa = write_file("da.bin", a, 0), pb = write_file("db.bin", b, 0), pc = write_file("dc.bin", c, 0);
    StubStream::copy_delay_us = 20000;   // every copy takes 20 ms: 27 MB in chunks of <= 8 MiB over three readers is > 60 ms
    std::vector<char> da(big), db(big), dc(c.size());
    Result ra{}, rb{}, rc{};
    std::thread ta([&] { const int fd = open(pa.c_str(), O_RDONLY); ra = upload(en, fd, 0, nullptr, da.data(), big, 0.03); close(fd); });
    std::thread tb([&] { const int fd = open(pb.c_str(), O_RDONLY); rb = upload(en, fd, 0, nullptr, db.data(), big, 0.0); close(fd); });
    std::thread tc([&] { const int fd = open(pc.c_str(), O_RDONLY); rc = upload(en, fd, 0, nullptr, dc.data(), c.size(), 0.0); close(fd); });
    ta.join(); tb.join(); tc.join();
    StubStream::copy_delay_us = 0;
    if (!ra.timed_out) return fail("the load with a 30 ms deadline did not time out");
    if (!rb.ok || memcmp(db.data(), b.data(), big) != 0) return fail("a load beside the timed-out one arrived damaged");
    if (!rc.ok || memcmp(dc.data(), c.data(), c.size()) != 0) return fail("the small load beside the timed-out one arrived damaged");
    // the engine serves on: the same file again, without a deadline
    {
        const int fd = open(pa.c_str(), O_RDONLY);
        std::fill(da.begin(), da.end(), 0);
        const Result r = upload(en, fd, 0, nullptr, da.data(), big, 0.0);
        close(fd);
        if (!r.ok || memcmp(da.data(), a.data(), big) != 0) return fail("the engine did not serve after a timed-out load");
    }
    printf("deadline: one load timed out, two beside it and one after it arrived whole\n");
    return 0;
}

int test_stall(UploadEngine *en) {
    const size_t big = (size_t)27 << 20;
    const std::vector<char> x = pattern(big, 11), y = pattern(big, 12);
    const std::string px = write_file("sx.bin", x, 0), py = write_file("sy.bin", y, 0);
    std::vector<char> dx(big), dy(big);
    StubStream::hole_lo = dx.data();
    StubStream::hole_hi = dx.data() + big;
    Result rx{};
    {   // X: every copy into dx blocks its stream for good.  Its own submit comes back: with the readers' slots taken by copies that
        // never end, the readers give the streams up after PDBEDA_SLOT_STALL_S and X, which has chunks on them, fails as stalled
        const int fd = open(px.c_str(), O_RDONLY);
        rx = upload(en, fd, 0, nullptr, dx.data(), big, 0.0, false);
        close(fd);
    }
    if (!rx.stalled) return fail("the load behind a copy that never ends was not reported as stalled");
    if (en->replaced.load() < 1) return fail("no reader took a new stream");
    for (int rep = 0; rep < 2; ++rep) {   // Y: whole, on the new streams
        const int fd = open(py.c_str(), O_RDONLY);
        std::fill(dy.begin(), dy.end(), 0);
        const Result ry = upload(en, fd, 0, nullptr, dy.data(), big, 0.0);
        close(fd);
        if (!ry.ok || memcmp(dy.data(), y.data(), big) != 0) return fail("a load after the stalled one did not arrive whole");
    }
    printf("stall: %d reader stream(s) given up and replaced; the stuck load failed as stalled, later loads arrived whole\n", en->replaced.load());
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    const char *which = argc > 1 ? argv[1] : "many";
    char tmpl[] = "/tmp/pdbeda_upload_XXXXXX";
    if (!mkdtemp(tmpl)) { perror("mkdtemp"); return 2; }
    g_dir = tmpl;
    if (!strcmp(which, "stall")) setenv("PDBEDA_SLOT_STALL_S", "0.2", 1);
    if (!strcmp(which, "deadline")) setenv("PDBEDA_FILE_CHUNK_KB", "1024", 1);   // (29 chunks of 20 ms each over three readers: the 30 ms deadline passes with most of the load still queued)
    UploadEngine *en = UploadEngine::create(&g_stub, 0);
    if (en->n_readers < 1) return fail("no readers");
    int rc = 2;
    if (!strcmp(which, "many")) rc = test_many(en);
    else if (!strcmp(which, "deadline")) rc = test_deadline(en);
    else if (!strcmp(which, "stall")) rc = test_stall(en);
    else fprintf(stderr, "usage: upload_harness many|deadline|stall\n");
    const std::string rm = "rm -rf " + g_dir;
    if (system(rm.c_str()) != 0) rc = rc ? rc : 3;
    fflush(stdout);
    _exit(rc);   // (the readers and the stand-in streams are parked threads: no static destructors under them)
}
