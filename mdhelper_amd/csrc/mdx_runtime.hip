// mdx_runtime.hip — runtime entry points of libmdx.so: errors, device memory,
// stream timers and the synthetic-trajectory generator used by bench.py/tests.
#include "mdx_common.hpp"

#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>

#include <unordered_map>

namespace mdx {

char *error_buffer()
{
    static thread_local char buf[1024] = "";
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 1024, fmt, ap);
    va_end(ap);
    return code;
}

// ---------------------------------------------------------------------------------------------------------------
// Process-level settings, made when the library is loaded — before this library's first HIP call, which is when the
// runtime reads its flags.
//
// (1) The runtime never page-locks caller memory on its own.  For a copy of more than GPU_PINNED_MIN_XFER_SIZE MiB
// between pageable memory and the device the HIP runtime locks the caller's pages for the DMA engine AND KEEPS the
// last few locked ranges, keyed by (host address, size), for the next copy out of the same buffer.  The entries are
// not dropped when the caller frees that memory.  When the allocator hands the same addresses out again — a new numpy
// array, a std::vector — a later copy finds the stale entry and lets the DMA engine use a registration whose pages
// are gone: the process aborts inside that copy.  This is what the two aborted test runs on record look like (round
// 4: a 52.8 MB hipMemcpy into a fresh numpy array after copies out of arrays that had been freed; round 5: the
// 4.4 MB copy of mdx_msd_cross into a std::vector, in the first build that handed pageable rows to the runtime
// again; both one run in several, each time in a copy INTO freshly allocated host memory; NOTES.md round 5), and a
// file mapping that loses its tail while such an entry exists blocks every later GPU call of the process
// (scripts/diag/mmap_truncate_probe.py).  With the threshold out of reach every pageable copy of the runtime goes
// through its own staging buffers; everything large in this library moves through the pinned ring or through pages
// it locks and unlocks itself around the copy (HostStager::copy_rows_locked).  A value the user has set stays.
//
// (2) MDX_ABORT_TRACE=<fd> (the test suite sets it; 1 or 2 = stderr): a SIGABRT handler that writes the native call
// stack to that descriptor before the handler that was there (Python's faulthandler) runs — an abort inside a runtime
// library then names the library.  When the descriptor is not 2 and descriptor 2 is a regular file (pytest's capture
// file, which dies with the process), the last 4 KB written to it — the runtime's own last words, e.g. "Memory
// access fault by GPU node ..." — are copied over first.
namespace {
struct sigaction g_prev_abort;
int g_abort_fd = 2;
void abort_trace(int sig)
{
    if (g_abort_fd != 2) {
        struct stat st;
        if (fstat(2, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            static char tail[4096];
            const off_t from = st.st_size > off_t(sizeof tail) ? st.st_size - off_t(sizeof tail) : 0;
            const ssize_t got = pread(2, tail, sizeof tail, from);
            if (got > 0) {
                static const char h1[] = "\nlibmdx: SIGABRT; last output on the captured stderr:\n";
                (void)!write(g_abort_fd, h1, sizeof h1 - 1);
                (void)!write(g_abort_fd, tail, size_t(got));
            }
        }
    }
    void *frames[64];
    const int n = backtrace(frames, 64);
    static const char head[] = "\nlibmdx: SIGABRT; native call stack (module(+offset)):\n";
    (void)!write(g_abort_fd, head, sizeof head - 1);
    backtrace_symbols_fd(frames, n, g_abort_fd);
    sigaction(SIGABRT, &g_prev_abort, nullptr);
    raise(sig);
}
__attribute__((constructor)) void mdx_process_init()
{
    setenv("GPU_PINNED_MIN_XFER_SIZE", "1048576", 0);       // MiB
    const char *t = getenv("MDX_ABORT_TRACE");
    if (t && *t && *t != '0') {
        const int fd = atoi(t);
        // (a child process inherits the variable but not the descriptor: what is not open here is stderr)
        g_abort_fd = fd > 2 && fcntl(fd, F_GETFD) != -1 ? fd : 2;
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = abort_trace;
        sigemptyset(&sa.sa_mask);
        sa.sa_flags = SA_NODEFER;
        sigaction(SIGABRT, &sa, &g_prev_abort);
    }
}
}  // namespace

int set_device(int dev)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(MDX_ERR_NO_DEVICE,
                    "no HIP device is visible (%s); libmdx has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (dev < 0 || dev >= n)
        return fail(MDX_ERR_NO_DEVICE, "device %d out of range [0, %d)", dev, n);
    MDX_HIP(hipSetDevice(dev));
    return MDX_OK;
}

namespace {
// Device blocks of destroyed handles and freed DeviceArrays, kept for the next analysis object.  Measured on
// MI355X / ROCm 7.2 (scripts/diag/onsager_stalls.py): an Onsager(...).run() per call at C4 size returned ~50 GB
// to the driver at the end of every analysis, and every other analysis then waited 1.1 - 3.3 s inside the
// hipMalloc of its 24.6 GB block; creating and freeing the small buffers cost another ~10 ms per object.
struct BlockCache {
    std::mutex m;
    std::vector<std::pair<size_t, void *>> blocks[64];   // per device: (bytes, pointer)
    size_t cached[64] = {};
    size_t limit[64] = {};                               // 0: not asked yet
    std::unordered_map<void *, size_t> live[64];         // blocks handed out by mdx_malloc
};
BlockCache &block_cache()
{
    static BlockCache *c = new BlockCache();   // never destroyed: the HIP runtime may be gone at exit
    return *c;
}

// at most a quarter of the device's free memory at first use (and at least 4 GiB) stays cached; MDX_CACHE_GB overrides
size_t cache_limit(BlockCache &c, int dev)
{
    if (c.limit[dev] == 0) {
        size_t lim = size_t(4) << 30;
        if (const char *e = getenv("MDX_CACHE_GB")) {
            lim = size_t(atoll(e) < 0 ? 0 : atoll(e)) << 30;
        } else {
            // a quarter of what is FREE now (first use: nothing of ours is cached yet), not of the total: another
            // rank or library sharing the device keeps its share
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 4 > lim)
                lim = free_b / 4;
            else
                (void)hipGetLastError();
        }
        c.limit[dev] = lim ? lim : 1;
    }
    return c.limit[dev];
}

// smallest cached block of `dev` that fits `want` without wasting more than half of itself (lock held)
bool take_cached(BlockCache &c, int dev, size_t want, void **ptr, size_t *bytes)
{
    auto &v = c.blocks[dev];
    size_t best = v.size();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i].first >= want && v[i].first <= 2 * want && (best == v.size() || v[i].first < v[best].first))
            best = i;
    if (best == v.size())
        return false;
    *ptr = v[best].second;
    *bytes = v[best].first;
    c.cached[dev] -= v[best].first;
    v[best] = v.back();
    v.pop_back();
    return true;
}

// hipFree every cached block of a device (the current one must be `dev`); returns the bytes given back
size_t flush_block_cache(int dev)
{
    if (dev < 0 || dev >= 64)
        return 0;
    BlockCache &c = block_cache();
    std::lock_guard<std::mutex> lk(c.m);
    for (auto &b : c.blocks[dev])
        (void)hipFree(b.second);
    c.blocks[dev].clear();
    const size_t freed = c.cached[dev];
    c.cached[dev] = 0;
    return freed;
}
}  // namespace

size_t cached_device_bytes(int dev)
{
    if (dev < 0 || dev >= 64)
        return 0;
    BlockCache &c = block_cache();
    std::lock_guard<std::mutex> lk(c.m);
    return c.cached[dev];
}

int DeviceBuffer::ensure(size_t need)
{
    if (need <= bytes)
        return MDX_OK;
    release();
    // round up so a slowly growing batch does not reallocate every call
    size_t want = (need + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        BlockCache &c = block_cache();
        std::lock_guard<std::mutex> lk(c.m);
        if (take_cached(c, dev, want, &ptr, &bytes))
            return MDX_OK;
    }
    hipError_t e = hipMalloc(&ptr, want);
    if (e == hipErrorOutOfMemory) {
        // give the cache back before giving up
        (void)hipGetLastError();
        flush_block_cache(dev);
        e = hipMalloc(&ptr, want);
    }
    if (e != hipSuccess)
        ptr = nullptr;
    MDX_HIP(e);
    bytes = want;
    return MDX_OK;
}

void DeviceBuffer::release()
{
    if (ptr)
        (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

void DeviceBuffer::recycle()
{
    if (!ptr)
        return;
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        BlockCache &c = block_cache();
        std::lock_guard<std::mutex> lk(c.m);
        if (c.cached[dev] + bytes <= cache_limit(c, dev)) {
            c.blocks[dev].emplace_back(bytes, ptr);
            c.cached[dev] += bytes;
            ptr = nullptr;
            bytes = 0;
            return;
        }
    }
    release();
}

namespace {
struct StreamPool {
    std::mutex m;
    std::vector<hipStream_t> idle[64];
};
StreamPool &stream_pool()
{
    static StreamPool *p = new StreamPool();
    return *p;
}
}  // namespace

int stream_acquire(hipStream_t *out)
{
    int dev = 0;
    MDX_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64) {
        StreamPool &p = stream_pool();
        std::lock_guard<std::mutex> lk(p.m);
        if (!p.idle[dev].empty()) {
            *out = p.idle[dev].back();
            p.idle[dev].pop_back();
            return MDX_OK;
        }
    }
    MDX_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return MDX_OK;
}

void stream_release(hipStream_t stream)
{
    if (!stream)
        return;
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        StreamPool &p = stream_pool();
        std::lock_guard<std::mutex> lk(p.m);
        if (p.idle[dev].size() < 16) {
            p.idle[dev].push_back(stream);
            return;
        }
    }
    (void)hipStreamDestroy(stream);
}

hipEvent_t StreamTimer::begin()
{
    if (!enabled)
        return nullptr;
    hipEvent_t ev;
    if (!pool.empty()) {
        ev = pool.back();
        pool.pop_back();
    } else if (hipEventCreate(&ev) != hipSuccess) {
        return nullptr;
    }
    (void)hipEventRecord(ev, stream);
    return ev;
}

void StreamTimer::end(hipEvent_t start)
{
    ++launches;
    if (!enabled || !start)
        return;
    hipEvent_t ev;
    if (!pool.empty()) {
        ev = pool.back();
        pool.pop_back();
    } else if (hipEventCreate(&ev) != hipSuccess) {
        pool.push_back(start);
        return;
    }
    (void)hipEventRecord(ev, stream);
    pending.emplace_back(start, ev);
}

static void timer_recycle(StreamTimer &t)
{
    // pairs dropped by reset() while their events were still in flight: recycle them once they have fired
    // (an event recorded again before it has fired gave elapsed times that spanned two uses)
    for (size_t i = 0; i < t.limbo.size();) {
        if (hipEventQuery(t.limbo[i].second) == hipSuccess) {
            t.pool.push_back(t.limbo[i].first);
            t.pool.push_back(t.limbo[i].second);
            t.limbo[i] = t.limbo.back();
            t.limbo.pop_back();
        } else {
            (void)hipGetLastError();
            ++i;
        }
    }
}

void StreamTimer::collect()
{
    timer_recycle(*this);
    for (auto &p : pending) {
        // a bracket whose end has not fired yet is waited for, not dropped: every collected figure
        // covers every bracket closed before the call, on whatever stream it ran
        float ms = 0.f;
        if (hipEventSynchronize(p.second) == hipSuccess &&
            hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            total_ms += ms;
            pool.push_back(p.first);
            pool.push_back(p.second);
        } else {
            (void)hipGetLastError();
            limbo.push_back(p);
        }
    }
    pending.clear();
}

void StreamTimer::reset()
{
    timer_recycle(*this);
    for (auto &p : pending)
        limbo.push_back(p);
    pending.clear();
    total_ms = 0.0;
    launches = 0;
}

void StreamTimer::destroy()
{
    collect();
    for (auto &p : limbo) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    limbo.clear();
    for (auto ev : pool)
        (void)hipEventDestroy(ev);
    pool.clear();
}

// ------------------------------------------------------------------ host workers + pinned ring

void HostWorkers::start(int n_threads)
{
    if (!threads.empty() || n_threads <= 1)
        return;
    for (int t = 1; t < n_threads; ++t)
        threads.emplace_back([this] { loop(); });
}

void HostWorkers::loop()
{
    uint64_t seen = 0;
    for (;;) {
        const std::function<void(int)> *fn;
        int total;
        {
            std::unique_lock<std::mutex> lk(m);
            cv_work.wait(lk, [&] { return stopping || generation != seen; });
            if (stopping)
                return;
            seen = generation;
            // woken after the call it was woken for has ended (its tasks were all claimed by others): nothing to
            // take part in — and `next` may already belong to the following call, so it is not touched
            if (!job)
                continue;
            fn = job;
            total = n_tasks;
            ++active;
        }
        for (int i; (i = next.fetch_add(1)) < total;)
            (*fn)(i);
        {
            std::lock_guard<std::mutex> lk(m);
            if (--active == 0)
                cv_done.notify_all();
        }
    }
}

void HostWorkers::parallel_for(int n, const std::function<void(int)> &fn)
{
    if (n <= 0)
        return;
    if (threads.empty() || n == 1) {
        for (int i = 0; i < n; ++i)
            fn(i);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(m);
        job = &fn;
        n_tasks = n;
        next.store(0);
        ++generation;
    }
    cv_work.notify_all();
    for (int i; (i = next.fetch_add(1)) < n;)
        fn(i);
    // every task has been claimed; wait for the workers that are still inside one (a worker joins a call — takes
    // `job`, counts itself in `active` — in ONE critical section, so none can be between the two here), then
    // close the call: a worker that wakes up late finds no job and goes back to sleep
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return active == 0; });
    job = nullptr;
}

void HostWorkers::stop()
{
    {
        std::lock_guard<std::mutex> lk(m);
        stopping = true;
    }
    cv_work.notify_all();
    for (std::thread &t : threads)
        t.join();
    threads.clear();
    stopping = false;
}

static int io_threads()
{
    static const int n = [] {
        const char *e = getenv("MDX_IO_THREADS");
        int v = e ? atoi(e) : 8;
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && v > hw)
            v = hw;
        return v < 1 ? 1 : (v > 64 ? 64 : v);
    }();
    return n;
}

int HostStager::ensure(int device, size_t chunk_bytes)
{
    if (dev >= 0 && dev != device)
        return fail(MDX_ERR_STATE, "staging ring is bound to device %d", dev);
    dev = device;
    if (!io) {
        MDX_HIP(hipStreamCreateWithFlags(&io, hipStreamNonBlocking));
        MDX_HIP(hipEventCreateWithFlags(&ev_batch, hipEventDisableTiming));
        for (int b = 0; b < NBUF; ++b)
            MDX_HIP(hipEventCreateWithFlags(&ev_sent[b], hipEventDisableTiming));
        workers.start(io_threads());
    }
    if (pinned_bytes < chunk_bytes) {
        MDX_TRY(drain());
        for (int b = 0; b < NBUF; ++b) {
            if (pinned[b])
                MDX_HIP(hipHostFree(pinned[b]));
            pinned[b] = nullptr;
        }
        pinned_bytes = 0;
        for (int b = 0; b < NBUF; ++b)
            MDX_HIP(hipHostMalloc(&pinned[b], chunk_bytes, hipHostMallocDefault));
        pinned_bytes = chunk_bytes;
    }
    return MDX_OK;
}

int HostStager::after(hipStream_t producer)
{
    // a throw-away event: recorded on the producer's stream, waited for by io, and gone before
    // the producer's stream can be (destroying an event that a stream still waits for is allowed;
    // the wait itself holds what it needs)
    hipEvent_t ev;
    MDX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, producer);
    if (e == hipSuccess)
        e = hipStreamWaitEvent(io, ev, 0);
    (void)hipEventDestroy(ev);
    MDX_HIP(e);
    return MDX_OK;
}

int HostStager::acquire(int *b, void **host)
{
    const int k = int(turn++ % NBUF);
    if (in_flight[k]) {
        MDX_HIP(hipEventSynchronize(ev_sent[k]));
        in_flight[k] = false;
    }
    *b = k;
    *host = pinned[k];
    return MDX_OK;
}

int HostStager::send(int b, void *d_dst, size_t bytes)
{
    MDX_HIP(hipMemcpyAsync(d_dst, pinned[b], bytes, hipMemcpyHostToDevice, io));
    MDX_HIP(hipEventRecord(ev_sent[b], io));
    in_flight[b] = true;
    return MDX_OK;
}

int HostStager::finish(hipStream_t consumer)
{
    MDX_HIP(hipEventRecord(ev_batch, io));
    MDX_HIP(hipStreamWaitEvent(consumer, ev_batch, 0));
    return MDX_OK;
}

int HostStager::drain()
{
    for (int b = 0; b < NBUF; ++b)
        if (in_flight[b]) {
            MDX_HIP(hipEventSynchronize(ev_sent[b]));
            in_flight[b] = false;
        }
    return MDX_OK;
}

HostStager &device_stager(int dev)
{
    // (never destroyed: at process exit the HIP runtime may already be gone)
    static HostStager *rings[64] = {};
    static std::mutex m;
    std::lock_guard<std::mutex> lk(m);
    const int i = dev < 0 ? 0 : dev % 64;
    if (!rings[i])
        rings[i] = new HostStager();
    return *rings[i];
}

// Memory the DMA engine can read where it lies: a range registered through mdx_host_register that covers
// [src, src + bytes) whole, or memory this library knows nothing about whose two ends the runtime reports as host
// allocations (hipHostMalloc, a caller's own hipHostRegister).  A range that only touches a registration of ours
// is pageable as far as the copy is concerned: a DMA that runs off the registered object faults.
static bool host_is_device_readable(const void *src, size_t bytes)
{
    const int known = host_range_registered(src, bytes);
    if (known != 0)
        return known > 0;
    hipPointerAttribute_t attr, attr_end;
    if (hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost &&
        hipPointerGetAttributes(&attr_end, static_cast<const uint8_t *>(src) + bytes - 1) == hipSuccess &&
        attr_end.type == hipMemoryTypeHost)
        return true;
    (void)hipGetLastError();
    return false;
}

int HostStager::upload(int device, hipStream_t consumer, void *d_dst, const void *src, size_t bytes)
{
    if (bytes == 0)
        return MDX_OK;
    std::lock_guard<std::mutex> guard(lock);
    // memory the device can read where it lies (hipHostMalloc / hipHostRegister, e.g. through
    // mdx_host_register): one DMA, no staging copy
    // (both ends are asked: a caller may have registered a shorter range than [src, src + bytes), and a DMA
    // that runs off the registered object faults; such a buffer goes through the ring like pageable memory)
    if (host_is_device_readable(src, bytes)) {
        MDX_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, consumer));
        return MDX_OK;
    }
    if (bytes < (size_t(1) << 20)) {
        // small: the runtime's own staging is as good
        MDX_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, consumer));
        return MDX_OK;
    }
    const size_t chunk = size_t(16) << 20;
    MDX_TRY(ensure(device, chunk));
    MDX_TRY(after(consumer));
    const uint8_t *from = static_cast<const uint8_t *>(src);
    uint8_t *to = static_cast<uint8_t *>(d_dst);
    const int parts = workers.size();
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = bytes - off < chunk ? bytes - off : chunk;
        int b;
        void *host;
        MDX_TRY(acquire(&b, &host));
        // 64-byte aligned slices, one per thread
        const size_t slice = ((n + parts - 1) / parts + 63) & ~size_t(63);
        const std::function<void(int)> copy = [&](int t) {
            const size_t lo = size_t(t) * slice;
            if (lo < n)
                memcpy(static_cast<uint8_t *>(host) + lo, from + off + lo, n - lo < slice ? n - lo : slice);
        };
        workers.parallel_for(parts, copy);
        MDX_TRY(send(b, to + off, n));
    }
    return finish(consumer);
}

int HostStager::download(int device, hipStream_t producer, void *dst, const void *d_src, size_t bytes)
{
    if (bytes == 0)
        return MDX_OK;
    std::lock_guard<std::mutex> guard(lock);
    const size_t chunk = size_t(16) << 20;
    MDX_TRY(ensure(device, chunk));
    if (producer)
        MDX_TRY(after(producer));
    uint8_t *to = static_cast<uint8_t *>(dst);
    const uint8_t *from = static_cast<const uint8_t *>(d_src);
    // two chunks in flight: while the host copies chunk k out of its pinned buffer, the DMA of chunk k + 1 runs
    int pend_b = -1;
    size_t pend_off = 0, pend_n = 0;
    auto retire = [&]() -> int {
        if (pend_b < 0)
            return MDX_OK;
        MDX_HIP(hipEventSynchronize(ev_sent[pend_b]));
        in_flight[pend_b] = false;
        const int parts = workers.size();
        const size_t slice = ((pend_n + parts - 1) / parts + 63) & ~size_t(63);
        uint8_t *host = static_cast<uint8_t *>(pinned[pend_b]);
        const std::function<void(int)> copy = [&](int t) {
            const size_t lo = size_t(t) * slice;
            if (lo < pend_n)
                memcpy(to + pend_off + lo, host + lo, pend_n - lo < slice ? pend_n - lo : slice);
        };
        workers.parallel_for(parts, copy);
        pend_b = -1;
        return MDX_OK;
    };
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = bytes - off < chunk ? bytes - off : chunk;
        int b;
        void *host;
        MDX_TRY(acquire(&b, &host));
        MDX_HIP(hipMemcpyAsync(host, from + off, n, hipMemcpyDeviceToHost, io));
        MDX_HIP(hipEventRecord(ev_sent[b], io));
        in_flight[b] = true;
        MDX_TRY(retire());               // the chunk before this one, while this one's DMA runs
        pend_b = b;
        pend_off = off;
        pend_n = n;
    }
    return retire();
}

// [ptr, ptr + bytes) lies in anonymous private memory (malloc / numpy / mmap(MAP_ANONYMOUS)) as /proc/self/maps
// tells it: no path, or [heap] / [stack] / [anon:...]; every page of the range covered.  Anything the table does not
// vouch for is treated as file-backed.
static bool host_range_is_anonymous(const void *ptr, size_t bytes)
{
    FILE *f = fopen("/proc/self/maps", "r");
    if (!f)
        return false;
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr), b = a + bytes;
    uintptr_t covered_to = a;
    bool ok = true;
    char line[512];
    while (ok && covered_to < b && fgets(line, sizeof line, f)) {
        unsigned long lo = 0, hi = 0, off = 0, ino = 0;
        char perms[8] = {0}, devs[16] = {0}, path[256] = {0};
        const int n = sscanf(line, "%lx-%lx %7s %lx %15s %lu %255s", &lo, &hi, perms, &off, devs, &ino, path);
        if (n < 6 || hi <= covered_to)
            continue;
        if (lo > covered_to)
            ok = false;                               // a hole in the range
        else if (ino != 0 || (n >= 7 && path[0] == '/') || perms[3] != 'p')
            ok = false;                               // file-backed or shared
        else
            covered_to = hi;
    }
    fclose(f);
    return ok && covered_to >= b;
}

int HostStager::upload_rows(int device, hipStream_t consumer, void *d_dst, const void *src, size_t row_bytes,
                            size_t src_stride, size_t n_rows)
{
    if (row_bytes == 0 || n_rows == 0)
        return MDX_OK;
    if (src_stride == row_bytes)
        return upload(device, consumer, d_dst, src, row_bytes * n_rows);
    // Pageable rows of a few KB and more, a couple of pages apart, in anonymous memory: locked slice by slice and
    // read by the DMA engine where they lie (copy_rows_locked: 53 - 54 GB/s for 15 .. 30 KB rows 120 KB apart, where
    // the copy threads' gather into the ring reaches 34 - 40; no host core touches the data).  Short rows keep the
    // gather: a DMA descriptor per 12-byte row is no way to move a particle's track.  MDX_RING_ROWS=1 (A/B hook)
    // keeps the ring for every row length.
    static const bool ring_rows = getenv("MDX_RING_ROWS") != nullptr;
    if (row_bytes >= 4096 && src_stride >= row_bytes + 8192 && !ring_rows &&
        host_range_registered(src, (n_rows - 1) * src_stride + row_bytes) == 0) {
        hipPointerAttribute_t attr;
        const bool pageable = hipPointerGetAttributes(&attr, src) != hipSuccess || attr.type == hipMemoryTypeUnregistered;
        (void)hipGetLastError();
        if (pageable && host_range_is_anonymous(src, (n_rows - 1) * src_stride + row_bytes)) {
            const int rc = copy_rows_locked(device, consumer, d_dst, src, row_bytes, src_stride, n_rows);
            if (rc <= 0)
                return rc;
        }
    }
    std::lock_guard<std::mutex> guard(lock);
    const uint8_t *from = static_cast<const uint8_t *>(src);
    // page-locked / registered memory (both ends of the strided range): one 2-D DMA where it lies
    if (host_is_device_readable(src, (n_rows - 1) * src_stride + row_bytes)) {
        MDX_HIP(hipMemcpy2DAsync(d_dst, row_bytes, src, src_stride, row_bytes, n_rows, hipMemcpyHostToDevice, consumer));
        return MDX_OK;
    }
    // pageable: whole rows gathered into the pinned ring by the copy threads, 16 MB at a time
    const size_t chunk = size_t(16) << 20;
    const size_t rows_per = std::max<size_t>(1, chunk / row_bytes);
    MDX_TRY(ensure(device, std::max(chunk, row_bytes)));
    MDX_TRY(after(consumer));
    uint8_t *to = static_cast<uint8_t *>(d_dst);
    const int parts = workers.size();
    for (size_t r0 = 0; r0 < n_rows; r0 += rows_per) {
        const size_t nr = std::min(rows_per, n_rows - r0);
        int b;
        void *host;
        MDX_TRY(acquire(&b, &host));
        const size_t per = (nr + parts - 1) / parts;
        const std::function<void(int)> copy = [&](int t) {
            const size_t lo = size_t(t) * per, hi = std::min(nr, lo + per);
            for (size_t r = lo; r < hi; ++r)
                memcpy(static_cast<uint8_t *>(host) + r * row_bytes, from + (r0 + r) * src_stride, row_bytes);
        };
        workers.parallel_for(parts, copy);
        MDX_TRY(send(b, to + r0 * row_bytes, nr * row_bytes));
    }
    return finish(consumer);
}

// n_rows rows of row_bytes, src_stride apart in pageable anonymous host memory -> contiguous rows in HBM, read by
// the DMA engine where they lie: the rows are cut into slices of ~128 MB, and every slice is page-locked
// (hipHostRegister on the whole pages its rows span), copied by ONE 2-D DMA on a stream of its own and unlocked
// again (hipHostUnregister) by one of the copy threads, several slices side by side.  Locking is cheap (0.25 ms per
// 128 MB) and every slice is unregistered before the call returns — unlike the runtime's own pageable path, whose kept
// registrations are what mdx_process_init switches off.  (Anonymous memory only: whatever the driver may still hold
// for pages the DMA engine has read is harmless there — pages that go away under a mapping that stays come back as
// fresh pages — and is not for a file mapping, whose pages beyond a later truncation never come back: NOTES.md
// round 5, the second mapped-file probe.)  Measured on the 12 GB of C4 in quarter-column chunks:
// 53 - 54 GB/s with 2 .. 8 threads (the runtime's implicit route 57.5, the ring's gather 34 - 40;
// profiles/r05_register_slices.json).
// Callers guarantee (upload_rows checks): rows of >= 4 KB that lie >= 2 pages apart, so the page spans of two slices
// never share a page; anonymous private memory (rows of a file mapping go through the ring: a registration is not
// something to hold on pages another program can truncate); no page of the range registered by the caller.
// Ordering: the copies start after what `consumer` has queued so far, and all of them have ended when the call
// returns.
int HostStager::copy_rows_locked(int device, hipStream_t consumer, void *d_dst, const void *src, size_t row_bytes,
                                 size_t src_stride, size_t n_rows)
{
    if (row_bytes == 0 || n_rows == 0)
        return MDX_OK;
    std::lock_guard<std::mutex> guard(lock);
    if (dev >= 0 && dev != device)
        return fail(MDX_ERR_STATE, "staging ring is bound to device %d", dev);
    dev = device;
    workers.start(io_threads());
    hipEvent_t ev_ready = nullptr;
    if (consumer) {
        MDX_HIP(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
        const hipError_t e = hipEventRecord(ev_ready, consumer);
        if (e != hipSuccess) {
            (void)hipEventDestroy(ev_ready);
            MDX_HIP(e);
        }
    }
    const size_t rows_per = std::max<size_t>(1, (size_t(128) << 20) / row_bytes);
    const int n_slices = int((n_rows + rows_per - 1) / rows_per);
    const uint8_t *from = static_cast<const uint8_t *>(src);
    uint8_t *to = static_cast<uint8_t *>(d_dst);
    const uintptr_t page = uintptr_t(sysconf(_SC_PAGESIZE) > 0 ? sysconf(_SC_PAGESIZE) : 4096);
    std::mutex err_lock;
    hipError_t first_err = hipSuccess;
    std::atomic<bool> not_lockable{false};
    const std::function<void(int)> one = [&](int k) {
        if (not_lockable.load())
            return;
        const size_t r0 = size_t(k) * rows_per, nr = std::min(rows_per, n_rows - r0);
        const uintptr_t a = reinterpret_cast<uintptr_t>(from + r0 * src_stride);
        const uintptr_t b = a + (nr - 1) * src_stride + row_bytes;
        void *lo = reinterpret_cast<void *>(a & ~(page - 1));
        const size_t span = size_t(((b + page - 1) & ~(page - 1)) - (a & ~(page - 1)));
        hipError_t e = hipSetDevice(device);
        hipStream_t s = nullptr;
        if (e == hipSuccess && stream_acquire(&s) != MDX_OK)
            e = hipErrorUnknown;
        bool locked = false;
        if (e == hipSuccess) {
            e = hipHostRegister(lo, span, hipHostRegisterDefault);
            locked = e == hipSuccess;
            if (!locked) {
                // pages that cannot be locked (a limit, a kind of memory the table did not tell apart): the caller
                // takes the ring for the whole range
                (void)hipGetLastError();
                not_lockable.store(true);
                if (s)
                    stream_release(s);
                return;
            }
        }
        if (e == hipSuccess && ev_ready)
            e = hipStreamWaitEvent(s, ev_ready, 0);
        if (e == hipSuccess)
            e = hipMemcpy2DAsync(to + r0 * row_bytes, row_bytes, from + r0 * src_stride, src_stride, row_bytes, nr,
                                 hipMemcpyHostToDevice, s);
        if (s) {
            const hipError_t e2 = hipStreamSynchronize(s);      // also after an error: no copy outlives its pages' lock
            if (e == hipSuccess)
                e = e2;
            stream_release(s);
        }
        if (locked) {
            const hipError_t e3 = hipHostUnregister(lo);
            if (e == hipSuccess)
                e = e3;
        }
        if (e != hipSuccess) {
            std::lock_guard<std::mutex> lk(err_lock);
            if (first_err == hipSuccess)
                first_err = e;
        }
    };
    workers.parallel_for(n_slices, one);
    if (ev_ready)
        (void)hipEventDestroy(ev_ready);
    if (first_err != hipSuccess) {
        (void)hipGetLastError();
        return fail(MDX_ERR_HIP, "copy of page-locked row slices failed: %s", hipGetErrorString(first_err));
    }
    return not_lockable.load() ? 1 : MDX_OK;          // 1: nothing went wrong, but the rows are not (all) there
}

void HostStager::destroy()
{
    if (dev >= 0)
        (void)hipSetDevice(dev);
    if (io)
        (void)hipStreamSynchronize(io);
    for (int b = 0; b < NBUF; ++b) {
        if (ev_sent[b])
            (void)hipEventDestroy(ev_sent[b]);
        ev_sent[b] = nullptr;
        in_flight[b] = false;
        if (pinned[b])
            (void)hipHostFree(pinned[b]);
        pinned[b] = nullptr;
    }
    pinned_bytes = 0;
    if (ev_batch)
        (void)hipEventDestroy(ev_batch);
    ev_batch = nullptr;
    if (io)
        (void)hipStreamDestroy(io);
    io = nullptr;
    workers.stop();
    dev = -1;
}

int StagePipeline::ensure()
{
    if (copy_stream)
        return MDX_OK;
    MDX_TRY(stream_acquire(&copy_stream));
    for (int b = 0; b < 2; ++b) {
        MDX_HIP(hipEventCreateWithFlags(&ev_filled[b], hipEventDisableTiming));
        MDX_HIP(hipEventCreateWithFlags(&ev_consumed[b], hipEventDisableTiming));
    }
    return MDX_OK;
}

void StagePipeline::destroy()
{
    if (!copy_stream)
        return;
    (void)hipStreamSynchronize(copy_stream);
    stream_release(copy_stream);
    copy_stream = nullptr;
    for (int b = 0; b < 2; ++b) {
        if (ev_filled[b]) (void)hipEventDestroy(ev_filled[b]);
        if (ev_consumed[b]) (void)hipEventDestroy(ev_consumed[b]);
        ev_filled[b] = ev_consumed[b] = nullptr;
        busy[b] = false;
    }
}

}  // namespace mdx

using namespace mdx;

// ----------------------------------------------------------------------------
// Philox-4x32-10 (Salmon et al., SC'11) keyed by (seed) with counter (atom, frame)
// ----------------------------------------------------------------------------
__device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0;
        k1 += W1;
    }
}

__device__ inline float u01(uint32_t x) { return (float(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

template <typename OutT>
__global__ __launch_bounds__(256) void synth_walk_kernel(OutT *__restrict__ out, int64_t n_frames,
                                                         int64_t n_atoms, float lx, float ly,
                                                         float lz, float sigma, uint32_t k0,
                                                         uint32_t k1, int wrap)
{
    int64_t atom = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (atom >= n_atoms)
        return;
    const float L[3] = {lx, ly, lz};
    float x[3];
    uint32_t c[4] = {uint32_t(atom), uint32_t(atom >> 32), 0u, 0x5eedu};
    philox4x32_10(c, k0, k1);
    for (int k = 0; k < 3; ++k)
        x[k] = u01(c[k]) * L[k];
    for (int64_t f = 0; f < n_frames; ++f) {
        if (f > 0) {
            uint32_t d[4] = {uint32_t(atom), uint32_t(atom >> 32), uint32_t(f), uint32_t(f >> 32)};
            philox4x32_10(d, k0, k1);
            // Box-Muller: two normals from (d0,d1), one from (d2,d3)
            float r0 = sqrtf(-2.0f * __logf(u01(d[0])));
            float r1 = sqrtf(-2.0f * __logf(u01(d[2])));
            float s0, c0, s1, c1;
            __sincosf(6.28318530718f * u01(d[1]), &s0, &c0);
            __sincosf(6.28318530718f * u01(d[3]), &s1, &c1);
            float g[3] = {r0 * c0, r0 * s0, r1 * c1};
            (void)s1;
            for (int k = 0; k < 3; ++k) {
                float v = x[k] + sigma * g[k];
                if (wrap) {
                    v -= L[k] * floorf(v / L[k]);
                    if (!(v < L[k]))
                        v -= L[k];
                    if (v < 0.0f)
                        v = 0.0f;
                }
                x[k] = v;
            }
        }
        OutT *o = out + (f * n_atoms + atom) * 3;
        o[0] = (OutT)x[0];
        o[1] = (OutT)x[1];
        o[2] = (OutT)x[2];
    }
}

extern "C" {

const char *mdx_last_error(void) { return error_buffer(); }

int mdx_version(void) { return MDX_VERSION; }

int mdx_device_count(int *count)
{
    MDX_REQUIRE(count != nullptr, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return MDX_OK;
}

int mdx_device_info(int dev, char *name, size_t name_len, int *compute_units, size_t *hbm_bytes,
                    size_t *hbm_free_bytes)
{
    MDX_TRY(set_device(dev));
    hipDeviceProp_t prop;
    MDX_HIP(hipGetDeviceProperties(&prop, dev));
    if (name && name_len) {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units)
        *compute_units = prop.multiProcessorCount;
    size_t free_b = 0, total_b = 0;
    MDX_HIP(hipMemGetInfo(&free_b, &total_b));
    if (hbm_bytes)
        *hbm_bytes = total_b;
    if (hbm_free_bytes)
        *hbm_free_bytes = free_b;
    return MDX_OK;
}

int mdx_malloc(int dev, size_t bytes, void **dptr)
{
    MDX_REQUIRE(dptr != nullptr, "dptr is NULL");
    MDX_TRY(set_device(dev));
    // blocks of a MiB and more come out of (and go back to, mdx_free) the per-device cache
    const bool large = bytes >= (size_t(1) << 20) && dev < 64;
    const size_t want = large ? (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1) : (bytes ? bytes : 1);
    BlockCache &c = block_cache();
    if (large) {
        std::lock_guard<std::mutex> lk(c.m);
        size_t got = 0;
        if (take_cached(c, dev, want, dptr, &got)) {
            c.live[dev][*dptr] = got;
            return MDX_OK;
        }
    }
    hipError_t e = hipMalloc(dptr, want);
    if (e == hipErrorOutOfMemory) {
        // the recycled blocks go back first
        (void)hipGetLastError();
        flush_block_cache(dev);
        e = hipMalloc(dptr, want);
    }
    MDX_HIP(e);
    if (large) {
        std::lock_guard<std::mutex> lk(c.m);
        c.live[dev][*dptr] = want;
    }
    return MDX_OK;
}

int mdx_free(int dev, void *dptr)
{
    MDX_TRY(set_device(dev));
    if (!dptr)
        return MDX_OK;
    BlockCache &c = block_cache();
    size_t bytes = 0;
    if (dev < 64) {
        std::lock_guard<std::mutex> lk(c.m);
        auto it = c.live[dev].find(dptr);
        if (it != c.live[dev].end()) {
            bytes = it->second;
            c.live[dev].erase(it);
        }
    }
    if (bytes) {
        // like hipFree: nothing queued may still touch the block when it changes hands
        MDX_HIP(hipDeviceSynchronize());
        std::lock_guard<std::mutex> lk(c.m);
        if (c.cached[dev] + bytes <= cache_limit(c, dev)) {
            c.blocks[dev].emplace_back(bytes, dptr);
            c.cached[dev] += bytes;
            return MDX_OK;
        }
    }
    MDX_HIP(hipFree(dptr));
    return MDX_OK;
}

int mdx_upload_rows(int dev, void *d_dst, const void *src, size_t row_bytes, size_t src_stride, size_t n_rows)
{
    MDX_REQUIRE(d_dst && src, "NULL argument");
    MDX_REQUIRE(src_stride >= row_bytes, "rows overlap: src_stride < row_bytes");
    MDX_TRY(set_device(dev));
    hipStream_t stream = nullptr;
    MDX_TRY(stream_acquire(&stream));
    const int rc = device_stager(dev).upload_rows(dev, stream, d_dst, src, row_bytes, src_stride, n_rows);
    const hipError_t e = hipStreamSynchronize(stream);   // also on an error: no copy outlives the call
    stream_release(stream);
    MDX_TRY(rc);
    MDX_HIP(e);
    return MDX_OK;
}

int mdx_cached_bytes(int dev, size_t *bytes)
{
    MDX_REQUIRE(bytes, "NULL argument");
    MDX_REQUIRE(dev >= 0 && dev < 64, "bad device");
    *bytes = cached_device_bytes(dev);
    return MDX_OK;
}

int mdx_trim_cache(int dev, size_t *freed_bytes)
{
    MDX_TRY(set_device(dev));
    MDX_HIP(hipDeviceSynchronize());
    const size_t freed = flush_block_cache(dev);
    if (freed_bytes)
        *freed_bytes = freed;
    return MDX_OK;
}

int mdx_upload(int dev, void *d_dst, const void *src, size_t bytes)
{
    MDX_REQUIRE(d_dst && src, "NULL argument");
    MDX_TRY(set_device(dev));
    hipStream_t stream = nullptr;
    MDX_TRY(stream_acquire(&stream));
    const int rc = device_stager(dev).upload(dev, stream, d_dst, src, bytes);
    const hipError_t e = hipStreamSynchronize(stream);   // also on an error: no copy outlives the call
    stream_release(stream);
    MDX_TRY(rc);
    MDX_HIP(e);
    return MDX_OK;
}

// Large whole-array copies between pageable host memory and HBM go through the library's own pinned ring: its
// copies are asynchronous to the engines' streams, which the staging pipelines are built on, and the runtime's own
// path for them page-locks caller memory and keeps the registrations (mdx_process_init above switches it off: the
// runtime stages what is left to it).
// Strided rows are locked and copied slice by slice (HostStager::copy_rows_locked).  hipMemcpy's ordering is kept:
// the copy starts after everything queued on the device before the call and has ended when the call returns.
int mdx_memcpy_h2d(int dev, void *dst, const void *src, size_t bytes)
{
    MDX_TRY(set_device(dev));
    if (bytes >= (size_t(1) << 20)) {
        MDX_HIP(hipDeviceSynchronize());
        return mdx_upload(dev, dst, src, bytes);
    }
    MDX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return MDX_OK;
}

int mdx_memcpy_d2h(int dev, void *dst, const void *src, size_t bytes)
{
    MDX_TRY(set_device(dev));
    if (bytes >= (size_t(1) << 20)) {
        // everything queued on the device first (hipMemcpy's own semantics on the null stream)
        MDX_HIP(hipDeviceSynchronize());
        return device_stager(dev).download(dev, nullptr, dst, src, bytes);
    }
    MDX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return MDX_OK;
}

int mdx_memset(int dev, void *dst, int value, size_t bytes)
{
    MDX_TRY(set_device(dev));
    MDX_HIP(hipMemset(dst, value, bytes));
    return MDX_OK;
}

}  // extern "C"

// Caller memory page-locked through this library: whole pages (the runtime locks pages; a range that starts or
// ends inside a page would leave its neighbours' state to the runtime), one registration per page, looked up by
// the caller's own pointer at unregistration.  The host-buffer entry points decide "DMA where it lies" from this
// table first (host_range_registered), so a buffer that is only partly covered never reaches the DMA engine.
namespace {
struct HostRange {
    uintptr_t lo, hi;        // page-aligned [lo, hi)
    uintptr_t user;          // the pointer the caller registered
    int dev;
};
struct HostRegistry {
    std::mutex m;
    std::vector<HostRange> ranges;
};
HostRegistry &host_registry()
{
    static HostRegistry *r = new HostRegistry();
    return *r;
}
uintptr_t page_bytes()
{
    static const uintptr_t p = [] {
        const long v = sysconf(_SC_PAGESIZE);
        return uintptr_t(v > 0 ? v : 4096);
    }();
    return p;
}
}  // namespace

namespace mdx {
// 1: [ptr, ptr + bytes) lies inside one range registered through mdx_host_register; -1: it touches one without
// lying inside it (must NOT be handed to the DMA engine); 0: this library knows nothing about it
int host_range_registered(const void *ptr, size_t bytes)
{
    if (!ptr || bytes == 0)
        return 0;
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr), b = a + bytes;
    HostRegistry &r = host_registry();
    std::lock_guard<std::mutex> lk(r.m);
    int state = 0;
    for (const HostRange &h : r.ranges) {
        if (a >= h.lo && b <= h.hi)
            return 1;
        if (a < h.hi && h.lo < b)
            state = -1;
    }
    return state;
}
}  // namespace mdx

extern "C" {

int mdx_host_register(int dev, void *ptr, size_t bytes)
{
    MDX_REQUIRE(ptr && bytes, "NULL buffer");
    MDX_TRY(set_device(dev));
    const uintptr_t page = page_bytes();
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    const uintptr_t lo = a & ~(page - 1), hi = (a + bytes + page - 1) & ~(page - 1);
    HostRegistry &r = host_registry();
    std::lock_guard<std::mutex> lk(r.m);
    for (const HostRange &h : r.ranges)
        if (lo < h.hi && h.lo < hi)
            return fail(MDX_ERR_STATE,
                        "mdx_host_register: [%p, +%zu) shares pages with a range that is still registered "
                        "(%p); unregister that one first", ptr, bytes, reinterpret_cast<void *>(h.user));
    MDX_HIP(hipHostRegister(reinterpret_cast<void *>(lo), size_t(hi - lo), hipHostRegisterDefault));
    r.ranges.push_back({lo, hi, a, dev});
    return MDX_OK;
}

int mdx_host_unregister(int dev, void *ptr)
{
    MDX_REQUIRE(ptr, "NULL buffer");
    MDX_TRY(set_device(dev));
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    HostRegistry &r = host_registry();
    std::lock_guard<std::mutex> lk(r.m);
    for (size_t i = 0; i < r.ranges.size(); ++i)
        if (r.ranges[i].user == a) {
            // nothing of this library may still read the pages: the entry points that hand registered memory to
            // the DMA engine return before their copies end only on handles' streams
            MDX_HIP(hipDeviceSynchronize());
            MDX_HIP(hipHostUnregister(reinterpret_cast<void *>(r.ranges[i].lo)));
            r.ranges[i] = r.ranges.back();
            r.ranges.pop_back();
            return MDX_OK;
        }
    return fail(MDX_ERR_INVALID_VALUE, "mdx_host_unregister: %p was not registered through mdx_host_register", ptr);
}

int mdx_device_synchronize(int dev)
{
    MDX_TRY(set_device(dev));
    MDX_HIP(hipDeviceSynchronize());
    return MDX_OK;
}

int mdx_synth_random_walk(int dev, float *d_out, int64_t n_frames, int64_t n_atoms,
                          const float box_lengths[3], float sigma, uint64_t seed, int wrap)
{
    MDX_REQUIRE(d_out && box_lengths, "NULL argument");
    MDX_REQUIRE(n_frames > 0 && n_atoms > 0, "n_frames and n_atoms must be positive");
    MDX_TRY(set_device(dev));
    dim3 block(256), grid((unsigned)ceil_div(n_atoms, 256));
    hipLaunchKernelGGL(synth_walk_kernel<float>, grid, block, 0, 0, d_out, n_frames, n_atoms,
                       box_lengths[0], box_lengths[1], box_lengths[2], sigma, uint32_t(seed),
                       uint32_t(seed >> 32), wrap);
    MDX_HIP(hipGetLastError());
    MDX_HIP(hipDeviceSynchronize());
    return MDX_OK;
}

int mdx_synth_random_walk_f64(int dev, double *d_out, int64_t n_frames, int64_t n_atoms,
                              const float box_lengths[3], float sigma, uint64_t seed)
{
    MDX_REQUIRE(d_out && box_lengths, "NULL argument");
    MDX_REQUIRE(n_frames > 0 && n_atoms > 0, "n_frames and n_atoms must be positive");
    MDX_TRY(set_device(dev));
    dim3 block(256), grid((unsigned)ceil_div(n_atoms, 256));
    hipLaunchKernelGGL(synth_walk_kernel<double>, grid, block, 0, 0, d_out, n_frames, n_atoms,
                       box_lengths[0], box_lengths[1], box_lengths[2], sigma, uint32_t(seed),
                       uint32_t(seed >> 32), 0);
    MDX_HIP(hipGetLastError());
    MDX_HIP(hipDeviceSynchronize());
    return MDX_OK;
}

}  // extern "C"
