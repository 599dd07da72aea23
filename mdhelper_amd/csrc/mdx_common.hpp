// mdx_common.hpp — shared host-side plumbing of libmdx.so (errors, device guard, timers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mdx.h"

namespace mdx {

// thread-local message behind mdx_last_error()
char *error_buffer();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define MDX_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            int _c = (_e == hipErrorOutOfMemory) ? MDX_ERR_OUT_OF_MEMORY                \
                     : (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice)          \
                         ? MDX_ERR_NO_DEVICE                                            \
                         : MDX_ERR_HIP;                                                 \
            return ::mdx::fail(_c, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                     \
        }                                                                               \
    } while (0)

#define MDX_TRY(expr)             \
    do {                          \
        int _rc = (expr);         \
        if (_rc != MDX_OK)        \
            return _rc;           \
    } while (0)

#define MDX_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond))                                            \
            return ::mdx::fail(MDX_ERR_INVALID_VALUE, __VA_ARGS__); \
    } while (0)

int set_device(int dev);

// grow-only device buffer owned by a handle
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);   // reallocates (contents lost) when need > bytes; takes a cached block when one fits
    void release();            // hipFree (waits for the device: safe while kernels may still read the block)
    // Hands the block to the per-device cache instead of freeing it — ONLY after every stream that may have
    // touched it has been synchronised (the destroy paths).  An analysis object per call, the reference's
    // usage, otherwise pays ~10 ms of hipMalloc / hipFree per object — and, for the multi-GB blocks of the
    // MSD engine, now and then seconds inside hipMalloc; the cache is bounded (a quarter of the device's
    // memory, MDX_CACHE_GB), is given back when an allocation fails or on mdx_trim_cache, and lives as long
    // as the process.
    void recycle();
    template <typename T> T *as() const { return static_cast<T *>(ptr); }
};

// bytes the cache of `dev` holds: memory hipMemGetInfo reports as used but an allocation of the library can have
size_t cached_device_bytes(int dev);

// Non-blocking streams are handed out from a per-device pool and go back to it when a handle is destroyed
// (after they have been synchronised): creating and destroying a stream costs about a millisecond each way,
// and an analysis object per call — the reference's usage — holds three.
int stream_acquire(hipStream_t *out);      // on the current device
void stream_release(hipStream_t stream);   // idle streams only

// HIP-event timer on one stream: accumulates the device time of bracketed regions
struct StreamTimer {
    bool enabled = false;
    hipStream_t stream = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> limbo;   // dropped before they fired (reset on a busy stream)
    std::vector<hipEvent_t> pool;
    double total_ms = 0.0;
    int64_t launches = 0;
    hipEvent_t begin();
    void end(hipEvent_t start);
    void collect();            // waits for the end event of every bracket still pending, then adds it up
    void reset();              // drops what is pending (recycled once fired) and zeroes the totals
    void destroy();
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Persistent host threads for the copies that feed pinned staging memory (memcpy out of caller
// buffers, pread out of the page cache).  parallel_for(n, fn) runs fn(0) ... fn(n-1), the caller
// taking part; tasks are claimed from one counter, so uneven tasks balance.
struct HostWorkers {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)> *job = nullptr;
    std::atomic<int> next{0};
    int n_tasks = 0, active = 0;
    uint64_t generation = 0;
    bool stopping = false;
    int size() const { return int(threads.size()) + 1; }
    void start(int n_threads);          // idempotent; n_threads counts the caller
    void parallel_for(int n, const std::function<void(int)> &fn);
    void stop();
    ~HostWorkers() { stop(); }
private:
    void loop();
};

// Pinned staging ring between host memory (caller buffers, the page cache) and HBM.  It owns its
// stream, and every event it owns is only ever recorded on that stream: no event can outlive the
// stream it was last recorded on.  (HIP keeps a raw pointer to that stream inside the event and
// reads its capture state in hipEventSynchronize / hipStreamWaitEvent; after the stream is
// destroyed that is a read of freed memory, which now and then looks like "capture active" and
// comes back as hipErrorCapturedEvent: "operation not permitted on an event last recorded in a
// capturing stream".  That was the intermittent failure of round 2, commit 5575fea.)
// Copies stay in flight across calls; a buffer is waited for only when it is needed again.
struct HostStager {
    static constexpr int NBUF = 3;
    int dev = -1;
    hipStream_t io = nullptr;
    void *pinned[NBUF] = {nullptr, nullptr, nullptr};
    size_t pinned_bytes = 0;
    hipEvent_t ev_sent[NBUF] = {nullptr, nullptr, nullptr};   // behind the last copy out of pinned[b]
    bool in_flight[NBUF] = {false, false, false};
    hipEvent_t ev_batch = nullptr;      // behind everything queued on io so far (consumers wait on it)
    int64_t turn = 0;
    HostWorkers workers;
    std::mutex lock;                    // one user at a time: the ring of a device is shared by its handles

    int ensure(int device, size_t chunk_bytes);
    int after(hipStream_t producer);                 // io waits for what `producer` holds so far
    int acquire(int *b, void **host);                // next pinned buffer, its last copy finished
    int send(int b, void *d_dst, size_t bytes);      // pinned[b] -> HBM on io
    int finish(hipStream_t consumer);                // consumer waits for everything queued on io
    // d_dst[0, bytes) <- src: directly when src is pinned / registered memory, else in chunks
    // through the ring with the workers copying; the data is ordered before later work on `consumer`
    int upload(int device, hipStream_t consumer, void *d_dst, const void *src, size_t bytes);
    // the same for n_rows rows of row_bytes that lie src_stride apart on the host and end up contiguous in d_dst
    // (a range of particles out of frames [T][N][3]: what one group of an MSD analysis needs of every frame)
    int upload_rows(int device, hipStream_t consumer, void *d_dst, const void *src, size_t row_bytes,
                    size_t src_stride, size_t n_rows);
    // the pageable case of upload_rows for rows of >= 4 KB, >= 2 pages apart, in anonymous memory: slices locked,
    // copied by 2-D DMA and unlocked by the copy threads, side by side
    int copy_rows_locked(int device, hipStream_t consumer, void *d_dst, const void *src, size_t row_bytes,
                         size_t src_stride, size_t n_rows);
    // dst[0, bytes) <- d_src (HBM -> pageable host memory) through the pinned buffers, the copy of chunk k + 1 in
    // flight while chunk k leaves its buffer; returns when dst is complete.  `producer`: the stream whose queued
    // work wrote d_src (nullptr: the caller has synchronised)
    int download(int device, hipStream_t producer, void *dst, const void *d_src, size_t bytes);
    int drain();                                     // host waits for every pinned buffer
    void destroy();
};

// The staging ring of a device, shared by every handle on it and kept for the life of the process: pinned
// allocations and stream creation are milliseconds, an analysis object per call (the reference's usage:
// RadialDistributionFunction(...).run()) must not pay them every time.  Users hold `lock` while they queue.
HostStager &device_stager(int dev);
// caller memory page-locked through mdx_host_register: 1 covered whole, -1 touched but not covered, 0 unknown
int host_range_registered(const void *ptr, size_t bytes);

// Double-buffered staging between a producer on a copy stream and the kernels on a handle's
// compute stream: the fill of slab k+1 overlaps the kernels of slab k.  A buffer is refilled
// only after the kernels that read it have finished (event), and run() returns once the last
// fill has completed — no host pointer is retained — while the kernels of the last slabs may
// still be in flight on the compute stream.
struct StagePipeline {
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_filled[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int ensure();
    void destroy();
    // fill(b, f0, nf): queue on copy_stream what brings items [f0, f0+nf) into staging set b;
    // consume(b, f0, nf): queue on `compute` the kernels that read staging set b.
    // ramp > 1: the first slabs are slab / 4 and slab / 2 items, rounded down to multiples of `ramp` — the first
    // fill is the one nothing hides, and large slabs (which the consumer wants: fewer, longer launches) make it long
    template <typename Fill, typename Consume>
    int run(hipStream_t compute, int64_t n_items, int64_t slab, Fill fill, Consume consume, int64_t ramp = 0)
    {
        MDX_TRY(ensure());
        slab = slab < 1 ? 1 : slab;
        int64_t step = slab;
        // an error exit waits for the copy stream too: a DMA out of caller memory queued by an earlier fill
        // must not be in flight after the call has returned
        auto body = [&]() -> int {
            for (int64_t f0 = 0, k = 0; f0 < n_items; f0 += step, ++k) {
                step = slab;
                if (ramp > 1 && k < 2) {
                    const int64_t part = (slab >> (2 - k)) / ramp * ramp;
                    if (part >= ramp)
                        step = part;
                }
                const int64_t nf = n_items - f0 < step ? n_items - f0 : step;
                const int b = int(k & 1);
                if (busy[b]) {
                    MDX_HIP(hipEventSynchronize(ev_consumed[b]));
                    busy[b] = false;
                }
                MDX_TRY(fill(b, f0, nf));
                MDX_HIP(hipEventRecord(ev_filled[b], copy_stream));
                MDX_HIP(hipStreamWaitEvent(compute, ev_filled[b], 0));
                MDX_TRY(consume(b, f0, nf));
                MDX_HIP(hipEventRecord(ev_consumed[b], compute));
                busy[b] = true;
            }
            return MDX_OK;
        };
        const int rc = body();
        if (rc != MDX_OK) {
            (void)hipStreamSynchronize(copy_stream);
            return rc;
        }
        MDX_HIP(hipStreamSynchronize(copy_stream));
        return MDX_OK;
    }
};

}  // namespace mdx
