/*
 * mm_hostcopy.hip -- a device buffer into the caller's HOST array at the speed of the link, whatever the array is.
 *
 * The reference's `run` returns host arrays (ChainRunner::run -> Array3, core.rs:176-186; HMC::run's tensor is read back
 * by the caller, hmc.rs:137-158), and a caller's `Vec` / numpy array is PAGEABLE memory: one hipMemcpyAsync into it
 * moved config 3's 315 MB sample at 9.9 GB/s (31.8 ms next to a 0.18 ms kernel; profiles/r4z_pcie_inclusive.json), into
 * pinned memory at 52.7 GB/s (5.97 ms).  Here the library stages: the DMA engine fills a ring of library-owned PINNED
 * bounce buffers chunk by chunk while a few host threads copy the finished chunks into the destination (each thread its
 * own slice of every chunk, so the destination's page faults and the DRAM traffic are spread over cores).  A destination
 * that IS pinned (hipHostMalloc / hipHostRegister / torch's pin_memory) gets the one direct copy as before.
 *
 * One stager per device (its own buffers and lock: shards of a device group copy side by side over their own links; they
 * share the copy-thread budget, and waiting threads yield the CPU after a short spin).  The 32 MB pinned ring of a device
 * that has staged once stays with the process (freeing pinned memory from a static destructor races the runtime's own
 * teardown); hosts with a single usable CPU get the plain copy.
 */
#include "mm_hostcopy.h"

#include <sched.h>
#include <unistd.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr size_t kChunk = 8u << 20; /* 0.15 ms of DMA per chunk: the ring's latency; 4 chunks in flight */
constexpr int kRing = 4;
constexpr size_t kDirectBelow = 4u << 20; /* small results: the plain copy (its staging inside the runtime is as good) */

struct Stager {
    std::mutex mu;
    void *pin[kRing] = {};
    hipEvent_t ev[kRing] = {};
    bool ready = false;
};

Stager &stager(int device)
{
    static Stager s[64];
    return s[device & 63];
}

int usable_cpus()
{
    cpu_set_t set;
    int n = 0;
    if (sched_getaffinity(0, sizeof set, &set) == 0)
        n = CPU_COUNT(&set);
    if (n <= 0)
        n = (int)sysconf(_SC_NPROCESSORS_ONLN);
    return n < 1 ? 1 : n;
}

/* staged copies in flight in this process (the shards of a device group copy side by side): they SHARE the copy-thread
 * budget -- eight shards no longer start 8 x (8 + 1) spinning threads on a 16-CPU affinity mask */
std::atomic<int> g_active{0};

int copy_threads(int cpus, int active)
{
    /* DRAM, not cores, is the limit from a handful of threads on; leave half the usable CPUs to the caller */
    int n = cpus / 2;
    n = n < 1 ? 1 : (n > 8 ? 8 : n);
    n = n / (active < 1 ? 1 : active);
    return n < 1 ? 1 : n;
}

bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError(); /* an ordinary malloc'ed pointer is "invalid value" to the runtime: not an error here */
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

/* waiting: `pause` for the first few hundred polls (a chunk is 0.15 ms of DMA), then give the CPU away between polls */
struct Backoff {
    unsigned int spins = 0;
    void wait()
    {
        if (++spins < 256) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        } else {
            sched_yield();
        }
    }
    void reset() { spins = 0; }
};

struct ActiveGuard {
    int n;
    ActiveGuard() : n(g_active.fetch_add(1, std::memory_order_relaxed) + 1) {}
    ~ActiveGuard() { g_active.fetch_sub(1, std::memory_order_relaxed); }
};

} // namespace

hipError_t mm_copy_to_host(void *dst, const void *d_src, size_t bytes, int device, hipStream_t stream)
{
    if (bytes == 0)
        return hipSuccess;
    hipError_t e;
    const int cpus = usable_cpus();
    /* with one usable CPU a copy thread and the DMA driver would only take turns: the plain copy */
    if (bytes < kDirectBelow || cpus < 2 || is_pinned_host(dst)) {
        if ((e = hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, stream)) != hipSuccess)
            return e;
        return hipStreamSynchronize(stream);
    }
    ActiveGuard active;
    Stager &st = stager(device);
    std::lock_guard<std::mutex> lock(st.mu);
    if (!st.ready) {
        for (int i = 0; i < kRing; ++i) {
            if (!st.pin[i] && (e = hipHostMalloc(&st.pin[i], kChunk, hipHostMallocDefault)) != hipSuccess)
                return e;
            if (!st.ev[i] && (e = hipEventCreateWithFlags(&st.ev[i], hipEventDisableTiming)) != hipSuccess)
                return e;
        }
        st.ready = true;
    }
    const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
    const int T = copy_threads(cpus, active.n);
    /* published: chunks whose DMA has finished (the workers may read them); consumed[i % kRing]: workers done with chunk i */
    std::atomic<size_t> published{0};
    std::atomic<unsigned int> consumed[kRing];
    std::atomic<bool> failed{false};
    for (auto &c : consumed)
        c.store(0, std::memory_order_relaxed);
    auto worker = [&](int t) {
        Backoff bo;
        for (size_t i = 0; i < n_chunks; ++i) {
            bo.reset();
            while (published.load(std::memory_order_acquire) <= i) {
                if (failed.load(std::memory_order_relaxed))
                    return;
                bo.wait();
            }
            const size_t off = i * kChunk, len = (off + kChunk <= bytes) ? kChunk : bytes - off;
            /* slices on 4 KB boundaries: every destination page is faulted in and written by one thread */
            const size_t per = ((len + (size_t)T - 1) / (size_t)T + 4095) & ~(size_t)4095;
            const size_t a = (size_t)t * per, b = a + per < len ? a + per : len;
            if (a < len)
                std::memcpy((char *)dst + off + a, (const char *)st.pin[i % kRing] + a, b - a);
            consumed[i % kRing].fetch_add(1, std::memory_order_release);
        }
    };
    /* T copy threads beside this one, which drives the DMA ring.  A host that cannot start threads (resource limits) gets
     * the plain copy: nothing may throw across the C ABI */
    std::vector<std::thread> threads;
    try {
        threads.reserve((size_t)T);
        for (int t = 0; t < T; ++t)
            threads.emplace_back(worker, t);
    } catch (...) {
        failed.store(true, std::memory_order_relaxed);
        for (std::thread &th : threads)
            th.join();
        if ((e = hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, stream)) != hipSuccess)
            return e;
        return hipStreamSynchronize(stream);
    }
    size_t issued = 0, pub = 0;
    e = hipSuccess;
    Backoff bo;
    while (pub < n_chunks) {
        bool progressed = false;
        if (issued < n_chunks) {
            /* slot free: never used, or every worker is done with the chunk that used it */
            const bool slot_free = issued < (size_t)kRing || consumed[issued % kRing].load(std::memory_order_acquire) >= (unsigned int)T;
            if (slot_free) {
                if (issued >= (size_t)kRing)
                    consumed[issued % kRing].store(0, std::memory_order_relaxed);
                const size_t off = issued * kChunk, len = (off + kChunk <= bytes) ? kChunk : bytes - off;
                if ((e = hipMemcpyAsync(st.pin[issued % kRing], (const char *)d_src + off, len, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
                    (e = hipEventRecord(st.ev[issued % kRing], stream)) != hipSuccess)
                    break;
                ++issued;
                progressed = true;
            }
        }
        if (pub < issued) {
            const hipError_t q = hipEventQuery(st.ev[pub % kRing]);
            if (q == hipSuccess) {
                published.store(++pub, std::memory_order_release);
                progressed = true;
            } else if (q != hipErrorNotReady) {
                e = q;
                break;
            }
        }
        if (progressed)
            bo.reset();
        else
            bo.wait();
    }
    if (e != hipSuccess) {
        failed.store(true, std::memory_order_relaxed);
        (void)hipGetLastError();
    }
    for (std::thread &th : threads)
        th.join();
    if (e != hipSuccess)
        (void)hipStreamSynchronize(stream); /* nothing of ours may still be in flight when the buffers are reused */
    return e;
}
