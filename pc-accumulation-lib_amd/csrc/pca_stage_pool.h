// pca_stage_pool.h -- the host threads behind pca_host_stage_h2d / pca_kitti_integrate (plain C++, no device code, so that
// tests/native/stage_pool_tsan.cpp can build it with g++ -fsanitize=thread).
//
// A job is a list of slices (dst, src, n <= 128 KB) copied with memcpy by whoever is awake: the caller and up to T helper
// threads.  Slices are CLAIMED one at a time -- a helper that is descheduled (more runnable threads than cores: eight ranks
// with their pools, a cgroup-limited container) claims nothing, and whoever is running finishes the job.
//
// Every claim is tied to ITS job: generation, slice count and next index live in ONE 64-bit word,
//     [63:40] generation   [39:20] slices of the job   [19:0] next unclaimed slice
// and a claim is a compare-and-swap on that word.  A helper that loaded the word at the tail of job N and was descheduled
// fails its CAS once job N + 1 has been published (the generation differs) and starts over; a successful CAS on (g, total, i)
// with i < total proves that job g is still open -- its caller waits for `done` to reach `total`, slice i included -- so the
// slice table read afterwards is job g's.  (Round 4's form kept `next` and `total` apart: a helper holding an index >= the
// old total could find it < the NEW total and copy a slice of the next job a second time, or read a freed slice table.)
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace pca_stage {

struct Slice { char *dst; const char *src; size_t n; };

// One slice, pageable source -> pinned staging block.  The destination is written ONCE and read next by the DMA engine, never
// by this CPU: streaming (non-temporal) stores skip the read-for-ownership of every destination line that a plain memcpy of
// 128 KB pays (glibc switches to them only for copies beyond the cache size), i.e. a third less memory traffic per byte staged.
// PCA_STAGING_NT=0: plain memcpy (A/B).  The caller fences (stream_fence) before it publishes the slice as done.
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void copy_nt_avx2(char *dst, const char *src, size_t n)
{
    const size_t head = (32 - (reinterpret_cast<uintptr_t>(dst) & 31)) & 31;
    if (head) { const size_t h = head < n ? head : n; memcpy(dst, src, h); dst += h; src += h; n -= h; }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 96), d);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
}
#endif
inline bool use_nt()
{
#if defined(__x86_64__)
    static const bool on = [] { const char *e = getenv("PCA_STAGING_NT"); return (!e || atoi(e) != 0) && __builtin_cpu_supports("avx2"); }();
    return on;
#else
    return false;
#endif
}
inline void copy_slice(char *dst, const char *src, size_t n)
{
#if defined(__x86_64__)
    if (n >= 4096 && use_nt()) { copy_nt_avx2(dst, src, n); return; }
#endif
    memcpy(dst, src, n);
}
inline void stream_fence()                                  // streaming stores are weakly ordered: drained before "done" is said
{
#if defined(__x86_64__)
    _mm_sfence();
#endif
}

struct Pool {
    static constexpr int IDX_BITS = 20;
    static constexpr uint64_t IDX_MASK = (1ull << IDX_BITS) - 1;
    static constexpr int MAX_SLICES = (int)IDX_MASK;          // a bigger job is copied by the caller alone (never seen: 128 GB)

    int T = 0;                                                // helper threads (the caller takes part too)
    long spin_us = 2000;
    std::vector<std::thread> th;
    std::atomic<uint64_t> claim{0};
    std::atomic<const Slice *> slices{nullptr};
    std::atomic<int> done{0}, sleepers{0};
    std::atomic<bool> stop{false};
    std::mutex m;
    std::condition_variable cv;
    uint64_t gen = 0;                                         // caller side only (callers take turns: see pca_stage_copy)

    static uint64_t pack(uint64_t g, uint64_t total, uint64_t idx) { return (g << (2 * IDX_BITS)) | (total << IDX_BITS) | idx; }
    static uint64_t gen_of(uint64_t c) { return c >> (2 * IDX_BITS); }

    // Copies slices of the job that is open NOW until it has none left.  Returns the generation it last looked at.
    uint64_t drain()
    {
        uint64_t c = claim.load(std::memory_order_acquire);
        for (;;) {
            const uint64_t total = (c >> IDX_BITS) & IDX_MASK, i = c & IDX_MASK;
            if (i >= total) return gen_of(c);
            if (!claim.compare_exchange_weak(c, c + 1, std::memory_order_acq_rel, std::memory_order_acquire)) continue;
            const Slice *s = slices.load(std::memory_order_relaxed);      // published before the claim word (release / acquire)
            copy_slice(s[i].dst, s[i].src, s[i].n);
            stream_fence();
            done.fetch_add(1, std::memory_order_release);
            c = claim.load(std::memory_order_acquire);
        }
    }
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            int polls = 0;
            while (gen_of(claim.load(std::memory_order_acquire)) == seen && !stop.load(std::memory_order_relaxed)) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                if ((++polls & 255) == 0 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) {
                    std::unique_lock<std::mutex> lk(m);
                    sleepers.fetch_add(1);
                    cv.wait(lk, [&] { return gen_of(claim.load(std::memory_order_acquire)) != seen || stop.load(); });
                    sleepers.fetch_sub(1);
                    break;
                }
            }
            if (stop.load()) return;
            seen = drain();
        }
    }
    void start(int threads)
    {
        T = threads;
        if (const char *e = getenv("PCA_STAGING_SPIN_US")) spin_us = atol(e);
        for (int w = 0; w < T; ++w) th.emplace_back([this] { worker(); });
    }
    // One job at a time (the caller serialises).  Returns when every slice has been copied.
    void run(const Slice *s, int n)
    {
        if (n <= 0) return;
        if (T == 0 || n > MAX_SLICES) { for (int i = 0; i < n; ++i) copy_slice(s[i].dst, s[i].src, s[i].n); stream_fence(); return; }
        gen = (gen + 1) & ((1ull << (64 - 2 * IDX_BITS)) - 1);
        if (gen == 0) gen = 1;                                // (0 = "no job yet" for a fresh helper)
        slices.store(s, std::memory_order_relaxed);
        done.store(0, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> lk(m); claim.store(pack(gen, (uint64_t)n, 0), std::memory_order_release); }
        if (sleepers.load() > 0) cv.notify_all();
        drain();
        // every slice is claimed by now; the last few may still be in a helper's memcpy (<= 128 KB each): a short, bounded
        // spin, then yield the core to whoever holds them
        for (int spins = 0; done.load(std::memory_order_acquire) < n; ++spins) {
#if defined(__x86_64__)
            if (spins < 4096) { __builtin_ia32_pause(); continue; }
#endif
            std::this_thread::yield();
        }
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(m); stop.store(true); }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
};

}  // namespace pca_stage
