// Host-side staging copy for the ingest path (no GPU work): a caller's pageable frame (24.9 MB at 3840x2160) is copied into the
// pinned staging buffer by a small persistent pool of threads before its H2D (apse_uav_amd/engines/track_predictor.py,
// FrameUploader).  The reference loop hands the tracker a fresh numpy array per frame (cv2.VideoCapture.read,
// dcnn/scripts/tests/visualize_uav.py:188-191); round 3 did this copy with a Python thread pool whose per-task dispatch cost
// (16 tasks per frame) was a third of the 0.8 ms it took.
#include "../../include/apse_hip.h"

#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// Streaming copy: the destination is a pinned staging buffer that only the DMA engine reads next, so its lines should not be
// pulled into the cache first (a plain store of a fresh line reads it: 3 bytes of memory traffic per byte copied instead of 2).
// glibc's memcpy switches to non-temporal stores only above a per-thread size threshold that a 3 MB share of a frame does not reach.
__attribute__((target("avx2"))) static void copy_stream_avx2(char* dst, const char* src, size_t n) {
    while (n && (reinterpret_cast<uintptr_t>(dst) & 31)) { *dst++ = *src++; --n; }
    size_t blocks = n / 128;
    for (; blocks; --blocks, dst += 128, src += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst), a);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + 96), d);
    }
    _mm_sfence();
    memcpy(dst, src, n % 128);
}
static void copy_part(char* dst, const char* src, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 4096) copy_stream_avx2(dst, src, n); else memcpy(dst, src, n);
}

struct CopyPool {
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::vector<std::thread> workers;
    unsigned long long gen = 0;
    int pending = 0, nparts = 0;
    bool quit = false;
    char* dst = nullptr; const char* src = nullptr; size_t bytes = 0;

    void part(int k) const {
        const size_t chunk = ((bytes / nparts) + 4095) & ~(size_t)4095;
        const size_t lo = chunk * (size_t)k;
        if (lo >= bytes) return;
        const size_t n = lo + chunk < bytes ? chunk : bytes - lo;
        copy_part(dst + lo, src + lo, n);
    }
    void worker(int k) {
        unsigned long long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
            const bool mine = k < nparts;
            lk.unlock();
            if (mine) part(k);
            lk.lock();
            if (mine && --pending == 0) cv_done.notify_one();
        }
    }
    void ensure(int n) {
        while ((int)workers.size() < n - 1) {
            const int k = (int)workers.size() + 1;           // part 0 is the caller's
            workers.emplace_back([this, k] { worker(k); });
        }
    }
    void run(void* d, const void* s, size_t b, int n) {
        ensure(n);
        {
            std::lock_guard<std::mutex> lk(mu);
            dst = (char*)d; src = (const char*)s; bytes = b; nparts = n; pending = n - 1;
            ++gen;
        }
        cv_go.notify_all();
        part(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_go.notify_all();
        for (auto& t : workers) t.join();
    }
};

CopyPool& pool() { static CopyPool p; return p; }
std::mutex g_call;            // one copy at a time (callers are frame loops; serialising keeps the pool's state single-owner)

}  // namespace

extern "C" int apse_host_copy(void* dst, const void* src, size_t bytes, int threads) {
    if ((!dst || !src) && bytes) return APSE_E_INVALID;
    if (threads < 1) threads = 1;
    if (threads > 32) threads = 32;
    if (bytes < ((size_t)1 << 20) || threads == 1) { memcpy(dst, src, bytes); return APSE_OK; }
    std::lock_guard<std::mutex> lk(g_call);
    pool().run(dst, src, bytes, threads);
    return APSE_OK;
}
