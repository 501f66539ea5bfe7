// copy_pool.h -- the copy threads of the host-buffer entry points (capi_pipeline.hip): a parallel memcpy of one region at a time between the
// caller's pageable buffers and the pinned staging ring.  The calling thread takes the first piece itself; the pool can be stopped and
// restarted with another thread count (option host_threads).  Plain C++ (no HIP): tests/host/test_copy_pool.cpp runs it under ThreadSanitizer.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

struct CopyPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  const char* src = nullptr; char* dst = nullptr; size_t len = 0, piece = 0;
  unsigned long long gen = 0; int pending = 0; bool quit = false;
  static constexpr size_t kSerialBelow = (size_t)256 << 10;      // smaller regions are copied by the caller alone

  ~CopyPool() { stop(); }
  int threads() const { return (int)th.size() + 1; }
  // seen: the generation at the time the thread was started -- a restarted pool must not replay the last job (it did, once: the workers of
  // the new pool copied into buffers that had been freed and decremented `pending` of the next job)
  void worker(int id, unsigned long long seen) {
    for (;;) {
      const char* s; char* d; size_t n, pc;
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return quit || gen != seen; }); if (quit) return; seen = gen; s = src; d = dst; n = len; pc = piece; }
      const size_t off = (size_t)(id + 1) * pc;
      if (off < n) memcpy(d + off, s + off, std::min(pc, n - off));
      { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cv_done.notify_one(); }
    }
  }
  void start(int nworkers) {
    unsigned long long g0;
    { std::lock_guard<std::mutex> lk(mu); g0 = gen; pending = 0; }
    for (int i = 0; i < nworkers; ++i) th.emplace_back([this, i, g0] { worker(i, g0); });
  }
  void stop() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    for (auto& t : th) t.join();
    th.clear();
    { std::lock_guard<std::mutex> lk(mu); quit = false; }
  }
  void copy(void* d, const void* s, size_t n) {
    const int T = threads();
    if (T == 1 || n < kSerialBelow) { memcpy(d, s, n); return; }
    size_t pc = (n + T - 1) / T; pc = (pc + 4095) & ~(size_t)4095;
    { std::lock_guard<std::mutex> lk(mu); src = (const char*)s; dst = (char*)d; len = n; piece = pc; pending = (int)th.size(); ++gen; }
    cv.notify_all();
    memcpy(d, s, std::min(pc, n));
    std::unique_lock<std::mutex> lk(mu); cv_done.wait(lk, [&] { return pending == 0; });
  }
};
