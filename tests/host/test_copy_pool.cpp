// test_copy_pool.cpp -- CPU (no GPU, no library): the copy threads of the host-buffer pipeline (fhe-si_amd/csrc/copy_pool.h) under
// ThreadSanitizer: regions of random sizes copied by pools of 1 .. 8 threads, the pool stopped and restarted with another thread count
// between batches (a restarted pool once replayed its last job), every copy compared with the source.
//   test_copy_pool [copies per pool = 200] [seed = 1]
#include <cstdio>
#include <cstdlib>

#include "../../fhe-si_amd/csrc/copy_pool.h"

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1;
  std::vector<char> a((size_t)6 << 20), b(a.size());
  for (size_t i = 0; i < a.size(); ++i) a[i] = (char)(i * 7 + 3 + seed);
  CopyPool p;
  long copies = 0;
  const int counts[] = {3, 7, 1, 2, 8, 4, 8, 1, 5};
  for (int T : counts) {
    p.stop();
    p.start(T - 1);
    if (p.threads() != T) { printf("pool reports %d threads, wanted %d\n", p.threads(), T); return 1; }
    for (int it = 0; it < reps; ++it) {
      // sizes on both sides of the serial threshold, unaligned ends
      const size_t n = (size_t)rand_r(&seed) % (it % 4 == 0 ? (size_t)512 << 10 : a.size()) + 1;
      const size_t off = (size_t)rand_r(&seed) % 64;
      if (off + n > a.size()) continue;
      std::fill(b.begin() + off, b.begin() + off + n, 0);
      p.copy(b.data() + off, a.data() + off, n);
      if (memcmp(a.data() + off, b.data() + off, n)) { printf("MISMATCH threads=%d n=%zu\n", T, n); return 1; }
      ++copies;
    }
  }
  p.stop();
  printf("copy pool: %ld copies over %zu pool configurations, 0 mismatches\nTest SUCCEEDED\n", copies, sizeof counts / sizeof counts[0]);
  return 0;
}
