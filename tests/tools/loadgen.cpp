// loadgen.cpp — TEST TOOL (not part of the product library): N host threads issuing single-query, host-I/O
// mi355_search calls against one open handle, the way the reference's callers do (tokio workers of a multi-thread
// runtime, /root/reference/python/src/runtime.rs:31-37; `BaseTable: Send + Sync`, rust/lancedb/src/table.rs:549).
// bench.py's first version of this leg drove the handle from Python threads: 64 of them spend most of their time
// waiting for the interpreter lock, so the figure measured Python, not the library.  Here the callers are plain
// std::threads; the search entry point arrives as a function pointer (the library is already loaded by the caller).
//
//   g++ -O2 -std=c++17 -shared -fPIC -pthread loadgen.cpp -o libloadgen.so        (bench_legs.build_loadgen)
#include <atomic>
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>

typedef int32_t (*search_fn)(void* index, const float* queries, uint32_t n_queries, const void* params, uint64_t* out_rowids,
                             float* out_dist, uint32_t* out_counts);

extern "C" int32_t loadgen_run(void* fn_ptr, void* index, const float* queries, uint32_t pool, uint32_t dim, const void* params,
                               uint32_t k, uint32_t n_threads, uint32_t per_thread, double* out_seconds, float* out_lat_us /*[n_threads * per_thread]*/,
                               uint64_t* out_rowid_sum) {
  search_fn search = (search_fn)fn_ptr;
  std::atomic<int32_t> status{0};
  std::atomic<uint32_t> ready{0};
  std::atomic<bool> go{false};
  std::atomic<uint64_t> sum{0};
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < n_threads; ++t)
    th.emplace_back([&, t] {
      std::vector<uint64_t> ids(k);
      std::vector<float> dist(k);
      uint32_t cnt = 0;
      uint64_t local = 0;
      ready.fetch_add(1);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      for (uint32_t i = 0; i < per_thread; ++i) {
        const float* q = queries + (size_t)((t * per_thread + i) % pool) * dim;
        auto t0 = std::chrono::steady_clock::now();
        int32_t s = search(index, q, 1, params, ids.data(), dist.data(), &cnt);
        auto t1 = std::chrono::steady_clock::now();
        if (s != 0) status.store(s);
        if (out_lat_us) out_lat_us[(size_t)t * per_thread + i] = std::chrono::duration<float, std::micro>(t1 - t0).count();
        for (uint32_t j = 0; j < cnt && j < k; ++j) local += ids[j];
      }
      sum.fetch_add(local);
    });
  while (ready.load() < n_threads) std::this_thread::yield();
  auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto& x : th) x.join();
  auto t1 = std::chrono::steady_clock::now();
  *out_seconds = std::chrono::duration<double>(t1 - t0).count();
  if (out_rowid_sum) *out_rowid_sum = sum.load();
  return status.load();
}
