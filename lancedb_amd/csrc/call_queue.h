// call_queue.h — the coalescing queue of concurrent host callers of one handle (SURVEY.md §8b threading: the reference's
// callers are tokio workers, /root/reference/python/src/runtime.rs:31-37; `BaseTable: Send + Sync`, rust/lancedb/src/table.rs:549).
// A caller that finds the handle busy parks its request; the caller that owns the device takes every parked request with
// equal parameters into ONE device batch when it starts, so N concurrent single-query calls cost about one launch sequence.
//
// Header-only and free of HIP: it is compiled into libmi355_ann.so (ann_index_search.hip) and, alone, into
// tests/tools/queue_stress.cpp, which runs it under ThreadSanitizer on the CPU (tests/test_build_host.py).
//
// Wake-ups are targeted: parked callers sleep on one of two futex words (Linux; the product is ROCm-only).  Everything a
// batch collects arrived before that collection and shares a word, so ONE wake call releases the callers a batch served,
// none of them needs the queue lock to leave, and the callers that parked meanwhile (the other word) sleep on.  A waiter
// lives on its caller's stack: whoever decides its fate stores its state LAST and never touches it again.
#pragma once
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

enum : uint32_t { QS_PARKED = 0, QS_SERVED = 1, QS_LEAD = 2 };

// what the queue needs of a parked request (the handle's request type derives from it)
struct QueueWaiter {
  uint32_t nq = 0;      // queries it carries (a batch is capped)
  int32_t status = 0;   // QS_SERVED: the status / message of the batch that carried it
  char error[256] = {0};  // fixed: delivering a failure must not allocate (it may be "out of host memory")
  // the owner of the batch already wrote THIS request's own outcome into status / error (its own deadline passed while
  // the others' did not): leave() delivers it instead of the batch's.  Only the owner writes it, before leave().
  bool decided = false;
  // QS_PARKED, QS_SERVED = another caller's batch carried it (status / error are final), QS_LEAD = handed the device:
  // this caller runs the next batch.  Written last by the thread that decides, read without the queue lock by the owner.
  std::atomic<uint32_t> state{QS_PARKED};
  uint32_t cohort = 0;  // which wake word it sleeps on (parity of the collection it arrived before)
};

static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "futex word");
static inline void cq_futex_wait(std::atomic<uint32_t>* a, uint32_t expected) {
  (void)syscall(SYS_futex, (uint32_t*)a, FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);  // EAGAIN / EINTR: the caller re-checks
}
static inline void cq_futex_wake_all(std::atomic<uint32_t>* a) {
  (void)syscall(SYS_futex, (uint32_t*)a, FUTEX_WAKE_PRIVATE, 0x7fffffff, nullptr, nullptr, 0);
}

// The deadline rule of a coalesced batch (QueryExecutionOptions.timeout is per query, rust/lancedb/src/query.rs:641): every call is
// held against ITS OWN entry time.  -> the calls that still have budget (`live`, indices into t0) and the smallest budget left
// among them (what the device is armed with); `worst_wait_ms` = the longest wait among the expired ones (for the message).
// Free of HIP like the queue: tests/tools/queue_stress.cpp checks it on the CPU.
struct DeadlineSplit {
  std::vector<size_t> live;
  uint32_t timeout_left = 0;
  long long worst_wait_ms = 0;
};
static inline DeadlineSplit split_by_deadline(const std::vector<std::chrono::steady_clock::time_point>& t0,
                                              std::chrono::steady_clock::time_point now, uint32_t timeout_ms) {
  DeadlineSplit d;
  d.timeout_left = timeout_ms;
  for (size_t i = 0; i < t0.size(); ++i) {
    const long long waited = std::chrono::duration_cast<std::chrono::milliseconds>(now - t0[i]).count();
    if (timeout_ms && waited >= (long long)timeout_ms) {
      d.worst_wait_ms = waited > d.worst_wait_ms ? waited : d.worst_wait_ms;
    } else {
      if (timeout_ms) {
        const uint32_t left = timeout_ms - (uint32_t)(waited > 0 ? waited : 0);
        d.timeout_left = left < d.timeout_left ? left : d.timeout_left;
      }
      d.live.push_back(i);
    }
  }
  return d;
}

template <class W>  // W derives from QueueWaiter
struct CallQueue {
  std::mutex mu;
  std::vector<W*> queue;          // parked requests, oldest first
  bool busy = false;              // somebody owns the device (or has been handed it)
  uint32_t collect_gen = 0;       // collections so far: later arrivals sleep on the other word
  uint32_t last_batch_calls = 0;  // calls the last batch carried (> 1 arms the owner's batching window)
  std::atomic<uint32_t> wake_word[2] = {{0}, {0}};

  // false: another caller's batch carried `me` (me.status / me.error are final).  true: the caller owns the device and
  // must run `me` together with `served` (parked requests `same` accepted, already off the queue, at most `max_queries`
  // queries in total), then call leave().
  template <class Same>
  bool enter(W& me, Same&& same, uint32_t max_queries, std::vector<W*>& served) {
    std::unique_lock<std::mutex> ql(mu);
    if (busy) {
      me.cohort = collect_gen & 1u;
      queue.push_back(&me);
      ql.unlock();
      std::atomic<uint32_t>& word = wake_word[me.cohort];
      uint32_t st;
      for (;;) {  // the word is read BEFORE the state: a wake between the two makes the kernel's compare fail
        const uint32_t w = word.load(std::memory_order_acquire);
        st = me.state.load(std::memory_order_acquire);
        if (st != QS_PARKED) break;
        cq_futex_wait(&word, w);
      }
      if (st == QS_SERVED) return false;
      // nobody served it: the previous owner took it off the queue and handed it the device (busy stays set)
      ql.lock();
    }
    busy = true;
    // Batching window.  Closed-loop callers split into two cohorts that alternate — the calls that arrived while batch n
    // ran form batch n + 1, whose callers are back just AFTER batch n + 2 was collected — so 64 callers ran as batches of
    // ~32.  When the last batch carried more than one call, the owner waits for stragglers: until nothing new has arrived
    // for ~15 us, at most ~60 us + 1 us per caller of the last batch (one futex call wakes them all, but the kernel
    // releases them one by one) — a single caller never waits.
    if (last_batch_calls > 1) {
      using clk = std::chrono::steady_clock;
      const auto cap = std::chrono::microseconds(60 + (last_batch_calls < 512u ? last_batch_calls : 512u));
      const auto t0 = clk::now();
      auto t_last = t0;
      size_t seen = queue.size();
      for (;;) {
        ql.unlock();
        std::this_thread::yield();
        ql.lock();
        const auto now = clk::now();
        if (queue.size() > seen) {
          seen = queue.size();
          t_last = now;
        }
        if (now - t_last > std::chrono::microseconds(15) || now - t0 > cap || seen >= (size_t)last_batch_calls * 2u) break;
      }
    }
    // (the only allocation of an owner: made before anything is taken off the queue, so that running out of host
    //  memory here hands the device on — leave() with nothing served — instead of stranding the parked callers)
    try {
      served.reserve(served.size() + queue.size());
    } catch (...) {
      ql.unlock();
      leave(std::vector<W*>(), 0, nullptr);
      throw;
    }
    ++collect_gen;
    uint32_t total = me.nq;
    for (auto it = queue.begin(); it != queue.end();) {
      if (same(**it) && total + (*it)->nq <= max_queries) {
        total += (*it)->nq;
        served.push_back(*it);
        it = queue.erase(it);
      } else {
        ++it;
      }
    }
    last_batch_calls = (uint32_t)served.size() + 1u;
    return true;
  }

  // The owner's batch is done: deliver its status to the requests it carried, then hand the device to the oldest
  // parked request (directly: busy stays set) or release it.
  void leave(const std::vector<W*>& served, int32_t status, const char* err) noexcept {
    uint32_t wake_mask = 0;
    {
      std::lock_guard<std::mutex> ql(mu);
      if (!queue.empty()) {
        W* next = queue.front();
        queue.erase(queue.begin());
        wake_mask |= 1u << next->cohort;
        next->state.store(QS_LEAD, std::memory_order_release);
      } else {
        busy = false;
      }
      for (W* f : served) {
        if (!f->decided) {
          f->status = status;
          snprintf(f->error, sizeof f->error, "%s", err ? err : "");
        }
        wake_mask |= 1u << f->cohort;
        f->state.store(QS_SERVED, std::memory_order_release);
      }
    }
    for (uint32_t c = 0; c < 2; ++c)
      if (wake_mask >> c & 1u) {
        wake_word[c].fetch_add(1u, std::memory_order_release);
        cq_futex_wake_all(&wake_word[c]);
      }
  }
};
