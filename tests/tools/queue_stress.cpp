// queue_stress.cpp — TEST TOOL: the handle's coalescing queue (lancedb_amd/csrc/call_queue.h, the very header the library
// compiles) driven by N threads on the CPU, with a mutex standing in for the device.  Built with -fsanitize=thread by
// tests/test_build_host.py: data races in the hand-over (who may touch a parked request, when), lost wake-ups (the
// program would hang: the test runs it under a timeout) and mixed-up results all show here, without a GPU.
//
//   g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -I lancedb_amd/csrc tests/tools/queue_stress.cpp -o queue_stress
//   ./queue_stress [threads] [calls per thread]      -> "ok ..." and exit code 0
#include <cstring>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "call_queue.h"

struct Req : QueueWaiter {
  int param = 0;          // only requests with equal parameters share a batch
  uint64_t payload = 0;   // "query"
  uint64_t result = 0;    // "output buffer" of the caller, written by whoever runs the batch
};

// the deadline rule of a coalesced batch (call_queue.h split_by_deadline): every call against its own entry time
static int check_deadline_rule() {
  using clk = std::chrono::steady_clock;
  using ms = std::chrono::milliseconds;
  int bad = 0;
  const clk::time_point now = clk::now();
  // budget 50 ms: waited 70 (expired), 10 (40 left), 49 (1 left), 50 (expired: the budget is spent), 0 (50 left)
  const std::vector<clk::time_point> t0 = {now - ms(70), now - ms(10), now - ms(49), now - ms(50), now};
  DeadlineSplit d = split_by_deadline(t0, now, 50);
  bad += !(d.live == std::vector<size_t>{1, 2, 4});
  bad += !(d.timeout_left == 1u);      // the device is armed with the SMALLEST budget left among the live calls
  bad += !(d.worst_wait_ms == 70);
  d = split_by_deadline(t0, now, 0);   // no timeout: everybody is live, nothing to arm
  bad += !(d.live.size() == 5 && d.timeout_left == 0u);
  d = split_by_deadline({now - ms(5)}, now, 5);  // a single call whose budget is exactly spent
  bad += !(d.live.empty() && d.worst_wait_ms == 5);
  d = split_by_deadline({now + ms(3)}, now, 20);  // (a clock read before the call's own stamp: never more than the budget)
  bad += !(d.live.size() == 1 && d.timeout_left == 20u);
  return bad;
}

int main(int argc, char** argv) {
  const unsigned n_threads = argc > 1 ? (unsigned)atoi(argv[1]) : 48, per = argc > 2 ? (unsigned)atoi(argv[2]) : 400;
  CallQueue<Req> q;
  std::atomic<int> in_device{0};
  std::atomic<uint64_t> batches{0}, carried{0}, failed_calls{0}, errors{0}, biggest{0}, own_outcomes{0}, own_seen{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < n_threads; ++t)
    th.emplace_back([&, t] {
      for (unsigned i = 0; i < per; ++i) {
        Req me;
        me.nq = 1 + (t + i) % 3;
        me.param = (int)(t % 3);
        me.payload = ((uint64_t)t << 32) | i;
        std::vector<Req*> served;
        int32_t status;
        if (q.enter(me, [&](const Req& o) { return o.param == me.param; }, 64u, served)) {
          // this caller owns the device: nobody else may be in here
          if (in_device.fetch_add(1) != 0) errors.fetch_add(1);
          uint32_t total = me.nq;
          for (Req* o : served) {
            if (o->param != me.param) errors.fetch_add(1);
            total += o->nq;
          }
          if (total > 64u) errors.fetch_add(1);
          const uint64_t b = batches.fetch_add(1);
          status = (b % 97 == 96) ? 7 : 0;  // every 97th batch "fails": its callers must all see the error
          if (status == 0) {
            me.result = me.payload * 3 + 1;
            for (Req* o : served) {
              // a call's deadline is its own: every 5th carried request "timed out before the device was reached" — the
              // owner writes that request's own outcome, and leave() must deliver it instead of the batch's status
              if (o->payload % 5 == 4) {
                o->status = 3;
                snprintf(o->error, sizeof o->error, "late");
                o->decided = true;
                own_outcomes.fetch_add(1);
              } else {
                o->result = o->payload * 3 + 1;
              }
            }
          }
          carried.fetch_add(served.size() + 1);
          uint64_t big = biggest.load();
          while (served.size() + 1 > big && !biggest.compare_exchange_weak(big, served.size() + 1)) {
          }
          if ((b & 7) == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));  // (a batch takes a while)
          in_device.fetch_sub(1);
          q.leave(served, status, status ? "boom" : "");
        } else {
          status = me.status;
          if (status == 7 && strcmp(me.error, "boom") != 0) errors.fetch_add(1);
          if (status == 3 && (strcmp(me.error, "late") != 0 || me.payload % 5 != 4)) errors.fetch_add(1);
          if (status != 0 && status != 3 && status != 7) errors.fetch_add(1);
          if (status == 3) own_seen.fetch_add(1);
        }
        if (status == 0) {
          if (me.result != me.payload * 3 + 1) errors.fetch_add(1);
        } else {
          failed_calls.fetch_add(1);
        }
      }
    });
  for (auto& x : th) x.join();
  const uint64_t calls = (uint64_t)n_threads * per;
  if (carried.load() != calls) errors.fetch_add(1);  // every call ran in exactly one batch
  if (q.busy || !q.queue.empty()) errors.fetch_add(1);
  if (own_outcomes.load() != own_seen.load()) errors.fetch_add(1);  // every per-request outcome reached its caller
  errors.fetch_add((uint64_t)check_deadline_rule());
  std::printf("%s: %llu calls in %llu batches (largest %llu), %llu calls saw their batch fail, %llu errors\n",
              errors.load() ? "FAILED" : "ok", (unsigned long long)calls, (unsigned long long)batches.load(),
              (unsigned long long)biggest.load(), (unsigned long long)failed_calls.load(), (unsigned long long)errors.load());
  return errors.load() ? 1 : 0;
}
