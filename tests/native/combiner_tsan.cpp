// tests/native/combiner_tsan.cpp -- the flat-combining protocol of the handle-level calls (sunscreen_amd/csrc/flat_combiner.hpp)
// under ThreadSanitizer with a mock executor: 48 threads, three request kinds, one or two leaders.  Checks: every request is
// executed exactly once, in a batch of its own kind, no larger than the cap; results written by a leader are visible to the owner;
// no request is lost (the program terminates) and the sanitizer reports no race.
//
//   g++ -O1 -g -std=c++17 -fsanitize=thread -Isunscreen_amd/csrc tests/native/combiner_tsan.cpp -lpthread
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "flat_combiner.hpp"

struct Req {
  int kind = 0;
  long in = 0;
  long out = 0;        // written by whoever executes the request
  int executions = 0;  // must end up 1
  int batch_size = 0;
  std::atomic<bool> done{false};
};

int main(int argc, char** argv) {
  const int leaders = argc > 1 ? atoi(argv[1]) : 1;
  const int T = 48, iters = 400;
  const size_t cap = 16;
  hipbfv::FlatCombiner<Req> comb;
  std::atomic<long> batches{0}, items{0}, bad{0};
  std::atomic<int> in_flight{0}, max_in_flight{0};
  auto execute = [&](const std::vector<Req*>& batch) {
    const int now = ++in_flight;
    int seen = max_in_flight.load();
    while (now > seen && !max_in_flight.compare_exchange_weak(seen, now)) {
    }
    if (batch.empty() || batch.size() > cap) bad++;
    for (Req* r : batch) {
      if (r->kind != batch[0]->kind) bad++;
      r->out = r->in * 3 + r->kind;
      r->executions++;
      r->batch_size = (int)batch.size();
    }
    batches++;
    items += (long)batch.size();
    std::this_thread::sleep_for(std::chrono::microseconds(30 + 5 * batch.size()));  // the "device" is busy
    --in_flight;
  };
  std::vector<std::thread> ths;
  for (int t = 0; t < T; t++)
    ths.emplace_back([&, t] {
      for (int it = 0; it < iters; it++) {
        Req r;
        r.kind = (t + it) % 3;
        r.in = (long)t * 100000 + it;
        comb.run(r, leaders, cap, [](const Req& h, const Req& x) { return h.kind == x.kind; }, execute);
        if (r.executions != 1 || r.out != r.in * 3 + r.kind || r.batch_size < 1) bad++;
      }
    });
  for (auto& th : ths) th.join();
  if (bad.load() || items.load() != (long)T * iters || max_in_flight.load() > leaders) {
    std::printf("FAILED bad=%ld items=%ld max_in_flight=%d\n", bad.load(), items.load(), max_in_flight.load());
    return 2;
  }
  std::printf("combiner ok: %ld requests in %ld batches (%.1f per batch), at most %d in flight\n", items.load(), batches.load(),
              (double)items.load() / (double)batches.load(), max_in_flight.load());
  return 0;
}
