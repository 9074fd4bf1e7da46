// sunscreen_amd/csrc/flat_combiner.hpp -- the queueing protocol behind the combined handle-level calls (capi.cpp Combiner),
// free of any device code so that it can be exercised on its own (tests/native/combiner_tsan.cpp: ThreadSanitizer, mock executor).
//
// Flat combining: run(req) queues the request; the caller then either finds it completed by somebody else or, when fewer than
// max_leaders batches are in flight, leads: it takes the oldest queued request and everything queued behind it that
// compatible(head, r) admits (at most max_batch), runs execute(batch) outside the lock and marks the batch done.  Requests that
// arrive while a batch is executing pile up and form the next batch.  Waiting callers watch two atomics -- their own request's
// `done` and `leaders` -- and take the mutex only to lead: with sixty-odd waiters a condition variable's mutex hand-over took
// longer than a batch.
#pragma once
#include <time.h>

#include <atomic>
#include <cstddef>
#include <deque>
#include <mutex>
#include <vector>

namespace hipbfv {

// Req must hold `std::atomic<bool> done` (initially false).  Whatever execute() writes into a request happens-before its owner's
// return from run().  A request's owner may return -- and the request go out of scope -- the moment it sees `done`: a leader
// touches nothing of a request after that store.
template <class Req>
class FlatCombiner {
 public:
  template <class Compatible, class Execute>
  void run(Req& req, int max_leaders, size_t max_batch, Compatible&& compatible, Execute&& execute) {
    {
      std::lock_guard<std::mutex> g(mu_);
      q_.push_back(&req);  // may throw (nothing is queued then)
    }
    for (unsigned spins = 0;; spins++) {
      if (req.done.load(std::memory_order_acquire)) return;
      if (leaders_.load(std::memory_order_relaxed) < max_leaders) {
        std::unique_lock<std::mutex> lk(mu_);
        if (req.done.load(std::memory_order_acquire)) return;
        if (leaders_.load(std::memory_order_relaxed) < max_leaders && !q_.empty()) {
          // Everything that can throw happens BEFORE any state changes: once the leader slot is taken and requests are off the
          // queue, nothing below allocates (push_back into reserved storage, deque::erase of pointers), so a batch can neither
          // be lost nor leaders_ stay raised.  If the reservation itself fails, this caller cannot lead: it withdraws its own
          // request when that is still queued (then nobody else holds a pointer to it) and reports; if another leader already
          // took the request, it keeps waiting for it -- the request must not go out of scope under that leader.
          std::vector<Req*> batch;
          try {
            batch.reserve(q_.size() < max_batch ? q_.size() : max_batch);
          } catch (...) {
            for (auto it = q_.begin(); it != q_.end(); ++it) {
              if (*it == &req) {
                q_.erase(it);
                throw;
              }
            }
            lk.unlock();
            continue;
          }
          leaders_.fetch_add(1, std::memory_order_relaxed);
          Req* head = q_.front();
          for (auto it = q_.begin(); it != q_.end() && batch.size() < max_batch;) {
            if (*it == head || compatible(*head, **it)) {
              batch.push_back(*it);
              it = q_.erase(it);
            } else {
              ++it;
            }
          }
          lk.unlock();
          execute(batch);  // must not throw
          for (Req* r : batch) r->done.store(true, std::memory_order_release);
          leaders_.fetch_sub(1, std::memory_order_release);
          spins = 0;
          continue;
        }
      }
      if (spins < 4096) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
      } else {
        // a long operation (large degree, big batch) is running: stop burning the core
        struct timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
      }
    }
  }

 private:
  std::mutex mu_;
  std::deque<Req*> q_;
  std::atomic<int> leaders_{0};  // batches in flight
};

}  // namespace hipbfv
