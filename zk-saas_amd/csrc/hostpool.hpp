// A small persistent pool of host worker threads owned by a context.
//
// What runs here is the host-side part of the reference's per-party work that is a few hundred group operations on
// single points: the king's unpack2 + sum over n masked points (dist-primitives/src/dmsm/mod.rs:85-86), the scalar
// multiples of CRS constants and of the out-masks in prove.rs:40-56, 99-110, 229-235, and the final window fold of an
// MSM.  All of it is submitted when a proof starts and runs beside the device work; nothing spawns threads per call.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <thread>
#include <vector>

namespace zk {

class HostPool {
 public:
  explicit HostPool(int nthreads, std::function<void()> on_start = nullptr) {
    for (int i = 0; i < nthreads; i++)
      workers_.emplace_back([this, on_start]() {
        if (on_start) on_start();
        for (;;) {
          std::packaged_task<void()> job;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
            if (q_.empty()) return;       // stop_ and drained
            job = std::move(q_.front());
            q_.pop_front();
          }
          busy_.fetch_add(1, std::memory_order_relaxed);
          job();
          busy_.fetch_sub(1, std::memory_order_relaxed);
        }
      });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  std::future<void> submit(std::function<void()> fn) {
    std::packaged_task<void()> job(std::move(fn));
    std::future<void> f = job.get_future();
    {
      std::lock_guard<std::mutex> lk(mu_);
      q_.push_back(std::move(job));
    }
    cv_.notify_one();
    return f;
  }
  // Runs ONE queued task on the calling thread (false: the queue is empty).  What a thread that waits for sub-tasks does
  // instead of blocking: a pool whose every worker waits for work queued behind it cannot starve (round 6; round 5 guarded
  // its nested submits with idle() >= jobs, a check-then-act race).
  bool run_one() {
    std::packaged_task<void()> job;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (q_.empty()) return false;
      job = std::move(q_.front());
      q_.pop_front();
    }
    busy_.fetch_add(1, std::memory_order_relaxed);
    job();
    busy_.fetch_sub(1, std::memory_order_relaxed);
    return true;
  }
  // Waits for `f`, running queued tasks meanwhile; false when `deadline` passes first (`f` stays valid).
  // Tasks picked up here may themselves wait on the device or on a flag raised by a task dequeued EARLIER (the only
  // kinds of wait the library's tasks contain), never on the caller's frame: every task raises its flags before it
  // reaches a point where it waits for sub-tasks.
  bool wait_helping(std::future<void>& f, std::chrono::steady_clock::time_point deadline) {
    using namespace std::chrono;
    for (int spins = 0;; spins++) {
      if (f.wait_for(seconds(0)) == std::future_status::ready) return true;
      if (run_one()) continue;
      if ((spins & 63) == 63 && steady_clock::now() > deadline) return false;
      if (f.wait_for(microseconds(50)) == std::future_status::ready) return true;
    }
  }
  int size() const { return (int)workers_.size(); }
  // workers not running a task right now (a task that fans out to the pool and waits checks this first: waiting for
  // sub-tasks that no free worker can pick up would be a deadlock)
  int idle() const { return (int)workers_.size() - busy_.load(std::memory_order_relaxed); }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::packaged_task<void()>> q_;
  std::vector<std::thread> workers_;
  bool stop_ = false;
  std::atomic<int> busy_{0};
};

}  // namespace zk
