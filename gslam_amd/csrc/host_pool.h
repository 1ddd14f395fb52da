// Host thread pool shared by the host-side set-up code of the BA path (ba.hip: index lists; ba_order.hip: camera order).
#pragma once
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// A few persistent host threads for the index lists below (spawning threads per solve cost more than the lists).
class HostPool {
 public:
  static HostPool& get() {
    static HostPool p;
    return p;
  }
  int size() const { return (int)workers_.size() + 1; }
  // runs f(t) for t in [0, n_tasks) on the pool threads and the caller; returns when all are done
  template <typename F>
  void run(int n_tasks, F&& f) {
    if (n_tasks <= 1 || workers_.empty()) {
      for (int t = 0; t < n_tasks; ++t) f(t);
      return;
    }
    std::unique_lock<std::mutex> call(call_mu_);  // one parallel region at a time
    std::function<void(int)> fn = f;
    {
      std::lock_guard<std::mutex> l(mu_);
      fn_ = &fn;
      n_tasks_ = n_tasks;
      next_ = 0;
      pending_ = n_tasks;
      ++epoch_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> l(mu_);
    done_cv_.wait(l, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  HostPool() {
    unsigned hw = std::thread::hardware_concurrency();
    int n = (int)(hw ? hw : 4) - 1;
    if (n > 15) n = 15;
    if (const char* e = getenv("GSLAM_HIP_HOST_THREADS")) n = atoi(e) - 1;
    for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_ = true;
      ++epoch_;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  void work() {
    for (;;) {
      int t;
      std::function<void(int)>* fn;
      {
        std::lock_guard<std::mutex> l(mu_);
        if (!fn_ || next_ >= n_tasks_) return;
        t = next_++;
        fn = fn_;
      }
      (*fn)(t);
      std::lock_guard<std::mutex> l(mu_);
      if (--pending_ == 0) done_cv_.notify_all();
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (stop_) return;
      }
      work();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_cv_;
  std::function<void(int)>* fn_ = nullptr;
  int n_tasks_ = 0, next_ = 0, pending_ = 0;
  unsigned long long epoch_ = 0;
  bool stop_ = false;
};

