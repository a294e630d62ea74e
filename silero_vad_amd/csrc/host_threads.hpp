// host_threads.hpp -- the host-side worker threads of the native helpers (staging.cpp, segmenter.cpp): how many there may
// be, where they run, and a persistent pool so that no call pays for thread creation.
//
// How many.  std::thread::hardware_concurrency() reports the machine (256 on the MI355X hosts); a container usually owns
// far fewer CPUs through its cgroup quota (16 there), and running more busy threads than the quota gets the whole process
// throttled for the rest of the scheduling period -- measured: staging 4 GB on 32 threads under a 16-CPU quota took 458 ms
// instead of 35.  The quota belongs to the NODE, not to the process: with one process per GPU (the reference's
// process-per-worker pattern, examples/parallel_example.ipynb cells 5, 7; here torchrun --nproc-per-node N) every rank
// may use 1/N of it, or N ranks x 16 threads under a 16-CPU quota is exactly the oversubscription above.  So:
//     min(affinity mask, cgroup CPU quota) / LOCAL_WORLD_SIZE, at least 1, at most `cap`
// (LOCAL_WORLD_SIZE is what torchrun exports; SILERO_VAD_AMD_HOST_THREADS overrides the result).
//
// Where.  vad_bind_host_to_device (engine.hip) narrows the calling thread's affinity to the CPUs of the GPU's NUMA node;
// pool threads are created afterwards and inherit it, and pinned buffers allocated afterwards are first touched there.
#pragma once
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace vad {

inline int cgroup_cpu_quota() {            // CPUs' worth of quota, 0 = unlimited / unknown
    long quota = -1, period = -1;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {               // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        if (std::fscanf(f, "%31s %ld", q, &period) == 2 && q[0] != 'm') quota = std::atol(q);
        std::fclose(f);
    } else {                                                                 // cgroup v1
        if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (std::fscanf(fq, "%ld", &quota) != 1) quota = -1;
            std::fclose(fq);
        }
        if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(fp, "%ld", &period) != 1) period = -1;
            std::fclose(fp);
        }
    }
    if (quota <= 0 || period <= 0) return 0;
    return (int)std::max(1L, quota / period);
}

inline int env_int(const char *name) {
    const char *v = std::getenv(name);
    if (!v || !*v) return 0;
    const long x = std::strtol(v, nullptr, 10);
    return x > 0 && x < (1 << 20) ? (int)x : 0;
}

// CPUs this PROCESS may keep busy: the node's budget (affinity mask, cgroup quota) divided among the ranks of the node.
inline int host_cpu_budget() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    const int q = cgroup_cpu_quota();
    if (q > 0) n = std::min(n, q);
    const int ranks = env_int("LOCAL_WORLD_SIZE");
    if (ranks > 1) n = std::max(1, n / ranks);
    return n;
}

inline int default_host_threads(int cap) {
    const int forced = env_int("SILERO_VAD_AMD_HOST_THREADS");
    if (forced) return std::max(1, std::min(forced, 256));
    return std::max(1, std::min(host_cpu_budget(), cap));
}

// A persistent pool: run(nthreads, n, fn) executes fn(k) for k in [0, n) on at most nthreads - 1 pool threads plus the
// caller and returns when all are done.  Threads are created on first use (after any affinity binding the process did) and sleep on a condition
// variable between calls; one run at a time (callers are serialised by a mutex: the engine's users are single-threaded
// per engine, but the pool is process-wide).
//
// fork(): only the forking thread exists in the child, the workers do not.  pthread_atfork handlers take both mutexes around the
// fork (so the child never inherits one held by a thread that is not there) and mark the child's copy of the pool; the child LEAKS
// the stale thread handles (they name threads of the parent: joining or detaching them is undefined, forgetting them is not) before
// its first run and in its destructor, and grows a pool of its own.  Consequences a host should know: a fork() issued while another
// thread is inside a long run() waits for that run; and the handlers can never be unregistered, so the library must not be
// dlclose()d -- it is linked with -z nodelete (__graft_entry__.build) so that a dlclose() leaves it mapped.
class HostPool {
public:
    static HostPool &get() {
        static HostPool p;
        return p;
    }
    int size() {                               // worker threads + the caller
        std::lock_guard<std::mutex> g(api_);
        drop_stale();
        return (int)threads_.size() + 1;
    }
    void run(int nthreads, int n, const std::function<void(int)> &fn) {
        if (n <= 0) return;
        nthreads = std::min(nthreads, n);
        if (nthreads <= 1) {
            for (int k = 0; k < n; ++k) fn(k);
            return;
        }
        std::lock_guard<std::mutex> g(api_);
        drop_stale();
        ensure(nthreads - 1);
        {
            std::lock_guard<std::mutex> l(m_);
            limit_ = nthreads - 1;
            fn_ = &fn;
            next_ = 0;
            total_ = n;
            pending_ = n;
            ++epoch_;
        }
        cv_.notify_all();
        work();                                // the caller takes items too
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    HostPool() { pthread_atfork(&HostPool::before_fork, &HostPool::after_fork_parent, &HostPool::after_fork_child); }
    ~HostPool() {
        drop_stale();
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
            ++epoch_;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    static void before_fork() {
        get().api_.lock();
        get().m_.lock();
    }
    static void after_fork_parent() {
        get().m_.unlock();
        get().api_.unlock();
    }
    static void after_fork_child() {
        HostPool &p = get();
        p.forked_ = true;                       // threads_ lists threads of the PARENT process
        p.fn_ = nullptr;
        p.next_ = p.total_ = p.pending_ = p.limit_ = 0;
        // the parent's workers were asleep on cv_: its copy here still counts waiters that do not exist (a glibc condition variable in
        // that state can block its next signaller for good).  Fresh objects in place, the old ones are not destroyed.
        new (&p.cv_) std::condition_variable();
        new (&p.done_) std::condition_variable();
        new (&p.m_) std::mutex();
        new (&p.api_) std::mutex();
    }
    void drop_stale() {
        if (!forked_) return;
        // handles of threads that do not exist in this process: moved into a heap object that is never destroyed
        new std::vector<std::thread>(std::move(threads_));
        threads_.clear();
        forked_ = false;
    }
    void ensure(int want) {
        want = std::min(want, 255);
        while ((int)threads_.size() < want) {
            const int id = (int)threads_.size();
            threads_.emplace_back([this, id] { loop(id); });
        }
    }
    void work() {
        for (;;) {
            int k;
            const std::function<void(int)> *f;
            {
                std::lock_guard<std::mutex> l(m_);
                if (!fn_ || next_ >= total_) return;
                k = next_++;
                f = fn_;
            }
            (*f)(k);
            std::lock_guard<std::mutex> l(m_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop(int id) {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
                if (id >= limit_) continue;        // this run asked for fewer threads than the pool holds
            }
            work();
        }
    }
    std::mutex api_, m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    const std::function<void(int)> *fn_ = nullptr;
    int next_ = 0, total_ = 0, pending_ = 0, limit_ = 0;
    unsigned long epoch_ = 0;
    bool stop_ = false;
    bool forked_ = false;
};

}  // namespace vad
