// host_pool.h -- ONE budget of host CPUs for everything in this library that runs on the host beside the GPU (round 5).
//
// The reference's `--cpu N` (pyseer/__main__.py:541-568) is a pool of N worker processes; here the per-variant work is on the device and the
// host threads only feed and drain it: the k-mer readers (reader.cpp), the stager of pageable rows (api.hip), the result formatter
// (writer.cpp).  Until round 4 each of them sized itself alone (a 16-thread OpenMP team per formatter call, up to 16 std::threads per staged
// chunk per context from hardware_concurrency(), a parser pool per reader): eight device streams on a 16-CPU quota were 8-16x oversubscribed.
//
//  * host_cpus()      the CPUs this process may really use: the affinity mask cut by the cgroup quota (a GPU box shows 256 logical CPUs
//                     under a quota of 16), or what sh_set_host_threads() says.
//  * HostPool         one persistent pool of host_cpus() - 1 workers for the whole process.  run(count, chunk, fn) shares `count` items out
//                     between the calling thread and whatever workers are idle; any number of callers may be inside run() at once (each
//                     brings its own thread, the workers go where work is), so two sinks format concurrently and a third caller's copy
//                     takes the workers they leave.  Nothing is created or joined per call.
//  * ParPool          the fixed-size fork-join pool the readers keep for themselves (a reader's decoder threads hold state between calls);
//                     its size comes from host_cpus() / host_streams().
//  * cpu accounting   every piece of host work adds its THREAD CPU time (CLOCK_THREAD_CPUTIME_ID) to a per-stage counter; sh_host_cpu_seconds()
//                     reports them (tools/gpu_e2e_job.py -> profiles/r05/host_budget.json).
#ifndef SEERHIP_HOST_POOL_H
#define SEERHIP_HOST_POOL_H
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>
#include <sched.h>

namespace shost {

enum Stage { ST_STAGE_COPY = 0, ST_FORMAT, ST_REGISTER, ST_SUBMIT, ST_COLLECT, ST_READER_PARSE, ST_READER_DECODE, ST_READER_CRC, ST_OTHER, ST_COUNT };
static const char *const kStageName[ST_COUNT] = {"stage_copy", "format", "register", "submit", "collect", "reader_parse", "reader_decode", "reader_crc", "other"};

inline std::atomic<int64_t> *stage_ns() { static std::atomic<int64_t> a[ST_COUNT]; return a; }
inline int64_t thread_cpu_ns() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec; }
struct CpuScope {                                  // adds the calling thread's CPU time between construction and destruction to a stage
    int st; int64_t t0;
    explicit CpuScope(int stage) : st(stage), t0(thread_cpu_ns()) {}
    ~CpuScope() { stage_ns()[st].fetch_add(thread_cpu_ns() - t0, std::memory_order_relaxed); }
};

inline std::atomic<int> &cpus_override() { static std::atomic<int> v{0}; return v; }
inline std::atomic<int> &streams() { static std::atomic<int> v{1}; return v; }

// CPUs this process can use: the affinity mask, cut by the cgroup (v2, then v1) CPU quota
inline int detected_cpus()
{
    static const int n = [] {
        long t = (long)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set; CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) t = std::min<long>(t, c); }
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32]; long per = 0;
            if (fscanf(f, "%31s %ld", q, &per) == 2 && q[0] != 'm' && per > 0) { const long qq = atol(q); if (qq > 0) t = std::min(t, (qq + per - 1) / per); }
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            long qq = -1, per = 0;
            if (fscanf(g, "%ld", &qq) == 1 && qq > 0)
                if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &per) == 1 && per > 0) t = std::min(t, (qq + per - 1) / per); fclose(h); }
            fclose(g);
        }
        return (int)std::max<long>(1, t);
    }();
    return n;
}
inline int host_cpus() { const int o = cpus_override().load(); return o > 0 ? o : detected_cpus(); }
// device streams (or readers) the caller runs at once: what a per-stream component may take of the budget
inline int host_streams() { return std::max(1, streams().load()); }
inline int per_stream_cpus(int at_least = 1) { return std::max(at_least, (host_cpus() + host_streams() - 1) / host_streams()); }

// ---- fixed-size fork-join pool (one client): the readers' parser / decoder teams ---------------------------------------------------------
struct ParPool {
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv_work, cv_done;
    std::function<void(int64_t)> fn; int64_t n = 0, chunk = 1; std::atomic<int64_t> next{0}; int busy = 0; uint64_t gen = 0; bool quit = false;
    int stage = ST_OTHER;
    explicit ParPool(int nthreads, int stage_ = ST_OTHER) : stage(stage_) {
        for (int t = 0; t < nthreads; ++t) th.emplace_back([this] {
            uint64_t seen = 0;
            for (;;) {
                { std::unique_lock<std::mutex> lk(mu); cv_work.wait(lk, [&] { return quit || gen != seen; }); if (quit) return; seen = gen; }
                { CpuScope cs(stage); work(); }
                { std::lock_guard<std::mutex> lk(mu); if (--busy == 0) cv_done.notify_all(); }
            }
        });
    }
    void work() { for (;;) { const int64_t i0 = next.fetch_add(chunk); if (i0 >= n) return; const int64_t i1 = std::min(n, i0 + chunk); for (int64_t i = i0; i < i1; ++i) fn(i); } }
    void run(int64_t count, int64_t chunk_, std::function<void(int64_t)> f) {
        if (count <= 0) return;
        if (th.empty() || count == 1) { for (int64_t i = 0; i < count; ++i) f(i); return; }
        { std::lock_guard<std::mutex> lk(mu); fn = std::move(f); n = count; chunk = std::max<int64_t>(1, chunk_); next = 0; busy = (int)th.size(); ++gen; }
        cv_work.notify_all();
        work();                                                       // the caller works too
        std::unique_lock<std::mutex> lk(mu); cv_done.wait(lk, [&] { return busy == 0; });
    }
    ~ParPool() { { std::lock_guard<std::mutex> lk(mu); quit = true; } cv_work.notify_all(); for (auto &t : th) t.join(); }
};

// ---- the process-wide pool (any number of clients) ---------------------------------------------------------------------------------------
class HostPool {
    struct Job { const std::function<void(int64_t)> *fn; int64_t n, chunk; std::atomic<int64_t> next{0}; int active = 0, want = 0, stage = ST_OTHER; };
    std::mutex mu; std::condition_variable cv_work, cv_done;
    std::vector<Job *> jobs; std::vector<std::thread> th; bool quit = false;
    static void work(Job *j) { for (;;) { const int64_t i0 = j->next.fetch_add(j->chunk); if (i0 >= j->n) return; const int64_t i1 = std::min(j->n, i0 + j->chunk); for (int64_t i = i0; i < i1; ++i) (*j->fn)(i); } }
    Job *pick() { for (Job *x : jobs) if (x->active < x->want && x->next.load(std::memory_order_relaxed) < x->n) return x; return nullptr; }   // (mu held)
    void worker() {
        pthread_setname_np(pthread_self(), "sh-pool");
        for (;;) {
            Job *j = nullptr;
            { std::unique_lock<std::mutex> lk(mu); cv_work.wait(lk, [&] { return quit || (j = pick()) != nullptr; }); if (quit) return; ++j->active; }
            { CpuScope cs(j->stage); work(j); }
            { std::lock_guard<std::mutex> lk(mu); if (--j->active == 0) cv_done.notify_all(); }
        }
    }
    void grow(int n) { while ((int)th.size() < n) th.emplace_back([this] { worker(); }); }   // (mu held)
public:
    // count items in chunks of `chunk`, on the caller and at most max_threads - 1 pool workers (0 = the whole budget); the caller's own
    // share of the CPU time is accounted by the caller (CpuScope), the workers' here under `stage`
    void run(int64_t count, int64_t chunk, const std::function<void(int64_t)> &f, int stage = ST_OTHER, int max_threads = 0) {
        if (count <= 0) return;
        const int budget = host_cpus();
        int want = max_threads > 0 ? std::min(max_threads, budget) : budget;
        chunk = std::max<int64_t>(1, chunk);
        want = (int)std::min<int64_t>(want, (count + chunk - 1) / chunk);
        if (want <= 1) { for (int64_t i = 0; i < count; ++i) f(i); return; }
        Job j; j.fn = &f; j.n = count; j.chunk = chunk; j.want = want - 1; j.stage = stage;
        { std::lock_guard<std::mutex> lk(mu); grow(budget - 1); jobs.push_back(&j); }
        cv_work.notify_all();
        work(&j);
        std::unique_lock<std::mutex> lk(mu);
        jobs.erase(std::find(jobs.begin(), jobs.end(), &j));          // no worker can pick it any more; wait for those inside
        cv_done.wait(lk, [&] { return j.active == 0; });
    }
    int workers() { std::lock_guard<std::mutex> lk(mu); return (int)th.size(); }
    ~HostPool() { { std::lock_guard<std::mutex> lk(mu); quit = true; } cv_work.notify_all(); for (auto &t : th) t.join(); }
};
// (leaked on purpose: worker threads must not be joined from a static destructor while Python threads may still be inside the library)
inline HostPool &pool() { static HostPool *p = new HostPool(); return *p; }

}  // namespace shost
#endif
