// dsp_threads.h -- the one rule of the host thread pools (dsp_text / dsp_freq / dsp_gz / dsp_pgz / dsp_featfile / dsp_sites):
// no exception leaves a std::thread, and none leaves a pool while its threads are joinable.
//
// std::thread's constructor throws std::system_error when the process may not have another thread (RLIMIT_NPROC, a pids
// cgroup: eight ranks with their parser, deflate and runtime threads on one box), a worker that grows a vector may throw
// std::bad_alloc -- and an exception that leaves a std::thread, or unwinds past a vector of joinable threads, or crosses
// the C ABI into ctypes, is std::terminate: the process ends with "Aborted" and no message.
#ifndef DSP_THREADS_H
#define DSP_THREADS_H

#include <atomic>
#include <thread>
#include <vector>

namespace dsp {

// work(t) for every t in [0, nt): t >= 1 on threads of their own where the system gives them, on the calling thread where it
// does not (fewer threads, the same result: every index still runs exactly once).  Returns false when a call of work threw:
// the caller reports DSP_ENOMEM.
template <class F>
bool run_indexed(int nt, F&& work) {
    std::atomic<bool> threw(false);
    auto guarded = [&](int t) {
        try { work(t); } catch (...) { threw.store(true); }
    };
    if (nt <= 1) {
        guarded(0);
        return !threw.load();
    }
    std::vector<std::thread> th;
    int started = 1;
    try {
        th.reserve((size_t)nt - 1);
        for (; started < nt; ++started) th.emplace_back(guarded, started);
    } catch (...) {
    }   // (no more threads to be had: indices [started, nt) run here)
    guarded(0);
    for (int t = started; t < nt; ++t) guarded(t);
    for (auto& x : th) x.join();
    return !threw.load();
}

}  // namespace dsp

#endif
