// stress test of csrc/launch_combiner.h (built and run by tests/test_launch_combiner.py; no HIP)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "../../mamdr_amd/csrc/launch_combiner.h"

using namespace mamdr;

struct Desc { int member, kind, seq; };
struct Log {
    std::atomic<long> delivered{0}, launches{0}, batched{0};
    std::atomic<int> bad{0};
    std::vector<std::atomic<int>> next;          // per member: the sequence number expected next
    explicit Log(int n) : next(n) { for (auto& a : next) a = 0; }
};

static void flush(void* user, int kind, int n, const int* members, const void* const* descs) {
    Log* log = static_cast<Log*>(user);
    log->launches += 1;
    if (n > 1) log->batched += n;
    for (volatile int spin = 0; spin < 2000; ++spin) { }         // (a launch takes a few microseconds)
    for (int i = 0; i < n; ++i) {
        const Desc* d = static_cast<const Desc*>(descs[i]);
        if (d->kind != kind || d->member != members[i]) log->bad += 1;
        if (i && members[i] <= members[i - 1]) log->bad += 1;                 // ascending member order
        if (log->next[d->member].fetch_add(1) != d->seq) log->bad += 1;      // every descriptor exactly once, in order
        log->delivered += 1;
    }
}

int main(int argc, char** argv) {
    const int n_members = argc > 1 ? atoi(argv[1]) : 4;
    const int calls = argc > 2 ? atoi(argv[2]) : 300;
    Log log(n_members);
    LaunchCombiner comb(n_members, flush, &log);
    std::atomic<long> submitted{0};
    std::vector<std::thread> th;
    for (int m = 0; m < n_members; ++m)
        th.emplace_back([&, m]() {
            std::mt19937 rng(1234 + m);
            int seq = 0;
            for (int c = 0; c < calls; ++c) {
                const int steps = (rng() % 9 == 0) ? 0 : 4 + (int)(rng() % 28);     // passes of 0 or 4..31 steps, ragged between members
                if (rng() % 5 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 200));   // "Python" between calls
                if (steps == 0) continue;
                comb.enter(m);
                for (int s = 0; s < steps; ++s)
                    for (int kind = 0; kind < COMB_KINDS; ++kind) {
                        Desc d{m, kind, seq++};
                        comb.submit(m, kind, &d);
                        submitted += 1;
                    }
                comb.leave(m);
            }
        });
    // a watchdog instead of a hang: the test harness also bounds the run
    std::atomic<bool> done{false};
    std::thread dog([&]() {
        for (int i = 0; i < 600 && !done; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!done) { fprintf(stderr, "DEADLOCK: %ld submitted, %ld delivered\n", submitted.load(), log.delivered.load()); _Exit(3); }
    });
    for (auto& t : th) t.join();
    done = true;
    dog.join();
    printf("members %d submitted %ld delivered %ld launches %ld batched %ld bad %d carried %llu\n", n_members, submitted.load(),
           log.delivered.load(), log.launches.load(), log.batched.load(), log.bad.load(), (unsigned long long)comb.carried());
    return (log.bad == 0 && submitted == log.delivered && (long)comb.carried() == submitted.load()) ? 0 : 1;
}
