// Host-side rendezvous that turns the step launches of several contexts into ONE launch each (round 6; pure C++, no HIP:
// tests/test_launch_combiner.py compiles it with g++ and hammers it from threads).
//
// The members of a group are contexts driven by host threads of their own (the lanes of mamdr_amd/parallel.py), all on ONE
// stream.  While a member is inside mamdr_train_steps(_n) it is ACTIVE.  At each of the three launch sites of a slab-path
// step (tower, weight gradients, update) an active member SUBMITS its launch descriptor instead of launching and waits;
// when every active member is waiting (or has left), the last one to arrive issues the launches of the EARLIEST stage
// anybody waits at -- together, in member order, through the group's callback -- and releases those members; members
// further along in their step wait on, so lanes that drifted out of phase fall back into step.  A member that is not inside a
// call is not waited for (it may be in a collective, an evaluation, Python): whoever is stepping at the moment is batched,
// nobody else is held up, and since a waiting member only ever waits for members that are themselves inside a call and
// therefore on their way to their next submit or to leave, there is no cycle to deadlock on.  Every member's arithmetic is
// independent of who it was batched with: the bodies a batched launch runs are the bodies of the single launches.
//
// Waiting is a spin on a generation counter (a condition variable's wake-up costs 5 - 20 us, three of them per step would
// make the host the bottleneck of a 40 us batched step); after a few thousand spins the waiter yields its time slice.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>
#include <mutex>
#include <thread>

namespace mamdr {

enum { COMB_TOWER = 0, COMB_WGRAD = 1, COMB_UPDATE = 2, COMB_KINDS = 3 };

class LaunchCombiner {
  public:
    static constexpr int MAX_MEMBERS = 8;
    // issue the launches of `n` descriptors of one kind (members in ascending order); called with the group locked
    using Flush = void (*)(void* user, int kind, int n, const int* members, const void* const* descs);

    LaunchCombiner(int n_members, Flush fn, void* user) : n_(n_members), fn_(fn), user_(user) {
        for (int i = 0; i < MAX_MEMBERS; ++i) {
            inside_[i] = present_[i] = false;
            released_[i].store(0);
        }
    }
    int members() const { return n_; }

    // the member's thread is inside a training call from now on: its submits are waited for
    void enter(int member) {
        std::lock_guard<std::mutex> lk(mu_);
        if (!inside_[member]) {
            inside_[member] = true;
            active_ += 1;
        }
    }
    // ... and no longer: members waiting for it alone are released
    void leave(int member) {
        std::lock_guard<std::mutex> lk(mu_);
        if (inside_[member]) {
            inside_[member] = false;
            active_ -= 1;
            while (arrived_ > 0 && arrived_ >= active_) flush_kind_locked(-1, -1);
        }
    }
    // returns once the launch that carries `desc` has been issued (`desc` must stay valid until then: the caller's frame)
    void submit(int member, int kind, const void* desc) {
        uint64_t g;
        {
            std::lock_guard<std::mutex> lk(mu_);
            present_[member] = true;
            kind_[member] = kind;
            desc_[member] = desc;
            arrived_ += 1;
            g = released_[member].load(std::memory_order_relaxed);
            if (!inside_[member]) {                 // (a member that never entered is served at once, alone)
                flush_kind_locked(kind, member);
            } else {
                while (arrived_ > 0 && arrived_ >= active_) flush_kind_locked(-1, -1);
            }
            if (released_[member].load(std::memory_order_relaxed) != g) return;
        }
        int spins = 0;
        auto t0 = std::chrono::steady_clock::now();
        while (released_[member].load(std::memory_order_acquire) == g) {
            if ((++spins & 63) == 0 && wait_ns_ > 0) {
                // a bounded wait: members that are held active between their calls (hold) but busy elsewhere for longer than
                // this are not waited for -- whoever is here goes (earliest stage first), the late member joins the next launch
                const auto now = std::chrono::steady_clock::now();
                if (std::chrono::duration_cast<std::chrono::nanoseconds>(now - t0).count() > wait_ns_) {
                    std::lock_guard<std::mutex> lk(mu_);
                    if (released_[member].load(std::memory_order_relaxed) == g) flush_kind_locked(-1, -1);
                    t0 = now;
                }
            }
            if (spins > 4096) {
                std::this_thread::yield();
                spins = 0;
            } else {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
            }
        }
    }
    void set_wait_ns(int64_t ns) { wait_ns_ = ns; }
    // launches issued / descriptors carried so far (reports and tests)
    uint64_t launches() const { return launches_.load(); }
    uint64_t carried() const { return carried_.load(); }

  private:
    // Every active member is waiting: issue the launches of ONE kind -- the earliest stage of a step anybody waits at
    // (tower < weight gradients < update) -- and release those members only.  Members further along in their step keep
    // waiting until the others have caught up: lanes that drifted out of phase (one at its tower, one at its update) are
    // back in step within one step instead of sharing no launch ever again.
    // only >= 0: that member alone (a submit from outside a call).
    void flush_kind_locked(int kind, int only) {
        if (kind < 0) {
            kind = COMB_KINDS;
            for (int i = 0; i < n_; ++i)
                if (present_[i] && kind_[i] < kind) kind = kind_[i];
            if (kind == COMB_KINDS) return;
        }
        int who[MAX_MEMBERS];
        const void* what[MAX_MEMBERS];
        int n = 0;
        for (int i = 0; i < n_; ++i)
            if (present_[i] && kind_[i] == kind && (only < 0 || i == only)) {
                who[n] = i;
                what[n] = desc_[i];
                n += 1;
            }
        if (!n) return;
        fn_(user_, kind, n, who, what);
        launches_.fetch_add(1, std::memory_order_relaxed);
        carried_.fetch_add((uint64_t)n, std::memory_order_relaxed);
        for (int j = 0; j < n; ++j) {
            present_[who[j]] = false;
            released_[who[j]].fetch_add(1, std::memory_order_release);
        }
        arrived_ -= n;
    }

    const int n_;
    Flush fn_;
    void* user_;
    std::mutex mu_;
    int active_ = 0, arrived_ = 0;
    bool inside_[MAX_MEMBERS], present_[MAX_MEMBERS];
    int kind_[MAX_MEMBERS];
    const void* desc_[MAX_MEMBERS];
    std::atomic<uint64_t> released_[MAX_MEMBERS];
    std::atomic<uint64_t> launches_{0}, carried_{0};
    int64_t wait_ns_ = 0;           // 0: wait for every active member however long
};

}  // namespace mamdr
