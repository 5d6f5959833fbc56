// input_guard.hpp -- a mapped input file that somebody cuts short while it is being read.
//
// The reference reads its input with fread() and reports a short read as "Read input file failed" /
// "Invalid file length" (/root/reference/src/gpu_compressor.cpp:146-150, 299-318).  The GPU pipeline here reads
// through a shared mapping of the file (gpu_compressor.cpp, MappedInput): truncating the file under such a mapping
// turns every access behind the new end into SIGBUS -- in whichever thread touches it, ours (the decoder's header
// walk) or one of the HIP runtime's (a staging copy) -- and the default action kills the process with the
// output half written.  InputGuard answers that signal for the mappings it watches: the rest of the mapping, from
// the faulting page on, is replaced by zero pages (everything from that page on lies behind the new end of the file,
// or it would not have faulted), the access restarts and reads zeros, and the mapping is marked `cut`.  The pipeline
// looks at the mark after every chunk and at the end of the job and fails with the reference's message instead.
// A SIGBUS anywhere else goes to whoever handled it before.
//
// POSIX only, no HIP: compiled into the CLI and, by tests/input_guard_test.cpp, on its own.
#ifndef GPUAR_INPUT_GUARD_HPP
#define GPUAR_INPUT_GUARD_HPP

#include <signal.h>
#include <stdint.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstddef>

namespace gip {

class InputGuard {
  public:
    static constexpr int kSlots = 8;            // mappings watched at a time (one per job; a process runs one job)

    // watches [base, base + size); returns a slot for cut() / unwatch(), or -1 when there is none left (the
    // mapping is then simply not protected, as before)
    static int watch(const void *base, size_t size) {
        install();
        for (int s = 0; s < kSlots; ++s) {
            uintptr_t expect = 0;
            if (slots()[s].base.compare_exchange_strong(expect, 1)) {      // 1: claimed, not yet armed
                slots()[s].cut.store(false);
                slots()[s].size.store(size);
                slots()[s].base.store(reinterpret_cast<uintptr_t>(base));
                return s;
            }
        }
        return -1;
    }
    static void unwatch(int slot) {
        if (slot >= 0 && slot < kSlots) slots()[slot].base.store(0);
    }
    // an access to the mapping has faulted since watch(): the file is shorter than it was when it was mapped
    static bool cut(int slot) { return slot >= 0 && slot < kSlots && slots()[slot].cut.load(); }

  private:
    struct Slot {
        std::atomic<uintptr_t> base{0};
        std::atomic<size_t> size{0};
        std::atomic<bool> cut{false};
    };
    static Slot *slots() {
        static Slot table[kSlots];
        return table;
    }
    static struct sigaction &previous() {
        static struct sigaction old;
        return old;
    }
    static std::atomic<uintptr_t> &pageSize() {
        static std::atomic<uintptr_t> bytes{4096};
        return bytes;
    }
    static void install() {
        static std::atomic<bool> done{false};
        bool expect = false;
        if (!done.compare_exchange_strong(expect, true)) return;
        (void)slots();
        (void)previous();
        // everything the handler touches exists before it can run: no function-local static is first initialised (and
        // sysconf is not called) inside the signal handler
        const long p = ::sysconf(_SC_PAGESIZE);
        if (p > 0) pageSize().store(static_cast<uintptr_t>(p));
        struct sigaction act;
        act.sa_sigaction = &InputGuard::onSigbus;
        sigemptyset(&act.sa_mask);
        act.sa_flags = SA_SIGINFO;
        (void)::sigaction(SIGBUS, &act, &previous());
    }
    // async-signal-safe: atomics, mmap and sigaction only
    static void onSigbus(int sig, siginfo_t *info, void *context) {
        const uintptr_t addr = reinterpret_cast<uintptr_t>(info->si_addr);
        const uintptr_t page = pageSize().load();
        for (int s = 0; s < kSlots; ++s) {
            const uintptr_t base = slots()[s].base.load();
            const size_t size = slots()[s].size.load();
            if (base > 1 && addr >= base && addr < base + size) {
                const uintptr_t from = addr & ~(page - 1);
                void *r = ::mmap(reinterpret_cast<void *>(from), base + size - from, PROT_READ, MAP_FIXED | MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                if (r != MAP_FAILED) {
                    slots()[s].cut.store(true);
                    return;                       // the faulting access restarts and reads zeros
                }
            }
        }
        // not one of ours: whoever was there before; with nobody there, the default action on the re-executed access
        const struct sigaction &old = previous();
        if ((old.sa_flags & SA_SIGINFO) && old.sa_sigaction) {
            old.sa_sigaction(sig, info, context);
        } else if (old.sa_handler != SIG_DFL && old.sa_handler != SIG_IGN) {
            old.sa_handler(sig);
        } else {
            struct sigaction dfl;
            dfl.sa_handler = SIG_DFL;
            sigemptyset(&dfl.sa_mask);
            dfl.sa_flags = 0;
            (void)::sigaction(SIGBUS, &dfl, nullptr);
        }
    }
};

}  // namespace gip
#endif  // GPUAR_INPUT_GUARD_HPP
