// input_guard_test.cpp -- host/input_guard.hpp on its own (no HIP, no GPU): a file is mapped, cut short, and read
// through the mapping, by this thread and by another one; the process must survive, read zeros behind the cut, see the
// mark -- and still die of a SIGBUS that is none of the guard's business (checked by the caller through the exit status).
//   input_guard_test FILE [foreign]
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "input_guard.hpp"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const size_t n = 8u << 20;
    const int fd = ::open(argv[1], O_RDWR | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) return 2;
    std::vector<uint8_t> ones(n, 0xAB);
    if (::write(fd, ones.data(), n) != static_cast<ssize_t>(n)) return 2;
    const uint8_t *m = static_cast<const uint8_t *>(::mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0));
    if (m == MAP_FAILED) return 2;
    if (argc > 2 && !std::strcmp(argv[2], "foreign")) {
        // a mapping nobody watches (but the handler is installed: another mapping is watched): the default action must still apply
        static uint8_t other[4096];
        const int slot = gip::InputGuard::watch(other, sizeof other);
        if (::ftruncate(fd, 4096) != 0) return 2;
        volatile uint8_t v = m[n - 1];
        (void)v;
        std::printf("survived a foreign SIGBUS (slot %d)\n", slot);
        return 1;
    }
    const int slot = gip::InputGuard::watch(m, n);
    if (slot < 0 || gip::InputGuard::cut(slot)) return 3;
    unsigned long sum = 0;
    for (size_t i = 0; i < n; i += 4096) sum += m[i];
    if (sum != 0xABul * (n / 4096) || gip::InputGuard::cut(slot)) return 4;      // intact file: nothing happens
    if (::ftruncate(fd, 1u << 20) != 0) return 2;                                 // 1 MiB left of 8
    unsigned long head = 0, tail = 0;
    std::thread other([&] {                                                       // the fault is taken on another thread first
        for (size_t i = 5u << 20; i < n; i += 4096) tail += m[i];
    });
    other.join();
    for (size_t i = 0; i < (1u << 20); i += 4096) head += m[i];
    for (size_t i = 1u << 20; i < (5u << 20); i += 4096) tail += m[i];            // ... then here, below the first fault
    if (!gip::InputGuard::cut(slot)) return 5;
    if (head != 0xABul * ((1u << 20) / 4096) || tail != 0) return 6;             // what survives is intact, what is gone reads as zeros
    std::vector<uint8_t> copy(n);
    std::memcpy(copy.data(), m, n);                                               // a bulk copy across the cut
    gip::InputGuard::unwatch(slot);
    if (gip::InputGuard::cut(slot) && gip::InputGuard::watch(m, n) != slot) return 7;   // the slot is free again
    ::munmap(const_cast<uint8_t *>(m), n);
    std::printf("input guard ok\n");
    return 0;
}
