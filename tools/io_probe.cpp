// io_probe.cpp -- what the host side of the CLI pipeline can move on this box: pinned allocation, pread / pwrite
// of page-cached files by thread count, PCIe copies, and whether a mapped file can be registered for DMA.
// Build: g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o tools/io_probe.bin tools/io_probe.cpp -L/opt/rocm/lib -lamdhip64 -lpthread
// Run (GPU box): ./tools/io_probe.bin [dir, default /tmp] [GiB, default 2]
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)

template <typename F>
static double parallel(int threads, size_t n, F &&f) {       // f(begin, end) over [0, n) in `threads` contiguous parts
    const double t0 = now();
    std::vector<std::thread> ts;
    const size_t per = ((n + threads - 1) / threads + 4095) & ~size_t(4095);
    for (int t = 0; t < threads; ++t) ts.emplace_back([&, t] { const size_t b = std::min(n, t * per), e = std::min(n, b + per); if (b < e) f(b, e); });
    for (auto &t : ts) t.join();
    return now() - t0;
}

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t total = size_t(argc > 2 ? atof(argv[2]) * (1 << 30) : 2.0 * (1 << 30));
    const std::string in = dir + "/io_probe.in", out = dir + "/io_probe.out";
    printf("cpus online %ld, affinity-usable %u, dir %s\n", sysconf(_SC_NPROCESSORS_ONLN), std::thread::hardware_concurrency(), dir.c_str());
    double t0 = now();
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    printf("hip init                          %7.3f s\n", now() - t0);
    for (size_t mib : {64, 256, 1024}) {
        void *p = nullptr;
        t0 = now();
        CK(hipHostMalloc(&p, mib << 20, hipHostMallocDefault));
        const double ta = now() - t0;
        t0 = now();
        memset(p, 1, mib << 20);
        const double tm = now() - t0;
        t0 = now();
        CK(hipHostFree(p));
        printf("hipHostMalloc %5zu MiB            %7.3f s (%6.2f GB/s), first memset %6.3f s, free %6.3f s\n", mib, ta, (mib << 20) / ta / 1e9, tm, now() - t0);
    }
    {   // registering ordinary memory instead
        const size_t n = 1u << 30;
        void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
        t0 = now();
        madvise(p, n, MADV_HUGEPAGE);
        memset(p, 1, n);
        const double tm = now() - t0;
        t0 = now();
        const hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault);
        printf("anonymous 1 GiB: touch %6.3f s, hipHostRegister %6.3f s (%s)\n", tm, now() - t0, hipGetErrorString(e));
        if (e == hipSuccess) CK(hipHostUnregister(p));
        munmap(p, n);
    }
    uint8_t *pin = nullptr, *pin2 = nullptr, *dev = nullptr;
    const size_t chunk = 1u << 30;
    CK(hipHostMalloc(reinterpret_cast<void **>(&pin), chunk, hipHostMallocDefault));
    CK(hipHostMalloc(reinterpret_cast<void **>(&pin2), chunk, hipHostMallocDefault));
    CK(hipMalloc(reinterpret_cast<void **>(&dev), chunk));
    for (size_t i = 0; i < chunk; i += 4096) pin[i] = uint8_t(i >> 12), pin2[i] = 1;
    // the input file, written once (page cache)
    {
        const int fd = open(in.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
        t0 = now();
        for (size_t at = 0; at < total; at += chunk) if (pwrite(fd, pin, std::min(chunk, total - at), at) < 0) perror("pwrite");
        printf("write input %4.1f GiB 1 thread      %7.3f s (%6.2f GB/s)\n", total / double(1 << 30), now() - t0, total / (now() - t0) / 1e9);
        close(fd);
    }
    for (int threads : {1, 2, 4, 8, 12, 16}) {
        const int fd = open(in.c_str(), O_RDONLY);
        const size_t n = std::min(chunk, total);
        const double t = parallel(threads, n, [&](size_t b, size_t e) { while (b < e) { ssize_t g = pread(fd, pin + b, e - b, b); if (g <= 0) break; b += g; } });
        printf("pread  1 GiB page cache -> pinned, %2d threads %7.3f s (%6.2f GB/s)\n", threads, t, n / t / 1e9);
        close(fd);
    }
    for (int threads : {1, 4, 8, 16}) {
        const double t = parallel(threads, chunk, [&](size_t b, size_t e) { memcpy(pin2 + b, pin + b, e - b); });
        printf("memcpy 1 GiB pinned -> pinned,     %2d threads %7.3f s (%6.2f GB/s)\n", threads, t, chunk / t / 1e9);
    }
    for (int pass = 0; pass < 2; ++pass)
        for (int threads : {1, 2, 4, 8, 16}) {
            if (pass == 0) unlink(out.c_str());
            const int fd = open(out.c_str(), O_CREAT | O_WRONLY, 0644);
            const double t = parallel(threads, chunk, [&](size_t b, size_t e) { while (b < e) { ssize_t g = pwrite(fd, pin + b, e - b, b); if (g <= 0) break; b += g; } });
            printf("pwrite 1 GiB pinned -> %s file, %2d threads %7.3f s (%6.2f GB/s)\n", pass ? "existing" : "new     ", threads, t, chunk / t / 1e9);
            close(fd);
        }
    {
        t0 = now();
        const int fd = open(out.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
        printf("O_TRUNC of the 1 GiB output       %7.3f s\n", now() - t0);
        close(fd);
        t0 = now();
        const int fd2 = open(out.c_str(), O_CREAT | O_WRONLY, 0644);
        const int rc = posix_fallocate(fd2, 0, chunk);
        printf("posix_fallocate 1 GiB             %7.3f s (rc %d)\n", now() - t0, rc);
        const double t = parallel(8, chunk, [&](size_t b, size_t e) { while (b < e) { ssize_t g = pwrite(fd2, pin + b, e - b, b); if (g <= 0) break; b += g; } });
        printf("pwrite 1 GiB into fallocated file, 8 threads %7.3f s (%6.2f GB/s)\n", t, chunk / t / 1e9);
        close(fd2);
    }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now();
        CK(hipMemcpyAsync(dev, pin, chunk, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        const double th = now() - t0;
        t0 = now();
        CK(hipMemcpyAsync(pin2, dev, chunk, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        printf("H2D 1 GiB pinned %6.2f GB/s, D2H %6.2f GB/s\n", chunk / th / 1e9, chunk / (now() - t0) / 1e9);
    }
    {   // both directions at once on two streams
        hipStream_t s2;
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        uint8_t *dev2 = nullptr;
        CK(hipMalloc(reinterpret_cast<void **>(&dev2), chunk));
        t0 = now();
        CK(hipMemcpyAsync(dev, pin, chunk, hipMemcpyHostToDevice, s));
        CK(hipMemcpyAsync(pin2, dev2, chunk, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s));
        CK(hipStreamSynchronize(s2));
        printf("H2D + D2H of 1 GiB each at once   %7.3f s (%6.2f GB/s per direction)\n", now() - t0, chunk / (now() - t0) / 1e9);
        CK(hipFree(dev2));
    }
    {   // pageable source: what the runtime's own staging does
        std::vector<uint8_t> heap(chunk, 3);
        t0 = now();
        CK(hipMemcpy(dev, heap.data(), chunk, hipMemcpyHostToDevice));
        printf("H2D 1 GiB from pageable memory    %7.3f s (%6.2f GB/s)\n", now() - t0, chunk / (now() - t0) / 1e9);
    }
    {   // the input file mapped and registered: DMA straight out of the page cache?
        const int fd = open(in.c_str(), O_RDONLY);
        const size_t n = std::min<size_t>(256u << 20, total);
        t0 = now();
        void *m = mmap(nullptr, n, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
        const double tm = now() - t0;
        t0 = now();
        hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault);
        const double tr = now() - t0;
        printf("mmap(MAP_POPULATE) 256 MiB of the file %6.3f s, hipHostRegister %6.3f s (%s)\n", tm, tr, hipGetErrorString(e));
        if (e == hipSuccess) {
            t0 = now();
            CK(hipMemcpyAsync(dev, m, n, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            printf("H2D 256 MiB from the registered mapping  %6.3f s (%6.2f GB/s)\n", now() - t0, n / (now() - t0) / 1e9);
            t0 = now();
            CK(hipHostUnregister(m));
            printf("hipHostUnregister                        %6.3f s\n", now() - t0);
        } else {
            (void)hipGetLastError();
        }
        munmap(m, n);
        close(fd);
    }
    {   // the output file mapped (shared, writable) and registered: DMA straight into the page cache?
        const int fd = open(out.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
        const size_t n = 256u << 20;
        if (ftruncate(fd, n) != 0) perror("ftruncate");
        t0 = now();
        void *m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
        const double tm = now() - t0;
        t0 = now();
        hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault);
        const double tr = now() - t0;
        printf("mmap(shared, populate) 256 MiB of a new file %6.3f s, hipHostRegister %6.3f s (%s)\n", tm, tr, hipGetErrorString(e));
        if (e == hipSuccess) {
            t0 = now();
            CK(hipMemcpyAsync(m, dev, n, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            printf("D2H 256 MiB into the registered mapping      %6.3f s (%6.2f GB/s)\n", now() - t0, n / (now() - t0) / 1e9);
            CK(hipHostUnregister(m));
        } else {
            (void)hipGetLastError();
        }
        munmap(m, n);
        close(fd);
    }
    // ---- can the output bypass pwrite's one-writer-per-file limit?  (a) CPU copies into a shared mapping of the new
    //      file, each thread faulting its own pages; (b) the same into a preallocated file; (c) per-thread windows of the
    //      mapping registered for DMA and filled by D2H copies
    for (int prealloc = 0; prealloc < 2; ++prealloc)
        for (int threads : {1, 4, 8, 16}) {
            unlink(out.c_str());
            const int fd = open(out.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
            if (prealloc ? posix_fallocate(fd, 0, chunk) != 0 : ftruncate(fd, chunk) != 0) perror("size");
            t0 = now();
            uint8_t *m = static_cast<uint8_t *>(mmap(nullptr, chunk, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
            const double t = parallel(threads, chunk, [&](size_t b, size_t e) { memcpy(m + b, pin + b, e - b); });
            const double tu0 = now();
            munmap(m, chunk);
            close(fd);
            printf("memcpy 1 GiB pinned -> shared mapping of a %s file, %2d threads %7.3f s (%6.2f GB/s), munmap+close %6.3f s\n",
                   prealloc ? "fallocated" : "truncated ", threads, t, chunk / t / 1e9, now() - tu0);
        }
    for (int threads : {1, 2, 4, 8}) {
        unlink(out.c_str());
        const int fd = open(out.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
        if (posix_fallocate(fd, 0, chunk) != 0) perror("fallocate");
        const size_t win = 64u << 20;
        std::vector<hipStream_t> ss(threads);
        for (auto &x : ss) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        t0 = now();
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; ++t)
            ts.emplace_back([&, t] {
                (void)hipSetDevice(0);
                for (size_t at = size_t(t) * win; at < chunk; at += size_t(threads) * win) {
                    void *m = mmap(nullptr, win, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, at);
                    if (hipHostRegister(m, win, hipHostRegisterDefault) == hipSuccess) {
                        (void)hipMemcpyAsync(m, dev + at, win, hipMemcpyDeviceToHost, ss[t]);
                        (void)hipStreamSynchronize(ss[t]);
                        (void)hipHostUnregister(m);
                    }
                    munmap(m, win);
                }
            });
        for (auto &t : ts) t.join();
        const double t = now() - t0;
        printf("D2H 1 GiB into registered 64 MiB windows of the output mapping, %d threads %7.3f s (%6.2f GB/s)\n", threads, t, chunk / t / 1e9);
        close(fd);
    }
    for (int threads : {1, 4, 8}) {   // O_DIRECT: past the page cache, to whatever device backs the directory
        unlink(out.c_str());
        const int fd = open(out.c_str(), O_CREAT | O_WRONLY | O_TRUNC | O_DIRECT, 0644);
        if (fd < 0) { printf("O_DIRECT open failed: %s\n", strerror(errno)); break; }
        if (posix_fallocate(fd, 0, chunk) != 0) perror("fallocate");
        bool bad = false;
        const double t = parallel(threads, chunk, [&](size_t b, size_t e) {
            while (b < e) { const size_t io = std::min<size_t>(16u << 20, e - b); ssize_t g = pwrite(fd, pin + b, io, b); if (g <= 0) { bad = true; break; } b += g; } });
        printf("pwrite 1 GiB pinned -> O_DIRECT file, %2d threads %7.3f s (%6.2f GB/s)%s\n", threads, t, chunk / t / 1e9, bad ? "  FAILED" : "");
        close(fd);
    }
    {   // is the data really in the file?
        const int fd = open(out.c_str(), O_RDONLY);
        std::vector<uint8_t> head(4096);
        const ssize_t got = pread(fd, head.data(), head.size(), 0);
        std::vector<uint8_t> want(4096);
        CK(hipMemcpy(want.data(), dev, want.size(), hipMemcpyDeviceToHost));
        printf("read back through pread: %s\n", got == 4096 && memcmp(head.data(), want.data(), 4096) == 0 ? "equal" : "DIFFERENT");
        close(fd);
    }
    unlink(in.c_str());
    unlink(out.c_str());
    return 0;
}
