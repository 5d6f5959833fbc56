// gpu_compressor.cpp -- host pipeline of the GPU path.
//
// The reference moves one 8704-byte packet per memcpy and per fwrite, all on one thread
// (/root/reference/src/gpu_compressor.cpp:134-171, 280-340).  Here the file is cut into CHUNKS of
// whole packets (a multiple of 64; by default large enough to fill the chip: 65536 packets = 512 MiB
// of input = 1024 wavefronts) and every chunk goes down a lane:
//
//     input file, MAPPED and registered for DMA: H2D straight out of the page cache (no pread, no staging copy;
//         if the file cannot be mapped or registered: pread into the lane's pinned pieces instead)
//       -> kernels (+ device-side compaction when compressing)
//       -> D2H in PIECES of 64 MiB into the lane's two pinned buffers
//       -> the writer thread: one stream of pwrite()s, each cut into slices on a few helper threads
//
// Each GPU runs kLanesPerDevice lanes at once, each a host thread with its own non-blocking HIP stream,
// device buffers, pinned pieces and status word, so the copies and kernels of one chunk overlap the
// draining of another.  The file is cut into contiguous packet ranges (chunks) in file order, sized so that every lane of
// every device gets one (ensureBuffers: at most ceil(packets / (G * lanes)) packets each, whole wavefronts), and chunk c
// goes to device c mod G: the packet-range sharding of SURVEY.md section 8(e) at CHUNK grain -- every device codes the same
// number of chunks to within one, hence the same bytes to within one chunk -- dealt in file order because the one writer
// consumes in file order (one contiguous range per device would have devices 1..G-1 finish their first chunks and then hold
// them, pinned, until the writer is through all of device 0's range).  What bounds the wall time
// on a page-cached file is the ONE thing that cannot be spread: buffered writes to a single file are
// serialised by the file system (tools/io_probe.cpp: ~9-11 GB/s into a new file whatever the number of
// threads, against 57 GB/s per direction over PCIe and 76+ GB/s of pread), so the pipeline is built to keep
// exactly that writer busy from the first piece to the last.  The only serial quantity besides it is the
// running output offset of the compressor, which the ordered writer carries by itself.
// Decoding works the same way in the other direction; where the packets of a chunk start is known from the
// index trailer (packet_index.hpp) or from a scanner thread that walks the packet headers through the mapping
// (`off += clen`, 4 bytes per packet) ahead of the lanes.  The files produced are byte-identical to those
// of the reference's loop from byte 20 on.
#include "gpu_compressor.hpp"

#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <thread>

#include "file_header.hpp"
#include "gpuar_hip.h"
#include "input_guard.hpp"
#include "packet_index.hpp"

namespace gip {

namespace {

constexpr size_t kPacket = GPUAR_PACKET_BYTES;
constexpr size_t kSlot = GPUAR_SLOT_BYTES;
constexpr size_t kLanesPerDevice = 3;      // chunks in flight per GPU
constexpr size_t kPieceBytes = 64u << 20;  // what one D2H copy / one pwrite moves
constexpr size_t kPiecesPerLane = 2;       // pinned buffers per lane: one being filled while the other is written
constexpr size_t kIoSlice = 16u << 20;     // a pread / pwrite is cut into slices of this size, one helper thread each
constexpr size_t kIoThreads = 4;           // ... at most this many at a time

// GPUAR_TRACE=1: a timeline of the pipeline on stderr (seconds since the first call), for tools/cli_timing.sh
double trace_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void trace(const char *what, double value = -1) {
    static const bool on = std::getenv("GPUAR_TRACE") != nullptr;
    static const double t0 = trace_now();
    static std::mutex lock;
    if (!on) return;
    std::lock_guard<std::mutex> hold(lock);
    if (value >= 0) std::fprintf(stderr, "[gpuar %8.3f s] %s %.3f\n", trace_now() - t0, what, value);
    else std::fprintf(stderr, "[gpuar %8.3f s] %s\n", trace_now() - t0, what);
}

void hip_check(hipError_t e, const char *what) {
    // same message shape as src/gpu_compressor.cpp:189-192
    if (e != hipSuccess) throw std::runtime_error(std::string("Fail to execute kernel code: ") + what + ": " + hipGetErrorString(e));
}
void gpuar_check(int code, const char *what) {
    if (code != GPUAR_OK) throw std::runtime_error(std::string("Fail to execute kernel code: ") + what + ": " + gpuar_hip_error_string(code));
}

// pread / pwrite of a large range, cut into slices that run on a few threads
template <bool kWrite>
void sliced_io(int fd, uint8_t *buf, size_t n, uint64_t at, const char *error) {
    const size_t slices = std::max<size_t>(1, std::min(kIoThreads, (n + kIoSlice - 1) / kIoSlice));
    const size_t per = ((n + slices - 1) / slices + 4095) & ~static_cast<size_t>(4095);
    std::atomic<bool> failed{false};
    auto run = [&](size_t k) {
        size_t done = std::min(n, k * per);
        const size_t end = std::min(n, done + per);
        while (done < end) {
            const ssize_t got = kWrite ? ::pwrite(fd, buf + done, end - done, static_cast<off_t>(at + done))
                                       : ::pread(fd, buf + done, end - done, static_cast<off_t>(at + done));
            if (got <= 0) {
                failed = true;
                return;
            }
            done += static_cast<size_t>(got);
        }
    };
    std::vector<std::thread> helpers;
    for (size_t k = 1; k < slices; ++k) helpers.emplace_back(run, k);
    run(0);
    for (auto &h : helpers) h.join();
    if (failed) throw std::runtime_error(error);
}

// The input file mapped read-only and registered with the HIP runtime, so that hipMemcpyAsync reads the page
// cache directly (tools/io_probe.cpp: 57 GB/s, the same as from hipHostMalloc memory, with no copy on the CPU).
// Registration costs ~15 ms per GiB and is therefore done window by window (256 MiB), by the lane that is about
// to copy from the window, while the other lanes and the writer are already at work -- registering an 8 GiB file
// in one go would hold the whole pipeline up for 0.12 s.  The FIRST registration of a process costs ~35 ms whatever its
// size: the jobs start it on a thread of its own, next to the lanes' buffer allocation (warmFirstWindow).  `data() == nullptr` means the file could not be mapped,
// `require()` returning false that a window could not be registered: callers fall back to pread.
class MappedInput {
  public:
    ~MappedInput() { close(); }
    void open(int fd, size_t bytes) {
        close();
        if (bytes == 0 || std::getenv("GPUAR_NO_MMAP")) return;
        void *m = ::mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) return;
        base = static_cast<const uint8_t *>(m);
        size = bytes;
        guard = InputGuard::watch(base, size);     // a file cut short under the mapping: zeros and a mark instead of SIGBUS (input_guard.hpp)
        (void)::madvise(m, bytes, MADV_SEQUENTIAL);
        state.assign((bytes + kWindow - 1) / kWindow, 0);
        users.assign(state.size(), 0);
        live = 0;
        frontier = 0;
        if (const char *cap = std::getenv("GPUAR_MAX_WINDOWS")) max_live = std::max<size_t>(1, std::strtoul(cap, nullptr, 10));
        // (tests only: time for somebody to cut the file short between the mapping and the first read of it;
        // and says on stderr that it is there, so the test need not guess how long start-up took)
        if (const char *hold = std::getenv("GPUAR_TEST_HOLD_AFTER_MAP_MS")) {
            std::fprintf(stderr, "[gpuar] input mapped\n");
            std::fflush(stderr);
            ::usleep(1000u * static_cast<useconds_t>(std::strtoul(hold, nullptr, 10)));
        }
    }
    // makes [at, at + n) DMA-able and counts the caller as a user of its windows until release(at, n);
    // false: this file cannot be registered (the caller reads it with pread instead, and owes no release)
    bool require(uint64_t at, size_t n) {
        if (!base || n == 0) return base != nullptr;
        std::unique_lock<std::mutex> hold(lock);
        if (refused) return false;
        const size_t first = at / kWindow, last = (at + n - 1) / kWindow;
        // The caller counts as a user of EVERY window of its range before anything below can wait: dropped.wait() lets go of
        // the lock, and a release() on another lane only ever marks windows with no users -- so a window of this range that is
        // already registered cannot be taken away while a later one is waited for (with the count taken after the
        // loop, a chunk straddling a window boundary could be handed a window that was being unregistered).
        for (size_t w = first; w <= last; ++w) ++users[w];
        for (size_t w = first; w <= last; ++w) {
            dropped.wait(hold, [&] { return state[w] != kDropping; });      // (a trim of this very window is on its way out: let it finish)
            if (state[w] == kRegistered) continue;
            const size_t begin = w * kWindow, len = std::min(kWindow, size - begin);
            if (refused || hipHostRegister(const_cast<uint8_t *>(base) + begin, len, hipHostRegisterPortable) != hipSuccess) {
                if (!refused) (void)hipGetLastError();
                refused = true;       // e.g. the memlock / userptr limit: from here on everybody reads with pread
                for (size_t u = first; u <= last; ++u) --users[u];            // (the caller owes no release)
                return false;         // (windows registered so far stay until their users are done)
            }
            state[w] = kRegistered;
            ++live;
        }
        frontier = std::max(frontier, last);
        return true;
    }
    // the copies out of [at, at + n) have completed (the caller synchronised its stream).  Windows nobody uses any more
    // are unregistered, oldest first, once more than `max_live` are registered: the registered (pinned, DMA-mapped)
    // footprint of a job stays at max_live * 256 MiB however large the file is, instead of growing to the whole input.
    // Never the window the furthest lane has reached (the next chunk starts in it: it would be registered again a moment
    // later), and never under the lock: hipHostUnregister may synchronise the device, and every other lane's
    // require() / release() would wait behind it -- the windows to drop are marked under the lock and let go
    // after it.
    void release(uint64_t at, size_t n) {
        if (!base || n == 0) return;
        std::vector<size_t> drop;
        {
            std::lock_guard<std::mutex> hold(lock);
            for (size_t w = at / kWindow; w <= (at + n - 1) / kWindow; ++w)
                if (users[w]) --users[w];
            for (size_t w = 0; w < state.size() && w < frontier && live > max_live; ++w)
                if (state[w] == kRegistered && !users[w]) {      // (no users AT MARKING TIME, and require() counts before it waits)
                    state[w] = kDropping;
                    --live;
                    drop.push_back(w);
                }
        }
        if (drop.empty()) return;
        const double t0 = trace_now();
        for (size_t w : drop) (void)hipHostUnregister(const_cast<uint8_t *>(base) + w * kWindow);
        {
            std::lock_guard<std::mutex> hold(lock);
            for (size_t w : drop) state[w] = kFree;
            trimmed += drop.size();
            trim_seconds += trace_now() - t0;
        }
        dropped.notify_all();
    }
    void close() {
        if (warmer.joinable()) warmer.join();
        if (!base) return;
        for (size_t w = 0; w < state.size(); ++w)
            if (state[w] == kRegistered) (void)hipHostUnregister(const_cast<uint8_t *>(base) + w * kWindow);
        InputGuard::unwatch(guard);
        guard = -1;
        ::munmap(const_cast<uint8_t *>(base), size);
        base = nullptr;
        state.clear();
        users.clear();
        live = 0;
        refused = false;
    }
    // registers the first window on a thread of its own (the first registration of a process costs ~35 ms, as long as a lane
    // takes to allocate its buffers: the two then run side by side); close() waits for it
    void warmFirstWindow(int device) {
        if (!base) return;
        warmer = std::thread([this, device] {
            const size_t n = std::min(kWindow, size);
            if (hipSetDevice(device) == hipSuccess && require(0, n)) release(0, n);
        });
    }
    const uint8_t *data() const { return base; }
    // Is the file still what was mapped?  false once an access to the mapping has faulted (somebody cut the file short:
    // what was read behind the new end are zeros, input_guard.hpp) or the file's length is no longer the mapped one.
    // The reference's fread() would have come back short (src/gpu_compressor.cpp:146-150); callers throw its message.
    bool intact(int fd) const {
        if (!base) return true;
        if (InputGuard::cut(guard)) return false;
        struct stat st;
        return ::fstat(fd, &st) == 0 && static_cast<uint64_t>(st.st_size) >= size;
    }
    // a copy must not straddle two registrations: the end of the window `at` lies in
    static size_t windowEnd(uint64_t at) { return (at / kWindow + 1) * kWindow; }
    size_t windowsTrimmed() {
        std::lock_guard<std::mutex> hold(lock);
        return trimmed;
    }
    double trimSeconds() {
        std::lock_guard<std::mutex> hold(lock);
        return trim_seconds;
    }
    bool wasRefused() {
        std::lock_guard<std::mutex> hold(lock);
        return refused;
    }

  private:
    static constexpr size_t kWindow = 256u << 20;
    static constexpr uint8_t kFree = 0, kRegistered = 1, kDropping = 2;
    std::thread warmer;
    const uint8_t *base = nullptr;
    size_t size = 0;
    int guard = -1;
    std::mutex lock;
    std::condition_variable dropped;
    std::vector<uint8_t> state;      // per window: kFree / kRegistered / kDropping (being unregistered by a release())
    std::vector<uint32_t> users;     // per window: chunks whose copies out of it may still be in flight
    size_t live = 0;                 // registered windows
    size_t frontier = 0;             // the highest window any lane has required so far: never trimmed
    size_t max_live = 16;            // ... of which at most this many are kept once nobody uses them (4 GiB; GPUAR_MAX_WINDOWS)
    size_t trimmed = 0;
    double trim_seconds = 0;         // spent inside hipHostUnregister by the trims (GPUAR_TRACE)
    bool refused = false;
};

// A pinned buffer handed from a lane to the writer and back.
struct Piece {
    uint8_t *data = nullptr;
    size_t bytes = 0;
    uint64_t at = 0;           // file offset (unordered mode); ignored when the writer keeps the running offset
    bool last_of_chunk = false;
    size_t plain_bytes = 0;    // uncompressed bytes this piece completes (progress line)
    std::function<void()> give_back;
};

// The writer: ONE thread that turns pieces into pwrite()s.
//   ordered (compress): chunk c is written behind chunk c-1 -- where it starts is the sum of what came before,
//       the pipeline's one serial quantity, which this thread simply carries; pieces of a chunk arrive in order.
//   unordered (decompress): every piece knows its own file offset; pieces are written as they arrive.
class Writer {
  public:
    Writer(int fd, bool ordered, uint64_t start, const char *error) : fd(fd), ordered(ordered), at(start), error(error) {
        worker = std::thread([this] { run(); });
    }
    ~Writer() {
        abort();
        if (worker.joinable()) worker.join();
    }
    void submit(size_t chunk, Piece p) {
        std::lock_guard<std::mutex> hold(lock);
        if (aborted) {
            if (p.give_back) p.give_back();
            return;
        }
        queues[ordered ? chunk : 0].push_back(std::move(p));
        wake.notify_all();
    }
    // no more pieces will come; waits for everything submitted to be written and rethrows the writer's failure
    uint64_t finish(size_t n_chunks) {
        {
            std::lock_guard<std::mutex> hold(lock);
            expect_chunks = n_chunks;
            closing = true;
            wake.notify_all();
        }
        if (worker.joinable()) worker.join();
        if (failure) std::rethrow_exception(failure);
        return at;
    }
    void abort() {
        std::lock_guard<std::mutex> hold(lock);
        aborted = true;
        wake.notify_all();
    }
    std::function<void(size_t)> on_progress;      // called with the plain bytes of every piece written
    std::function<void()> on_failure;             // the writer itself failed (disk full ...): lets the lanes go

  private:
    void run() {
        try {
            for (;;) {
                Piece p;
                {
                    std::unique_lock<std::mutex> hold(lock);
                    const size_t key = 0;
                    wake.wait(hold, [&] {
                        if (aborted) return true;
                        auto it = queues.find(ordered ? next_chunk : key);
                        if (it != queues.end() && !it->second.empty()) return true;
                        return closing && (!ordered || next_chunk >= expect_chunks);
                    });
                    if (aborted) break;
                    auto it = queues.find(ordered ? next_chunk : key);
                    if (it == queues.end() || it->second.empty()) break;          // closing and drained
                    p = std::move(it->second.front());
                    it->second.pop_front();
                    if (ordered && p.last_of_chunk) {
                        queues.erase(it);
                        ++next_chunk;
                    }
                }
                try {
                    const double w0 = trace_now();
                    if (first_write < 0) first_write = w0, trace("writer: first piece");
                    if (p.bytes) sliced_io<true>(fd, p.data, p.bytes, ordered ? at : p.at, error);
                    busy_s += trace_now() - w0;
                    bytes_written += p.bytes;
                } catch (...) {
                    if (p.give_back) p.give_back();
                    throw;
                }
                if (ordered) at += p.bytes;
                if (p.give_back) p.give_back();
                if (on_progress) on_progress(p.plain_bytes);
            }
        } catch (...) {
            failure = std::current_exception();
        }
        if (first_write >= 0) {
            trace("writer: done; seconds inside pwrite:", busy_s);
            trace("writer: GB/s while writing:", bytes_written / std::max(busy_s, 1e-9) / 1e9);
            trace("writer: seconds between first piece and last:", trace_now() - first_write);
        }
        // whatever is still queued goes back to its lane (which may be waiting for it)
        {
            std::lock_guard<std::mutex> hold(lock);
            aborted = true;
            for (auto &q : queues)
                for (auto &p : q.second)
                    if (p.give_back) p.give_back();
            queues.clear();
        }
        if (failure && on_failure) on_failure();
    }

    int fd;
    bool ordered;
    uint64_t at;
    const char *error;
    std::mutex lock;
    std::condition_variable wake;
    std::map<size_t, std::deque<Piece>> queues;
    size_t next_chunk = 0, expect_chunks = 0;
    bool closing = false, aborted = false;
    std::exception_ptr failure;
    std::thread worker;
    double first_write = -1, busy_s = 0, bytes_written = 0;
};

}  // namespace

// Keeps the first failure of any lane and tells the others to stop.
struct GPUCompressor::Failure {
    std::mutex lock;
    std::exception_ptr first;
    std::atomic<bool> stop{false};
    void set(std::exception_ptr e) {
        std::lock_guard<std::mutex> hold(lock);
        if (!first) first = e;
        stop = true;
    }
};

// Everything one lane needs for one chunk of `cap` packets.
struct GPUCompressor::DeviceBuffers {
    int device = 0;
    size_t cap = 0;                 // packets
    size_t piece_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    uint8_t *d_plain = nullptr;     // cap * 8192
    uint8_t *d_slots = nullptr;     // cap * 8704 (compress only)
    uint8_t *d_stream = nullptr;    // cap * 8704 (+16)
    uint64_t *d_offsets = nullptr;  // cap + 1, then this lane's status word
    uint64_t *h_offsets = nullptr;  // pinned, cap + 2: the word behind the offsets receives the status word
    uint32_t *d_status = nullptr;   // this lane's own status word (device): its launches report here, nobody else's do
    hipEvent_t epoch = nullptr;     // the device's common time base (owned by the GPUCompressor)
    std::vector<std::pair<float, float>> busy;   // [begin, end) of every chunk's kernels, ms since `epoch`
    // the lane's pinned pieces and which of them are free
    uint8_t *pieces[kPiecesPerLane] = {};
    std::mutex piece_lock;
    std::condition_variable piece_free;
    bool piece_busy[kPiecesPerLane] = {};
    bool aborted = false;

    void allocate(int dev, size_t packets, bool compressing) {
        device = dev;
        cap = packets;
        piece_bytes = std::min(kPieceBytes, cap * kSlot + 16);
        hip_check(hipSetDevice(device), "hipSetDevice");
        // non-blocking: a lane must never wait for another lane's copies or kernels through the NULL stream
        hip_check(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipEventCreate(&t0), "hipEventCreate");
        hip_check(hipEventCreate(&t1), "hipEventCreate");
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_plain), cap * kPacket), "hipMalloc");
        if (compressing) hip_check(hipMalloc(reinterpret_cast<void **>(&d_slots), cap * kSlot), "hipMalloc");
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_stream), cap * kSlot + 16), "hipMalloc");
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_offsets), (cap + 2) * sizeof(uint64_t)), "hipMalloc");
        d_status = reinterpret_cast<uint32_t *>(d_offsets + cap + 1);
        hip_check(hipHostMalloc(reinterpret_cast<void **>(&h_offsets), (cap + 2) * sizeof(uint64_t), hipHostMallocDefault), "hipHostMalloc");
        // (the pinned pieces are allocated when the first one is asked for: by then the lane's first copy and kernels are
        // on their way, and pinning 128 MiB takes as long as they do)
        has_slots = compressing;
    }
    bool has_slots = false;
    void release() {
        if (!cap) return;
        (void)hipSetDevice(device);
        (void)hipFree(d_plain);
        if (d_slots) (void)hipFree(d_slots);
        (void)hipFree(d_stream);
        (void)hipFree(d_offsets);
        (void)hipHostFree(h_offsets);
        for (auto &p : pieces) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
        }
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
        (void)hipStreamDestroy(stream);
        d_slots = nullptr;
        cap = 0;
    }

    // a free pinned piece (blocks while the writer still holds both)
    size_t takePiece() {
        std::unique_lock<std::mutex> hold(piece_lock);
        for (;;) {
            if (aborted) throw std::runtime_error("pipeline stopped");
            for (size_t k = 0; k < kPiecesPerLane; ++k)
                if (!piece_busy[k]) {
                    piece_busy[k] = true;
                    if (!pieces[k]) {
                        hold.unlock();
                        const hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&pieces[k]), piece_bytes, hipHostMallocDefault);
                        if (e != hipSuccess) {
                            givePiece(k);
                            hip_check(e, "hipHostMalloc");
                        }
                    }
                    return k;
                }
            piece_free.wait(hold);
        }
    }
    void givePiece(size_t k) {
        std::lock_guard<std::mutex> hold(piece_lock);
        piece_busy[k] = false;
        piece_free.notify_all();
    }
    void abortWaits() {
        std::lock_guard<std::mutex> hold(piece_lock);
        aborted = true;
        piece_free.notify_all();
    }
    void resetPieces() {
        std::lock_guard<std::mutex> hold(piece_lock);
        aborted = false;
        for (auto &b : piece_busy) b = false;
    }

    // where this chunk's kernels ran on the device's clock
    void noteBusy() {
        float begin = 0, end = 0;
        hip_check(hipEventElapsedTime(&begin, epoch, t0), "event");
        hip_check(hipEventElapsedTime(&end, epoch, t1), "event");
        busy.emplace_back(begin, end);
    }

    // `n` bytes of the input file at `at` -> dst (device): straight from the mapping when there is one, else through
    // the lane's pinned pieces
    // Returns true when the copies were queued straight out of the mapping: the caller then owes in.release(at, n)
    // once it has synchronised this lane's stream.
    bool upload(uint8_t *dst, MappedInput &in, int fd, uint64_t at, size_t n, const char *error) {
        if (in.data() && in.require(at, n)) {
            for (size_t done = 0; done < n;) {           // one copy per registered window
                const size_t part = std::min<uint64_t>(n - done, MappedInput::windowEnd(at + done) - (at + done));
                hip_check(hipMemcpyAsync(dst + done, in.data() + at + done, part, hipMemcpyHostToDevice, stream), "H2D");
                done += part;
            }
            return true;
        }
        for (size_t done = 0; done < n;) {
            const size_t k = takePiece();
            const size_t part = std::min(piece_bytes, n - done);
            try {
                sliced_io<false>(fd, pieces[k], part, at + done, error);
                hip_check(hipMemcpyAsync(dst + done, pieces[k], part, hipMemcpyHostToDevice, stream), "H2D");
                hip_check(hipStreamSynchronize(stream), "sync");
            } catch (...) {
                givePiece(k);
                throw;
            }
            givePiece(k);
            done += part;
        }
        return false;
    }

    // src[0..n) (device) -> the writer, piece by piece; `at`: file offset of the first byte (unordered writer).
    // `closes_chunk`: the last piece of these is the chunk's last (the ordered writer then moves on to chunk + 1).
    void drain(Writer &writer, size_t chunk, const uint8_t *src, size_t n, uint64_t at, size_t plain_bytes, bool closes_chunk = true) {
        size_t done = 0;
        do {
            const size_t k = takePiece();
            const size_t part = std::min(piece_bytes, n - done);
            try {
                if (part) {
                    hip_check(hipMemcpyAsync(pieces[k], src + done, part, hipMemcpyDeviceToHost, stream), "D2H");
                    hip_check(hipStreamSynchronize(stream), "sync");
                }
            } catch (...) {
                givePiece(k);
                throw;
            }
            Piece p;
            p.data = pieces[k];
            p.bytes = part;
            p.at = at + done;
            done += part;
            p.last_of_chunk = closes_chunk && done >= n;
            p.plain_bytes = done >= n ? plain_bytes : 0;
            p.give_back = [this, k] { givePiece(k); };
            writer.submit(chunk, std::move(p));
        } while (done < n);
    }

    // d_plain[0..n_plain) -> d_stream[0..n_stream), h_offsets[0..n_packets]; returns n_stream and, through `flags`,
    // what THIS chunk's launches reported (GPUAR_STATUS_*): the status word travels back with the offsets, on the
    // lane's own stream -- no device-wide synchronisation, no flag shared with another lane
    // `mode`: GPUAR_MODE_* -- which encode kernel (the caller knows whether this launch has the chip to itself)
    size_t encodeChunk(size_t n_plain, uint32_t &flags, int mode) {
        const size_t n_packets = (n_plain + kPacket - 1) / kPacket;
        hip_check(hipMemsetAsync(d_status, 0, sizeof(uint32_t), stream), "memset");
        hip_check(hipEventRecord(t0, stream), "event");
        gpuar_check(gpuar_hip_encode_mode(d_plain, n_plain, d_slots, d_status, stream, mode), "gpuar_hip_encode_mode");
        gpuar_check(gpuar_hip_compact(d_slots, n_packets, d_stream, d_offsets, stream), "gpuar_hip_compact");
        hip_check(hipEventRecord(t1, stream), "event");
        hip_check(hipMemcpyAsync(h_offsets, d_offsets, (n_packets + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipMemcpyAsync(h_offsets + n_packets + 1, d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipStreamSynchronize(stream), "sync");
        flags = *reinterpret_cast<const uint32_t *>(h_offsets + n_packets + 1);
        noteBusy();
        return static_cast<size_t>(h_offsets[n_packets]);
    }

    // d_stream[0..n_stream) with h_offsets[0..n_packets] -> d_plain[0..n_packets*8192); returns this chunk's status flags
    uint32_t decodeChunk(size_t n_packets) {
        hip_check(hipMemsetAsync(d_status, 0, sizeof(uint32_t), stream), "memset");
        hip_check(hipMemcpyAsync(d_offsets, h_offsets, (n_packets + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream), "H2D");
        hip_check(hipEventRecord(t0, stream), "event");
        gpuar_check(gpuar_hip_decode_stream(d_stream, d_offsets, n_packets, d_plain, d_status, stream), "gpuar_hip_decode_stream");
        hip_check(hipEventRecord(t1, stream), "event");
        hip_check(hipMemcpyAsync(h_offsets + n_packets + 1, d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipStreamSynchronize(stream), "sync");
        noteBusy();
        return *reinterpret_cast<const uint32_t *>(h_offsets + n_packets + 1);
    }
};

GPUCompressor::GPUCompressor() {
    trace("GPUCompressor: constructing (the HIP runtime starts here)");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw std::runtime_error("No HIP device found (use --host to run the codec on the CPU)");
    devices.push_back(0);
    trace("GPUCompressor: devices counted");
    initConstantRange();     // kept for parity with src/gpu_compressor.cpp:19; a no-op for these kernels
}

GPUCompressor::~GPUCompressor() { releaseBuffers(); }

void GPUCompressor::releaseBuffers() {
    for (DeviceBuffers *b : buffers) {
        b->release();
        delete b;
    }
    buffers.clear();
    for (size_t g = 0; g < epochs.size(); ++g)
        if (epochs[g]) {
            (void)hipSetDevice(devices[g]);
            (void)hipEventDestroy(static_cast<hipEvent_t>(epochs[g]));
        }
    epochs.clear();
}

// kLanesPerDevice buffer sets per device, sized for the job at hand: a file of `total_packets` is cut
// into chunks of at most batchPackets packets, small enough that every lane of every device gets one
// (whole wavefronts: multiples of 64 packets).  A lane's buffers are allocated when it first gets a
// chunk -- pinned allocations are what a short run of the CLI spends most of its time on.
void GPUCompressor::ensureBuffers(size_t total_packets, bool compressing) {
    const size_t G = devices.size();
    const size_t lanes = G * kLanesPerDevice;
    const size_t share = ((total_packets + lanes - 1) / lanes + 63) / 64 * 64;
    const size_t cap = std::max<size_t>(64, std::min(batchPackets, share));
    bool have = buffers.size() == lanes && !buffers.empty() && chunkPackets == cap;
    if (have && compressing)
        for (DeviceBuffers *b : buffers) have = have && (!b->cap || b->has_slots);
    if (have) {
        for (DeviceBuffers *b : buffers) b->resetPieces();
        return;
    }
    releaseBuffers();
    chunkPackets = cap;
    for (size_t l = 0; l < lanes; ++l) {
        buffers.push_back(new DeviceBuffers());
        buffers.back()->device = devices[l / kLanesPerDevice];
    }
}

void GPUCompressor::chooseDevice(const int id) {
    int count = 0;
    hip_check(hipGetDeviceCount(&count), "hipGetDeviceCount");
    if (id < 0 || id >= count) throw std::runtime_error("No such HIP device: " + std::to_string(id));
    releaseBuffers();
    devices.assign(1, id);
}

void GPUCompressor::useDevices(const int n) {
    int count = 0;
    hip_check(hipGetDeviceCount(&count), "hipGetDeviceCount");
    // GPUAR_OVERSUBSCRIBE_DEVICES=1 lets logical device d run on physical device d % count, so the
    // sharding / ordered-concatenation path can be exercised on a box with fewer GPUs (tests only)
    const char *over = std::getenv("GPUAR_OVERSUBSCRIBE_DEVICES");
    const bool oversubscribe = over && over[0] == '1';
    if (n < 1 || (n > count && !oversubscribe))
        throw std::runtime_error("Asked for " + std::to_string(n) + " GPUs, " + std::to_string(count) + " visible");
    releaseBuffers();
    devices.clear();
    for (int d = 0; d < n; ++d) devices.push_back(d % count);
}

// How many packets chunk c holds.  The first chunks of a job are small and double from one to the next (64 MiB of input,
// 128, 256 ...) until they reach the full chunk size: the writer gets its first piece a few milliseconds after the
// start instead of after the first 512 MiB have crossed PCIe three lanes abreast, and it is the writer that bounds the
// wall time.  (Chunks stay multiples of 64 packets: whole wavefronts.)
static size_t rampPackets(size_t c, size_t full) {
    constexpr size_t kFirst = 8192;
    if (c >= 8 || full <= kFirst) return full;
    return std::min(full, kFirst << c);
}

// Runs `work(lane, buffers, failure)` on every lane (lanes g*kLanesPerDevice .. belong to device g) and
// rethrows the first failure.  `work` pulls chunk numbers until there are none left; `on_failure` runs on the
// failing lane's thread, so that whoever the other lanes are waiting for lets them go.
template <typename Work, typename OnFailure>
void GPUCompressor::runLanes(Work &&work, OnFailure &&on_failure) {
    Failure failure;
    std::vector<std::thread> threads;
    for (size_t l = 0; l < buffers.size(); ++l)
        threads.emplace_back([&, l] {
            try {
                work(l, *buffers[l], failure);
            } catch (...) {
                failure.set(std::current_exception());      // (what a lane woken by an abort throws comes second: never the cause)
                for (DeviceBuffers *b : buffers) b->abortWaits();
                on_failure();
            }
        });
    for (auto &t : threads) t.join();
    if (failure.first) {
        // A lane that failed may have left copies out of the mapped input queued on its stream; the caller's unwinding
        // unregisters and unmaps that file next.  Nothing may still be reading it: drain every lane's stream first
        // (results ignored -- the failure that is about to be reported is the first one).
        for (DeviceBuffers *b : buffers)
            if (b->cap && b->stream) {
                (void)hipSetDevice(b->device);
                (void)hipStreamSynchronize(b->stream);
            }
        std::rethrow_exception(failure.first);
    }
}

CompressionInfo GPUCompressor::compress(ProgressMonitor *monitor) {
    CompressionInfo info;
    monitor->reset();
    process_timer.reset();
    io_timer.reset();
    io_timer.start();
    openFiles(false);
    const int out_fd = fileno(saveFile);
    try {
        info.uncompressedFileSize = getFileSize(openFile);
        const size_t total_packets = (info.uncompressedFileSize + kPacket - 1) / kPacket;
        ensureBuffers(total_packets, true);
        const size_t G = devices.size();
        // contiguous packet ranges in file order: where chunk c starts and how long it is
        std::vector<uint64_t> chunk_at;
        for (uint64_t at = 0; at < info.uncompressedFileSize; at += rampPackets(chunk_at.size() - 1, chunkPackets) * kPacket) chunk_at.push_back(at);
        const size_t n_chunks = chunk_at.size();
        chunk_at.push_back(info.uncompressedFileSize);
        // The latency-mode encode kernel is faster only for a launch that has the chip to itself (include/gpuar_hip.h):
        // a job of one chunk.  With several chunks the lanes of a device keep three launches in flight (the ramped first
        // chunks are small enough for GPUAR_MODE_AUTO to pick the latency kernel), so the pipeline names the kernel itself.
        const int encode_mode = n_chunks == 1 ? GPUAR_MODE_AUTO : GPUAR_MODE_THROUGHPUT;
        const int in_fd = fileno(openFile);
        hip_check(hipSetDevice(devices[0]), "hipSetDevice");
        trace("compress: files open, device set");
        MappedInput mapped;
        mapped.open(in_fd, info.uncompressedFileSize);
        trace(mapped.data() ? "compress: input mapped" : "compress: input NOT mapped (pread path)");
        mapped.warmFirstWindow(devices[0]);
        // room for the worst case up front: writing into allocated blocks is faster than growing the file, and the real
        // length is set at the end (best effort: a file system without fallocate just grows the file as it goes)
        // (the fallocate system call, not posix_fallocate: where the file system cannot do it, glibc's stand-in would write zeros)
        (void)::fallocate(out_fd, 0, 0, static_cast<off_t>(FileHeader::HEADER_LENGTH + total_packets * kSlot));
        std::vector<std::vector<uint16_t>> chunk_clens(writeIndex ? n_chunks : 0);     // for the optional index trailer
        std::vector<std::atomic<size_t>> next_of_device(G);
        for (auto &n : next_of_device) n = 0;
        std::vector<std::atomic<uint64_t>> device_bytes(G), device_chunks(G);      // what each device coded (GPUAR_TRACE)
        for (size_t g = 0; g < G; ++g) device_bytes[g] = 0, device_chunks[g] = 0;
        std::mutex progress_lock;
        {
            Writer writer(out_fd, true, FileHeader::HEADER_LENGTH, "Write compressed data to output file failed");
            writer.on_progress = [&](size_t plain) {
                if (!plain) return;
                std::lock_guard<std::mutex> hold(progress_lock);
                info.processedUncompressedSize += plain;
                monitor->updateProgress(&info);
            };
            writer.on_failure = [&] {
                for (DeviceBuffers *b : buffers) b->abortWaits();
            };
            std::exception_ptr lane_failure;
            try {
                runLanes(
                    [&](size_t lane, DeviceBuffers &b, Failure &failure) {
                        const size_t g = lane / kLanesPerDevice;
                        for (;;) {
                            // contiguous packet ranges in file order, dealt round-robin: chunk c belongs to device c mod G
                            const size_t c = g + G * next_of_device[g].fetch_add(1);
                            if (c >= n_chunks || failure.stop) break;
                            if (c == 0) trace("compress: lane of chunk 0 starts");
                            if (!b.cap) b.allocate(b.device, chunkPackets, true);
                            if (c == 0) trace("compress: its buffers allocated");
                            hip_check(hipSetDevice(b.device), "hipSetDevice");
                            b.epoch = static_cast<hipEvent_t>(epochOf(g));
                            const uint64_t at = chunk_at[c];
                            const size_t n_plain = static_cast<size_t>(chunk_at[c + 1] - at);
                            const bool from_mapping = b.upload(b.d_plain, mapped, in_fd, at, n_plain, "Read input file failed");
                            if (c == 0) trace("compress: its input on its way (window registered, copy queued)");
                            uint32_t flags = 0;
                            const size_t n_stream = b.encodeChunk(n_plain, flags, encode_mode);     // (synchronises the lane's stream)
                            if (from_mapping) mapped.release(at, n_plain);
                            // the file was cut short (or replaced) under the mapping: what the reference's fread() reports
                            // (src/gpu_compressor.cpp:146-150)
                            if (!mapped.intact(in_fd)) throw std::runtime_error("Read input file failed");
                            if (c < 3) trace("compress: kernels of an early chunk done, chunk", static_cast<double>(c));
                            if (flags & GPUAR_STATUS_SLOT_OVERFLOW)
                                throw std::runtime_error("a packet outgrew its 8704-byte slot (input bytes " + std::to_string(at) + " .. " +
                                                         std::to_string(at + n_plain) + ")");
                            const size_t n_packets = (n_plain + kPacket - 1) / kPacket;
                            if (writeIndex) {
                                chunk_clens[c].resize(n_packets);
                                for (size_t p = 0; p < n_packets; ++p) chunk_clens[c][p] = static_cast<uint16_t>(b.h_offsets[p + 1] - b.h_offsets[p]);
                            }
                            device_bytes[g] += n_plain;
                            device_chunks[g] += 1;
                            b.drain(writer, c, b.d_stream, n_stream, 0, n_plain);     // the ordered writer knows where chunk c goes
                        }
                    },
                    [&] { writer.abort(); });
            } catch (...) {
                lane_failure = std::current_exception();
            }
            if (lane_failure) {
                // the writer's own failure (disk full) is the cause when it has one: the lanes only saw "pipeline stopped"
                writer.abort();
                (void)writer.finish(0);      // rethrows the writer's own failure when it has one: that is the cause then
                std::rethrow_exception(lane_failure);
            }
            trace("compress: lanes done");
            trace(mapped.wasRefused() ? "compress: a window could not be registered: the rest was read with pread" : "compress: every window registered");
            trace("compress: input windows unregistered while the job ran:", static_cast<double>(mapped.windowsTrimmed()));
            trace("compress: seconds inside hipHostUnregister for that:", mapped.trimSeconds());
            for (size_t g = 0; g < G; ++g)
                trace(("compress: device " + std::to_string(g) + " coded " + std::to_string(device_bytes[g].load()) + " bytes in " +
                       std::to_string(device_chunks[g].load()) + " chunks of at most " + std::to_string(chunkPackets * kPacket) + " bytes").c_str());
            info.compressedFileSize = static_cast<size_t>(writer.finish(n_chunks));
        }
        uint64_t file_end = info.compressedFileSize;
        if (writeIndex) {
            std::vector<uint16_t> all;
            for (const auto &v : chunk_clens) all.insert(all.end(), v.begin(), v.end());
            if (std::fseek(saveFile, static_cast<long>(info.compressedFileSize), SEEK_SET) != 0) throw std::runtime_error("Seek file failed");
            PacketIndex::write(saveFile, all);
            if (std::fflush(saveFile) != 0) throw std::runtime_error("Write packet index failed");
            file_end = static_cast<uint64_t>(std::ftell(saveFile));
        }
        FileHeader header;
        header.setCompressedFileSize(info.compressedFileSize);
        header.setUncompressedFileSize(info.uncompressedFileSize);
        if (::pwrite(out_fd, header.getData(), FileHeader::HEADER_LENGTH, 0) != FileHeader::HEADER_LENGTH)
            throw std::runtime_error("Write data to file failed");
        if (::ftruncate(out_fd, static_cast<off_t>(file_end)) != 0) throw std::runtime_error("Write data to file failed");
        trace("compress: header written, length set");
        mapped.close();
        closeFiles();
        trace("compress: files closed");
    } catch (...) {
        if (::ftruncate(out_fd, 0) != 0) {}      // a failed job leaves an empty file, not a half-written one
        closeFiles();
        throw;
    }
    io_timer.stop();
    finishTimes(info);
    return info;
}

// "Compute time" = kernels + their sync, as src/gpu_compressor.cpp:184-194.  The lanes of a device run their
// kernels at the same time, so their spans overlap (and each span includes time queued behind another lane's
// kernels): what is reported is the time the device had at least one chunk's kernels in flight -- the UNION of
// the lanes' [first kernel submitted, last kernel done) intervals on the device's own clock -- of the busiest
// device; the rest of the wall time is I/O.
void GPUCompressor::finishTimes(CompressionInfo &info) {
    double busiest = 0;
    for (size_t g = 0; g < devices.size(); ++g) {
        std::vector<std::pair<float, float>> spans;
        for (size_t l = g * kLanesPerDevice; l < (g + 1) * kLanesPerDevice && l < buffers.size(); ++l) {
            spans.insert(spans.end(), buffers[l]->busy.begin(), buffers[l]->busy.end());
            buffers[l]->busy.clear();
        }
        std::sort(spans.begin(), spans.end());
        double total = 0;
        float open_begin = 0, open_end = -1;
        for (const auto &sp : spans) {
            if (open_end < open_begin || sp.first > open_end) {      // a gap: close the run so far
                if (open_end >= open_begin) total += open_end - open_begin;
                open_begin = sp.first;
                open_end = sp.second;
            } else {
                open_end = std::max(open_end, sp.second);
            }
        }
        if (open_end >= open_begin) total += open_end - open_begin;
        busiest = std::max(busiest, total);
    }
    info.processTime = busiest;
    info.ioTime = std::max(0.0, io_timer.value() - info.processTime);
}

// One event per device that every lane of the device measures its kernel intervals against.
void *GPUCompressor::epochOf(size_t g) {
    std::lock_guard<std::mutex> hold(epochLock);
    if (epochs.size() < devices.size()) epochs.resize(devices.size(), nullptr);
    if (!epochs[g]) {
        hipEvent_t e = nullptr;
        hip_check(hipSetDevice(devices[g]), "hipSetDevice");
        hip_check(hipEventCreate(&e), "hipEventCreate");
        hip_check(hipEventRecord(e, nullptr), "event");
        hip_check(hipEventSynchronize(e), "event");
        epochs[g] = e;
    }
    return epochs[g];
}

namespace {

// Where the packets of each chunk sit in the stream.  Filled by the index trailer in one go, or by
// the scanner thread as it walks the headers; lanes wait for the chunk they are about to take.
struct ChunkMap {
    struct Chunk {
        uint64_t begin = 0, end = 0;     // file offsets of the chunk's first byte / one past its last
        size_t n_packets = 0;
    };
    std::mutex lock;
    std::condition_variable more;
    std::vector<Chunk> chunks;
    bool complete = false;
    std::exception_ptr failure;

    void push(const Chunk &c) {
        std::lock_guard<std::mutex> hold(lock);
        chunks.push_back(c);
        more.notify_all();
    }
    void finish(std::exception_ptr e = nullptr) {
        std::lock_guard<std::mutex> hold(lock);
        complete = true;
        failure = e;
        more.notify_all();
    }
    // false: there is no chunk c
    bool get(size_t c, Chunk &out) {
        std::unique_lock<std::mutex> hold(lock);
        more.wait(hold, [&] { return c < chunks.size() || complete; });
        if (c < chunks.size()) {
            out = chunks[c];
            return true;
        }
        if (failure) std::rethrow_exception(failure);
        return false;
    }
};

// The running total of what the chunks before chunk c decode to (a stream may hold short packets in its middle:
// the output offset of a chunk is then not c * chunk bytes).  take(c, n) blocks until chunks 0..c-1 have called and
// returns the total before chunk c's own n bytes; it is called before the chunk's kernels, so nobody waits for it long.
class OrderedOffsets {
  public:
    uint64_t take(size_t chunk, uint64_t bytes) {
        std::unique_lock<std::mutex> hold(lock);
        turn.wait(hold, [&] { return aborted || next == chunk; });
        if (aborted) throw std::runtime_error("pipeline stopped");
        const uint64_t mine = total;
        total += bytes;
        ++next;
        turn.notify_all();
        return mine;
    }
    void abort() {
        std::lock_guard<std::mutex> hold(lock);
        aborted = true;
        turn.notify_all();
    }
    uint64_t sum() {
        std::lock_guard<std::mutex> hold(lock);
        return total;
    }

  private:
    std::mutex lock;
    std::condition_variable turn;
    size_t next = 0;
    uint64_t total = 0;
    bool aborted = false;
};

}  // namespace

CompressionInfo GPUCompressor::decompress(ProgressMonitor *monitor) {
    CompressionInfo info;
    monitor->reset();
    process_timer.reset();
    io_timer.reset();
    io_timer.start();
    openFiles(false);
    const int out_fd = fileno(saveFile);
    try {
        FileHeader header;
        const size_t fileSize = getFileSize(openFile);
        if (std::fread(header.getData(), FileHeader::HEADER_LENGTH, 1, openFile) != 1 || !header.checkHeaderVersion())
            throw std::runtime_error("Incorrect file format");
        info = header.getInfo(fileSize);
        const size_t G = devices.size();
        const uint64_t stream_end = streamEnd(info, fileSize);
        // How many packets to size the chunks for.  Every packet but the last holds 8192 bytes, so the header's
        // uncompressed size says; but a file written by the reference keeps only the low 32 bits of that size
        // (src/file_header.hpp:31-36) and a header can lie, so the count is bounded from below by the stream itself
        // (a packet is at most 8704 bytes long) and from above by it as well (at least 4).
        const size_t stream_bytes = static_cast<size_t>(stream_end - FileHeader::HEADER_LENGTH);
        const size_t by_header = (info.uncompressedFileSize + kPacket - 1) / kPacket;
        const size_t at_least = (stream_bytes + kSlot - 1) / kSlot, at_most = stream_bytes / GPUAR_PACKET_HEADER_BYTES + 1;
        const size_t expect_packets = std::min(std::max(by_header, at_least), at_most);
        ensureBuffers(expect_packets, false);
        const int in_fd = fileno(openFile);
        hip_check(hipSetDevice(devices[0]), "hipSetDevice");
        trace("decompress: files open, device set");
        MappedInput mapped;
        mapped.open(in_fd, fileSize);
        mapped.warmFirstWindow(devices[0]);
        // packet lengths from the index trailer when the file has one (packet_index.hpp)
        std::vector<uint16_t> index;
        const bool indexed = PacketIndex::read(openFile, FileHeader::HEADER_LENGTH, stream_end, fileSize, index);

        ChunkMap map;
        std::thread scanner;
        std::atomic<bool> stop_scan{false};
        // whatever way this scope is left, the scanner is told to stop and joined before `map` and `mapped` go away
        struct JoinScanner {
            std::thread &t;
            std::atomic<bool> &stop;
            ~JoinScanner() {
                stop = true;
                if (t.joinable()) t.join();
            }
        } join_scanner{scanner, stop_scan};
        if (indexed) {                   // prefix sums of the stored lengths (already checked against the stream size)
            uint64_t at = FileHeader::HEADER_LENGTH;
            for (size_t p = 0; p < index.size();) {
                ChunkMap::Chunk c;
                c.begin = at;
                const size_t want = rampPackets(map.chunks.size(), chunkPackets);
                while (p < index.size() && c.n_packets < want) {
                    const size_t clen = index[p++];
                    if (clen < GPUAR_PACKET_HEADER_BYTES || clen > kSlot) throw std::runtime_error("Invalid file length");
                    at += clen;
                    ++c.n_packets;
                }
                c.end = at;
                map.push(c);
            }
            map.finish();
        } else {
            // the header walk of src/gpu_compressor.cpp:299-312 (`off += clen`), four bytes per packet, ahead of the lanes:
            // through the mapping when there is one, else by pread
            scanner = std::thread([&] {
                try {
                    uint64_t at = FileHeader::HEADER_LENGTH;
                    size_t n_pushed = 0;
                    while (at < stream_end && !stop_scan) {
                        ChunkMap::Chunk c;
                        c.begin = at;
                        const size_t want = rampPackets(n_pushed, chunkPackets);
                        while (at < stream_end && c.n_packets < want) {
                            uint8_t h[GPUAR_PACKET_HEADER_BYTES];
                            if (stream_end - at < sizeof h) throw std::runtime_error("Incorrect file format");
                            if (mapped.data()) std::memcpy(h, mapped.data() + at, sizeof h);
                            else if (::pread(in_fd, h, sizeof h, static_cast<off_t>(at)) != static_cast<ssize_t>(sizeof h))
                                throw std::runtime_error("Incorrect file format");
                            const size_t clen = getPacketSize(h);
                            if (clen < GPUAR_PACKET_HEADER_BYTES || clen > kSlot || at + clen > stream_end)
                                throw std::runtime_error("Invalid file length");
                            at += clen;
                            ++c.n_packets;
                        }
                        c.end = at;
                        map.push(c);
                        ++n_pushed;
                    }
                    map.finish();
                } catch (...) {
                    map.finish(std::current_exception());
                }
            });
        }

        // what the packet count promises, allocated up front (best effort); the real length is set at the end.  A header
        // may claim anything up to 2048 times the stream's size (a packet is at least 4 bytes): the preallocation is bounded
        // by what packets can plausibly hold -- 64 times the stream (8192 zeros code to 210 bytes, 39 times) -- and the file
        // simply grows beyond that if the packets really say so.
        const size_t prealloc_packets = std::min(expect_packets, 64 * at_least);
        (void)::fallocate(out_fd, 0, 0, static_cast<off_t>(static_cast<uint64_t>(prealloc_packets) * kPacket));
        OrderedOffsets place;
        std::vector<std::atomic<size_t>> next_of_device(G);
        for (auto &n : next_of_device) n = 0;
        std::vector<std::atomic<uint64_t>> device_bytes(G), device_chunks(G);      // what each device decoded (GPUAR_TRACE)
        for (size_t g = 0; g < G; ++g) device_bytes[g] = 0, device_chunks[g] = 0;
        std::mutex progress_lock;
        {
            Writer writer(out_fd, false, 0, "Write uncompressed data to output file failed");
            writer.on_progress = [&](size_t plain) {
                if (!plain) return;
                std::lock_guard<std::mutex> hold(progress_lock);
                info.processedUncompressedSize += plain;
                monitor->updateProgress(&info);
            };
            writer.on_failure = [&] {
                for (DeviceBuffers *b : buffers) b->abortWaits();
                place.abort();
            };
            std::exception_ptr lane_failure;
            try {
                runLanes(
                    [&](size_t lane, DeviceBuffers &b, Failure &failure) {
                        const size_t g = lane / kLanesPerDevice;
                        std::vector<uint8_t> staged;        // unmapped input only: the chunk's bytes, for the header walk below
                        for (;;) {
                            const size_t c = g + G * next_of_device[g].fetch_add(1);
                            ChunkMap::Chunk chunk;
                            if (failure.stop || !map.get(c, chunk)) break;
                            if (!b.cap) b.allocate(b.device, chunkPackets, false);
                            hip_check(hipSetDevice(b.device), "hipSetDevice");
                            b.epoch = static_cast<hipEvent_t>(epochOf(g));
                            const size_t n_stream = static_cast<size_t>(chunk.end - chunk.begin);
                            const bool from_mapping = b.upload(b.d_stream, mapped, in_fd, chunk.begin, n_stream, "Invalid file length");
                            const uint8_t *bytes = mapped.data() ? mapped.data() + chunk.begin : nullptr;
                            if (!bytes) {
                                staged.resize(n_stream);
                                sliced_io<false>(in_fd, staged.data(), n_stream, chunk.begin, "Invalid file length");
                                bytes = staged.data();
                            }
                            // packet offsets inside the chunk, and what every packet says it holds (u16 at +2): all but the
                            // file's last one hold 8192 bytes (src/gpu_compressor.cpp:326-331).  What is written comes from
                            // the packets, not from the header's size field: a file written by the reference carries garbage
                            // in the upper half of that field (src/file_header.hpp:31-36), and --host decodes by ulen too.
                            uint64_t produced = 0;
                            size_t off = 0;
                            bool all_full = true;
                            for (size_t p = 0; p < chunk.n_packets; ++p) {
                                b.h_offsets[p] = off;
                                if (n_stream - off < GPUAR_PACKET_HEADER_BYTES) throw std::runtime_error("Invalid file length");
                                const uint8_t *pkt = bytes + off;
                                const size_t clen = getPacketSize(pkt);
                                if (clen < GPUAR_PACKET_HEADER_BYTES || off + clen > n_stream) throw std::runtime_error("Invalid file length");
                                const size_t ulen = std::min<size_t>(kPacket, pkt[2] | (static_cast<size_t>(pkt[3]) << 8));
                                all_full = all_full && (ulen == kPacket || p + 1 == chunk.n_packets);
                                produced += ulen;
                                off += clen;
                            }
                            if (off != n_stream) throw std::runtime_error("Invalid file length");
                            b.h_offsets[chunk.n_packets] = off;
                            const uint64_t out_at = place.take(c, produced);
                            device_bytes[g] += produced;
                            device_chunks[g] += 1;
                            const uint32_t flags = b.decodeChunk(chunk.n_packets);        // (synchronises the lane's stream)
                            if (from_mapping) mapped.release(chunk.begin, n_stream);
                            // the file was cut short under the mapping (what was read behind its new end are zeros,
                            // input_guard.hpp): the reference's short fread(), src/gpu_compressor.cpp:299-307
                            if (!mapped.intact(in_fd)) throw std::runtime_error("Invalid file length");
                            if (flags & GPUAR_STATUS_BAD_PACKET)
                                throw std::runtime_error("Incorrect file format (malformed packet between file offsets " + std::to_string(chunk.begin) +
                                                         " and " + std::to_string(chunk.end) + ")");
                            if (all_full) {
                                b.drain(writer, c, b.d_plain, static_cast<size_t>(produced), out_at, static_cast<size_t>(produced));
                            } else {             // short packets inside the chunk: one piece per packet
                                uint64_t at = out_at;
                                for (size_t p = 0; p < chunk.n_packets; ++p) {
                                    const uint8_t *pkt = bytes + b.h_offsets[p];
                                    const size_t ulen = std::min<size_t>(kPacket, pkt[2] | (static_cast<size_t>(pkt[3]) << 8));
                                    if (ulen) b.drain(writer, c, b.d_plain + p * kPacket, ulen, at, ulen);
                                    at += ulen;
                                }
                            }
                        }
                    },
                    [&] {
                        writer.abort();
                        place.abort();
                    });
            } catch (...) {
                lane_failure = std::current_exception();
            }
            stop_scan = true;
            if (scanner.joinable()) scanner.join();
            if (lane_failure) {
                writer.abort();
                (void)writer.finish(0);      // rethrows the writer's own failure when it has one: that is the cause then
                std::rethrow_exception(lane_failure);
            }
            trace("decompress: lanes done");
            trace("decompress: input windows unregistered while the job ran:", static_cast<double>(mapped.windowsTrimmed()));
            for (size_t g = 0; g < G; ++g)
                trace(("decompress: device " + std::to_string(g) + " decoded " + std::to_string(device_bytes[g].load()) + " bytes in " +
                       std::to_string(device_chunks[g].load()) + " chunks").c_str());
            (void)writer.finish(0);
        }
        info.uncompressedFileSize = info.processedUncompressedSize = static_cast<size_t>(place.sum());     // what the packets held
        if (::ftruncate(out_fd, static_cast<off_t>(place.sum())) != 0) throw std::runtime_error("Write uncompressed data to output file failed");
        mapped.close();
        closeFiles();
    } catch (...) {
        if (::ftruncate(out_fd, 0) != 0) {}      // a failed job leaves an empty file, not a half-written one
        closeFiles();
        throw;
    }
    io_timer.stop();
    finishTimes(info);
    return info;
}

}  // namespace gip
