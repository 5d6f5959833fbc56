// gpu_compressor.cpp -- host pipeline of the GPU path.
//
// The reference moves one 8704-byte packet per memcpy and per fwrite, all on one thread
// (/root/reference/src/gpu_compressor.cpp:134-171, 280-340).  Here the file is cut into CHUNKS of
// whole packets (a multiple of 64, 64 MiB of input by default) and every chunk goes down a lane:
//
//     pread (sliced over helper threads, straight into pinned memory)  ->  H2D  ->  kernels  ->  D2H
//         ->  its place in the output file is known  ->  pwrite (sliced, straight from pinned memory)
//
// Each GPU runs kLanesPerDevice lanes at once, each a host thread with its own HIP stream and buffers,
// so the file read of one chunk, the PCIe copies and kernels of another and the file write of a third
// overlap without any explicit scheduling; chunks are dealt to the GPUs round-robin (chunk c -> device
// c mod G), which is the contiguous-packet-range sharding of SURVEY.md section 8(e) at chunk grain.
// The only serial step is the running output offset: a chunk may be written once every chunk before
// it has said how many bytes it produced (OrderedOffsets).  Decoding works the same way in the other
// direction; where the packets of a chunk start is known from the index trailer (packet_index.hpp) or
// from a scanner thread that walks the packet headers (`off += clen`, 4 bytes read per packet) ahead of
// the lanes.  The files produced are byte-identical to those of the reference's loop from byte 20 on.
#include "gpu_compressor.hpp"

#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "file_header.hpp"
#include "gpuar_hip.h"
#include "packet_index.hpp"

namespace gip {

namespace {

constexpr size_t kPacket = GPUAR_PACKET_BYTES;
constexpr size_t kSlot = GPUAR_SLOT_BYTES;
constexpr size_t kLanesPerDevice = 3;      // chunks in flight per GPU: one reading, one on the GPU, one writing
constexpr size_t kIoSlice = 16u << 20;     // a pread / pwrite is cut into slices of this size, one helper thread each
constexpr size_t kIoThreads = 4;           // ... at most this many at a time per lane

void hip_check(hipError_t e, const char *what) {
    // same message shape as src/gpu_compressor.cpp:189-192
    if (e != hipSuccess) throw std::runtime_error(std::string("Fail to execute kernel code: ") + what + ": " + hipGetErrorString(e));
}
void gpuar_check(int code, const char *what) {
    if (code != GPUAR_OK) throw std::runtime_error(std::string("Fail to execute kernel code: ") + what + ": " + gpuar_hip_error_string(code));
}

// pread / pwrite of a large range, cut into slices that run on a few threads: one thread moves
// page-cache pages at 2-5 GB/s, which is what bounded the previous, single-threaded pipeline.
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

// The one serial quantity of the pipeline: where the output of chunk c starts = what all chunks
// before it produced.  A lane calls take(c, n): it blocks until chunks 0..c-1 have called, returns
// the running total before its own n bytes and lets chunk c+1 go.
class OrderedOffsets {
  public:
    explicit OrderedOffsets(uint64_t start) : total(start) {}
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
    uint64_t total;
    bool aborted = false;
};

}  // namespace

// Keeps the first failure of any lane and tells the others to stop.
struct GPUCompressor::Failure {
    std::mutex lock;
    std::exception_ptr first;
    std::atomic<bool> stop{false};
    void set(std::exception_ptr e) {
        std::lock_guard<std::mutex> hold(lock);
        // "pipeline stopped" is what the lanes woken by an abort throw: never the cause
        if (!first) first = e;
        stop = true;
    }
};

// Everything one lane needs for one chunk of `cap` packets.
struct GPUCompressor::DeviceBuffers {
    int device = 0;
    size_t cap = 0;                 // packets
    hipStream_t stream = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    uint8_t *d_plain = nullptr;     // cap * 8192
    uint8_t *d_slots = nullptr;     // cap * 8704
    uint8_t *d_stream = nullptr;    // cap * 8704 (+16)
    uint64_t *d_offsets = nullptr;  // cap + 1
    uint8_t *h_plain = nullptr;     // pinned, cap * 8192
    uint8_t *h_stream = nullptr;    // pinned, cap * 8704 (+16)
    uint64_t *h_offsets = nullptr;  // pinned, cap + 2: the word behind the offsets receives this lane's status word
    uint32_t *d_status = nullptr;   // this lane's own status word (device): its launches report here, nobody else's do
    hipEvent_t epoch = nullptr;     // the device's common time base (owned by the device's first lane)
    std::vector<std::pair<float, float>> busy;   // [begin, end) of every chunk's kernels, ms since `epoch`

    void allocate(int dev, size_t packets) {
        device = dev;
        cap = packets;
        hip_check(hipSetDevice(device), "hipSetDevice");
        // non-blocking: a lane must never wait for another lane's copies or kernels through the NULL stream
        hip_check(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipEventCreate(&t0), "hipEventCreate");
        hip_check(hipEventCreate(&t1), "hipEventCreate");
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_plain), cap * kPacket), "hipMalloc");
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_slots), cap * kSlot), "hipMalloc");
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_stream), cap * kSlot + 16), "hipMalloc");
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_offsets), (cap + 2) * sizeof(uint64_t)), "hipMalloc");
        d_status = reinterpret_cast<uint32_t *>(d_offsets + cap + 1);
        hip_check(hipHostMalloc(reinterpret_cast<void **>(&h_plain), cap * kPacket, hipHostMallocDefault), "hipHostMalloc");
        hip_check(hipHostMalloc(reinterpret_cast<void **>(&h_stream), cap * kSlot + 16, hipHostMallocDefault), "hipHostMalloc");
        hip_check(hipHostMalloc(reinterpret_cast<void **>(&h_offsets), (cap + 2) * sizeof(uint64_t), hipHostMallocDefault), "hipHostMalloc");
    }
    void release() {
        if (!cap) return;
        (void)hipSetDevice(device);
        (void)hipFree(d_plain);
        (void)hipFree(d_slots);
        (void)hipFree(d_stream);
        (void)hipFree(d_offsets);
        (void)hipHostFree(h_plain);
        (void)hipHostFree(h_stream);
        (void)hipHostFree(h_offsets);
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
        (void)hipStreamDestroy(stream);
        cap = 0;
    }

    // where this chunk's kernels ran on the device's clock
    void noteBusy() {
        float begin = 0, end = 0;
        hip_check(hipEventElapsedTime(&begin, epoch, t0), "event");
        hip_check(hipEventElapsedTime(&end, epoch, t1), "event");
        busy.emplace_back(begin, end);
    }

    // h_plain[0..n_plain) -> h_stream[0..n_stream), h_offsets[0..n_packets]; returns n_stream and, through `flags`,
    // what THIS chunk's launches reported (GPUAR_STATUS_*): the status word travels back with the offsets, on the
    // lane's own stream -- no device-wide synchronisation, no flag shared with another lane
    size_t encodeChunk(size_t n_plain, uint32_t &flags) {
        const size_t n_packets = (n_plain + kPacket - 1) / kPacket;
        hip_check(hipMemsetAsync(d_status, 0, sizeof(uint32_t), stream), "memset");
        hip_check(hipMemcpyAsync(d_plain, h_plain, n_plain, hipMemcpyHostToDevice, stream), "H2D");
        hip_check(hipEventRecord(t0, stream), "event");
        gpuar_check(gpuar_hip_encode(d_plain, n_plain, d_slots, d_status, stream), "gpuar_hip_encode");
        gpuar_check(gpuar_hip_compact(d_slots, n_packets, d_stream, d_offsets, stream), "gpuar_hip_compact");
        hip_check(hipEventRecord(t1, stream), "event");
        hip_check(hipMemcpyAsync(h_offsets, d_offsets, (n_packets + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipMemcpyAsync(h_offsets + n_packets + 1, d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipStreamSynchronize(stream), "sync");
        const size_t n_stream = static_cast<size_t>(h_offsets[n_packets]);
        flags = *reinterpret_cast<const uint32_t *>(h_offsets + n_packets + 1);
        hip_check(hipMemcpyAsync(h_stream, d_stream, n_stream, hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipStreamSynchronize(stream), "sync");
        noteBusy();
        return n_stream;
    }

    // h_stream[0..n_stream) with h_offsets[0..n_packets] -> h_plain[0..n_packets*8192); returns this chunk's status flags
    uint32_t decodeChunk(size_t n_stream, size_t n_packets) {
        hip_check(hipMemsetAsync(d_status, 0, sizeof(uint32_t), stream), "memset");
        hip_check(hipMemcpyAsync(d_stream, h_stream, n_stream, hipMemcpyHostToDevice, stream), "H2D");
        hip_check(hipMemcpyAsync(d_offsets, h_offsets, (n_packets + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream), "H2D");
        hip_check(hipEventRecord(t0, stream), "event");
        gpuar_check(gpuar_hip_decode_stream(d_stream, d_offsets, n_packets, d_plain, d_status, stream), "gpuar_hip_decode_stream");
        hip_check(hipEventRecord(t1, stream), "event");
        hip_check(hipMemcpyAsync(h_plain, d_plain, n_packets * kPacket, hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipMemcpyAsync(h_offsets + n_packets + 1, d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipStreamSynchronize(stream), "sync");
        noteBusy();
        return *reinterpret_cast<const uint32_t *>(h_offsets + n_packets + 1);
    }
};

GPUCompressor::GPUCompressor() {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw std::runtime_error("No HIP device found (use --host to run the codec on the CPU)");
    devices.push_back(0);
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
void GPUCompressor::ensureBuffers(size_t total_packets) {
    const size_t G = devices.size();
    const size_t lanes = G * kLanesPerDevice;
    const size_t share = ((total_packets + lanes - 1) / lanes + 63) / 64 * 64;
    const size_t cap = std::max<size_t>(64, std::min(batchPackets, share));
    const bool have = buffers.size() == lanes && !buffers.empty() && chunkPackets == cap;
    if (have) return;
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

// Runs `work(lane, buffers, failure)` on every lane (lanes g*kLanesPerDevice .. belong to device g) and
// rethrows the first failure.  `work` pulls chunk numbers until there are none left.
template <typename Work>
void GPUCompressor::runLanes(Work &&work) {
    Failure failure;
    std::vector<std::thread> threads;
    for (size_t l = 0; l < buffers.size(); ++l)
        threads.emplace_back([&, l] {
            try {
                work(l, *buffers[l], failure);
            } catch (...) {
                failure.set(std::current_exception());
            }
        });
    for (auto &t : threads) t.join();
    if (failure.first) std::rethrow_exception(failure.first);
}

CompressionInfo GPUCompressor::compress(ProgressMonitor *monitor) {
    CompressionInfo info;
    monitor->reset();
    process_timer.reset();
    io_timer.reset();
    io_timer.start();
    openFiles();
    try {
        info.uncompressedFileSize = getFileSize(openFile);
        const size_t total_packets = (info.uncompressedFileSize + kPacket - 1) / kPacket;
        ensureBuffers(total_packets);
        const size_t G = devices.size();
        const size_t chunk_bytes = chunkPackets * kPacket;
        const size_t n_chunks = (info.uncompressedFileSize + chunk_bytes - 1) / chunk_bytes;
        const int in_fd = fileno(openFile), out_fd = fileno(saveFile);
        OrderedOffsets place(FileHeader::HEADER_LENGTH);
        std::vector<std::vector<uint16_t>> chunk_clens(writeIndex ? n_chunks : 0);     // for the optional index trailer
        std::vector<std::atomic<size_t>> next_of_device(G);
        for (auto &n : next_of_device) n = 0;
        std::mutex progress_lock;

        runLanes([&](size_t lane, DeviceBuffers &b, Failure &failure) {
            const size_t g = lane / kLanesPerDevice;
            try {
                for (;;) {
                    // contiguous packet ranges in file order, dealt round-robin: chunk c belongs to device c mod G
                    const size_t c = g + G * next_of_device[g].fetch_add(1);
                    if (c >= n_chunks || failure.stop) break;
                    if (!b.cap) b.allocate(b.device, chunkPackets);
                    hip_check(hipSetDevice(b.device), "hipSetDevice");
                    b.epoch = static_cast<hipEvent_t>(epochOf(g));
                    const uint64_t at = static_cast<uint64_t>(c) * chunk_bytes;
                    const size_t n_plain = static_cast<size_t>(std::min<uint64_t>(chunk_bytes, info.uncompressedFileSize - at));
                    sliced_io<false>(in_fd, b.h_plain, n_plain, at, "Read input file failed");
                    uint32_t flags = 0;
                    const size_t n_stream = b.encodeChunk(n_plain, flags);
                    if (flags & GPUAR_STATUS_SLOT_OVERFLOW)
                        throw std::runtime_error("a packet outgrew its 8704-byte slot (input bytes " + std::to_string(at) + " .. " + std::to_string(at + n_plain) + ")");
                    const size_t n_packets = (n_plain + kPacket - 1) / kPacket;
                    if (writeIndex) {
                        chunk_clens[c].resize(n_packets);
                        for (size_t p = 0; p < n_packets; ++p) chunk_clens[c][p] = static_cast<uint16_t>(b.h_offsets[p + 1] - b.h_offsets[p]);
                    }
                    const uint64_t out_at = place.take(c, n_stream);      // every chunk before this one has said its size
                    sliced_io<true>(out_fd, b.h_stream, n_stream, out_at, "Write compressed data to output file failed");
                    std::lock_guard<std::mutex> hold(progress_lock);
                    info.processedUncompressedSize += n_plain;
                    monitor->updateProgress(&info);
                }
            } catch (...) {
                failure.set(std::current_exception());
                place.abort();
            }
        });

        info.compressedFileSize = static_cast<size_t>(place.sum());
        if (writeIndex) {
            std::vector<uint16_t> all;
            for (const auto &v : chunk_clens) all.insert(all.end(), v.begin(), v.end());
            if (std::fseek(saveFile, static_cast<long>(info.compressedFileSize), SEEK_SET) != 0) throw std::runtime_error("Seek file failed");
            PacketIndex::write(saveFile, all);
            if (std::fflush(saveFile) != 0) throw std::runtime_error("Write packet index failed");
        }
        FileHeader header;
        header.setCompressedFileSize(info.compressedFileSize);
        header.setUncompressedFileSize(info.uncompressedFileSize);
        if (::pwrite(out_fd, header.getData(), FileHeader::HEADER_LENGTH, 0) != FileHeader::HEADER_LENGTH)
            throw std::runtime_error("Write data to file failed");
        closeFiles();
    } catch (...) {
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

}  // namespace

CompressionInfo GPUCompressor::decompress(ProgressMonitor *monitor) {
    CompressionInfo info;
    monitor->reset();
    process_timer.reset();
    io_timer.reset();
    io_timer.start();
    openFiles();
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
        ensureBuffers(std::min(std::max(by_header, at_least), at_most));
        const int in_fd = fileno(openFile), out_fd = fileno(saveFile);
        // packet lengths from the index trailer when the file has one (packet_index.hpp)
        std::vector<uint16_t> index;
        const bool indexed = PacketIndex::read(openFile, FileHeader::HEADER_LENGTH, stream_end, fileSize, index);

        ChunkMap map;
        std::thread scanner;
        std::atomic<bool> stop_scan{false};
        if (indexed) {                   // prefix sums of the stored lengths (already checked against the stream size)
            uint64_t at = FileHeader::HEADER_LENGTH;
            for (size_t p = 0; p < index.size();) {
                ChunkMap::Chunk c;
                c.begin = at;
                while (p < index.size() && c.n_packets < chunkPackets) {
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
            // the header walk of src/gpu_compressor.cpp:299-312 (`off += clen`), four bytes read per packet, ahead of the lanes
            scanner = std::thread([&] {
                try {
                    uint64_t at = FileHeader::HEADER_LENGTH;
                    while (at < stream_end && !stop_scan) {
                        ChunkMap::Chunk c;
                        c.begin = at;
                        while (at < stream_end && c.n_packets < chunkPackets) {
                            uint8_t h[GPUAR_PACKET_HEADER_BYTES];
                            if (stream_end - at < sizeof h || ::pread(in_fd, h, sizeof h, static_cast<off_t>(at)) != static_cast<ssize_t>(sizeof h))
                                throw std::runtime_error("Incorrect file format");
                            const size_t clen = getPacketSize(h);
                            if (clen < GPUAR_PACKET_HEADER_BYTES || clen > kSlot || at + clen > stream_end)
                                throw std::runtime_error("Invalid file length");
                            at += clen;
                            ++c.n_packets;
                        }
                        c.end = at;
                        map.push(c);
                    }
                    map.finish();
                } catch (...) {
                    map.finish(std::current_exception());
                }
            });
        }

        OrderedOffsets place(0);
        std::vector<std::atomic<size_t>> next_of_device(G);
        for (auto &n : next_of_device) n = 0;
        std::mutex progress_lock;
        try {
            runLanes([&](size_t lane, DeviceBuffers &b, Failure &failure) {
                const size_t g = lane / kLanesPerDevice;
                try {
                    for (;;) {
                        const size_t c = g + G * next_of_device[g].fetch_add(1);
                        ChunkMap::Chunk chunk;
                        if (failure.stop || !map.get(c, chunk)) break;
                        if (!b.cap) b.allocate(b.device, chunkPackets);
                        hip_check(hipSetDevice(b.device), "hipSetDevice");
                        b.epoch = static_cast<hipEvent_t>(epochOf(g));
                        const size_t n_stream = static_cast<size_t>(chunk.end - chunk.begin);
                        sliced_io<false>(in_fd, b.h_stream, n_stream, chunk.begin, "Invalid file length");
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
                            const uint8_t *pkt = b.h_stream + off;
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
                        const uint32_t flags = b.decodeChunk(n_stream, chunk.n_packets);
                        if (flags & GPUAR_STATUS_BAD_PACKET)
                            throw std::runtime_error("Incorrect file format (malformed packet between file offsets " + std::to_string(chunk.begin) + " and " + std::to_string(chunk.end) + ")");
                        if (all_full) {
                            sliced_io<true>(out_fd, b.h_plain, static_cast<size_t>(produced), out_at, "Write uncompressed data to output file failed");
                        } else {             // short packets inside the chunk: one write per packet
                            uint64_t at = out_at;
                            for (size_t p = 0; p < chunk.n_packets; ++p) {
                                const uint8_t *pkt = b.h_stream + b.h_offsets[p];
                                const size_t ulen = std::min<size_t>(kPacket, pkt[2] | (static_cast<size_t>(pkt[3]) << 8));
                                if (ulen) sliced_io<true>(out_fd, b.h_plain + p * kPacket, ulen, at, "Write uncompressed data to output file failed");
                                at += ulen;
                            }
                        }
                        std::lock_guard<std::mutex> hold(progress_lock);
                        info.processedUncompressedSize += static_cast<size_t>(produced);
                        monitor->updateProgress(&info);
                    }
                } catch (...) {
                    failure.set(std::current_exception());
                    place.abort();
                }
            });
        } catch (...) {
            stop_scan = true;
            if (scanner.joinable()) scanner.join();
            throw;
        }
        if (scanner.joinable()) scanner.join();
        info.uncompressedFileSize = info.processedUncompressedSize;     // what the packets held
        closeFiles();
    } catch (...) {
        closeFiles();
        throw;
    }
    io_timer.stop();
    finishTimes(info);
    return info;
}

}  // namespace gip
