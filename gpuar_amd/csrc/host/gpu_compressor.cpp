#include "gpu_compressor.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "file_header.hpp"
#include "packet_index.hpp"
#include "gpuar_hip.h"

namespace gip {

namespace {

constexpr size_t kPacket = GPUAR_PACKET_BYTES;
constexpr size_t kSlot = GPUAR_SLOT_BYTES;

void hip_check(hipError_t e, const char *what) {
    // same message shape as src/gpu_compressor.cpp:189-192
    if (e != hipSuccess) throw std::runtime_error(std::string("Fail to execute kernel code: ") + what + ": " + hipGetErrorString(e));
}
void gpuar_check(int code, const char *what) {
    if (code != GPUAR_OK) throw std::runtime_error(std::string("Fail to execute kernel code: ") + what + ": " + gpuar_hip_error_string(code));
}

}  // namespace

// Everything one GPU needs for one round of `cap` packets.
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
    uint64_t *h_offsets = nullptr;  // pinned, cap + 1
    // per-round results
    size_t n_plain = 0, n_packets = 0, n_stream = 0;
    float kernel_ms = 0;
    std::exception_ptr failure;

    void allocate(int dev, size_t packets) {
        device = dev;
        cap = packets;
        hip_check(hipSetDevice(device), "hipSetDevice");
        hip_check(hipStreamCreate(&stream), "hipStreamCreate");
        hip_check(hipEventCreate(&t0), "hipEventCreate");
        hip_check(hipEventCreate(&t1), "hipEventCreate");
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_plain), cap * kPacket), "hipMalloc");
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_slots), cap * kSlot), "hipMalloc");
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_stream), cap * kSlot + 16), "hipMalloc");
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_offsets), (cap + 1) * sizeof(uint64_t)), "hipMalloc");
        hip_check(hipHostMalloc(reinterpret_cast<void **>(&h_plain), cap * kPacket, hipHostMallocDefault), "hipHostMalloc");
        hip_check(hipHostMalloc(reinterpret_cast<void **>(&h_stream), cap * kSlot + 16, hipHostMallocDefault), "hipHostMalloc");
        hip_check(hipHostMalloc(reinterpret_cast<void **>(&h_offsets), (cap + 1) * sizeof(uint64_t), hipHostMallocDefault), "hipHostMalloc");
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

    // h_plain[0..n_plain) -> h_stream[0..n_stream), h_offsets[0..n_packets]
    void encodeRound() {
        hip_check(hipSetDevice(device), "hipSetDevice");
        n_packets = (n_plain + kPacket - 1) / kPacket;
        hip_check(hipMemcpyAsync(d_plain, h_plain, n_plain, hipMemcpyHostToDevice, stream), "H2D");
        hip_check(hipEventRecord(t0, stream), "event");
        gpuar_check(gpuar_hip_encode(d_plain, n_plain, d_slots, stream), "gpuar_hip_encode");
        gpuar_check(gpuar_hip_compact(d_slots, n_packets, d_stream, d_offsets, stream), "gpuar_hip_compact");
        hip_check(hipEventRecord(t1, stream), "event");
        hip_check(hipMemcpyAsync(h_offsets, d_offsets, (n_packets + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipStreamSynchronize(stream), "sync");
        n_stream = static_cast<size_t>(h_offsets[n_packets]);
        hip_check(hipMemcpyAsync(h_stream, d_stream, n_stream, hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipStreamSynchronize(stream), "sync");
        hip_check(hipEventElapsedTime(&kernel_ms, t0, t1), "event");
        uint32_t flags = 0;
        gpuar_check(gpuar_hip_status(&flags), "gpuar_hip_status");
        if (flags & GPUAR_STATUS_SLOT_OVERFLOW) throw std::runtime_error("a packet outgrew its 8704-byte slot");
    }

    // h_stream[0..n_stream) with h_offsets[0..n_packets] -> h_plain[0..n_packets*8192)
    void decodeRound() {
        hip_check(hipSetDevice(device), "hipSetDevice");
        hip_check(hipMemcpyAsync(d_stream, h_stream, n_stream, hipMemcpyHostToDevice, stream), "H2D");
        hip_check(hipMemcpyAsync(d_offsets, h_offsets, (n_packets + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream), "H2D");
        hip_check(hipEventRecord(t0, stream), "event");
        gpuar_check(gpuar_hip_decode_stream(d_stream, d_offsets, n_packets, d_plain, stream), "gpuar_hip_decode_stream");
        hip_check(hipEventRecord(t1, stream), "event");
        hip_check(hipMemcpyAsync(h_plain, d_plain, n_packets * kPacket, hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipStreamSynchronize(stream), "sync");
        hip_check(hipEventElapsedTime(&kernel_ms, t0, t1), "event");
        uint32_t flags = 0;
        gpuar_check(gpuar_hip_status(&flags), "gpuar_hip_status");
        if (flags & GPUAR_STATUS_BAD_PACKET) throw std::runtime_error("Incorrect file format");
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
}

// Two buffer sets per device: while the GPUs work on one round, the host reads the next round's
// input into the other set and writes the previous round's output.  Sized for the job at hand: a
// file of `total_packets` needs at most ceil(total / devices) packets per device per round, and a
// job that fits one round never touches the second set -- pinned allocations are what a short run
// of the CLI spends most of its time on.
void GPUCompressor::ensureBuffers(size_t total_packets) {
    const size_t G = devices.size();
    const size_t per_dev = ((total_packets + G - 1) / G + 63) / 64 * 64;
    const size_t cap = std::max<size_t>(64, std::min(batchPackets, per_dev));
    const bool one_round = total_packets <= G * cap;
    const bool have = buffers.size() == 2 * G && !buffers.empty() && buffers[0]->cap >= cap && buffers[0]->cap <= batchPackets &&
                      (one_round || buffers[G]->cap == buffers[0]->cap);
    if (have) return;
    releaseBuffers();
    for (int set = 0; set < 2; ++set)
        for (int dev : devices) {
            DeviceBuffers *b = new DeviceBuffers();
            buffers.push_back(b);
            if (set == 0 || !one_round) b->allocate(dev, cap);
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

namespace {

// One round in flight: the device work of every GPU on one buffer set, each on its own host thread.
struct Round {
    std::vector<std::thread> workers;
    bool active = false;
    void join() {
        for (auto &w : workers) w.join();
        workers.clear();
    }
};

}  // namespace

// Pipeline over rounds r = 0, 1, ... with buffer set r & 1:
//   read input of round r  ||  GPUs run round r-1   (then)   GPUs run round r  ||  write output of round r-1
// File order is kept because rounds, and devices within a round, are drained in order.
CompressionInfo GPUCompressor::compress(ProgressMonitor *monitor) {
    CompressionInfo info;
    monitor->reset();
    process_timer.reset();
    io_timer.reset();
    io_timer.start();
    openFiles();
    double kernel_ms_total = 0;
    Round in_flight;
    try {
        info.uncompressedFileSize = getFileSize(openFile);
        ensureBuffers((info.uncompressedFileSize + kPacket - 1) / kPacket);
        const size_t roundPackets = buffers[0]->cap;       // packets per device per round
        if (std::fseek(saveFile, FileHeader::HEADER_LENGTH, SEEK_SET) != 0) throw std::runtime_error("Seek file failed");
        info.compressedFileSize = FileHeader::HEADER_LENGTH;
        const size_t G = devices.size();
        size_t remaining = info.uncompressedFileSize;
        int set = 0;
        std::vector<uint16_t> all_clens;       // for the optional index trailer
        auto write_out = [&](int which) {      // output of the finished round that used buffer set `which`
            float slowest = 0;
            for (size_t g = 0; g < G; ++g) {
                DeviceBuffers *b = buffers[which * G + g];
                if (!b->n_plain) continue;
                if (b->failure) std::rethrow_exception(b->failure);
                slowest = std::max(slowest, b->kernel_ms);
                if (b->n_stream && std::fwrite(b->h_stream, 1, b->n_stream, saveFile) != b->n_stream)
                    throw std::runtime_error("Write compressed data to output file failed");
                if (writeIndex)
                    for (size_t p = 0; p < b->n_packets; ++p)
                        all_clens.push_back(static_cast<uint16_t>(b->h_offsets[p + 1] - b->h_offsets[p]));
                info.compressedFileSize += b->n_stream;
                info.processedUncompressedSize += b->n_plain;
            }
            kernel_ms_total += slowest;
            monitor->updateProgress(&info);
        };
        while (remaining > 0) {
            // contiguous packet ranges, device order = file order
            const size_t round_bytes = std::min(remaining, G * roundPackets * kPacket);
            const size_t round_packets = (round_bytes + kPacket - 1) / kPacket;
            const size_t per_dev = ((round_packets + G - 1) / G + 63) / 64 * 64;      // whole wavefronts
            size_t given = 0;
            for (size_t g = 0; g < G; ++g) {       // overlaps with the previous round's GPU work
                DeviceBuffers *b = buffers[set * G + g];
                if (!b->cap) b->allocate(devices[g], roundPackets);       // second set, first needed now
                b->n_plain = std::min(round_bytes - given, std::min(per_dev, b->cap) * kPacket);
                b->failure = nullptr;
                if (b->n_plain && std::fread(b->h_plain, 1, b->n_plain, openFile) != b->n_plain)
                    throw std::runtime_error("Read input file failed");
                given += b->n_plain;
            }
            std::vector<std::thread> next;
            const bool had_previous = in_flight.active;
            if (had_previous) in_flight.join();    // GPUs are free again
            for (size_t g = 0; g < G; ++g) {
                DeviceBuffers *b = buffers[set * G + g];
                if (b->n_plain) next.emplace_back([b] {
                    try {
                        b->encodeRound();
                    } catch (...) {
                        b->failure = std::current_exception();
                    }
                });
            }
            in_flight.workers = std::move(next);
            in_flight.active = true;
            if (had_previous) write_out(set ^ 1);   // this file write overlaps with the new round's GPU work
            remaining -= round_bytes;
            set ^= 1;
        }
        if (in_flight.active) {
            in_flight.join();
            write_out(set ^ 1);
        }
        if (writeIndex) PacketIndex::write(saveFile, all_clens);
        FileHeader header;
        header.setCompressedFileSize(info.compressedFileSize);
        header.setUncompressedFileSize(info.uncompressedFileSize);
        if (std::fseek(saveFile, 0, SEEK_SET) != 0) throw std::runtime_error("Seek file failed");
        if (std::fwrite(header.getData(), FileHeader::HEADER_LENGTH, 1, saveFile) != 1)
            throw std::runtime_error("Write data to file failed");
        closeFiles();
    } catch (...) {
        in_flight.join();
        closeFiles();
        throw;
    }
    io_timer.stop();
    // "Compute time" = kernels + their sync, as src/gpu_compressor.cpp:184-194; the rest is I/O
    info.processTime = kernel_ms_total;
    info.ioTime = std::max(0.0, io_timer.value() - kernel_ms_total);
    return info;
}

CompressionInfo GPUCompressor::decompress(ProgressMonitor *monitor) {
    CompressionInfo info;
    monitor->reset();
    process_timer.reset();
    io_timer.reset();
    io_timer.start();
    openFiles();
    double kernel_ms_total = 0;
    Round in_flight;
    try {
        FileHeader header;
        const size_t fileSize = getFileSize(openFile);
        if (std::fread(header.getData(), FileHeader::HEADER_LENGTH, 1, openFile) != 1 || !header.checkHeaderVersion())
            throw std::runtime_error("Incorrect file format");
        info = header.getInfo(fileSize);
        // every packet but the last holds 8192 bytes; a packet is at least 4 bytes long, which bounds a lying header
        ensureBuffers(std::min((info.uncompressedFileSize + kPacket - 1) / kPacket, fileSize / GPUAR_PACKET_HEADER_BYTES + 1));
        const size_t G = devices.size();
        const size_t stream_end = streamEnd(info, fileSize);
        size_t file_pos = FileHeader::HEADER_LENGTH;
        // packet lengths from the index trailer when the file has one (packet_index.hpp)
        std::vector<uint16_t> index;
        const bool indexed = PacketIndex::read(openFile, FileHeader::HEADER_LENGTH, stream_end, fileSize, index);
        size_t next_packet = 0;
        bool more = file_pos < stream_end;
        int set = 0;
        auto write_out = [&](int which) {
            float slowest = 0;
            for (size_t g = 0; g < G; ++g) {
                DeviceBuffers *b = buffers[which * G + g];
                if (!b->n_packets) continue;
                if (b->failure) std::rethrow_exception(b->failure);
                slowest = std::max(slowest, b->kernel_ms);
                // Every packet says how many bytes it holds (u16 at +2); all but the file's last one hold
                // 8192 (src/gpu_compressor.cpp:326-331).  The lengths written come from the packets, not
                // from the header's size field: a file written by the reference carries garbage in the
                // upper half of that field (src/file_header.hpp:31-36), and --host decodes by ulen too.
                size_t run_begin = 0, run_bytes = 0;       // contiguous bytes of h_plain not yet written
                auto flush = [&] {
                    if (run_bytes && std::fwrite(b->h_plain + run_begin, 1, run_bytes, saveFile) != run_bytes)
                        throw std::runtime_error("Write uncompressed data to output file failed");
                    info.processedUncompressedSize += run_bytes;
                    run_bytes = 0;
                };
                for (size_t p = 0; p < b->n_packets; ++p) {
                    const uint8_t *pkt = b->h_stream + b->h_offsets[p];
                    const size_t ulen = std::min<size_t>(kPacket, pkt[2] | (static_cast<size_t>(pkt[3]) << 8));
                    if (run_bytes == 0) run_begin = p * kPacket;
                    run_bytes += ulen;
                    if (ulen != kPacket) flush();          // a short packet ends the contiguous run
                }
                flush();
            }
            kernel_ms_total += slowest;
            monitor->updateProgress(&info);
        };
        while (more) {
            // Fill each device with up to cap packets.  One bulk read per device: with an index the
            // range and its offsets come from the stored lengths (prefix sum); without one the bytes
            // are read first and `off += clen` is walked in memory (src/gpu_compressor.cpp:299-312
            // walks it through the file, two reads per packet), then the file is wound back to the
            // end of the last whole packet.
            for (size_t g = 0; g < G; ++g) {
                DeviceBuffers *b = buffers[set * G + g];
                if (!b->cap) b->allocate(devices[g], buffers[g]->cap);    // second set, first needed now (the header understated the file)
                b->n_packets = 0;
                b->n_stream = 0;
                b->h_offsets[0] = 0;
                b->failure = nullptr;
                if (!more) continue;
                if (indexed) {
                    while (next_packet < index.size() && b->n_packets < b->cap) {
                        const size_t clen = index[next_packet++];
                        if (clen < GPUAR_PACKET_HEADER_BYTES || clen > kSlot) throw std::runtime_error("Invalid file length");
                        b->n_stream += clen;
                        b->h_offsets[++b->n_packets] = b->n_stream;
                    }
                    if (b->n_stream && std::fread(b->h_stream, 1, b->n_stream, openFile) != b->n_stream)
                        throw std::runtime_error("Invalid file length");
                } else {
                    const size_t want = std::min(stream_end - file_pos, b->cap * kSlot);
                    if (std::fread(b->h_stream, 1, want, openFile) != want) throw std::runtime_error("Invalid file length");
                    size_t off = 0;
                    while (off < want && b->n_packets < b->cap) {
                        if (want - off < GPUAR_PACKET_HEADER_BYTES) {
                            if (file_pos + want == stream_end) throw std::runtime_error("Incorrect file format");
                            break;                     // header cut by the read window: next round
                        }
                        const size_t clen = getPacketSize(b->h_stream + off);
                        if (clen < GPUAR_PACKET_HEADER_BYTES || clen > kSlot || file_pos + off + clen > stream_end)
                            throw std::runtime_error("Invalid file length");
                        if (off + clen > want) break;  // packet cut by the read window: next round
                        off += clen;
                        b->h_offsets[++b->n_packets] = off;
                    }
                    b->n_stream = off;
                    if (off != want && std::fseek(openFile, static_cast<long>(file_pos + off), SEEK_SET) != 0)
                        throw std::runtime_error("Seek file failed");
                }
                file_pos += b->n_stream;
                more = file_pos < stream_end;
            }
            std::vector<std::thread> next;
            const bool had_previous = in_flight.active;
            if (had_previous) in_flight.join();
            for (size_t g = 0; g < G; ++g) {
                DeviceBuffers *b = buffers[set * G + g];
                if (b->n_packets) next.emplace_back([b] {
                    try {
                        b->decodeRound();
                    } catch (...) {
                        b->failure = std::current_exception();
                    }
                });
            }
            in_flight.workers = std::move(next);
            in_flight.active = true;
            if (had_previous) write_out(set ^ 1);
            set ^= 1;
        }
        if (in_flight.active) {
            in_flight.join();
            write_out(set ^ 1);
        }
        info.uncompressedFileSize = info.processedUncompressedSize;     // what the packets held
        closeFiles();
    } catch (...) {
        in_flight.join();
        closeFiles();
        throw;
    }
    io_timer.stop();
    info.processTime = kernel_ms_total;
    info.ioTime = std::max(0.0, io_timer.value() - kernel_ms_total);
    return info;
}

}  // namespace gip
