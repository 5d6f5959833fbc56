// cpu_compressor.cpp -- host packet loop of `gpuar --host`
// (what src/cpu_compressor.cpp:10-206 does), running the same per-lane codec
// source the GPU kernels run (../lane_codec.h compiled for the host).  This is
// an explicit user-selected mode, as in the reference; it is never a fallback
// of the GPU path.
#include "cpu_compressor.hpp"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "../lane_codec.h"
#include "file_header.hpp"
#include "packet_index.hpp"

namespace gip {

namespace {

const gpuar::RecipTable kRecip = gpuar::RecipTable();
const gpuar::DecodeConstTable kDecode = gpuar::DecodeConstTable();
constexpr size_t kBatchPackets = 4096;   // 32 MiB of input per batch

// one packet through the same three lane programs the GPU's encoder wavefronts run
size_t encode_one(const uint8_t *in, uint32_t len, uint8_t *slot) {
    uint16_t table[gpuar::kTreeRows];
    uint8_t *col = reinterpret_cast<uint8_t *>(table);
    gpuar::TopModeler<1> top;
    gpuar::LowModeler<1> low;
    top.open(col, 0, in[0]);
    low.open(col, 0, in[0]);
    gpuar::CarryCoderLane coder;     // (the carry form: the same bytes in fewer operations, lane_codec.h)
    coder.open(slot, 0);
    for (uint32_t i = 0; i < len; ++i) {
        const uint32_t x = in[i], next = i + 1 < len ? in[i + 1] : 0u;
        coder.step(top.step(x, 256u + i, next) + low.step(x, 256u + i, next), kRecip.r[i]);
    }
    bool overflowed = false;
    const uint32_t clen = coder.finish(len, overflowed);
    if (overflowed) throw std::runtime_error("a packet outgrew its 8704-byte slot");
    return clen;
}

// ... and the decoder lane program
size_t decode_one(const uint8_t *pkt, const uint8_t *limit, uint8_t *out) {
    alignas(16) uint8_t records[gpuar::kDecodeRecords * 16];
    gpuar::DecoderLane<3> dec;
    const size_t readable = static_cast<size_t>(limit - pkt);
    dec.open(records, pkt, 0, readable < 0x7FFFFFFFu ? static_cast<uint32_t>(readable) : 0x7FFFFFFFu, true);
    for (uint32_t i = 0; i < dec.ulen; ++i) dec.step(i, kDecode.c[i], out);
    dec.finish(out);
    if (dec.bad) throw std::runtime_error("Incorrect file format");
    return dec.ulen;
}

// f(p) for p in [0, n) on `threads` host threads.  The first exception wins and stops the others
// at their next packet.
template <typename F>
void for_each_packet(size_t n, unsigned threads, F &&f) {
    if (threads <= 1 || n < 2) {
        for (size_t p = 0; p < n; ++p) f(p);
        return;
    }
    std::vector<std::thread> pool;
    std::exception_ptr failure;
    std::mutex failure_lock;
    std::atomic<bool> stop{false};
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            try {
                for (size_t p = t; p < n && !stop.load(std::memory_order_relaxed); p += threads) f(p);
            } catch (...) {
                std::lock_guard<std::mutex> hold(failure_lock);
                if (!failure) failure = std::current_exception();
                stop.store(true, std::memory_order_relaxed);
            }
        });
    for (auto &th : pool) th.join();
    if (failure) std::rethrow_exception(failure);
}

}  // namespace

CPUCompressor::CPUCompressor() {}
CPUCompressor::~CPUCompressor() {}

CompressionInfo CPUCompressor::compress(ProgressMonitor *monitor) {
    CompressionInfo info;
    monitor->reset();
    process_timer.reset();
    io_timer.reset();
    io_timer.start();
    openFiles();
    info.uncompressedFileSize = getFileSize(openFile);
    if (std::fseek(saveFile, FileHeader::HEADER_LENGTH, SEEK_SET) != 0) throw std::runtime_error("Seek file failed");
    io_timer.stop();
    info.compressedFileSize = FileHeader::HEADER_LENGTH;

    const unsigned nthreads = threads ? threads : std::max(1u, std::thread::hardware_concurrency());
    std::vector<uint8_t> in(kBatchPackets * gpuar::kPacket + 16), slots(kBatchPackets * gpuar::kSlot);
    std::vector<uint32_t> clen(kBatchPackets);
    std::vector<uint16_t> all_clens;                   // for the optional index trailer
    try {
        for (;;) {
            io_timer.start();
            const size_t got = std::fread(in.data(), 1, kBatchPackets * gpuar::kPacket, openFile);
            io_timer.stop();
            if (got == 0) break;
            const size_t np = (got + gpuar::kPacket - 1) / gpuar::kPacket;
            process_timer.start();     // model init + codec only, as src/cpu_compressor.cpp:157-161
            for_each_packet(np, nthreads, [&](size_t p) {
                const size_t off = p * gpuar::kPacket;
                const uint32_t len = static_cast<uint32_t>(std::min<size_t>(gpuar::kPacket, got - off));
                clen[p] = static_cast<uint32_t>(encode_one(in.data() + off, len, slots.data() + p * gpuar::kSlot));
            });
            process_timer.stop();
            io_timer.start();
            for (size_t p = 0; p < np; ++p) {
                if (std::fwrite(slots.data() + p * gpuar::kSlot, clen[p], 1, saveFile) != 1)
                    throw std::runtime_error("Write data to file failed");
                info.compressedFileSize += clen[p];
                if (writeIndex) all_clens.push_back(static_cast<uint16_t>(clen[p]));
            }
            io_timer.stop();
            info.processedUncompressedSize += got;
            monitor->updateProgress(&info);
        }
        io_timer.start();
        if (writeIndex) PacketIndex::write(saveFile, all_clens);
        FileHeader header;
        header.setCompressedFileSize(info.compressedFileSize);
        header.setUncompressedFileSize(info.uncompressedFileSize);
        if (std::fseek(saveFile, 0, SEEK_SET) != 0) throw std::runtime_error("Seek file failed");
        if (std::fwrite(header.getData(), FileHeader::HEADER_LENGTH, 1, saveFile) != 1)
            throw std::runtime_error("Write data to file failed");
        closeFiles();
        io_timer.stop();
    } catch (...) {
        closeFiles();
        throw;
    }
    info.processTime = process_timer.value();
    info.ioTime = io_timer.value();
    return info;
}

CompressionInfo CPUCompressor::decompress(ProgressMonitor *monitor) {
    CompressionInfo info;
    monitor->reset();
    process_timer.reset();
    io_timer.reset();
    io_timer.start();
    openFiles();
    FileHeader header;
    const size_t fileSize = getFileSize(openFile);
    try {
        if (std::fread(header.getData(), FileHeader::HEADER_LENGTH, 1, openFile) != 1 || !header.checkHeaderVersion())
            throw std::runtime_error("Incorrect file format");
        info = header.getInfo(fileSize);
        const size_t stream_end = streamEnd(info, fileSize);
        std::vector<uint16_t> index;
        const bool indexed = PacketIndex::read(openFile, FileHeader::HEADER_LENGTH, stream_end, fileSize, index);
        io_timer.stop();

        // The stream is taken in windows of at most kBatchPackets packets, as GPUCompressor does:
        // memory stays bounded whatever the file size (the 64-bit header exists for 8-64 GiB inputs).
        // With an index the window's extent comes from the stored lengths; without one the bytes are
        // read first and `off += clen` is walked in memory (src/cpu_compressor.cpp:47-56 walks it
        // through the file), then the file is wound back to the end of the last whole packet.
        const unsigned nthreads = threads ? threads : std::max(1u, std::thread::hardware_concurrency());
        std::vector<uint8_t> window(kBatchPackets * gpuar::kSlot + 65536 + 16), out(kBatchPackets * gpuar::kPacket);
        std::vector<size_t> offsets(kBatchPackets + 1);
        std::vector<uint32_t> ulen(kBatchPackets);
        size_t file_pos = FileHeader::HEADER_LENGTH, next_packet = 0;
        while (file_pos < stream_end) {
            size_t np = 0, bytes = 0;
            io_timer.start();
            if (indexed) {
                offsets[0] = 0;
                while (next_packet < index.size() && np < kBatchPackets && bytes + index[next_packet] + 16 <= window.size()) {
                    const size_t c = index[next_packet++];
                    if (c < gpuar::kHdr) throw std::runtime_error("Incorrect file format");
                    bytes += c;
                    offsets[++np] = bytes;
                }
                if (!np) throw std::runtime_error("Incorrect file format");
                if (std::fread(window.data(), 1, bytes, openFile) != bytes) throw std::runtime_error("Invalid file length");
            } else {
                const size_t want = std::min(stream_end - file_pos, kBatchPackets * static_cast<size_t>(gpuar::kSlot) + 65536);
                if (std::fread(window.data(), 1, want, openFile) != want) throw std::runtime_error("Invalid file length");
                offsets[0] = 0;
                while (bytes < want && np < kBatchPackets) {
                    if (want - bytes < gpuar::kHdr) {
                        if (file_pos + want == stream_end) throw std::runtime_error("Incorrect file format");
                        break;                             // header cut by the window: next round
                    }
                    const size_t c = window[bytes] | (static_cast<size_t>(window[bytes + 1]) << 8);
                    if (c < gpuar::kHdr || file_pos + bytes + c > stream_end) throw std::runtime_error("Incorrect file format");
                    if (bytes + c > want) break;           // packet cut by the window: next round
                    bytes += c;
                    offsets[++np] = bytes;
                }
                if (!np) throw std::runtime_error("Incorrect file format");
                if (bytes != want && std::fseek(openFile, static_cast<long>(file_pos + bytes), SEEK_SET) != 0)
                    throw std::runtime_error("Seek file failed");
            }
            io_timer.stop();
            file_pos += bytes;
            process_timer.start();
            for_each_packet(np, nthreads, [&](size_t p) {
                ulen[p] = static_cast<uint32_t>(decode_one(window.data() + offsets[p], window.data() + bytes,
                                                           out.data() + p * gpuar::kPacket));
            });
            process_timer.stop();
            io_timer.start();
            for (size_t p = 0; p < np; ++p) {
                if (ulen[p] && std::fwrite(out.data() + p * gpuar::kPacket, ulen[p], 1, saveFile) != 1)
                    throw std::runtime_error("Write raw data to file failed");
                info.processedUncompressedSize += ulen[p];
            }
            io_timer.stop();
            monitor->updateProgress(&info);
        }
        info.uncompressedFileSize = info.processedUncompressedSize;     // what the packets held
        io_timer.start();
        closeFiles();
        io_timer.stop();
    } catch (...) {
        closeFiles();
        throw;
    }
    info.processTime = process_timer.value();
    info.ioTime = io_timer.value();
    return info;
}

}  // namespace gip
