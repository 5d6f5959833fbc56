// cpu_compressor.cpp -- host packet loop of `gpuar --host`
// (what src/cpu_compressor.cpp:10-206 does), running the same per-lane codec
// source the GPU kernels run (../lane_codec.h compiled for the host).  This is
// an explicit user-selected mode, as in the reference; it is never a fallback
// of the GPU path.
#include "cpu_compressor.hpp"

#include <algorithm>
#include <thread>
#include <vector>

#include "../lane_codec.h"
#include "file_header.hpp"
#include "packet_index.hpp"

namespace gip {

namespace {

const gpuar::RecipTable kRecip = gpuar::RecipTable();
constexpr size_t kBatchPackets = 4096;   // 32 MiB of input per batch

// one packet through the same three lane programs the GPU's encoder wavefronts run
size_t encode_one(const uint8_t *in, uint32_t len, uint8_t *slot) {
    uint16_t table[gpuar::kTreeRows];
    uint8_t *col = reinterpret_cast<uint8_t *>(table);
    gpuar::TopModeler<1> top;
    gpuar::LowModeler<1> low;
    top.open(col, 0, in[0]);
    low.open(col, 0, in[0]);
    gpuar::CoderLane coder;
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
    gpuar::DecoderLane<4> dec;
    const size_t readable = static_cast<size_t>(limit - pkt);
    dec.open(records, pkt, 0, readable < 0x7FFFFFFFu ? static_cast<uint32_t>(readable) : 0x7FFFFFFFu, true);
    for (uint32_t i = 0; i < dec.ulen; ++i) dec.step(i, kRecip.r[i], out);
    dec.finish(out);
    if (dec.bad) throw std::runtime_error("Incorrect file format");
    return dec.ulen;
}

template <typename F>
void for_each_packet(size_t n, unsigned threads, F &&f) {
    if (threads <= 1 || n < 2) {
        for (size_t p = 0; p < n; ++p) f(p);
        return;
    }
    std::vector<std::thread> pool;
    std::exception_ptr failure;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            try {
                for (size_t p = t; p < n; p += threads) f(p);
            } catch (...) {
                failure = std::current_exception();
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
        info = header.getInfo();
        const size_t n_stream = streamEnd(info, fileSize) - FileHeader::HEADER_LENGTH;
        std::vector<uint16_t> index;
        const bool indexed = PacketIndex::read(openFile, FileHeader::HEADER_LENGTH, FileHeader::HEADER_LENGTH + n_stream, fileSize, index);
        std::vector<uint8_t> stream(n_stream + 16);
        if (n_stream && std::fread(stream.data(), 1, n_stream, openFile) != n_stream)
            throw std::runtime_error("Invalid file length");
        io_timer.stop();

        std::vector<size_t> offsets;
        if (indexed) {                                 // prefix sum of the stored lengths (already checked against n_stream)
            offsets.reserve(index.size());
            size_t off = 0;
            for (uint16_t c : index) {
                if (c < gpuar::kHdr) throw std::runtime_error("Incorrect file format");
                offsets.push_back(off);
                off += c;
            }
        } else {                                       // the serial header walk of src/cpu_compressor.cpp:47-56: off += clen
            for (size_t off = 0; off < n_stream;) {
                if (n_stream - off < gpuar::kHdr) throw std::runtime_error("Incorrect file format");
                const size_t c = stream[off] | (static_cast<size_t>(stream[off + 1]) << 8);
                if (c < gpuar::kHdr || c > n_stream - off) throw std::runtime_error("Incorrect file format");
                offsets.push_back(off);
                off += c;
            }
        }
        const unsigned nthreads = threads ? threads : std::max(1u, std::thread::hardware_concurrency());
        std::vector<uint8_t> out(kBatchPackets * gpuar::kPacket);
        std::vector<uint32_t> ulen(kBatchPackets);
        for (size_t first = 0; first < offsets.size(); first += kBatchPackets) {
            const size_t np = std::min(kBatchPackets, offsets.size() - first);
            process_timer.start();
            for_each_packet(np, nthreads, [&](size_t p) {
                ulen[p] = static_cast<uint32_t>(decode_one(stream.data() + offsets[first + p], stream.data() + n_stream,
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
