// packet_index.hpp -- optional packet-offset index behind the packet stream of a .gip file
// (SURVEY.md section 8(f) row 2).
//
// The reference finds packet boundaries by walking `off += clen` through the file, one packet
// header at a time (/root/reference/src/gpu_compressor.cpp:299-312, src/cpu_compressor.cpp:47-56).
// With the index a reader cuts the stream into per-device ranges, reads each range with one bulk
// read and gets the packet offsets from a prefix sum.
//
// The index is a TRAILER: it follows the last packet and is not counted in the header's
// compressed-size field, which keeps meaning "20 + bytes of packets" (src/cpu_compressor.cpp:136,162).
// The reference stops at that size, so a file with an index still decodes there, and bytes 0..size of
// the file are exactly what is written without `--index`.  Layout, little-endian:
//     "GIPX"  u32 version = 1  u64 n_packets  |  u16 clen[n_packets]  |  zero pad to 8  |
//     u64 trailer_bytes (everything from "GIPX" to the end)  "XPIG"
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace gip {

class PacketIndex {
  public:
    static constexpr uint32_t kVersion = 1;

    // appends the trailer at the current position of `f`
    static void write(FILE *f, const std::vector<uint16_t> &clens) {
        const uint64_t n = clens.size();
        const uint64_t body = 2 * n, pad = (8 - body % 8) % 8, total = 16 + body + pad + 12;
        uint8_t head[16] = {'G', 'I', 'P', 'X'};
        put32(head + 4, kVersion);
        put64(head + 8, n);
        uint8_t tail[12];
        put64(tail, total);
        std::memcpy(tail + 8, "XPIG", 4);
        const uint8_t zeros[8] = {0};
        std::vector<uint8_t> le(body);
        for (uint64_t i = 0; i < n; ++i) le[2 * i] = static_cast<uint8_t>(clens[i]), le[2 * i + 1] = static_cast<uint8_t>(clens[i] >> 8);
        if (std::fwrite(head, sizeof head, 1, f) != 1 || (body && std::fwrite(le.data(), body, 1, f) != 1) ||
            (pad && std::fwrite(zeros, pad, 1, f) != 1) || std::fwrite(tail, sizeof tail, 1, f) != 1)
            throw std::runtime_error("Write packet index failed");
    }

    // Looks for a trailer in [stream_end, file_size); on success fills `clens`, restores the file
    // position and returns true.  A trailer whose lengths do not add up to the stream is rejected.
    static bool read(FILE *f, uint64_t stream_begin, uint64_t stream_end, uint64_t file_size, std::vector<uint16_t> &clens) {
        clens.clear();
        if (file_size < stream_end + 28) return false;
        const long here = std::ftell(f);
        bool ok = false;
        uint8_t tail[12], head[16];
        if (std::fseek(f, static_cast<long>(file_size - 12), SEEK_SET) == 0 && std::fread(tail, sizeof tail, 1, f) == 1 &&
            std::memcmp(tail + 8, "XPIG", 4) == 0 && get64(tail) == file_size - stream_end &&
            std::fseek(f, static_cast<long>(stream_end), SEEK_SET) == 0 && std::fread(head, sizeof head, 1, f) == 1 &&
            std::memcmp(head, "GIPX", 4) == 0 && get32(head + 4) == kVersion) {
            const uint64_t n = get64(head + 8), body = 2 * n, pad = (8 - body % 8) % 8;
            if (16 + body + pad + 12 == file_size - stream_end) {
                std::vector<uint8_t> le(body);
                if (!body || std::fread(le.data(), body, 1, f) == 1) {
                    clens.resize(n);
                    uint64_t sum = 0;
                    for (uint64_t i = 0; i < n; ++i) sum += clens[i] = static_cast<uint16_t>(le[2 * i] | (le[2 * i + 1] << 8));
                    ok = sum == stream_end - stream_begin;
                }
            }
        }
        if (!ok) clens.clear();
        std::fseek(f, here, SEEK_SET);
        return ok;
    }

  private:
    static void put32(uint8_t *p, uint32_t v) { for (int b = 0; b < 4; ++b) p[b] = static_cast<uint8_t>(v >> (8 * b)); }
    static void put64(uint8_t *p, uint64_t v) { for (int b = 0; b < 8; ++b) p[b] = static_cast<uint8_t>(v >> (8 * b)); }
    static uint32_t get32(const uint8_t *p) { uint32_t v = 0; for (int b = 0; b < 4; ++b) v |= static_cast<uint32_t>(p[b]) << (8 * b); return v; }
    static uint64_t get64(const uint8_t *p) { uint64_t v = 0; for (int b = 0; b < 8; ++b) v |= static_cast<uint64_t>(p[b]) << (8 * b); return v; }
};

}  // namespace gip
