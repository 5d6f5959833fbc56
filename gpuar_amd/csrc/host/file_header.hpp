// file_header.hpp -- the 20-byte .gip container header (src/file_header.hpp:15-78).
//
//   [0..2]   version 0.1.0
//   [3]      0            (uninitialised in the reference)
//   [4..11]  uncompressed size, little-endian u64   (reference: u32 at [4..7], [8..11] uninitialised)
//   [12..19] compressed file size INCLUDING this header, little-endian u64
//            (reference: u32 at [12..15], [16..19] uninitialised)
//
// For files under 4 GiB bytes 0-2, 4-7 and 12-15 are exactly what the reference
// writes; the other bytes are garbage there and zero here.  Writing the full
// 64-bit value keeps 8 GiB / 64 GiB inputs representable (the reference's
// reader, src/file_header.hpp:50-51, overflows at 2 GiB).
#pragma once
#include <cstdint>
#include <cstring>

#include "compress_info.hpp"

namespace gip {

class FileHeader {
  public:
    constexpr static int VERSION_POSITION = 0;
    constexpr static int UNCOMPRESSED_FILE_SIZE_POSITION = 4;
    constexpr static int COMPRESSED_FILE_SIZE_POSITION = 12;
    constexpr static int HEADER_LENGTH = 20;
    constexpr static unsigned char GLZ_VERSION_MAJOR = 0;
    constexpr static unsigned char GLZ_VERSION_MINOR = 1;
    constexpr static unsigned char GLZ_VERSION_REVISION = 0;

    FileHeader() {
        std::memset(data, 0, sizeof data);
        data[0] = GLZ_VERSION_MAJOR;
        data[1] = GLZ_VERSION_MINOR;
        data[2] = GLZ_VERSION_REVISION;
    }
    void *getData() { return data; }
    void setData(const void *src) { std::memcpy(data, src, HEADER_LENGTH); }

    // `fileSize` = bytes in the .gip file being read (0: unknown).  The reference writes only the
    // low 4 bytes of each size and leaves the 4 above them uninitialised, so a 64-bit value is
    // believed only when the file could hold it: the packet stream ends inside the file, and the
    // stream is long enough for that many bytes (a packet of <= 8192 bytes takes >= 4).  Otherwise
    // the low 32 bits are what the writer meant.
    CompressionInfo getInfo(uint64_t fileSize = 0) const {
        CompressionInfo info;
        uint64_t unc = get64(UNCOMPRESSED_FILE_SIZE_POSITION), comp = get64(COMPRESSED_FILE_SIZE_POSITION);
        if (fileSize) {
            if (comp > fileSize) comp &= 0xFFFFFFFFull;
            const uint64_t stream = (comp >= HEADER_LENGTH && comp <= fileSize ? comp : fileSize) - HEADER_LENGTH;
            if (unc > (stream / 4 + 1) * 8192) unc &= 0xFFFFFFFFull;
        }
        info.uncompressedFileSize = static_cast<size_t>(unc);
        info.compressedFileSize = static_cast<size_t>(comp);
        return info;
    }
    void setUncompressedFileSize(size_t size) { put64(UNCOMPRESSED_FILE_SIZE_POSITION, size); }
    void setCompressedFileSize(size_t size) { put64(COMPRESSED_FILE_SIZE_POSITION, size); }
    bool checkHeaderVersion() const {
        return data[0] == GLZ_VERSION_MAJOR && data[1] == GLZ_VERSION_MINOR && data[2] == GLZ_VERSION_REVISION;
    }

  private:
    unsigned char data[HEADER_LENGTH];
    uint64_t get64(int at) const {
        uint64_t v = 0;
        for (int b = 7; b >= 0; --b) v = (v << 8) | data[at + b];
        return v;
    }
    void put64(int at, uint64_t v) {
        for (int b = 0; b < 8; ++b) data[at + b] = static_cast<unsigned char>(v >> (8 * b));
    }
};

}  // namespace gip
