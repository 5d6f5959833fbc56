// compress_info.hpp -- statistics returned by Compressor::compress/decompress.
// Field names and units follow the reference (src/compress_info.hpp:11-16);
// times are milliseconds (common/helper_timer.h:330-342).
#pragma once
#include <cstddef>

namespace gip {

class CompressionInfo {
  public:
    double ratio = 0;
    double processTime = 0;   // ms inside the codec (GPU: kernels + sync; --host: codec calls)
    double ioTime = 0;        // ms everywhere else (file I/O, staging copies)
    size_t processedUncompressedSize = 0;
    size_t compressedFileSize = 0;
    size_t uncompressedFileSize = 0;
};

}  // namespace gip
