// compressor.hpp -- abstract base of the two compressors, API as
// src/compressor.hpp:10-65 (setOpenFileName, setSaveFileName, compress,
// decompress, closeFiles, getFileSize, generateRandomFile).  Unlike the
// reference's base class (src/compressor.cpp:23-25, cudaMallocHost) it touches
// no GPU runtime, so `--host` works on a machine without one.
#pragma once
#include <chrono>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "compress_info.hpp"
#include "progress_monitor.hpp"

namespace gip {

// millisecond stopwatch with start/stop accumulation (the role of StopWatchInterface)
class StopWatch {
  public:
    void reset() { total_ms = 0; }
    void start() { t0 = clock::now(); }
    void stop() { total_ms += std::chrono::duration<double, std::milli>(clock::now() - t0).count(); }
    double value() const { return total_ms; }

  private:
    using clock = std::chrono::steady_clock;
    clock::time_point t0;
    double total_ms = 0;
};

class Compressor {
  protected:
    std::string openFileName;
    std::string saveFileName;
    StopWatch process_timer;
    StopWatch io_timer;
    FILE *openFile = nullptr;
    FILE *saveFile = nullptr;
    bool writeIndex = false;          // append the packet-offset index trailer (packet_index.hpp)

    // where the packets of an open .gip end: the header's size field when it is sane (the reference's
    // reading, src/cpu_compressor.cpp:47-56), else the end of the file
    static size_t streamEnd(const CompressionInfo &info, size_t fileSize) {
        const size_t claimed = info.compressedFileSize;
        return claimed >= 20 && claimed <= fileSize ? claimed : fileSize;
    }

    // throws std::runtime_error naming the file.  truncate_output = false: an existing output file keeps its pages
    // (dropping 8 GiB of page cache costs most of a second); the caller sets the final length itself
    void openFiles(bool truncate_output = true);

  public:
    Compressor();
    virtual ~Compressor();

    size_t getFileSize(FILE *stream);
    void setOpenFileName(const std::string &fileName) { openFileName = fileName; }
    void setSaveFileName(const std::string &fileName) { saveFileName = fileName; }
    void setWriteIndex(bool on) { writeIndex = on; }
    virtual CompressionInfo compress(ProgressMonitor *monitor) = 0;
    virtual CompressionInfo decompress(ProgressMonitor *monitor) = 0;
    void closeFiles();
    void generateRandomFile(const size_t size);
};

}  // namespace gip
