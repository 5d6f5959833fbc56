// progress_monitor.hpp -- decile progress line, as src/progress_monitor.{hpp,cpp}.
#pragma once
#include <iostream>

#include "compress_info.hpp"

namespace gip {

class ProgressMonitor {
  public:
    ProgressMonitor() { reset(); }
    void reset() { currentRatio = 0; quiet = false; }
    void setQuiet(bool q) { quiet = q; }

    // prints "10%..", "20%.." ... whenever the decile changes (src/progress_monitor.cpp:17-33)
    void updateProgress(const CompressionInfo *info) {
        const unsigned last = static_cast<unsigned>(currentRatio * 100);
        currentRatio = info->uncompressedFileSize
                           ? static_cast<double>(info->processedUncompressedSize) / info->uncompressedFileSize
                           : 1.0;
        const unsigned now = static_cast<unsigned>(currentRatio * 100);
        if (!quiet && last / 10 != now / 10) {
            std::cout << now << "%.." << std::flush;
            if (now >= 100) std::cout << "Closing file..";
        }
    }

  private:
    double currentRatio;
    bool quiet;
};

}  // namespace gip
