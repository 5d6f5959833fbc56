// cpu_compressor.hpp -- the explicit `--host` mode (src/cpu_compressor.hpp:9-16).
#pragma once
#include "compressor.hpp"

namespace gip {

class CPUCompressor : public Compressor {
  public:
    CPUCompressor();
    ~CPUCompressor() override;
    CompressionInfo compress(ProgressMonitor *monitor) override;
    CompressionInfo decompress(ProgressMonitor *monitor) override;

    // packets are independent: 0 = one thread per hardware thread, 1 = the
    // reference's single-threaded behaviour (default)
    void setThreads(unsigned n) { threads = n; }

  private:
    unsigned threads = 1;
};

}  // namespace gip
