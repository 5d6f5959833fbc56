// gpu_compressor.hpp -- host pipeline of the GPU path, API as
// src/gpu_compressor.hpp:8-39 (chooseDevice, getPacketSize, compress,
// decompress) plus useDevices() for the multi-GPU sharding the north star
// adds.  Internals are new: bulk pinned transfers, device-side compaction and
// one host thread per GPU instead of the reference's per-packet memcpys
// (src/gpu_compressor.cpp:134-171).
#pragma once
#include <cstdint>
#include <vector>

#include "compressor.hpp"

namespace gip {

class GPUCompressor : public Compressor {
  public:
    GPUCompressor();                 // throws if no HIP device is visible: there is no CPU fallback
    ~GPUCompressor() override;
    CompressionInfo compress(ProgressMonitor *monitor) override;
    CompressionInfo decompress(ProgressMonitor *monitor) override;

    void chooseDevice(const int id);     // run on exactly this device
    void useDevices(const int count);    // shard packets over devices 0..count-1
    int deviceCount() const { return static_cast<int>(devices.size()); }

    // u16 LE packet length at the start of a packet (src/gpu_compressor.hpp:33-36)
    static unsigned short getPacketSize(const uint8_t *packet) {
        return static_cast<unsigned short>(packet[0] | (packet[1] << 8));
    }

    // packets each device takes per round (default 32768 = 256 MiB of input), kept a multiple of 64
    // (whole wavefronts: compress() deals every device a multiple of 64 packets, which must fit its
    // buffers); rounds are double-buffered, so file reads, GPU work and file writes of neighbouring rounds overlap
    void setBatchPackets(size_t n) { batchPackets = n < 64 ? 64 : n / 64 * 64; }

  private:
    struct DeviceBuffers;
    std::vector<int> devices;
    std::vector<DeviceBuffers *> buffers;
    size_t batchPackets = 32768;

    void releaseBuffers();
    void ensureBuffers(size_t total_packets);
};

}  // namespace gip
