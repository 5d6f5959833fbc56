// gpu_compressor.hpp -- host pipeline of the GPU path, API as
// src/gpu_compressor.hpp:8-39 (chooseDevice, getPacketSize, compress,
// decompress) plus useDevices() for the multi-GPU sharding the north star
// adds.  Internals are new: chunks of whole packets go down independent lanes
// (H2D straight out of the mapped input file, kernels, device-side compaction,
// D2H in pieces to one writer thread), several lanes per GPU, instead of the
// reference's per-packet memcpys on one thread (src/gpu_compressor.cpp:134-171).
#pragma once
#include <cstdint>
#include <mutex>
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

    // largest chunk a lane takes at a time, in packets (default 65536 = 512 MiB of input = 1024 wavefronts: one launch
    // fills the chip; the pinned staging is independent of it, 64 MiB pieces), kept a multiple of 64 (whole wavefronts);
    // a file is cut into chunks no larger than this, small enough that every lane of every device gets one
    void setBatchPackets(size_t n) { batchPackets = n < 64 ? 64 : n / 64 * 64; }

  private:
    struct DeviceBuffers;
    struct Failure;
    std::vector<int> devices;
    std::vector<DeviceBuffers *> buffers;      // one per lane, device-major
    size_t batchPackets = 65536;
    size_t chunkPackets = 0;                   // chunk size of the job the buffers were set up for
    std::vector<void *> epochs;                // one hipEvent_t per device: the time base of its lanes' kernel intervals
    std::mutex epochLock;

    void releaseBuffers();
    void ensureBuffers(size_t total_packets, bool compressing);
    template <typename Work, typename OnFailure>
    void runLanes(Work &&work, OnFailure &&on_failure);
    void finishTimes(CompressionInfo &info);
    void *epochOf(size_t device_index);
};

}  // namespace gip
