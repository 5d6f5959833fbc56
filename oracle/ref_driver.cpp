// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin C-ABI driver around the REFERENCE's own host-callable codec functions
// (arCompress / arDecompress / initializeAdaptiveProbabilityRangeList,
// declared extern "C" in /root/reference/src/gpuar.h:73,75,76 and defined in
// /root/reference/src/gpuar_kernel.cu:403-419, 487-531, 848-892).  It is
// linked by oracle/build_ref.sh against an object compiled from that file
// where it lies; no reference source is copied into this repository.
//
// The loops below are this repo's own restatement of what the reference's
// --host path does around those calls (src/cpu_compressor.cpp:144-173 encode,
// :47-78 decode): model re-initialised per packet, packets back to back.
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <thread>
#include <vector>
#include "gpuar.h"   // the reference header, found via -I/root/reference/src

namespace {
const size_t kIn = UNCOMPRESSED_PACKET_SIZE;   // 8192, src/gpu.h:13
const size_t kSlot = COMPRESSED_PACKET_SIZE;   // 8704, src/gpu.h:12
}

extern "C" {

size_t ref_encode_packet(const uint8_t *in, uint16_t n, uint8_t *out)
{
    // arCompress reads its input as 16-byte ulonglong2 elements and may touch
    // up to 15 bytes past `n` (src/gpuar_kernel.cu:496-517): give it a padded,
    // aligned private copy so the driver never reads outside the caller's buffer.
    static __thread unsigned char staged[kIn + 32] __attribute__((aligned(16)));
    AdaptiveProbabilityRange model;
    probability_t total;
    memset(staged, 0, sizeof staged);
    memcpy(staged, in, n);
    initializeAdaptiveProbabilityRangeList(&model, total);
    return arCompress(staged, n, out, model, total);
}

size_t ref_decode_packet(const uint8_t *pkt, size_t avail, uint8_t *out)
{
    // arDecompress has no bound on its bit reads (readBit :553-569); stage the
    // packet into a zero-padded buffer so trailing reads are defined.
    static __thread unsigned char staged[2 * kSlot + 64];
    AdaptiveProbabilityRange model;
    probability_t total;
    size_t clen = (size_t)pkt[0] | ((size_t)pkt[1] << 8);
    if (clen > avail) clen = avail;
    if (clen > 2 * kSlot) return 0;
    memset(staged, 0, sizeof staged);
    memcpy(staged, pkt, clen);
    initializeAdaptiveProbabilityRangeList(&model, total);
    return arDecompress(staged, (uint16_t)clen, out, model, total);
}

size_t ref_encode_stream(const uint8_t *in, size_t n_bytes, uint8_t *out)
{
    unsigned char tmp[2 * kIn + 64];
    size_t total = 0;
    for (size_t off = 0; off < n_bytes; off += kIn) {
        size_t n = n_bytes - off < kIn ? n_bytes - off : kIn;
        size_t len = ref_encode_packet(in + off, (uint16_t)n, tmp);
        memcpy(out + total, tmp, len);
        total += len;
    }
    return total;
}

size_t ref_decode_stream(const uint8_t *stream, size_t n_stream, uint8_t *out)
{
    size_t off = 0, produced = 0;
    while (off + 4 <= n_stream) {
        size_t clen = (size_t)stream[off] | ((size_t)stream[off + 1] << 8);
        if (clen < 4 || clen > n_stream - off) return (size_t)-1;
        produced += ref_decode_packet(stream + off, clen, out + produced);
        off += clen;
    }
    return produced;
}

// ---- the same two loops with the packets fanned out over native threads, one contiguous packet range per
// thread (packets are independent: every one starts from a fresh model).  bench.py's all-cores CPU row.
size_t ref_encode_stream_mt(const uint8_t *in, size_t n_bytes, uint8_t *out, unsigned threads)
{
    const size_t n_packets = (n_bytes + kIn - 1) / kIn;
    if (threads < 1) threads = 1;
    if (threads > n_packets) threads = n_packets ? (unsigned)n_packets : 1;
    const size_t per = (n_packets + threads - 1) / threads;
    std::vector<std::vector<uint8_t> > seg(threads);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            const size_t p0 = t * per, p1 = (t + 1) * per < n_packets ? (t + 1) * per : n_packets;
            if (p0 >= p1) return;
            const size_t b0 = p0 * kIn, b1 = p1 * kIn < n_bytes ? p1 * kIn : n_bytes;
            seg[t].resize((p1 - p0) * kSlot + 64);
            seg[t].resize(ref_encode_stream(in + b0, b1 - b0, seg[t].data()));
        });
    for (auto &th : pool) th.join();
    size_t total = 0;
    for (unsigned t = 0; t < threads; ++t) {
        memcpy(out + total, seg[t].data(), seg[t].size());
        total += seg[t].size();
    }
    return total;
}

size_t ref_decode_stream_mt(const uint8_t *stream, size_t n_stream, uint8_t *out, unsigned threads)
{
    std::vector<size_t> at;                                  // the header walk (`off += clen`), serial and cheap
    size_t off = 0;
    while (off + 4 <= n_stream) {
        const size_t clen = (size_t)stream[off] | ((size_t)stream[off + 1] << 8);
        if (clen < 4 || clen > n_stream - off) return (size_t)-1;
        at.push_back(off);
        off += clen;
    }
    at.push_back(off);
    const size_t n_packets = at.size() - 1;
    if (threads < 1) threads = 1;
    if (threads > n_packets) threads = n_packets ? (unsigned)n_packets : 1;
    const size_t per = (n_packets + threads - 1) / threads;
    std::vector<size_t> produced(threads, 0);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            const size_t p0 = t * per, p1 = (t + 1) * per < n_packets ? (t + 1) * per : n_packets;
            // every packet but the stream's last holds 8192 bytes: packet p decodes to out + p * 8192
            for (size_t p = p0; p < p1; ++p) produced[t] += ref_decode_packet(stream + at[p], at[p + 1] - at[p], out + p * kIn);
        });
    for (auto &th : pool) th.join();
    size_t total = 0;
    for (size_t v : produced) total += v;
    return total;
}

// ---- model-explicit variants: the caller owns the model state, exactly as the
// reference's own callers do (src/cpu_compressor.cpp:59-60,159-160; src/main.cpp:27-43).
// `ranges` is the 257-entry Fenwick array of AdaptiveProbabilityRange (src/gpuar.h:42-48).
void ref_model_init(uint16_t *ranges, uint16_t *total)
{
    AdaptiveProbabilityRange model;
    probability_t t;
    initializeAdaptiveProbabilityRangeList(&model, t);
    memcpy(ranges, model.ranges, sizeof model.ranges);
    *total = t;
}

size_t ref_encode_packet_model(const uint8_t *in, uint16_t n, uint8_t *out, uint16_t *ranges, uint16_t *total)
{
    static __thread unsigned char staged[kIn + 32] __attribute__((aligned(16)));
    AdaptiveProbabilityRange model;
    probability_t t = *total;
    memcpy(model.ranges, ranges, sizeof model.ranges);
    memset(staged, 0, sizeof staged);
    memcpy(staged, in, n);
    size_t len = arCompress(staged, n, out, model, t);
    memcpy(ranges, model.ranges, sizeof model.ranges);
    *total = t;
    return len;
}

size_t ref_decode_packet_model(const uint8_t *pkt, size_t avail, uint8_t *out, uint16_t *ranges, uint16_t *total)
{
    static __thread unsigned char staged[2 * kSlot + 64];
    AdaptiveProbabilityRange model;
    probability_t t = *total;
    size_t clen = (size_t)pkt[0] | ((size_t)pkt[1] << 8);
    if (clen > avail) clen = avail;
    if (clen > 2 * kSlot) return 0;
    memcpy(model.ranges, ranges, sizeof model.ranges);
    memset(staged, 0, sizeof staged);
    memcpy(staged, pkt, clen);
    size_t n = arDecompress(staged, (uint16_t)clen, out, model, t);
    memcpy(ranges, model.ranges, sizeof model.ranges);
    *total = t;
    return n;
}

}  // extern "C"
