// host_codec.cpp -- the host-callable packet codec behind include/gpuar_host.h.
//
// Scalar, caller-owned model state, no HIP.  Same results as the reference's
// arCompress / arDecompress / initializeAdaptiveProbabilityRangeList
// (/root/reference/src/gpuar_kernel.cu:487-531, 848-892, 403-419), reached a
// different way: the Fenwick array the caller hands in is unpacked into plain
// counts with 16-symbol block sums, the interval is renormalised in closed form
// (count the agreeing / straddling bits, shift once) instead of bit by bit,
// and bits move through a 64-bit window.  The adapted model is packed back into
// the caller's Fenwick array on return.
#include <stdint.h>
#include <string.h>

#include "gpuar_host.h"

namespace {

constexpr uint32_t kSymbols = 256;
constexpr uint32_t kBlock = 16;                  // symbols per block sum
constexpr uint32_t kHeader = 4;                  // PACKET_HEADER_LENGTH, src/gpu.h:14
constexpr uint32_t kPacketBytes = 8192;          // UNCOMPRESSED_PACKET_SIZE, src/gpu.h:13

inline uint32_t lowbit(uint32_t i) { return i & (0u - i); }
inline uint32_t clz16(uint32_t v) { return v ? static_cast<uint32_t>(__builtin_clz(v)) - 16u : 16u; }

// Order-0 adaptive counts.  below(x) = number of coded symbols < x (+ x initial ones).
struct Counts {
    uint16_t cnt[kSymbols];
    uint16_t blk[kSymbols / kBlock];
    uint32_t total;

    void unpack(const uint16_t *fenwick, uint32_t running_total) {
        uint32_t prefix[kSymbols + 1];
        prefix[0] = 0;
        for (uint32_t k = 1; k <= kSymbols; ++k) {
            uint32_t sum = 0;
            for (uint32_t i = k; i != 0; i &= i - 1) sum += fenwick[i];
            prefix[k] = sum & 0xFFFFu;           // the reference accumulates in u16 (getRange :215-227)
        }
        memset(blk, 0, sizeof blk);
        for (uint32_t s = 0; s < kSymbols; ++s) {
            cnt[s] = static_cast<uint16_t>(prefix[s + 1] - prefix[s]);
            blk[s / kBlock] = static_cast<uint16_t>(blk[s / kBlock] + cnt[s]);
        }
        total = running_total;
    }
    void pack(uint16_t *fenwick) const {
        uint32_t prefix[kSymbols + 1];
        prefix[0] = 0;
        for (uint32_t s = 0; s < kSymbols; ++s) prefix[s + 1] = prefix[s] + cnt[s];
        fenwick[0] = 0;
        for (uint32_t i = 1; i <= kSymbols; ++i) fenwick[i] = static_cast<uint16_t>(prefix[i] - prefix[i - lowbit(i)]);
    }
    uint32_t below(uint32_t x) const {
        uint32_t sum = 0;
        for (uint32_t b = 0; b < x / kBlock; ++b) sum += blk[b];
        for (uint32_t s = x & ~(kBlock - 1); s < x; ++s) sum += cnt[s];
        return sum & 0xFFFFu;
    }
    // the symbol whose [below, below + cnt) holds target; -1 if none (target >= sum of counts)
    int find(uint32_t target, uint32_t &cum_lo) const {
        uint32_t sum = 0, b = 0;
        for (; b < kSymbols / kBlock; ++b) {
            if (target < sum + blk[b]) break;
            sum += blk[b];
        }
        if (b == kSymbols / kBlock) return -1;
        uint32_t s = b * kBlock;
        while (target >= sum + cnt[s]) sum += cnt[s++];
        cum_lo = sum;
        return static_cast<int>(s);
    }
    void bump(uint32_t x) {
        ++cnt[x];
        ++blk[x / kBlock];
        ++total;
    }
};

// applySymbolRange :256-299 -- both ends from the old `lo`, truncating division, mod 2^16
inline void narrow(uint32_t &lo, uint32_t &hi, uint32_t cum_lo, uint32_t cum_hi, uint32_t total) {
    const uint32_t range = static_cast<uint32_t>(static_cast<int>(hi) - static_cast<int>(lo)) + 1u;
    const uint32_t up = (cum_hi * range) / total;
    const uint32_t dn = (cum_lo * range) / total;
    hi = (lo + (up & 0xFFFFu) - 1u) & 0xFFFFu;
    lo = (lo + (dn & 0xFFFFu)) & 0xFFFFu;
}

// How far one renormalisation moves: `agree` leading bits equal in lo and hi,
// then `straddle` positions where lo reads 1 and hi reads 0 below a 0/1 top bit.
struct Shift { uint32_t agree, straddle; };

inline Shift renormalise(uint32_t &lo, uint32_t &hi) {
    Shift s;
    s.agree = clz16(lo ^ hi);
    lo = (lo << s.agree) & 0xFFFFu;
    hi = ((hi << s.agree) | ((1u << s.agree) - 1u)) & 0xFFFFu;
    // now lo = 0..., hi = 1... (or the interval is degenerate and agree == 16: lo = 0, hi = 0xFFFF)
    s.straddle = clz16(~((lo & ~hi) << 1) & 0xFFFFu);
    if (s.straddle) {                            // each step drops the second bit and leaves the top bits 0 / 1
        lo = (lo << s.straddle) & 0x7FFFu;
        hi = (((hi << s.straddle) | ((1u << s.straddle) - 1u)) & 0xFFFFu) | 0x8000u;
    }
    return s;
}

// MSB-first bit sink (writeBit :128-151, putChar :76-84, writeClose :430-439)
struct BitSink {
    uint8_t *at;
    uint64_t window = 0;
    uint32_t held = 0;                           // bits in window, < 8 between calls

    void put(uint32_t bits, uint32_t count) {    // count <= 32, bits right-aligned
        window = (window << count) | (count == 32 ? bits : (bits & ((1u << count) - 1u)));
        held += count;
        while (held >= 8) {
            held -= 8;
            *at++ = static_cast<uint8_t>(window >> held);
        }
    }
    void run(uint32_t bit, uint32_t count) {
        const uint32_t fill = bit ? 0xFFFFFFFFu : 0u;
        for (; count >= 32; count -= 32) put(fill, 32);
        if (count) put(fill, count);
    }
    void close() {
        if (held) put(0, 8 - held);
    }
};

// MSB-first bit source; bytes at or past `end` read as zero (readBit :553-569 reads on)
struct BitSource {
    const uint8_t *at, *end;
    uint64_t window = 0;
    uint32_t held = 0;

    uint32_t take(uint32_t count) {              // count <= 32
        while (held < count) {
            window = (window << 8) | (at < end ? *at : 0u);
            ++at;
            held += 8;
        }
        held -= count;
        return static_cast<uint32_t>(window >> held) & (count == 32 ? 0xFFFFFFFFu : ((1u << count) - 1u));
    }
};

}  // namespace

extern "C" {

void initializeAdaptiveProbabilityRangeList(AdaptiveProbabilityRange *r, probability_t &cumProb) {
    r->ranges[0] = 0;
    for (uint32_t i = 1; i <= kSymbols; ++i) r->ranges[i] = static_cast<probability_t>(lowbit(i));
    cumProb = static_cast<probability_t>(kSymbols);
}

uint16_t arCompress(const uint8_t *fpIn, const uint16_t size, uint8_t *outFile, AdaptiveProbabilityRange &r,
                    probability_t &cumulativeProb) {
    Counts model;
    model.unpack(r.ranges, cumulativeProb);
    BitSink sink{outFile + kHeader};
    uint32_t lo = 0, hi = 0xFFFFu, pending = 0;

    for (uint32_t i = 0; i < size; ++i) {
        const uint32_t x = fpIn[i];
        const uint32_t cum_lo = model.below(x);
        narrow(lo, hi, cum_lo, (cum_lo + model.cnt[x]) & 0xFFFFu, model.total & 0xFFFFu);
        model.bump(x);
        const uint32_t settled = lo;             // its top `agree` bits are final
        const Shift s = renormalise(lo, hi);
        if (s.agree) {                           // writeEncodedBits :321-367
            const uint32_t first = settled >> 15;
            sink.put(first, 1);
            sink.run(first ^ 1u, pending);
            pending = 0;
            sink.put(settled >> (16 - s.agree), s.agree - 1);
        }
        pending += s.straddle;
    }
    // writeRemaining :379-388: the second bit of lo, then pending + 1 of its complement
    const uint32_t second = (lo >> 14) & 1u;
    sink.put(second, 1);
    sink.run(second ^ 1u, pending + 1);
    sink.close();

    const uint32_t clen = static_cast<uint32_t>(sink.at - outFile);
    outFile[0] = static_cast<uint8_t>(clen);     // u16 LE clen, u16 LE ulen (:525-528)
    outFile[1] = static_cast<uint8_t>(clen >> 8);
    outFile[2] = static_cast<uint8_t>(size);
    outFile[3] = static_cast<uint8_t>(size >> 8);
    model.pack(r.ranges);
    cumulativeProb = static_cast<probability_t>(model.total);
    return static_cast<uint16_t>(clen);
}

uint16_t arDecompress(const uint8_t *fpIn, const uint16_t inSize, uint8_t *fpOut, AdaptiveProbabilityRange &r,
                      probability_t &cumProb) {
    Counts model;
    model.unpack(r.ranges, cumProb);
    const uint32_t clen = static_cast<uint32_t>(fpIn[0]) | (static_cast<uint32_t>(fpIn[1]) << 8);
    const uint32_t ulen = static_cast<uint32_t>(fpIn[2]) | (static_cast<uint32_t>(fpIn[3]) << 8);
    // The reference ignores inSize and trusts the packet's own header (it reads whatever follows a
    // short packet and writes ulen bytes whatever the caller's buffer holds).  Here reads stop at the
    // caller's inSize when it is given (at clen otherwise), and a packet that claims more than one
    // packet's worth of output (> 8192 bytes) is refused -- the same guards as DecoderLane::open.
    if (ulen > kPacketBytes) return 0;
    const uint32_t readable = inSize ? inSize : clen;
    BitSource source{fpIn + kHeader, fpIn + (readable < kHeader ? kHeader : readable)};
    uint32_t lo = 0, hi = 0xFFFFu;
    uint32_t code = source.take(16);             // initializeDecoder :582-603
    uint32_t produced = 0;

    while (produced < ulen) {
        // getUnscaledCode :703-716
        const uint32_t range = static_cast<uint32_t>(static_cast<int>(hi) - static_cast<int>(lo)) + 1u;
        uint32_t target = static_cast<uint32_t>(static_cast<int>(code) - static_cast<int>(lo)) + 1u;
        target = (target * (model.total & 0xFFFFu) - 1u) / range;
        uint32_t cum_lo = 0;
        const int x = model.find(target & 0xFFFFu, cum_lo);
        if (x < 0) break;                        // :873-877
        fpOut[produced++] = static_cast<uint8_t>(x);
        narrow(lo, hi, cum_lo, (cum_lo + model.cnt[x]) & 0xFFFFu, model.total & 0xFFFFu);
        model.bump(static_cast<uint32_t>(x));
        const Shift s = renormalise(lo, hi);     // readEncodedBits :787-836
        const uint32_t moved = s.agree + s.straddle;
        if (moved) {
            code = ((code << moved) | source.take(moved)) & 0xFFFFu;
            if (s.straddle) code ^= 0x8000u;     // each straddle step flips the bit that becomes the top one
        }
    }
    model.pack(r.ranges);
    cumProb = static_cast<probability_t>(model.total);
    return static_cast<uint16_t>(produced);
}

}  // extern "C"
