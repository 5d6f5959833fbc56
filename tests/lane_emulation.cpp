// tests/lane_emulation.cpp -- TEST HARNESS (CPU).  Runs the per-lane codec of
// gpuar_amd/csrc/lane_codec.h (the code the gfx950 kernels execute per lane)
// one packet at a time on the host, so its closed forms can be compared with
// the oracle in a container without a GPU.  Built by tests/test_lane_emulation.py
// into tests/_build/liblane_emulation.so; never linked into a product library.
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../gpuar_amd/csrc/lane_codec.h"

using namespace gpuar;

static const RecipTable kRecip = RecipTable();
static const DecodeConstTable kDecode = DecodeConstTable();

// What the three encoder wavefronts do (TopModeler + LowModeler + a coder), packet by packet.
// slots: ceil(n/8192) * 8704 bytes.  Returns the OR of per-packet overflow flags.
template <typename Coder>
static int encode_slots_with(const uint8_t *in, size_t n_bytes, uint8_t *slots)
{
    int any_overflow = 0;
    const size_t np = (n_bytes + kPacket - 1) / kPacket;
    std::vector<uint16_t> table(kTreeRows);
    for (size_t p = 0; p < np; ++p) {
        const size_t off = p * kPacket;
        const uint32_t len = static_cast<uint32_t>(n_bytes - off < kPacket ? n_bytes - off : kPacket);
        TopModeler<1> top;
        LowModeler<1> low;
        top.open(reinterpret_cast<uint8_t *>(table.data()), 0, in[off]);
        low.open(reinterpret_cast<uint8_t *>(table.data()), 0, in[off]);
        Coder coder;
        coder.open(slots, static_cast<uint32_t>(p * kSlot));
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t next = i + 1 < len ? in[off + i + 1] : 0u;
            const uint32_t cums = top.step(in[off + i], 256u + i, next) + low.step(in[off + i], 256u + i, next);
            coder.step(cums, kRecip.r[i]);
        }
        bool ov;
        coder.finish(len, ov);
        any_overflow |= ov ? 1 : 0;
    }
    return any_overflow;
}

// The GPU's store region and rare path, restated for the CPU.  On the device CarryCoderLane::shift_and_store is
// hand-written gfx950 text that does not call take_top() / leave(): it keeps `held` 16 too high (its shifts look at the low
// five bits only), stores `cache + over` unconditionally for every lane with a whole dword, finds the lanes for which
// that was not the whole story with ONE compare against `key` (0xFFFFFFFF, or 0 while dwords of ones wait), and then, for
// those lanes only: not decided (32 ones, no carry) -> rewind `at`, put the old cache back, count the dword; decided ->
// let the waiting dwords go out behind what was just stored; key = nff ? 0 : ~0.  This class is that rule statement by
// statement in plain C++, so that it can be compared with leave() on inputs that keep carrying without a GPU.
static uint64_t g_rule_undecided = 0, g_rule_filled = 0;      // how often the rare path ran (so a test can tell it was exercised)
struct DeviceRuleCoder : CarryCoderLane {
    uint32_t key = 0xFFFFFFFFu;
    void open_device(uint8_t *uniform_base, uint32_t slot_offset) {
        open(uniform_base, slot_offset);
        held = 16u;                                              // what open() does under __HIP_DEVICE_COMPILE__
    }
    void store_at(uint32_t word) { store32(base + (at < last ? at : last), bswap32(word)); }      // v_min_u32, v_perm_b32, global_store_dword
    void step_device(uint32_t cums, Recip rc) {
        const Ahead a = ahead(cums, rc);
        const uint64_t sum = ((static_cast<uint64_t>(wh) << 32) | wl) + a.dn;                   // narrow(): w += dn
        wl = static_cast<uint32_t>(sum), wh = static_cast<uint32_t>(sum >> 32);
        const uint32_t n = renorm_count(wl & 0xFFFFu, a.wd);
        held += n;                                               // settled_mask(): v_add_u32, v_cmp_le_u32 48, v_lshlrev_b32
        const bool full = held >= 48u;
        range = a.wd << n;
        const uint64_t w = ((static_cast<uint64_t>(wh) << 32) | wl) << n;                       // shift_and_store(): the 64-bit shift
        wl = static_cast<uint32_t>(w), wh = static_cast<uint32_t>(w >> 32);
        if (!full) return;                                       // s_and_saveexec: lanes without a whole dword sit the region out
        const uint32_t s = held & 31u;                           // (every shift takes `held` itself: the low five bits count)
        const uint32_t over = wh >> s;
        const uint32_t sent = cache + over;
        cache = GPUAR_ALIGNBIT(wh, wl, s);
        const bool rare = cache >= key;
        store_at(sent);
        wl = s ? (wl & ((1u << s) - 1u)) : 0u;                   // v_bfe_u32 wl, 0, held
        wh = 0;
        at += 4u;
        held -= 32u;
        if (!rare) return;                                       // s_cbranch_scc1 over the rare path
        const bool undecided = cache == 0xFFFFFFFFu && over == 0u;
        const uint32_t fill = over == 0u ? 0xFFFFFFFFu : 0u;
        if (undecided) {
            at -= 4u;
            cache = sent;
            ++nff;
            ++g_rule_undecided;
        } else {
            while (nff) {
                store_at(fill);
                at += 4u;
                --nff;
                ++g_rule_filled;
            }
        }
        key = nff ? 0u : 0xFFFFFFFFu;
    }
    uint32_t finish_device(uint32_t ulen, bool &overflowed) {
        held -= 16u;                                             // finish() under __HIP_DEVICE_COMPILE__
        return finish(ulen, overflowed);
    }
};

static int encode_slots_device_rule(const uint8_t *in, size_t n_bytes, uint8_t *slots)
{
    int any_overflow = 0;
    const size_t np = (n_bytes + kPacket - 1) / kPacket;
    std::vector<uint16_t> table(kTreeRows);
    for (size_t p = 0; p < np; ++p) {
        const size_t off = p * kPacket;
        const uint32_t len = static_cast<uint32_t>(n_bytes - off < kPacket ? n_bytes - off : kPacket);
        TopModeler<1> top;
        LowModeler<1> low;
        top.open(reinterpret_cast<uint8_t *>(table.data()), 0, in[off]);
        low.open(reinterpret_cast<uint8_t *>(table.data()), 0, in[off]);
        DeviceRuleCoder coder;
        coder.open_device(slots, static_cast<uint32_t>(p * kSlot));
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t next = i + 1 < len ? in[off + i + 1] : 0u;
            coder.step_device(top.step(in[off + i], 256u + i, next) + low.step(in[off + i], 256u + i, next), kRecip.r[i]);
        }
        bool ov;
        coder.finish_device(len, ov);
        any_overflow |= ov ? 1 : 0;
    }
    return any_overflow;
}

extern "C" {
// encode_kernel's coder with the DEVICE's store rule (DeviceRuleCoder above) instead of take_top() + leave()
int emu_encode_slots_device_rule(const uint8_t *in, size_t n_bytes, uint8_t *slots) { return encode_slots_device_rule(in, n_bytes, slots); }
// dwords of 32 ones that had to wait (low half) and waiting dwords written out once decided (high half), since the library was loaded
uint64_t emu_device_rule_rare_events(void) { return (g_rule_filled << 32) | (g_rule_undecided & 0xFFFFFFFFu); }


// encode_kernel's coder: the carry form (CarryCoderLane)
int emu_encode_slots(const uint8_t *in, size_t n_bytes, uint8_t *slots) { return encode_slots_with<CarryCoderLane>(in, n_bytes, slots); }
// the reference's own shape -- 16-bit bounds, bits owed while they straddle the midpoint -- in closed form (CoderLane;
// cut in two it is the latency kernel's coder, emu_encode_slots_split)
int emu_encode_slots_e3(const uint8_t *in, size_t n_bytes, uint8_t *slots) { return encode_slots_with<CoderLane>(in, n_bytes, slots); }

// The same the way the kernel's three roles run it: in phases of 8 symbols, the top modeler's parts first, the
// low modeler one phase later adding its own ONTO them -- without a look at the next phase (prime() at the start
// of a phase, step_last() for its last symbol) -- and the coder behind both.
int emu_encode_slots_phased(const uint8_t *in, size_t n_bytes, uint8_t *slots)
{
    int any_overflow = 0;
    const size_t np = (n_bytes + kPacket - 1) / kPacket;
    std::vector<uint16_t> table(kTreeRows);
    constexpr uint32_t kPhase = 8;
    for (size_t p = 0; p < np; ++p) {
        const size_t off = p * kPacket;
        const uint32_t len = static_cast<uint32_t>(n_bytes - off < kPacket ? n_bytes - off : kPacket);
        TopModeler<1> top;
        LowModeler<1> low;
        top.open(reinterpret_cast<uint8_t *>(table.data()), 0, in[off]);
        low.open(reinterpret_cast<uint8_t *>(table.data()), 0, 0);
        CarryCoderLane coder;
        coder.open(slots, static_cast<uint32_t>(p * kSlot));
        for (uint32_t base = 0; base < len; base += kPhase) {
            const uint32_t count = len - base < kPhase ? len - base : kPhase;
            uint32_t sums[kPhase];
            for (uint32_t j = 0; j < count; ++j) {
                const uint32_t i = base + j;
                sums[j] = top.step(in[off + i], 256u + i, i + 1 < len ? in[off + i + 1] : 0u);
            }
            // (encode_kernel's low modeler gets the symbols as row tags, the way the top modeler formed them)
            uint32_t tags[kPhase];
            for (uint32_t j = 0; j < count; ++j) tags[j] = low.tree.tag(in[off + base + j]) & 0xFFFFu;
            low.prime_tag(tags[0]);
            for (uint32_t j = 0; j < count; ++j) {
                const uint32_t i = base + j;
                sums[j] = j + 1 < count ? low.step_tag(tags[j], 256u + i, tags[j + 1], sums[j])
                                        : low.step_last_tag(tags[j], 256u + i, sums[j]);
            }
            for (uint32_t j = 0; j < count; ++j) coder.step(sums[j], kRecip.r[base + j]);
        }
        bool ov;
        coder.finish(len, ov);
        any_overflow |= ov ? 1 : 0;
    }
    return any_overflow;
}

// The latency-mode encoder's five roles (encode_small_kernel): the tree dealt 3 + 3 + (0, 7, tail) to three modelers
// that add their parts onto each other in phases of 16 symbols, the carry-form coder cut into CarryIntervalLane and
// CarrySinkLane joined by one word per symbol (dn | n << 16).
int emu_encode_slots_split(const uint8_t *in, size_t n_bytes, uint8_t *slots)
{
    int any_overflow = 0;
    const size_t np = (n_bytes + kPacket - 1) / kPacket;
    std::vector<uint16_t> table(kTreeRows);
    constexpr uint32_t kPhase = 16;
    for (size_t p = 0; p < np; ++p) {
        const size_t off = p * kPacket;
        const uint32_t len = static_cast<uint32_t>(n_bytes - off < kPacket ? n_bytes - off : kPacket);
        PartialModeler<1, 1, 3, 0, false> upper;
        PartialModeler<1, 4, 3, 0, false> middle;
        DeepestModeler<1> low;
        uint8_t *t = reinterpret_cast<uint8_t *>(table.data());
        upper.open(t, 0, in[off]);
        middle.open(t, 0, 0);
        low.open(t, 0, 0);
        CarryIntervalLane interval;
        interval.open();
        CarrySinkLane sink;
        sink.open(slots, static_cast<uint32_t>(p * kSlot));
        for (uint32_t base = 0; base < len; base += kPhase) {
            const uint32_t count = len - base < kPhase ? len - base : kPhase;
            uint32_t sums[kPhase];
            for (uint32_t j = 0; j < count; ++j) {
                const uint32_t i = base + j;
                sums[j] = upper.step(in[off + i], 256u + i, i + 1 < len ? in[off + i + 1] : 0u);
            }
            middle.prime(in[off + base]);
            for (uint32_t j = 0; j < count; ++j) {
                const uint32_t i = base + j;
                sums[j] = j + 1 < count ? middle.step(in[off + i], 256u + i, in[off + i + 1], sums[j])
                                        : middle.step_last(in[off + i], 256u + i, sums[j]);
            }
            low.prime(in[off + base]);
            for (uint32_t j = 0; j < count; ++j) {
                const uint32_t i = base + j;
                sums[j] = j + 1 < count ? low.step(in[off + i], 256u + i, in[off + i + 1], sums[j])
                                        : low.step_last(in[off + i], 256u + i, sums[j]);
            }
            uint32_t words[kPhase];
            for (uint32_t j = 0; j < count; ++j) words[j] = interval.step(sums[j], kRecip.r[base + j]);
            for (uint32_t j = 0; j < count; ++j) sink.take(words[j]);
        }
        bool ov;
        sink.finish(len, ov);
        any_overflow |= ov ? 1 : 0;
    }
    return any_overflow;
}

// What a decoder wavefront's lane does (SubtreeModel + DecoderLane).
// pkt_offsets: np+1 byte offsets into `stream`; out: np * 8192 bytes.
// Returns the number of packets flagged bad.
int emu_decode_stream(const uint8_t *stream, const uint64_t *pkt_offsets, size_t np, uint8_t *out)
{
    int bad = 0;
    std::vector<uint64_t> records(kDecodeRecords * 2);       // 16 bytes each, 8-byte aligned storage
    const uint8_t *limit = stream + pkt_offsets[np];
    for (size_t p = 0; p < np; ++p) {
        DecoderLane<3> dec;
        uint8_t *o = out + p * kPacket;
        const uint64_t readable = static_cast<uint64_t>(limit - (stream + pkt_offsets[p]));
        dec.open(reinterpret_cast<uint8_t *>(records.data()), stream + pkt_offsets[p], 0,
                 readable < 0x7FFFFFFFu ? static_cast<uint32_t>(readable) : 0x7FFFFFFFu, true);
        for (uint32_t i = 0; i < dec.ulen; ++i) dec.step(i, kDecode.c[i], o);
        dec.finish(o);
        bad += dec.bad ? 1 : 0;
    }
    return bad;
}

// renorm_count(a, wd) against the reference's renormalisation loop (writeEncodedBits :321-367 / readEncodedBits :787-836:
// shift while the top bits agree, or while lo = 01.. and hi = 10..), for every a and every `stride`-th h >= a plus the
// neighbours of a and of 0xFFFF.  stride 1 = all 2^31 pairs (8 s).  Returns the number of mismatches.
uint64_t emu_check_renorm_count(uint32_t stride)
{
    uint64_t wrong = 0;
    auto by_loop = [](uint32_t a, uint32_t h) {
        uint16_t lo = static_cast<uint16_t>(a), hi = static_cast<uint16_t>(h);
        uint32_t n = 0;
        for (;;) {
            if (((hi ^ lo) & 0x8000u) == 0) {
            } else if ((lo & 0x4000u) && !(hi & 0x4000u)) {
                lo &= 0x3FFFu;
                hi |= 0x4000u;
            } else {
                return n;
            }
            lo = static_cast<uint16_t>(lo << 1);
            hi = static_cast<uint16_t>((hi << 1) | 1u);
            ++n;
        }
    };
    for (uint32_t a = 0; a < 65536u; ++a) {
        for (uint32_t h = a; h < 65536u; h += stride) wrong += renorm_count(a, h - a + 1u) != by_loop(a, h);
        for (uint32_t h = a; h < 65536u && h < a + 4u; ++h) wrong += renorm_count(a, h - a + 1u) != by_loop(a, h);
        for (uint32_t h = 65535u; h >= a && h + 4u > 65535u; --h) {
            wrong += renorm_count(a, h - a + 1u) != by_loop(a, h);
            if (h == 0) break;
        }
    }
    return wrong;
}

// Check of the reciprocal table: for every total d, the multiple boundaries
// k*d-1 and k*d (stepping k by `stride`) plus the largest numerator
// d*65536-1.  Returns the number of mismatches.
uint64_t emu_check_recip(uint32_t stride)
{
    uint64_t wrong = 0;
    for (uint32_t i = 0; i < kPacket; ++i) {
        const uint32_t d = 256u + i;
        const uint32_t top = d * 65536u - 1u;
        for (uint64_t k = 1; k * d <= top; k += stride) {
            const uint32_t a = static_cast<uint32_t>(k * d), b = a - 1u;
            wrong += div_total(a, kRecip.r[i]) != a / d;
            wrong += div_total(b, kRecip.r[i]) != b / d;
        }
        wrong += div_total(top, kRecip.r[i]) != top / d;
        wrong += div_total(0, kRecip.r[i]) != 0;
    }
    return wrong;
}

}  // extern "C"
