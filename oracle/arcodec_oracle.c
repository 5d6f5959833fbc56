/*
 * oracle/arcodec_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU-only restatement of the packet arithmetic codec of the
 * reference (jiahansu/GPUAR, /root/reference/src/gpuar_kernel.cu).  It is the
 * checker the parity tests, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg compare the HIP path against.  Nothing under gpuar_amd/
 * may link, import or call it.
 *
 * Parity status: PINNED.  This file is checked (tests/test_oracle_golden.py)
 *   (1) against the known-answer vectors and packet-stream md5s recorded from
 *       the reference's own --host path, committed as
 *       tests/golden/survey_vectors.json, and
 *   (2) in this container, byte-for-byte against oracle/_ref (the reference's
 *       unmodified arCompress/arDecompress compiled from /root/reference, see
 *       oracle/build_ref.sh) on seeded inputs, and against the fixtures that
 *       build emitted (tests/golden/, made by tests/golden/make_golden.py).
 *
 * The restatement deliberately keeps the reference's *structure* (Fenwick
 * model, bit-at-a-time renormalisation loop, byte-at-a-time bit I/O) so that
 * it is an independent check of the closed forms the HIP kernels use.
 * Every function cites the reference lines it follows.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define AR_SYMBOLS        256u
#define AR_PACKET_IN      8192u   /* src/gpu.h:13  UNCOMPRESSED_PACKET_SIZE */
#define AR_PACKET_SLOT    8704u   /* src/gpu.h:12  COMPRESSED_PACKET_SIZE   */
#define AR_PACKET_HDR     4u      /* src/gpu.h:14  PACKET_HEADER_LENGTH     */
#define AR_TOP            0x8000u /* src/gpuar.h:26 MASK_BIT(0) */
#define AR_SECOND         0x4000u /* src/gpuar.h:26 MASK_BIT(1) */

/* ---- adaptive order-0 model: Fenwick tree over 1..256 --------------------
 * src/gpuar.h:42-48 (AdaptiveProbabilityRange), src/gpuar_kernel.cu:205-238 */
typedef struct {
    uint16_t fen[AR_SYMBOLS + 1]; /* fen[0] unused */
    uint16_t total;               /* cumulativeProb  */
} ar_model;

/* sum of counts of symbols < k, k in 0..256   (getRange, :215-227) */
static uint16_t model_below(const ar_model *m, unsigned k)
{
    uint16_t acc = 0;
    for (; k != 0; k &= k - 1)
        acc = (uint16_t)(acc + m->fen[k]);
    return acc;
}

/* count of symbol s += 1   (update(UPPER(s)), :229-238) */
static void model_bump(ar_model *m, unsigned s)
{
    for (unsigned k = s + 1; k <= AR_SYMBOLS; k += k & (0u - k))
        m->fen[k]++;
}

/* every symbol starts with count 1, total 256
 * (initializeAdaptiveProbabilityRangeList, :403-419) */
static void model_reset(ar_model *m)
{
    memset(m, 0, sizeof *m);
    for (unsigned s = 0; s < AR_SYMBOLS; s++) {
        model_bump(m, s);
        m->total++;
    }
}

/* ---- interval narrowing, shared by both directions ------------------------
 * applySymbolRange, src/gpuar_kernel.cu:256-299 */
static void narrow(ar_model *m, unsigned s, uint16_t *lo, uint16_t *hi)
{
    uint32_t span = (uint32_t)((int)*hi - (int)*lo) + 1u;        /* :269 (int promotion, as there) */
    uint32_t up   = (uint32_t)model_below(m, s + 1) * span / m->total; /* :272-273 */
    uint32_t dn   = (uint32_t)model_below(m, s)     * span / m->total; /* :279-280 */
    uint16_t base = *lo;
    *hi = (uint16_t)(base + (uint16_t)up - 1u);                 /* :276 */
    *lo = (uint16_t)(base + (uint16_t)dn);                      /* :283 */
    m->total++;                                                 /* :286 */
    model_bump(m, s);                                           /* :288 */
}

/* ---- MSB-first bit sink ----------------------------------------------------
 * BitPointer src/gpuar.h:50-57, writeBit :128-151, putChar :76-84 */
typedef struct {
    uint8_t *at;
    uint8_t  held;
    uint8_t  nheld;
} bit_sink;

static void sink_bit(bit_sink *w, int bit)
{
    w->held = (uint8_t)((w->held << 1) | (bit ? 1 : 0));
    if (++w->nheld == 8) {
        *w->at++ = w->held;
        w->held = 0;
        w->nheld = 0;
    }
}

/* writeEncodedBits, :321-367 */
static void encoder_renorm(bit_sink *w, uint16_t *lo, uint16_t *hi, uint16_t *pending)
{
    for (;;) {
        if (((*hi ^ *lo) & AR_TOP) == 0) {
            int b = (*hi & AR_TOP) != 0;
            sink_bit(w, b);
            while (*pending) {
                sink_bit(w, !b);
                (*pending)--;
            }
        } else if ((*lo & AR_SECOND) && !(*hi & AR_SECOND)) {
            (*pending)++;
            *lo &= (uint16_t)~(AR_TOP | AR_SECOND);
            *hi |= AR_SECOND;
        } else {
            return;
        }
        *lo = (uint16_t)(*lo << 1);
        *hi = (uint16_t)((*hi << 1) | 1u);
    }
}

/*
 * Encode one packet (n <= 8192 bytes) into out[]; returns its length incl.
 * the 4-byte header.  arCompress, :487-531: state init :492-494; symbols in
 * memory order (the ulonglong2 walk of :496-517 with LSB-first extraction in
 * writeLongLong :462-475 is memory order on a little-endian host); flush
 * writeRemaining :379-388 + writeClose :430-439; header :525-528.
 * out must have room for the worst case (callers give >= 2*n + 16).
 */
size_t oracle_encode_packet(const uint8_t *in, uint16_t n, uint8_t *out)
{
    ar_model m;
    bit_sink w = { out + AR_PACKET_HDR, 0, 0 };
    uint16_t lo = 0, hi = 0xFFFFu, pending = 0;

    model_reset(&m);
    for (unsigned i = 0; i < n; i++) {
        narrow(&m, in[i], &lo, &hi);
        encoder_renorm(&w, &lo, &hi, &pending);
    }
    {
        int b = (lo & AR_SECOND) != 0;
        sink_bit(&w, b);
        for (pending++; pending; pending--)
            sink_bit(&w, !b);
    }
    if (w.nheld) {
        *w.at++ = (uint8_t)(w.held << (8 - w.nheld));
    }
    size_t len = (size_t)(w.at - out);
    out[0] = (uint8_t)len;
    out[1] = (uint8_t)(len >> 8);
    out[2] = (uint8_t)n;
    out[3] = (uint8_t)(n >> 8);
    return len;
}

/* ---- MSB-first bit source; bits past `end` read as 0 ----------------------
 * readBit :553-569 / getChar :533-541 (the reference reads whatever bytes
 * follow the packet; a well-formed packet decodes identically whatever they
 * are, so the oracle feeds zeros and never leaves the buffer). */
typedef struct {
    const uint8_t *at, *end;
    uint8_t held, nheld;
} bit_source;

static int source_bit(bit_source *r)
{
    if (r->nheld == 0) {
        r->held = (r->at < r->end) ? *r->at : 0;
        r->at++;
        r->nheld = 8;
    }
    r->nheld--;
    return (r->held >> r->nheld) & 1;
}

/* getSymbolFromProbability, :727-763 (binary search on the Fenwick sums) */
static int model_find(const ar_model *m, uint16_t target)
{
    int first = 0, last = (int)AR_SYMBOLS, mid = last >> 1;
    while (last >= first) {
        if (target < model_below(m, (unsigned)mid)) {
            last = mid - 1;
            mid = first + ((last - first) >> 1);
            continue;
        }
        if (target >= model_below(m, (unsigned)mid + 1)) {
            first = mid + 1;
            mid = first + ((last - first) >> 1);
            continue;
        }
        return mid;
    }
    return -1;
}

/*
 * Decode one packet.  `avail` = bytes readable at pkt (>= its clen).  Writes
 * at most min(ulen, out_cap) bytes, returns the number written.
 * arDecompress :848-892; initializeDecoder :582-603; getUnscaledCode :703-716;
 * readEncodedBits :787-836.
 */
size_t oracle_decode_packet(const uint8_t *pkt, size_t avail, uint8_t *out, size_t out_cap)
{
    if (avail < AR_PACKET_HDR)
        return 0;
    size_t ulen = (size_t)pkt[2] | ((size_t)pkt[3] << 8);
    bit_source r = { pkt + AR_PACKET_HDR, pkt + avail, 0, 0 };
    ar_model m;
    uint16_t lo = 0, hi = 0xFFFFu, code = 0;
    size_t produced = 0;

    model_reset(&m);
    for (int i = 0; i < 16; i++)
        code = (uint16_t)((code << 1) | source_bit(&r));

    while (produced < ulen && produced < out_cap) {
        uint32_t span = (uint32_t)((int)hi - (int)lo) + 1u;      /* :708 */
        uint32_t t = (uint32_t)((int)code - (int)lo) + 1u;       /* :711 */
        t = t * m.total - 1u;
        t /= span;
        int s = model_find(&m, (uint16_t)t);
        if (s < 0)
            break;                                   /* :873-877 */
        out[produced++] = (uint8_t)s;
        narrow(&m, (unsigned)s, &lo, &hi);
        for (;;) {
            if (((hi ^ lo) & AR_TOP) == 0) {
                /* shift the agreed bit out */
            } else if ((lo & AR_SECOND) && !(hi & AR_SECOND)) {
                lo &= (uint16_t)~(AR_TOP | AR_SECOND);
                hi |= AR_SECOND;
                code ^= AR_SECOND;
            } else {
                break;
            }
            lo = (uint16_t)(lo << 1);
            hi = (uint16_t)((hi << 1) | 1u);
            code = (uint16_t)((code << 1) | source_bit(&r));
        }
    }
    return produced;
}

/* ---- batch layouts ---------------------------------------------------------
 * Fixed-stride slots as the device kernels use them:
 * garCompress :894-914 (input stride 8192, output stride 8704) and
 * garDecompress :916-934. */
size_t oracle_packet_count(size_t n_bytes)
{
    return (n_bytes + AR_PACKET_IN - 1) / AR_PACKET_IN;
}

/* returns the sum of packet lengths; slots must hold packet_count*8704 bytes.
 * A packet longer than the slot (never seen; SURVEY.md section 7 risk 3) is
 * reported by returning (size_t)-1. */
size_t oracle_encode_slots(const uint8_t *in, size_t n_bytes, uint8_t *slots)
{
    uint8_t tmp[2 * AR_PACKET_IN + 64];
    size_t total = 0, np = oracle_packet_count(n_bytes);
    for (size_t p = 0; p < np; p++) {
        size_t off = p * AR_PACKET_IN;
        size_t n = n_bytes - off < AR_PACKET_IN ? n_bytes - off : AR_PACKET_IN;
        size_t len = oracle_encode_packet(in + off, (uint16_t)n, tmp);
        if (len > AR_PACKET_SLOT)
            return (size_t)-1;
        memcpy(slots + p * AR_PACKET_SLOT, tmp, len);
        total += len;
    }
    return total;
}

void oracle_decode_slots(const uint8_t *slots, size_t n_packets, uint8_t *out)
{
    for (size_t p = 0; p < n_packets; p++)
        oracle_decode_packet(slots + p * AR_PACKET_SLOT, AR_PACKET_SLOT,
                             out + p * AR_PACKET_IN, AR_PACKET_IN);
}

/* Packets back to back, as they sit in a .gip file after the 20-byte header
 * (src/cpu_compressor.cpp:144-173).  `out` must hold packet_count*8704. */
size_t oracle_encode_stream(const uint8_t *in, size_t n_bytes, uint8_t *out)
{
    uint8_t tmp[2 * AR_PACKET_IN + 64];
    size_t total = 0, np = oracle_packet_count(n_bytes);
    for (size_t p = 0; p < np; p++) {
        size_t off = p * AR_PACKET_IN;
        size_t n = n_bytes - off < AR_PACKET_IN ? n_bytes - off : AR_PACKET_IN;
        size_t len = oracle_encode_packet(in + off, (uint16_t)n, tmp);
        memcpy(out + total, tmp, len);
        total += len;
    }
    return total;
}

/* Walks `off += clen` like src/cpu_compressor.cpp:47-78.  Returns bytes
 * written, or (size_t)-1 on a truncated / malformed stream. */
size_t oracle_decode_stream(const uint8_t *stream, size_t n_stream, uint8_t *out, size_t out_cap)
{
    size_t off = 0, produced = 0;
    while (off < n_stream) {
        if (n_stream - off < AR_PACKET_HDR)
            return (size_t)-1;
        size_t clen = (size_t)stream[off] | ((size_t)stream[off + 1] << 8);
        if (clen < AR_PACKET_HDR || clen > n_stream - off)
            return (size_t)-1;
        produced += oracle_decode_packet(stream + off, clen, out + produced, out_cap - produced);
        off += clen;
    }
    return produced;
}
