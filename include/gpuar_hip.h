/*
 * gpuar_hip.h -- C ABI of the MI355X-native (gfx950) arithmetic-coding path.
 *
 * This is the drop-in boundary for the GPU encode/decode path of
 * jiahansu/GPUAR.  The first three entry points carry the reference's own
 * names and signatures, so the reference's GPUCompressor
 * (src/gpu_compressor.cpp:19,185,357) links against libgpuar_hip.so unchanged;
 * the gpuar_hip_* entry points are the stream- and device-explicit native ABI
 * the rest of this repository (C++ host classes, Python bindings, bench) uses.
 *
 * Plain pointers and sizes only.  All `d_*` / `source` / `destination`
 * pointers are DEVICE pointers owned by the caller (hipMalloc, or any
 * allocator that yields HIP device memory, e.g. a torch CUDA tensor).
 * Nothing here allocates device memory, and nothing throws across the boundary.
 * State kept between calls, all of it per device and none of it affecting
 * results: a fallback status word (used only by launches that were given no
 * status word of their own, i.e. the reference-named executors) and an arrival
 * counter per compute unit the encoder uses to place its wavefronts.
 *
 * Packet geometry (reference: src/gpu.h:8-14):
 *   input  is cut into 8192-byte packets, packet p = bytes [p*8192, ...)
 *   output packet p lives in the 8704-byte slot at p*8704; only its first
 *          `clen` bytes are defined: u16 LE clen (incl. this 4-byte header),
 *          u16 LE ulen, then the MSB-first arithmetic-coded bitstream
 *          (src/gpuar_kernel.cu:523-528).
 */
#ifndef GPUAR_HIP_H
#define GPUAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPUAR_PACKET_BYTES        8192u  /* UNCOMPRESSED_PACKET_SIZE, src/gpu.h:13 */
#ifndef GPUAR_SLOT_BYTES                 /* only tests/lane_emulation.cpp ever overrides it (to drive a lane into overflow) */
#define GPUAR_SLOT_BYTES          8704u  /* COMPRESSED_PACKET_SIZE,   src/gpu.h:12 */
#endif
#define GPUAR_PACKET_HEADER_BYTES 4u     /* PACKET_HEADER_LENGTH,     src/gpu.h:14 */

/* Error codes returned by the gpuar_hip_* calls (0 = ok).  Positive values
 * are hipError_t codes passed through unchanged. */
#define GPUAR_OK                 0
#define GPUAR_ERR_ALIGNMENT     (-1)  /* device pointer not 16-byte aligned            */
#define GPUAR_ERR_ARGUMENT      (-2)  /* null pointer / size out of range              */
#define GPUAR_ERR_NO_DEVICE     (-3)  /* no HIP device visible                         */

/* Bits a launch ORs into its status word (the caller's `d_status`, or the
 * device's fallback word that gpuar_hip_status reads). */
#define GPUAR_STATUS_SLOT_OVERFLOW  0x1u /* a packet outgrew its 8704-byte slot; its slot is truncated
                                            (the reference would write past the slot, SURVEY.md s.7 risk 3) */
#define GPUAR_STATUS_BAD_PACKET     0x2u /* decode met a malformed packet (ulen > 8192, clen < 4, or a code
                                            value outside the model, src/gpuar_kernel.cu:873-877)          */

/* ------------------------------------------------------------------------
 * Reference-named entry points (the reference's kernel object exports these;
 * declarations: /root/reference/src/gpuar.h:74,77,78).
 * ---------------------------------------------------------------------- */

/* Replaces initConstantRange (src/gpuar_kernel.cu:453-460), which uploads the
 * initial uniform model to __constant__ memory.  The HIP kernels build the
 * initial model in LDS themselves and read their reciprocal table from a
 * compile-time constant, so this only selects/initialises the current device
 * context; calling it is optional and idempotent. */
void initConstantRange(void);

/* Replaces garCompressExecutor (src/gpuar_kernel.cu:936-944): encodes the
 * `size` bytes at `source` into ceil(size/8192) slots at `destination`.
 * Asynchronous on the NULL stream, like the reference's <<<>>> launch; errors
 * surface at the caller's next synchronising HIP call and through
 * gpuar_hip_last_error().  `numBlocks` (the reference's grid size for
 * 32-thread blocks) is accepted and ignored: the launch shape is derived
 * from `size`. */
void garCompressExecutor(const uint8_t *source, size_t size, uint8_t *destination, uint32_t numBlocks);

/* Replaces garDecompressExecutor (src/gpuar_kernel.cu:946-954): `size` is
 * numPackets*8704 (src/gpu_compressor.cpp:357); every slot that STARTS inside
 * `size` is decoded (index*8704 < size, src/gpuar_kernel.cu:916-934) and nothing
 * at or beyond source + size is read; slot p decodes to destination + p*8192. */
void garDecompressExecutor(const uint8_t *source, size_t size, uint8_t *destination, uint32_t numBlocks);

/* ------------------------------------------------------------------------
 * Native ABI: explicit stream (a hipStream_t passed as void*; NULL = the
 * NULL stream), int return codes, usable from one host thread per GPU.
 * ---------------------------------------------------------------------- */

/* Number of packets / slots for an input of n_bytes. */
size_t gpuar_hip_packet_count(size_t n_bytes);

/* `d_status` (encode, decode, decode_stream): a caller-owned 32-bit DEVICE word
 * (4-byte aligned) that this launch -- and only this launch -- ORs its
 * GPUAR_STATUS_* bits into; the caller zeroes it and copies it back on the same
 * stream, so concurrent launches on other streams never share a flag and no
 * device-wide synchronisation is needed.  NULL: the bits go to the device's
 * fallback word instead (gpuar_hip_status). */

/* Encode n_bytes at d_in (16-byte aligned) into packet slots at d_slots
 * (16-byte aligned, gpuar_hip_packet_count(n_bytes)*8704 bytes). */
int gpuar_hip_encode(const uint8_t *d_in, size_t n_bytes, uint8_t *d_slots, uint32_t *d_status, void *stream);

/* The same with the ENCODE kernel named by the caller (there is one decode kernel per input layout and no
 * decode mode).  Both encode kernels write the same bytes; they differ in how a launch is cut into wavefronts:
 *   GPUAR_MODE_THROUGHPUT  three working wavefronts (and one that carries constants) per 64 packets:
 *                          the most bytes per second from a launch that fills the chip;
 *   GPUAR_MODE_LATENCY     a finer cut -- six working wavefronts (and a seventh that carries constants) per
 *                          64 packets -- with a shorter symbol step: faster while the launch cannot fill
 *                          the chip by itself;
 *   GPUAR_MODE_AUTO        what gpuar_hip_encode does: LATENCY up to 32768 packets (256 MiB of input),
 *                          THROUGHPUT above -- right for a launch that has the chip to itself; a pipeline
 *                          that keeps several launches in flight names THROUGHPUT.
 * The choice is an argument, never an environment variable: this library reads no environment.
 * Any other `mode` is GPUAR_ERR_ARGUMENT. */
#define GPUAR_MODE_AUTO        0
#define GPUAR_MODE_THROUGHPUT  1
#define GPUAR_MODE_LATENCY     2
int gpuar_hip_encode_mode(const uint8_t *d_in, size_t n_bytes, uint8_t *d_slots, uint32_t *d_status, void *stream, int mode);

/* Decode n_packets slots at d_slots into d_out (n_packets*8192 bytes; the
 * last packet writes only its ulen bytes). */
int gpuar_hip_decode(const uint8_t *d_slots, size_t n_packets, uint8_t *d_out, uint32_t *d_status, void *stream);

/* Device-side compaction (what the reference does with one 8704-byte D2H copy
 * and one fwrite per packet, src/gpu_compressor.cpp:138,161-168):
 *   d_offsets[p]   = sum of clen of packets < p   (n_packets+1 entries, u64)
 *   d_stream       = packets back to back, exactly the bytes that follow the
 *                    20-byte header in a .gip file.
 * d_stream (8-byte aligned) needs room for the sum of clen (<= n_packets*8704);
 * its first bytes serve as scan scratch before the packets are gathered into
 * it, so the call keeps no state outside its arguments and may run
 * concurrently on different streams and devices.  At most 16 777 215 packets
 * (128 GiB of input) per call: GPUAR_ERR_ARGUMENT above. */
int gpuar_hip_compact(const uint8_t *d_slots, size_t n_packets, uint8_t *d_stream,
                      uint64_t *d_offsets, void *stream);

/* Decode straight from a back-to-back packet stream: d_offsets[p] is the byte
 * offset of packet p in d_stream (n_packets+1 entries; the host builds it by
 * walking `off += clen`, src/gpu_compressor.cpp:299-312, or keeps the array
 * gpuar_hip_compact produced). */
int gpuar_hip_decode_stream(const uint8_t *d_stream, const uint64_t *d_offsets, size_t n_packets,
                            uint8_t *d_out, uint32_t *d_status, void *stream);

/* Reads and clears the FALLBACK status word of the current device: what
 * launches without a `d_status` of their own reported (the reference-named
 * executors above).  Synchronises the whole device -- meant for that
 * single-stream legacy use, not for pipelines (pass `d_status` there). */
int gpuar_hip_status(uint32_t *flags);

/* Last error recorded by a void-returning reference-named entry point on this
 * host thread (0 = none); cleared by the call. */
int gpuar_hip_last_error(void);

const char *gpuar_hip_error_string(int code);

/* Build identification, e.g. "gpuar-hip 0.2 gfx950".  A library built with timing switches (GPUAR_EXP_*: pieces of the kernels left
 * out to price them, WRONG output by design) says "... EXPERIMENT BUILD ..." here. */
const char *gpuar_hip_version(void);

/* The number of this header's ABI, GPUAR_HIP_ABI_VERSION at the time the library was built.  It changes whenever
 * the signature of an existing entry point does (2: encode / decode / decode_stream take `d_status` in front of
 * `stream`), so a caller built against another header can refuse to go on instead of passing a stream handle
 * where a status word is expected:  if (gpuar_hip_abi_version() != GPUAR_HIP_ABI_VERSION) ...  */
#define GPUAR_HIP_ABI_VERSION 2
int gpuar_hip_abi_version(void);

/* Synthetic streams of SURVEY.md section 8(d), generated on the device so
 * multi-GiB benchmark inputs never cross PCIe.  kind: 0 uniform, 1 zipf,
 * 2 text.  Fills d_out[0..n) with bytes [offset, offset+n) of the stream. */
int gpuar_hip_generate(int kind, uint64_t seed, uint64_t offset, size_t n, uint8_t *d_out, void *stream);

/* Measurement support: a plain device-to-device copy of n_bytes (a multiple of 16; both pointers 16-byte aligned),
 * 16 bytes per lane -- the practical HBM roof bench.py measures on the box and quotes next to the datasheet's
 * 8 TB/s.  Moves 2 * n_bytes through HBM.  n_bytes < 64 GiB: GPUAR_ERR_ARGUMENT above. */
int gpuar_hip_copy(const uint8_t *d_src, uint8_t *d_dst, size_t n_bytes, void *stream);

/* Measurement support: the shader clock while the throughput kernels ran.  Every 64th workgroup of encode_kernel
 * (which = 0) and of the two decode kernels (which = 1) notes, in a slot of its own on the current device, the
 * shader clock's counter and the constant 100 MHz clock's counter at its start and at its end.
 * Copies the GPUAR_CLOCK_SLOTS records {shader at start, 100 MHz at start, shader at end, 100 MHz at end} into
 * ticks[4 * GPUAR_CLOCK_SLOTS] (slots no sampled workgroup has written since the last reset are 0) and, if `reset`,
 * zeroes them.  Synchronises the device.  sum(shader end - start) / sum(100 MHz end - start) x 100 MHz = the clock
 * the vector pipes ran at -- under this load not the data sheet's 2.4 GHz, which is why bench.py quotes its
 * vector-issue roof against both. */
#define GPUAR_CLOCK_SLOTS 256
int gpuar_hip_clock_samples(int which, uint64_t *ticks, int reset);

#ifdef __cplusplus
}
#endif
#endif /* GPUAR_HIP_H */
