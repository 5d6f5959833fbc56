/*
 * gpuar_host.h -- host-callable packet codec of the drop-in boundary.
 *
 * The reference's kernel object also exports its per-packet codec as plain host
 * functions with C names (extern "C" block, /root/reference/src/gpuar.h:59-86);
 * its CPU compressor and self-test call them directly
 * (src/cpu_compressor.cpp:59-60,159-160; src/main.cpp:27-43).  libgpuar_hip.so
 * and the GPU-runtime-free libgpuar_host.so export the same three names with
 * the same signatures and the same meaning, so those callers link unchanged.
 *
 * The model is the caller's: a 257-entry Fenwick array over symbols 0..255
 * (entry i, 1 <= i <= 256, holds the count of symbols (i - lowbit(i), i];
 * entry 0 stays 0) plus the running total.  Both calls start from the state
 * they are given and leave the adapted state behind, as the reference does;
 * every caller in the reference re-initialises before each packet.
 *
 * No HIP, no allocation, re-entrant; one packet per call.
 */
#ifndef GPUAR_HOST_H
#define GPUAR_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifndef GPUAR_MODEL_ENTRIES
#define GPUAR_MODEL_ENTRIES 257u          /* UPPER(EOF_CHAR) + 1, src/gpuar.h:20,30,46 */
#endif

/* Layout-compatible with the reference's `struct AdaptiveProbabilityRange`
 * (src/gpuar.h:42-48) and `probability_t` (:32).  When the reference's own
 * header is in scope first (it defines MASK_BIT), its types are used as is. */
#ifndef MASK_BIT
typedef unsigned short probability_t;
struct AdaptiveProbabilityRange {
    probability_t ranges[GPUAR_MODEL_ENTRIES];
};
typedef struct AdaptiveProbabilityRange AdaptiveProbabilityRange;
#endif

#ifdef __cplusplus
#define GPUAR_REF(T) T &
extern "C" {
#else
#define GPUAR_REF(T) T *                /* a C++ reference parameter is a pointer at the ABI */
#endif

/* Replaces initializeAdaptiveProbabilityRangeList (src/gpuar_kernel.cu:403-419):
 * every symbol count 1, total 256. */
void initializeAdaptiveProbabilityRangeList(AdaptiveProbabilityRange *r, GPUAR_REF(probability_t) cumProb);

/* Replaces arCompress (src/gpuar_kernel.cu:487-531): codes `size` (<= 8192)
 * bytes at fpIn into one packet at outFile -- u16 LE clen, u16 LE ulen,
 * bitstream -- and returns clen.  Unlike the reference it reads exactly `size`
 * input bytes (the reference over-reads up to 15, :496-517).  outFile needs
 * room for the packet: 8704 bytes always suffice for a freshly initialised
 * model. */
uint16_t arCompress(const uint8_t *fpIn, const uint16_t size, uint8_t *outFile,
                    GPUAR_REF(AdaptiveProbabilityRange) r, GPUAR_REF(probability_t) cumulativeProb);

/* Replaces arDecompress (src/gpuar_kernel.cu:848-892): decodes the packet at
 * fpIn into fpOut and returns the number of bytes produced (the packet's ulen,
 * fewer if the code value leaves the model, :873-877).  The reference ignores
 * `inSize` and reads bits as far as the decoder asks; this implementation reads
 * max(inSize, the packet's own clen) bytes and zeros beyond, which decodes every
 * well-formed packet identically. */
uint16_t arDecompress(const uint8_t *fpIn, const uint16_t inSize, uint8_t *fpOut,
                      GPUAR_REF(AdaptiveProbabilityRange) r, GPUAR_REF(probability_t) cumProb);

#ifdef __cplusplus
}
#endif
#undef GPUAR_REF
#endif /* GPUAR_HOST_H */
