#!/usr/bin/env bash
# extracts decode_slots_kernel from build/gpuar_kernels.s into /tmp/dec.s
cd /root/repo/build || exit 1
s=$(grep -n "^_ZN5gpuar19decode_slots_kernelEPKhjPh:" gpuar_kernels.s | cut -d: -f1)
e=$(grep -n "\.Lfunc_end1:" gpuar_kernels.s | cut -d: -f1)
awk -v s=$s -v e=$e 'NR>=s && NR<=e' gpuar_kernels.s > /tmp/dec.s
wc -l /tmp/dec.s
