#!/bin/bash
# A/B of two builds of the GEMM probe in one session: tools/probes/lab_ab.sh <other binary> <tag> [variants on the nips-like shape]
# (build the other binary with the macro under test: hipcc ... -D<MACRO> -o tools/probes/gemm_probe_<x> tools/probes/gemm_probe.hip)
mkdir -p gpurun_out/$2
NIPS=${3:-"k128_cvt_256x320 k128_h_256x320 k64_cvt_256x320"}
(
for rep in 1 2 3; do
for v in $NIPS; do
timeout -k 10 120 tools/probes/gemm_probe 1500 12419 0.04 $v 200 || exit 1
timeout -k 10 120 $1 1500 12419 0.04 $v 200 || exit 1
done
for v in k512_b_256x256 k128_h_256x256; do
timeout -k 10 120 tools/probes/gemm_probe 4096 4096 0.1 $v 200 || exit 1
timeout -k 10 120 $1 4096 4096 0.1 $v 200 || exit 1
done
done
) > gpurun_out/$2/ab.txt 2>&1
awk '{print $1, $11, $12, $13, $14, $(NF-7), $(NF-6), $(NF-5), $(NF-4)}' gpurun_out/$2/ab.txt | paste - -
