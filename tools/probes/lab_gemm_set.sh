#!/bin/bash
# the GEMM probe on the shapes the profiles quote, product build and stamped lab build: tools/probes/lab_gemm_set.sh <tag>
mkdir -p gpurun_out/$1
P=tools/probes/gemm_probe
L=tools/probes/gemm_probe_lab
(
timeout -k 10 120 $P 1500 12419 0.04 k128_cvt_256x320 200 &&
timeout -k 10 120 $P 1500 12419 0.04 k128_h_256x320 200 &&
timeout -k 10 120 $P 1500 12419 0.04 k128_cvt_256x256 200 &&
timeout -k 10 120 $P 4096 4096 0.1 k512_b_256x256 200 &&
timeout -k 10 120 $P 4096 4096 0.1 k128_h_256x256 200 &&
timeout -k 10 120 $P 4096 4096 0.1 k64_h_256x256 200 &&
timeout -k 10 120 $P 1500 12419 0.04 k512_b_256x320 200 &&
timeout -k 10 120 $L 1500 12419 0.04 k128_cvt_256x320 200 0 &&
timeout -k 10 120 $L 1500 12419 0.04 k128_h_256x320 200 0
) > gpurun_out/$1/gemm_set.txt 2>&1
cat gpurun_out/$1/gemm_set.txt
