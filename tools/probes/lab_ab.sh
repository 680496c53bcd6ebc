mkdir -p gpurun_out/lab3
(
for rep in 1 2; do
for v in k128_cvt_256x320 k128_h_256x320; do
timeout -k 10 120 tools/probes/gemm_probe 1500 12419 0.04 $v 200 || exit 1
timeout -k 10 120 tools/probes/gemm_probe_b128 1500 12419 0.04 $v 200 || exit 1
done
for v in k512_b_256x256 k128_h_256x256; do
timeout -k 10 120 tools/probes/gemm_probe 4096 4096 0.1 $v 200 || exit 1
timeout -k 10 120 tools/probes/gemm_probe_b128 4096 4096 0.1 $v 200 || exit 1
done
done
) > gpurun_out/lab3/ab.txt 2>&1
cut -c1-150 gpurun_out/lab3/ab.txt
