#!/bin/bash
# builds the lab probes next to their sources (binaries are git-ignored; they travel to the GPU box with gpurun)
cd "$(dirname "$0")/../.." || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ibsmr-sddmm_amd/csrc"
/opt/rocm/bin/hipcc $F -o tools/probes/sweep_probe tools/probes/sweep_probe.hip &
/opt/rocm/bin/hipcc $F -DBSMR_SWEEP_STAMPS -o tools/probes/sweep_probe_stamps tools/probes/sweep_probe.hip &
/opt/rocm/bin/hipcc $F -o tools/probes/stream_probe tools/probes/stream_probe.hip &
/opt/rocm/bin/hipcc $F -o tools/probes/gemm_probe tools/probes/gemm_probe.hip &
/opt/rocm/bin/hipcc $F -DBSMR_GEMM_LAB -o tools/probes/gemm_probe_lab tools/probes/gemm_probe.hip &
wait
