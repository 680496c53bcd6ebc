#!/bin/bash
# device clustering under the measured rule and under each rule pinned -> gpurun_out/cluster/r04_cluster_device_lab.txt
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/cluster
(
echo "## BSMR_CLUSTER_RULE=0 (default): clusters ahead of the older ones' decisions when the kernel's evidence allows; row order checked against the host"
BSMR_CLUSTER_RULE=0 python tools/cluster_device_lab.py --check || exit 1
echo "## BSMR_CLUSTER_RULE=2: rule by measurement (rows per millisecond of trial batches)"
BSMR_CLUSTER_RULE=2 python tools/cluster_device_lab.py --check || exit 1
echo "## BSMR_CLUSTER_RULE=1: only behind them (round 2's rule)"
BSMR_CLUSTER_RULE=1 python tools/cluster_device_lab.py
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/cluster/r04_cluster_device_lab.txt
