export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4n
python -m pytest tests/test_gpu_cluster.py -x -q -m gpu 2>&1 | tail -3 &&
timeout -k 10 700 bash tools/r04_cluster.sh > /dev/null; cut -c1-260 gpurun_out/cluster/r04_cluster_device_lab.txt
