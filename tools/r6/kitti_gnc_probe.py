"""BASELINE configs[4] alone (for rocprofv3 --kernel-trace --stats): kitti_00 + 25 outliers, 4 agents, GNC-TLS with the
reference schedule; prints bench.kitti_gnc_gpu's record.  usage: python tools/r6/kitti_gnc_probe.py"""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
print(json.dumps(bench.kitti_gnc_gpu(5, 0)))
