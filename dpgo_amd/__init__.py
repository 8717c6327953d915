"""dpgo_amd -- MI355X-native local solver for dpgo's per-agent RBCD step.

Host-side mirror of the reference interface (PoseGraph, QuadraticProblem, QuadraticOptimizer,
LiftedSEManifold, ROptParameters, ROPTResult) over the C ABI of libdpgo_hip.so
(include/dpgo_hip.h).  Importing this package loads the shared library and fails loudly if it
has not been built; there is no CPU fallback.
"""
from . import lib as _lib

_lib.load()  # ImportError if libdpgo_hip.so is missing

from .lib import DpgoError, device_count  # noqa: E402
from .measurements import RelativeSEMeasurements, partition_contiguous, read_g2o_file  # noqa: E402
from .solver import (LiftedSEManifold, PoseGraph, QuadraticOptimizer, QuadraticProblem,  # noqa: E402
                     ROptParameters, ROPTResult)

from .trajectory import (load_trajectory, log_measurements, log_trajectory, round_trajectory,  # noqa: E402
                         round_trajectory_device)

__all__ = ["load_trajectory", "log_measurements", "log_trajectory", "round_trajectory", "round_trajectory_device",
           "DpgoError", "device_count", "RelativeSEMeasurements", "partition_contiguous", "read_g2o_file",
           "LiftedSEManifold", "PoseGraph", "QuadraticOptimizer", "QuadraticProblem", "ROptParameters",
           "ROPTResult"]
