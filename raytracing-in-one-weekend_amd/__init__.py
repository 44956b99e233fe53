"""MI355X-native sample-batch path for renaudbedard/raytracing-in-one-weekend.

Package layout:
  abi.py     ctypes mirror of include/rtow.h
  lib.py     loader for csrc/librtow_hip.so (the C-ABI product library; fails loudly if missing)
  host.py    host-side mirror of the reference's job structs (SampleBatchJob, CombineJob, ...)
  scenes.py  synthetic benchmark scenes (input preparation, shared by product and tests)
  csrc/      hand-written HIP (gfx950) kernels + the C ABI implementation

The directory name contains '-' so it is loaded with importlib (see tests/conftest.py, bench.py):
    rtow = importlib.import_module("raytracing-in-one-weekend_amd")
"""
from . import abi, host, lib, scenes  # noqa: F401
from .host import (CombineJob, Context, DeviceBuffer, FinalizeTexturesJob, ReduceMetricsJob, SampleBatchJob,  # noqa: F401
                   sample_batch_chain_device, sample_batch_chain_host, sample_batch_group_device, sample_batch_host)

__all__ = ["abi", "host", "lib", "scenes", "Context", "DeviceBuffer", "SampleBatchJob", "CombineJob", "FinalizeTexturesJob",
           "ReduceMetricsJob", "sample_batch_host", "sample_batch_chain_device", "sample_batch_chain_host", "sample_batch_group_device"]
