"""MI355X-native anchor-chaining DP behind the offload boundary of stormalex/minimap2_chaindp.

The product is the C-ABI library csrc/libchaindp_hip.so (include/chaindp.h, include/chaindp_fpga.h);
this package is its host-side mirror for tests and benchmarks:
  chaindp    ctypes front end of the batch API (HIP kernels; no CPU fallback)
  fpga       the reference's packet-level driver ABI (fpga.h) as served by the library
  params     DP parameter presets as the reference derives them
  anchorgen  seeded ONT-shaped synthetic anchor batches
  shard      read sharding across the GPUs of a node
  dump       anchor-dump files produced from real reads by the reference's own front half (oracle/mt_dump.c)
"""
__all__ = ["chaindp", "fpga", "params", "anchorgen", "shard", "dump"]
