"""space_filling_forest_star_amd — MI355X-native SFF / SFF* tree-expansion hot path.

Thin ctypes binding of libsffgpu.so (C ABI in include/sffgpu.h).  There is no CPU fallback:
importing works anywhere (so the symbol table can be checked), but creating a Context
without a gfx950 GPU raises.
"""
from ._lib import (Rrt, RrtCfg, RrtStats, exchange_records, run_distributed, host_staged_allgather, Context, Forest, ForestCfg, ForestStats, SffGpuError, lib, lib_path, build_library,
                   EXPORTED_SYMBOLS)

__all__ = ["Rrt", "RrtCfg", "RrtStats", "exchange_records", "run_distributed", "host_staged_allgather", "Context", "Forest", "ForestCfg", "ForestStats", "SffGpuError", "lib", "lib_path", "build_library",
           "EXPORTED_SYMBOLS"]
