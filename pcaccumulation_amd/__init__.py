"""MI355X-native implementation of the PCAccumulation per-frame forward hot path."""
import os as _os

# This pool's host driver (and any host with dmabuf-only IPC) needs the non-legacy IPC mode for RCCL and for sharing device memory between processes:
# without it the first collective at N > 1 fails with `hipIpcGetMemHandle: invalid argument`.  The HIP runtime reads the variable when it initialises, so it is
# set HERE -- at package import, before torch / HIP come up in any entry point (bench.py, the drop-in trainers, tests/dist_worker.py, tools) -- not inside
# distributed.init_from_env, which may run after the runtime is up.  An explicit setting in the environment wins.  
_os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

# The dense conv stack goes through MIOpen, whose kernels for gfx950 are JIT-compiled on first use (the PyTorch
# wheel ships no gfx950 kernel database).  Keep that cache inside the repo tree so that it travels with the
# snapshot to a fresh GPU box instead of being rebuilt (minutes) on every run.
_cache = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), '.miopen_cache')
_os.environ.setdefault('MIOPEN_USER_DB_PATH', _cache)
_os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', _cache)
try:
    _os.makedirs(_cache, exist_ok=True)
except OSError:
    pass
