"""MI355X-native implementation of the PCAccumulation per-frame forward hot path."""
import os as _os

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
