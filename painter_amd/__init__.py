"""painter_amd -- MI355X-native (gfx950) Painter / SegGPT ViT forward/backward hot path.

Host side: Python mirroring the reference's nn.Module interface (models_painter.py / models_seggpt.py);
compute: hand-written HIP kernels behind the C ABI in include/painter_hip.h (painter_amd/lib/libpainter_hip.so)."""
__version__ = "0.1.0"

import os as _os

# The MI355X boxes' host driver only supports dmabuf IPC; HSA reads this when the runtime initialises, i.e. at the first HIP call
# of the process, so it is set at import (multi-process RCCL / CUDA-tensor sharing fails without it).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
