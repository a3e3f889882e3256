"""painter_amd -- MI355X-native (gfx950) Painter / SegGPT ViT forward/backward hot path.

Host side: Python mirroring the reference's nn.Module interface (models_painter.py / models_seggpt.py);
compute: hand-written HIP kernels behind the C ABI in include/painter_hip.h (painter_amd/lib/libpainter_hip.so)."""
__version__ = "0.1.0"
