"""The unmodified reference ViT-L (PyTorch-ROCm eager, bf16 + fp16 autocast, B = 8, train forward + backward) alone on the GPU, for a
rocprofv3 --kernel-trace --stats summary of where ITS step goes (profiles/r04_reference_gpu_eager_kernel_stats.csv):
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -o ref -- python <repo>/tools/reference_gpu_prof.py
Baseline evidence only; bench.py's `reference_gpu` leg is the number quoted."""
import json
import sys

import torch

sys.path.insert(0, ".")
import bench   # noqa: E402
from painter_amd import models_painter   # noqa: E402

dev = torch.device("cuda", 0)
model = models_painter.painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1(compute_dtype="bf16")
bench.randomize_parameters(model, seed=1)
model = model.to(dev).train()
c = model._cfg
inputs = bench.synthetic_inputs(8, c.H, c.W, c.L, 1234, dev)
print(json.dumps(bench.reference_gpu_baseline(model, inputs, dev, steps=3, warmup=1)))
