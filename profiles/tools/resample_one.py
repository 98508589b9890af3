"""One rate pair of the resampler, a few launches (for rocprofv3 passes): python profiles/tools/resample_one.py [orig target]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sound-event-detection_amd"))
import sed_amd  # noqa: E402

orig, target = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (44100, 16000)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.randn(512, orig * 10, device=dev) * 0.1
rs = sed_amd.Resampler(orig, target)
for _ in range(20):
    y = rs(x)
torch.cuda.synchronize()
print(float(y.double().abs().mean()))
