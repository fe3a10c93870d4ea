"""sweep the fused-dgrad kernel test over batch sizes / shapes (GPU box)"""
import sys, traceback
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import test_kernels_gpu as T
dev = torch.device("cuda:0")
bad = 0
for N in (1, 2, 3, 4, 5, 6, 8):
    for (H, C, Co, k, s, p) in ((10, 256, 256, 3, 1, 1), (10, 128, 256, 3, 2, 1), (20, 128, 128, 3, 1, 1), (5, 512, 512, 3, 1, 1),
                                (10, 128, 256, 1, 2, 0), (40, 64, 64, 3, 1, 1)):
        for mode, second in ((1, True), (1, False), (2, False)):
            case = (N, H, H, C, Co, k, s, p)
            try:
                T.test_conv2d_dgrad_fused_bn_backward(dev, case, mode, second)
            except AssertionError as e:
                bad += 1
                print("FAIL", case, mode, second, str(e).splitlines()[0][:150])
        try:
            T.test_conv2d(dev, (N, H, H, C, Co, k, s, p), False)
            T.test_conv2d(dev, (N, H, H, C, Co, k, s, p), True)
        except AssertionError as e:
            bad += 1
            print("FAIL conv2d", (N, H, H, C, Co, k, s, p), str(e).splitlines()[0][:150])
print("bad", bad)
