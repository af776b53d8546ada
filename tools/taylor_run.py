import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B
o = torch.empty(1 << 26, dtype=torch.int32, device="cuda")
for w in (16, 32):
    pt = bhw.make_params(1, 26, w, sin_type=B.SIN_TAYLOR, combine=B.COMBINE_VHDL, lut_size=9)
    for _ in range(20): bhw.generate(pt, 0, 1 << 26, out=o)
torch.cuda.synchronize()
