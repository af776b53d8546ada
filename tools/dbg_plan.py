import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B
for model in (0, 1):
    p = bhw.make_params(7, 26, 32, model=model)
    print("before:", B.describe_plan(p, 0, 1 << 26))
    out = bhw.generate(p, 0, 1 << 26)
    torch.cuda.synchronize()
    print("after: ", B.describe_plan(p, 0, 1 << 26))
