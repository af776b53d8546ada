import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, blackman_harris_win_amd as bhw
p4 = bhw.make_params(4, 16, 24)
o4 = torch.empty((1024, 1 << 16), dtype=torch.int32, device="cuda")
for _ in range(30): bhw.generate_batched(p4, 1024, out=o4)
torch.cuda.synchronize()
