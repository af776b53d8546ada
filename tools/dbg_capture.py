import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
st = torch.cuda.Stream()
status = ctypes.c_int(-1)
x = torch.zeros(16, device="cuda")
print("before:", hip.hipStreamIsCapturing(ctypes.c_void_p(st.cuda_stream), ctypes.byref(status)), status.value)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(st):
    with torch.cuda.graph(g, stream=st):
        cur = torch.cuda.current_stream().cuda_stream
        print("in capture: cur == st:", cur == st.cuda_stream)
        rc = hip.hipStreamIsCapturing(ctypes.c_void_p(cur), ctypes.byref(status))
        print("in capture:", rc, status.value)
        x += 1
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B
py = B.make_params(3, 16, 27, combine=B.COMBINE_VHDL, sin_type=B.SIN_TAYLOR, lut_size=7)
out = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
g2 = torch.cuda.CUDAGraph()
try:
    with torch.cuda.stream(st):
        with torch.cuda.graph(g2, stream=st):
            bhw.generate(py, 0, 1 << 16, out=out)
    print("no raise")
except Exception as e:
    print("raised", repr(e))
