import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deadtrees_amd import ops, _lib
lib = _lib.load()
B, H, W, Ci, Co = 1, 16, 16, 64, 64
x = torch.zeros((B, H, W, Ci)); dy = torch.zeros((B, H, W, Co))
x[0, 5, 6, 3] = 1.0
dy[0, 5, 6, 7] = 1.0
xg, dyg = x.cuda(), dy.cuda()
d = ops.conv_desc(B, H, W, Ci, 0, 0, Co, 3, 1, 1)
nbytes = lib.dt_conv2d_wgrad_winograd_workspace(C.byref(d))
ws = torch.full((nbytes // 4,), float("nan"), device="cuda")
dw = torch.zeros((3, 3, Ci, Co), device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
rc = lib.dt_conv2d_wgrad_winograd(C.byref(d), p(xg), None, p(dyg), p(dw), p(ws), nbytes, None, None, None)
torch.cuda.synchronize()
print("rc", rc, "ws floats", ws.numel(), "nan count", int(torch.isnan(ws).sum()))
w = ws[:4 * 16 * 4096].view(4, 16, 64, 64).cpu()
for k in range(4):
    nz = (w[k].abs() > 1e-9).nonzero()
    print("split", k, "nonzeros", len(nz), [tuple(i.tolist()) for i in nz[:8]], "nan", int(torch.isnan(w[k]).sum()))
