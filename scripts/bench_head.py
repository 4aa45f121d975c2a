"""time dt_head_bwd alone (fp32 B=32 and bf16 B=64 at 512x512, K=2): python scripts/bench_head.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deadtrees_amd import ops, _lib
import ctypes as C
dev = "cuda"
lib = _lib.load()
for name, B, bf in (("fp32 B=32", 32, False), ("bf16 B=64", 64, True)):
    H = W = 512
    x = torch.randn(B, H, W, 16, device=dev)
    if bf:
        x = x.to(torch.bfloat16)
    w = torch.randn(2 * 9 * 16, device=dev) * 0.1
    dl = torch.randn(B, 2, H, W, device=dev)
    dx = torch.empty_like(x)
    red = torch.empty(lib.dt_head_bwd_red_floats(B, H, W, 16, 2), device=dev)
    fn = lib.dt_head_bwd_bf16 if bf else lib.dt_head_bwd
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        fn(p(x), p(w), p(dl), p(dx), p(red), B, H, W, 16, 2, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(p(x), p(w), p(dl), p(dx), p(red), B, H, W, 16, 2, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    nbytes = B * H * W * (16 * (2 if bf else 4) * 2 + 8)
    print(f"{os.environ.get('DT_HIP_LIB', 'in-tree'):20s} {name}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s algorithmic")
