"""what the vendor GEMM (hipBLASLt via torch) reaches on the plain-linear shapes of the UNet (development tool)"""
import torch, sys
shapes = [(65536, 320, 320), (65536, 320, 1280), (16384, 640, 640), (16384, 640, 2560), (4096, 1280, 1280), (4096, 1280, 5120),
          (65536, 2560, 320), (16384, 5120, 640), (4096, 10240, 1280), (1024, 1280, 1280), (256, 1280, 1280), (4096, 320, 320)]
for M, N, K in shapes:
    a = torch.randn(M, K, dtype=torch.float16, device="cuda")
    w = torch.randn(N, K, dtype=torch.float16, device="cuda") * K ** -0.5
    b = torch.randn(N, dtype=torch.float16, device="cuda")
    r = torch.randn(M, N, dtype=torch.float16, device="cuda")
    for _ in range(3):
        y = torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = torch.nn.functional.linear(a, w, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    e0.record()
    for _ in range(20):
        y2 = torch.addmm(r, a, w.t())
    e1.record(); torch.cuda.synchronize()
    us2 = e0.elapsed_time(e1) / 20 * 1e3
    print(f"M{M} N{N} K{K}: linear+bias {us:7.1f} us {2.0*M*N*K/us/1e6:6.0f} TF | addmm(residual) {us2:7.1f} us {2.0*M*N*K/us2/1e6:6.0f} TF", flush=True)
