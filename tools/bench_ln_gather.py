import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from stable_renderer_amd import ops as O, _lib as L
from stable_renderer_amd.plan import PlanBuilder
dev = torch.device("cuda")
for (B, HW, Cc) in [(2, 4096, 320), (2, 1024, 640), (2, 256, 1280), (2, 64, 1280), (16, 4096, 320)]:
    res = {}
    for mode in ("ln_only", "gather+ln", "fused"):
        pb = PlanBuilder(dev, torch.float16)
        x = pb.buf(B, HW, Cc); x.normal_()
        g, b = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
        sel = torch.tensor([1], dtype=torch.int32, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        src, y = pb.buf(1, HW, Cc), pb.buf(1, HW, Cc)
        for i in range(50):
            if mode == "ln_only":
                pb.layernorm(x[1:2], g, b, y, HW, Cc)
            elif mode == "gather+ln":
                pb.gather_rows(x, sel, src, 1, HW * Cc * 2, B, err)
                pb.layernorm(src, g, b, y, HW, Cc)
            else:
                pb.layernorm_gather(x, sel, 1, HW, B, g, b, y, Cc, err_flag=err)
        plan = pb.take()
        st = torch.cuda.Stream(); plan.capture(st); torch.cuda.synchronize()
        with torch.cuda.stream(st):
            for _ in range(3): plan.launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): plan.launch()
            e1.record()
        torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 5 / 50 * 1e3
    print(f"B{B} HW{HW} C{Cc}: " + "  ".join(f"{k} {v:.2f} us" for k, v in res.items()), flush=True)
