import sys, time, torch
sys.path.insert(0, '/root/repo')
import studiosr_amd as S
dev = torch.device('cuda')
for kind, B in (("SwinIR", 8), ("EDSR", 16)):
    m = getattr(S, kind)(scale=4).eval().to(dev).set_precision("fp32")
    x = torch.rand(B, 3, 64, 64, device=dev)
    with torch.no_grad():
        for _ in range(2): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(kind, "fp32 exact path:", round(dt * 1e3, 2), "ms per batch of", B, "=", round(B * 0.065536 / dt, 1), "HR-Mpix/s")
