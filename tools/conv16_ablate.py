import sys, torch
sys.path.insert(0, "/root/repo")
from gen6d_amd import lib, ops
lib.load()
dev = torch.device("cuda", 0)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(1)
for (n, h, w, ci, co) in ((16, 176, 232, 256, 256), (16, 88, 116, 512, 512), (16, 352, 464, 64, 128)):
    wt = ((torch.rand((co, 9, ci), generator=g) * 2 - 1) * 0.02).to(dev)
    x32 = torch.rand((n, h, w, ci), generator=g).to(dev)
    bias = torch.zeros(co, device=dev)
    fl = 2.0 * n * h * w * co * 9 * ci
    for tag, mode, layout in (("fp16 LDS-B", 2, 0), ("fp16 reg-B", 2, 1), ("pairs reg-B", 3, 1)):
        f = ops.conv16_pack(wt, mode, layout)
        x = torch.stack([x32.half(), (x32 - x32.half().float()).half()], -2).contiguous() if mode == 3 else x32.half()
        row = []
        for abl in (0, 1, 2, 3):
            lib.set_knob("c16_ablate", abl)
            t = timed(lambda: ops.conv16_direct_multi([x], f, bias, relu=True, full="t16"))
            row.append(f"abl{abl}: {t:.0f} us ({fl / t / 1e6:.0f} TF)")
        lib.set_knob("c16_ablate", 0)
        print(n, h, w, ci, co, tag, " | ".join(row))
