import torch, torch.nn.functional as F, sys
sys.path.insert(0, ".")
from gen6d_amd import ops
def t(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
w = torch.randn(64, 3, 3, 3, device="cuda"); b = torch.randn(64, device="cuda")
for shp in ((1, 3, 704, 928), (1, 3, 480, 640), (1, 3, 256, 320), (7, 3, 128, 128), (1, 3, 128, 128)):
    x = torch.randn(shp, device="cuda")
    a = t(lambda: ops.vgg_conv1_pool(x, w, b))
    m = t(lambda: ops.bias_relu_pool_nchw(F.conv2d(x, w, None, padding=1), b, True, True))
    print(shp, f"own {a:.1f} us  miopen+glue {m:.1f} us")
