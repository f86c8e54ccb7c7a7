import sys, os, torch
sys.path.insert(0, os.getcwd())
from driving_dirty_amd import ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
b, h, w = 32, 256, 1836
x4 = torch.rand(b, h, w, 4, device=dev); x4[..., 3] = 0
w1 = torch.randn(32, 3, 3, 3, device=dev) * 0.2
bias = torch.randn(32, device=dev) * 0.1
d1 = ops.conv_desc(b, h, w, 3, 1)
p1 = ops.conv_pack(w1, d1, 0)
print("c1 fwd with bits %.4f ms" % timeit(lambda: ops.conv_fwd_bits(x4, p1, bias, d1), 20))
print("c1 fwd no bits   %.4f ms" % timeit(lambda: ops.conv_fwd(x4, p1, bias, d1), 20))
