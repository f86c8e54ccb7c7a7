import sys, os, torch
sys.path.insert(0, os.getcwd())
from driving_dirty_amd import ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
feat = torch.rand(32, 128, 918, 32, device=dev) - 0.3
pooled, codes = ops.pool4_fwd_idx(feat)
gp = torch.rand_like(pooled)
print("pool4_fwd_idx %.4f ms" % timeit(lambda: ops.pool4_fwd_idx(feat), 10))
print("pool4_idx_relu_bwd %.4f ms" % timeit(lambda: ops.pool4_idx_relu_bwd(gp, codes, tuple(feat.shape)), 10))
