#!/usr/bin/env python3
"""One shape of the split-bf16 GEMM, 20 plain launches (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egtr_amd import ops  # noqa: E402

M, K, N = 12537, int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") / K ** 0.5
b = torch.randn(N, device="cuda")
wt = ops.gemm_split_weights(w)
with torch.no_grad():
    for _ in range(20):
        ops.linear_split_bf16(x, wt, b, N)
torch.cuda.synchronize()
