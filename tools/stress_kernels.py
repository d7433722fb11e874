import torch, json, sys
sys.path.insert(0, '/root/repo')
import bench
from egtr_amd.runtime import enable_gemm_tuning
enable_gemm_tuning()
out = bench.stress_bench(torch.device('cuda:0'), steps=6, warmup=3)
print(out['value'], out['ms_per_step'])
for k in out.get('roofline_kernels', []): print(k['kernel'], k['avg_launch_us'], k['frac'])
