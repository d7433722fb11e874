#!/usr/bin/env python3
"""Which PyTorch operators (with input shapes) are behind the remaining ATen kernels of one eager inference forward of the
bench model: python tools/op_trace.py [substring of the kernel name, default: elementwise]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else "elementwise"
    dev = torch.device("cuda:0")
    model, cfg, _ = bench.build_model(dev)
    pv = torch.randn(1, 3, bench.H_IMG, bench.W_IMG, device=dev)
    pm = torch.ones(1, bench.H_IMG, bench.W_IMG, dtype=torch.long, device=dev)
    kw = dict(output_attentions=False, output_attention_states=True, output_hidden_states=True)
    with torch.no_grad():
        for _ in range(3):
            model(pixel_values=pv, pixel_mask=pm, **kw)
        torch.cuda.synchronize()
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
            model(pixel_values=pv, pixel_mask=pm, **kw)
            torch.cuda.synchronize()
    evs = prof.events()
    for e in evs:
        if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
            for k in e.kernels:
                if pat in k.name or "Cat" in k.name or "reduce_kernel" in k.name:
                    st = [s for s in (e.stack or []) if "egtr_amd" in s][:2]
                    print(f"{k.duration:8.1f} us  {e.name:28s} {str(e.input_shapes)[:70]:70s} {k.name[:50]} | {' <- '.join(st)}")


if __name__ == "__main__":
    main()
