#!/usr/bin/env python3
"""Time ResNet-50 feature extraction (fp32, 1 x 3 x 600 x 1000, frozen BN folded) in three forms, each replayed from a
HIP graph:  (a) the shipped NCHW path (MIOpen convs + fused bias/residual/ReLU kernel),
            (b) channels_last: 3x3 / 7x7 convs through MIOpen NHWC, 1x1 convs as GEMMs on the [H*W, C] view
                (bias + ReLU in the GEMM epilogue), residual add + ReLU as one elementwise pass,
            (c) (b) with every conv left to MIOpen (NHWC) -- isolates what the GEMM form buys.
    python tools/backbone_probe.py [--tune 1]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def graph_time(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = fn()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


def conv1x1_gemm(x, w2d, b, relu):
    """x: NHWC-contiguous [B,C,H,W] (channels_last); w2d [Cout, Cin]; returns channels_last [B,Cout,H,W]."""
    B, C, H, W = x.shape
    x2 = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
    if relu:
        y = torch._addmm_activation(b, x2, w2d.t(), use_gelu=False)
    else:
        y = torch.addmm(b, x2, w2d.t())
    return y.view(B, H, W, -1).permute(0, 3, 1, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tune", type=int, default=1)
    ap.add_argument("--find", type=int, default=0, help="1 = MIOpen find mode (torch.backends.cudnn.benchmark)")
    a = ap.parse_args()
    if a.find:
        torch.backends.cudnn.benchmark = True
    from egtr_amd.backbone import ResNet50Features
    from egtr_amd import runtime
    dev = "cuda:0"
    torch.manual_seed(0)
    net = ResNet50Features().to(dev).eval()
    for m in net.modules():
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 1.5)
            m.running_mean.normal_(0, 0.1)
            m.weight.uniform_(0.5, 1.5)
            m.bias.normal_(0, 0.1)
    if a.tune:
        runtime.enable_gemm_tuning()
    x = torch.randn(1, 3, 600, 1000, device=dev)
    with torch.no_grad():
        t_a, fa = graph_time(lambda: net(x))
        print(f"(a) shipped NCHW path: {t_a:.3f} ms")
        P = net._folded
        CL = torch.channels_last
        stem_w = P["stem"][0].contiguous(memory_format=CL)
        stem_b = P["stem"][1]
        blocks = []
        for li in range(1, 5):
            for blk, p in zip(getattr(net, f"layer{li}"), P[li]):
                d = dict(stride=blk.conv2.stride, w1=p[0][0].flatten(1).contiguous(), b1=p[0][1],
                         w2=p[1][0].contiguous(memory_format=CL), b2=p[1][1], w3=p[2][0].flatten(1).contiguous(),
                         b3=p[2][1], w1c=p[0][0].contiguous(memory_format=CL),
                         w3c=p[2][0].contiguous(memory_format=CL))
                if len(p) >= 4:
                    d.update(wd=p[3][0].flatten(1).contiguous(), bd=p[3][1], wdc=p[3][0].contiguous(memory_format=CL),
                             dstride=blk.downsample[0].stride)
                blocks.append((li, d))
        xc = x.contiguous(memory_format=CL)

        def fwd(gemm):
            y = torch.relu_(F.conv2d(xc, stem_w, stem_b, stride=2, padding=3))
            y = F.max_pool2d(y, 3, 2, 1)
            feats = {}
            for li, d in blocks:
                idt = y
                if "wd" in d:
                    if gemm and d["dstride"] == (1, 1):
                        idt = conv1x1_gemm(y, d["wd"], d["bd"], False)
                    elif gemm:
                        idt = conv1x1_gemm(y[:, :, ::2, ::2].contiguous(memory_format=CL), d["wd"], d["bd"], False)
                    else:
                        idt = F.conv2d(y, d["wdc"], d["bd"], stride=d["dstride"])
                if gemm:
                    z = conv1x1_gemm(y, d["w1"], d["b1"], True)
                else:
                    z = torch.relu_(F.conv2d(y, d["w1c"], d["b1"]))
                z = torch.relu_(F.conv2d(z, d["w2"], d["b2"], stride=d["stride"], padding=1))
                if gemm:
                    z = conv1x1_gemm(z, d["w3"], d["b3"], False)
                else:
                    z = F.conv2d(z, d["w3c"], d["b3"])
                y = torch.relu_(z.add_(idt))
                feats[li] = y
            return [feats[2], feats[3], feats[4]]

        t_b, fb = graph_time(lambda: fwd(True))
        print(f"(b) channels_last, 1x1 as GEMM: {t_b:.3f} ms")
        t_c, fc = graph_time(lambda: fwd(False))
        print(f"(c) channels_last, all MIOpen:  {t_c:.3f} ms")
        # (d) NCHW throughout (3x3 / 7x7 / strided convs through MIOpen as shipped), stride-1 1x1 convs as W . X GEMMs on
        #     the [C, H*W] view (tuned by TunableOp instead of MIOpen's own rocBLAS call) + the shipped epilogue kernel
        from egtr_amd import ops

        def conv1x1_nchw(xin, w2d):
            B_, C_, H_, W_ = xin.shape
            return torch.matmul(w2d, xin.reshape(B_, C_, H_ * W_)).view(B_, -1, H_, W_)

        def fwd_d():
            w, b = P["stem"]
            y = net.maxpool(ops.bias_act_(F.conv2d(x, w, None, stride=2, padding=3), b))
            feats = []
            for li in range(1, 5):
                for blk, p in zip(getattr(net, f"layer{li}"), P[li]):
                    s_ = blk.conv2.stride
                    idt = y
                    if blk.downsample is not None:
                        if blk.downsample[0].stride == (1, 1):
                            idt = ops.bias_act_(conv1x1_nchw(y, p[3][0].flatten(1)), p[3][1], None, relu=False)
                        else:
                            idt = ops.bias_act_(F.conv2d(y, p[3][0], None, stride=blk.downsample[0].stride), p[3][1],
                                                None, relu=False)
                    z = ops.bias_act_(conv1x1_nchw(y, p[0][0].flatten(1)), p[0][1])
                    z = ops.bias_act_(F.conv2d(z, p[1][0], None, stride=s_, padding=1), p[1][1])
                    y = ops.bias_act_(conv1x1_nchw(z, p[2][0].flatten(1)), p[2][1], idt)
                if li in net.out_indices:
                    feats.append(y)
            return feats

        t_d, fd = graph_time(fwd_d)
        print(f"(d) NCHW, stride-1 1x1 as W.X GEMM:  {t_d:.3f} ms")

        # (e) NCHW, EVERY convolution through MIOpen (its own choice for the 1x1 ones) + the shipped epilogue kernel
        def fwd_e():
            w, b = P["stem"]
            y = net.maxpool(ops.bias_act_(F.conv2d(x, w, None, stride=2, padding=3), b))
            feats = []
            for li in range(1, 5):
                for blk, p in zip(getattr(net, f"layer{li}"), P[li]):
                    s_ = blk.conv2.stride
                    idt = y
                    if blk.downsample is not None:
                        idt = ops.bias_act_(F.conv2d(y, p[3][0], None, stride=blk.downsample[0].stride), p[3][1], None,
                                            relu=False)
                    z = ops.bias_act_(F.conv2d(y, p[0][0]), p[0][1])
                    z = ops.bias_act_(F.conv2d(z, p[1][0], None, stride=s_, padding=1), p[1][1])
                    y = ops.bias_act_(F.conv2d(z, p[2][0]), p[2][1], idt)
                if li in net.out_indices:
                    feats.append(y)
            return feats

        t_e, fe = graph_time(fwd_e)
        print(f"(e) NCHW, every convolution through MIOpen:  {t_e:.3f} ms")
        for i in range(3):
            print(f"  C{i + 3}: max |d - a| = {(fd[i] - fa[i]).abs().max().item():.3e}")
        for i in range(3):
            print(f"  C{i + 3}: max |b - a| = {(fb[i] - fa[i]).abs().max().item():.3e}  (|a| max {fa[i].abs().max().item():.2f}), "
                  f"max |c - a| = {(fc[i] - fa[i]).abs().max().item():.3e}")


if __name__ == "__main__":
    main()
