#!/usr/bin/env python3
"""Time the ResNet-50 convolution stack (fp32, 1 x 3 x 600 x 1000) through MIOpen in NCHW and channels_last, each
replayed from a HIP graph: python tools/conv_probe.py [--find 1]"""
import argparse
import torch
import torch.nn.functional as F


def resnet50_convs():
    """(cin, cout, k, stride, pad) in execution order, with the H, W each conv sees tracked by the caller."""
    layers = [("stem", 3, 64, 7, 2, 3)]
    cfg = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]
    cin = 64
    out = []
    out.append(("conv", 3, 64, 7, 2, 3))
    out.append(("pool",))
    for planes, blocks, stride in cfg:
        for b in range(blocks):
            s = stride if b == 0 else 1
            blk = [("conv", cin, planes, 1, 1, 0), ("conv", planes, planes, 3, s, 1), ("conv", planes, planes * 4, 1, 1, 0)]
            if b == 0:
                blk.append(("down", cin, planes * 4, 1, s, 0))
            out.append(("block", blk))
            cin = planes * 4
    return out


def build(dev, fmt):
    torch.manual_seed(0)
    ws = []
    for item in resnet50_convs():
        if item[0] == "conv":
            ws.append(torch.randn(item[2], item[1], item[3], item[3], device=dev).mul_(0.05).contiguous(memory_format=fmt))
        elif item[0] == "block":
            ws.append([torch.randn(c[2], c[1], c[3], c[3], device=dev).mul_(0.05).contiguous(memory_format=fmt)
                       for c in item[1]])
        else:
            ws.append(None)
    return ws


def forward(x, ws):
    for item, w in zip(resnet50_convs(), ws):
        if item[0] == "conv":
            x = F.conv2d(x, w, None, stride=item[4], padding=item[5])
        elif item[0] == "pool":
            x = F.max_pool2d(x, 3, 2, 1)
        else:
            blk = item[1]
            idt = x
            y = F.conv2d(x, w[0])
            y = F.conv2d(y, w[1], None, stride=blk[1][4], padding=1)
            y = F.conv2d(y, w[2])
            if len(blk) == 4:
                idt = F.conv2d(x, w[3], None, stride=blk[3][4])
            x = y + idt
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--find", type=int, default=0)
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = bool(a.find)
    dev = "cuda:0"
    for name, fmt in (("NCHW", torch.contiguous_format), ("channels_last", torch.channels_last)):
        ws = build(dev, fmt)
        x = torch.randn(1, 3, 600, 1000, device=dev).contiguous(memory_format=fmt)
        with torch.no_grad():
            for _ in range(3):
                forward(x, ws)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    y = forward(x, ws)
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 20:.3f} ms per ResNet-50 conv stack (find={a.find}), out {tuple(y.shape)} "
              f"stride {y.stride()}")


if __name__ == "__main__" and "--nhwc-gemm" not in __import__("sys").argv:
    main()


def forward_nhwc_gemm(x, ws):
    """Same stack with activations as channels-last matrices [B*H*W, C]: 1x1 convolutions are GEMMs with the bias +
    ReLU epilogue of hipBLASLt (torch._addmm_activation), 3x3 / 7x7 / strided ones go to MIOpen in channels_last."""
    B = x.shape[0]
    for item, w in zip(resnet50_convs(), ws):
        if item[0] == "conv":
            x = torch.relu_(F.conv2d(x, w, None, stride=item[4], padding=item[5]))
        elif item[0] == "pool":
            x = F.max_pool2d(x, 3, 2, 1)
        else:
            blk = item[1]
            _, C, H, W = x.shape
            xm = x.permute(0, 2, 3, 1).reshape(B * H * W, C)  # view of the channels_last tensor
            w1, w2, w3 = w[0], w[1], w[2]
            b1 = torch.zeros(w1.shape[0], device=x.device)
            y = torch._addmm_activation(b1, xm, w1.view(w1.shape[0], -1).t())
            y = y.view(B, H, W, -1).permute(0, 3, 1, 2)  # channels_last view
            y = torch.relu_(F.conv2d(y, w2, None, stride=blk[1][4], padding=1))
            _, C2, H2, W2 = y.shape
            ym = y.permute(0, 2, 3, 1).reshape(B * H2 * W2, C2)
            b3 = torch.zeros(w3.shape[0], device=x.device)
            y3 = torch.addmm(b3, ym, w3.view(w3.shape[0], -1).t())
            if len(blk) == 4:
                s = blk[3][4]
                xi = x[:, :, ::s, ::s] if s > 1 else x
                xim = xi.permute(0, 2, 3, 1).reshape(B * H2 * W2, C)
                idt = torch.addmm(b3, xim, w[3].view(w[3].shape[0], -1).t())
            else:
                idt = xm
            x = torch.relu_(y3 + idt).view(B, H2, W2, -1).permute(0, 3, 1, 2)
    return x


def main_nhwc():
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_filename("/tmp/conv_probe_tunable.csv", insert_device_ordinal=True)
    torch.backends.cudnn.benchmark = True
    dev = "cuda:0"
    ws = build(dev, torch.channels_last)
    x = torch.randn(1, 3, 600, 1000, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            forward_nhwc_gemm(x, ws)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                y = forward_nhwc_gemm(x, ws)
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    print(f"NHWC, 1x1 convs as tuned GEMMs (+bias/ReLU epilogue): {e0.elapsed_time(e1) / 20:.3f} ms per ResNet-50 conv "
          f"stack, out {tuple(y.shape)}")


if __name__ == "__main__" and "--nhwc-gemm" in __import__("sys").argv:
    main_nhwc()
