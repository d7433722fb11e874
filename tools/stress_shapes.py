import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
model, cfg, _ = bench.build_model(dev, {"num_queries": 300, "decoder_layers": 8})
B, H, W = 2, 800, 1333
pv = torch.randn(B, 3, H, W, device=dev)
pm = torch.ones(B, H, W, dtype=torch.long, device=dev)
pm[1, 700:, :] = 0
pm[1, :, 1200:] = 0
with torch.no_grad():
    o32 = model(pixel_values=pv, pixel_mask=pm, output_attentions=False, output_attention_states=True, output_hidden_states=True)
    print("fp32 ok", o32["pred_rel"].shape, float(o32["pred_rel"].mean()))
    mb = model.to(torch.bfloat16)
    try:
        ob = mb(pixel_values=pv.to(torch.bfloat16), pixel_mask=pm, output_attentions=False, output_attention_states=True, output_hidden_states=True)
        print("bf16 ok", ob["pred_rel"].dtype, float((ob["pred_rel"].float() - o32["pred_rel"]).abs().max()),
              float((ob["logits"].float() - o32["logits"]).abs().max()))
    except Exception as e:
        import traceback; traceback.print_exc()
