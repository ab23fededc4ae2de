#!/usr/bin/env python3
"""Diagnostic: config-5 network (conditional PUNetG-64, 4x256x256) single evaluation against the CPU oracle, under the
precision / fusion variants, to localise a full-size discrepancy."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import embedder_ref, punetg_ref  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    import diffsci_amd.models as M
    dev = torch.device("cuda:0")
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    mcfg = dict(input_channels=4, output_channels=4)
    torch.manual_seed(0)
    net = M.PUNetG(M.PUNetGConfig(**mcfg), conditional_embedding=M.nets.PorosityEmbedder(dembed=64))
    with torch.no_grad():
        for k, w in net.state_dict().items():
            if "gnorm" in k or k.endswith("bias"):
                w.add_(0.1 * torch.randn_like(w))
    sd = {k: w.detach().clone() for k, w in net.state_dict().items()}
    ocfg = punetg_ref.default_config(**mcfg)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, size, size, generator=g) * 0.56
    t = torch.tensor([0.265])
    y = {"porosity": torch.tensor([[0.2]])}
    t0 = time.time()
    with torch.inference_mode():
        ye = embedder_ref.porosity_embed(sd, "conditional_embedding.", y)
        want = punetg_ref.punetg_forward(sd, ocfg, x, t, ye)
        sd64 = {k: w.double() for k, w in sd.items()}
        want64 = punetg_ref.punetg_forward(sd64, ocfg, x.double(), t.double(), ye.double())
        wantu = punetg_ref.punetg_forward(sd, ocfg, x, t, None)
    print(f"oracle {time.time()-t0:.1f}s; oracle fp32 vs fp64: {rel(want, want64):.3e}", flush=True)
    net = net.to(dev).eval()
    yd = {"porosity": y["porosity"].to(dev)}
    for prec in ("fp16x3", "bf16x6", "fp32"):
        for fuse in (True, False):
            if prec != "fp16x3" and fuse:
                continue
            net.conv_precision, net.fuse_norm = prec, fuse
            with torch.inference_mode():
                got = net(x.to(dev), t.to(dev), yd).cpu()
                gotu = net(x.to(dev), t.to(dev)).cpu()
            print(f"{prec:7s} fuse={fuse}: cond vs f32 {rel(got, want):.3e}  vs f64 {rel(got, want64):.3e}   uncond vs f32 {rel(gotu, wantu):.3e}", flush=True)
    # stages (fp16x3, fused): encoder output, bottom, decoder
    net.conv_precision, net.fuse_norm = "fp16x3", True
    import torch.nn.functional as F
    with torch.inference_mode():
        te_o = punetg_ref.fourier_features(t, sd["time_projection.W"]) + ye
        te_g = net.embed_time(t.to(dev), net.embed_condition(yd))
        print("te", rel(te_g.cpu(), te_o))
        h_o = punetg_ref.conv3x3(sd, "convin", x)
        h_g = net._conv(net.convin, x.to(dev), net.packed_weights())
        print("convin", rel(h_g.cpu(), h_o))
        xe_g, skips_g = net.encode(h_g, te_g)
        # oracle encode
        ho = h_o
        skips_o = []
        for lv in range(2):
            for r in range(2):
                ho = punetg_ref.resnet_block(sd, f"downward_blocks.{lv}.{r}.", ho, te_o)
            skips_o.append(ho)
            ho = punetg_ref.conv3x3(sd, f"downsamplers.{lv}.conv", F.max_pool2d(ho, 2))
        for i, (a, b) in enumerate(zip(skips_g, skips_o)):
            print(f"skip{i}", rel(a.cpu(), b))
        print("encoded", rel(xe_g.cpu(), ho))
        hb = ho
        for r in range(2):
            hb = punetg_ref.resnet_block(sd, f"before_block.{r}.", hb, te_o)
        gb = net.resnet_block_forward(xe_g, te_g, net.before_block)
        print("before", rel(gb.cpu(), hb))
        xa = hb
        xa = punetg_ref.resnet_block(sd, "attn_resnet_block.0.", xa, te_o)
        ga = net.resnet_block_forward(gb, te_g, [net.attn_resnet_block[0]])
        print("attn_res0", rel(ga.cpu(), xa))
        xa2 = punetg_ref.attention_2d(sd, "attn_block.0.", xa)
        ga2 = net._attention(net.attn_block[0], ga, net.packed_weights(), net._ws)
        print("attention (own input)", rel(ga2.cpu(), xa2))
        ga2b = net._attention(net.attn_block[0], xa.to(dev), net.packed_weights(), net._ws)
        print("attention (oracle input)", rel(ga2b.cpu(), xa2))
        net.conv_precision = "fp32"
        ga2c = net._attention(net.attn_block[0], xa.to(dev), net.packed_weights(), net._ws)
        print("attention fp32 (oracle input)", rel(ga2c.cpu(), xa2))
        xa64 = punetg_ref.attention_2d(sd64, "attn_block.0.", xa.double())
        print("oracle attention f32 vs f64", rel(xa2, xa64), " gpu fp16x3 vs f64", rel(ga2b.cpu(), xa64), " gpu fp32 vs f64", rel(ga2c.cpu(), xa64))


if __name__ == "__main__":
    main()
