"""(name, shape) lists of the reference checkpoints' parameter tensors, in the reference modules' ``state_dict()``
order (UNetModel.__init__, openaimodel.py:452-839; Decoder.__init__, model.py:541-615).  Used to create seeded
synthetic weights of the real architecture when no checkpoint is available (there is no network)."""


def _res(p, cin, cout, emb, out):
    out += [(f"{p}.in_layers.0.weight", (cin,)), (f"{p}.in_layers.0.bias", (cin,)),
            (f"{p}.in_layers.2.weight", (cout, cin, 3, 3)), (f"{p}.in_layers.2.bias", (cout,)),
            (f"{p}.emb_layers.1.weight", (cout, emb)), (f"{p}.emb_layers.1.bias", (cout,)),
            (f"{p}.out_layers.0.weight", (cout,)), (f"{p}.out_layers.0.bias", (cout,)),
            (f"{p}.out_layers.3.weight", (cout, cout, 3, 3)), (f"{p}.out_layers.3.bias", (cout,))]
    if cin != cout:
        out += [(f"{p}.skip_connection.weight", (cout, cin, 1, 1)), (f"{p}.skip_connection.bias", (cout,))]
    return [f"{p}.in_layers.0.weight", f"{p}.out_layers.0.weight"]


def _st(p, c, ctx, depth, out, linear=False):
    norms = [f"{p}.norm.weight"]
    pshape = (c, c) if linear else (c, c, 1, 1)        # use_linear_in_transformer (SDXL): nn.Linear instead of a 1x1 conv
    out += [(f"{p}.norm.weight", (c,)), (f"{p}.norm.bias", (c,)), (f"{p}.proj_in.weight", pshape), (f"{p}.proj_in.bias", (c,))]
    for i in range(depth):
        b = f"{p}.transformer_blocks.{i}"
        out += [(f"{b}.attn1.to_q.weight", (c, c)), (f"{b}.attn1.to_k.weight", (c, c)), (f"{b}.attn1.to_v.weight", (c, c)),
                (f"{b}.attn1.to_out.0.weight", (c, c)), (f"{b}.attn1.to_out.0.bias", (c,)),
                (f"{b}.ff.net.0.proj.weight", (8 * c, c)), (f"{b}.ff.net.0.proj.bias", (8 * c,)),
                (f"{b}.ff.net.2.weight", (c, 4 * c)), (f"{b}.ff.net.2.bias", (c,)),
                (f"{b}.attn2.to_q.weight", (c, c)), (f"{b}.attn2.to_k.weight", (c, ctx)), (f"{b}.attn2.to_v.weight", (c, ctx)),
                (f"{b}.attn2.to_out.0.weight", (c, c)), (f"{b}.attn2.to_out.0.bias", (c,)),
                (f"{b}.norm2.weight", (c,)), (f"{b}.norm2.bias", (c,)), (f"{b}.norm1.weight", (c,)), (f"{b}.norm1.bias", (c,)),
                (f"{b}.norm3.weight", (c,)), (f"{b}.norm3.bias", (c,))]
        norms += [f"{b}.norm2.weight", f"{b}.norm1.weight", f"{b}.norm3.weight"]
    out += [(f"{p}.proj_out.weight", pshape), (f"{p}.proj_out.bias", (c,))]
    return norms


def unet_names_shapes(cfg):
    mc, emb, ctx = cfg["model_channels"], 4 * cfg["model_channels"], cfg["context_dim"]
    out, norms = [], []
    out += [("time_embed.0.weight", (emb, mc)), ("time_embed.0.bias", (emb,)), ("time_embed.2.weight", (emb, emb)), ("time_embed.2.bias", (emb,))]
    lin = bool(cfg.get("use_linear_in_transformer"))
    if cfg.get("adm_in_channels"):                      # num_classes="sequential" (SDXL): label_emb = [Linear, SiLU, Linear]
        adm = cfg["adm_in_channels"]
        out += [("label_emb.0.0.weight", (emb, adm)), ("label_emb.0.0.bias", (emb,)), ("label_emb.0.2.weight", (emb, emb)),
                ("label_emb.0.2.bias", (emb,))]
    out += [("input_blocks.0.0.weight", (mc, cfg["in_channels"], 3, 3)), ("input_blocks.0.0.bias", (mc,))]
    chans = [mc]
    ch, bi = mc, 1
    td = list(cfg["transformer_depth"])
    nlev = len(cfg["channel_mult"])
    for lev in range(nlev):
        cout = mc * cfg["channel_mult"][lev]
        for _ in range(cfg["num_res_blocks"][lev]):
            norms += _res(f"input_blocks.{bi}.0", ch, cout, emb, out)
            ch = cout
            d = td.pop(0)
            if d > 0:
                norms += _st(f"input_blocks.{bi}.1", ch, ctx, d, out, lin)
            chans.append(ch)
            bi += 1
        if lev != nlev - 1:
            out += [(f"input_blocks.{bi}.0.op.weight", (ch, ch, 3, 3)), (f"input_blocks.{bi}.0.op.bias", (ch,))]
            chans.append(ch)
            bi += 1
    norms += _res("middle_block.0", ch, ch, emb, out)
    norms += _st("middle_block.1", ch, ctx, cfg["transformer_depth_middle"], out, lin)
    norms += _res("middle_block.2", ch, ch, emb, out)
    tdo = list(cfg["transformer_depth_output"])
    bo = 0
    for lev in reversed(range(nlev)):
        cout = mc * cfg["channel_mult"][lev]
        for i in range(cfg["num_res_blocks"][lev] + 1):
            cs = chans.pop()
            norms += _res(f"output_blocks.{bo}.0", ch + cs, cout, emb, out)
            ch = cout
            d = tdo.pop()
            j = 1
            if d > 0:
                norms += _st(f"output_blocks.{bo}.1", ch, ctx, d, out, lin)
                j = 2
            if lev > 0 and i == cfg["num_res_blocks"][lev]:
                out += [(f"output_blocks.{bo}.{j}.conv.weight", (ch, ch, 3, 3)), (f"output_blocks.{bo}.{j}.conv.bias", (ch,))]
            bo += 1
    out += [("out.0.weight", (ch,)), ("out.0.bias", (ch,)), ("out.2.weight", (cfg["out_channels"], ch, 3, 3)),
            ("out.2.bias", (cfg["out_channels"],))]
    norms.append("out.0.weight")
    return out, norms


def vae_decoder_names_shapes(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, out_ch=3):
    out, norms = [], []

    def res(p, cin, cout):
        out.extend([(f"{p}.norm1.weight", (cin,)), (f"{p}.norm1.bias", (cin,)), (f"{p}.conv1.weight", (cout, cin, 3, 3)),
                    (f"{p}.conv1.bias", (cout,)), (f"{p}.norm2.weight", (cout,)), (f"{p}.norm2.bias", (cout,)),
                    (f"{p}.conv2.weight", (cout, cout, 3, 3)), (f"{p}.conv2.bias", (cout,))])
        norms.extend([f"{p}.norm1.weight", f"{p}.norm2.weight"])
        if cin != cout:
            out.extend([(f"{p}.nin_shortcut.weight", (cout, cin, 1, 1)), (f"{p}.nin_shortcut.bias", (cout,))])
    bi = ch * ch_mult[-1]
    out += [("conv_in.weight", (bi, z_channels, 3, 3)), ("conv_in.bias", (bi,))]
    res("mid.block_1", bi, bi)
    out += [("mid.attn_1.norm.weight", (bi,)), ("mid.attn_1.norm.bias", (bi,))]
    norms.append("mid.attn_1.norm.weight")
    for n in ("q", "k", "v", "proj_out"):
        out += [(f"mid.attn_1.{n}.weight", (bi, bi, 1, 1)), (f"mid.attn_1.{n}.bias", (bi,))]
    res("mid.block_2", bi, bi)
    # the decoder builds levels top-down but registers them with up.insert(0, ...): state_dict order is up.0 first
    cins = {}
    cur = bi
    for lev in reversed(range(len(ch_mult))):
        cins[lev] = cur
        cur = ch * ch_mult[lev]
    for lev in range(len(ch_mult)):
        cin, cout = cins[lev], ch * ch_mult[lev]
        for i in range(num_res_blocks + 1):
            res(f"up.{lev}.block.{i}", cin, cout)
            cin = cout
        if lev != 0:
            out += [(f"up.{lev}.upsample.conv.weight", (cout, cout, 3, 3)), (f"up.{lev}.upsample.conv.bias", (cout,))]
    c0 = ch * ch_mult[0]
    out += [("norm_out.weight", (c0,)), ("norm_out.bias", (c0,)), ("conv_out.weight", (out_ch, c0, 3, 3)), ("conv_out.bias", (out_ch,))]
    norms.append("norm_out.weight")
    return out, norms


def vae_encoder_names_shapes(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, in_channels=3):
    """Encoder state-dict names/shapes (comfy/ldm/modules/diffusionmodules/model.py:441-500) followed by AutoencoderKL's
    ``quant_conv`` (comfy/ldm/models/autoencoder.py:160-170, embed_dim = z_channels, double_z)"""
    out, norms = [], []

    def res(p, cin, cout):
        out.extend([(f"{p}.norm1.weight", (cin,)), (f"{p}.norm1.bias", (cin,)), (f"{p}.conv1.weight", (cout, cin, 3, 3)),
                    (f"{p}.conv1.bias", (cout,)), (f"{p}.norm2.weight", (cout,)), (f"{p}.norm2.bias", (cout,)),
                    (f"{p}.conv2.weight", (cout, cout, 3, 3)), (f"{p}.conv2.bias", (cout,))])
        norms.extend([f"{p}.norm1.weight", f"{p}.norm2.weight"])
        if cin != cout:
            out.extend([(f"{p}.nin_shortcut.weight", (cout, cin, 1, 1)), (f"{p}.nin_shortcut.bias", (cout,))])
    out += [("conv_in.weight", (ch, in_channels, 3, 3)), ("conv_in.bias", (ch,))]
    cin = ch
    for lev in range(len(ch_mult)):
        cout = ch * ch_mult[lev]
        for i in range(num_res_blocks):
            res(f"down.{lev}.block.{i}", cin, cout)
            cin = cout
        if lev != len(ch_mult) - 1:
            out += [(f"down.{lev}.downsample.conv.weight", (cin, cin, 3, 3)), (f"down.{lev}.downsample.conv.bias", (cin,))]
    res("mid.block_1", cin, cin)
    out += [("mid.attn_1.norm.weight", (cin,)), ("mid.attn_1.norm.bias", (cin,))]
    norms.append("mid.attn_1.norm.weight")
    for n in ("q", "k", "v", "proj_out"):
        out += [(f"mid.attn_1.{n}.weight", (cin, cin, 1, 1)), (f"mid.attn_1.{n}.bias", (cin,))]
    res("mid.block_2", cin, cin)
    out += [("norm_out.weight", (cin,)), ("norm_out.bias", (cin,)), ("conv_out.weight", (2 * z_channels, cin, 3, 3)),
            ("conv_out.bias", (2 * z_channels,))]
    norms.append("norm_out.weight")
    out += [("quant_conv.weight", (2 * z_channels, 2 * z_channels, 1, 1)), ("quant_conv.bias", (2 * z_channels,))]
    return out, norms


def controlnet_names_shapes(cfg, hint_channels=3):
    """cldm.ControlNet state-dict names/shapes (comfy/cldm/cldm.py:30-282) for a UNet config: the UNet's time embedding,
    encoder and middle block, the 8-conv hint encoder (16,16,32,32,96,96,256 -> model_channels), one 1x1 zero conv per
    encoder output and middle_block_out."""
    ns, norms = unet_names_shapes(cfg)
    keep = ("time_embed.", "input_blocks.", "middle_block.")
    out = [(n, sh) for n, sh in ns if n.startswith(keep)]
    norm_out = [n for n in norms if n.startswith(keep)]
    mc = cfg["model_channels"]
    chans = [(0, hint_channels, 16), (2, 16, 16), (4, 16, 32), (6, 32, 32), (8, 32, 96), (10, 96, 96), (12, 96, 256), (14, 256, mc)]
    for i, cin, cout in chans:
        out += [(f"input_hint_block.{i}.weight", (cout, cin, 3, 3)), (f"input_hint_block.{i}.bias", (cout,))]
    # zero convs: one per encoder feature (conv_in, every res block, every downsample)
    zc = [mc]
    ch = mc
    nlev = len(cfg["channel_mult"])
    for lev in range(nlev):
        for _ in range(cfg["num_res_blocks"][lev]):
            ch = mc * cfg["channel_mult"][lev]
            zc.append(ch)
        if lev != nlev - 1:
            zc.append(ch)
    for i, c in enumerate(zc):
        out += [(f"zero_convs.{i}.0.weight", (c, c, 1, 1)), (f"zero_convs.{i}.0.bias", (c,))]
    out += [("middle_block_out.0.weight", (ch, ch, 1, 1)), ("middle_block_out.0.bias", (ch,))]
    # the reference module's state_dict() order (cldm.py registers zero_convs and input_hint_block before middle_block): seeded
    # synthetic weights are drawn per position in this table, so the order is part of the contract with the golden vectors
    rank = {"time_embed": 0, "input_blocks": 1, "zero_convs": 2, "input_hint_block": 3, "middle_block": 4, "middle_block_out": 5}
    out = [e for _, e in sorted(enumerate(out), key=lambda ie: (rank[ie[1][0].split(".")[0]], ie[0]))]
    return out, norm_out
