"""CompVis (ldm UNetModel) -> diffusers (UNet2DConditionModel) checkpoint export, the save path of the SD unlearning scripts.

Mirrors /root/reference/SD/train-scripts/convertModels.py:1006-1128 ``savemodelDiffusers`` as the scripts call it
(nsfw_removal.py:217-244 ``save_model``): the UNet weights ``model.diffusion_model.*`` of a CompVis state dict are re-keyed
for ``UNet2DConditionModel`` (create_unet_diffusers_config :242-301 + convert_ldm_unet_checkpoint :348-591) and written with
``torch.save``; tensors are passed through unchanged.  The mapping below is derived from the two block structures (it is not a
table copied from the script): an ldm ``input_blocks`` entry is conv_in, a (ResBlock [, SpatialTransformer]) pair or a
Downsample; ``output_blocks`` entries are (ResBlock [, SpatialTransformer] [, Upsample]).  tests/golden/compvis_export.npz holds
the key mapping the imported reference function produces for the v1-inference.yaml UNet and for a small config.
"""
import torch

UNET_PREFIX = "model.diffusion_model."

_RES = {"in_layers.0": "norm1", "in_layers.2": "conv1", "emb_layers.1": "time_emb_proj", "out_layers.0": "norm2",
        "out_layers.3": "conv2", "skip_connection": "conv_shortcut"}


def unet_diffusers_config(model_channels=320, channel_mult=(1, 2, 4, 4), attention_resolutions=(4, 2, 1), num_res_blocks=2,
                          in_channels=4, out_channels=4, context_dim=768, num_heads=8, image_size=512, vae_ch_mult=(1, 2, 4, 4)):
    """create_unet_diffusers_config (:242-301) for the options of v1-inference.yaml."""
    boc = [model_channels * m for m in channel_mult]
    down, res = [], 1
    for i in range(len(boc)):
        down.append("CrossAttnDownBlock2D" if res in attention_resolutions else "DownBlock2D")
        if i != len(boc) - 1:
            res *= 2
    up = []
    for _ in range(len(boc)):
        up.append("CrossAttnUpBlock2D" if res in attention_resolutions else "UpBlock2D")
        res //= 2
    return dict(sample_size=image_size // 2 ** (len(vae_ch_mult) - 1), in_channels=in_channels, out_channels=out_channels,
                down_block_types=tuple(down), up_block_types=tuple(up), block_out_channels=tuple(boc), layers_per_block=num_res_blocks,
                cross_attention_dim=context_dim, attention_head_dim=num_heads, use_linear_projection=False)


def _rename_resnet(rest):
    for old, new in _RES.items():
        if rest.startswith(old + "."):
            return new + rest[len(old):]
    raise KeyError(f"unexpected ResBlock parameter {rest!r}")


def unet_key_map(keys, layers_per_block=2):
    """ldm UNet parameter name (without the ``model.diffusion_model.`` prefix) -> diffusers name, for every name in ``keys``."""
    keys = list(keys)
    L = layers_per_block
    # which sub-module index of each input / output block is what: .0 is always the ResBlock (or the lone conv / Downsample)
    def subs(prefix, i):
        return sorted({k.split(".")[2] for k in keys if k.startswith(f"{prefix}.{i}.")}, key=int)
    out = {}
    for k in keys:
        p = k.split(".")
        if p[0] == "time_embed":
            out[k] = "time_embedding.linear_%d.%s" % ({"0": 1, "2": 2}[p[1]], p[2])
        elif p[0] == "out":
            out[k] = ("conv_norm_out." if p[1] == "0" else "conv_out.") + p[2]
        elif p[0] == "input_blocks":
            i, sub, rest = int(p[1]), p[2], ".".join(p[3:])
            if i == 0:
                out[k] = "conv_in." + rest
                continue
            block, layer = (i - 1) // (L + 1), (i - 1) % (L + 1)
            if rest.startswith("op."):                        # Downsample closes the level
                out[k] = f"down_blocks.{block}.downsamplers.0.conv.{rest[3:]}"
            elif sub == "0":
                out[k] = f"down_blocks.{block}.resnets.{layer}.{_rename_resnet(rest)}"
            else:
                out[k] = f"down_blocks.{block}.attentions.{layer}.{rest}"
        elif p[0] == "middle_block":
            sub, rest = p[1], ".".join(p[2:])
            out[k] = {"0": "mid_block.resnets.0." + _rename_resnet(rest) if sub == "0" else None,
                      "1": "mid_block.attentions.0." + rest if sub == "1" else None,
                      "2": "mid_block.resnets.1." + _rename_resnet(rest) if sub == "2" else None}[sub]
        elif p[0] == "output_blocks":
            i, sub, rest = int(p[1]), p[2], ".".join(p[3:])
            block, layer = i // (L + 1), i % (L + 1)
            present = subs("output_blocks", i)
            if sub == "0":
                out[k] = f"up_blocks.{block}.resnets.{layer}.{_rename_resnet(rest)}"
            elif rest.startswith("conv.") and sub == present[-1] and not any(
                    kk.startswith(f"output_blocks.{i}.{sub}.norm.") for kk in keys):
                out[k] = f"up_blocks.{block}.upsamplers.0.conv.{rest[5:]}"          # Upsample: the last sub-module, a bare conv
            else:
                out[k] = f"up_blocks.{block}.attentions.{layer}.{rest}"
        else:
            raise KeyError(f"unexpected UNet parameter {k!r}")
    return out


def compvis_unet_to_diffusers(checkpoint, layers_per_block=2):
    """CompVis state dict (optionally wrapped as {"state_dict": ...}; keys ``model.diffusion_model.*`` plus whatever else the
    LatentDiffusion module holds) -> diffusers UNet2DConditionModel state dict (same tensors, new keys)."""
    sd = checkpoint.get("state_dict", checkpoint)
    unet = {k[len(UNET_PREFIX):]: v for k, v in sd.items() if k.startswith(UNET_PREFIX)}
    if not unet:
        raise KeyError(f"no {UNET_PREFIX}* entries: not a CompVis LatentDiffusion state dict")
    m = unet_key_map(unet.keys(), layers_per_block)
    return {m[k]: v for k, v in unet.items()}


def save_model(state_dict, path_compvis=None, path_diffusers=None, layers_per_block=2):
    """nsfw_removal.py:217-244 save_model: the CompVis state dict as it is, and the converted UNet (models/<name>/<name with
    'compvis' -> 'diffusers'>.pt in the reference's layout)."""
    if path_compvis:
        torch.save(state_dict, path_compvis)
    if path_diffusers:
        torch.save(compvis_unet_to_diffusers(state_dict, layers_per_block), path_diffusers)
