#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE ITSELF (build container only).

/root/reference cannot be imported as a whole here: `diffusers`, `torchvision`, `cv2`, `comfy` and
`folder_paths` are absent.  Its own pure-torch pieces do run once those imports are
satisfied by empty stand-in *modules* (no arithmetic is stubbed except torchvision's `Resize` /
`Normalize`, which are thin wrappers over `F.interpolate(bilinear, antialias)` / `(x-mean)/std` -
assumption A.8 in SURVEY.md, recorded in the fixture metadata).  What is captured:

  G1  SDMatteApply.apply_matte (sdmatte_nodes.py:257-405) with a FakeCore standing in for the
      network: the exact `data` dict handed to the model + alpha/matted outputs for every
      output_mode x mask_refine x trimap_constraint.
  G2  custom_prepare_attention_mask (replace.py:20-72) + the mask build of meta_arch.py:200-204 and
      replace.py:401-403: additive key bias at every U-Net level for heads 5/10/20.
  G3  custom_get_attention_scores (replace.py:75-122): fp32 probabilities with / without bias.
  G4  replace_unet_conv_in / add_aux_conv_in (utils.py:13-41) weight surgery.

The outputs are DATA (inputs + expected outputs), written to tests/golden/*.npz.  This script never
runs on the GPU box (no /root/reference there); the fixtures travel instead.

usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PARENT = "/root"
REF_NAME = "reference"


def _install_stubs(tmp_models_dir):
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    # ---- torchvision.transforms: Resize / Normalize only (assumption A.8) ----
    class Resize:
        def __init__(self, size, antialias=True):
            self.size = tuple(size)
            self.antialias = antialias

        def __call__(self, x):
            if tuple(x.shape[-2:]) == self.size:
                return x
            return F.interpolate(x, size=self.size, mode="bilinear", align_corners=False, antialias=self.antialias)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, x):
            m = torch.tensor(self.mean, dtype=x.dtype)[None, :, None, None]
            s = torch.tensor(self.std, dtype=x.dtype)[None, :, None, None]
            return (x - m) / s

    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Resize=Resize, Normalize=Normalize)

    # ---- ComfyUI host modules ----
    paths = {"SDMatte": [os.path.join(tmp_models_dir, "SDMatte")], "diffusers": [os.path.join(tmp_models_dir, "diffusers")]}
    mod("folder_paths", models_dir=tmp_models_dir,
        add_model_folder_path=lambda name, p: paths.setdefault(name, []).append(p),
        get_folder_paths=lambda name: paths.get(name, []))
    comfy = mod("comfy")
    comfy.model_management = mod("comfy.model_management", get_torch_device=lambda: torch.device("cpu"))

    # ---- diffusers / cv2: import-time names only, never executed by the captured functions ----
    class _Any:
        def __init__(self, *a, **k):
            pass

    d = mod("diffusers", UNet2DConditionModel=type("UNet2DConditionModel", (torch.nn.Module,), {}),
            DDIMScheduler=_Any, AutoencoderKL=_Any)
    mod("diffusers.models")
    mod("diffusers.models.embeddings", Timesteps=_Any, TimestepEmbedding=_Any, get_timestep_embedding=None)
    mod("diffusers.models.unets")
    mod("diffusers.models.unets.unet_2d_blocks", get_down_block=None, get_up_block=None, get_mid_block=None)
    mod("diffusers.models.activations", get_activation=None)
    mod("diffusers.models.unets.unet_2d_condition", UNet2DConditionOutput=_Any)
    mod("diffusers.utils", USE_PEFT_BACKEND=False, scale_lora_layers=None, unscale_lora_layers=None)
    mod("diffusers.models.attention_processor", Attention=type("Attention", (), {}), AttnProcessor=_Any)
    mod("cv2", MORPH_ELLIPSE=2, getStructuringElement=lambda *a, **k: None)
    del d


class _AttnSelf:
    """Stand-in for the diffusers `Attention` instance the reference binds its methods onto
    (utils.py:47-52); only the attributes the two functions read."""

    def __init__(self, heads, scale):
        self.heads = heads
        self.scale = scale
        self.upcast_attention = False
        self.upcast_softmax = False


def main():
    import tempfile
    from safetensors.torch import save_file

    tmp = tempfile.mkdtemp(prefix="sdmatte_golden_")
    os.makedirs(os.path.join(tmp, "SDMatte"))
    os.makedirs(os.path.join(tmp, "diffusers", "stable-diffusion-2-1-base"))
    save_file({"dummy": torch.zeros(1)}, os.path.join(tmp, "SDMatte", "SDMatte.safetensors"))
    _install_stubs(tmp)
    sys.path.insert(0, REF_PARENT)
    import importlib
    nodes = importlib.import_module(f"{REF_NAME}.sdmatte_nodes")
    replace = importlib.import_module(f"{REF_NAME}.src.utils.replace")
    utils = importlib.import_module(f"{REF_NAME}.src.utils.utils")

    # ------------------------------------------------------------------ G1
    g = torch.Generator().manual_seed(7)
    B, H, W, S = 2, 37, 53, 64
    image = torch.rand(B, H, W, 3, generator=g)
    tri = torch.zeros(B, H, W)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    for b in range(B):
        r = torch.sqrt((yy - H / 2 - b) ** 2 + (xx - W / 2 + 2 * b) ** 2)
        tri[b][r < 16] = 0.5
        tri[b][r < 10] = 1.0
    tri = (tri + 0.02 * torch.rand(B, H, W, generator=g)).clamp(0, 1)   # not exactly three-valued
    fake_alpha = torch.rand(B, 1, S, S, generator=g) * 1.2 - 0.1          # exercises the clamp
    captured = {}

    class FakeCore(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()
            captured["ctor"] = dict(kw)

        def load_state_dict(self, sd, strict=True):
            captured["n_keys"] = len(sd)
            return None

        def forward(self, data):
            captured["data"] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
            return fake_alpha.clone()

    nodes.SDMatteCore = FakeCore                       # lazy-import guard sdmatte_nodes.py:262-264
    nodes.SDMatteApply.INPUT_TYPES()                   # must at least evaluate
    node = nodes.SDMatteApply()
    out = {"image": image.numpy(), "trimap": tri.numpy(), "fake_alpha": fake_alpha.numpy(),
           "inference_size": np.int64(S)}
    cases = []
    for mode in ("alpha_only", "matted_rgba", "matted_rgb"):
        for refine in (False, True):
            for c in ((0.8,) if not refine else (0.6, 0.8, 0.9)):
                a, m = node.apply_matte("SDMatte.safetensors", image.clone(), tri.clone(), S, False, mode, refine, c,
                                        force_cpu=True)
                tag = f"{mode}__refine{int(refine)}__c{int(round(c * 10))}"
                out[f"alpha__{tag}"] = a.numpy()
                out[f"matted__{tag}"] = m.numpy()
                cases.append(tag)
    d = captured["data"]
    out["data_image"] = d["image"].numpy()
    out["data_trimap"] = d["trimap"].numpy()
    out["data_is_trans"] = d["is_trans"].numpy()
    out["data_trimap_coords"] = d["trimap_coords"].numpy()
    out["data_caption_len"] = np.int64(len(d["caption"]))
    a, m = node.apply_matte("SDMatte.safetensors", image.clone(), tri.clone(), S, True, "alpha_only", False, 0.8, force_cpu=True)
    out["data_is_trans_transparent"] = captured["data"]["is_trans"].numpy()
    out["cases"] = np.array(cases)
    out["ctor_keys"] = np.array(sorted(captured["ctor"].keys()))
    out["input_types_required"] = np.array(list(nodes.SDMatteApply.INPUT_TYPES()["required"].keys()))
    out["return_types"] = np.array(list(nodes.SDMatteApply.RETURN_TYPES))
    out["return_names"] = np.array(list(nodes.SDMatteApply.RETURN_NAMES))
    out["display_name"] = np.array([nodes.NODE_DISPLAY_NAME_MAPPINGS["SDMatteApply"]])
    np.savez_compressed(os.path.join(HERE, "g1_node_prepost.npz"), **out)

    # ------------------------------------------------------------------ G2
    Bm, l = 2, 32
    trimap_m11 = torch.full((Bm, 1, 8 * l, 8 * l), -1.0)
    yy, xx = torch.meshgrid(torch.arange(8 * l), torch.arange(8 * l), indexing="ij")
    for b in range(Bm):
        r = torch.sqrt((yy - 4 * l - 5 * b) ** 2.0 + (xx - 4 * l + 9 * b) ** 2.0)
        trimap_m11[b, 0][r < 3.2 * l] = 0.0
        trimap_m11[b, 0][r < 2.3 * l] = 1.0
    m = (trimap_m11 + 1) / 2                                              # meta_arch.py:202
    m = F.interpolate(m, scale_factor=1 / 8, mode="nearest")              # meta_arch.py:203
    m = m.flatten(start_dim=1)                                            # meta_arch.py:204
    bias = ((1 - m) * -10000.0).unsqueeze(1)                              # replace.py:402-403
    g2 = {"trimap_m11": trimap_m11.numpy(), "attention_mask": m.numpy(), "bias_level0": bias.numpy()}
    for lev, heads in ((0, 5), (1, 10), (2, 20), (3, 20)):
        t = (l >> lev) ** 2
        pm = replace.custom_prepare_attention_mask(_AttnSelf(heads, 0.125), bias.clone(), t, Bm)
        g2[f"prepared_level{lev}_heads{heads}"] = pm.numpy()
    assert replace.custom_prepare_attention_mask(_AttnSelf(5, 0.125), None, 16, Bm) is None
    np.savez_compressed(os.path.join(HERE, "g2_mask_pyramid.npz"), **g2)

    # ------------------------------------------------------------------ G3
    g = torch.Generator().manual_seed(11)
    BH, Lq, Lk, dh = 4, 48, 256, 64
    q = torch.randn(BH, Lq, dh, generator=g)
    k = torch.randn(BH, Lk, dh, generator=g)
    v = torch.randn(BH, Lk, dh, generator=g)
    keep = (torch.rand(BH // 2, 1, Lk, generator=g) > 0.4).float()
    keep[:, :, 0] = 1.0
    kb = ((1 - keep) * -10000.0).repeat_interleave(2, dim=0)              # heads=2, image-major
    s = _AttnSelf(2, dh ** -0.5)
    p_b = replace.custom_get_attention_scores(s, q, k, kb)
    p_n = replace.custom_get_attention_scores(s, q, k, None)
    np.savez_compressed(os.path.join(HERE, "g3_attention_scores.npz"), q=q.numpy(), k=k.numpy(), v=v.numpy(),
                        key_bias=kb.numpy(), probs_bias=p_b.numpy(), probs_nobias=p_n.numpy(),
                        out_bias=torch.bmm(p_b, v).numpy(), out_nobias=torch.bmm(p_n, v).numpy(),
                        scale=np.float64(s.scale))

    # ------------------------------------------------------------------ G4
    class _U:
        pass

    u = _U()
    g = torch.Generator().manual_seed(3)
    u.conv_in = torch.nn.Conv2d(4, 320, 3, padding=1)
    with torch.no_grad():
        u.conv_in.weight.copy_(torch.randn(320, 4, 3, 3, generator=g) * 0.1)
        u.conv_in.bias.copy_(torch.randn(320, generator=g) * 0.1)
    u.config = {}
    w0, b0 = u.conv_in.weight.detach().clone(), u.conv_in.bias.detach().clone()
    torch.manual_seed(0)
    u = utils.add_aux_conv_in(u)                                          # meta_arch.py:64-65 order
    u = utils.replace_unet_conv_in(u, 2)                                  # meta_arch.py:70-71
    np.savez_compressed(os.path.join(HERE, "g4_conv_in_surgery.npz"), w0=w0.numpy(), b0=b0.numpy(),
                        conv_in_w=u.conv_in.weight.detach().numpy(), conv_in_b=u.conv_in.bias.detach().numpy(),
                        aux_w=u.aux_conv_in.weight.detach().numpy(), aux_b=u.aux_conv_in.bias.detach().numpy(),
                        in_channels=np.int64(u.config["in_channels"]))
    print("golden vectors written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("  ", f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
