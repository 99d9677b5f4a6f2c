"""`SDMatte` model-load API on top of the native engine.

Mirrors the constructor / `load_state_dict` / `eval` / `to` / `__call__(data)` surface of the reference core
(/root/reference/src/modeling/SDMatte/meta_arch.py:30-77,127) as used by the node
(/root/reference/sdmatte_nodes.py:286-296,321-323,358), but owns no torch modules: weights go straight
from the checkpoint tensors into the engine's packed fp16 arena (HIP pack kernels), and the forward is
one C-ABI call.  All prompt types of the reference core are served by the same engine graph
(meta_arch.py:22-28,131-206): the aux image `data[aux_input]` ("trimap", "bbox_mask", "mask", "auto_mask",
"point_mask") is VAE-encoded and used as cross-attention context; its coordinates go through
bbox_embedding (4 values) or, for point prompts, point_embedding; the aux key mask of the self-attention
is applied when `aux_input` is listed in `attn_mask_aux_input`.  What stays unsupported raises
NotImplementedError: noise / multi-step inference, the CLIP text context, random prompt choice
(aux_input=None), partial attention-mask / context stage lists.
"""
import os

import torch

from .config import SDMatteConfig
from .engine import Engine

# meta_arch.py:22-28: prompt image key -> coordinate key of `data`
AUX_INPUT_DIT = {"auto_mask": "auto_coords", "point_mask": "point_coords", "bbox_mask": "bbox_coords", "mask": "mask_coords",
                 "trimap": "trimap_coords"}


class SDMatte:
    def __init__(self, pretrained_model_name_or_path=None, conv_scale=3, num_inference_steps=1, aux_input="bbox_mask",
                 use_aux_input=False, use_coor_input=True, use_dis_loss=True, use_attention_mask=True,
                 use_encoder_attention_mask=False, add_noise=False, attn_mask_aux_input=("point_mask", "bbox_mask", "mask"),
                 aux_input_list=("point_mask", "bbox_mask", "mask"), use_encoder_hidden_states=True, residual_connection=False,
                 use_attention_mask_list=(True, True, True), use_encoder_hidden_states_list=(True, True, True), load_weight=True,
                 config: SDMatteConfig = None, stream_f32: bool = True, precision=None):
        # `pretrained_model_name_or_path` only supplied SD-2.1 config JSONs to the reference (meta_arch.py:95-118);
        # the constants are embedded (config.py), so it is accepted and ignored.
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        unsupported = []
        if aux_input not in AUX_INPUT_DIT:
            unsupported.append(f"aux_input={aux_input!r} (expected one of {sorted(AUX_INPUT_DIT)}; random choice per call is a training feature)")
        if not use_aux_input:
            unsupported.append("use_aux_input=False (the reference itself cannot run it: torch.cat([rgb_latent, None]), meta_arch.py:244)")
        if add_noise or num_inference_steps != 1:
            unsupported.append("add_noise / multi-step inference")
        if not use_encoder_hidden_states or not all(use_encoder_hidden_states_list):
            unsupported.append("text-encoder context (use_encoder_hidden_states=False)")
        if not all(use_attention_mask_list):
            unsupported.append("per-stage attention-mask lists other than [True]*3")
        if use_encoder_attention_mask:
            unsupported.append("encoder attention mask")
        if unsupported:
            raise NotImplementedError("SDMatte (MI355X engine): unsupported configuration: " + "; ".join(unsupported))
        self.use_coor_input = bool(use_coor_input)
        # meta_arch.py:199-206: the aux image masks the self-attention keys only for the listed prompt types
        self.use_attention_mask = bool(use_attention_mask) and aux_input in tuple(attn_mask_aux_input)
        self.aux_input = aux_input
        self.config = config or SDMatteConfig.full()
        self.stream_f32 = stream_f32
        self.precision = precision          # None -> engine.DEFAULT_PRECISION ("fp16x3": within 1e-3 of the reference's fp32 path)
        self.engine = None
        self._pending = None
        self.training = False
        self.missing_keys, self.ignored = [], 0

    # ---- nn.Module-like surface used by the node -------------------------------------------------
    def load_state_dict(self, state_dict, strict=False):
        self._pending = state_dict
        if self.engine is not None:
            self._upload()
        return self

    def eval(self):
        self.training = False
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the MI355X-native SDMatte engine runs on a gfx950 GPU only; it has no CPU path "
                               f"(requested device: {device})")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self.engine is None or self.engine.device != idx:
            if self.engine is not None:
                self.engine.close()
            self.engine = Engine(self.config, idx, self.stream_f32, precision=self.precision)
            if self._pending is not None:
                self._upload()
        return self

    def _upload(self):
        self.missing_keys, self.ignored = self.engine.load_state_dict(self._pending)
        self._pending = None
        # The reference loads with strict=False and would silently keep its random initialisation; a native engine would run on
        # zeros.  What THIS model's graph consumes must be present (a checkpoint without it is a wrong / truncated file: fail
        # loudly; SDMATTE_ALLOW_MISSING_KEYS=1 restores the lenient behaviour for experiments); tensors that only another prompt
        # type reads (unet.point_embedding.* is consumed by point prompts alone, replace.py:446-450) are reported and left zero,
        # like the reference's strict=False.
        optional = [k for k in self.missing_keys if k.startswith("unet.point_embedding.") and self.aux_input != "point_mask"]
        required = [k for k in self.missing_keys if k not in optional]
        if optional:
            print(f"[SDMatte] note: {len(optional)} tensors not used by aux_input={self.aux_input!r} are absent from the checkpoint "
                  f"(first: {optional[0]})")
        if required:
            msg = (f"[SDMatte] {len(required)} tensors the model needs are absent from the checkpoint "
                   f"(first: {', '.join(required[:3])}); tools/check_checkpoint.py lists every difference from the expected key schema")
            if os.environ.get("SDMATTE_ALLOW_MISSING_KEYS") == "1":
                print(msg + "; they stay zero")
            else:
                raise RuntimeError(msg)

    def __call__(self, data):
        return self.forward(data)

    @torch.no_grad()
    def forward(self, data):
        """data: {"image" [B,3,S,S] in [-1,1], data[aux_input] [B,1,S,S] in [-1,1], "is_trans" int [B],
        data[AUX_INPUT_DIT[aux_input]] coordinates ([B,4], or [B,N] for point prompts), "caption" ignored}
        -> alpha [B,1,S,S] fp32 (meta_arch.py:127-261)."""
        if self.engine is None:
            raise RuntimeError("SDMatte: call .to('cuda') (and load_state_dict) before forward")
        img, aux = data["image"], data[self.aux_input]
        it = data.get("is_trans")
        it = it.detach().cpu().numpy() if torch.is_tensor(it) else it
        coor_name = AUX_INPUT_DIT[self.aux_input]
        co = data.get(coor_name)
        co = co.detach().cpu().float().numpy() if torch.is_tensor(co) else co
        if coor_name == "point_coords":
            if co is None:
                raise KeyError("point prompt: data['point_coords'] is required (meta_arch.py:151)")
            if not self.use_coor_input:
                co = co * 0.0                                   # meta_arch.py:160,170-176: zero coordinates, same padding
            return self.engine.forward(img, aux, is_trans=it, point_coords=co, use_attention_mask=self.use_attention_mask)
        if not self.use_coor_input:
            co = None                                           # meta_arch.py:188-197: default box [0,0,1,1]
        return self.engine.forward(img, aux, is_trans=it, coords=co, use_attention_mask=self.use_attention_mask)
