"""`SDMatte` model-load API on top of the native engine.

Mirrors the constructor / `load_state_dict` / `eval` / `to` / `__call__(data)` surface of the reference core
(/root/reference/src/modeling/SDMatte/meta_arch.py:30-77,127) as used by the node
(/root/reference/sdmatte_nodes.py:286-296,321-323,358), but owns no torch modules: weights go straight
from the checkpoint tensors into the engine's packed fp16 arena (HIP pack kernels), and the forward is
one C-ABI call.  The configuration the node uses (trimap aux input, trimap-latent cross-attention
context, trimap key mask, no noise) is the one implemented; the other prompt types of the reference
core (point/bbox/mask) are "next" scope and raise NotImplementedError.
"""
import torch

from .config import SDMatteConfig
from .engine import Engine

_TRIMAP_LISTS = ["point_mask", "bbox_mask", "mask", "trimap"]


class SDMatte:
    def __init__(self, pretrained_model_name_or_path=None, conv_scale=3, num_inference_steps=1, aux_input="bbox_mask",
                 use_aux_input=False, use_coor_input=True, use_dis_loss=True, use_attention_mask=True,
                 use_encoder_attention_mask=False, add_noise=False, attn_mask_aux_input=("point_mask", "bbox_mask", "mask"),
                 aux_input_list=("point_mask", "bbox_mask", "mask"), use_encoder_hidden_states=True, residual_connection=False,
                 use_attention_mask_list=(True, True, True), use_encoder_hidden_states_list=(True, True, True), load_weight=True,
                 config: SDMatteConfig = None, stream_f32: bool = True):
        # `pretrained_model_name_or_path` only supplied SD-2.1 config JSONs to the reference (meta_arch.py:95-118);
        # the constants are embedded (config.py), so it is accepted and ignored.
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        unsupported = []
        if aux_input != "trimap" or not use_aux_input:
            unsupported.append(f"aux_input={aux_input!r}/use_aux_input={use_aux_input} (only the trimap prompt is implemented)")
        if add_noise or num_inference_steps != 1:
            unsupported.append("add_noise / multi-step inference")
        if not use_encoder_hidden_states or not all(use_encoder_hidden_states_list):
            unsupported.append("text-encoder context (use_encoder_hidden_states=False)")
        if not use_attention_mask or not all(use_attention_mask_list) or "trimap" not in attn_mask_aux_input:
            unsupported.append("attention without the trimap key mask")
        if use_encoder_attention_mask:
            unsupported.append("encoder attention mask")
        if not use_coor_input:
            unsupported.append("use_coor_input=False")
        if unsupported:
            raise NotImplementedError("SDMatte (MI355X engine): unsupported configuration: " + "; ".join(unsupported))
        self.aux_input = aux_input
        self.config = config or SDMatteConfig.full()
        self.stream_f32 = stream_f32
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
            self.engine = Engine(self.config, idx, self.stream_f32)
            if self._pending is not None:
                self._upload()
        return self

    def _upload(self):
        self.missing_keys, self.ignored = self.engine.load_state_dict(self._pending)
        self._pending = None
        if self.missing_keys:
            print(f"[SDMatte] warning: {len(self.missing_keys)} expected tensors absent from the checkpoint "
                  f"(first: {self.missing_keys[0]}); they stay zero (strict=False semantics)")

    def __call__(self, data):
        return self.forward(data)

    @torch.no_grad()
    def forward(self, data):
        """data: {"image" [B,3,S,S] in [-1,1], "trimap" [B,1,S,S] in [-1,1], "is_trans" int [B],
        "trimap_coords" [B,4] (optional), "caption" ignored} -> alpha [B,1,S,S] fp32 (meta_arch.py:127-261)."""
        if self.engine is None:
            raise RuntimeError("SDMatte: call .to('cuda') (and load_state_dict) before forward")
        img, tri = data["image"], data["trimap"]
        it = data.get("is_trans")
        it = it.detach().cpu().numpy() if torch.is_tensor(it) else it
        co = data.get("trimap_coords")
        co = co.detach().cpu().float().numpy() if torch.is_tensor(co) else co
        return self.engine.forward(img, tri, is_trans=it, coords=co)
