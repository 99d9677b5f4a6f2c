"""ctypes binding of the C ABI in include/sdmatte.h (libsdmatte_hip.so, hipcc/gfx950).

This is the only bridge between the Python host code (node, model-load API) and the hand-written HIP
kernels.  There is NO fallback: if the shared library is missing or no MI355X-class GPU is visible,
`load_library()` / `Engine()` raise.  PyTorch tensors are used only as device/host memory holders at the
node boundary (`tensor.data_ptr()`).
"""
import ctypes as C
import os

import numpy as np
import torch

from .config import SDMatteConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsdmatte_hip.so")

SDM_PTR_HOST, SDM_PTR_DEVICE = 0, 1
SDM_F32, SDM_F16, SDM_BF16 = 0, 1, 2
_DT = {torch.float32: SDM_F32, torch.float16: SDM_F16, torch.bfloat16: SDM_BF16}


class SdmConfig(C.Structure):
    _fields_ = [
        ("vae_channels", C.c_int32 * 4), ("vae_layers_per_block", C.c_int32),
        ("unet_channels", C.c_int32 * 4), ("unet_heads", C.c_int32 * 4), ("unet_layers_per_block", C.c_int32),
        ("cross_attention_dim", C.c_int32), ("unet_in_channels", C.c_int32), ("unet_out_channels", C.c_int32),
        ("bbox_embeddings_input_dim", C.c_int32), ("groups", C.c_int32),
        ("vae_eps", C.c_float), ("unet_res_eps", C.c_float), ("unet_tf_gn_eps", C.c_float), ("unet_ln_eps", C.c_float),
        ("vae_scaling_factor", C.c_float), ("attn_mask_value", C.c_float),
        ("stream_f32", C.c_int32), ("point_embeddings_input_dim", C.c_int32), ("precise_mask", C.c_int32), ("reserved", C.c_int32 * 5),
    ]


# sdm_precise_stage bits (include/sdmatte.h)
PRECISE_VAE_ENC, PRECISE_VAE_DEC, PRECISE_VAE_ATTN_LIN, PRECISE_UNET_RES, PRECISE_UNET_TF, PRECISE_UNET_ATTN = 1, 2, 4, 8, 16, 32
PRECISE_ALL = 63
PRECISIONS = {
    # fp16 MFMA operands everywhere: fastest; alpha within ~4e-3 of the reference's fp32 CPU path (the rounding floor of any
    # fp16-operand evaluation of this graph, the reference's own CUDA autocast path included)
    "fp16": 0,
    # split-fp16 operands (hi + lo, 3 MFMAs per product) and fp32 activations in every stage: alpha within 1e-3 (measured ~1e-4)
    "fp16x3": PRECISE_ALL,
}
DEFAULT_PRECISION = os.environ.get("SDMATTE_PRECISION", "fp16x3")


def precise_mask_of(precision) -> int:
    if precision is None:
        precision = DEFAULT_PRECISION
    if isinstance(precision, int):
        return precision & PRECISE_ALL
    if precision not in PRECISIONS:
        raise ValueError(f"unknown precision {precision!r}; expected one of {sorted(PRECISIONS)} or a stage bit mask")
    return PRECISIONS[precision]


def to_c_config(cfg: SDMatteConfig, stream_f32: bool = True, precision=None) -> SdmConfig:
    c = SdmConfig()
    for i in range(4):
        c.vae_channels[i] = cfg.vae_channels[i]
        c.unet_channels[i] = cfg.unet_channels[i]
        c.unet_heads[i] = cfg.unet_heads[i]
    c.vae_layers_per_block = cfg.vae_layers_per_block
    c.unet_layers_per_block = cfg.unet_layers_per_block
    c.cross_attention_dim = cfg.cross_attention_dim
    c.unet_in_channels = cfg.unet_in_channels
    c.unet_out_channels = cfg.unet_out_channels
    c.bbox_embeddings_input_dim = cfg.bbox_embeddings_input_dim
    c.groups = cfg.unet_groups
    c.vae_eps, c.unet_res_eps = cfg.vae_eps, cfg.unet_res_eps
    c.unet_tf_gn_eps, c.unet_ln_eps = cfg.unet_tf_gn_eps, cfg.unet_ln_eps
    c.vae_scaling_factor, c.attn_mask_value = cfg.vae_scaling_factor, cfg.attn_mask_value
    c.stream_f32 = 1 if stream_f32 else 0
    c.point_embeddings_input_dim = cfg.point_embeddings_input_dim
    c.precise_mask = precise_mask_of(precision)
    return c


EXPORTS = [
    "sdm_default_config", "sdm_create", "sdm_destroy", "sdm_last_error", "sdm_load_tensor", "sdm_finalize_weights",
    "sdm_weight_stats", "sdm_missing_key", "sdm_weight_blob_bytes", "sdm_export_weight_blob", "sdm_import_weight_blob",
    "sdm_host_blob_bytes", "sdm_export_host_blob", "sdm_import_host_blob", "sdm_forward", "sdm_forward_ex", "sdm_forward_rect", "sdm_apply_matte", "sdm_apply_matte_node",
    "sdm_synchronize", "sdm_release_memory", "sdm_resident_bytes", "sdm_weight_bytes", "sdm_last_forward_ms", "sdm_profile_enable", "sdm_profile_count", "sdm_profile_get", "sdm_profile_dump",
    "sdm_op_conv", "sdm_op_conv_ex", "sdm_op_gemm_p3", "sdm_debug_run_layer", "sdm_debug_set_input_cmask", "sdm_debug_temb_row", "sdm_conv_num_cfgs", "sdm_bench_conv", "sdm_bench_attn", "sdm_bench_gemm_p3", "sdm_op_groupnorm", "sdm_op_layernorm", "sdm_op_attention", "sdm_op_attention_split", "sdm_op_resize_aa",
    "sdm_op_mask_bias",
    "sdm_set_option", "sdm_get_option", "sdm_reset_options", "sdm_option_name", "sdm_option_help", "sdm_kernel_counts", "sdm_kernel_counts_reset",
]


class Bindings:
    """Typed view of an already dlopen()ed engine library."""

    def __init__(self, cdll):
        self.dll = cdll
        vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
        sig = {
            "sdm_default_config": (None, [C.POINTER(SdmConfig)]),
            "sdm_create": (i32, [C.POINTER(vp), i32, C.POINTER(SdmConfig)]),
            "sdm_destroy": (None, [vp]),
            "sdm_last_error": (C.c_char_p, [vp]),
            "sdm_load_tensor": (i32, [vp, C.c_char_p, i32, i32, C.POINTER(i64), vp]),
            "sdm_finalize_weights": (i32, [vp]),
            "sdm_weight_stats": (i32, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
            "sdm_missing_key": (C.c_char_p, [vp, i64]),
            "sdm_weight_blob_bytes": (i64, [vp]),
            "sdm_export_weight_blob": (i32, [vp, vp]),
            "sdm_import_weight_blob": (i32, [vp, vp]),
            "sdm_host_blob_bytes": (i64, [vp]),
            "sdm_export_host_blob": (i32, [vp, vp]),
            "sdm_import_host_blob": (i32, [vp, vp]),
            "sdm_forward": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, i32, vp]),
            "sdm_forward_ex": (i32, [vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, vp, i32, vp]),
            "sdm_forward_rect": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, i32, i32, i32, vp, i32, vp]),
            "sdm_apply_matte": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
            "sdm_apply_matte_node": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, C.c_double, vp, vp, i32, vp]),
            "sdm_synchronize": (i32, [vp]),
            "sdm_release_memory": (i32, [vp]),
            "sdm_resident_bytes": (i64, [vp]),
            "sdm_weight_bytes": (i64, [vp]),
            "sdm_last_forward_ms": (f32, [vp]),
            "sdm_profile_enable": (i32, [vp, i32]),
            "sdm_profile_count": (i32, [vp]),
            "sdm_profile_dump": (C.c_char_p, [vp]),
            "sdm_profile_get": (i32, [vp, i32, C.POINTER(C.c_char_p), C.POINTER(f32), C.POINTER(i64), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double)]),
            "sdm_op_conv": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, i32, i32,
                                  f32, i32]),
            "sdm_op_conv_ex": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, i32, i32,
                                     f32, i32, i32, vp, vp, f32, i32, i32]),
            "sdm_op_gemm_p3": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, f32, i32, vp, vp, C.POINTER(i32)]),
            "sdm_debug_run_layer": (i32, [vp, C.c_char_p, vp, i32, i32, i32, vp, i32]),
            "sdm_debug_set_input_cmask": (i32, [vp, vp]),
            "sdm_debug_temb_row": (i32, [vp, i32, i32, vp, vp, i32]),
            "sdm_conv_num_cfgs": (i32, [i32, i32]),
            "sdm_bench_conv": (f32, [vp] + [i32] * 11),
            "sdm_bench_attn": (f32, [vp] + [i32] * 7),
            "sdm_bench_gemm_p3": (f32, [vp, C.c_long, i32, i32, i32, i32]),
            "sdm_op_groupnorm": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, f32, i32, vp]),
            "sdm_op_layernorm": (i32, [vp, vp, i32, C.c_long, i32, vp, vp, f32, vp]),
            "sdm_op_attention": (i32, [vp, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, i32]),
            "sdm_op_attention_split": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
            "sdm_op_resize_aa": (i32, [vp, vp, i32, i32, i32, vp, i32, i32]),
            "sdm_op_mask_bias": (i32, [vp, vp, i32, i32, i32, vp]),
            "sdm_set_option": (i32, [C.c_char_p, i32]),
            "sdm_get_option": (i32, [C.c_char_p, C.POINTER(i32)]),
            "sdm_reset_options": (None, []),
            "sdm_option_name": (C.c_char_p, [i32]),
            "sdm_option_help": (C.c_char_p, [i32]),
            "sdm_kernel_counts": (i32, [C.c_char_p, i32]),
            "sdm_kernel_counts_reset": (None, []),
        }
        for name in EXPORTS:
            fn = getattr(cdll, name)          # AttributeError = missing export: fail loudly
            fn.restype, fn.argtypes = sig[name]
            setattr(self, name, fn)

    # ---- kernel-selection options (process-wide; tests and tools/ only - the library reads no environment variable) ----
    def set_option(self, name, value):
        if self.sdm_set_option(name.encode(), int(value)) != 0:
            raise KeyError(f"unknown engine option {name!r}; known: {', '.join(self.options())}")

    def get_option(self, name):
        v = C.c_int(0)
        if self.sdm_get_option(name.encode(), C.byref(v)) != 0:
            raise KeyError(name)
        return v.value

    def options(self):
        out, i = {}, 0
        while True:
            n = self.sdm_option_name(i)
            if n is None:
                return out
            out[n.decode()] = self.sdm_option_help(i).decode()
            i += 1

    def reset_options(self):
        self.sdm_reset_options()

    def kernel_counts(self, reset=False):
        """{kernel variant: launches since the last reset}"""
        n = self.sdm_kernel_counts(None, 0)
        buf = C.create_string_buffer(n + 1)
        self.sdm_kernel_counts(buf, n + 1)
        out = {}
        for item in buf.value.decode().split(";"):
            if item:
                k, v = item.rsplit("=", 1)
                out[k] = int(v)
        if reset:
            self.sdm_kernel_counts_reset()
        return out


_PRODUCT = None


def load_library() -> Bindings:
    """dlopen the gfx950 engine.  Raises if it has not been built (`python -m ... build` / __graft_entry__.build())."""
    global _PRODUCT
    if _PRODUCT is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"SDMatte HIP engine not built: {LIB_PATH} is missing (run __graft_entry__.build()); "
                               "there is no CPU fallback")
        _PRODUCT = Bindings(C.CDLL(LIB_PATH))
    return _PRODUCT


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    """One engine per GPU (one process per GPU under torch.distributed, or one per device inside ComfyUI)."""

    def __init__(self, cfg: SDMatteConfig = None, device: int = 0, stream_f32: bool = True, _lib: Bindings = None, precision=None):
        self.lib = _lib or load_library()
        self.cfg = cfg or SDMatteConfig.full()
        self.device = device
        self.precise_mask = precise_mask_of(precision)
        self._ccfg = to_c_config(self.cfg, stream_f32, self.precise_mask)
        h = C.c_void_p()
        rc = self.lib.sdm_create(C.byref(h), device, C.byref(self._ccfg))
        if rc != 0:
            raise RuntimeError(f"sdm_create failed ({rc}): {self.lib.sdm_last_error(None).decode()}")
        self.h = h
        self._on_device = _lib is None       # emulator builds (tests) address host memory

    def close(self):
        if getattr(self, "h", None):
            self.lib.sdm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.sdm_last_error(self.h).decode()}")
        return rc

    # ---- weights -------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = False):
        """Mirror of `load_state_dict(sd, strict=False)` (sdmatte_nodes.py:321): unknown keys are ignored,
        shape mismatches raise.  Returns (missing_keys, n_ignored)."""
        for k, t in state_dict.items():
            if not torch.is_tensor(t):
                continue
            t = t.detach()
            if t.device.type != "cpu":
                t = t.cpu()
            if t.dtype not in _DT:
                t = t.float()
            t = t.contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            self._check(self.lib.sdm_load_tensor(self.h, k.encode(), _DT[t.dtype], t.dim(), shape, _ptr(t)), f"load {k}")
        self._check(self.lib.sdm_finalize_weights(self.h), "finalize")
        nl, nm, ni = C.c_int64(), C.c_int64(), C.c_int64()
        self.lib.sdm_weight_stats(self.h, C.byref(nl), C.byref(nm), C.byref(ni))
        missing = [self.lib.sdm_missing_key(self.h, i).decode() for i in range(nm.value)]
        if strict and missing:
            raise RuntimeError(f"missing keys: {missing[:8]}...")
        return missing, ni.value

    def weight_blob_bytes(self):
        return int(self.lib.sdm_weight_blob_bytes(self.h))

    def export_weights(self, device_u8: torch.Tensor, host_u8: torch.Tensor):
        self._check(self.lib.sdm_export_weight_blob(self.h, _ptr(device_u8)), "export blob")
        self._check(self.lib.sdm_export_host_blob(self.h, _ptr(host_u8)), "export host blob")

    def import_weights(self, device_u8: torch.Tensor, host_u8: torch.Tensor):
        self._check(self.lib.sdm_import_weight_blob(self.h, _ptr(device_u8)), "import blob")
        self._check(self.lib.sdm_import_host_blob(self.h, _ptr(host_u8)), "import host blob")
        self._check(self.lib.sdm_finalize_weights(self.h), "finalize")

    def host_blob_bytes(self):
        return int(self.lib.sdm_host_blob_bytes(self.h))

    # ---- forward -------------------------------------------------------------------------------
    def _kind(self, t):
        return SDM_PTR_DEVICE if (t.device.type == "cuda") else (SDM_PTR_DEVICE if not self._on_device else SDM_PTR_HOST)

    def _check_io(self, what, *tensors):
        """All tensors of one call live in one place: host memory, or THIS engine's GPU (raw pointers cross the C ABI, so a
        tensor on another device would be read as garbage or fault).  Returns the hipStream_t the caller's work is queued on
        (torch's current stream of that device; the engine orders itself after it and makes it wait for the outputs)."""
        devs = {(t.device.type, t.device.index) for t in tensors if t is not None}
        if len(devs) != 1:
            raise ValueError(f"{what}: image, trimap and out must share one device, got {sorted(devs)}")
        (kind, idx), = devs
        if kind == "cuda":
            if not self._on_device or idx != self.device:
                raise ValueError(f"{what}: tensors live on cuda:{idx}, this engine drives cuda:{self.device}")
            return C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)
        if kind != "cpu":
            raise ValueError(f"{what}: unsupported device {kind}")
        return None

    def forward(self, image_b3ss: torch.Tensor, trimap_b1ss: torch.Tensor, is_trans=None, coords=None, out=None, sync=True,
                point_coords=None, use_attention_mask=True):
        """SDMatte.forward(data): image [B,3,S,S] in [-1,1], aux prompt image (trimap / bbox_mask / mask / point_mask) [B,1,S,S]
        in [-1,1] -> alpha [B,1,S,S].  `coords` [B,4] feed bbox_embedding (None -> [0,0,1,1]); `point_coords` [B,N] select the
        point prompt (point_embedding) instead; use_attention_mask=False drops the aux key mask of the self-attention."""
        B, _, SH, SW = image_b3ss.shape                    # square in the reference; rectangles are an extension (sdm_forward_rect)
        image_b3ss = image_b3ss.float().contiguous()
        trimap_b1ss = trimap_b1ss.float().contiguous()
        if out is None:
            out = torch.empty(B, 1, SH, SW, dtype=torch.float32, device=image_b3ss.device)
        elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != B * SH * SW:
            raise ValueError("forward: out must be a contiguous fp32 tensor of B*SH*SW elements")
        stream = self._check_io("forward", image_b3ss, trimap_b1ss, out)
        it = np.ascontiguousarray(np.zeros(B, np.int32) if is_trans is None else np.asarray(is_trans, np.int32).reshape(B))
        if point_coords is not None:
            co = np.ascontiguousarray(np.asarray(point_coords, np.float32).reshape(B, -1))
            kind, dim = 1, co.shape[1]
        else:
            co = None if coords is None else np.ascontiguousarray(np.asarray(coords, np.float32).reshape(B, 4))
            kind, dim = 0, 4
        self._check(self.lib.sdm_forward_rect(self.h, _ptr(image_b3ss), _ptr(trimap_b1ss), B, SH, SW, it.ctypes.data_as(C.c_void_p),
                                              co.ctypes.data_as(C.c_void_p) if co is not None else None, dim, kind,
                                              1 if use_attention_mask else 0, _ptr(out), self._kind(image_b3ss), stream), "sdm_forward_rect")
        if sync:
            self.synchronize()
        return out

    def apply_matte(self, image_bhwc: torch.Tensor, trimap_bhw: torch.Tensor, S: int, is_transparent=False, out=None, sync=True):
        """Device part of SDMatteApply.apply_matte: raw image [B,H,W,3] + trimap [B,H,W] in [0,1] -> alpha [B,H,W]."""
        B, H, W, Cc = image_bhwc.shape
        if Cc != 3:
            raise ValueError(f"apply_matte: image must be [B,H,W,3], got {tuple(image_bhwc.shape)}")
        image_bhwc = image_bhwc.float().contiguous()
        trimap_bhw = trimap_bhw.float().contiguous()
        if tuple(trimap_bhw.shape) != (B, H, W):
            raise ValueError(f"apply_matte: trimap must be [B,H,W] = {(B, H, W)}, got {tuple(trimap_bhw.shape)}")
        if out is None:
            out = torch.empty(B, H, W, dtype=torch.float32, device=image_bhwc.device)
        elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != B * H * W:
            raise ValueError("apply_matte: out must be a contiguous fp32 tensor of B*H*W elements")
        stream = self._check_io("apply_matte", image_bhwc, trimap_bhw, out)
        self._check(self.lib.sdm_apply_matte(self.h, _ptr(image_bhwc), _ptr(trimap_bhw), B, H, W, int(S), 1 if is_transparent else 0,
                                             _ptr(out), self._kind(image_bhwc), stream), "sdm_apply_matte")
        if sync:
            self.synchronize()
        return out

    OUTPUT_MODES = {"alpha_only": 0, "matted_rgba": 1, "matted_rgb": 2}

    def apply_matte_node(self, image_bhwc, trimap_bhw, S, is_transparent, output_mode, mask_refine, trimap_constraint, sync=True):
        """The whole node body in one C-ABI call: preprocess, model, resize back, mask_refine and output composition, all on
        the GPU.  Returns (alpha [B,H,W], matted [B,H,W,3|4]) on the inputs' device."""
        if output_mode not in self.OUTPUT_MODES:
            raise ValueError(f"unknown output_mode {output_mode!r}")
        B, H, W, Cc = image_bhwc.shape
        if Cc != 3:
            raise ValueError(f"apply_matte_node: image must be [B,H,W,3], got {tuple(image_bhwc.shape)}")
        image_bhwc = image_bhwc.float().contiguous()
        trimap_bhw = trimap_bhw.float().contiguous()
        if trimap_bhw.dim() != 3 or trimap_bhw.shape[0] != B:
            raise ValueError(f"apply_matte_node: trimap must be [B,h,w] with B = {B}, got {tuple(trimap_bhw.shape)}")
        TH, TW = int(trimap_bhw.shape[1]), int(trimap_bhw.shape[2])
        mode = self.OUTPUT_MODES[output_mode]
        if (TH, TW) != (H, W) and (mask_refine or mode == 2):
            # where the reference fails too: it indexes the (H, W) alpha with the trimap (sdmatte_nodes.py:365-380,390-394)
            raise IndexError(f"apply_matte_node: the trimap {(TH, TW)} must match the image {(H, W)} for mask_refine / matted_rgb")
        alpha = torch.empty(B, H, W, dtype=torch.float32, device=image_bhwc.device)
        matted = torch.empty(B, H, W, 4 if mode == 1 else 3, dtype=torch.float32, device=image_bhwc.device)
        stream = self._check_io("apply_matte_node", image_bhwc, trimap_bhw, alpha, matted)
        self._check(self.lib.sdm_apply_matte_node(self.h, _ptr(image_bhwc), _ptr(trimap_bhw), B, H, W, TH, TW, int(S), 1 if is_transparent else 0, mode,
                                                  1 if mask_refine else 0, float(trimap_constraint), _ptr(alpha), _ptr(matted),
                                                  self._kind(image_bhwc), stream), "sdm_apply_matte_node")
        if sync:
            self.synchronize()
        return alpha, matted

    def synchronize(self):
        self._check(self.lib.sdm_synchronize(self.h), "sdm_synchronize")

    def last_forward_ms(self):
        return float(self.lib.sdm_last_forward_ms(self.h))

    def resident_bytes(self):
        """Device memory held by the engine (weights + activation arena + I/O staging), invisible to torch's allocator."""
        return int(self.lib.sdm_resident_bytes(self.h))

    def weight_bytes(self):
        """The weight part of resident_bytes(): canonical blob + derived kernel layouts (stays resident across release_memory())."""
        return int(self.lib.sdm_weight_bytes(self.h))

    def release_memory(self):
        """Free the activation arena and staging buffers (weights stay); the next call re-allocates what it needs."""
        self._check(self.lib.sdm_release_memory(self.h), "sdm_release_memory")

    def profile(self, on: bool):
        self.lib.sdm_profile_enable(self.h, 1 if on else 0)

    def profile_results(self):
        res = {}
        for i in range(self.lib.sdm_profile_count(self.h)):
            name, ms, n, fl, by = C.c_char_p(), C.c_float(), C.c_int64(), C.c_double(), C.c_double()
            self.lib.sdm_profile_get(self.h, i, C.byref(name), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by))
            res[name.value.decode()] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        return res

    def profile_dump(self):
        return self.lib.sdm_profile_dump(self.h).decode()

    # ---- single operators (parity tests) ---------------------------------------------------------
    def op_conv(self, x0, w, bias=None, x1=None, stride=1, pad_mode=0, up=0, res=None, geglu=False, out_f32=False, out_scale=1.0,
                tile_cfg=-1, split=False, gn=None, cmask=None):
        """x0/x1: NHWC fp16|fp32 tensors; w: fp32 OIHW or [O,I]; returns NHWC.  split=True: split-fp16 operands (fp32 inputs);
        gn=(gamma, beta, eps, groups, silu): GroupNorm(+SiLU) of the input fused into the conv's operand staging.
        cmask (test hook): uint8 [N,H,W] class plane of x0 (0 = nothing known, 1..4 = pixels of one constant region class)."""
        N, H, W_, C0 = x0.shape
        C1 = x1.shape[-1] if x1 is not None else 0
        ntaps = 9 if (w.dim() == 4 and w.shape[-1] == 3) else 1
        O = w.shape[0]
        Ho, Wo = (H << up), (W_ << up)
        if stride == 2:
            Ho, Wo = Ho // 2, Wo // 2
        Creal = O // 2 if geglu else O
        Cst = (Creal + 3) // 4 * 4          # rows are stored with 4-channel vector stores
        out = torch.empty(N, Ho, Wo, Cst, dtype=torch.float32 if out_f32 else torch.float16, device=x0.device)
        w = w.float().contiguous()
        b = bias.float().contiguous() if bias is not None else None
        gam = gn[0].float().contiguous() if gn is not None else None
        bet = gn[1].float().contiguous() if gn is not None else None
        if cmask is not None:
            cmask = cmask.to(torch.uint8).contiguous()
            self._check(self.lib.sdm_debug_set_input_cmask(self.h, _ptr(cmask)), "sdm_debug_set_input_cmask")
        self._check(self.lib.sdm_op_conv_ex(self.h, _ptr(x0), _ptr(x1), C0, C1, int(x0.dtype == torch.float32), N, H, W_, up, stride,
                                            pad_mode, ntaps, _ptr(w), _ptr(b), O, _ptr(out), int(out_f32), _ptr(res),
                                            int(res is not None and res.dtype == torch.float32), int(geglu), float(out_scale), tile_cfg,
                                            int(split), _ptr(gam), _ptr(bet), float(gn[2]) if gn is not None else 0.0,
                                            int(gn[3]) if gn is not None else 32, int(gn[4]) if gn is not None else 0),
                    "sdm_op_conv_ex")
        return out[..., :Creal]

    def op_gemm_p3(self, x, w, bias=None, mode=0, res=None, ln=None, lo_cols=-1):
        """Test hook: the plane-fed GEMM (k_gemm.h) on x fp32 [N,H,W,K].  mode 0 fp32 (+res), 1 GEGLU, 3 planes (+res), 4 fp32 + statistics -> (out, stats[N,srows,O,2]),
        2 -> (hi fp16 [N,H,W,O], pair uint8 [N,H,W,O,2]); ln = (gamma, beta, eps): LayerNorm with plane output in front."""
        N, H, W_, K = x.shape
        O = w.shape[0]
        x = x.float().contiguous(); w = w.float().contiguous()
        b = bias.float().contiguous() if bias is not None else None
        r = res.float().contiguous() if res is not None else None
        gam = ln[0].float().contiguous() if ln is not None else None
        bet = ln[1].float().contiguous() if ln is not None else None
        Cst = O // 2 if mode == 1 else O
        if mode == 2:
            out = torch.zeros(2 * N * H * W_ * O, dtype=torch.float16, device=x.device)
        else:
            out = torch.empty(N, H, W_, Cst, dtype=torch.float32, device=x.device)
        srows_max = 2 * ((H * W_ + 63) // 64)
        stats = torch.zeros(N * srows_max * O * 2, dtype=torch.float32, device=x.device) if mode == 4 else None
        srows = C.c_int(0)
        self._check(self.lib.sdm_op_gemm_p3(self.h, _ptr(x), N, H, W_, K, _ptr(w), _ptr(b), O, mode, _ptr(r), _ptr(gam), _ptr(bet),
                                            float(ln[2]) if ln is not None else 0.0, lo_cols, _ptr(out), _ptr(stats), C.byref(srows)), "sdm_op_gemm_p3")
        if mode == 2:
            n = N * H * W_ * O
            return out[:n].view(N, H, W_, O), out[n:].view(torch.uint8).view(N, H, W_, O, 2)
        if mode == 4:
            return out, stats[:N * srows.value * O * 2].view(N, srows.value, O, 2)
        return out

    def debug_run_layer(self, name, x_nhwc, cout):
        """Test hook: one packed layer of the loaded model on an fp32 NHWC input -> fp32 NHWC [N,H,W,cout]."""
        N, H, W_, _ = x_nhwc.shape
        x = x_nhwc.float().contiguous()
        out = torch.empty(N, H, W_, cout, dtype=torch.float32, device=x.device)
        self._check(self.lib.sdm_debug_run_layer(self.h, name.encode(), _ptr(x), N, H, W_, _ptr(out), cout), "sdm_debug_run_layer")
        return out

    def debug_temb_row(self, index, is_trans, coords, cout):
        """Test hook: folded conv1 bias row of the index-th time-embedded ResBlock for one (is_trans, box) conditioning."""
        out = np.zeros(cout, np.float32)
        co = None if coords is None else np.ascontiguousarray(np.asarray(coords, np.float32).reshape(4))
        self._check(self.lib.sdm_debug_temb_row(self.h, index, int(is_trans), co.ctypes.data_as(C.c_void_p) if co is not None else None,
                                                out.ctypes.data_as(C.c_void_p), cout), "sdm_debug_temb_row")
        return torch.from_numpy(out)

    def bench_conv(self, N, H, W, Cin, Cout, ntaps=9, stride=1, in_f32=0, tile_cfg=-1, ablate=0, iters=10):
        return float(self.lib.sdm_bench_conv(self.h, N, H, W, Cin, Cout, ntaps, stride, in_f32, tile_cfg, ablate, iters))

    def bench_gemm_p3(self, M, K, O, epi=0, res=False, iters=10):
        return float(self.lib.sdm_bench_gemm_p3(self.h, M, K, O, epi | (256 if res else 0), iters))

    def bench_attn(self, B, heads, Lq, Lk, qt=1, ablate=0, iters=10):
        return float(self.lib.sdm_bench_attn(self.h, B, heads, Lq, Lk, qt, ablate, iters))

    def op_groupnorm(self, x0, gamma, beta, eps, silu, groups=32, x1=None):
        N, H, W_, C0 = x0.shape
        C1 = x1.shape[-1] if x1 is not None else 0
        out = torch.empty(N, H, W_, C0 + C1, dtype=torch.float16, device=x0.device)
        self._check(self.lib.sdm_op_groupnorm(self.h, _ptr(x0), _ptr(x1), C0, C1, int(x0.dtype == torch.float32), N, H * W_, groups,
                                              _ptr(gamma), _ptr(beta), float(eps), int(silu), _ptr(out)), "sdm_op_groupnorm")
        return out

    def op_layernorm(self, x, gamma, beta, eps):
        rows, Cc = x.numel() // x.shape[-1], x.shape[-1]
        out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
        self._check(self.lib.sdm_op_layernorm(self.h, _ptr(x), int(x.dtype == torch.float32), rows, Cc, _ptr(gamma), _ptr(beta),
                                              float(eps), _ptr(out)), "sdm_op_layernorm")
        return out

    def op_attention(self, q, k, v, heads, bias=None):
        """q [B,Lq,h*D], k/v [B,Lk,h*D] fp16 (may be views with a row stride); bias fp32 [B,Lk] or None."""
        B, Lq, HD = q.shape
        Lk = k.shape[1]
        D = HD // heads
        out = torch.empty(B, Lq, HD, dtype=torch.float16, device=q.device)
        self._check(self.lib.sdm_op_attention(self.h, _ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v), v.stride(1), _ptr(bias), B,
                                              heads, Lq, Lk, D, _ptr(out), HD), "sdm_op_attention")
        return out

    def op_attention_split(self, q, k, v, heads, bias=None):
        """Split-precision attention cores (head dim 64): q [B,Lq,h*64], k / v [B,Lk,h*64] fp32; the C side splits them into the operand
        planes the producing GEMM epilogues write in the engine (fp16 high parts + fp8 residual pairs for Q.K^T); fp32 output."""
        B, Lq, HD = q.shape
        Lk = k.shape[1]
        qf, kf, vf = q.float().contiguous(), k.float().contiguous(), v.float().contiguous()
        out = torch.empty(B, Lq, HD, dtype=torch.float32, device=q.device)
        self._check(self.lib.sdm_op_attention_split(self.h, _ptr(qf), _ptr(kf), _ptr(vf), _ptr(bias), B, heads, Lq, Lk, _ptr(out)), "sdm_op_attention_split")
        return out

    def op_resize_aa(self, planes, Hout, Wout):
        P, Hin, Win = planes.shape
        out = torch.empty(P, Hout, Wout, dtype=torch.float32, device=planes.device)
        self._check(self.lib.sdm_op_resize_aa(self.h, _ptr(planes), P, Hin, Win, _ptr(out), Hout, Wout), "sdm_op_resize_aa")
        return out

    def op_mask_bias(self, plane_bss, level):
        B, S, _ = plane_bss.shape
        lk = (S // 8) >> level
        out = torch.empty(B, lk * lk, dtype=torch.float32, device=plane_bss.device)
        self._check(self.lib.sdm_op_mask_bias(self.h, _ptr(plane_bss), B, S, level, _ptr(out)), "sdm_op_mask_bias")
        return out
