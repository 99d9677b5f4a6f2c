"""ComfyUI node `Apply SDMatte` + model-load API, MI355X-native.

Drop-in for /root/reference/sdmatte_nodes.py: same `SDMatteApply` class attributes and `apply_matte`
signature (sdmatte_nodes.py:217-257), same `NODE_CLASS_MAPPINGS` / `NODE_DISPLAY_NAME_MAPPINGS`
(:408-414), same module-level `MODEL_DIR`, `MODEL_URLS`, `download_model`, `ensure_sd21_from_manojb`
(:9-17,34,103).  What changes underneath:
  * the model is the hand-written HIP engine (engine.py -> libsdmatte_hip.so), not diffusers modules;
  * the built model is cached per (checkpoint path, mtime, device) instead of being rebuilt and re-read on every
    call (the reference does both per call, :286-323);
  * resize / normalise / forward / resize-back / clamp / `mask_refine` / output composition (:339-397) run on the GPU in
    one C-ABI call (`sdm_apply_matte_node`); `refine_and_compose` below is the same tail on CPU tensors, kept as the
    bit-exact restatement the tests compare with;
  * `force_cpu=True` is rejected: this node has no CPU path (the reference's own force_cpu branch cannot run either:
    meta_arch.py hard-codes `.cuda()`).
"""
import os
import threading

import torch

try:  # inside ComfyUI
    import folder_paths
except Exception:  # headless use (bench / tests): minimal stand-in with the three functions the node needs
    class _FolderPaths:
        def __init__(self):
            self.models_dir = os.environ.get("SDMATTE_MODELS_DIR", os.path.join(os.path.expanduser("~"), ".cache", "sdmatte_models"))
            self._paths = {}

        def add_model_folder_path(self, name, path):
            self._paths.setdefault(name, [])
            if path not in self._paths[name]:
                self._paths[name].append(path)

        def get_folder_paths(self, name):
            return list(self._paths.get(name, []))

    folder_paths = _FolderPaths()

try:
    import comfy.model_management as _comfy_mm
except Exception:
    _comfy_mm = None

MODEL_DIR = os.path.join(folder_paths.models_dir, "SDMatte")
folder_paths.add_model_folder_path("SDMatte", MODEL_DIR)

MODEL_URLS = {
    "SDMatte.safetensors": "https://huggingface.co/1038lab/SDMatte/resolve/main/SDMatte.safetensors",
    "SDMatte_plus.safetensors": "https://huggingface.co/1038lab/SDMatte/resolve/main/SDMatte_plus.safetensors",
}

# The reference needs these SD-2.1 config files to instantiate diffusers modules; the native engine embeds the constants
# (config.py) and never reads them.  The helper is kept so that callers of the reference API keep working.
SD21_MANOJB_FILES = {p: p for p in (
    "model_index.json", "text_encoder/config.json", "vae/config.json", "unet/config.json", "scheduler/scheduler_config.json",
    "tokenizer/tokenizer_config.json", "tokenizer/merges.txt", "tokenizer/vocab.json", "tokenizer/special_tokens_map.json",
    "feature_extractor/preprocessor_config.json")}


def _fetch(url, target):
    """Stream `url` to `target` through a .tmp file + atomic rename; verifies content-length when known."""
    os.makedirs(os.path.dirname(target), exist_ok=True)
    tmp = target + ".tmp"
    try:
        try:
            import requests
        except ImportError:
            import urllib.request
            urllib.request.urlretrieve(url, tmp)
        else:
            with requests.get(url, stream=True, timeout=60) as resp:
                resp.raise_for_status()
                expected = int(resp.headers.get("content-length", 0) or 0)
                with open(tmp, "wb") as fh:
                    for block in resp.iter_content(1 << 20):
                        if block:
                            fh.write(block)
                if expected and os.path.getsize(tmp) != expected:
                    raise IOError(f"[SDMatte] Incomplete download: {os.path.getsize(tmp)} != {expected}")
        if os.path.isfile(target) and os.path.getsize(target) > 0:   # someone else finished first
            os.remove(tmp)
        else:
            os.replace(tmp, target)
    except BaseException:
        if os.path.exists(tmp):
            try:
                os.remove(tmp)
            except OSError:
                pass
        raise
    return target


def ensure_sd21_from_manojb(sd21_base_dir=None):
    """API-compatible with the reference (sdmatte_nodes.py:34-101): make sure the SD-2.1 config files exist.
    Best effort (failures are printed, not raised); the native engine does not depend on the result."""
    if sd21_base_dir is None:
        roots = folder_paths.get_folder_paths("diffusers") or [os.path.join(folder_paths.models_dir, "diffusers")]
        sd21_base_dir = os.path.join(roots[0], "stable-diffusion-2-1-base")
    os.makedirs(sd21_base_dir, exist_ok=True)
    base = "https://huggingface.co/Manojb/stable-diffusion-2-1-base/resolve/main"
    for rel in SD21_MANOJB_FILES:
        dst = os.path.join(sd21_base_dir, rel)
        if os.path.isfile(dst):
            continue
        try:
            _fetch(f"{base}/{rel}", dst)
            print(f"[SDMatte] Downloaded {rel}")
        except Exception as exc:  # noqa: BLE001 - mirror the reference: warn and continue
            print(f"[SDMatte] Warning: failed to download {rel}: {exc}")
    return sd21_base_dir


def download_model(model_name, models_dir=MODEL_DIR, model_urls=MODEL_URLS):
    """Locate (every registered "SDMatte" folder first) or download a checkpoint; ValueError on unknown names
    (reference: sdmatte_nodes.py:103-199)."""
    for root in folder_paths.get_folder_paths("SDMatte") or []:
        cand = os.path.join(root, model_name)
        try:
            if os.path.isfile(cand) and os.path.getsize(cand) > 0:
                print(f"[SDMatte] Found model at: {cand}")
                return cand
        except OSError:
            continue
    url = model_urls.get(model_name)
    if not url:
        raise ValueError(f"[SDMatte] Unknown model name: {model_name}")
    target = os.path.join(models_dir, model_name)
    if os.path.isfile(target) and os.path.getsize(target) > 0:
        return target
    print(f"[SDMatte] Model '{model_name}' not found. Downloading to {target}...")
    _fetch(url, target)
    print(f"[SDMatte] Download complete: {target}")
    return target


SDMatteCore = None            # lazily bound, like the reference's module global (sdmatte_nodes.py:201,262-264)
_MODEL_CACHE = {}
_CACHE_LOCK = threading.Lock()


def _torch_device():
    if _comfy_mm is not None:
        return _comfy_mm.get_torch_device()
    if not torch.cuda.is_available():
        raise RuntimeError("[SDMatte] no ROCm GPU visible: the MI355X-native node has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


class LazyCheckpoint:
    """`state_dict`-like view of a .safetensors file that materialises ONE tensor at a time from the memory map (the engine packs
    every tensor into its own arena as it arrives; a dict of all tensors would hold the whole 3.8 GB checkpoint on the host).
    Only `text_encoder.*` is skipped up front: dead on this path (meta_arch.py:220-234 is never consumed, replace.py:414-416)."""

    def __init__(self, path):
        self.path = path

    def keys(self):
        from safetensors import safe_open
        with safe_open(self.path, framework="pt", device="cpu") as f:
            return [k for k in f.keys()]

    def items(self):
        from safetensors import safe_open
        with safe_open(self.path, framework="pt", device="cpu") as f:
            for key in f.keys():
                if key.startswith("text_encoder."):
                    continue
                yield key, f.get_tensor(key)


def load_checkpoint_state_dict(path):
    return LazyCheckpoint(path)


def get_model(ckpt_name, device):
    """Build (once) and cache the engine for (checkpoint file, mtime, device)."""
    global SDMatteCore
    if SDMatteCore is None:
        from .core import SDMatte as SDMatteCore
    path = download_model(ckpt_name)
    key = (os.path.realpath(path), os.path.getmtime(path), str(device))
    with _CACHE_LOCK:
        model = _MODEL_CACHE.get(key)
        if model is None:
            model = SDMatteCore(
                pretrained_model_name_or_path=None, load_weight=False, use_aux_input=True, aux_input="trimap",
                aux_input_list=["point_mask", "bbox_mask", "mask", "trimap"],
                attn_mask_aux_input=["point_mask", "bbox_mask", "mask", "trimap"],
                use_encoder_hidden_states=True, use_attention_mask=True, add_noise=False)
            model.load_state_dict(load_checkpoint_state_dict(path), strict=False)
            model.eval()
            model.to(device)
            _MODEL_CACHE.clear()          # one resident checkpoint per process is enough for the node
            _MODEL_CACHE[key] = model
    return model


def _trim_engine_memory(model):
    """The engine's activation arena lives outside torch's / ComfyUI's allocators and is sized by the largest call so far.  Give it
    back after the call when it is large (SDMATTE_KEEP_ARENA_GB, default 8), so that other nodes of the workflow can use the memory;
    the packed weights stay resident (that is the point of the model cache)."""
    try:
        limit = float(os.environ.get("SDMATTE_KEEP_ARENA_GB", "8")) * 2 ** 30
        engines = [model.engine] + (list(model._fan.engines[1:]) if getattr(model, "_fan", None) is not None else [])
        for eng in engines:
            if eng.resident_bytes() - eng.weight_bytes() > limit:      # arena + I/O staging only: ALL weight layouts stay
                eng.release_memory()
    except Exception as exc:  # noqa: BLE001 - best effort, like the reference's empty_cache block (sdmatte_nodes.py:399-403)
        print(f"[SDMatte] note: could not trim engine memory ({exc})")


def unload_models():
    """Drop the cached engine(s): every byte the node holds on the GPU is released."""
    with _CACHE_LOCK:
        for model in list(_MODEL_CACHE.values()):
            fan = getattr(model, "_fan", None)
            if fan is not None:
                fan.close()
            if model.engine is not None:
                model.engine.close()
        _MODEL_CACHE.clear()


def _fan_out(model, batch):
    """Multi-GPU fan-out of one node call.  OPT-IN (SDMATTE_MULTI_GPU=1): every extra GPU receives its own copy of the packed weights
    (~12 GB in the default precision) plus an activation arena, outside ComfyUI's memory manager, and those GPUs may belong to
    other models or processes.  Used when the batch has more than one image and more than one GPU is visible; the extra engines
    live as long as the cached model they were copied from (`unload_models()` frees them)."""
    if batch < 2 or os.environ.get("SDMATTE_MULTI_GPU", "0") != "1" or not torch.cuda.is_available():
        return None
    ndev = torch.cuda.device_count()
    if ndev < 2:
        return None
    fan = getattr(model, "_fan", None)
    if fan is None:
        from .parallel import MultiGpuEngine
        first = model.engine
        fan = MultiGpuEngine.around(first, [d for d in range(ndev) if d != first.device])
        model._fan = fan
    return fan


def refine_and_compose(alpha_bhw, image, trimap, output_mode, mask_refine, trimap_constraint):
    """CPU tail of the node, same arithmetic and order as sdmatte_nodes.py:365-397."""
    out = alpha_bhw
    image_cpu, tri = image.cpu(), trimap.cpu()
    if mask_refine:
        fg = tri > trimap_constraint
        bg = tri < (1.0 - trimap_constraint)
        unknown = ~(fg | bg)
        ref = out.clone()
        ref[bg] = 0.0
        ref[fg] = torch.clamp(ref[fg] * 1.2, 0, 1)
        ref[(ref < 0.3) & unknown] = 0.0
        out = ref
    a4 = out.unsqueeze(-1)
    if output_mode == "alpha_only":
        matted = torch.zeros_like(image_cpu)
    elif output_mode == "matted_rgba":
        matted = torch.cat([image_cpu, a4], dim=-1)
    elif output_mode == "matted_rgb":
        matted = image_cpu * ((tri.unsqueeze(-1) > 0.2) & (a4 > 0.1)).float()
    else:
        matted = image_cpu * a4
    return out, matted


class SDMatteApply:

    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "ckpt_name": (list(MODEL_URLS.keys()), ),
                "image": ("IMAGE", {"tooltip": "image to matte"}),
                "trimap": ("MASK", {"tooltip": "trimap: white = foreground, black = background, gray = unknown"}),
                "inference_size": ([512, 640, 768, 896, 1024], {"default": 1024, "tooltip": "inference resolution: higher = better and slower (1024 best quality, 768 balanced)"}),
                "is_transparent": ("BOOLEAN", {"default": False, "tooltip": "enable when the source image has a transparent background / depicts a transparent object"}),
                "output_mode": (["alpha_only", "matted_rgba", "matted_rgb"], {"default": "alpha_only", "tooltip": "alpha_only = mask only; matted_rgba = cut-out on a transparent background; matted_rgb = cut-out on black"}),
                "mask_refine": ("BOOLEAN", {"default": True, "tooltip": "refine the mask with the trimap: filters unwanted regions, less background interference"}),
                "trimap_constraint": ("FLOAT", {"default": 0.8, "min": 0.1, "max": 1.0, "step": 0.1, "tooltip": "strength of the trimap constraint (0.1-1.0): higher = stricter; 0.8 balanced, 0.9 strict, 0.6 permissive"}),
            },
            "optional": {
                "force_cpu": ("BOOLEAN", {"default": False}),
            },
        }

    RETURN_TYPES = ("MASK", "IMAGE")
    RETURN_NAMES = ("alpha_mask", "matted_image")
    FUNCTION = "apply_matte"
    CATEGORY = "Matting/SDMatte"

    def apply_matte(self, ckpt_name, image, trimap, inference_size, is_transparent, output_mode, mask_refine, trimap_constraint,
                    force_cpu=False):
        if force_cpu:
            raise RuntimeError("[SDMatte] force_cpu=True is not available: this node runs hand-written gfx950 kernels only "
                               "(no CPU path).  Use the reference plugin for CPU inference.")
        if image.dim() != 4 or image.shape[-1] != 3:
            raise ValueError(f"[SDMatte] image must be [B,H,W,3], got {tuple(image.shape)}")
        # the trimap is resized to the inference size on its own, as in the reference (sdmatte_nodes.py:212-214,349); its size only has
        # to equal the image's where the reference indexes the alpha with it (mask_refine, matted_rgb): the engine raises there
        if trimap.dim() != 3 or trimap.shape[0] != image.shape[0]:
            raise ValueError(f"[SDMatte] trimap must be [B,h,w] with the image's batch size, got {tuple(trimap.shape)}")
        model = get_model(ckpt_name, _torch_device())
        fan = _fan_out(model, image.shape[0])
        # one C-ABI call per device: resize / normalise / model / resize back / clamp AND mask_refine + output composition, all on
        # the GPU at the original resolution (bit-identical to the reference's CPU tail, tests/test_emu_e2e.py); with several GPUs
        # the batch is split over them (one engine + host thread per device)
        runner = fan if fan is not None else model.engine
        out, matted = runner.apply_matte_node(image, trimap, int(inference_size), bool(is_transparent), output_mode, bool(mask_refine),
                                              float(trimap_constraint))
        out, matted = out.detach().cpu(), matted.detach().cpu()
        _trim_engine_memory(model)
        return (out, matted)


NODE_CLASS_MAPPINGS = {
    "SDMatteApply": SDMatteApply,
}

NODE_DISPLAY_NAME_MAPPINGS = {
    "SDMatteApply": "Apply SDMatte",
}
