"""Key schema / config sanity: the schema walk must reproduce the published SD-2.1 parameter counts
(U-Net 865.91 M + 7.12 M SDMatte extras = 873.03 M; VAE 83.65 M) - SURVEY.md 2.2."""
import torch


def test_param_counts(pkg):
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import count_params, unet_up_resnet_in_channels
    cfg = SDMatteConfig.full()
    assert count_params(cfg, "unet.") == 873_030_852
    assert count_params(cfg, "vae.") == 83_653_863
    assert unet_up_resnet_in_channels(cfg) == [(2560, 2560, 2560), (2560, 2560, 1920), (1920, 1280, 960), (960, 640, 640)]


def test_oracle_runs_on_tiny(pkg):
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    from oracle import sdmatte_oracle as O
    cfg = SDMatteConfig.tiny()
    w = synthetic_state_dict(cfg, 0)
    img, tri = synthetic_inputs(2, 64, 64)
    a, m = O.apply_matte(w, cfg.as_dict(), img, tri, 64)
    assert a.shape == (2, 64, 64) and m.shape == (2, 64, 64, 3)
    assert 0.0 <= float(a.min()) and float(a.max()) <= 1.0 and float(a.std()) > 0.05
    # batch invariance: image 1 alone gives the same alpha (no cross-image coupling)
    a1, _ = O.apply_matte(w, cfg.as_dict(), img[1:], tri[1:], 64)
    assert torch.allclose(a1[0], a[1], atol=1e-5)
    # all-foreground trimap == unmasked attention (bias is identically 0)
    d = O.preprocess(img[:1], torch.ones(1, 64, 64), 64, False)
    p1 = O.sdmatte_forward(w, cfg.as_dict(), d)
    assert torch.isfinite(p1).all()
