"""Compare a real `SDMatte*.safetensors` with the key schema the engine expects (comfyui-sdmatte_amd/weights.py::weight_schema,
inferred from the reference's module attribute names: SURVEY.md Appendix C - no real checkpoint was ever available to the build).
Reads only the safetensors header (names, dtypes, shapes): no tensor data is loaded, no GPU is needed.

  python tools/check_checkpoint.py /path/to/SDMatte.safetensors [--write-manifest out.json] [--config full|tiny]
  python tools/check_checkpoint.py --write-expected expected.json [--config full|tiny]      # what the engine expects, as a manifest
  python tools/check_checkpoint.py --diff real_manifest.json [--config full|tiny]            # a manifest written elsewhere vs the schema

A manifest is {tensor name: {"dtype", "shape"}}: `--write-manifest` on the machine that has the real file + `--diff` here (or
`--write-expected` here + any JSON diff there) answers "will it load" in one command, without moving 4 GB.

Exit code 0 when every tensor the engine consumes is present with the expected shape (extra tensors such as text_encoder.* are
listed but fine: the engine ignores them, like the reference's load_state_dict(strict=False))."""
import json
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

LEGACY = ((".query.", ".to_q."), (".key.", ".to_k."), (".value.", ".to_v."), (".proj_attn.", ".to_out.0."))


def read_header(path):
    with open(path, "rb") as fh:
        n = struct.unpack("<Q", fh.read(8))[0]
        hdr = json.loads(fh.read(n))
    hdr.pop("__metadata__", None)
    return {k: (v["dtype"], tuple(v["shape"])) for k, v in hdr.items()}


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    load_package()
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import weight_schema
    if "--write-expected" in sys.argv:
        out = sys.argv[sys.argv.index("--write-expected") + 1]
        cfg_name = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "full"
        want = weight_schema(getattr(SDMatteConfig, cfg_name)())
        with open(out, "w") as fh:
            json.dump({"_comment": "tensors the SDMatte engine consumes (any float dtype; legacy VAE attention names query/key/value/proj_attn are "
                                   "accepted for to_q/to_k/to_v/to_out.0; everything else in a checkpoint, e.g. text_encoder.*, is ignored)",
                       "tensors": {k: {"dtype": "F32|F16|BF16", "shape": list(v)} for k, v in want.items()}}, fh, indent=0)
        print(f"expected manifest of {len(want)} tensors ({cfg_name} architecture) written to {out}")
        return
    if "--diff" in sys.argv:
        with open(sys.argv[sys.argv.index("--diff") + 1]) as fh:
            man = json.load(fh)
        man = man.get("tensors", man)
        have = {k: (v["dtype"], tuple(v["shape"])) for k, v in man.items() if isinstance(v, dict)}
    else:
        have = read_header(sys.argv[1])
    if "--write-manifest" in sys.argv:
        out = sys.argv[sys.argv.index("--write-manifest") + 1]
        with open(out, "w") as fh:
            json.dump({k: {"dtype": d, "shape": list(s)} for k, (d, s) in sorted(have.items())}, fh, indent=0)
        print(f"manifest of {len(have)} tensors written to {out}")
    cfg_name = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "full"
    want = weight_schema(getattr(SDMatteConfig, cfg_name)())
    norm = {}
    for k, v in have.items():
        kk = k
        if "mid_block.attentions.0" in kk:
            for a, b in LEGACY:
                kk = kk.replace(a, b)
        norm[kk] = v
    missing = [k for k in want if k not in norm]
    numel = lambda s: int(__import__("math").prod(s)) if len(s) else 1
    bad = [(k, norm[k][1], tuple(want[k])) for k in want if k in norm and numel(norm[k][1]) != numel(tuple(want[k]))]
    extra = [k for k in norm if k not in want]
    groups = {}
    for k in extra:
        groups[k.split(".")[0]] = groups.get(k.split(".")[0], 0) + 1
    print(f"checkpoint: {len(have)} tensors; engine schema: {len(want)} tensors")
    print(f"missing (engine would refuse to load): {len(missing)}")
    for k in missing[:40]:
        print("   -", k, tuple(want[k]))
    print(f"shape mismatches: {len(bad)}")
    for k, a, b in bad[:40]:
        print("   -", k, "checkpoint", a, "expected", b)
    print(f"ignored by the engine: {len(extra)} {groups}")
    dts = {}
    for d, _ in have.values():
        dts[d] = dts.get(d, 0) + 1
    print(f"dtypes: {dts}")
    sys.exit(1 if (missing or bad) else 0)


if __name__ == "__main__":
    main()
