"""Import shims for running the *real* reference (``/root/reference``) in the
build container.  TEST INFRASTRUCTURE ONLY (see clip_oracle.py header).

The reference checkout does not exist on the GPU box: everything here is
guarded by :func:`reference_available` and is used only by
``tools/make_golden.py`` and the ``-m "not gpu"`` tests that pin the oracle.

Shims: (1) ``easynlp.appzoo`` is pre-registered as a bare
package so its eager ``__init__`` (which imports every app and needs
uninstalled packages) does not run; (2) ``datasets.list_datasets`` stub
(removed upstream, imported at appzoo/dataset.py:28); (3) stub ``tensorboardX``
/ ``rouge`` / ``ftfy`` modules.
"""
from __future__ import annotations

import json
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("EASYNLP_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "easynlp", "appzoo", "clip"))


_installed = False


def install_shims() -> None:
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference checkout not present at %s" % REFERENCE_ROOT)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import easynlp  # noqa: F401  (package root: light)
    pkg = types.ModuleType("easynlp.appzoo")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "easynlp", "appzoo")]
    sys.modules["easynlp.appzoo"] = pkg
    try:
        import datasets
        if not hasattr(datasets, "list_datasets"):
            datasets.list_datasets = lambda *a, **k: []
    except Exception:  # pragma: no cover
        pass
    for name in ("tensorboardX", "rouge", "ftfy", "jieba"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                if name == "tensorboardX":
                    class SummaryWriter:  # noqa: D401
                        def __init__(self, *a, **k):
                            pass

                        def __getattr__(self, _):
                            return lambda *a, **k: None
                    m.SummaryWriter = SummaryWriter
                if name == "rouge":
                    m.Rouge = object
                if name == "ftfy":
                    m.fix_text = lambda s: s
                sys.modules[name] = m
    import numpy as _np
    if not hasattr(_np, "long"):        # removed in numpy 2; text2video_retrieval/data.py:253 still spells the mask dtype so
        _np.long = _np.int64
    _installed = True


def reference_chinese_clip(cfg: dict, state_dict: dict):
    """Instantiate the reference ``CHINESE_CLIP`` with given weights (eval mode)."""
    install_shims()
    from easynlp.modelzoo.models.clip.modeling_chineseclip import CHINESE_CLIP
    m = CHINESE_CLIP(**cfg)
    missing, unexpected = m.load_state_dict(state_dict, strict=False)
    # only the non-parameter buffer may be missing
    assert all("position_ids" in k for k in missing), missing
    assert not unexpected, unexpected
    return m.eval()


def write_checkpoint_dir(path: str, cfg: dict, state_dict: dict, vocab_size: int = 0) -> None:
    """Synthetic checkpoint in the reference's on-disk format
    (appzoo/clip/model.py:52-72): config.json + pytorch_model.bin with
    ``chinese_clip.``-prefixed keys (+ a toy vocab.txt)."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save({"chinese_clip." + k: v for k, v in state_dict.items()},
               os.path.join(path, "pytorch_model.bin"))
    with open(os.path.join(path, "vocab.txt"), "w") as f:
        toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
        n = vocab_size or cfg["vocab_size"]
        toks += ["tok%d" % i for i in range(n - len(toks))]
        f.write("\n".join(toks) + "\n")


def reference_clip_app(ckpt_dir: str):
    """The reference ``CLIPApp`` loaded from a checkpoint dir."""
    install_shims()
    from easynlp.appzoo.clip.model import CLIPApp
    return CLIPApp(ckpt_dir)


def write_hf_checkpoint_dir(path: str, cfg: dict, state_dict: dict) -> None:
    """Synthetic checkpoint of the huggingface_clip flavour (appzoo/clip/model.py:73-104): config.json with
    text_config / vision_config (no model_type), pytorch_model.bin with ``text_encoder.*`` / ``vision_encoder.*`` /
    ``text_projection.*`` / ``vision_projection.*`` / ``logit_scale`` keys, a toy vocab.txt."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save(dict(state_dict), os.path.join(path, "pytorch_model.bin"))
    with open(os.path.join(path, "vocab.txt"), "w") as f:
        toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
        toks += ["tok%d" % i for i in range(cfg["text_config"]["vocab_size"] - len(toks))]
        f.write("\n".join(toks) + "\n")
