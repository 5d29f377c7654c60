"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/ezclip.h declares, the parameter table equals the
reference's state_dict contract, and the host logic fails loudly (no compute
calls here -- there is no GPU in the build container)."""
import ctypes as C
import os
import re
import sys

import pytest

from easynlp_amd import lib as L
from oracle import clip_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "ezclip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ezclip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "include/ezclip.h declares %s but the .so does not export it" % s
    # and the Python binding knows a signature for each of them
    assert set(syms) == set(L.SIGNATURES), set(syms) ^ set(L.SIGNATURES)
    assert b"gfx950" in lib.ezclip_version()


def make_handle(cfg_name, dtype):
    from easynlp_amd.appzoo.clip.model import _cfg_struct
    lib = L.load()
    cs = _cfg_struct(O.CONFIGS[cfg_name], dtype)
    h = C.c_void_p()
    rc = lib.ezclip_create(C.byref(cs), C.byref(h))
    return lib, h, rc


@pytest.mark.parametrize("cfg_name", ["tiny", "small", "vitb16_bertbase", "vitl14_robertabase", "large_text"])
@pytest.mark.parametrize("dtype", [L.DTYPE_F32, L.DTYPE_BF16])
def test_param_table_matches_reference_state_dict(cfg_name, dtype):
    lib, h, rc = make_handle(cfg_name, dtype)
    assert rc == 0, L.last_error()
    shapes = O.param_shapes(O.CONFIGS[cfg_name])
    n = lib.ezclip_num_params(h)
    name, shape, ndim = C.c_char_p(), (C.c_int64 * 8)(), C.c_int()
    got = {}
    for i in range(n):
        assert lib.ezclip_param_info(h, i, C.byref(name), shape, C.byref(ndim)) == 0
        got[name.value.decode()] = tuple(shape[j] for j in range(ndim.value))
    # (+ the two optional projection biases of the huggingface_clip branch, include/ezclip.h)
    E = O.CONFIGS[cfg_name]["embed_dim"]
    assert got == dict(shapes, **{"visual.proj_bias": (E,), "text_projection_bias": (E,)})
    # workspace sizes are monotone in batch and larger with save_for_backward
    a = lib.ezclip_image_workspace_bytes(h, 2, 0)
    b = lib.ezclip_image_workspace_bytes(h, 4, 0)
    c = lib.ezclip_image_workspace_bytes(h, 4, 1)
    assert 0 < a < b < c
    t0 = lib.ezclip_text_workspace_bytes(h, 4, 16, 0)
    t1 = lib.ezclip_text_workspace_bytes(h, 4, 32, 0)
    t2 = lib.ezclip_text_workspace_bytes(h, 4, 32, 1)
    assert 0 < t0 < t1 < t2
    assert lib.ezclip_shadow_bytes(h, 1) > lib.ezclip_shadow_bytes(h, 0) > 0
    lib.ezclip_destroy(h)


def test_bad_config_and_bad_calls_fail_loudly():
    from easynlp_amd.appzoo.clip.model import _cfg_struct
    lib = L.load()
    bad = dict(O.CONFIGS["tiny"], vision_width=100)
    h = C.c_void_p()
    cs = _cfg_struct(bad, L.DTYPE_BF16)
    assert lib.ezclip_create(C.byref(cs), C.byref(h)) != 0
    assert "vision_width" in L.last_error()
    lib, h, rc = make_handle("tiny", L.DTYPE_F32)
    assert rc == 0
    shp = (C.c_int64 * 1)(7)
    assert lib.ezclip_bind_param(h, b"no.such.param", 16, None, shp, 1) != 0
    assert "unknown parameter" in L.last_error()
    assert lib.ezclip_bind_param(h, b"visual.class_embedding", 16, None, shp, 1) != 0   # wrong shape
    # forward before binding parameters must refuse, not crash
    assert lib.ezclip_encode_image(h, 256, 1, 256, 256, 1 << 30, 0, None) != 0
    assert "refresh_weights" in L.last_error()
    lib.ezclip_destroy(h)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "easynlp_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt, fn


def test_no_cpu_fallback_in_host_module(tmp_path):
    """The app refuses CPU tensors instead of silently computing elsewhere."""
    import torch
    from easynlp_amd.appzoo.clip import CLIPApp
    from oracle import ref_harness as R
    cfg = O.CONFIGS["tiny"]
    R.write_checkpoint_dir(str(tmp_path), cfg, O.make_state_dict(cfg, 3))
    app = CLIPApp(str(tmp_path))
    # reference checkpoint contract: same keys as the reference state_dict (+ chinese_clip. prefix)
    keys = set(app.state_dict().keys())
    want = {"chinese_clip." + k for k in O.param_shapes(cfg)} | {"chinese_clip.bert.embeddings.position_ids"}
    assert keys == want
    sd = O.make_state_dict(cfg, 3)
    for k, v in sd.items():
        assert torch.equal(app.state_dict()["chinese_clip." + k], v), k
    px, ids = O.make_inputs(cfg, 2, 8, 0)
    with pytest.raises(L.EzclipError):
        app({"pixel_values": px, "input_ids": ids})


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_pipelined_gemm_isa_audit(tmp_path):
    """The 8-phase GEMM kernels issue their VMEM by hand (inline asm) and count their own waits; hipcc knows nothing
    about those loads and stores.  Compile the two kernel files for gfx950 and audit the ISA: no register spills (a
    spill reload carries a compiler-counted vmcnt that drains the DMA pipeline), no instruction touching an asm-load
    destination before the counted wait that guards it, no VALU overwriting the data of a 16-byte buffer store within
    its wait states (tools/audit_asm_loads.py)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "easynlp_amd", "csrc")
    import concurrent.futures

    def compile_one(src):
        return src, subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps",
                              "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(csrc, src), "-I", csrc,
                              "-o", str(tmp_path / (src + ".o"))], cwd=str(tmp_path), capture_output=True, text=True)

    # the weight-gradient kernel + dispatcher, and the three translation units holding the NT kernel's instantiations
    with concurrent.futures.ThreadPoolExecutor(4) as ex:
        results = list(ex.map(compile_one, ("gemm8p.hip", "gemm8p_nt_a.hip", "gemm8p_nt_b.hip", "gemm8p_nt_c.hip")))
    for src, out in results:
        assert out.returncode == 0, out.stderr[-2000:]
        spills = [l for l in out.stderr.splitlines() if "VGPRs Spill" in l and not l.rstrip().endswith("Spill: 0 [-Rpass-analysis=kernel-resource-usage]")]
        assert not spills, spills
        asm = str(tmp_path / (src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s"))
        assert os.path.exists(asm)
        audit = subprocess.run([sys.executable, os.path.join(root, "tools", "audit_asm_loads.py"), asm],
                               capture_output=True, text=True)
        assert audit.returncode == 0 and "audit: ok" in audit.stdout, audit.stdout[-2000:]


def test_all_three_checkpoint_flavours_keep_the_reference_state_dict(tmp_path):
    """chinese_clip / huggingface_clip / open_clip (appzoo/clip/model.py:52-104): the drop-in CLIPApp exposes exactly the
    reference's state-dict keys and round-trips the weights (checked here without a GPU: construction and loading are
    host-side)."""
    import json
    import torch
    from easynlp_amd.appzoo.clip import CLIPApp
    from oracle import hf_clip_oracle as H
    from oracle import open_clip_oracle as OC
    from oracle import ref_harness as R
    # huggingface_clip
    d1 = tmp_path / "hf"
    cfg = H.HF_CONFIGS["hf_small"]
    sd = H.make_state_dict(cfg, 2)
    R.write_hf_checkpoint_dir(str(d1), cfg, sd)
    app = CLIPApp(str(d1))
    assert app.model_type == "huggingface_clip"
    got = app.state_dict()
    assert set(got) == set(H.param_shapes(cfg)) | {"text_encoder.embeddings.position_ids",
                                                   "vision_encoder.vision_model.embeddings.position_ids"}
    for k, v in sd.items():
        assert torch.equal(got[k].reshape(v.shape), v), k
    assert {n for n, _ in app.named_parameters()} - {"logit_scale_param"} == set(H.param_shapes(cfg)) - {"logit_scale"}
    # the library-side view: packed in_proj = [q; k; v], transposed projections
    t = app._hf.library_tensors(app._engine.names)
    vm = "vision_encoder.vision_model.encoder.layers.1.self_attn."
    assert torch.equal(t["visual.transformer.resblocks.1.attn.in_proj_weight"],
                       torch.cat([sd[vm + "q_proj.weight"], sd[vm + "k_proj.weight"], sd[vm + "v_proj.weight"]]))
    assert torch.equal(t["text_projection"], sd["text_projection.weight"].t())
    assert t["bert.pooler.dense.bias"].data_ptr() == app._hf_params["text_encoder.pooler.dense.bias"].data_ptr()
    # open_clip
    d2 = tmp_path / "oc"
    os.makedirs(str(d2))
    ocfg = OC.OPENCLIP_CONFIGS["oc_tiny"]
    osd = OC.make_state_dict(ocfg, 4)
    with open(os.path.join(str(d2), "config.json"), "w") as f:
        json.dump(ocfg, f)
    torch.save({"open_clip." + k: v for k, v in osd.items()}, os.path.join(str(d2), "pytorch_model.bin"))
    app2 = CLIPApp(str(d2))
    assert app2.model_type == "open_clip"
    got2 = app2.state_dict()
    assert set(got2) == {"open_clip." + k for k in OC.param_shapes(ocfg)}
    for k, v in osd.items():
        assert torch.equal(got2["open_clip." + k], v), k


def test_wukong_and_text2video_applications_construct_from_reference_checkpoints(tmp_path):
    """WukongCLIP (appzoo/wukong_clip/model.py:14-42) and Text2VideoRetrieval (appzoo/text2video_retrieval/model.py:46-62)
    load the reference's checkpoint directories and expose the reference's state-dict keys; the two Wukong options reach
    the library (construction, option setting and loading are host-side: no GPU needed)."""
    import json
    import torch
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.text2video_retrieval import Text2VideoRetrieval
    from easynlp_amd.appzoo.wukong_clip import WukongCLIP
    from oracle import clip_oracle as O
    from oracle import open_clip_oracle as OC
    from oracle import ref_harness as R
    from oracle import wukong_oracle as WK
    d1 = tmp_path / "wk"
    os.makedirs(str(d1))
    cfg = WK.WUKONG_CONFIGS["wk_small"]
    sd = WK.make_state_dict(cfg, 2)
    with open(os.path.join(str(d1), "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save(sd, os.path.join(str(d1), "pytorch_model.bin"))
    app = WukongCLIP.from_pretrained(str(d1), user_defined_parameters={})
    got = app.state_dict()
    assert set(got) == set(sd) and {n for n, _ in app.named_parameters()} == set(sd)
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    assert json.loads(app.config.to_json_string()) == cfg
    # the library sees its own names bound to the same storage
    assert app._params["token_embedding.weight"].data_ptr() == dict(app.named_parameters())["model.text_encoder.embedding_table"].data_ptr()
    # options: accepted on this handle, rejected where they do not apply / out of range
    app._engine.set_option(L.OPT_BLOCK_LN_EPS, 1e-7)
    app._engine.set_option(L.OPT_TEXT_EOT_ID, 102)
    with pytest.raises(L.EzclipError):
        app._engine.set_option(L.OPT_TEXT_EOT_ID, 10 ** 7)
    with pytest.raises(L.EzclipError):
        app._engine.set_option(L.OPT_BLOCK_LN_EPS, 0.0)
    # a checkpoint with a missing tower key does not load silently (WukongModel loads each tower strictly)
    d1b = tmp_path / "wk_bad"
    os.makedirs(str(d1b))
    with open(os.path.join(str(d1b), "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save({k: v for k, v in sd.items() if not k.endswith("ln_final.bias")}, os.path.join(str(d1b), "pytorch_model.bin"))
    with pytest.raises(RuntimeError):
        WukongCLIP(str(d1b))
    # the token-id option is only defined for the CLIP text tower
    d0 = tmp_path / "cc"
    ccfg = O.CONFIGS["tiny"]
    R.write_checkpoint_dir(str(d0), ccfg, O.make_state_dict(ccfg, 1))
    from easynlp_amd.appzoo.clip import CLIPApp
    capp = CLIPApp(str(d0))
    with pytest.raises(L.EzclipError):
        capp._engine.set_option(L.OPT_TEXT_EOT_ID, 102)
    with pytest.raises(L.EzclipError):
        Text2VideoRetrieval(str(d0))                      # only open_clip checkpoints (model.py:54-62)
    d2 = tmp_path / "oc"
    os.makedirs(str(d2))
    ocfg = OC.OPENCLIP_CONFIGS["oc_tiny"]
    osd = OC.make_state_dict(ocfg, 4)
    with open(os.path.join(str(d2), "config.json"), "w") as f:
        json.dump(ocfg, f)
    torch.save({"open_clip." + k: v for k, v in osd.items()}, os.path.join(str(d2), "pytorch_model.bin"))
    t2v = Text2VideoRetrieval.from_pretrained(str(d2))
    assert set(t2v.state_dict()) == {"open_clip." + k for k in OC.param_shapes(ocfg)}


def test_huggingface_checkpoint_without_logit_scale_keeps_the_default(tmp_path):
    """appzoo/clip/model.py:101-104: 'logit_scale' is optional in a huggingface_clip checkpoint (ln(1 / 0.07) otherwise);
    any other missing tensor is an error"""
    import math
    import torch
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.clip import CLIPApp
    from oracle import hf_clip_oracle as H
    from oracle import ref_harness as R
    cfg = H.HF_CONFIGS["hf_tiny"]
    sd = H.make_state_dict(cfg, 2)
    R.write_hf_checkpoint_dir(str(tmp_path), cfg, {k: v for k, v in sd.items() if k != "logit_scale"})
    app = CLIPApp(str(tmp_path))
    assert abs(float(app.state_dict()["logit_scale"].reshape(-1)[0]) - math.log(1 / 0.07)) < 1e-6
    assert torch.equal(app.state_dict()["text_projection.weight"], sd["text_projection.weight"])
    R.write_hf_checkpoint_dir(str(tmp_path), cfg, {k: v for k, v in sd.items() if k != "text_projection.bias"})
    with pytest.raises(L.EzclipError):
        CLIPApp(str(tmp_path))
