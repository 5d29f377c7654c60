"""Drop-in boundary, end to end on the CPU: the REAL reference ``Trainer`` (easynlp/core/trainer.py) trains

  (a) the reference CLIPApp (or WukongCLIP) fed by the reference CLIPDataset (WukongCLIPDataset), and
  (b) the drop-in application fed by the drop-in dataset,

from the same checkpoint directory, TSV, arguments and sampler seed, and saves both with ``Trainer.save_checkpoint``.
Device compute of (b) is stood in for by the CPU oracle (tower encodes, similarity, loss, image pre-processing) -- so what is
under test is everything AROUND the kernels: construction from the checkpoint directory, parameter names / order (AdamW's
weight-decay groups are chosen by name, optimizers.py:490), autograd plumbing of forward / compute_loss, the dataset's
batch contract, ``config.to_json_string``, ``label_enumerate_values``, ``state_dict`` as the Trainer writes it.  After two
AdamW steps the two checkpoints must hold the same files, the same keys and the same weights.  Runs in a subprocess: the
reference keeps its arguments and its process group in globals."""
import os
import subprocess
import sys

import pytest

from oracle import ref_harness as R

_SCRIPT = r'''
import sys, os, json
ROOT, WORK, FLAVOUR = sys.argv[1], sys.argv[2], sys.argv[3]
AMP = FLAVOUR.endswith("+amp")          # --use_amp (trainer.py:57-62 GradScaler, :297-299 autocast, :327-329, :658-659)
FLAVOUR = FLAVOUR.split("+")[0]
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import ref_harness as R, clip_oracle as O, preprocess_oracle as P
R.install_shims()
from oracle import hf_clip_oracle as H, open_clip_oracle as OC
g = np.load(os.path.join(ROOT, "tests", "golden", "dataset_tsv_b7.npz"))
vocab = g["vocab"].tobytes().decode().split("\n")
ck = os.path.join(WORK, "ckpt")
# the datasets emit 224 x 224 images: 224-resolution models with tiny widths
if FLAVOUR == "chinese_clip":
    cfg = dict(O.CONFIGS["tiny"], vocab_size=len(vocab), image_resolution=224, vision_patch_size=32, vision_width=64, vision_layers=1)
    init = {"chinese_clip." + k: v for k, v in O.make_state_dict(cfg, 5).items()}
    R.write_checkpoint_dir(ck, cfg, O.make_state_dict(cfg, 5))
    open(os.path.join(ck, "vocab.txt"), "wb").write(g["vocab"].tobytes() + b"\n")
elif FLAVOUR == "huggingface_clip":
    cfg = dict(text_config=dict(H.HF_CONFIGS["hf_tiny"]["text_config"], vocab_size=len(vocab)),
               vision_config=dict(H.HF_CONFIGS["hf_tiny"]["vision_config"], hidden_size=64, intermediate_size=256, num_hidden_layers=1,
                                  num_attention_heads=1, image_size=224, patch_size=32), projection_dim=64)
    init = H.make_state_dict(cfg, 5)
    R.write_hf_checkpoint_dir(ck, cfg, init)
    open(os.path.join(ck, "vocab.txt"), "wb").write(g["vocab"].tobytes() + b"\n")
elif FLAVOUR == "wukong":
    from oracle import wukong_oracle as WK
    g = np.load(os.path.join(ROOT, "tests", "golden", "wukong_dataset_b5.npz"))       # RGB rows, BERT-layout vocab ([SEP] = 102)
    vocab = g["vocab"].tobytes().decode().split("\n")
    cfg = {"model": {"visual": dict(input_resolution=224, patch_size=32, width=64, layers=1, heads=1, output_dim=64),
                     "text": dict(context_length=32, vocab_size=len(vocab), output_dim=64, width=64, layers=2, heads=1)}}
    init = WK.make_state_dict(cfg, 5, small_embeddings=False)
    os.makedirs(ck)
    json.dump(cfg, open(os.path.join(ck, "config.json"), "w"))
    torch.save(init, os.path.join(ck, "pytorch_model.bin"))
    open(os.path.join(ck, "vocab.txt"), "wb").write(g["vocab"].tobytes() + b"\n")
elif FLAVOUR == "text2video":
    import gzip
    from oracle import text2video_oracle as TV
    bpe = np.load(os.path.join(ROOT, "tests", "golden", "openclip_bpe_corpus.npz"))
    g = np.load(os.path.join(ROOT, "tests", "golden", "t2v_dataset_b3.npz"))
    cfg = dict(OC.OPENCLIP_CONFIGS["oc_tiny"], image_resolution=224, vision_patch_size=32, vision_width=64, vision_layers=1,
               context_length=77, vocab_size=int(bpe["meta"][0]))
    init = {"open_clip." + k: v for k, v in OC.make_state_dict(cfg, 5).items()}
    os.makedirs(ck)
    json.dump(cfg, open(os.path.join(ck, "config.json"), "w"))
    torch.save(init, os.path.join(ck, "pytorch_model.bin"))
    with gzip.open(os.path.join(ck, "vocab.txt"), "wb") as f:
        f.write(bpe["merges"].tobytes())
    for k in g.files:                                   # the clips: one directory of frame images each
        if k.startswith("png/"):
            os.makedirs(os.path.dirname(os.path.join(WORK, k[4:])), exist_ok=True)
            open(os.path.join(WORK, k[4:]), "wb").write(g[k].tobytes())
    g = {"tsv": np.frombuffer("".join("%s\t%s\n" % (c, os.path.join(WORK, "clip%d" % i)) for i, c in enumerate(g["captions"])).encode(), dtype=np.uint8)}
else:
    import gzip
    bpe = np.load(os.path.join(ROOT, "tests", "golden", "openclip_bpe_corpus.npz"))
    cfg = dict(OC.OPENCLIP_CONFIGS["oc_tiny"], image_resolution=224, vision_patch_size=32, vision_width=64, vision_layers=1,
               context_length=77, vocab_size=int(bpe["meta"][0]))
    init = {"open_clip." + k: v for k, v in OC.make_state_dict(cfg, 5).items()}
    os.makedirs(ck)
    json.dump(cfg, open(os.path.join(ck, "config.json"), "w"))
    torch.save(init, os.path.join(ck, "pytorch_model.bin"))
    with gzip.open(os.path.join(ck, "vocab.txt"), "wb") as f:
        f.write(bpe["merges"].tobytes())
tsv = os.path.join(WORK, "train.tsv")
open(tsv, "wb").write(g["tsv"].tobytes())
sys.argv = ["x", "--mode", "train", "--tables", tsv + "," + tsv, "--input_schema", "text:str:1,image:str:1",
            "--first_sequence", "text", "--second_sequence", "image", "--checkpoint_dir", os.path.join(WORK, "out"),
            "--learning_rate", "1e-4", "--epoch_num", "1", "--random_seed", "42", "--save_checkpoint_steps", "2",
            "--sequence_length", "20", "--micro_batch_size", "3" if FLAVOUR in ("wukong", "text2video") else "4",
            "--app_name", {"wukong": "wukong_clip", "text2video": "clip4clip"}.get(FLAVOUR, "clip"), "--worker_gpu", "0",
            "--user_defined_parameters", "pretrain_model_name_or_path=" + ck] + (["--use_amp"] if AMP else [])
from easynlp.utils import initialize_easynlp
args = initialize_easynlp()
from easynlp.core.trainer import Trainer
assert bool(args.use_amp) == AMP
SCHEMA = dict(input_schema="text:str:1,image:str:1", first_sequence="text", second_sequence="image")

SCORES = {}
def train(app, ds, out, evaluator):
    args.checkpoint_dir = out
    os.makedirs(out, exist_ok=True)
    tr = Trainer(model=app, train_dataset=ds, evaluator=evaluator)
    assert hasattr(tr, "_scaler") == AMP          # (on this CPU the scaler and autocast are constructed and disable themselves)
    torch.manual_seed(123)                       # the RandomSampler draws its permutation from the global generator
    tr.train()                                   # evaluates at step 2 and saves the best checkpoint itself (trainer.py:367-384)
    SCORES[out] = (evaluator.best_valid_score, os.path.exists(os.path.join(out, "pytorch_model.bin")))

if FLAVOUR == "wukong":
    from easynlp.appzoo.wukong_clip.model import WukongCLIP as RefApp
    from easynlp.appzoo.wukong_clip.data import WukongCLIPDataset as RefDataset
elif FLAVOUR == "text2video":
    from easynlp.appzoo.text2video_retrieval.model import Text2VideoRetrieval as RefApp
    from easynlp.appzoo.text2video_retrieval.data import Text2VideoRetrievalDataset as RefDataset
else:
    from easynlp.appzoo.clip.model import CLIPApp as RefApp
    from easynlp.appzoo.clip.data import CLIPDataset as RefDataset
if FLAVOUR == "wukong":
    from easynlp.appzoo.wukong_clip.evaluator import WukongCLIPEvaluator as RefEval
    ref_eval = RefEval(valid_dataset=RefDataset(ck, tsv, 20, **SCHEMA), user_defined_parameters={}, eval_batch_size=4)
elif FLAVOUR == "text2video":
    from easynlp.appzoo.text2video_retrieval.evaluator import Text2VideoRetrievalEvaluator as RefEval
    ref_eval = RefEval(valid_dataset=RefDataset(ck, tsv, 20, **SCHEMA), eval_batch_size=4)
else:
    from easynlp.appzoo.clip.evaluator import CLIPEvaluator as RefEval
    ref_eval = RefEval(valid_dataset=RefDataset(ck, tsv, 20, **SCHEMA), eval_batch_size=4)
train(RefApp(ck), RefDataset(ck, tsv, 20, **SCHEMA), os.path.join(WORK, "out_reference"), ref_eval)

from easynlp_amd import lib as L
from easynlp_amd.appzoo.clip import model as CM
from easynlp_amd.appzoo.clip.data import CLIPDataset
def oracle_preprocess(images, size=224, crop=224, mean=L.CLIP_MEAN, std=L.CLIP_STD, device="cpu"):
    outs = []
    for im in images:
        a = np.asarray(im)
        outs.append(P.preprocess(np.repeat(a[:, :, None], 3, axis=2) if a.ndim == 2 else a, size=size, crop=crop))
    return torch.from_numpy(np.stack(outs))
def oracle_encode(self, pixel_values=None, input_ids=None, token_type_ids=None, attention_mask=None, pack_hint=None):
    if FLAVOUR == "wukong":
        fo = WK.wukong_forward({"model." + n: p for n, p in self.model.named_parameters()}, cfg, pixel_values, input_ids)
        return fo["image_features"], fo["text_features"]
    if FLAVOUR == "huggingface_clip":
        out = H.hf_clip_forward(dict(self._hf_params), cfg, pixel_values, input_ids, token_type_ids, attention_mask)
        return out["image_embeds"], out["text_embeds"]
    if FLAVOUR == "text2video":          # (the application pools the frames itself: encode returns per-frame features)
        sd = {n: p for n, p in self.open_clip.named_parameters()}
        img = O.l2_normalize(O.vit_forward(sd, OC.chinese_style_config(cfg), pixel_values)) if pixel_values is not None else None
        txt = O.l2_normalize(OC.text_forward(sd, cfg, input_ids)) if input_ids is not None else None
        return img, txt
    if FLAVOUR == "open_clip":
        out = OC.open_clip_forward({n: p for n, p in self.open_clip.named_parameters()}, cfg, pixel_values, input_ids)
        return out["image_embeds"], out["text_embeds"]
    sd = {n: p for n, p in self.chinese_clip.named_parameters()}
    return (O.encode_image(sd, self.raw_config, pixel_values) if pixel_values is not None else None,
            O.encode_text(sd, self.raw_config, input_ids) if input_ids is not None else None)
class OracleSimilarity:
    apply = staticmethod(lambda t, i, ls: (t @ i.t()) * ls.exp())
class OracleInfoNCE:
    apply = staticmethod(lambda logits: O.clip_loss(logits))
L.preprocess_images = oracle_preprocess
CM.CLIPApp.encode = oracle_encode
CM._SimilarityFn, CM._InfoNCEFn = OracleSimilarity, OracleInfoNCE
DropApp = CM.CLIPApp
if FLAVOUR == "wukong":
    import easynlp_amd.appzoo.wukong_clip.model as WM
    from easynlp_amd.appzoo.wukong_clip import WukongCLIPDataset as CLIPDataset
    WM._SimilarityFn, WM._InfoNCEFn = OracleSimilarity, OracleInfoNCE
    DropApp = WM.WukongCLIP
if FLAVOUR == "text2video":
    import easynlp_amd.appzoo.text2video_retrieval.model as TM
    from easynlp_amd.appzoo.text2video_retrieval import Text2VideoRetrievalDataset as CLIPDataset
    TM._SimilarityFn = OracleSimilarity
    DropApp = TM.Text2VideoRetrieval
def oracle_recall(t, v, ks=(1, 5, 10)):
    r = O.recall_at_k(t.float(), v.float())
    return r, tuple(int(round(x * t.shape[0])) for x in r[1:])
from easynlp_amd.appzoo.clip import evaluator as EV
EV.recall_at_k = oracle_recall
if FLAVOUR == "wukong":
    import easynlp_amd.appzoo.wukong_clip.evaluator as WE
    WE.recall_at_k = oracle_recall
    my_eval = WE.WukongCLIPEvaluator(valid_dataset=CLIPDataset(ck, tsv, 20, **SCHEMA), user_defined_parameters={}, eval_batch_size=4)
elif FLAVOUR == "text2video":
    import easynlp_amd.appzoo.text2video_retrieval.evaluator as TE
    TE.recall_at_k = oracle_recall
    my_eval = TE.Text2VideoRetrievalEvaluator(valid_dataset=CLIPDataset(ck, tsv, 20, **SCHEMA), eval_batch_size=4)
else:
    my_eval = EV.CLIPEvaluator(valid_dataset=CLIPDataset(ck, tsv, 20, **SCHEMA), eval_batch_size=4)
train(DropApp(ck), CLIPDataset(ck, tsv, 20, **SCHEMA), os.path.join(WORK, "out_dropin"), my_eval)

fa, fb = (sorted(os.listdir(os.path.join(WORK, d))) for d in ("out_reference", "out_dropin"))
a = torch.load(os.path.join(WORK, "out_reference", "pytorch_model.bin"), map_location="cpu")
b = torch.load(os.path.join(WORK, "out_dropin", "pytorch_model.bin"), map_location="cpu")
diff = max(float((a[k].float() - b[k].float()).abs().max()) for k in a) if set(a) == set(b) else -1.0
moved = max(float((a[k].float() - init[k].float().reshape(a[k].shape)).abs().max()) for k in a if "position_ids" not in k)
ca, cb = (json.load(open(os.path.join(WORK, d, "config.json"))) for d in ("out_reference", "out_dropin"))
# each implementation loads the directory the other one's training run wrote
ref_on_dropin = RefApp(os.path.join(WORK, "out_dropin")).state_dict()
dropin_on_ref = DropApp(os.path.join(WORK, "out_reference")).state_dict()
cross = max(max(float((ref_on_dropin[k].float() - b[k].float().reshape(ref_on_dropin[k].shape)).abs().max()) for k in b if "position_ids" not in k),
            max(float((dropin_on_ref[k].float() - a[k].float().reshape(dropin_on_ref[k].shape)).abs().max()) for k in a if "position_ids" not in k))
print("RESULT " + json.dumps({"files_equal": fa == fb, "files": fa, "keys_equal": set(a) == set(b), "n_keys": len(a),
                              "max_diff": diff, "moved": moved, "config_equal": ca == cb, "cross_load_diff": cross,
                              "scores": [SCORES[os.path.join(WORK, d)] for d in ("out_reference", "out_dropin")]}))
'''


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
@pytest.mark.parametrize("flavour", ["chinese_clip", "chinese_clip+amp", "huggingface_clip", "open_clip", "wukong", "text2video"])
def test_reference_trainer_trains_the_dropin_like_the_reference(tmp_path, flavour):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text(_SCRIPT)
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, str(script), root, str(tmp_path), flavour], capture_output=True, text=True, cwd=str(tmp_path),
                       timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads(lines[-1][len("RESULT "):])
    assert res["files_equal"] and "pytorch_model.bin" in res["files"] and "config.json" in res["files"], res
    assert res["keys_equal"], res
    # config.json: the raw dict for chinese_clip / open_clip (Config_Wrapper, model.py:32-38).  In the huggingface_clip flavour the
    # reference writes its expanded CLIPConfig (every PretrainedConfig default), the drop-in the text_config / vision_config it
    # was given; both re-load in both implementations, which is what cross_load_diff checks
    assert res["config_equal"] or flavour == "huggingface_clip", res
    assert res["cross_load_diff"] == 0.0, res
    # the Trainer evaluated both with their own evaluator class, got the same mean recall and saved the best checkpoint itself
    (ref_score, ref_saved), (my_score, my_saved) = res["scores"]
    assert ref_saved and my_saved and 0.0 < ref_score <= 1.0 and abs(ref_score - my_score) < 1e-9, res
    assert res["moved"] > 5e-5, res                       # the two AdamW steps did change the weights ...
    assert 0 <= res["max_diff"] < 2e-6, res                # ... and both runs ended in the same place
