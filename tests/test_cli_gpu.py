"""SURVEY.md 8(f)2, producer side, on the GPU: ``coldrec_amd.main.main()`` end to end -- dataset files in, trained tables
and result file out -- in the formats a stock ColdRec checkout writes and reads.

Reference: main.py:149-301 (driver, summary, result file), model/MF.py:44-46 and model/LightGCN.py:45-47 (``torch.save``
of the tables: MF stores its live ``nn.Parameter``s, LightGCN the plain tensors its ``forward()`` built), read back by
model/DropoutNet.py:95-100 (``nn.ParameterDict`` over ``torch.load(./emb/<dataset>_cold_<object>_<backbone>_*.pt)``)."""
import json
import re

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

COMMON = ["--dataset", "toy", "--cold_object", "item", "--emb_size", "64", "--bs", "512", "--save_emb", "true",
          "--seed", "2024"]


def _json_blocks(path):
    """Every run block of a result file -> its parsed JSON tail (main.py:207-301: text block, then the JSON)."""
    text = open(path, encoding="utf-8").read()
    blocks = [b for b in text.split("\n" + "=" * 80 + "\n") if b.strip()]
    out = []
    for b in blocks:
        assert b.lstrip().startswith("=== ColdRec Run Result ===")
        for head in ("--- Hyperparameters ---", "--- Test Metrics (mean ± std) ---", "--- Efficiency ---"):
            assert head in b
        out.append((b, json.loads(b.split("--- JSON (machine-readable) ---\n", 1)[1])))
    return out


def test_main_writes_tables_and_result_file_that_the_generator_trainer_reads(tmp_path, monkeypatch, capsys):
    from coldrec_amd.main import main
    monkeypatch.chdir(tmp_path)                      # ./emb and ./result are relative to the working directory
    root = ["--data_root", str(tmp_path / "data"), "--result_dir", str(tmp_path / "result")]
    assert main(["--make_synthetic", "toy"] + COMMON + root) is None
    base = tmp_path / "data" / "toy" / "cold_item"
    for f in ("warm_train", "warm_val", "cold_item_val", "overall_val", "warm_test", "cold_item_test", "overall_test"):
        assert (base / (f + ".csv")).is_file()
    assert (base / "info_dict.pkl").is_file() and (tmp_path / "data" / "toy" / "toy_item_content.npy").is_file()

    # ---- BPR-MF: nn.Parameter files (model/MF.py:44-46 saves the live parameters)
    capsys.readouterr()
    pay_mf = main(["--model", "MF", "--epochs", "3"] + COMMON + root)
    out = capsys.readouterr().out
    assert re.search(r"Time: \d+\.\d{4}±\d+\.\d{4} seconds per completed training epoch\.", out)
    assert "Top-20 Cold-Start Test Performance:" in out and "Registered model: MF." in out
    import pickle
    info = pickle.load(open(base / "info_dict.pkl", "rb"))
    shapes = {}
    for side, rows in (("user", info["user_num"]), ("item", info["item_num"])):
        p = tmp_path / "emb" / f"toy_cold_item_MF_{side}_emb.pt"
        assert p.is_file()
        t = torch.load(p, map_location="cpu")        # what model/DropoutNet.py:97-98 does (torch's default loader)
        assert isinstance(t, nn.Parameter) and t.dtype == torch.float32 and t.shape == (rows, 64)
        assert torch.isfinite(t).all() and float(t.detach().abs().max()) > 0
        shapes[side] = t
    # ---- LightGCN: plain tensors (model/LightGCN.py:45-47 saves what forward() returned under no_grad)
    pay_lg = main(["--model", "LightGCN", "--layers", "3", "--epochs", "2"] + COMMON + root)
    for side in ("user", "item"):
        t = torch.load(tmp_path / "emb" / f"toy_cold_item_LightGCN_{side}_emb.pt", map_location="cpu")
        assert torch.is_tensor(t) and not isinstance(t, nn.Parameter) and not t.requires_grad
        assert t.shape == shapes[side].shape and torch.isfinite(t).all()
    # ---- the generator trainer reads either backbone's files (model/DropoutNet.py:95-100) and saves its own
    for backbone in ("MF", "LightGCN"):
        pay = main(["--model", "DropoutNet", "--backbone", backbone, "--epochs", "2", "--bs", "128"] +
                   [a for a in COMMON if a not in ("--bs", "512")] + root)
        assert set(pay) == {"10", "20"}
    gen = torch.load(tmp_path / "emb" / "toy_cold_item_DropoutNet_item_emb.pt", map_location="cpu")
    assert torch.is_tensor(gen) and gen.shape == shapes["item"].shape
    # a backbone that was never trained: the reference fails in torch.load; so does the mirror (no silent fallback)
    with pytest.raises(FileNotFoundError):
        main(["--model", "DropoutNet", "--backbone", "NGCF", "--epochs", "1"] + COMMON + root)

    # ---- result files (main.py:207-301): one block per run, appended, JSON tail == the returned payload
    mf_runs = _json_blocks(tmp_path / "result" / "MF" / "history.txt")
    assert len(mf_runs) == 1
    text, js = mf_runs[0]
    assert js["method"] == "MF" and js["metrics"] == pay_mf and js["hyperparameters"]["emb_size"] == 64
    assert set(js["metrics"]["20"]) == {"all", "cold", "warm"}
    assert set(js["metrics"]["20"]["all"]) == {"Hit", "Precision", "Recall", "NDCG"}
    assert js["efficiency"]["seconds_per_completed_epoch_mean"] > 0
    m = re.search(r"Top-20 Overall: Hit=(\d\.\d{4})±", text)
    assert m and abs(float(m.group(1)) - js["metrics"]["20"]["all"]["Hit"]["mean"]) < 5.1e-5
    assert "dataset: toy" in text and "cold_object: item" in text and "method: MF" in text
    lg_runs = _json_blocks(tmp_path / "result" / "LightGCN" / "history.txt")
    assert len(lg_runs) == 1 and lg_runs[0][1]["metrics"] == pay_lg and lg_runs[0][1]["hyperparameters"]["layers"] == 3
    dn_runs = _json_blocks(tmp_path / "result" / "DropoutNet" / "history.txt")
    assert [r[1]["hyperparameters"]["backbone"] for r in dn_runs] == ["MF", "LightGCN"]          # appended in order
    # --result_file + --result_overwrite replace instead of appending
    one = tmp_path / "one.txt"
    for _ in range(2):
        main(["--model", "MF", "--epochs", "1", "--result_file", str(one), "--result_overwrite"] + COMMON + root)
    assert len(_json_blocks(one)) == 1
    for top in pay_mf.values():                    # (the synthetic interactions are random: no quality bar, only sanity)
        for setting in top.values():
            assert all(0.0 <= m["mean"] <= 1.0 and m["std"] == 0.0 for m in setting.values())


def test_saved_mf_tables_equal_the_trainers_and_reload_bit_exact(tmp_path, monkeypatch):
    """_save_tables (model/MF.py:44-46): what lands on disk is the trainer's final (best) tables, bit for bit."""
    import argparse
    import types
    from coldrec_amd.model import AVAILABLE_MODELS
    from coldrec_amd.util.utils import set_seed
    from tests.test_host_logic import builder
    monkeypatch.chdir(tmp_path)
    _, data = builder()
    a = dict(dataset="toy", model="MF", epochs=2, layers=2, topN="10,20", bs=512, emb_size=64, lr=0.001, reg=0.0001,
             runs=1, seed=2024, use_gpu=True, save_emb=True, gpu_id=0, cold_object="item", backbone="MF", early_stop=10,
             eval_every=1)
    for model, is_param in (("MF", True), ("LightGCN", False)):
        set_seed(2024, True)
        a["model"] = model
        tr = AVAILABLE_MODELS[model](types.SimpleNamespace(args=argparse.Namespace(**a), data=data,
                                                           device=torch.device("cuda:0")))
        tr.run()
        for side, live in (("user", tr.user_emb), ("item", tr.item_emb)):
            t = torch.load(tmp_path / "emb" / f"toy_cold_item_{model}_{side}_emb.pt", map_location="cpu")
            assert isinstance(t, nn.Parameter) == is_param
            assert torch.equal(t.detach(), live.detach().cpu())
