"""Pin the CPU oracle against the fixtures captured from the reference itself.

Runs without a GPU.  Every golden file under tests/golden is replayed through
oracle/carca_oracle.py; tolerances are fp32 round-off (the reference's own
fp32-vs-fp64 gap is 4.8e-7).
"""
import numpy as np
import pytest
import torch

from oracle import carca_oracle as O
from tests.golden_util import G1_NAMES, G7_NAMES, G9_NAMES, load, oracle_config

ATOL = 2e-6


def _eval_inputs(ins, prefix=""):
    g = lambda k: ins[prefix + k]  # noqa: E731
    return (g("p_x"), g("p_a"), g("p_c")), [(g("o_x"), g("o_a"), g("o_c"))]


def _train_inputs(ins, prefix=""):
    g = lambda k: ins[prefix + k]  # noqa: E731
    L = g("p_x").shape[1]
    pos = tuple(g(k)[:, :L] for k in ("o_x", "o_a", "o_c"))
    neg = tuple(g(k)[:, L:] for k in ("o_x", "o_a", "o_c"))
    return (g("p_x"), g("p_a"), g("p_c")), [pos, neg]


def _check_eval(fx, prefix=""):
    cfg = oracle_config(fx.cfg)
    profile, targets = _eval_inputs(fx.ins, prefix)
    trace = {}
    y = O.carca_forward(fx.params, cfg, profile, targets, training=False, trace=trace)
    assert y.shape == fx.outs["y"].shape
    assert torch.allclose(y, fx.outs["y"], atol=ATOL, rtol=0)
    for k in ["p_mask", "p_embed", "p_final", "o_embed0", "dec_w0"] + [f"block{i}" for i in range(cfg.n_blocks)]:
        assert torch.allclose(trace[k], fx.outs[k], atol=1e-5, rtol=1e-5), k
    loss = O.bce_loss(y, fx.ins[prefix + "y_true"], O.get_mask(targets[0][0]))
    assert abs(float(loss) - float(fx.outs["loss"])) < 1e-6


def _check_train(fx, in_prefix="", out_prefix=""):
    cfg = oracle_config(fx.cfg)
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "enc.pe" not in k else v)
              for k, v in fx.params.items()}
    profile, targets = _train_inputs(fx.ins, in_prefix)
    y = O.carca_forward(params, cfg, profile, targets, training=True)
    assert torch.allclose(y, fx.outs[out_prefix + "y"], atol=ATOL if cfg.decoder == "ca" else 1e-5, rtol=0)
    mask = O.get_mask(fx.ins[in_prefix + "o_x"])
    loss = O.bce_loss(y, fx.ins[in_prefix + "y_true"], mask)
    ref_loss = float(fx.outs[out_prefix + "loss"])
    assert abs(float(loss.detach()) - ref_loss) < 1e-6 * max(1.0, abs(ref_loss))
    loss.backward()
    for k, v in params.items():
        gk = out_prefix + "grad/" + k
        if gk not in fx.outs:
            continue
        ref = fx.outs[gk]
        got = v.grad if v.grad is not None else torch.zeros_like(ref)
        scale = float(ref.abs().max()) + 1e-12
        assert float((got - ref).abs().max()) <= 2e-5 * scale + 1e-8, k


@pytest.mark.parametrize("name", G1_NAMES)
def test_g1_eval_forward_and_intermediates(name):
    _check_eval(load("g1_" + name))


@pytest.mark.parametrize("name", G1_NAMES)
def test_g2_train_forward_loss_grads(name):
    _check_train(load("g2_" + name))


def test_g2_first_valid_target_sees_no_keys():
    """tril(-1) in training: the first valid target slot attends nothing (carca.py:339)."""
    fx = load("g2_d90h2")
    cfg = oracle_config(fx.cfg)
    profile, targets = _train_inputs(fx.ins)
    trace = {}
    O.carca_forward(fx.params, cfg, profile, targets, training=True, trace=trace)
    p_x = fx.ins["p_x"]
    for u in range(p_x.shape[0]):
        nz = torch.nonzero(p_x[u])
        if len(nz) == 0:
            continue
        first = int(nz[0])
        assert float(trace["dec_w0"][u, :, first].abs().sum()) == 0.0


def test_g3_adam_three_steps():
    fx = load("g3_adam")
    cfg = oracle_config(fx.cfg)
    params = {k: v.clone().requires_grad_(True) for k, v in fx.params.items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3, weight_decay=0.0, betas=(0.9, 0.98))
    profile, targets = _train_inputs(fx.ins)
    mask = O.get_mask(fx.ins["o_x"])
    for step in range(3):
        opt.zero_grad()
        y = O.carca_forward(params, cfg, profile, targets, training=True)
        loss = O.bce_loss(y, fx.ins["y_true"], mask)
        assert abs(float(loss) - float(fx.outs[f"loss{step}"])) < 2e-6
        loss.backward()
        opt.step()
    for k, v in params.items():
        # A key bias shifts every score of a softmax row equally, so its true gradient is 0 and
        # what reaches Adam is round-off that Adam rescales to +-lr steps: not comparable.
        if k.endswith("WK.bias"):
            continue
        # Adam's first steps are +-lr whatever the gradient size, so allow a few 1e-6
        assert torch.allclose(v.detach(), fx.outs["final/" + k], atol=5e-5, rtol=0), k


def test_g4_metrics():
    fx = load("g4_metrics")
    scores, y_true = fx.ins["scores"], fx.ins["y_true"]
    for k in (1, 5, 10, 20):
        hr, ndcg = O.hr_ndcg_sums(scores, k)
        assert hr == float(fx.outs[f"hr{k}"])
        assert abs(ndcg - float(fx.outs[f"ndcg{k}"])) < 1e-4
        hr2, ndcg2 = O.hr_ndcg_sort(scores, y_true, k)
        assert hr2 == hr and abs(ndcg2 - ndcg) < 1e-4


def test_g6_squeeze_quirks():
    fx = load("g6_shapes")
    cfg = oracle_config(fx.cfg)
    for tag, key in (("b1/", "y_b1"), ("n1/", "y_n1")):
        profile, targets = _eval_inputs(fx.ins, tag)
        y = O.carca_forward(fx.params, cfg, profile, targets, training=False)
        assert tuple(y.shape) == tuple(fx.outs[key].shape)
        assert torch.allclose(y, fx.outs[key], atol=ATOL, rtol=0)
    assert fx.outs["y_b1"].dim() == 1 and fx.outs["y_n1"].dim() == 1


@pytest.mark.parametrize("name", G7_NAMES)
def test_g7_variants(name):
    fx = load("g7_" + name)
    _check_eval(fx)
    _check_train(fx, in_prefix="train/", out_prefix="train/")


@pytest.mark.parametrize("name", G9_NAMES)
def test_g9_ablation_embeddings_and_decoders(name):
    """The reference's other embeddings / decoders (carca.py:98-198,352-399): eval scores + loss, train scores + grads."""
    fx = load("g9_" + name)
    cfg = oracle_config(fx.cfg)
    profile, targets = _eval_inputs(fx.ins)
    y = O.carca_forward(fx.params, cfg, profile, targets, training=False)
    assert y.shape == fx.outs["y"].shape
    # un-normalised dot-product logits reach +-40 (x8.6 more with the decay weights): fp32 noise of 3e-6 relative on
    # the logit is up to 1e-5 on the score; still 10x inside the 1e-4 bar
    assert torch.allclose(y, fx.outs["y"], atol=1e-5, rtol=0)
    loss = O.bce_loss(y, fx.ins["y_true"], O.get_mask(targets[0][0]))
    ref = float(fx.outs["loss"])  # saturated scores: the loss is O(10), compare relatively
    assert abs(float(loss) - ref) < 1e-6 * max(1.0, abs(ref))
    _check_train(fx, in_prefix="train/", out_prefix="train/")


@pytest.mark.parametrize("tag", ["f32", "f37"])
def test_g10_knn_baseline(tag):
    """The KNN baseline model (knn.py:8-21): one eval group and the two train groups."""
    fx = load("g10_knn")
    L = fx.dim["L"]
    i = {k[len(tag) + 1:]: v for k, v in fx.ins.items() if k.startswith(tag + "/")}
    y = O.knn_forward((i["p_x"], i["p_a"], i["p_c"]), [(i["o_x"], i["o_a"], i["o_c"])])
    assert y.shape == fx.outs[tag + "/y"].shape
    assert torch.allclose(y, fx.outs[tag + "/y"], atol=1e-5, rtol=1e-6)
    pos = tuple(i["train/" + k][:, :L] for k in ("o_x", "o_a", "o_c"))
    neg = tuple(i["train/" + k][:, L:] for k in ("o_x", "o_a", "o_c"))
    yt = O.knn_forward((i["train/p_x"], i["train/p_a"], i["train/p_c"]), [pos, neg])
    assert torch.allclose(yt, fx.outs[tag + "/train/y"], atol=1e-5, rtol=1e-6)


@pytest.mark.parametrize("name", ["g8_ranking", "g11_ranking_c2dims"])
def test_g8_ranking_after_training(name):
    fx = load(name)
    cfg = oracle_config(fx.cfg)
    profile, targets = _eval_inputs(fx.ins)
    y = O.carca_forward(fx.params, cfg, profile, targets, training=False)
    assert torch.allclose(y, fx.outs["y"], atol=ATOL, rtol=0)
    assert int(fx.outs["ties"]) == 0
    assert torch.equal(O.positive_rank(y), fx.outs["rank"])
    hr, ndcg = O.hr_ndcg_sums(y, 10)
    assert hr == float(fx.outs["hr10"])
    assert abs(ndcg - float(fx.outs["ndcg10"])) < 1e-4


def test_fp64_oracle_floor():
    """fp64 run of the same restatement stays within 1e-5 of the fp32 fixtures."""
    fx = load("g1_d90h3")
    cfg = oracle_config(fx.cfg)
    P = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in fx.params.items()}
    (px, pa, pc), [(ox, oa, oc)] = _eval_inputs(fx.ins)
    y = O.carca_forward(P, cfg, (px, pa.double(), pc.double()), [(ox, oa.double(), oc.double())], training=False)
    assert float((y.float() - fx.outs["y"]).abs().max()) < 1e-5
