"""shared helpers for the parity tests (test infrastructure)."""
import torch


def rel_l2(a, b):
    a = a.detach().double().flatten().cpu()
    b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def make_config(resnet_type=18, classes=101, **train):
    from core.config import config
    cfg = config.clone()
    cfg.MODEL.RESNET_TYPE = resnet_type
    cfg.MODEL.CLFDIM = classes
    cfg.MODEL.INIT_WEIGHTS = False
    for k, v in train.items():
        cfg.TRAIN[k] = v
    return cfg


def oracle_to_product_state(oracle_sd):
    """oracle keys backbone.* / clf_head.* -> product keys backbone.* / clf_haed.*"""
    out = {}
    for k, v in oracle_sd.items():
        out[k.replace("clf_head.", "clf_haed.")] = v.clone()
    return out


def build_pair(resnet_type=18, classes=101, seed=0, device="cuda"):
    """(oracle fp32 CPU model, product model on `device`) with identical weights."""
    from oracle.r3d_torch import R3DOracle, init_weights_
    from models.model import create_model
    oracle = init_weights_(R3DOracle(resnet_type, classes), seed)
    cfg = make_config(resnet_type, classes)
    prod = create_model(cfg, is_train=True)
    missing, unexpected = prod.load_state_dict(oracle_to_product_state(oracle.state_dict()), strict=True)
    assert not missing and not unexpected
    return oracle, prod.to(device), cfg
