"""Parameter containers named and shaped exactly like the reference's module tree, so that `state_dict()`s
interchange with volta checkpoints.  They hold weights only -- the arithmetic runs in the
HIP engine (volta_amd/engine.py) -- and are generated from small declarative specs instead of one class
per layer.  Reference sites: volta/embeddings.py:39-53,127-160,184-238,304-334,401-431 (embeddings),
volta/encoders.py:163-218,361-396,452-484,504-539 (gated sub-layers), :596-637 (poolers), :643-764 (heads).
"""
import copy
import math

import torch
from torch import nn


class Holder(nn.Module):
    """A node of the parameter tree; never executed."""

    def forward(self, *a, **k):
        raise RuntimeError("volta_amd parameter holders are not callable: run the model through "
                           "BertForVLPreTraining / BertModel, which dispatch to the HIP engine")


class LayerNormParams(Holder):
    def __init__(self, n):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))
        self.bias = nn.Parameter(torch.zeros(n))
        self.variance_epsilon = 1e-12


class LinearParams(Holder):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in))
        self.bias = nn.Parameter(torch.zeros(n_out))
        self.in_features, self.out_features = n_in, n_out


class TableParams(Holder):
    def __init__(self, rows, width):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(rows, width))
        self.num_embeddings, self.embedding_dim = rows, width


def _make(kind, *dims):
    return {"ln": LayerNormParams, "lin": LinearParams, "tab": TableParams}[kind](*dims)


def build(spec, cfg):
    """spec: [(attribute name, kind, dims...)] with dims given as config attribute names or ints."""
    node = Holder()
    for name, kind, *dims in spec:
        vals = [getattr(cfg, d) if isinstance(d, str) else d for d in dims]
        node.add_module(name, _make(kind, *vals))
    return node


_TEXT = [("word_embeddings", "tab", "vocab_size", "hidden_size"),
         ("position_embeddings", "tab", "max_position_embeddings", "hidden_size"),
         ("token_type_embeddings", "tab", "type_vocab_size", "hidden_size"),
         ("LayerNorm", "ln", "hidden_size")]

EMBEDDING_SPECS = {
    # dual-stream: text table set under bert.embeddings, image projection under bert.v_embeddings
    "text": _TEXT,
    "vilbert": [("image_embeddings", "lin", "v_feature_size", "v_hidden_size"),
                ("image_location_embeddings", "lin", "num_locs", "v_hidden_size"),
                ("LayerNorm", "ln", "v_hidden_size")],
    "lxmert": [("image_embeddings", "lin", "v_feature_size", "v_hidden_size"),
               ("image_location_embeddings", "lin", "num_locs", "v_hidden_size"),
               ("ImgLayerNorm", "ln", "v_hidden_size"), ("LocLayerNorm", "ln", "v_hidden_size")],
    # single-stream: everything under bert.embeddings
    "uniter": _TEXT + [("image_embeddings", "lin", "v_feature_size", "v_hidden_size"),
                       ("image_location_embeddings", "lin", "num_locs", "v_hidden_size"),
                       ("image_layer_norm", "ln", "hidden_size"), ("image_location_layer_norm", "ln", "hidden_size"),
                       ("v_LayerNorm", "ln", "hidden_size")],
    "visualbert": _TEXT + [("projection", "lin", "v_feature_size", "hidden_size"),
                           ("token_type_embeddings_visual", "tab", "type_vocab_size", "hidden_size"),
                           ("position_embeddings_visual", "tab", "max_position_embeddings", "hidden_size")],
}
DUAL = ("vilbert", "lxmert")
SHARED = ("vl-bert", "visualbert", "uniter")


def build_vlbert_embeddings(cfg):
    """VL-BERT's container has a Sequential child (`obj_downsample.1` is the 2F -> Hv linear)."""
    node = Holder()
    seq = Holder()
    seq.add_module("1", LinearParams(2 * cfg.v_feature_size, cfg.v_hidden_size))
    node.add_module("obj_downsample", seq)
    node.add_module("object_linguistic_embeddings", TableParams(1, cfg.hidden_size))
    if cfg.visual_target_weights.get("6", 0) > 0:
        node.add_module("object_mask_word_embedding", TableParams(1, cfg.hidden_size))
    node.add_module("object_mask_visual_embedding", TableParams(1, cfg.v_feature_size))
    node.add_module("end_embedding", TableParams(1, cfg.hidden_size))
    for name, kind, *dims in _TEXT[:3]:
        node.add_module(name, _make(kind, *[getattr(cfg, d) for d in dims]))
    if cfg.v_hidden_size != cfg.hidden_size:
        node.add_module("visual_1x1_text", LinearParams(cfg.v_hidden_size, cfg.hidden_size))
        node.add_module("visual_1x1_object", LinearParams(cfg.v_hidden_size, cfg.hidden_size))
    node.add_module("visual_ln_text", LayerNormParams(cfg.hidden_size))
    node.add_module("visual_ln_object", LayerNormParams(cfg.hidden_size))
    node.add_module("LayerNorm", LayerNormParams(cfg.hidden_size))
    return node


def build_attention_sublayer(cfg, n):
    """`attention_self` + `attention_output` holders of sub-layer n with the reference's gating, aliasing
    (shared sub-layers register the text modules under the v_* names too) and error behaviour."""
    H = cfg.sublayer2attn_hidden_size.get(str(n), cfg.hidden_size)
    nh = cfg.sublayer2num_attention_heads.get(str(n), cfg.num_attention_heads)
    Hv = cfg.sublayer2v_attn_hidden_size.get(str(n), cfg.v_hidden_size)
    vnh = cfg.sublayer2v_num_attention_heads.get(str(n), cfg.v_num_attention_heads)
    if H % nh != 0:
        raise ValueError("The text hidden size (%d) is not a multiple of the number of attention heads (%d)" % (H, nh))
    if Hv % vnh != 0:
        raise ValueError("The vision hidden size (%d) is not a multiple of the number of attention heads (%d)" % (Hv, nh))
    tt, tv = n in cfg.tt_attn_sublayers, n in cfg.tv_attn_sublayers
    vt, vv = n in cfg.vt_attn_sublayers, n in cfg.vv_attn_sublayers
    has_t, has_v, shared = tt or tv, vv or vt, n in cfg.shared_sublayers
    if tv or vt:
        assert H == Hv, "hidden_size != v_hidden_size"
        assert nh == vnh, "num_attention_heads != v_num_attention_heads"
    if n in cfg.single_ln_sublayers:
        assert has_t and has_v and shared, "Missing language, vision or sharing"
    sa, so = Holder(), Holder()
    if has_t:
        for k in ("query", "key", "value"):
            sa.add_module(k, LinearParams(cfg.hidden_size, H))
        so.add_module("dense", LinearParams(H, cfg.hidden_size))
        so.add_module("LayerNorm", LayerNormParams(cfg.hidden_size))
    if has_t and has_v and shared:
        assert H == Hv and cfg.hidden_size == cfg.v_hidden_size, "hidden_size != v_hidden_size"
        for k in ("query", "key", "value"):
            sa.add_module("v_" + k, getattr(sa, k))
        so.add_module("v_dense", so.dense)
        so.add_module("v_LayerNorm", so.LayerNorm)
    elif has_v:
        for k in ("query", "key", "value"):
            sa.add_module("v_" + k, LinearParams(cfg.v_hidden_size, Hv))
        so.add_module("v_dense", LinearParams(Hv, cfg.v_hidden_size))
        so.add_module("v_LayerNorm", LayerNormParams(Hv))
    node = Holder()
    node.add_module("attention_self", sa)
    node.add_module("attention_output", so)
    return node


def build_ffn_sublayer(cfg, n):
    I = cfg.sublayer2intermediate_size.get(str(n), cfg.intermediate_size)
    Iv = cfg.sublayer2v_intermediate_size.get(str(n), cfg.v_intermediate_size)
    has_t, has_v, shared = n in cfg.t_ff_sublayers, n in cfg.v_ff_sublayers, n in cfg.shared_sublayers
    if n in cfg.single_ln_sublayers:
        assert has_t and has_v and shared, "Missing language, vision or sharing"
    inter, out = Holder(), Holder()
    if has_t:
        inter.add_module("dense", LinearParams(cfg.hidden_size, I))
        out.add_module("dense", LinearParams(I, cfg.hidden_size))
        out.add_module("LayerNorm", LayerNormParams(cfg.hidden_size))
    if has_t and has_v and shared:
        assert cfg.hidden_size == cfg.v_hidden_size, "hidden_size != v_hidden_size"
        assert I == Iv, "intermediate_size != v_intermediate_size"
        inter.add_module("v_dense", inter.dense)
        out.add_module("v_dense", out.dense)
        out.add_module("v_LayerNorm", out.LayerNorm)
    elif has_v:
        inter.add_module("v_dense", LinearParams(cfg.v_hidden_size, Iv))
        out.add_module("v_dense", LinearParams(Iv, cfg.v_hidden_size))
        out.add_module("v_LayerNorm", LayerNormParams(cfg.v_hidden_size))
    node = Holder()
    node.add_module("intermediate", inter)
    node.add_module("output", out)
    return node


def sublayer_schedule(cfg):
    """[(n, 'attn' | 'ff')] in execution order, with the reference's wiring assertions (encoders.py:830-843)."""
    attn = set(cfg.tt_attn_sublayers + cfg.tv_attn_sublayers + cfg.vt_attn_sublayers + cfg.vv_attn_sublayers)
    ff = set(cfg.t_ff_sublayers + cfg.v_ff_sublayers)
    assert not (attn & ff), "Overlapping attn-ff sublayer numbers"
    nums = sorted(attn | ff)
    assert nums and nums[0] == 0 and nums[-1] == len(nums) - 1, "Non contiguous sublayer numbers"
    return [(n, "attn" if n in attn else "ff") for n in nums]


def init_bert_(module, std):
    """N(0, std) for tables and linear weights, zero biases, LayerNorm (1, 0) (encoders.py:904-915)."""
    for m in module.modules():
        if isinstance(m, (LinearParams, TableParams)):
            m.weight.data.normal_(mean=0.0, std=std)
        if isinstance(m, LinearParams):
            m.bias.data.zero_()
        if isinstance(m, LayerNormParams):
            m.weight.data.fill_(1.0)
            m.bias.data.zero_()


def init_heads_(module):
    """Pre-training heads: xavier-uniform linears, N(0, 0.02) tables (encoders.py:753-764)."""
    for m in module.modules():
        if isinstance(m, TableParams):
            m.weight.data.normal_(mean=0.0, std=0.02)
        elif isinstance(m, LinearParams):
            bound = math.sqrt(6.0 / (m.in_features + m.out_features))
            m.weight.data.uniform_(-bound, bound)
            m.bias.data.zero_()
        elif isinstance(m, LayerNormParams):
            m.weight.data.fill_(1.0)
            m.bias.data.zero_()


def torch_default_init_(module):
    """What torch's own constructors leave in nn.Embedding (N(0, 1)) and nn.Linear (weight and bias uniform in
    +-1 / sqrt(fan_in): kaiming_uniform_(a = sqrt 5))."""
    for m in module.modules():
        if isinstance(m, TableParams):
            m.weight.data.normal_(mean=0.0, std=1.0)
        elif isinstance(m, LinearParams):
            bound = 1.0 / math.sqrt(m.in_features)
            m.weight.data.uniform_(-bound, bound)
            m.bias.data.uniform_(-bound, bound)
        elif isinstance(m, LayerNormParams):
            m.weight.data.fill_(1.0)
            m.bias.data.zero_()


def special_init_embeddings_(emb, kind, cfg):
    """Family-specific initial values.  The reference builds the SHARED (single-stream) embedding modules AFTER
    `BertModel.apply(init_weights)` has run (encoders.py:949-952) and never re-applies it, so their tensors keep torch's
    constructor defaults -- tables N(0, 1) with a zero padding row, linears uniform +-1/sqrt(fan_in) incl. the bias --
    except what their own constructors set (embeddings.py:229-238, 328-334, 428-431).  Kept: the oracle is the truth."""
    if kind not in SHARED:
        return
    torch_default_init_(emb)
    emb.word_embeddings.weight.data[0].zero_()              # padding_idx = 0 (embeddings.py:207,313,410)
    if kind == "visualbert":
        emb.token_type_embeddings_visual.weight = nn.Parameter(copy.deepcopy(emb.token_type_embeddings.weight.data))
        emb.position_embeddings_visual.weight = nn.Parameter(copy.deepcopy(emb.position_embeddings.weight.data))
    elif kind == "uniter":
        emb.v_LayerNorm.weight = nn.Parameter(copy.deepcopy(emb.LayerNorm.weight.data))
        emb.v_LayerNorm.bias = nn.Parameter(copy.deepcopy(emb.LayerNorm.bias.data))
    elif kind == "vl-bert":
        lin = emb.obj_downsample._modules["1"]
        bound = math.sqrt(6.0 / (lin.in_features + lin.out_features))
        lin.weight.data.uniform_(-bound, bound)              # xavier; the bias keeps the constructor default
        emb.object_mask_visual_embedding.weight.data.fill_(0.0)
        emb.object_linguistic_embeddings.weight.data.normal_(mean=0.0, std=cfg.initializer_range)
        if hasattr(emb, "object_mask_word_embedding"):
            emb.object_mask_word_embedding.weight.data.normal_(mean=0.0, std=cfg.initializer_range)
        emb.visual_ln_text.weight.data.fill_(0.0)
        emb.visual_ln_object.weight.data.fill_(0.0)


def init_tied_decoder_(weight):
    """The LM decoder is an nn.Linear(hidden, vocab) whose weight IS the word-embedding table (encoders.py:688-691), and the
    heads' init (encoders.py:753-764) xavier-initialises every nn.Linear: the table ends up uniform in +-sqrt(6 / (V + H))."""
    bound = math.sqrt(6.0 / (weight.shape[0] + weight.shape[1]))
    weight.data.uniform_(-bound, bound)
