"""Drop-in model classes: same constructor / forward signatures, parameter names and error behaviour as the
reference's `BertModel` / `BertForVLPreTraining` (volta/encoders.py:918-1114, volta/utils.py:250-360), with the
whole forward + backward dispatched to the HIP engine (volta_amd/engine.py).  There is no eager / CPU
fallback: without a GPU and libvolta_hip.so the forward raises.

Differences a caller can observe, all deliberate:
  * the three losses come back from one fused step; the two host synchronisations of the reference
    (encoders.py:1089, 1111) do not happen -- `img_loss` is numerically 0 when no region is labelled;
  * activations are bf16 with fp32 accumulation / statistics (north_star), parameters stay fp32;
  * dropout uses a counter-based Philox stream seeded from `torch.initial_seed()` (or `set_dropout_seed`).
"""
import logging
import os

import torch
from torch import nn

from . import modules as M
from .config import BertConfig

logger = logging.getLogger(__name__)


class ArenaParameters:
    """What `parameters()` of a volta_amd root model returns: nn.Module's parameter iterator, carrying the model it walks.
    `clip_grad_norm_(model.parameters(), max_norm)` -- the reference's call, train_concap.py:307 -- reads `vk_model` and takes the whole
    gradient arena in one pass instead of walking 600 tensors (an explicit hook: no generator-frame introspection)."""

    __slots__ = ("vk_model", "started", "_it")

    def __init__(self, model, it):
        self.vk_model, self.started, self._it = model, False, it

    def __iter__(self):
        return self

    def __next__(self):
        self.started = True
        return next(self._it)


class PreTrainedModel(nn.Module):
    """Base with the helpers the reference's driver touches (volta/utils.py:250-360)."""

    def parameters(self, recurse=True):
        it = super().parameters(recurse)
        return ArenaParameters(self, it) if (recurse and getattr(self, "_vk_is_model", False)) else it

    config_class = BertConfig
    base_model_prefix = "bert"

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        self.config = config

    def init_weights(self, module):
        M.init_bert_(module, self.config.initializer_range)

    def _get_resized_embeddings(self, old_embeddings, new_num_tokens=None):
        if new_num_tokens is None:
            return old_embeddings
        old_num_tokens, dim = old_embeddings.weight.size()
        if old_num_tokens == new_num_tokens:
            return old_embeddings
        new = M.TableParams(new_num_tokens, dim).to(old_embeddings.weight.device)
        new.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        n = min(old_num_tokens, new_num_tokens)
        new.weight.data[:n, :] = old_embeddings.weight.data[:n, :]
        self._invalidate_engine()
        return new

    def _tie_or_clone_weights(self, first_module, second_module):
        first_module.weight = second_module.weight

    def _invalidate_engine(self):
        root = getattr(self, "_root", None) or self
        root.__dict__["_arena"] = None
        root.__dict__["_engines"] = {}

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)       # .cuda() / .to() / .float(): parameters are re-allocated
        self._invalidate_engine()
        return out

    def save_pretrained(self, save_directory):
        assert os.path.isdir(save_directory), "Saving path should be a directory where the model and configuration can be saved"
        model_to_save = self.module if hasattr(self, "module") else self
        with open(os.path.join(save_directory, "config.json"), "w", encoding="utf-8") as f:
            f.write(model_to_save.config.to_json_string())
        torch.save({k: v.detach().cpu().clone() for k, v in model_to_save.state_dict().items()},
                   os.path.join(save_directory, "pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, state_dict=None, **kwargs):
        """Local directory / file branch of the reference loader (volta/utils.py:404-550): old `gamma` / `beta` names,
        the HuggingFace-BERT key renumbering of `from_hf=True` (`bert_layer2attn_sublayer` / `bert_layer2ff_sublayer`,
        utils.py:475-498), base-model checkpoints without the `bert.` prefix (utils.py:513-521), tied decoder, eval mode,
        `output_loading_info`.  Named archives would need the network (utils.py:417-418) and are refused."""
        kwargs.pop("default_gpu", None)
        kwargs.pop("cache_dir", None)
        from_hf = kwargs.pop("from_hf", False)
        output_loading_info = kwargs.pop("output_loading_info", False)
        if kwargs.pop("from_tf", False):
            raise NotImplementedError("TensorFlow checkpoints (volta/utils.py:424-426) are not supported")
        assert config is not None
        path = pretrained_model_name_or_path
        if path is not None and os.path.isdir(path):
            path = os.path.join(path, "pytorch_model.bin")
        if state_dict is None:
            if path is None or not os.path.isfile(path):
                raise EnvironmentError("'%s' is not a local checkpoint; downloading named archives is not available "
                                       "offline (volta/utils.py:417-418)" % pretrained_model_name_or_path)
            state_dict = torch.load(path, map_location="cpu")
        model = cls(config, *model_args, **kwargs)
        sd = {}
        for k, v in state_dict.items():                       # old LayerNorm parameter names (utils.py:461-472)
            if "gamma" in k:
                k = k.replace("gamma", "weight")
            if "beta" in k:
                k = k.replace("beta", "bias")
            sd[k[len("module."):] if k.startswith("module.") else k] = v     # (DDP-wrapped saves; train_utils.py:326-330)
        if from_hf:                                           # one BERT layer -> one attention + one FF sub-layer
            a2s, f2s = config.bert_layer2attn_sublayer, config.bert_layer2ff_sublayer
            renames = []
            for k in sd:
                if ".layer." not in k:
                    continue
                num = int(k.split(".layer.")[-1].split(".")[0])
                nk = None
                if ".attention." in k:
                    nk = k.replace(".layer.%d.attention." % num, ".layer.%d.attention_" % a2s.get(str(num), num))
                elif ".intermediate." in k:
                    nk = k.replace(".layer.%d.intermediate." % num, ".layer.%d.intermediate." % f2s.get(str(num), num))
                elif ".output." in k:
                    nk = k.replace(".layer.%d.output." % num, ".layer.%d.output." % f2s.get(str(num), num))
                if nk:
                    renames.append((k, nk, num))
            for k, nk, _ in sorted(renames, key=lambda x: x[2], reverse=True):      # highest layer first: no clobbering
                sd[nk] = sd.pop(k)
        # derived model <- base checkpoint, or base model <- derived checkpoint (utils.py:511-521)
        prefix = cls.base_model_prefix + "."
        has_prefix = any(k.startswith(prefix) for k in sd)
        target, rel = model, ""
        if hasattr(model, cls.base_model_prefix) and not has_prefix:
            target = getattr(model, cls.base_model_prefix)
        elif not hasattr(model, cls.base_model_prefix) and has_prefix:
            rel = prefix
        if rel:
            sd = {k[len(rel):]: v for k, v in sd.items() if k.startswith(rel)}
        res = target.load_state_dict(sd, strict=False)
        missing, unexpected = list(res.missing_keys), list(res.unexpected_keys)
        if missing:
            print("Weights of {} not initialized from pretrained model: {}".format(cls.__name__, missing))
        if unexpected:
            print("Weights from pretrained model not used in {}: {}".format(cls.__name__, unexpected))
        if hasattr(model, "tie_weights"):
            model.tie_weights()
        model.eval()
        if output_loading_info:
            return model, {"missing_keys": missing, "unexpected_keys": unexpected, "error_msgs": []}
        return model


class BertEncoder(M.Holder):
    def __init__(self, config):
        super().__init__()
        self.num2type = {}
        layers = []
        for n, typ in M.sublayer_schedule(config):
            self.num2type[n] = typ
            layers.append(M.build_attention_sublayer(config, n) if typ == "attn" else M.build_ffn_sublayer(config, n))
        self.layer = nn.ModuleList(layers)


class BertModel(PreTrainedModel):
    """Embeddings + gated encoder + poolers (volta/encoders.py:918-1017)."""

    def __init__(self, config):
        super().__init__(config)
        kind = config.image_embeddings
        self.shared_embeddings = kind in M.SHARED
        if config.model != "bert":
            raise NotImplementedError("RoBERTa embeddings are out of scope (SURVEY.md 2.1 #3)")
        if kind in M.DUAL:
            self.embeddings = M.build(M.EMBEDDING_SPECS["text"], config)
            self.v_embeddings = M.build(M.EMBEDDING_SPECS[kind], config)
        elif kind == "vl-bert":
            self.embeddings = M.build_vlbert_embeddings(config)
        elif kind in M.SHARED:
            self.embeddings = M.build(M.EMBEDDING_SPECS[kind], config)
        else:
            raise ValueError("unknown image_embeddings %r" % kind)
        self.encoder = BertEncoder(config)
        self.fusion_method = fm = config.fusion_method
        if fm not in ("sum", "mul", "text", "vl-bert_vqa", "none"):
            raise ValueError("Invalid fusion method: %s" % fm)
        # encoders.py:936-947: "none" has no poolers, "text" / "vl-bert_vqa" (VLBertTextPooler, :610-623) no vision pooler
        if fm != "none":
            self.t_pooler = M.Holder()
            self.t_pooler.add_module("dense", M.LinearParams(config.hidden_size, config.pooler_size))
        if fm in ("sum", "mul"):
            assert config.pooler_size == config.v_pooler_size, "pooler_size != v_pooler_size"
            self.v_pooler = M.Holder()
            self.v_pooler.add_module("dense", M.LinearParams(config.v_hidden_size, config.v_pooler_size))
        M.init_bert_(self, config.initializer_range)
        M.special_init_embeddings_(self.embeddings, kind, config)

    def forward(self, input_txt, input_imgs, image_loc, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, output_all_encoded_layers=False, output_all_attention_masks=False):
        root = self.__dict__.get("_root")
        if root is None:
            raise NotImplementedError("BertModel runs as part of BertForVLPreTraining (the HIP plan includes the heads)")
        return root.encode(input_txt, input_imgs, image_loc, token_type_ids, attention_mask, image_attention_mask,
                           output_all_encoded_layers=output_all_encoded_layers, output_all_attention_masks=output_all_attention_masks)


class BertPreTrainingHeads(M.Holder):
    """cls.predictions / cls.bi_seq_relationship / cls.imagePredictions (volta/encoders.py:643-764)."""

    def __init__(self, config, bert_model_embedding_weights):
        super().__init__()
        H, Hv = config.hidden_size, config.v_hidden_size
        pred, tr = M.Holder(), M.Holder()
        tr.add_module("dense", M.LinearParams(H, H))
        tr.add_module("LayerNorm", M.LayerNormParams(H))
        pred.add_module("transform", tr)
        dec = M.Holder()
        dec.weight = bert_model_embedding_weights            # tied (encoders.py:691)
        pred.add_module("decoder", dec)
        pred.bias = nn.Parameter(torch.zeros(bert_model_embedding_weights.size(0)))
        self.add_module("predictions", pred)
        if config.fusion_method not in ("none", "vl-bert_vqa"):            # encoders.py:744-747
            self.add_module("bi_seq_relationship", M.LinearParams(config.pooler_size, 2))
        img, itr = M.Holder(), M.Holder()
        itr.add_module("dense", M.LinearParams(Hv, Hv))
        if config.image_head_ln:
            itr.add_module("LayerNorm", M.LayerNormParams(Hv))
        img.add_module("transform", itr)
        widths = {"0": 1601, "1": 2048, "2": 2048, "3": 1600, "4": 400, "5": 2048, "6": 1601}   # losses.py:129-137
        img.add_module("decoder_dict", nn.ModuleDict({ix: M.LinearParams(Hv, n) for ix, n in widths.items()
                                                       if config.visual_target_weights.get(ix, 0) > 0}))
        self.add_module("imagePredictions", img)
        self.fusion_method = config.fusion_method
        M.init_heads_(self)
        M.init_tied_decoder_(bert_model_embedding_weights)


class _PretrainStep(torch.autograd.Function):
    """One autograd node for the whole model: forward = engine forward list, backward = engine backward list.
    Parameter gradients are written straight into the flat gradient arena and attached as `.grad` views."""

    @staticmethod
    def forward(ctx, model, anchor, tensors):
        losses = model._engine_forward(tensors)
        ctx.model = model
        return losses[0:1].clone(), losses[1:2].clone(), losses[2:3].clone()

    @staticmethod
    def backward(ctx, g_lm, g_img, g_nsp):
        ctx.model._engine_backward(g_lm, g_img, g_nsp)
        return None, None, None


class BertForVLPreTraining(PreTrainedModel):
    """BERT model with multimodal pre-training heads (volta/encoders.py:1020-1114)."""

    def __init__(self, config):
        super().__init__(config)
        self.bert = BertModel(config)
        self.cls = BertPreTrainingHeads(config, self.bert.embeddings.word_embeddings.weight)
        self.visual_target_weights = config.visual_target_weights
        logger.info("model's visual targets are %s", [ix for ix, w in config.visual_target_weights.items() if w > 0])
        self.add_global_imgfeat = int(config.add_global_imgfeat is not None)
        self.tie_weights()
        self.__dict__["_arena"] = None
        self.__dict__["_engines"] = {}
        self.__dict__["_step"] = 0
        self.__dict__["_seed_base"] = None
        self.__dict__["_last"] = None
        self.__dict__["_ddp"] = None
        self.bert.__dict__["_root"] = self
        for mod in self.modules():
            if mod is not self and isinstance(mod, PreTrainedModel):
                mod.__dict__["_root"] = self
        for p in self.parameters():            # lets an optimizer find (and materialize) its model before the first forward,
            p._vk_owner = self                 # e.g. AdamW.load_state_dict() in the reference's resume() order

    def tie_weights(self):
        self._tie_or_clone_weights(self.cls.predictions.decoder, self.bert.embeddings.word_embeddings)

    # ------------------------------------------------------------------ engine plumbing
    def set_dropout_seed(self, seed):
        self.__dict__["_seed_base"] = int(seed)
        self.__dict__["_step"] = 0

    _vk_is_model = True      # lets clip_grad_norm_(model.parameters()) recognise the whole arena without walking it

    def materialize(self, device=None):
        """Build (or rebuild) the flat parameter arenas on `device`; parameters become views of them."""
        from .engine import ParamArena
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("volta_amd runs on an MI355X only: move the model to the GPU first (model.cuda()); "
                               "there is no CPU execution path")
        arena = self.__dict__.get("_arena")
        if arena is None or arena.device != dev or not arena.intact():
            self.__dict__["_arena"] = ParamArena(self, dev)
            self.__dict__["_engines"] = {}
            for p in self.parameters():
                p._vk_owner = self
        return self.__dict__["_arena"]

    def _engine(self, B, T, Rv, train):
        from .engine import StepEngine
        arena = self.materialize()
        fp8 = bool(self.__dict__.get("_fp8", False))
        task_id = self.__dict__.get("_cur_task")
        maps = bool(self.__dict__.get("_want_attn_maps", False))
        key = (B, T, Rv, bool(train), fp8, task_id, maps)
        eng = self._engines.get(key)
        if eng is None:
            for k in [k for k in self._engines if k[3] == key[3] and k[5] == task_id and k[6] == maps]:      # one plan per mode (and task head) keeps memory bounded
                del self._engines[k]
            task = (task_id, self.task_cfg[task_id]) if task_id is not None else None
            eng = StepEngine(self.config, arena, B, T, Rv, train, heads=getattr(self, "_heads_mode", "pretrain"), fp8=fp8, task=task,
                             task_dropout=self.__dict__.get("_task_dropout", 0.1), attn_maps=maps)
            self._engines[key] = eng
        return eng

    def set_projection_dtype(self, dtype):
        """"bf16" (default) or "fp8": run the forward Q|K|V / FFN projections of the encoder on the e4m3 MFMA path (csrc/fp8.hip)."""
        if dtype not in ("bf16", "fp8"):
            raise ValueError("projection dtype %r (bf16 | fp8)" % (dtype,))
        self.__dict__["_fp8"] = dtype == "fp8"

    def _prep_inputs(self, input_ids, image_feat, image_loc, token_type_ids, attention_mask, image_attention_mask,
                     masked_lm_labels, image_label, image_cls, next_sentence_label, obj_labels=None, obj_confs=None,
                     attr_labels=None, attr_confs=None):
        dev = next(self.parameters()).device
        B, T = input_ids.shape
        Rv = image_feat.shape[1]
        i64 = dict(device=dev, dtype=torch.int64)
        f32 = dict(device=dev, dtype=torch.float32)
        if attention_mask is None:
            attention_mask = torch.ones(B, T, **i64)
        if token_type_ids is None:
            token_type_ids = torch.zeros(B, T, **i64)
        if image_attention_mask is None:
            image_attention_mask = torch.ones(B, Rv, **i64)
        R = Rv - self.add_global_imgfeat
        t = dict(input_ids=input_ids.to(**i64).contiguous(), token_type_ids=token_type_ids.to(**i64).contiguous(),
                 attention_mask=attention_mask.to(**i64).contiguous(), image_attention_mask=image_attention_mask.to(**i64).contiguous(),
                 image_feat=image_feat.to(**f32).contiguous(), image_loc=image_loc.to(**f32).contiguous())
        assert t["image_feat"].shape == (B, Rv, self.config.v_feature_size), "image_feat must be [B, regions, v_feature_size]"
        assert t["image_loc"].shape == (B, Rv, self.config.num_locs), "image_loc must be [B, regions, num_locs]"
        if masked_lm_labels is not None:
            t["masked_lm_labels"] = masked_lm_labels.to(**i64).contiguous()
            t["image_label"] = image_label.to(**i64).contiguous()
            assert t["masked_lm_labels"].shape == (B, T) and t["image_label"].shape == (B, R)
            # what each configured visual target reads (volta/losses.py); the reference silently drops a target whose inputs are
            # missing -- here that is an error, the plan was compiled for the configured targets
            need = set()
            for ix, w in self.config.visual_target_weights.items():
                if w > 0:
                    need |= {"0": {"image_cls"}, "3": {"obj_labels", "obj_confs"}, "4": {"attr_labels", "attr_confs"}, "6": {"obj_labels"}}.get(ix, set())
            given = dict(image_cls=image_cls, obj_labels=obj_labels, obj_confs=obj_confs, attr_labels=attr_labels, attr_confs=attr_confs)
            for name in sorted(need):
                if given[name] is None:
                    raise ValueError("visual target weights %r need `%s`" % (self.config.visual_target_weights, name))
                x = given[name].to(**(i64 if name.endswith("labels") else f32)).contiguous()
                assert x.shape[:2] == (B, R), "%s must be [B, regions%s]" % (name, ", 1601" if name == "image_cls" else "")
                t[name] = x
            if "image_cls" in t:
                assert t["image_cls"].shape == (B, R, 1601)
            if self.bert.fusion_method in ("mul", "sum", "text"):
                if next_sentence_label is None:
                    raise ValueError("fusion method %r has an ITM head: next_sentence_label is required" % self.bert.fusion_method)
                t["next_sentence_label"] = next_sentence_label.to(**i64).contiguous()
                assert t["next_sentence_label"].numel() == B
            # ids / labels are range-clamped inside the kernels: no host synchronisation on the hot path
        return t, B, T, Rv

    def _engine_forward(self, tensors):
        B, T = tensors["input_ids"].shape
        Rv = tensors["image_feat"].shape[1]
        eng = self._engine(B, T, Rv, self.training)
        eng.arena.refresh_shadow()
        if eng.fp8:
            eng.arena.sync_optimizer()         # the re-quantisation reads the fp32 masters
            eng.arena.refresh_fp8()
        eng.bind_inputs(tensors)
        if self._seed_base is None:
            self.__dict__["_seed_base"] = int(torch.initial_seed())
        eng.prepare_step(self._seed_base + self._step)
        self.__dict__["_step"] += 1
        eng.run_forward()
        self.__dict__["_last"] = (eng, tensors)      # keeps the step's input tensors alive until backward
        return getattr(eng, "losses", None)

    def _engine_backward(self, g_lm, g_img, g_nsp):
        eng, tensors = self._last
        state = self._backward_begin(eng)
        torch.cat([g_lm.reshape(1), g_img.reshape(1), g_nsp.reshape(1)], out=eng.gout)      # one launch, no temporary
        self._backward_run(eng, state)

    def _backward_begin(self, eng):
        """Gradient-accumulation bookkeeping shared by the pre-training and the task models.  Parameters of torch-side head
        modules (`_torch_param_prefixes`) get their gradients from autograd (redirected into the arena by a hook)."""
        arena = eng.arena
        from .optimization import flush_clip
        flush_clip(arena)                   # a clip coefficient no optimizer step has consumed applies to the gradients it was computed for
        skip = getattr(self, "_torch_param_prefixes", ())
        # frozen parameters (requires_grad False: volta/train_utils.py:250-255) get no .grad, as under autograd; the
        # optimizer and clip_grad_norm_ then leave their arena chunks alone
        unused = eng.unused_params          # e.g. the VQA text pooler in pre-training: no launch reads it, .grad stays None
        pairs = [((n, p), g) for (n, p), g in zip(arena.param_list(), arena.grad_views())
                 if p.requires_grad and not (skip and n.startswith(skip)) and n not in unused]
        params, gviews = [x[0] for x in pairs], [x[1] for x in pairs]
        n_have = sum(p.grad is not None for _, p in params)
        accumulate = n_have == len(params)
        if n_have and not accumulate:
            raise RuntimeError("volta_amd: some parameters carry a .grad and some do not; zero_grad() all of them")
        old = None
        if accumulate:
            if not all(p.grad is g or p.grad.data_ptr() == g.data_ptr() for (_, p), g in zip(params, gviews)):
                raise RuntimeError("volta_amd: .grad tensors were replaced by foreign tensors; call zero_grad(set_to_none=True)")
            old = arena.grad.clone()
        return params, gviews, accumulate, old

    def _backward_run(self, eng, state):
        params, gviews, accumulate, old = state
        arena = eng.arena
        skip = getattr(self, "_torch_param_prefixes", ())
        if skip:            # the engine list zero-fills / overwrites only what it owns; torch-side head gradients are already in the arena
            keep = [(g, g.clone()) for (n, _), g in zip(arena.param_list(), arena.grad_views()) if n.startswith(skip)] if accumulate else []
        ddp = self.__dict__.get("_ddp")
        if ddp is not None:
            ddp.run_backward(eng)
        else:
            eng.bwd.run()
        if accumulate:
            from . import _lib as L
            L.check(L.lib.vk_axpy_f32(L.ptr(arena.grad), L.ptr(old), 1.0, arena.total, L.stream_ptr()))
            if skip:
                for g, saved in keep:
                    g.copy_(saved)
        else:
            for (_, p), g in zip(params, gviews):
                p.grad = g

    # ------------------------------------------------------------------ public API
    def forward(self, input_ids, image_feat, image_loc, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, masked_lm_labels=None, image_label=None, image_cls=None, obj_labels=None,
                obj_confs=None, attr_labels=None, attr_confs=None, image_attrs=None, next_sentence_label=None,
                output_all_attention_masks=False):
        # (with losses the reference returns the three losses only, encoders.py:1111-1112: the attention maps are not part of that result)
        if masked_lm_labels is None and next_sentence_label is None:
            # encoders.py:1113-1114: without text labels and without a masked region every loss is zero and the reference returns the heads' scores
            return self._scores(input_ids, image_feat, image_loc, token_type_ids, attention_mask, image_attention_mask, image_label,
                                output_all_attention_masks)
        if masked_lm_labels is None or image_label is None:
            raise NotImplementedError("the pre-training step needs masked_lm_labels and image_label")
        tensors, B, T, Rv = self._prep_inputs(input_ids, image_feat, image_loc, token_type_ids, attention_mask,
                                              image_attention_mask, masked_lm_labels, image_label, image_cls, next_sentence_label,
                                              obj_labels, obj_confs, attr_labels, attr_confs)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            self.materialize()
            anchor = next(p for p in self.parameters() if p.requires_grad)
            return _PretrainStep.apply(self, anchor, tensors)
        losses = self._engine_forward(tensors)
        return losses[0:1].clone(), losses[1:2].clone(), losses[2:3].clone()

    def _scores(self, input_ids, image_feat, image_loc, token_type_ids, attention_mask, image_attention_mask, image_label, output_all_attention_masks=False):
        """The score-returning branch of BertForVLPreTraining.forward (volta/encoders.py:1065-1068,1113-1114): (prediction_scores_t [B,T,V],
        {target: prediction_scores_v [B,Rv,C]}, seq_relationship_score [B,2] or None, attention maps = ([], []), pooled_output or None), the heads
        applied to EVERY position.  Inference only (no gradient, the heads' dropout is the identity); the projections run on the library's GEMM /
        LayerNorm kernels through volta_amd.ops."""
        from . import _lib as L, ops
        if image_label is not None and bool((image_label == 1).any()):
            raise ValueError("masked regions without text labels: the reference returns the losses here, which need masked_lm_labels")
        with torch.no_grad():
            was_training = self.training
            self.eval()
            try:
                seq_t, seq_v, pt, pv, attn_maps = self.encode(input_ids, image_feat, image_loc, token_type_ids, attention_mask, image_attention_mask,
                                                              output_all_attention_masks=output_all_attention_masks)
            finally:
                self.train(was_training)
            eng = self._last[0]
            arena = eng.arena
            W, Pm = (lambda n: arena.view(n, "shadow")), (lambda n: arena.view(n, "master"))
            dev = seq_t.device
            B, T, H = seq_t.shape
            Rv, Hv = seq_v.shape[1], seq_v.shape[2]
            cfg = self.config

            def head(x, rows, K, pre, with_ln, decoders):
                """x bf16 [rows, K] -> transform (dense + GELU (+ LayerNorm)) -> {name: fp32 scores}"""
                h, gp = torch.empty(rows, K, device=dev, dtype=torch.bfloat16), torch.empty(rows, K, device=dev, dtype=torch.bfloat16)
                ops.gemm_grouped(L.NT, L.EPI_GELU, [ops.gemm_problem(x, W(pre + "transform.dense.weight"), h, L.NT, rows, K, K, bias=Pm(pre + "transform.dense.bias"), C2=gp)])
                if with_ln:
                    hn = torch.empty_like(h)
                    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
                    ops.ln_fwd(h, None, Pm(pre + "transform.LayerNorm.weight"), Pm(pre + "transform.LayerNorm.bias"), hn, None, mean, rstd, rows, K)
                    h = hn
                out = {}
                for name, (wt, bias, C) in decoders.items():
                    Cp = -(-C // 64) * 64
                    sc = torch.empty(rows, Cp, device=dev)
                    ops.gemm_grouped(L.NT, L.EPI_F32, [ops.gemm_problem(h, wt, sc, L.NT, rows, C, K, bias=bias, n_store=Cp)])
                    out[name] = sc[:, :C]
                return out

            xt = eng.taps["seq_t"]
            scores_t = head(xt, B * T, H, "cls.predictions.", True,
                            {"t": (W("bert.embeddings.word_embeddings.weight"), Pm("cls.predictions.bias"), cfg.vocab_size)})["t"].reshape(B, T, -1)
            widths = {"0": 1601, "1": 2048, "2": 2048, "3": 1600, "4": 400, "5": 2048, "6": 1601}
            decs = {ix: (W("cls.imagePredictions.decoder_dict.%s.weight" % ix), Pm("cls.imagePredictions.decoder_dict.%s.bias" % ix), widths[ix])
                    for ix, w in cfg.visual_target_weights.items() if w > 0}
            sv = head(eng.taps["seq_v"], B * Rv, Hv, "cls.imagePredictions.", bool(cfg.image_head_ln), decs)
            scores_v = {ix: t.reshape(B, Rv, -1) for ix, t in sv.items()}
            fm = self.bert.fusion_method
            pooled = None if fm == "none" else (pt * pv if fm == "mul" else pt + pv if fm == "sum" else pt)
            itm = None
            if fm in ("mul", "sum", "text"):
                P = pooled.shape[1]
                pb = pooled.to(torch.bfloat16).contiguous()
                sc = torch.empty(B, 64, device=dev)
                ops.gemm_grouped(L.NT, L.EPI_F32, [ops.gemm_problem(pb, W("cls.bi_seq_relationship.weight"), sc, L.NT, B, 2, P, bias=Pm("cls.bi_seq_relationship.bias"), n_store=64)])
                itm = sc[:, :2]
            return scores_t, scores_v, itm, attn_maps, pooled

    def encode(self, input_ids, image_feat, image_loc, token_type_ids=None, attention_mask=None, image_attention_mask=None,
               output_all_encoded_layers=False, output_all_attention_masks=False):
        """BertModel.forward: (seq_t [B,T,H], seq_v [B,Rv,H], pooled_t, pooled_v, attention maps = ([], [])); with
        `output_all_encoded_layers` the two sequences are lists with both streams' states after EVERY sub-layer (encoders.py:868-881)."""
        B, T = input_ids.shape
        R = image_feat.shape[1] - self.add_global_imgfeat
        dev = next(self.parameters()).device
        dummy = dict(masked_lm_labels=torch.full((B, T), -1, dtype=torch.int64, device=dev),
                     image_label=torch.full((B, R), -1, dtype=torch.int64, device=dev),
                     image_cls=torch.zeros(B, R, 1601, device=dev), next_sentence_label=torch.zeros(B, dtype=torch.int64, device=dev),
                     obj_labels=torch.zeros(B, R, dtype=torch.int64, device=dev), obj_confs=torch.zeros(B, R, device=dev),
                     attr_labels=torch.zeros(B, R, dtype=torch.int64, device=dev), attr_confs=torch.zeros(B, R, device=dev))
        tensors, B, T, Rv = self._prep_inputs(input_ids, image_feat, image_loc, token_type_ids, attention_mask, image_attention_mask, **dummy)
        self.__dict__["_want_attn_maps"] = bool(output_all_attention_masks and self.config.visualization)
        try:
            with torch.no_grad():
                self._engine_forward(tensors)
        finally:
            self.__dict__["_want_attn_maps"] = False
        eng = self._last[0]
        attn_maps = _attention_maps(self.config, eng, output_all_attention_masks)
        H = self.config.hidden_size
        pt, pv = eng.taps["pooled_t"], eng.taps["pooled_v"]          # None where the fusion method has no such pooler
        Hv = self.config.v_hidden_size
        if output_all_encoded_layers:
            seq_t = [eng.taps["t%d" % n].view(B, T, H).float() for n in eng.sublayer_ids]
            seq_v = [eng.taps["v%d" % n].view(B, Rv, Hv).float() for n in eng.sublayer_ids]
        else:
            seq_t, seq_v = eng.taps["seq_t"].view(B, T, H).float(), eng.taps["seq_v"].view(B, Rv, Hv).float()
        return seq_t, seq_v, None if pt is None else pt.float(), None if pv is None else pv.float(), attn_maps


def _attention_maps(config, eng, requested):
    """all_attention_mask of BertEncoder.forward (volta/encoders.py:858-886): empty lists unless requested; one entry per attention
    sub-layer otherwise -- the dictionaries of encoders.py:342-356 under config.visualization, None without it (:357-358)."""
    if not requested:
        return ([], [])
    if config.visualization:
        return eng.attention_maps()
    n = sum(1 for _, typ in M.sublayer_schedule(config) if typ == "attn")
    return ([None] * n, [None] * n)


# ======================================================================================== downstream tasks
class SimpleClassifier(nn.Module):
    """Linear -> GELU -> LayerNorm -> Linear (volta/encoders.py:787-815): the parameter container with the reference's names; the arithmetic
    runs on the HIP engine (engine.py:_heads_tasks), `forward` is kept for reference use on CPU copies only."""

    def __init__(self, in_dim, hid_dim, out_dim, dropout_prob=0.0):
        super().__init__()
        self.logit_fc = nn.Sequential(nn.Linear(in_dim, hid_dim), nn.GELU(), nn.LayerNorm(hid_dim, eps=1e-12), nn.Linear(hid_dim, out_dim))

    def forward(self, hidden_states):
        return self.logit_fc(hidden_states)


class _TaskStep(torch.autograd.Function):
    """Encoder + poolers + the task's classifier as one autograd node: the prediction leaves the engine as fp32 logits, its gradient
    re-enters it as the seed of the backward list."""

    @staticmethod
    def forward(ctx, model, anchor, tensors):
        model._engine_forward(tensors)
        eng = model._last[0]
        ctx.model = model
        C = eng.pred_shape[-1]
        return eng.pred[:, :C].reshape(eng.pred_shape).clone()

    @staticmethod
    def backward(ctx, g_pred):
        model = ctx.model
        eng, _ = model._last
        state = model._backward_begin(eng)
        C = eng.pred_shape[-1]
        eng.d_pred[:, :C].copy_(g_pred.reshape(-1, C))        # the pad columns stay zero
        model._backward_run(eng, state)
        return None, None, None


class BertForVLTasks(PreTrainedModel):
    """Fine-tuning / evaluation model of the downstream tasks (volta/encoders.py:1117-1206): encoder, poolers, fusion + dropout and the
    task's classifier all run on the HIP engine, forward and backward; `clfs_dict` holds the classifiers' parameters under the reference's
    names and is never called."""

    _heads_mode = "tasks"
    _torch_param_prefixes = ()
    _vk_is_model = True
    materialize = BertForVLPreTraining.materialize
    _engine = BertForVLPreTraining._engine
    _prep_inputs = BertForVLPreTraining._prep_inputs
    _engine_forward = BertForVLPreTraining._engine_forward
    _backward_begin = BertForVLPreTraining._backward_begin
    _backward_run = BertForVLPreTraining._backward_run
    set_dropout_seed = BertForVLPreTraining.set_dropout_seed
    set_projection_dtype = BertForVLPreTraining.set_projection_dtype

    def __init__(self, config, task_cfg, task_ids, dropout_prob=0.1):
        super().__init__(config)
        self.bert = BertModel(config)
        if not 0.0 <= float(dropout_prob) < 1.0:
            raise ValueError("dropout probability has to be between 0 and 1, but got {}".format(dropout_prob))
        self.__dict__["_task_dropout"] = float(dropout_prob)      # nn.Dropout(dropout_prob) of the reference (encoders.py:1122): an engine op here
        self.task_cfg = task_cfg
        task2clf = {}
        for task_id in task_ids:
            task_type = task_cfg[task_id]["type"]
            if task_type in {"VL-classifier", "VL-classifier-GQA"}:
                task2clf[task_id] = SimpleClassifier(config.pooler_size, config.clf_hidden_size, task_cfg[task_id]["num_labels"])
            elif task_type == "VL-binary-classifier":
                task2clf[task_id] = SimpleClassifier(config.pooler_size * 2, config.clf_hidden_size, 2)
            elif task_type == "VL-tri-classifier":
                task2clf[task_id] = nn.Linear(config.pooler_size, 3)
            elif task_type == "VL-logit":
                task2clf[task_id] = nn.Linear(config.pooler_size, 1)
            elif task_type.startswith("V-logit"):
                if task_cfg[task_id].get("num_clf_layers", 1) == 2:
                    task2clf[task_id] = nn.Sequential(nn.Linear(config.v_hidden_size, config.v_hidden_size), nn.GELU(),
                                                      nn.Dropout(config.v_attention_probs_dropout_prob, inplace=False),
                                                      nn.Linear(config.v_hidden_size, 1))
                else:
                    task2clf[task_id] = nn.Linear(config.v_hidden_size, 1)
            else:
                raise ValueError("Undefined task type: %s" % task_type)
        self.clfs_dict = nn.ModuleDict(task2clf)
        self.fusion_method = config.fusion_method
        for mod in self.clfs_dict.modules():                     # the outer apply(init_weights) of the reference (utils.py:377-389)
            if isinstance(mod, nn.Linear):
                mod.weight.data.normal_(mean=0.0, std=config.initializer_range)
                if mod.bias is not None:
                    mod.bias.data.zero_()
            elif isinstance(mod, nn.LayerNorm):
                mod.bias.data.zero_()
                mod.weight.data.fill_(1.0)
        self.add_global_imgfeat = int(config.add_global_imgfeat is not None)
        self.__dict__["_arena"] = None
        self.__dict__["_engines"] = {}
        self.__dict__["_step"] = 0
        self.__dict__["_seed_base"] = None
        self.__dict__["_last"] = None
        self.__dict__["_ddp"] = None
        self.bert.__dict__["_root"] = self
        for mod in self.modules():
            if mod is not self and isinstance(mod, PreTrainedModel):
                mod.__dict__["_root"] = self
        for n, p in self.named_parameters():
            p._vk_owner = self

    def encode(self, input_ids, image_feat, image_loc, token_type_ids=None, attention_mask=None, image_attention_mask=None,
               output_all_encoded_layers=False, output_all_attention_masks=False):
        """BertModel.forward under no_grad (BertModel.forward delegates here)."""
        tensors, B, T, Rv = self._prep_inputs(input_ids, image_feat, image_loc, token_type_ids, attention_mask, image_attention_mask,
                                              None, None, None, None)
        self.__dict__["_want_attn_maps"] = bool(output_all_attention_masks and self.config.visualization)
        try:
            with torch.no_grad():
                self._engine_forward(tensors)
        finally:
            self.__dict__["_want_attn_maps"] = False
        eng = self._last[0]
        attn_maps = _attention_maps(self.config, eng, output_all_attention_masks)
        H = self.config.hidden_size
        pt, pv = eng.taps["pooled_t"], eng.taps["pooled_v"]          # None where the fusion method has no such pooler
        Hv = self.config.v_hidden_size
        if output_all_encoded_layers:
            seq_t = [eng.taps["t%d" % n].view(B, T, H).float() for n in eng.sublayer_ids]
            seq_v = [eng.taps["v%d" % n].view(B, Rv, Hv).float() for n in eng.sublayer_ids]
        else:
            seq_t, seq_v = eng.taps["seq_t"].view(B, T, H).float(), eng.taps["seq_v"].view(B, Rv, Hv).float()
        return seq_t, seq_v, None if pt is None else pt.float(), None if pv is None else pv.float(), attn_maps

    def forward(self, input_txt, input_imgs, image_loc, task_id, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, output_all_encoded_layers=False, output_all_attention_masks=False):
        if task_id not in self.task_cfg or task_id not in self.clfs_dict:
            raise KeyError("unknown task id %r" % (task_id,))
        tensors, B, T, Rv = self._prep_inputs(input_txt, input_imgs, image_loc, token_type_ids, attention_mask, image_attention_mask,
                                              None, None, None, None)
        self.__dict__["_cur_task"] = task_id                    # selects the plan whose head is this task's classifier
        self.__dict__["_want_attn_maps"] = bool(output_all_attention_masks and self.config.visualization)
        try:
            self.materialize()
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                anchor = next(p for p in self.parameters() if p.requires_grad)
                vil_prediction = _TaskStep.apply(self, anchor, tensors)
            else:
                with torch.no_grad():
                    self._engine_forward(tensors)
                eng = self._last[0]
                C = eng.pred_shape[-1]
                vil_prediction = eng.pred[:, :C].reshape(eng.pred_shape).clone()
        finally:
            self.__dict__["_cur_task"] = None
            self.__dict__["_want_attn_maps"] = False
        attn_maps = _attention_maps(self.config, self._last[0], output_all_attention_masks)
        if self.task_cfg[task_id]["type"].startswith("V-logit"):   # padded regions are masked out of the region scores (encoders.py:1198-1199)
            mask = tensors["image_attention_mask"].to(vil_prediction.dtype)
            vil_prediction = vil_prediction + ((1.0 - mask) * -10000.0).unsqueeze(2)
        return vil_prediction, None, None, attn_maps
