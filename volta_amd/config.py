"""Host-side mirror of the reference's model configuration object (volta/config.py:11-181).

Same constructor forms, attribute names and defaults: a flat attribute bag filled with the defaults and
then overwritten by the JSON keys, so that `config/ctrl_*.json` files load unchanged and keys absent
from a file keep their default (objective=0, image_head_ln=True, model="bert", fixed_layers=[], ...)."""
import copy
import json

_DEFAULTS = dict(
    hidden_size=768, num_attention_heads=12, intermediate_size=3072, pooler_size=768, hidden_act="gelu",
    hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
    num_locs=5, v_coordinate_embeddings_dim=None, add_global_imgfeat=None, image_embeddings="vilbert",
    initializer_range=0.02, v_feature_size=2048, v_hidden_size=768, v_num_attention_heads=12,
    v_intermediate_size=3072, v_pooler_size=1024, v_attention_probs_dropout_prob=0.1, v_hidden_act="gelu",
    v_hidden_dropout_prob=0.1, v_initializer_range=0.2, visual_target_weights={"0": 1}, fixed_layers=[],
    fusion_method="mul", objective=0, clf_hidden_size=1536, image_head_ln=True, model="bert", visualization=False,
    tt_attn_sublayers=[], tv_attn_sublayers=[], vt_attn_sublayers=[], vv_attn_sublayers=[], t_ff_sublayers=[],
    v_ff_sublayers=[], shared_sublayers=[], single_ln_sublayers=[], sublayer2attn_hidden_size={},
    sublayer2num_attention_heads={}, sublayer2intermediate_size={}, sublayer2v_attn_hidden_size={},
    sublayer2v_num_attention_heads={}, sublayer2v_intermediate_size={}, bert_layer2attn_sublayer={},
    bert_layer2ff_sublayer={},
)


class BertConfig(object):
    def __init__(self, vocab_size_or_config_json_file, **kwargs):
        if isinstance(vocab_size_or_config_json_file, str):
            with open(vocab_size_or_config_json_file, "r", encoding="utf-8") as reader:
                for key, value in json.loads(reader.read()).items():
                    self.__dict__[key] = value
        elif isinstance(vocab_size_or_config_json_file, int):
            unknown = set(kwargs) - set(_DEFAULTS)
            if unknown:
                raise TypeError("unexpected BertConfig arguments: %s" % sorted(unknown))
            self.vocab_size = vocab_size_or_config_json_file
            for key, value in _DEFAULTS.items():
                self.__dict__[key] = copy.deepcopy(kwargs.get(key, value))
        else:
            raise ValueError("First argument must be either a vocabulary size (int)"
                             "or the path to a pretrained model config file (str)")

    @classmethod
    def from_dict(cls, json_object):
        config = BertConfig(vocab_size_or_config_json_file=-1)
        for key, value in json_object.items():
            config.__dict__[key] = value
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as reader:
            text = reader.read()
        return cls.from_dict(json.loads(text))

    def __repr__(self):
        return str(self.to_json_string())

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"
