"""Minimal, dependency-free stand-ins for the Hugging Face base classes the reference builds on
(transformers 4.18: PretrainedConfig, ModelOutput, PreTrainedModel).  The GPU box ships transformers 5.x, whose
versions of these classes changed behaviour, and the hot path must not depend on it.
Only the behaviours the reference's callers rely on are provided:

* ModelOutput: ordered mapping + attribute access, ``out["k"]``, ``"k" in out`` (None fields are absent, as in
  HF), ``out["k"] = v`` (model/egtr.py:329,333), integer indexing / ``to_tuple()``.
* PretrainedConfig: open attribute bag, ``from_pretrained(dir_or_json)`` / ``save_pretrained(dir)`` /
  ``to_dict()``, ``use_return_dict``, ``num_labels`` <-> ``id2label``.
* PreTrainedModel: ``from_pretrained(dir, config=, ignore_mismatched_sizes=, output_loading_info=, **kwargs)``
  from a local directory or file (``pytorch_model.bin`` / ``model.safetensors`` / ``*.pt``),
  ``save_pretrained(dir)``, ``post_init()`` applying ``_init_weights``, ``.device``.
"""
import copy
import json
import os
import warnings
from collections import OrderedDict
from dataclasses import fields, is_dataclass

import torch
from torch import nn


class ModelOutput(OrderedDict):
    def __post_init__(self):
        if not is_dataclass(self):
            return
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                OrderedDict.__setitem__(self, f.name, v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return OrderedDict.__getitem__(self, k)
        return self.to_tuple()[k]

    def __setattr__(self, name, value):
        if name in self.keys() and value is not None:
            OrderedDict.__setitem__(self, name, value)
        super().__setattr__(name, value)

    def __setitem__(self, key, value):
        if value is None:
            # HF semantics: a None field is not part of the mapping
            if key in self.keys():
                OrderedDict.__delitem__(self, key)
        else:
            OrderedDict.__setitem__(self, key, value)
        object.__setattr__(self, key, value)

    def to_tuple(self):
        return tuple(OrderedDict.__getitem__(self, k) for k in self.keys())

    def __reduce__(self):
        return (dict, (dict(self),))


class PretrainedConfig:
    model_type = ""
    attribute_map = {}

    def __init__(self, **kwargs):
        self.return_dict = kwargs.pop("return_dict", True)
        self.output_hidden_states = kwargs.pop("output_hidden_states", False)
        self.output_attentions = kwargs.pop("output_attentions", False)
        self.is_encoder_decoder = kwargs.pop("is_encoder_decoder", False)
        id2label = kwargs.pop("id2label", None)
        num_labels = kwargs.pop("num_labels", None)
        self.id2label = {int(k): v for k, v in id2label.items()} if id2label is not None else None
        if num_labels is not None:
            self.num_labels = num_labels
        elif self.id2label is None:
            self.num_labels = 2
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def use_return_dict(self):
        return self.return_dict

    @property
    def num_labels(self):
        return len(self.id2label)

    @num_labels.setter
    def num_labels(self, n):
        if getattr(self, "id2label", None) is None or len(self.id2label) != n:
            self.id2label = {i: f"LABEL_{i}" for i in range(int(n))}
            self.label2id = {v: k for k, v in self.id2label.items()}

    def to_dict(self):
        d = {k: v for k, v in copy.deepcopy(self.__dict__).items()}
        d["model_type"] = self.model_type
        if d.get("id2label") is not None:
            d["id2label"] = {str(k): v for k, v in d["id2label"].items()}
        return d

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True, default=str) + "\n"

    def save_pretrained(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            f.write(self.to_json_string())

    @classmethod
    def from_dict(cls, d, **kwargs):
        d = dict(d)
        d.pop("model_type", None)
        d.pop("label2id", None)
        d.update(kwargs)
        return cls(**d)

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        """Local directory (containing config.json) or a json file.  There is no hub access in this build."""
        f = os.path.join(path, "config.json") if os.path.isdir(path) else path
        if not os.path.isfile(f):
            raise OSError(f"{path}: no local config.json (hub downloads are not available; pass a local path)")
        with open(f) as fh:
            return cls.from_dict(json.load(fh), **kwargs)

    def __repr__(self):
        return f"{self.__class__.__name__} {self.to_json_string()}"


def _load_state_file(path):
    if os.path.isdir(path):
        for name in ("pytorch_model.bin", "model.safetensors", "model.pt"):
            f = os.path.join(path, name)
            if os.path.isfile(f):
                path = f
                break
        else:
            raise OSError(f"{path}: no pytorch_model.bin / model.safetensors found")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    try:  # plain tensor checkpoints (pytorch_model.bin)
        sd = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as err:
        # Lightning .ckpt files pickle hyper-parameter namespaces / callbacks next to the tensors (the reference's
        # evaluate_egtr.py:232-240 loads them with full unpickling).  Full unpickling executes code from the file, so it is
        # limited to that case (a .ckpt path) or an explicit opt-in, and never silent.
        if not (path.endswith(".ckpt") or os.environ.get("EGTR_TRUST_CHECKPOINT_PICKLE") == "1"):
            raise RuntimeError(
                f"{path}: not loadable with weights_only=True ({type(err).__name__}: {err}).  If this is a trusted "
                "pickled checkpoint, set EGTR_TRUST_CHECKPOINT_PICKLE=1 (full unpickling runs code from the file).") from err
        warnings.warn(f"{path}: falling back to full unpickling (weights_only=False); only do this for trusted files")
        sd = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(sd, dict) and "state_dict" in sd and not any(torch.is_tensor(v) for v in sd.values()):
        sd = sd["state_dict"]
    return sd


class PreTrainedModel(nn.Module):
    config_class = PretrainedConfig
    base_model_prefix = ""
    main_input_name = "input_ids"

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        self.config = config

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def _init_weights(self, module):
        pass

    def post_init(self):
        self.apply(self._init_weights)

    def save_pretrained(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        torch.save(self.state_dict(), os.path.join(save_directory, "pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, ignore_mismatched_sizes=False,
                        output_loading_info=False, **kwargs):
        if config is None:
            config = cls.config_class.from_pretrained(pretrained_model_name_or_path)
        model = cls(config, *model_args, **kwargs)
        state_dict = _load_state_file(pretrained_model_name_or_path)
        own = model.state_dict()
        # checkpoints of the bare base model carry keys without the base prefix, and vice versa
        prefix = cls.base_model_prefix + "." if cls.base_model_prefix else ""
        if prefix and not any(k.startswith(prefix) for k in state_dict) and any(k.startswith(prefix) for k in own):
            state_dict = {prefix + k: v for k, v in state_dict.items()}
        mismatched = []
        for k in list(state_dict.keys()):
            if k in own and tuple(own[k].shape) != tuple(state_dict[k].shape):
                if not ignore_mismatched_sizes:
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(state_dict[k].shape)} vs model "
                                       f"{tuple(own[k].shape)} (pass ignore_mismatched_sizes=True)")
                mismatched.append((k, tuple(state_dict[k].shape), tuple(own[k].shape)))
                del state_dict[k]
        res = model.load_state_dict(state_dict, strict=False)
        model.eval()
        # the transformers base class logs these; a silent strict=False load would hand back random weights
        matched = len(own) - len(res.missing_keys)
        if matched == 0:
            raise RuntimeError(
                f"{pretrained_model_name_or_path}: none of the {len(state_dict)} checkpoint keys matches the model "
                f"(first checkpoint keys: {list(state_dict)[:3]}; first model keys: {list(own)[:3]}) -- e.g. a Lightning "
                "checkpoint whose keys carry a 'model.' prefix must be stripped first (evaluate_egtr.py:232-240)")
        if res.missing_keys or res.unexpected_keys or mismatched:
            import warnings
            warnings.warn(
                f"from_pretrained({pretrained_model_name_or_path}): {len(res.missing_keys)} missing (newly initialised), "
                f"{len(res.unexpected_keys)} unexpected, {len(mismatched)} size-mismatched keys; "
                f"missing: {list(res.missing_keys)[:8]} unexpected: {list(res.unexpected_keys)[:8]} "
                f"mismatched: {[m[0] for m in mismatched][:8]}")
        if output_loading_info:
            info = {"missing_keys": list(res.missing_keys) , "unexpected_keys": list(res.unexpected_keys),
                    "mismatched_keys": mismatched, "error_msgs": []}
            return model, info
        return model
