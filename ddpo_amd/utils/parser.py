"""Config reader with command-line overrides — the flag surface of `pipeline/policy_gradient.py`.

Behavioural mirror of /root/reference/ddpo/utils/parser.py:71-214 without the `tap` dependency (not installable here):
  precedence  base[experiment] < <dataset>["common"] < <dataset>[experiment] < `--key value` pairs on the CLI;
  dataset names treat '-' and '_' alike (:94); unknown keys are an error (:135-137); overrides are cast to the type of
  the value they replace, "None" -> None, bool / None-typed values are eval'ed (:141-153); strings starting with "f:"
  are lazy f-strings over the final arguments (:157-164); `savepath` (and relative loadpath / modelpath) are joined
  under `logbase` (:196-214); the seed is offset by the process index (:174-179) — here the data-parallel rank.
"""
import ast
import importlib
import os
import random
import sys

import numpy as np
import torch


class Args:
    """Attribute bag; `_dict` holds exactly the config-derived keys (what the reference dumps to args.json)."""

    def __init__(self):
        object.__setattr__(self, "_dict", {})

    def set(self, key, value, config_key=True):
        object.__setattr__(self, key, value)
        if config_key:
            self._dict[key] = value

    def __setattr__(self, key, value):
        self.set(key, value, config_key=key in self._dict)

    def __contains__(self, key):
        return hasattr(self, key)


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _cast(raw, old):
    if raw == "None":
        return None
    if raw == "latest":
        return "latest"
    if isinstance(old, bool) or old is None:
        try:
            return ast.literal_eval(raw)
        except (ValueError, SyntaxError):
            print(f"[ utils/setup ] Warning: could not parse {raw} (old: {old}, {type(old)}), using str")
            return raw
    if isinstance(old, (dict, list, tuple)):
        return ast.literal_eval(raw)
    return type(old)(raw)


class Parser:
    """Subclass and add class attributes `config` / `dataset` for defaults (as the reference's entrypoint does)."""

    config: str = "config.base"
    dataset: str = "consistent_imagenet"

    def __init__(self, argv=None):
        self._argv = list(sys.argv[1:] if argv is None else argv)

    def _split_argv(self):
        known = {"config": self.config, "dataset": self.dataset}
        extras = []
        i = 0
        while i < len(self._argv):
            tok = self._argv[i]
            name = tok[2:] if tok.startswith("--") else None
            if name in known and i + 1 < len(self._argv):
                known[name] = self._argv[i + 1]
                i += 2
            else:
                extras.append(tok)
                i += 1
        return known, extras

    def parse_args(self, experiment=None, process_index=0):
        known, extras = self._split_argv()
        args = Args()
        args.set("config", known["config"], config_key=False)
        args.set("dataset", known["dataset"], config_key=False)
        args.set("extra_args", extras, config_key=False)
        self.read_config(args, experiment)
        self.add_extras(args)
        self.eval_fstrings(args)
        self.mkdir(args)
        self.set_seed(args, process_index)
        self.report(args)
        return args

    def read_config(self, args, experiment):
        dataset = args.dataset.replace("-", "_")
        print(f"[ utils/parser ] Reading config: {args.config}:{dataset}:{experiment}")
        module = importlib.import_module(args.config)
        params = dict(getattr(module, "base")[experiment])
        if hasattr(module, dataset):
            print(f"[ utils/parser ] Using overrides | config: {args.config} | dataset: {dataset}")
            overrides = getattr(module, dataset)
            params.update(overrides.get("common", {}))
            params.update(overrides.get(experiment, {}))
        else:
            print(f"[ utils/parser ] Not using overrides | config: {args.config} | dataset: {dataset}")
        for key, val in params.items():
            args.set(key, val)

    def add_extras(self, args):
        extras = args.extra_args
        if not extras:
            return
        print(f"[ utils/setup ] Found extras: {extras}")
        assert len(extras) % 2 == 0, f"Found odd number ({len(extras)}) of extras: {extras}"
        for key, raw in zip(extras[0::2], extras[1::2]):
            key = key.replace("--", "")
            assert hasattr(args, key), f"[ utils/setup ] {key} not found in config: {args.config}"
            old = getattr(args, key)
            new = _cast(raw, old)
            print(f"[ utils/setup ] Overriding config | {key} : {old} --> {new}")
            args.set(key, new)

    def eval_fstrings(self, args):
        for key, old in list(args._dict.items()):
            if isinstance(old, str) and old.startswith("f:"):
                # a real f-string evaluation, like the reference ("f:models/{iteration+1}" needs expressions, not only names)
                ns = {k: getattr(args, k) for k in vars(args) if not k.startswith("_")}
                new = eval("f" + repr(old[2:]), {"__builtins__": {}}, ns)
                print(f"[ utils/setup ] Lazy fstring | {key} : {old} --> {new}")
                args.set(key, new)

    def mkdir(self, args):
        if "logbase" in args and "savepath" in args:
            args.set("savepath", os.path.join(args.logbase, args.savepath))
            if not args.savepath.startswith("gs://"):
                os.makedirs(args.savepath, exist_ok=True)
                print(f"[ utils/setup ] Made savepath: {args.savepath}")
        for key in ("loadpath", "modelpath"):
            if "logbase" in args and key in args:
                val = getattr(args, key)
                if val.startswith("/") or val.startswith("gs://"):
                    continue
                args.set(key, os.path.join(args.logbase, val))

    def set_seed(self, args, process_index=0):
        if "seed" not in args or args.seed is None:
            args.set("seed", int(np.random.randint(0, int(1e6))))
        args.set("seed", args.seed + process_index)
        print(f"[ utils/setup ] Setting seed: {args.seed}")
        set_seed(args.seed)

    def report(self, args):
        lines = [f"[ utils/setup ] Parser [ {args.dataset} ]"] + [f"        {k}: {v}" for k, v in args._dict.items()]
        print("\n".join(lines), "\n")
