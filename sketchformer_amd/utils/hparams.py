"""Typed hyper-parameter container with the behaviour of the reference's flag system
(utils/hparams.py: HParams :347-697, parse_values :232-343, combine_hparams_into_one :49-55,
save_config / load_config :58-84).  Independent implementation; behaviours are pinned by
tests/golden/reference_goldens.json ("hparams_parse", "hparams_combine", "hparams_to_json_sorted").

    hps = HParams(num_layers=4, dropout_rate=0.1, do_classification=True)
    hps.parse("num_layers=6,do_classification=false")   # unknown key / bad type -> ValueError
"""
import json
import re

import numpy as np

_CLAUSE = re.compile(r"""\s*(?P<name>[a-zA-Z][\w\.]*)\s*(\[\s*(?P<index>\d+)\s*\])?\s*=\s*
                         (?:\[(?P<vals>[^\]]*)\]|(?P<val>[^,\[]*))(?:$|,)""", re.VERBOSE)


def _to_bool(text):
    t = text.strip()
    if t in ("true", "True"):
        return True
    if t in ("false", "False"):
        return False
    try:
        return bool(int(t))
    except ValueError:
        raise ValueError("could not parse %r as a bool" % text)


def _caster(kind):
    if kind is bool:
        return _to_bool
    if kind is int:
        return lambda t: int(t)
    if kind is float:
        return lambda t: float(t)
    return lambda t: t


def _kind_of(value):
    if isinstance(value, bool):
        return bool
    if isinstance(value, (int, np.integer)):
        return int
    if isinstance(value, (float, np.floating)):
        return float
    if isinstance(value, str):
        return str
    raise ValueError("unsupported hparam type %s" % type(value))


class HParams(object):
    def __init__(self, **kwargs):
        self._kinds = {}     # name -> (scalar type, is_list)
        for name, value in kwargs.items():
            self.add_hparam(name, value)

    # ---- definition
    def add_hparam(self, name, value):
        if name in self._kinds or getattr(self, name, None) is not None and not name.startswith("_") and name in self.__dict__:
            raise ValueError("Hyperparameter name is reserved: %s" % name)
        if isinstance(value, (list, tuple)):
            if not value:
                raise ValueError("Multi-valued hyperparameters cannot be empty: %s" % name)
            self._kinds[name] = (_kind_of(value[0]), True)
            value = list(value)
        else:
            self._kinds[name] = (_kind_of(value), False)
        setattr(self, name, value)

    def _coerce(self, name, value):
        kind, is_list = self._kinds[name]
        if is_list != isinstance(value, (list, tuple)):
            raise ValueError("Must %spass a list for hyperparameter: %s" % ("" if is_list else "not ", name))

        def one(v):
            if kind is bool:
                if isinstance(v, bool):
                    return v
            elif kind is int:
                if isinstance(v, (int, np.integer)) and not isinstance(v, bool):
                    return int(v)
            elif kind is float:
                if isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool):
                    return float(v)
            elif isinstance(v, str):
                return v
            raise ValueError("Could not cast hparam '%s' of type '%s' from value %r" % (name, kind.__name__, v))
        return [one(v) for v in value] if is_list else one(value)

    def set_hparam(self, name, value):
        if name not in self._kinds:
            raise ValueError("Unknown hyperparameter: %s" % name)
        setattr(self, name, self._coerce(name, value))

    def del_hparam(self, name):
        if name in self._kinds:
            delattr(self, name)
            del self._kinds[name]

    # ---- parsing "a=1,b=[2,3],c[0]=4"
    def parse(self, values):
        pos, seen = 0, set()
        text = values
        while pos < len(text):
            m = _CLAUSE.match(text, pos)
            if not m:
                raise ValueError("Malformed hyperparameter value: %s" % text[pos:])
            pos = m.end()
            name = m.group("name")
            if name not in self._kinds:
                raise ValueError("Unknown hyperparameter type for %s" % name)
            kind, is_list = self._kinds[name]
            cast = _caster(kind)
            try:
                if m.group("vals") is not None:
                    if not is_list:
                        raise ValueError("list value for scalar hyperparameter %s" % name)
                    parsed = [cast(v) for v in re.split(r"[ ,]", m.group("vals")) if v]
                    if name in seen:
                        raise ValueError("Multiple assignments to variable %r" % name)
                    seen.add(name)
                    setattr(self, name, parsed)
                elif m.group("index") is not None:
                    if not is_list:
                        raise ValueError("index on scalar hyperparameter %s" % name)
                    cur = list(getattr(self, name))
                    idx = int(m.group("index"))
                    while len(cur) <= idx:
                        cur.append(cur[-1])
                    cur[idx] = cast(m.group("val"))
                    setattr(self, name, cur)
                else:
                    if is_list:
                        raise ValueError("scalar value for list hyperparameter %s" % name)
                    if name in seen:
                        raise ValueError("Multiple assignments to variable %r" % name)
                    seen.add(name)
                    setattr(self, name, cast(m.group("val")))
            except ValueError as e:
                raise ValueError("Could not parse hparam '%s' of type '%s' with value %r (%s)"
                                 % (name, kind.__name__, m.group("val") or m.group("vals"), e))
        return self

    def override_from_dict(self, values_dict):
        for name, value in values_dict.items():
            self.set_hparam(name, value)
        return self

    # ---- (de)serialisation
    def values(self):
        return {n: getattr(self, n) for n in self._kinds}

    def get(self, key, default=None):
        return getattr(self, key) if key in self._kinds else default

    def to_json(self, indent=None, separators=None, sort_keys=False):
        return json.dumps(self.values(), indent=indent, separators=separators, sort_keys=sort_keys)

    def parse_json(self, values_json):
        return self.override_from_dict(json.loads(values_json))

    def __contains__(self, key):
        return key in self._kinds

    def __str__(self):
        return str(sorted(self.values().items()))

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, self.__str__())


def copy_hparams(hparams):
    return HParams(**hparams.values())


def combine_hparams_into_one(*args):
    combined = args[0].values()
    for more in args[1:]:
        combined.update(more.values())
    return HParams(**combined)


def save_config(output_file, hps, verbose=True):
    def convert(o):
        if isinstance(o, np.integer):
            return int(o)
        raise TypeError
    if verbose:
        for key, val in hps.values().items():
            print("%s = %s" % (key, str(val)))
    with open(output_file, "w") as f:
        json.dump(hps.values(), f, indent=True, default=convert)


def load_config(hps, config_file, verbose=True):
    try:
        with open(config_file, "r") as fin:
            hps.parse_json(fin.read())
        if verbose:
            for key, val in hps.values().items():
                print("%s = %s" % (key, str(val)))
    except Exception as e:  # noqa: BLE001  (the reference reports and carries on)
        print("Error reading config file %s: %s.\nConfig will not be updated." % (config_file, e))
