"""Parser for the reference's config DSL (ae_configs/*, pc_configs/*).

The reference parses these files with ``fjcommon.config_parser.parse`` (un-vendored
dependency fjcommon==0.1.69; call sites code/train.py:65-66, code/val.py:71-72), which
returns ``(config_object, rel_path)``.  The DSL, as used by the reference's config files
(code/ae_configs/base:1-43, code/pc_configs/base:1-26):

    use <relative path>           inherit every key of another config file first
    constrain <key> :: A, B, C    declare the legal values of an enum-like key; the bare
                                  words A, B, C become usable as values ( x = A -> 'A')
    <key> = <python expression>   evaluated with earlier keys and enum words in scope
    # comment / blank lines

Unknown keys raise AttributeError on access, a value outside its ``constrain`` set raises
ValueError at parse time -- the same failure behaviour the reference relies on.
"""
import os


class ConfigError(ValueError):
    pass


class Config(object):
    """Attribute-style view of the parsed key/value pairs."""

    def __init__(self, values, constraints, path):
        object.__setattr__(self, '_values', dict(values))
        object.__setattr__(self, '_constraints', dict(constraints))
        object.__setattr__(self, '_path', path)

    def __getattr__(self, key):
        try:
            return self._values[key]
        except KeyError:
            raise AttributeError('config {} has no parameter {!r}'.format(self._path, key))

    def __setattr__(self, key, value):
        self._check(key, value)
        self._values[key] = value

    def _check(self, key, value):
        allowed = self._constraints.get(key)
        if allowed is not None and value not in allowed:
            raise ConfigError('{}: {} = {!r} violates constraint {}'.format(self._path, key, value, allowed))

    def all_params_and_values(self):
        return sorted(self._values.items())

    def as_dict(self):
        return dict(self._values)

    def __contains__(self, key):
        return key in self._values

    def __str__(self):
        return '\n'.join('{} = {!r}'.format(k, v) for k, v in self.all_params_and_values())


def _parse_into(path, values, constraints, enum_words, seen):
    path = os.path.abspath(path)
    if path in seen:
        raise ConfigError('cyclic `use` involving {}'.format(path))
    seen = seen | {path}
    if not os.path.isfile(path):
        raise FileNotFoundError('config file not found: {}'.format(path))
    with open(path) as f:
        lines = f.readlines()
    for lineno, raw in enumerate(lines, 1):
        line = raw.split('#', 1)[0].strip()
        if not line:
            continue
        where = '{}:{}'.format(path, lineno)
        if line.startswith('use '):
            target = os.path.join(os.path.dirname(path), line[4:].strip())
            _parse_into(target, values, constraints, enum_words, seen)
        elif line.startswith('constrain '):
            try:
                key, allowed = line[len('constrain '):].split('::')
            except ValueError:
                raise ConfigError('{}: expected `constrain key :: A, B`'.format(where))
            words = tuple(w.strip() for w in allowed.split(',') if w.strip())
            constraints[key.strip()] = words
            for w in words:
                enum_words[w] = w
        elif '=' in line:
            key, expr = line.split('=', 1)
            key = key.strip()
            if not key.isidentifier():
                raise ConfigError('{}: invalid key {!r}'.format(where, key))
            scope = dict(enum_words)
            scope.update(values)
            try:
                value = eval(expr.strip(), {'__builtins__': {}}, scope)
            except Exception as e:
                raise ConfigError('{}: cannot evaluate {!r}: {}'.format(where, expr.strip(), e))
            allowed = constraints.get(key)
            if allowed is not None and value not in allowed:
                raise ConfigError('{}: {} = {!r} not in {}'.format(where, key, value, allowed))
            values[key] = value
        else:
            raise ConfigError('{}: cannot parse line {!r}'.format(where, raw.rstrip()))


def _rel_path(path):
    """Path of the config relative to the directory that holds the `*_configs` root,
    e.g. '.../code/ae_configs/cvpr/low' -> 'ae_configs/cvpr/low' (used in log-dir names)."""
    parts = os.path.abspath(path).split(os.sep)
    for i in range(len(parts) - 1, -1, -1):
        if parts[i].endswith('_configs'):
            return '/'.join(parts[i:])
    return os.path.basename(path)


def parse(config_path):
    """-> (Config, rel_path), the signature of fjcommon.config_parser.parse."""
    values, constraints, enum_words = {}, {}, {}
    _parse_into(config_path, values, constraints, enum_words, frozenset())
    return Config(values, constraints, config_path), _rel_path(config_path)


def builtin_config_path(*rel):
    """Path to a config shipped with this package, e.g. ('ae_configs', 'cvpr', 'low')."""
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), *rel)
