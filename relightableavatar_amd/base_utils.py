"""dotdict: dict with attribute access, the container the reference passes around.

Mirrors the *behaviour* the hot path relies on from lib/utils/base_utils.py:7-67
(item + attribute access on the same storage, nested plain dicts are left alone).
"""


class dotdict(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        try:
            del self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def copy(self):
        return dotdict(super().copy())


class lazydict(dotdict):
    """dotdict whose entries may be thunks, evaluated (once) when first read.  The renderers use it for outputs whose SHAPE is
    data dependent (the per-hit arrays `raw`, `volume_albedo`, `volume_roughness` of render_human): producing them needs the hit
    count on the host, i.e. a device synchronisation that the frame loop must not pay unless somebody reads them."""

    class _Thunk:
        __slots__ = ('fn', 'deps')

        def __init__(self, fn, deps=()):
            self.fn, self.deps = fn, tuple(deps)

    def lazy(self, key, fn, deps=()):
        """deps: the tensors `fn` closes over — whoever hands the dict to another stream (pipeline.Pending.result) must record them on it"""
        dict.__setitem__(self, key, lazydict._Thunk(fn, deps))

    def pending_tensors(self):
        """tensors captured by entries that have not been evaluated yet"""
        out = []
        for v in dict.values(self):
            if isinstance(v, lazydict._Thunk):
                out.extend(v.deps)
        return out

    def __iter__(self):
        return iter(list(dict.keys(self)))

    def __reduce__(self):          # pickling / copy.copy evaluate: a raw _Thunk must never leave the dict
        return (dotdict, (dict(self.items()),))

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if isinstance(v, lazydict._Thunk):
            v = v.fn()
            dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in dict.keys(self)]

    def values(self):
        return [self[k] for k in dict.keys(self)]

    def copy(self):
        c = lazydict()
        for k in dict.keys(self):
            dict.__setitem__(c, k, dict.__getitem__(self, k))
        return c
