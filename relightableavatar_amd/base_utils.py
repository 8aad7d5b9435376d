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
