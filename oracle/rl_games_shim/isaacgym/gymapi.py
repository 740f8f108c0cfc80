"""Stub of an Isaac Gym module: the reference's task files import it at module level; nothing of it runs in the oracle
(only the jit-scripted observation functions of those files are called).  Any attribute resolves to a placeholder."""


class _Anything:
    def __init__(self, name='isaacgym'):
        self._name = name

    def __getattr__(self, k):
        return _Anything(self._name + '.' + k)

    def __call__(self, *a, **kw):
        return _Anything(self._name + '()')


def __getattr__(name):
    return _Anything(name)
