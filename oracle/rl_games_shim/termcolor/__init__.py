"""termcolor stub (utils/logger.py imports it at module level; nothing is printed in the oracle)."""


def colored(text, *a, **kw):
    return text
