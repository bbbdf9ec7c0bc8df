"""The package's one exception type (mirrors xmhw/exception.py:18-19)."""


class XmhwException(Exception):
    pass
