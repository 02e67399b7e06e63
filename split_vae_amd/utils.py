"""vae/utils.py:3-7 -- dotdict: attribute access, missing keys read as None."""


class dotdict(dict):
    """dot.notation access to dictionary attributes (missing key -> None, like dict.get)"""
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__
