class MuscleCommandline(object):     # imported by metamlst-merge.py:26; used only by --outseqformat (out of scope)
    def __init__(self, *a, **k):
        raise RuntimeError("MUSCLE is not part of the pinned path")
