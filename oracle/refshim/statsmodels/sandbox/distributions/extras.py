"""See refshim/statsmodels/__init__.py: the one name the reference touches, never called by the goldens."""


def pdf_mvsk(*args, **kwargs):
    raise NotImplementedError("statsmodels is not installed; refshim only makes Visualization/utils.py importable")
