"""Import alias for the package directory ``s2vt-video-caption_amd/``.

The directory name is fixed by the build contract and is not a valid Python
identifier, so this module turns itself into a package whose ``__path__`` is
that directory: ``import s2vt_video_caption_amd.capi`` etc. then resolve to
``s2vt-video-caption_amd/capi.py``.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "s2vt-video-caption_amd")]

with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f, _os
