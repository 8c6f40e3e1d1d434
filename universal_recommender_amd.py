"""Import alias: `import universal_recommender_amd` -> the package in ./universal-recommender_amd/
(the directory name required by the project layout contains a hyphen, which is not importable)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "universal-recommender_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
