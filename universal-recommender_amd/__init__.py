"""universal-recommender_amd: MI355X-native Correlated Cross-Occurrence model build for the Universal Recommender.

Import name: `universal_recommender_amd` (see universal_recommender_amd.py at the repo root: the directory name
carries a hyphen, which Python cannot import directly).
"""
__version__ = "0.1.0"
