import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
from universal_recommender_amd import _lib, synth
cfg = synth.config3(1.0); data = synth.generate(cfg)
lib = _lib.load(_lib.DEFAULT_PATH)
print(bench.host_level_leg(lib, data, cfg, 1, 0)["ms"])
