#!/bin/bash
# round 5: LLVM scheduling strategies for the whole library, one box: max-ilp, AMDGPU register-pressure trackers (both spill in some row kernels)
STEPS=20 tools/lib_ab.sh r05_sched_strategy_ab 2 tools/_variants/maxilp.so tools/_variants/trackers.so
