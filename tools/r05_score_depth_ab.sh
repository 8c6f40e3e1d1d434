#!/bin/bash
# round 5: candidates scored together per lane (their count gathers travel together), one box: 256/4Ki class 3 / 1 (tree: 2), 256/8Ki class 3 (tree: 2),
# 512-thread class 2 (tree: 1), 1024-thread class 2 (tree: 1)
STEPS=20 tools/lib_ab.sh r05_score_depth_ab 2 tools/_variants/ubs3.so tools/_variants/ubs1.so tools/_variants/ub3.so tools/_variants/uh2.so tools/_variants/uc2.so
