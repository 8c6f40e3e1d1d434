#!/bin/bash
# round 5: B' column gathers in flight per lane in the pair loops, one box: one-wave class 4 / 8 (tree: 2), 256-thread classes 2 / 4 (tree: 1), 512/1024-thread classes 2 (tree: 1)
STEPS=20 tools/lib_ab.sh r05_gather_depth_ab 2 tools/_variants/gw4.so tools/_variants/gw8.so tools/_variants/gb2.so tools/_variants/gb4.so tools/_variants/gc2.so
