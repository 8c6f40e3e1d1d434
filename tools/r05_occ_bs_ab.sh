#!/bin/bash
# round 5: the 256-thread / 4Ki class compiled for 8 blocks per CU (64 registers, 7 spilled) against 7 (71 registers), one box
STEPS=20 tools/lib_ab.sh r05_occ_bs_ab 2 tools/_variants/occbs8.so
