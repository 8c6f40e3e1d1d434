#!/bin/bash
# round 5: knobs of the one-wave class after the instruction diet, one box: ambiguous-set limit 32 / 16 (tree: 64), two column gathers in flight
# per lane, shared key bytes tracked while scoring, four count gathers per lane in flight (the last two spill 4-5 registers)
STEPS=20 tools/lib_ab.sh r05_wave_knobs_ab 2 tools/_variants/amb32.so tools/_variants/amb16.so tools/_variants/g2.so tools/_variants/skipw.so tools/_variants/u4.so
