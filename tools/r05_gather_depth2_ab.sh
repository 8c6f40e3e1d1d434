#!/bin/bash
# round 5: deeper column-gather batches after two were adopted everywhere: 512/1024-thread classes 4 / 8, 256-thread classes 3, one box
STEPS=20 tools/lib_ab.sh r05_gather_depth2_ab 2 tools/_variants/gc4.so tools/_variants/gc8.so tools/_variants/gb3.so
