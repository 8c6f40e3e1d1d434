#!/bin/bash
# round 5: knobs around the single ranking of the select's finish, one box: waits = the select goes on until bin + above fit the set (URCCO_MERGE_WAITS=1),
# selm192 = 192-entry sets for the teams of several waves, both, amb32w = the one-wave class finishes at <= 32 members of the cut bin
STEPS=20 tools/lib_ab.sh r05_merge_knobs_ab 2 tools/_variants/waits.so tools/_variants/selm192.so tools/_variants/both.so tools/_variants/amb32w.so
