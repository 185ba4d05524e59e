#!/bin/bash
# developer (round 6): the tree's library against tools/exp/ab/libdbfr_base.so (an older revision's api.cpp / conv.hip / graph.hip) -- same poses bit for bit? seconds per call?
cd $GRAFT_REPO_ROOT
for r in 1 2; do
python tools/exp/pose_hash.py cfg1 bs16 cfg1x40 c5p16 p160
DBFR_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/libdbfr_base.so python tools/exp/pose_hash.py cfg1 bs16 cfg1x40 c5p16 p160
done
python -m pytest tests -m gpu -q -x -k "chunk or bitwise or batch_independ or native_library" 2>&1 | tail -3
bash tools/exp/r6_small.sh cfg1 2>&1 | grep -v "^$"
