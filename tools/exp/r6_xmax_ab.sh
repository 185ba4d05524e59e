cd $GRAFT_REPO_ROOT
python tools/exp/pose_hash.py cfg1 bs16 cfg1x40 c5p16
echo "== tree"; bash tools/exp/r6_small.sh bs16 cfg1 2>&1 | grep "poses/s\|absmax\|span"
echo "== base"; DBFR_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/libdbfr_base.so bash tools/exp/r6_small.sh bs16 cfg1 2>&1 | grep "poses/s\|absmax\|span"
cd $GRAFT_REPO_ROOT; bash tools/exp/bench_ab.sh 1 base tree
